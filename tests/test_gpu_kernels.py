"""Per-kernel parity on the GPU: every HIP kernel, called through the C ABI, against the
same op evaluated by torch on CPU in fp32 (the oracle arithmetic)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------ convolution
CONV_CASES = [
    # B, Cin, Cout, H, k, stride, tile
    (2, 128, 128, 32, 3, 1, 0),
    (1, 128, 128, 16, 3, 1, 1),
    (2, 128, 256, 16, 3, 1, 2),
    (2, 256, 128, 16, 1, 1, 0),
    (1, 64, 96, 16, 3, 1, 2),      # Cout not a multiple of the N tile
    (2, 128, 128, 32, 3, 2, 0),    # Downsample: pad (0,1,0,1), stride 2
    (1, 32, 128, 32, 3, 1, 0),     # conv_in (image staged to 32 channels)
    (2, 128, 3, 32, 3, 1, 0),      # conv_out, tile 128x32
    (8, 512, 512, 8, 3, 1, 0),     # 8x8 level: M tile = one image, split-K
    (2, 512, 512, 16, 3, 1, 0),    # split-K with the 128x128 tile
    (1, 256, 256, 32, 3, 1, 0),    # TW = 32 halo
    (1, 128, 128, 64, 3, 1, 1),
    (2, 256, 512, 8, 1, 1, 0),     # 1x1 on the gather kernel with split-K
]


@pytest.mark.parametrize("B,Cin,Cout,H,k,stride,tile", CONV_CASES)
def test_conv_plain(hip, B, Cin, Cout, H, k, stride, tile):
    from ddnm_amd import ops
    x = gen(B, Cin, H, H, seed=1)
    w = gen(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = gen(Cout, seed=3)
    if stride == 2:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        out = ops.conv2d(nhwc(x).cuda(), ops.pack_conv_weight(w.cuda()), Cout, k, bias=b.cuda(), stride=2, pad=0,
                         out_hw=(H // 2, H // 2), tile=tile)
    else:
        ref = F.conv2d(x, w, b, padding=k // 2)
        out = ops.conv2d(nhwc(x).cuda(), ops.pack_conv_weight(w.cuda()), Cout, k, bias=b.cuda(), tile=tile)
    torch.cuda.synchronize()
    assert rel(nchw(out.cpu()), ref) < 2e-6


def test_conv_fused_everything(hip):
    """GN-affine+swish prologue, concat of two sources, nearest-x2 upsample, temb addend, residual."""
    from ddnm_amd import ops
    B, C0, C1, Cout, Hs = 2, 128, 64, 128, 8
    a, bsrc = gen(B, C0, Hs, Hs, seed=4), gen(B, C1, Hs, Hs, seed=5)
    w = gen(Cout, C0 + C1, 3, 3, seed=6, scale=(9 * (C0 + C1)) ** -0.5)
    bias, badd = gen(Cout, seed=7), gen(B, 300, seed=8)
    scale, shift = gen(B, C0 + C1, seed=9), gen(B, C0 + C1, seed=10)
    res = gen(B, Cout, 2 * Hs, 2 * Hs, seed=11)
    xin = torch.cat([a, bsrc], 1)
    act = xin * scale[:, :, None, None] + shift[:, :, None, None]
    act = act * torch.sigmoid(act)
    act = F.interpolate(act, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(act, w, bias, padding=1) + badd[:, 100:100 + Cout, None, None] + res
    out = ops.conv2d(nhwc(a).cuda(), ops.pack_conv_weight(w.cuda()), Cout, 3, src1=nhwc(bsrc).cuda(), bias=bias.cuda(),
                     badd=badd.cuda()[:, 100:], badd_stride=300, res=nhwc(res).cuda(),
                     gn=(scale.cuda().contiguous(), shift.cuda().contiguous()), gn_silu=True, ups=True)
    torch.cuda.synchronize()
    assert rel(nchw(out.cpu()), ref) < 3e-6


def test_conv_nchw_out(hip):
    from ddnm_amd import ops
    x = gen(2, 128, 32, 32, seed=12)
    w = gen(3, 128, 3, 3, seed=13, scale=0.03)
    b = gen(3, seed=14)
    out = ops.conv2d(nhwc(x).cuda(), ops.pack_conv_weight(w.cuda()), 3, 3, bias=b.cuda(), out_nchw=True)
    torch.cuda.synchronize()
    assert out.shape == (2, 3, 32, 32)
    assert rel(out.cpu(), F.conv2d(x, w, b, padding=1)) < 2e-6


def test_conv_rejects_bad_shapes(hip):
    from ddnm_amd import ops
    from ddnm_amd._lib import DDNMHipError
    x = torch.zeros(1, 8, 8, 48, device="cuda")       # C0 % 32 != 0
    with pytest.raises(DDNMHipError):
        ops.conv2d(x, torch.zeros(128, 9, 48, device="cuda"), 128, 3)


# ------------------------------------------------------------------ GroupNorm
@pytest.mark.parametrize("B,C0,C1,H", [(2, 128, 0, 32), (2, 256, 128, 16), (1, 512, 256, 8), (3, 1024, 0, 8),
                                       (1, 128, 0, 64),
                                       (2, 32, 0, 16), (2, 64, 32, 16), (1, 192, 0, 8)])      # 1, 3, 6 channels / group
def test_groupnorm_affine(hip, B, C0, C1, H):
    from ddnm_amd import ops
    a = gen(B, C0, H, H, seed=20) * 2 + 0.7
    b2 = gen(B, C1, H, H, seed=21) if C1 else None
    C = C0 + C1
    gamma, beta = 1 + 0.1 * gen(C, seed=22), 0.1 * gen(C, seed=23)
    xin = a if b2 is None else torch.cat([a, b2], 1)
    ref = F.group_norm(xin, 32, gamma, beta, eps=1e-6)
    ws = ops.GroupNormWorkspace("cuda", B, C, B * ops.gn_nchunk(H * H, C) * 64)
    sc, sh = ops.group_norm_affine(nhwc(a).cuda(), None if b2 is None else nhwc(b2).cuda(), gamma.cuda(), beta.cuda(),
                                   1e-6, ws)
    torch.cuda.synchronize()
    sc, sh = sc[:B * C].reshape(B, C).cpu(), sh[:B * C].reshape(B, C).cpu()
    got = xin * sc[:, :, None, None] + sh[:, :, None, None]
    assert rel(got, ref) < 2e-6


# ------------------------------------------------------------------ GEMM / softmax / linear / embedding
@pytest.mark.parametrize("M,N,K,transb,batch", [(64, 64, 512, True, 3), (256, 256, 512, True, 2),
                                                (256, 512, 256, False, 2), (64, 256, 256, False, 6),
                                                (16, 16, 64, True, 2), (8, 32, 32, False, 3),
                                                (256, 256, 64, True, 24)])
def test_bgemm(hip, M, N, K, transb, batch):
    from ddnm_amd import ops
    A = gen(batch, M, K, seed=30)
    Bm = gen(batch, N, K, seed=31) if transb else gen(batch, K, N, seed=31)
    D = gen(batch, M, N, seed=32)
    ref = 0.5 * torch.bmm(A, Bm.transpose(1, 2) if transb else Bm) - D
    C = torch.empty(batch, M, N, device="cuda")
    ops.bgemm(A.cuda(), Bm.cuda(), C, M, N, K, lda=K, ldb=K if transb else N, ldc=N, transb=transb, batch=batch,
              sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), D=D.cuda(), ldd=N, sD=(M * N, 0), alpha=0.5, beta=-1.0)
    torch.cuda.synchronize()
    assert rel(C.cpu(), ref) < 2e-6


def test_bgemm_shared_operand_and_inner_stride(hip):
    from ddnm_amd import ops
    A = gen(64, 256, seed=33)                      # shared across the batch (stride 0)
    X = gen(2, 3, 256, 256, seed=34)
    ref = torch.matmul(A, X)
    C = torch.empty(2, 3, 64, 256, device="cuda")
    ops.bgemm(A.cuda(), X.cuda(), C, 64, 256, 256, lda=256, ldb=256, ldc=256, transb=False, batch=6, inner=3,
              sB=(3 * 256 * 256, 256 * 256), sC=(3 * 64 * 256, 64 * 256))
    torch.cuda.synchronize()
    assert rel(C.cpu(), ref) < 2e-6


@pytest.mark.parametrize("n", [64, 256, 1024, 100])
def test_softmax_rows(hip, n):
    from ddnm_amd import ops
    x = gen(37, n, seed=35) * 3
    got = ops.softmax_rows_(x.cuda().contiguous(), 37, n, n, 0.25)
    torch.cuda.synchronize()
    assert rel(got.cpu(), F.softmax(x * 0.25, dim=1)) < 2e-6


@pytest.mark.parametrize("silu", [False, True])
def test_linear(hip, silu):
    from ddnm_amd import ops
    x, W, b = gen(5, 512, seed=36), gen(300, 512, seed=37, scale=0.05), gen(300, seed=38)
    xin = x * torch.sigmoid(x) if silu else x
    got = ops.linear(x.cuda(), W.cuda(), b.cuda(), silu_in=silu)
    torch.cuda.synchronize()
    assert rel(got.cpu(), F.linear(xin, W, b)) < 2e-6


@pytest.mark.parametrize("K", [512, 1024])
def test_linear_wide_rows(hip, K):
    """The wide FiLM projection (N >= 4096 rows) takes the row-streaming kernel: 5 batch rows cross its 4-row register
    tile, and a row's result must not depend on how many rows the launch has."""
    from ddnm_amd import ops
    x, W, b = gen(5, K, seed=41), gen(4100, K, seed=42, scale=0.05), gen(4100, seed=43)
    xin = x * torch.sigmoid(x)
    got = ops.linear(x.cuda(), W.cuda(), b.cuda(), silu_in=True)
    one = ops.linear(x[3:4].cuda().contiguous(), W.cuda(), b.cuda(), silu_in=True)
    torch.cuda.synchronize()
    assert rel(got.cpu(), F.linear(xin, W, b)) < 2e-6
    assert torch.equal(got[3:4].cpu(), one.cpu())


def test_timestep_embedding(hip):
    from ddnm_amd import ops
    from oracle import unet_celeba
    import math
    t = torch.tensor([990.0, 430.0, 0.0, 10.0])
    half = 64
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    got = ops.timestep_embedding(t.cuda(), freq.cuda(), 0)
    torch.cuda.synchronize()
    assert (got.cpu() - unet_celeba.timestep_embedding(t, 128)).abs().max().item() < 2e-4   # sin(990) cond.
    got1 = ops.timestep_embedding(t.cuda(), freq.cuda(), 1).cpu()
    assert torch.equal(got1[:, :half], got.cpu()[:, half:]) and torch.equal(got1[:, half:], got.cpu()[:, :half])


def test_nchw_to_nhwc_pad(hip):
    from ddnm_amd import ops
    x = gen(2, 3, 16, 16, seed=39)
    got = ops.nchw_to_nhwc_pad(x.cuda(), 32).cpu()
    assert torch.equal(got[..., :3], nhwc(x)) and got[..., 3:].abs().max().item() == 0.0


# ------------------------------------------------------------------ operators (A, A^+) vs oracle
@pytest.mark.parametrize("name", ["sr_averagepooling", "sr_bicubic", "colorization", "inpainting",
                                  "cs_walshhadamard", "denoising"])
@pytest.mark.parametrize("d", [64, 256])
def test_operator_A_and_pinv(hip, name, d, golden_dir):
    from oracle import cases
    from tests.helpers import engine_operator, real_mask
    mask = real_mask(golden_dir) if (name == "inpainting" and d == 256) else None
    orc = cases.make_operator(name, d, mask)
    eng = engine_operator(name, d, mask)
    x = cases.operator_input(d, 2)
    y_o = orc.A(x)
    y_e = eng.A(x.cuda())
    p_o = orc.A_pinv(y_o)
    p_e = eng.A_pinv(y_o.cuda())
    torch.cuda.synchronize()
    assert y_e.shape == y_o.shape and p_e.shape == p_o.shape
    assert rel(y_e, y_o) < 3e-6 and rel(p_e, p_o) < 3e-6
    # golden from the real reference classes
    g = np.load(f"{golden_dir}/operators.npz")
    gy = torch.from_numpy(g[f"{name}_{d}_y"])
    if d == 64:
        assert rel(y_e, gy) < 3e-6
        assert rel(p_e.reshape(2, 3, d, d), torch.from_numpy(g[f"{name}_{d}_pinv"])) < 3e-6
    else:
        assert rel(y_e.cpu()[:, ::31], gy) < 3e-6
        assert rel(p_e.cpu().reshape(2, 3, d, d)[..., ::8, ::8], torch.from_numpy(g[f"{name}_{d}_pinv"])) < 3e-6
    # Moore-Penrose property A A^+ y = y (SURVEY.md section 4 invariants)
    assert rel(eng.A(p_e), y_o) < 1e-5


# ------------------------------------------------------------------ sampler step kernels vs oracle
@pytest.mark.parametrize("name", ["sr_averagepooling", "sr_bicubic", "colorization", "inpainting",
                                  "cs_walshhadamard", "denoising"])
@pytest.mark.parametrize("six", [False, True])
def test_ddnm_step(hip, name, six):
    from ddnm_amd import ops
    from oracle import cases, schedule
    from tests.helpers import engine_operator
    d, B = 64, 2
    orc, eng = cases.make_operator(name, d), engine_operator(name, d)
    xt, et6, nz = gen(B, 3, d, d, seed=40), gen(B, 6, d, d, seed=41), gen(B, 3, d, d, seed=42)
    et = et6[:, :3]
    y = orc.A(gen(B, 3, d, d, seed=43))
    betas = cases.betas()
    at, at_next = schedule.alpha_bar(betas, 500), schedule.alpha_bar(betas, 490)
    eta = 0.85
    x0 = (xt - et * (1 - at).sqrt()) / at.sqrt()
    x0h = x0 - orc.A_pinv(orc.A(x0.reshape(B, -1)) - y).reshape(x0.shape)
    ref = at_next.sqrt() * x0h + (1 - at_next).sqrt() * eta * nz + (1 - at_next).sqrt() * ((1 - eta ** 2) ** 0.5) * et
    s = ops.step_scalars(at, at_next, eta)
    et_dev = et6.cuda()[:, :3] if six else et.contiguous().cuda()
    x0_e, out = torch.empty(B, 3, d, d, device="cuda"), torch.empty(B, 3, d, d, device="cuda")
    eng.ddnm_step(xt.cuda(), et_dev, nz.cuda(), y.cuda(), s, x0_e, out)
    torch.cuda.synchronize()
    assert rel(x0_e, x0) < 1e-6
    assert rel(out, ref) < 3e-6
    if name in ("sr_averagepooling", "denoising", "inpainting"):
        assert torch.equal(x0_e.cpu(), x0), "x0 must be bit-exact (same fp32 evaluation order)"


def test_renoise_and_finalize(hip):
    from ddnm_amd import ops
    from oracle import sampler
    x0, nz, xo = gen(2, 3, 32, 32, seed=44), gen(2, 3, 32, 32, seed=45), gen(2, 3, 32, 32, seed=46)
    got = ops.renoise(x0.cuda(), nz.cuda(), 0.8, 0.6)
    img, psnr = ops.finalize_psnr(x0.cuda(), xo.cuda())
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), 0.8 * x0 + nz * 0.6)
    assert torch.equal(img.cpu(), torch.clamp((x0 + 1.0) / 2.0, 0.0, 1.0))
    assert (psnr.cpu().float() - sampler.psnr(x0, xo)).abs().max().item() < 1e-4


# ------------------------------------------------------------------ GroupNorm statistics from the conv epilogue
@pytest.mark.parametrize("B,C0,C1,H", [(4, 256, 0, 64), (4, 256, 256, 64), (8, 128, 128, 64), (2, 128, 384, 128),
                                       (2, 256, 0, 16), (1, 512, 128, 8)])       # split-K launches: stats from the reduction
def test_groupnorm_from_conv_epilogue_stats(hip, B, C0, C1, H):
    """The producing convolutions emit per-(tile, channel) partials; the consumer's GroupNorm affine is built
    from them (two sources = skip concat, possibly with different tilings) without re-reading the tensors."""
    from ddnm_amd import ops
    x = gen(B, 128, H, H, seed=50)
    w0 = gen(C0, 128, 3, 3, seed=51, scale=0.04)
    bias0, r0 = gen(C0, seed=55), gen(B, C0, H, H, seed=56)
    a = ops.conv2d(nhwc(x).cuda(), ops.pack_conv_weight(w0.cuda()), C0, 3, emit_stats=True, bias=bias0.cuda(),
                   res=nhwc(r0).cuda())
    ref0 = F.conv2d(x, w0, bias0, padding=1) + r0
    assert a.stats is not None and a.tiles > 0
    assert rel(nchw(a.t.cpu()), ref0) < 3e-6
    a1, ref1 = None, None
    if C1:
        w1 = gen(C1, 128, 1, 1, seed=52, scale=0.1)
        a1 = ops.conv2d(nhwc(x).cuda(), ops.pack_conv_weight(w1.cuda()), C1, 1, emit_stats=True, tile=2)
        ref1 = F.conv2d(x, w1, None)
        assert a1.stats is not None
    C = C0 + C1
    gamma, beta = 1 + 0.1 * gen(C, seed=53), 0.1 * gen(C, seed=54)
    xin = ref0 if ref1 is None else torch.cat([ref0, ref1], 1)
    ref = F.group_norm(xin, 32, gamma, beta, eps=1e-6)
    ws = ops.GroupNormWorkspace("cuda", B, C, 16)          # partial buffer unused on this path
    sc, sh = ops.group_norm_affine(a, a1, gamma.cuda(), beta.cuda(), 1e-6, ws)
    torch.cuda.synchronize()
    sc, sh = sc[:B * C].reshape(B, C).cpu(), sh[:B * C].reshape(B, C).cpu()
    got = xin * sc[:, :, None, None] + sh[:, :, None, None]
    assert rel(got, ref) < 3e-6


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,C,C0,C1,H", [(4, 128, 128, 128, 32),      # plain launch
                                         (2, 256, 256, 128, 16)])     # split-K launch: shortcut chunks are shared too
def test_conv_fused_shortcut(hip, f16, B, C, C0, C1, H):
    """conv2(3x3, GN+swish prologue) + 1x1 shortcut over the concat of two raw tensors, one launch."""
    from ddnm_amd import ops
    h = gen(B, C, H, H, seed=60)
    a, b2 = gen(B, C0, H, H, seed=61), gen(B, C1, H, H, seed=62)
    w2 = gen(C, C, 3, 3, seed=63, scale=(9 * C) ** -0.5)
    ws = gen(C, C0 + C1, 1, 1, seed=64, scale=(C0 + C1) ** -0.5)
    bias = gen(C, seed=65)
    sc, sh = gen(B, C, seed=66), gen(B, C, seed=67)
    act = h * sc[:, :, None, None] + sh[:, :, None, None]
    act = act * torch.sigmoid(act)
    ref = F.conv2d(act, w2, bias, padding=1) + F.conv2d(torch.cat([a, b2], 1), ws)
    out = ops.conv2d(nhwc(h).cuda(), ops.pack_conv_weight(w2.cuda()), C, 3, gn=(sc.cuda().contiguous(), sh.cuda().contiguous()),
                     bias=bias.cuda(), skip=(nhwc(a).cuda(), nhwc(b2).cuda()), skip_weight=ops.pack_skip_weight(ws.cuda()),
                     skip_weight_f16=ops.pack_skip_weight(ws.cuda(), f16=True) if f16 else None,
                     weight_f16=ops.pack_conv_weight_f16(w2.cuda()) if f16 else None)
    torch.cuda.synchronize()
    assert rel(nchw(out.cpu()), ref) < (2e-3 if f16 else 3e-6)


@pytest.mark.parametrize("name", ["sr_averagepooling", "sr_bicubic", "colorization", "inpainting", "cs_walshhadamard",
                                  "denoising"])
def test_operator_svd_surface(hip, name, golden_dir):
    """V / Vt / U / Ut / add_zeros / At / A_pinv_eta of the BASELINE operators (functions/svd_operators.py:9-97 and the
    per-class definitions).  Algebra first (basis-independent): V Vt = I, U Ut = I, A = U S Vt[:n], At and A_pinv_eta as
    the reference base class derives them; then the reference's own outputs (tests/golden/spectral.npz, every 5th
    entry): exact orderings for the permutation-type operators, and the basis-independent products for the operators
    built on a LAPACK SVD (their V_small columns are unique only up to sign / rotation of degenerate subspaces)."""
    from tests.helpers import engine_operator
    d, B = 64, 2
    op = engine_operator(name, d)
    from oracle import cases
    x = cases.operator_input(d, B).cuda()
    xf = x.reshape(B, -1)
    assert rel(op.V(op.Vt(x)), xf) < 2e-6 and rel(op.Vt(op.V(xf)), xf) < 2e-6
    y = op.A(x)
    assert rel(op.U(op.Ut(y)), y) < 2e-6
    s = op.singulars().float()
    n = s.numel()
    assert rel(op.U(s * op.Vt(x)[:, :n]), y) < 2e-5                       # A = U S V^T (:52-58)
    z = op.add_zeros(y)
    assert z.shape == (B, 3 * d * d) and torch.equal(z[:, :n], y) and not bool(z[:, n:].any())
    # <A x, w> = <x, At w>
    g = torch.Generator().manual_seed(5)
    w = torch.randn(B, n, generator=g).cuda()
    lhs = (y.double() * w.double()).sum()
    rhs = (xf.double() * op.At(w).double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4
    # A_pinv_eta(., 0) restricted to the non-zero singular values is A_pinv
    if float(s.min()) > 0:
        assert rel(op.A_pinv_eta(w, 0.0), op.A_pinv(w)) < 2e-5
    if name == "denoising":
        return
    gold = np.load(f"{golden_dir}/spectral.npz")
    gz, gw = torch.from_numpy(gold[f"{name}_z"]).cuda(), torch.from_numpy(gold[f"{name}_w"]).cuda()
    sp = lambda t: t[:, ::5].cpu()                                          # noqa: E731
    assert rel(sp(op.At(gw)), torch.from_numpy(gold[f"{name}_At"])) < 2e-5
    assert rel(sp(op.A_pinv_eta(gw, 0.3)), torch.from_numpy(gold[f"{name}_A_pinv_eta"])) < 2e-5
    assert rel(op.add_zeros(gw)[:, ::7].cpu(), torch.from_numpy(gold[f"{name}_add_zeros"])) == 0.0
    if name in ("inpainting", "cs_walshhadamard"):                          # permutations (+ FWHT): unique
        assert rel(sp(op.Vt(x)), torch.from_numpy(gold[f"{name}_Vt"])) < 2e-6
        assert rel(sp(op.V(gz)), torch.from_numpy(gold[f"{name}_V"])) < 2e-6
        assert rel(op.U(gw).cpu(), torch.from_numpy(gold[f"{name}_U"])) == 0.0


def test_superresolution_ratio16_surface(hip, golden_dir):
    """SuperResolution at ratio 16 (evaluation.sh:18: `--deg sr_averagepooling --deg_scale 16`): 256 entries per site, so
    V / Vt run as ONE MFMA GEMM over all sites (ddnm_bgemm_f32) instead of the register-resident site kernel; every entry
    point against the outputs of the reference class (functions/svd_operators.py:479-623, tests/golden/spectral_sr16.npz)."""
    from tests.helpers import engine_operator
    from oracle import cases
    d, B = 64, 2
    op = engine_operator("sr_averagepooling_x16", d)
    g = {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/spectral_sr16.npz").items()}
    x = cases.operator_input(d, B).cuda()
    xf = x.reshape(B, -1)
    z, w, e = g["z"].cuda(), g["w"].cuda(), g["e"].cuda()
    assert rel(op.A(x), g["A"]) < 2e-6 and rel(op.A_pinv(w), g["A_pinv"]) < 2e-6
    assert rel(op.V(op.Vt(x)), xf) < 2e-6 and rel(op.Vt(op.V(xf)), xf) < 2e-6
    s = op.singulars().float()
    n = s.numel()
    assert rel(op.U(s * op.Vt(x)[:, :n]), op.A(x)) < 2e-5
    # the reference's V_small comes from the same LAPACK call on a 1 x 256 row; compare the full vectors too
    assert rel(op.Vt(x), g["Vt"]) < 2e-5 and rel(op.V(z), g["V"]) < 2e-5
    assert rel(op.U(w), g["U"]) == 0.0 and rel(op.Ut(w), g["Ut"]) == 0.0
    assert rel(op.add_zeros(w), g["add_zeros"]) == 0.0
    assert rel(op.At(w), g["At"]) < 2e-5 and rel(op.A_pinv_eta(w, 0.3), g["A_pinv_eta"]) < 2e-5
    for tag, (a, st) in {"early": (0.2, 0.97), "late": (0.98, 0.15)}.items():
        assert rel(op.Lambda(z, a, 0.4, st, 0.85), g[f"Lambda_{tag}"]) < 1e-5
        assert rel(op.Lambda_noise(z, a, 0.4, st, 0.85, e), g[f"Lambda_noise_{tag}"]) < 1e-5


def test_general_a_dense_operator(hip, golden_dir):
    """GeneralA (functions/svd_operators.py:173-208): a dense measurement matrix with a full host SVD; every product is a
    ddnm_bgemm_f32 launch.  Basis-independent outputs against the reference class (tests/golden/general_a.npz: rank-deficient
    matrix, so the 1e-3 singular-value threshold acts), the SVD algebra, and a DDNM run with it as the operator."""
    from ddnm_amd.functions.svd_operators import GeneralA
    g = {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/general_a.npz").items()}
    op = GeneralA(g["A_mat"].cuda())
    x, w = g["x"].cuda(), g["w"].cuda()
    assert torch.equal((op.singulars() == 0).cpu(), g["singulars"] == 0)
    assert rel(op.singulars(), g["singulars"]) < 1e-6
    assert rel(op.A(x), g["A"]) < 2e-5 and rel(op.A_pinv(w), g["A_pinv"]) < 2e-5
    assert rel(op.At(w), g["At"]) < 2e-5 and rel(op.A_pinv_eta(w, 0.3), g["A_pinv_eta"]) < 2e-5
    xf = x.reshape(4, -1)
    assert rel(op.V(op.Vt(x)), xf) < 2e-6 and rel(op.U(op.Ut(w)), w) < 2e-6
    assert rel(op.A(op.A_pinv(op.A(x))), op.A(x)) < 2e-5            # A A^+ A = A
    z = op.add_zeros(w)
    assert z.shape == (4, 192) and torch.equal(z[:, :64], w) and not bool(z[:, 64:].any())
    with pytest.raises(NotImplementedError):                        # like the reference: no Lambda (svd_operators.py:93-97)
        op.Lambda(xf, 0.9, 0.2, 0.3, 0.85)


@pytest.mark.parametrize("B,C,H,gn", [(2, 128, 64, True), (1, 64, 32, False), (3, 128, 32, True)])
def test_small_cout_output_conv(hip, B, C, H, gn):
    """conv_out of the celeba Model (128 -> 3, GroupNorm + swish fused, NCHW result) on the vector-ALU kernel."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(3, C, 3, 3, generator=g) * (9 * C) ** -0.5
    b = torch.randn(3, generator=g)
    sc, sh = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    act = F.silu(x * sc[:, :, None, None] + sh[:, :, None, None]) if gn else x
    ref = F.conv2d(act, w, b, padding=1)
    got = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.pack_conv_weight(w.cuda()), 3, 3, bias=b.cuda(),
                     gn=(sc.cuda().contiguous(), sh.cuda().contiguous()) if gn else None, gn_silu=True, out_nchw=True)
    torch.cuda.synchronize()
    assert got.shape == (B, 3, H, H)
    assert rel(got, ref) < 2e-6


def test_conv_in_as_im2col_gemm(hip):
    """conv_in (3 -> 128, 3x3) as im2col (27 taps in one 32-wide K chunk) + 1x1 convolution."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 3, 64, 64, generator=g)
    w = torch.randn(128, 3, 3, 3, generator=g) * 27 ** -0.5
    b = torch.randn(128, generator=g)
    col = ops.nchw_im2col3x3_pad(x.cuda(), 32)
    out = ops.conv2d(col, ops.pack_conv_in_weight_im2col(w.cuda(), 32), 128, 1, bias=b.cuda(), emit_stats=True)
    torch.cuda.synchronize()
    assert rel(out.t.cpu().permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=1)) < 2e-6
    assert out.stats is not None


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,mag", [(8, 256, 512, 1.0), (3, 64, 512, 1.0), (2, 256, 256, 40.0), (1, 96, 128, 1e-3)])
def test_fused_single_head_attention(hip, B, T, C, mag):
    """ddnm_attn_fused_f32 (csrc/attn_d512.hip; the AttnBlock of guided_diffusion/models.py:171-185 in one launch) against an
    fp64 evaluation of the reference's formula and against the three-launch route (bgemm -> softmax_rows -> bgemm on the fp32
    MFMA): fp32 grade, for operand magnitudes far from 1 as well (the power-of-two operand scales)."""
    import math
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(T + C)
    qkv = (torch.randn(B, T, 3 * C, generator=g) * mag).cuda()
    qkv[..., :2 * C] *= (4.0 / (mag * C ** 0.25))          # scores of a few units: a softmax that is neither flat nor one-hot
    q, k, v = (qkv[..., i * C:(i + 1) * C].double() for i in range(3))
    want = torch.softmax(q @ k.transpose(1, 2) * (int(C) ** (-0.5)), dim=2) @ v
    bound = lambda t: 2.0 ** (14 - math.ceil(math.log2(float(t.abs().max()) * 37.0)))      # noqa: E731   a loose bound, like the model's
    got = ops.attn_fused(qkv, B, T, C, bound(qkv[..., :2 * C]), bound(qkv[..., 2 * C:]), float(int(C) ** (-0.5)))
    S = torch.empty(B, T, T, device="cuda")
    f = qkv.view(-1)
    ops.bgemm(f[0:], f[C:], S, T, T, C, lda=3 * C, ldb=3 * C, ldc=T, transb=True, batch=B, sA=(T * 3 * C, 0), sB=(T * 3 * C, 0), sC=(T * T, 0))
    ops.softmax_rows_(S, B * T, T, T, float(int(C) ** (-0.5)))
    o3 = torch.empty(B, T, C, device="cuda")
    ops.bgemm(S, f[2 * C:], o3, T, C, T, lda=T, ldb=3 * C, ldc=C, transb=False, batch=B, sA=(T * T, 0), sB=(T * 3 * C, 0), sC=(T * C, 0))
    torch.cuda.synchronize()
    rel = lambda a: ((a.double().cpu() - want.cpu()).norm() / want.cpu().norm()).item()      # noqa: E731
    e_f, e_3 = rel(got), rel(o3)
    assert bool(torch.isfinite(got).all())
    assert e_f < 2e-6 and e_f <= 2.0 * e_3 + 3e-7, (e_f, e_3)
