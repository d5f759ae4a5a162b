"""fp16-activation convolution kernel (csrc/conv16.hip, C ABI ddnm_conv16) against torch convolutions of the same
fp16-rounded operands: the `use_fp16` torso of guided_diffusion/unet.py:619-625 with fp32 accumulation."""
import pytest
import torch

from tests.helpers import rel

pytestmark = pytest.mark.gpu


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.float16).cuda()


def run_case(B, Cin, Cout, H, k, res=False, ups=False, res_ups=False, skip=None, bias=True, seed=21):
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(seed)
    Hs = H // 2 if ups else H
    x = torch.randn(B, Cin, Hs, Hs, generator=g).half().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (k * k * Cin) ** -0.5).half().float()
    b = torch.randn(Cout, generator=g) if bias else None
    Hr = H // 2 if res_ups else H
    r = torch.randn(B, Cout, Hr, Hr, generator=g).half().float() if res else None
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=k // 2)
    if res:
        ref = ref + (F.interpolate(r, scale_factor=2.0, mode="nearest") if res_ups else r)
    sk = skw = None
    if skip is not None:
        s0 = torch.randn(B, skip[0], H, H, generator=g).half().float()
        s1 = torch.randn(B, skip[1], H, H, generator=g).half().float() if skip[1] else None
        sw = (torch.randn(Cout, skip[0] + skip[1], 1, 1, generator=g) * (skip[0] + skip[1]) ** -0.5).half().float()
        ref = ref + F.conv2d(s0 if s1 is None else torch.cat([s0, s1], 1), sw)
        sk = (nhwc16(s0), None if s1 is None else nhwc16(s1))
        skw = ops.pack_conv_weight16(sw.cuda()).reshape(-1, skip[0] + skip[1]).contiguous()
    out = ops.conv16(nhwc16(x), ops.pack_conv_weight16(w.cuda()), Cout, k, bias=None if b is None else b.cuda(),
                     res=None if r is None else nhwc16(r), res_ups=res_ups, ups=ups, skip=sk, skip_weight=skw)
    torch.cuda.synchronize()
    got = out.t.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = rel(got, ref)
    assert err < 6e-4, err                                    # one fp16 rounding of the result
    assert rel(got, ref.half().float()) < 4e-4
    return out, got


@pytest.mark.parametrize("B,Cin,Cout,H,kw", [
    (1, 256, 256, 64, dict(res=True)),                       # 256-pixel tiles (8 x 32 patches)
    (2, 128, 256, 32, dict()),                               # 128-pixel tiles, two images
    (1, 64, 128, 32, dict(res=True)),                        # Cout = 128: half of the waves idle in the epilogue
    (2, 512, 512, 16, dict(res=True)),                       # 16 x 16 patches, split-K
    (1, 256, 256, 64, dict(ups=True)),                       # operand read through nearest x2
    (2, 256, 256, 32, dict(res=True, res_ups=True)),         # residual through nearest x2
    (1, 256, 512, 32, dict(skip=(128, 128))),                # fused 1x1 shortcut over a two-tensor concat
    (1, 128, 256, 64, dict(skip=(128, 0), bias=False)),
    (4, 1024, 1024, 16, dict(res=True)),                     # deep split-K
])
def test_conv16_3x3(hip, B, Cin, Cout, H, kw):
    run_case(B, Cin, Cout, H, 3, **kw)


@pytest.mark.parametrize("B,Cin,Cout,H,kw", [
    (2, 512, 1536, 32, dict()),                              # qkv of a 32 x 32 attention block
    (2, 512, 512, 32, dict(res=True)),                       # proj_out + residual
    (3, 1024, 3072, 8, dict()),                              # 8 x 8 level: tiles span images, ragged M (192 rows)
    (1, 9 * 256, 256, 8, dict(res=True)),                    # an im2col'ed 3x3 layer (K = 9*Cin)
    (1, 128, 384, 16, dict()),
])
def test_conv16_1x1(hip, B, Cin, Cout, H, kw):
    run_case(B, Cin, Cout, H, 1, **kw)


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(2, 256, 256, 64, 3), (2, 512, 512, 16, 3), (2, 256, 768, 32, 1),
                                            (4, 512, 1024, 8, 1)])
def test_conv16_groupnorm_partials(hip, B, Cin, Cout, H, k):
    """The emitted per-(tile, channel) partials reproduce F.group_norm of the ROUNDED fp16 output."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    out, got = run_case(B, Cin, Cout, H, k, res=True)
    if out.stats is None:
        assert k == 1 and (H * H) % 128 != 0
        return
    gamma, beta = torch.ones(Cout), torch.zeros(Cout)
    ws = ops.GroupNormWorkspace("cuda", B, Cout, 16)
    sc, sh = ops.group_norm_affine(out, None, gamma.cuda(), beta.cuda(), 1e-5, ws)
    torch.cuda.synchronize()
    sc, sh = sc[:B * Cout].reshape(B, Cout).cpu(), sh[:B * Cout].reshape(B, Cout).cpu()
    assert rel(got * sc[:, :, None, None] + sh[:, :, None, None], F.group_norm(got, 32, eps=1e-5)) < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,H,kw", [
    (4, 128, 128, 128, dict(res=True)),                      # 256 tiles of 8 x 32 pixels, two 64-channel chunks
    (4, 64, 128, 128, dict()),                               # one chunk (the input convolution's shape class)
    (4, 256, 128, 128, dict(res=True, bias=False)),          # four chunks (data gradient of a 128 -> 256 layer)
    (1, 128, 128, 256, dict(res=True)),                      # one image at 256 x 256
    (1, 128, 128, 256, dict(ups=True)),                      # operand through nearest x2
    (4, 128, 128, 128, dict(res=True, res_ups=True)),
])
def test_conv16_n128_tile(hip, B, Cin, Cout, H, kw):
    """Cout = 128 launches with >= 128 pixel tiles run conv16_n128_kernel (256 x 128 tile, 4 waves, two workgroups per
    CU; round 5); the result equals the 256-channel-tile kernel's bit for bit (same products, same summation order per
    output), GroupNorm partials included."""
    out, _ = run_case(B, Cin, Cout, H, 3, **kw)
    assert out.stats is not None and out.tiles == H * H // 256


@pytest.mark.parametrize("B,C0,C1,H,silu,res", [(4, 128, 0, 128, True, True), (4, 64, 64, 128, True, False),
                                               (1, 128, 0, 256, False, True), (5, 256, 0, 128, True, True)])
def test_conv16_n128_fused_groupnorm(hip, B, C0, C1, H, silu, res):
    """The n128 kernel's fused GroupNorm-in-LDS (single halo buffer, activated between chunks): equals the pre-pass
    route bit for bit and the torch convolution of the fp16-rounded activated tensor."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(29)
    Cin, Cout = C0 + C1, 128
    a = torch.randn(B, C0, H, H, generator=g).half().float()
    b = torch.randn(B, C1, H, H, generator=g).half().float() if C1 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().float()
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
    r = torch.randn(B, Cout, H, H, generator=g).half().float() if res else None
    x = a if b is None else torch.cat([a, b], 1)
    act = x * sc[:, :, None, None] + sh[:, :, None, None]
    act = (F.silu(act) if silu else act).half().float()
    ref = F.conv2d(act, w, bias, padding=1) + (r if res else 0)
    gn = (sc.cuda().contiguous(), sh.cuda().contiguous())
    w16 = ops.pack_conv_weight16(w.cuda())
    out = ops.conv16(nhwc16(a), w16, Cout, 3, src1=None if b is None else nhwc16(b), gn=gn, gn_silu=silu, bias=bias.cuda(),
                     res=None if r is None else nhwc16(r))
    pre = ops.gn_apply16(nhwc16(a), None if b is None else nhwc16(b), gn, silu)
    out2 = ops.conv16(pre, w16, Cout, 3, bias=bias.cuda(), res=None if r is None else nhwc16(r))
    torch.cuda.synchronize()
    assert rel(out.t.float().cpu().permute(0, 3, 1, 2), ref) < 8e-4
    assert torch.equal(out.t, out2.t) and torch.equal(out.stats, out2.stats)


def test_conv16_n128_equals_wide_tile(hip):
    """A/B of the two kernels on one launch through the plan switch of a child process is not possible in-process (the
    switch is read once); instead: the n128 result of a Cout = 128 layer equals rows 0..127 of the same layer padded to
    Cout = 256 with zero weights, which runs on the 256-channel tile."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(30)
    B, Cin, H = 4, 128, 128
    x = nhwc16(torch.randn(B, Cin, H, H, generator=g))
    w = (torch.randn(128, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5)
    wpad = torch.cat([w, torch.zeros_like(w)], 0)
    bias = torch.randn(128, generator=g)
    a = ops.conv16(x, ops.pack_conv_weight16(w.cuda()), 128, 3, bias=bias.cuda())
    b = ops.conv16(x, ops.pack_conv_weight16(wpad.cuda()), 256, 3, bias=torch.cat([bias, torch.zeros(128)]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(a.t, b.t[..., :128].contiguous())
    assert bool((b.t[..., 128:] == 0).all())


@pytest.mark.parametrize("B,Cin,Cout,H,scale", [(4, 1024, 1024, 16, 40.0),      # 8 slices: fp16 slabs
                                                 (1, 2048, 1024, 16, 40.0),      # 16 slices: fp32 slabs
                                                 (2, 512, 512, 16, 200.0)])
def test_conv16_splitk_large_magnitudes(hip, B, Cin, Cout, H, scale):
    """Split-K partial sums of large activations: slices rounded to fp16 (<= 8 slices) or kept fp32 (deeper splits); the
    result stays within one fp16 rounding of the fp32-accumulated reference and finite."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(41)
    x = (scale * torch.randn(B, Cin, H, H, generator=g)).half().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().float()
    ref = F.conv2d(x, w, None, padding=1)
    out = ops.conv16(nhwc16(x), ops.pack_conv_weight16(w.cuda()), Cout, 3)
    torch.cuda.synchronize()
    got = out.t.float().cpu().permute(0, 3, 1, 2)
    assert bool(torch.isfinite(got).all())
    assert rel(got, ref) < 8e-4, rel(got, ref)


def test_conv16_is_deterministic(hip):
    a, _ = run_case(2, 512, 512, 16, 3, res=True)
    b, _ = run_case(2, 512, 512, 16, 3, res=True)
    assert torch.equal(a.t, b.t) and torch.equal(a.stats, b.stats)


@pytest.mark.parametrize("B,C0,C1,Cout,H,silu,ups,res", [(2, 256, 0, 256, 32, True, False, True),
                                                         (1, 128, 128, 256, 64, True, False, False),   # concat, 256-pixel tiles
                                                         (2, 512, 256, 512, 16, True, False, True),    # concat + split-K
                                                         (1, 256, 0, 256, 64, True, True, False),      # through nearest x2
                                                         (2, 64, 0, 128, 32, False, False, False)])    # affine only
def test_conv16_fused_groupnorm_and_concat(hip, B, C0, C1, Cout, H, silu, ups, res):
    """GroupNorm affine (+ swish) and the channel concat fused into the loader: equals the convolution of the
    fp16-ROUNDED activated tensor (the same rounding point as the ddnm_gn_apply_h16 pre-pass), zero padding applied
    after the activation like the reference."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(23)
    Cin, Hs = C0 + C1, (H // 2 if ups else H)
    a = torch.randn(B, C0, Hs, Hs, generator=g).half().float()
    b = torch.randn(B, C1, Hs, Hs, generator=g).half().float() if C1 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().float()
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
    r = torch.randn(B, Cout, H, H, generator=g).half().float() if res else None
    x = a if b is None else torch.cat([a, b], 1)
    act = x * sc[:, :, None, None] + sh[:, :, None, None]
    if silu:
        act = F.silu(act)
    act = act.half().float()
    if ups:
        act = F.interpolate(act, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(act, w, bias, padding=1) + (r if res else 0)
    gn = (sc.cuda().contiguous(), sh.cuda().contiguous())
    out = ops.conv16(nhwc16(a), ops.pack_conv_weight16(w.cuda()), Cout, 3, src1=None if b is None else nhwc16(b), gn=gn,
                     gn_silu=silu, bias=bias.cuda(), res=None if r is None else nhwc16(r), ups=ups)
    torch.cuda.synchronize()
    got = out.t.float().cpu().permute(0, 3, 1, 2)
    assert rel(got, ref) < 8e-4
    # and bit-identical to the pre-pass route
    pre = ops.gn_apply16(nhwc16(a), None if b is None else nhwc16(b), gn, silu)
    out2 = ops.conv16(pre, ops.pack_conv_weight16(w.cuda()), Cout, 3, bias=bias.cuda(), res=None if r is None else nhwc16(r),
                      ups=ups)
    torch.cuda.synchronize()
    assert torch.equal(out.t, out2.t)


# ------------------------------------------------------------------ the other kernels of the fp16-activation path
def test_conv16_out_small_cout(hip):
    """Output convolution: 256 -> 6 channels, fp32 NCHW result (unet.py:627-631)."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 256, 64, 64, generator=g).half().float()
    w = (torch.randn(6, 256, 3, 3, generator=g) * (9 * 256) ** -0.5).half().float()
    b = torch.randn(6, generator=g)
    got = ops.conv16_out(nhwc16(x), ops.pack_conv_weight16(w.cuda()), 6, bias=b.cuda())
    torch.cuda.synchronize()
    assert got.shape == (2, 6, 64, 64) and got.dtype == torch.float32
    assert rel(got, F.conv2d(x, w, b, padding=1)) < 2e-5
    # GroupNorm affine + swish fused into its loader: bit-identical to the pre-pass route
    sc, sh = torch.randn(2, 256, generator=g).cuda(), torch.randn(2, 256, generator=g).cuda()
    pre = ops.gn_apply16(nhwc16(x), None, (sc, sh), True)
    want = ops.conv16_out(pre, ops.pack_conv_weight16(w.cuda()), 6, bias=b.cuda())
    fused = ops.conv16_out(nhwc16(x), ops.pack_conv_weight16(w.cuda()), 6, bias=b.cuda(), gn=(sc, sh), gn_silu=True)
    torch.cuda.synchronize()
    assert torch.equal(fused, want)


@pytest.mark.parametrize("C0,C1,silu,pool", [(256, 0, True, False), (256, 128, True, False), (128, 0, False, False),
                                             (256, 0, True, True), (512, 0, False, True)])
def test_gn_apply_h16(hip, C0, C1, silu, pool):
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(32)
    B, H = 2, 16
    a = torch.randn(B, C0, H, H, generator=g).half().float()
    b = torch.randn(B, C1, H, H, generator=g).half().float() if C1 else None
    C = C0 + C1
    sc, sh = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    x = a if b is None else torch.cat([a, b], 1)
    ref = x * sc[:, :, None, None] + sh[:, :, None, None]
    if silu:
        ref = F.silu(ref)
    if pool:
        ref = F.avg_pool2d(ref, 2, 2)
    got = ops.gn_apply16(nhwc16(a), None if b is None else nhwc16(b), (sc.cuda().contiguous(), sh.cuda().contiguous()), silu,
                         pool=pool)
    torch.cuda.synchronize()
    assert rel(got.float().cpu().permute(0, 3, 1, 2), ref) < 4e-4
    if pool:       # the raw (un-normalised) pooling of the shortcut branch
        got = ops.gn_apply16(nhwc16(a), None, None, False, pool=True)
        assert rel(got.float().cpu().permute(0, 3, 1, 2), F.avg_pool2d(a, 2, 2)) < 4e-4


def test_im2col_nchw_stats_h16(hip):
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(33)
    B, C, H = 3, 128, 8
    x = torch.randn(B, C, H, H, generator=g).half().float()
    sc, sh = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    act = F.silu(x * sc[:, :, None, None] + sh[:, :, None, None])
    col = ops.im2col16(nhwc16(x), None, (sc.cuda().contiguous(), sh.cuda().contiguous()), True)
    torch.cuda.synchronize()
    ref = F.unfold(act, 3, padding=1).reshape(B, C, 9, H * H).permute(0, 3, 2, 1).reshape(B, H, H, 9 * C)   # [tap][c]
    assert rel(col.float().cpu(), ref) < 4e-4
    # fp32 NCHW image -> zero-padded fp16 NHWC
    img = torch.randn(2, 3, 32, 32, generator=g)
    got = ops.nchw_to_nhwc16(img.cuda(), 64).float().cpu()
    assert torch.equal(got[..., :3], img.half().float().permute(0, 2, 3, 1)) and got[..., 3:].abs().max() == 0
    # stand-alone statistics == group_norm of the fp16 tensor
    t16 = nhwc16(x)
    a = ops.gn_stats16(t16)
    ws = ops.GroupNormWorkspace("cuda", B, C, 16)
    s2, h2 = ops.group_norm_affine(a, None, torch.ones(C).cuda(), torch.zeros(C).cuda(), 1e-5, ws)
    torch.cuda.synchronize()
    s2, h2 = s2[:B * C].reshape(B, C).cpu(), h2[:B * C].reshape(B, C).cpu()
    assert rel(x * s2[:, :, None, None] + h2[:, :, None, None], F.group_norm(x, 32, eps=1e-5)) < 2e-5


@pytest.mark.parametrize("B,T,C", [(2, 1024, 512), (1, 256, 1024), (3, 64, 256), (1, 128, 64)])
def test_attn16_matches_legacy_qkv_attention(hip, B, T, C):
    """QKVAttentionLegacy (unet.py:339-354) on the head-major fused qkv tensor: fp32 reference of the fp16 inputs."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(34)
    nh = C // 64
    qkv = (torch.randn(B, 3 * C, T, generator=g) * 1.5).half().float()          # reference layout [N, 3C, T]
    q, k, v = qkv.reshape(B * nh, 192, T).split(64, dim=1)
    scale = 1 / (64 ** 0.25)
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v).reshape(B, C, T)
    side = int(T ** 0.5) if int(T ** 0.5) ** 2 == T else None
    H, W = (side, side) if side else (T // 8, 8)
    got = ops.attn16(qkv.permute(0, 2, 1).contiguous().half().cuda().reshape(B, H, W, 3 * C), C)
    torch.cuda.synchronize()
    err = rel(got.float().cpu().reshape(B, T, C).permute(0, 2, 1), ref)
    assert err < 2e-3, err                                  # probabilities rounded to fp16 like the reference


@pytest.mark.parametrize("B,Cin,Cout,H,k,film", [(4, 1024, 1024, 16, 3, True), (2, 512, 512, 16, 3, False),
                                                 (4, 512, 1024, 32, 3, True), (4, 9 * 512, 512, 8, 1, True),
                                                 (1, 9 * 256, 256, 8, 1, False)])
def test_conv16_splitk_finalizes_consumer_groupnorm(hip, B, Cin, Cout, H, k, film):
    """Split-K launches can finalize the GroupNorm(+FiLM) that will consume their output in the reduction pass
    (ddnm_conv16_desc::fin_*): same output tensor as the plain launch, and the affine equals what
    ddnm_gn_finalize_tiles_f32 makes of the plain launch's partials (fp64 combination in another fixed order: <= 1e-6)
    and F.group_norm of the rounded output."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, H, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (k * k * Cin) ** -0.5).half()
    bias = torch.randn(Cout, generator=g).cuda()
    r = nhwc16(torch.randn(B, Cout, H, H, generator=g))
    gamma, beta = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(), (0.1 * torch.randn(Cout, generator=g)).cuda()
    fl = torch.randn(B, 2 * Cout + 40, generator=g).cuda() if film else None
    fs = 2 * Cout + 40
    w16 = ops.pack_conv_weight16(w.cuda())
    plain = ops.conv16(nhwc16(x), w16, Cout, k, bias=bias, res=r)
    ws0 = ops.GroupNormWorkspace("cuda", B, Cout, 16)
    sc0, sh0 = ops.group_norm_affine(plain, None, gamma, beta, 1e-5, ws0, film=fl, film_stride=fs if film else 0)
    sc0, sh0 = sc0[:B * Cout].clone(), sh0[:B * Cout].clone()
    ws1 = ops.GroupNormWorkspace("cuda", B, Cout, 16)
    fused = ops.conv16(nhwc16(x), w16, Cout, k, bias=bias, res=r, fin=("gn", gamma, beta, fl, fs if film else 0, 1e-5, ws1))
    torch.cuda.synchronize()
    assert fused.gn is not None and fused.gn[2] == "gn", "this shape must take the finalizing reduction"
    assert torch.equal(fused.t, plain.t)
    sc1, sh1 = fused.gn[0][:B * Cout], fused.gn[1][:B * Cout]
    assert rel(sc1, sc0) < 1e-6 and rel(sh1, sh0) < 1e-6
    assert fused.tiles == 1
    # its one-tile partials serve any other consumer (e.g. a later skip concat)
    sc2, sh2 = ops.group_norm_affine(fused, None, gamma, beta, 1e-5, ops.GroupNormWorkspace("cuda", B, Cout, 16),
                                     film=fl, film_stride=fs if film else 0)
    assert rel(sc2[:B * Cout], sc0) < 1e-6 and rel(sh2[:B * Cout], sh0) < 1e-6
    got = fused.t.float().cpu().permute(0, 3, 1, 2)
    want = F.group_norm(got, 32, gamma.cpu(), beta.cpu(), eps=1e-5)
    if film:
        s_, t_ = fl.cpu()[:, :Cout], fl.cpu()[:, Cout:2 * Cout]
        want = want * (1 + s_[:, :, None, None]) + t_[:, :, None, None]
    sc, sh = sc1.reshape(B, Cout).cpu(), sh1.reshape(B, Cout).cpu()
    assert rel(got * sc[:, :, None, None] + sh[:, :, None, None], want) < 2e-5


def test_conv16_unsplit_launch_ignores_fin(hip):
    """A launch that is not split (enough tiles for the chip) does not finalize: the caller falls back to the finalize
    kernel (Act.gn is None) and the partials keep their per-tile layout."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 256, 64, 64, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.02
    gamma, beta = torch.ones(256).cuda(), torch.zeros(256).cuda()
    ws = ops.GroupNormWorkspace("cuda", 2, 256, 16)
    out = ops.conv16(nhwc16(x), ops.pack_conv_weight16(w.cuda()), 256, 3, fin=("gn", gamma, beta, None, 0, 1e-5, ws))
    assert out.gn is None and out.tiles > 1
