"""ADM (ImageNet) UNet engine parity on the GPU (fp32 MFMA kernels) vs reference goldens / oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import engine_operator, rel

pytestmark = pytest.mark.gpu


def build(cfg, sd):
    from ddnm_amd.guided_diffusion.unet import create_model
    m = create_model(**vars(cfg.model))
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("kind,batch", [("small", 2), ("mid", 2), ("full", 1)])
def test_adm_forward_matches_reference_golden(hip, kind, batch, golden_dir):
    from oracle import cases
    cfg, sd = cases.adm_net(kind)
    x, t, y = cases.adm_forward_inputs(cfg, batch)
    m = build(cfg, sd)
    e = m(x.cuda(), t.cuda(), y.cuda()) if y is not None else m(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    e = e.cpu()
    g = np.load(f"{golden_dir}/adm_forward.npz")
    ref = torch.from_numpy(g[f"{kind}_eps"])
    got = e if kind != "full" else e[..., ::4, ::4]
    assert got.shape == ref.shape
    assert rel(got, ref) < 1e-5
    asum = g[f"{kind}_stats"][2]
    assert abs(e.double().abs().sum().item() - asum) / asum < 1e-5


def test_adm_kernels_film_avgpool_embedding(hip):
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C, H = 2, 128, 16
    x = torch.randn(B, C, H, H, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    film = torch.randn(B, 700, generator=g)
    s, t = film[:, 100:100 + C], film[:, 100 + C:100 + 2 * C]
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5) * (1 + s[:, :, None, None]) + t[:, :, None, None]
    ws = ops.GroupNormWorkspace("cuda", B, C, B * ops.gn_nchunk(H * H, C) * 64)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    sc, sh = ops.group_norm_affine(xn, None, gamma.cuda(), beta.cuda(), 1e-5, ws, film=film.cuda()[:, 100:], film_stride=700)
    got = x * sc[:B * C].reshape(B, C).cpu()[:, :, None, None] + sh[:B * C].reshape(B, C).cpu()[:, :, None, None]
    assert rel(got, ref) < 2e-6
    # avgpool2 with the activation prologue
    pooled = ops.avgpool2_nhwc(xn, gn=(sc, sh), silu=True).cpu().permute(0, 3, 1, 2)
    assert rel(pooled, F.avg_pool2d(F.silu(ref), 2, 2)) < 2e-6
    assert rel(ops.avgpool2_nhwc(xn).cpu().permute(0, 3, 1, 2), F.avg_pool2d(x, 2, 2)) < 1e-6
    # class-label embedding
    emb, table = torch.randn(3, 64, generator=g), torch.randn(10, 64, generator=g)
    idx = torch.tensor([9, 0, 4])
    got = ops.embedding_add_(emb.clone().cuda(), table.cuda(), idx.cuda()).cpu()
    assert torch.equal(got, emb + table[idx])
    # a label outside the table never reads out of bounds: its row is poisoned instead (nn.Embedding would raise)
    bad = ops.embedding_add_(emb.clone().cuda(), table.cuda(), torch.tensor([9, 10, -1]).cuda()).cpu()
    assert torch.equal(bad[0], emb[0] + table[9]) and bool(torch.isnan(bad[1:]).all())


def test_conv_residual_through_upsample(hip):
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, 16, 16, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    r = torch.randn(2, 128, 8, 8, generator=g)
    ref = F.conv2d(x, w, None, padding=1) + F.interpolate(r, scale_factor=2, mode="nearest")
    out = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.pack_conv_weight(w.cuda()), 128, 3,
                     res=r.permute(0, 2, 3, 1).contiguous().cuda(), res_ups=True)
    assert rel(out.cpu().permute(0, 3, 1, 2), ref) < 2e-6


@pytest.mark.parametrize("name", ["colorization", "inpainting"])
def test_adm_sampler_vs_reference_golden(hip, name, golden_dir):
    """imagenet-style net (6-channel learn_sigma head, FiLM, attention heads), 56-iteration time-travel run."""
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from oracle import cases, schedule
    cfg, sd = cases.adm_net("mid")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
    y = cases.make_operator(name, cfg.data.image_size).A(x_orig)
    op = engine_operator(name, cfg.data.image_size)
    xs, x0s = ddnm_diffusion(x_T.cuda(), build(cfg, sd), cases.betas().cuda(), 0.85, op, y.cuda(), cls_fn=None,
                             classes=None, config=cfg, noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    g = np.load(f"{golden_dir}/adm_forward.npz")
    assert rel(xs[0], torch.from_numpy(g[f"mid_{name}_x"])) < 2e-4
    assert rel(x0s[0], torch.from_numpy(g[f"mid_{name}_x0"])) < 2e-4


# ------------------------------------------------------------------ fp16-operand MFMA path (use_fp16 torso)
@pytest.mark.parametrize("B,C0,C1,Cout,H,gn,res,ups", [(1, 256, 0, 256, 32, True, True, False),
                                                       (2, 256, 256, 256, 16, True, False, False),
                                                       (1, 128, 0, 128, 64, False, True, False),
                                                       (1, 256, 0, 256, 16, True, True, True),
                                                       (4, 512, 0, 512, 16, True, True, False),
                                                       (4, 1024, 0, 1024, 8, True, True, False),     # 8x8: im2col + fp16 GEMM
                                                       (4, 512, 512, 512, 8, True, False, False),    # ... over a concat
                                                       (4, 256, 0, 256, 8, False, True, False)])     # ... without GroupNorm
def test_conv3x3_f16_operands(hip, B, C0, C1, Cout, H, gn, res, ups):
    """fp16 MFMA operands, fp32 accumulate.  Checked two ways: (a) against an fp32 convolution of the
    fp16-ROUNDED operands (isolates the kernel: only accumulation order differs, tol 2e-5), (b) against
    the unrounded fp32 convolution (the precision contract: fp16-class error, tol 2e-3)."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(11)
    Cin = C0 + C1
    a = torch.randn(B, C0, H, H, generator=g)
    b = torch.randn(B, C1, H, H, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
    Ho = 2 * H if ups else H
    r = torch.randn(B, Cout, Ho, Ho, generator=g) if res else None
    x = a if b is None else torch.cat([a, b], 1)
    act = x
    if gn:
        act = x * sc[:, :, None, None] + sh[:, :, None, None]
        act = act * torch.sigmoid(act)
    if ups:
        act = F.interpolate(act, scale_factor=2.0, mode="nearest")
    ref32 = F.conv2d(act, w, bias, padding=1) + (r if res else 0)
    ref16 = F.conv2d(act.half().float(), w.half().float(), bias, padding=1) + (r if res else 0)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()       # noqa: E731
    out = ops.conv2d(nh(a), ops.pack_conv_weight(w.cuda()), Cout, 3, src1=None if b is None else nh(b), bias=bias.cuda(),
                     gn=(sc.cuda().contiguous(), sh.cuda().contiguous()) if gn else None, gn_silu=True,
                     res=nh(r) if res else None, ups=ups, emit_stats=True, weight_f16=ops.pack_conv_weight_f16(w.cuda()))
    torch.cuda.synchronize()
    got = out.t.cpu().permute(0, 3, 1, 2)
    assert rel(got, ref16) < 2e-5
    assert rel(got, ref32) < 2e-3


@pytest.mark.parametrize("B,C0,C1,Cout,H,gn,res", [(4, 512, 0, 1536, 32, True, False),     # qkv of a 32x32 attention block
                                                   (4, 1024, 0, 1024, 16, False, True),    # proj_out + residual
                                                   (4, 1024, 0, 3072, 8, True, False),     # 8x8 level: batch folded, split-K
                                                   (2, 512, 256, 256, 32, False, False),   # un-fused shortcut over a concat
                                                   (1, 256, 0, 768, 16, True, False)])
def test_conv1x1_f16_operands(hip, B, C0, C1, Cout, H, gn, res):
    """1x1 convolutions of the fp16 torso (qkv / proj_out / shortcut) on the fp16 GEMM kernel: against an fp32
    convolution of the fp16-rounded operands (2e-5) and of the unrounded ones (fp16-class, 2e-3); the emitted
    GroupNorm partials reproduce F.group_norm of the output."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(13)
    Cin = C0 + C1
    a = torch.randn(B, C0, H, H, generator=g)
    b = torch.randn(B, C1, H, H, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * Cin ** -0.5
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.randn(B, Cin, generator=g), torch.randn(B, Cin, generator=g)
    r = torch.randn(B, Cout, H, H, generator=g) if res else None
    x = a if b is None else torch.cat([a, b], 1)
    act = x * sc[:, :, None, None] + sh[:, :, None, None] if gn else x
    ref32 = F.conv2d(act, w, bias) + (r if res else 0)
    ref16 = F.conv2d(act.half().float(), w.half().float(), bias) + (r if res else 0)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()       # noqa: E731
    out = ops.conv2d(nh(a), ops.pack_conv_weight(w.cuda()), Cout, 1, src1=None if b is None else nh(b), bias=bias.cuda(),
                     gn=(sc.cuda().contiguous(), sh.cuda().contiguous()) if gn else None, gn_silu=False,
                     res=nh(r) if res else None, emit_stats=True, weight_f16=ops.pack_conv_weight_f16(w.cuda()))
    torch.cuda.synchronize()
    got = out.t.cpu().permute(0, 3, 1, 2)
    assert rel(got, ref16) < 2e-5
    assert rel(got, ref32) < 2e-3
    if out.stats is not None:
        gamma, beta = torch.ones(Cout), torch.zeros(Cout)
        ws = ops.GroupNormWorkspace("cuda", B, Cout, 16)
        s2, h2 = ops.group_norm_affine(out, None, gamma.cuda(), beta.cuda(), 1e-5, ws)
        torch.cuda.synchronize()
        s2, h2 = s2[:B * Cout].reshape(B, Cout).cpu(), h2[:B * Cout].reshape(B, Cout).cpu()
        assert rel(got * s2[:, :, None, None] + h2[:, :, None, None], F.group_norm(got, 32, eps=1e-5)) < 1e-5
    else:
        assert H * H % 256 != 0


@pytest.mark.parametrize("C0,C1,Cout,H,ups", [(256, 0, 256, 32, False), (256, 256, 256, 32, False), (128, 0, 512, 16, True)])
def test_conv3x3_f16_groupnorm_prepass_is_bit_identical(hip, monkeypatch, C0, C1, Cout, H, ups):
    """ddnm_gn_apply_f16 + conv(src_f16) == the fused GroupNorm/swish prologue, bit for bit (same fp32 math,
    same rounding point, same MFMA order)."""
    from ddnm_amd import ops
    g = torch.Generator().manual_seed(12)
    B, Cin = 2, C0 + C1
    a = torch.randn(B, H, H, C0, generator=g).cuda()
    b = torch.randn(B, H, H, C1, generator=g).cuda() if C1 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).cuda()
    sc, sh = torch.randn(B, Cin, generator=g).cuda(), torch.randn(B, Cin, generator=g).cuda()
    w32, w16 = ops.pack_conv_weight(w), ops.pack_conv_weight_f16(w)
    outs = []
    for min_cout in (1 << 30, 0):
        monkeypatch.setattr(ops, "_F16_PREPASS_MIN_COUT", min_cout)
        outs.append(ops.conv2d(a, w32, Cout, 3, src1=b, gn=(sc, sh), gn_silu=True, ups=ups, weight_f16=w16).clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind,batch", [("mid", 2), ("full", 1)])
def test_adm_forward_fp16_torso(hip, kind, batch, golden_dir):
    """`convert_to_fp16()` engine vs the fp32 goldens of the reference: single-forward rel-L2 <= 3e-3
    (SURVEY.md section 8c bar for half-precision kernels; reference fp16 vs fp32 itself: 1.4e-3)."""
    from oracle import cases
    cfg, sd = cases.adm_net(kind)
    x, t, y = cases.adm_forward_inputs(cfg, batch)
    m = build(cfg, sd)
    m.convert_to_fp16()
    e = (m(x.cuda(), t.cuda(), y.cuda()) if y is not None else m(x.cuda(), t.cuda())).cpu()
    g = np.load(f"{golden_dir}/adm_forward.npz")
    ref = torch.from_numpy(g[f"{kind}_eps"])
    got = e if kind != "full" else e[..., ::4, ::4]
    err = rel(got, ref)
    assert 1e-7 < err < 3e-3, err            # > 1e-7: the fp16 path really ran


@pytest.mark.parametrize("name", ["colorization", "inpainting"])
def test_adm_sampler_fp16_torso_psnr_bar(hip, name, golden_dir):
    """The same 56-iteration time-travel run with `convert_to_fp16()`: against the reference's fp32 golden the
    restored images agree to the bar of SURVEY.md section 8c -- |PSNR(engine, x_orig) - PSNR(reference, x_orig)|
    <= 0.1 dB -- and image-to-image PSNR is high (the loop is not chaotic under fp16-class perturbations)."""
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from oracle import cases, sampler, schedule
    cfg, sd = cases.adm_net("mid")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
    y = cases.make_operator(name, cfg.data.image_size).A(x_orig)
    op = engine_operator(name, cfg.data.image_size)
    m = build(cfg, sd)
    m.convert_to_fp16()
    xs, _ = ddnm_diffusion(x_T.cuda(), m, cases.betas().cuda(), 0.85, op, y.cuda(), cls_fn=None, classes=None,
                           config=cfg, noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    got = xs[0].cpu()
    ref = torch.from_numpy(np.load(f"{golden_dir}/adm_forward.npz")[f"mid_{name}_x"])
    err = rel(got, ref)
    assert 1e-7 < err < 2e-2, err
    assert (sampler.psnr(got, x_orig) - sampler.psnr(ref, x_orig)).abs().max().item() <= 0.1
    assert sampler.psnr(got, ref).min().item() > 35.0


@pytest.mark.parametrize("two_streams", [False, True])
def test_adm_graph_replay_matches_eager(hip, two_streams):
    """The captured-graph forward (ddnm_amd/graph.py; optionally the two half-batches as concurrent branches) returns
    what the eager forward returns, replay after replay, for changing inputs."""
    from oracle import cases
    cfg, sd = cases.adm_net("mid")
    m = build(cfg, sd)
    m.convert_to_fp16()
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(4, 3, 64, 64, generator=g).cuda() for _ in range(3)]
    ts = [torch.tensor([990.0, 500.0, 20.0, 0.0]).cuda(), torch.full((4,), 430.0).cuda(), torch.full((4,), 10.0).cuda()]
    eager = [m(x, t).clone() for x, t in zip(xs, ts)]
    m.enable_graphs(two_streams=two_streams)
    for rep in range(2):
        for x, t, want in zip(xs, ts, eager):
            got = m(x, t)
            torch.cuda.synchronize()
            if two_streams:        # half batches take other split-K plans: fp32 summation order, then fp16 roundings
                assert rel(got, want) < 2e-3
            else:
                assert torch.equal(got, want)
    m.disable_graphs()
