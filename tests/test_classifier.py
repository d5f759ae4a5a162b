"""Classifier guidance (BASELINE config 5): oracle restatement (torch autograd on CPU) pinned to goldens of the
reference's EncoderUNetModel + cond_fn; HIP engine forward and explicit backward against both."""
import json

import numpy as np
import pytest
import torch

from oracle import cases, classifier as OC, weights
from tests.helpers import rel

KINDS = {"small": dict(image_size=32, classifier_depth=1, classifier_attention_resolutions="16,8"),
         "mid": dict(image_size=64, classifier_depth=1), "full": dict()}


def _inputs(r):
    g = torch.Generator().manual_seed(cases.SEED + 11)
    return torch.randn(2, 3, r, r, generator=g), torch.tensor([430.0, 10.0]), torch.tensor([951, 17])


def _engine(cc, sd):
    from ddnm_amd.guided_diffusion.classifier import EncoderUNetModel
    mult = {256: (1, 1, 2, 2, 4, 4), 64: (1, 2, 3, 4), 32: (1, 2)}[cc.image_size]
    ads = tuple(cc.image_size // int(r) for r in cc.classifier_attention_resolutions.split(","))
    m = EncoderUNetModel(image_size=cc.image_size, in_channels=3, model_channels=cc.classifier_width, out_channels=1000,
                         num_res_blocks=cc.classifier_depth, attention_resolutions=ads, channel_mult=mult,
                         num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True, pool="attention")
    m.load_state_dict(sd)
    return m


def test_classifier_keys_golden(golden_dir):
    keys = json.load(open(f"{golden_dir}/classifier_state_dict_keys.json"))
    mine = weights.classifier_shapes(weights.classifier_config())
    assert [[k, list(v)] for k, v in mine.items()] == keys
    assert sum(int(np.prod(v)) for v in mine.values()) == 54_096_360          # 54.10 M
    from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier
    cc = weights.classifier_config()
    eng = create_classifier(**{k: getattr(cc, k) for k in classifier_defaults()})
    assert [[k, list(v)] for k, v in eng.state_dict_shapes().items()] == keys


@pytest.mark.parametrize("kind", ["small", "mid"])
def test_oracle_classifier_golden(kind, golden_dir):
    g = np.load(f"{golden_dir}/classifier.npz")
    cc = weights.classifier_config(**KINDS[kind])
    sd = weights.classifier_state_dict(cc)
    x, t, y = _inputs(cc.image_size)
    with torch.no_grad():
        logits = OC.forward(sd, cc, x, t)
    assert torch.equal(logits, torch.from_numpy(g[f"{kind}_logits"]))
    grad = OC.cond_fn(sd, cc, x, t, y)
    assert rel(grad, torch.from_numpy(g[f"{kind}_grad"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["small", "mid", "full"])
def test_engine_classifier_logits_and_gradient(hip, kind, golden_dir):
    g = np.load(f"{golden_dir}/classifier.npz")
    cc = weights.classifier_config(**KINDS[kind])
    sd = weights.classifier_state_dict(cc)
    x, t, y = _inputs(cc.image_size)
    m = _engine(cc, sd)
    logits = m(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert rel(logits, torch.from_numpy(g[f"{kind}_logits"])) < 2e-5
    grad = m.log_prob_grad(x.cuda(), t.cuda(), y.cuda()).cpu()
    ref = torch.from_numpy(g[f"{kind}_grad"])
    got = grad if kind != "full" else grad[..., ::4, ::4]
    assert got.shape == ref.shape
    assert rel(got, ref) < 2e-4
    assert abs(grad.double().norm().item() - float(g[f"{kind}_grad_norm"][0])) < 2e-4 * float(g[f"{kind}_grad_norm"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,gen", [("small", "h16"), ("mid", "h16"), ("full", "h16"), ("mid", "gen1"), ("full", "gen1")])
def test_engine_classifier_fp16_operands(hip, kind, gen, golden_dir, monkeypatch):
    """`classifier.convert_to_fp16()` (imagenet_256_cc.yml: classifier_use_fp16 true) against the fp32 autograd goldens of
    the reference at the half-precision bar of SURVEY.md section 8c.  "h16" (default since round 5): the fp16-ACTIVATION
    path -- activations and activation gradients fp16 NHWC, ddnm_conv16 forward and data-gradient convolutions, fused
    attention forward / backward, fp16 GroupNorm backward; "gen1" (DDNM_CLS_GEN1=1): fp32 tensors, fp16 MFMA operands
    for the convolutions only."""
    monkeypatch.setenv("DDNM_CLS_GEN1", "1" if gen == "gen1" else "0")
    g = np.load(f"{golden_dir}/classifier.npz")
    cc = weights.classifier_config(**KINDS[kind])
    x, t, y = _inputs(cc.image_size)
    m = _engine(cc, weights.classifier_state_dict(cc))
    m.convert_to_fp16()
    assert m.h16 == (gen == "h16")
    logits = m(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    err = rel(logits, torch.from_numpy(g[f"{kind}_logits"]))
    assert 1e-7 < err < 3e-3, err                    # > 1e-7: the fp16 kernels really ran
    grad = m.log_prob_grad(x.cuda(), t.cuda(), y.cuda()).cpu()
    ref = torch.from_numpy(g[f"{kind}_grad"])
    got = grad if kind != "full" else grad[..., ::4, ::4]
    assert rel(got, ref) < 1e-2, rel(got, ref)
    cos = (got.double() * ref.double()).sum() / (got.double().norm() * ref.double().norm())
    assert cos > 0.9999


@pytest.mark.gpu
@pytest.mark.parametrize("wscale,xscale", [(1.0, 1.0), (1.6, 3.0), (0.5, 0.3)])
def test_fp16_activation_engine_tracks_the_fp32_engine_off_the_golden_point(hip, wscale, xscale):
    """The fp16-activation engine against the engine's own fp32 path (itself <= 2e-4 from the reference's autograd) away from
    the golden inputs: weights and inputs scaled up / down (activation and gradient magnitudes move by orders of magnitude
    through the 2^10 gradient pre-scale), extreme timesteps, B = 5 (ragged against every tile size).  Finite everywhere,
    gradient within the half-precision bar and aligned."""
    cc = weights.classifier_config(**KINDS["mid"])
    sd = {k: (v * wscale if v.dim() > 1 else v) for k, v in weights.classifier_state_dict(cc).items()}
    g = torch.Generator().manual_seed(23)
    r = cc.image_size
    x = (torch.randn(5, 3, r, r, generator=g) * xscale).cuda()
    t = torch.tensor([0.0, 999.0, 430.0, 10.0, 750.0]).cuda()
    y = torch.tensor([951, 0, 999, 17, 500]).cuda()
    ref = _engine(cc, sd)
    want_l, want_g = ref(x, t).float().cpu(), ref.log_prob_grad(x, t, y).float().cpu()
    m = _engine(cc, sd)
    m.convert_to_fp16()
    assert m.h16
    got_l, got_g = m(x, t).float().cpu(), m.log_prob_grad(x, t, y).float().cpu()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got_l).all()) and bool(torch.isfinite(got_g).all())
    assert rel(got_l, want_l) < 5e-3, rel(got_l, want_l)
    for i in range(5):                                   # per image: one bad sample must not hide in the batch norm
        e = rel(got_g[i], want_g[i])
        cos = (got_g[i].double() * want_g[i].double()).sum() / (got_g[i].double().norm() * want_g[i].double().norm())
        assert e < 2e-2 and cos > 0.9998, (i, e, float(cos))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,G", [("mid", 3, 4), ("full", 2, 3)])
def test_grouped_pass_shares_the_timestep_independent_prefix(hip, monkeypatch, kind, n, G):
    """A grouped guidance pass is G copies of the same n images with G timesteps (svd_ddnm.py::_GuidanceAhead).  The input
    convolution and the first ResBlock's in_layers do not see the timestep, so the fp16-activation engine evaluates them once
    for the n distinct images (`replicas=G`) -- bit-identical to evaluating every copy (classifier.SHARE_PREFIX = False), and each
    copy equals the un-grouped call at its own timestep."""
    from ddnm_amd.guided_diffusion.classifier import make_cond_fn
    cc = weights.classifier_config(**KINDS[kind])
    m = _engine(cc, weights.classifier_state_dict(cc))
    m.convert_to_fp16()
    fn = make_cond_fn(m, 1.0)
    g = torch.Generator().manual_seed(5)
    r = cc.image_size
    x = torch.randn(n, 3, r, r, generator=g).cuda()
    xg = torch.cat([x] * G, 0)
    tg = torch.tensor([float(10 + 240 * k) for k in range(G) for _ in range(n)]).cuda()
    yg = torch.full((G * n,), 951).cuda()
    shared = fn(xg, tg, yg, replicas=G)
    from ddnm_amd.guided_diffusion import classifier as _cl
    monkeypatch.setattr(_cl, "SHARE_PREFIX", False)
    plain = fn(xg, tg, yg, replicas=G)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(shared).all()) and torch.equal(shared, plain)
    assert torch.equal(shared, fn(xg, tg, yg))
    one = fn(x, tg[n:2 * n].contiguous(), yg[:n])              # the second timestep alone: other launch plans (batch n)
    assert rel(shared[n:2 * n], one) < 5e-3


@pytest.mark.gpu
def test_gn_backward_kernel(hip):
    """GroupNorm(+FiLM)+SiLU backward against torch autograd on CPU, incl. the half-resolution (avg-pool) mapping."""
    import torch.nn.functional as F
    from ddnm_amd import _lib, ops
    from ddnm_amd._lib import check
    gen = torch.Generator().manual_seed(2)
    B, C, H = 2, 128, 16
    x = torch.randn(B, C, H, H, generator=gen, requires_grad=True)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)
    dA_half = torch.randn(B, C, H // 2, H // 2, generator=gen)
    add_half = torch.randn(B, C, H // 2, H // 2, generator=gen)
    a = F.avg_pool2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), 2, 2)
    (a * dA_half).sum().backward()
    ref = x.grad + F.interpolate(add_half, scale_factor=2, mode="nearest") * 0.25
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()      # noqa: E731
    ws = ops.GroupNormWorkspace("cuda", B, C, B * ops.gn_nchunk(H * H, C) * 64)
    keep = {}
    ops.group_norm_affine(nh(x), None, gamma.cuda(), beta.cuda(), 1e-5, ws, keep=keep)
    L = _lib.lib()
    nchunk = L.ddnm_gn_bwd_nchunk(H * H, C)
    partial = torch.empty(B * nchunk * 64, dtype=torch.float64, device="cuda")
    coef = torch.empty(B * 64, device="cuda")
    dx = torch.empty(B, H, H, C, device="cuda")
    xa, da, ad = nh(x), nh(dA_half), nh(add_half)
    check(L.ddnm_gn_bwd_f32(xa.data_ptr(), da.data_ptr(), 1, keep["scale"].data_ptr(), keep["shift"].data_ptr(),
                            keep["mean_rstd"].data_ptr(), 1, ad.data_ptr(), 1, B, H, H, C, 32, partial.data_ptr(), nchunk,
                            coef.data_ptr(), dx.data_ptr(), ops._stream()), "gn_bwd")
    torch.cuda.synchronize()
    assert rel(dx.cpu().permute(0, 3, 1, 2), ref) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,ups", [(128, 16, True), (128, 32, False), (384, 8, False), (512, 16, True)])
def test_gn_backward_kernel_h16(hip, C, H, ups):
    """ddnm_gn_bwd_h16 (fp16 activations and gradients, fp32 arithmetic) against torch autograd on the same fp16-rounded
    inputs; the result is rounded to fp16 once."""
    import torch.nn.functional as F
    from ddnm_amd import _lib, ops
    from ddnm_amd._lib import check
    gen = torch.Generator().manual_seed(5)
    B = 2
    r16 = lambda t: t.half().float()       # noqa: E731
    x = r16(torch.randn(B, C, H, H, generator=gen)).requires_grad_(True)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)
    Hd = H // 2 if ups else H
    dA = r16(torch.randn(B, C, Hd, Hd, generator=gen))
    add = r16(torch.randn(B, C, Hd, Hd, generator=gen))
    a = F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5))
    if ups:
        a = F.avg_pool2d(a, 2, 2)
    (a * dA).sum().backward()
    ref = x.grad + (F.interpolate(add, scale_factor=2, mode="nearest") * 0.25 if ups else add)
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().half().cuda()      # noqa: E731
    ws = ops.GroupNormWorkspace("cuda", B, C, B * ops.gn_nchunk(H * H, C) * 64)
    keep = {}
    xa = nh(x)
    ops.group_norm_affine(xa, None, gamma.cuda(), beta.cuda(), 1e-5, ws, keep=keep)
    L = _lib.lib()
    nchunk = L.ddnm_gn_bwd_nchunk(H * H, C)
    partial = torch.empty(B * nchunk * 64, dtype=torch.float64, device="cuda")
    coef = torch.empty(B * 64, device="cuda")
    dx = torch.empty(B, H, H, C, device="cuda", dtype=torch.float16)
    da, ad = nh(dA), nh(add)
    check(L.ddnm_gn_bwd_h16(xa.data_ptr(), da.data_ptr(), int(ups), keep["scale"].data_ptr(), keep["shift"].data_ptr(),
                            keep["mean_rstd"].data_ptr(), 1, ad.data_ptr(), int(ups), B, H, H, C, 32, partial.data_ptr(),
                            nchunk, coef.data_ptr(), dx.data_ptr(), ops._stream()), "gn_bwd_h16")
    torch.cuda.synchronize()
    err = rel(dx.float().cpu().permute(0, 3, 1, 2), ref)
    assert err < 6e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C", [(2, 64, 128), (1, 256, 256), (2, 1024, 64), (3, 128, 192)])
def test_attention16_backward(hip, B, T, C):
    """ddnm_attn16_d64_lse + ddnm_attn16_d64_bwd against torch autograd of QKVAttentionLegacy (unet.py:328-354) on the
    same fp16-rounded qkv: output, log-sum-exp and the three input gradients."""
    import math
    from ddnm_amd import ops
    gen = torch.Generator().manual_seed(31)
    nh = C // 64
    qkv = (0.8 * torch.randn(B, T, 3 * C, generator=gen)).half()
    dO = torch.randn(B, T, C, generator=gen).half()
    q32 = qkv.float().requires_grad_(True)
    v = q32.view(B, T, nh, 3, 64)
    q, k, vv = v[:, :, :, 0], v[:, :, :, 1], v[:, :, :, 2]                 # [B, T, nh, 64]
    logits = torch.einsum("bthd,bshd->bhts", q, k) / 8.0
    ref_lse = torch.logsumexp(logits, dim=-1) / math.log(2.0)                # base 2, [B, nh, T]
    o_ref = torch.einsum("bhts,bshd->bthd", torch.softmax(logits, dim=-1), vv).reshape(B, T, C)
    (o_ref * dO.float()).sum().backward()
    side = int(math.isqrt(T))
    H, W = (side, side) if side * side == T else (T // 8, 8)
    qd = qkv.view(B, H, W, 3 * C).cuda()
    lse = torch.empty(B, nh, T, device="cuda")
    o = ops.attn16(qd, C, lse=lse)
    dq = ops.attn16_bwd(qd, o, dO.view(B, H, W, C).cuda(), lse)
    torch.cuda.synchronize()
    assert rel(o.float().cpu().view(B, T, C), o_ref.detach()) < 2e-3
    assert (lse.cpu() - ref_lse.detach()).abs().max().item() < 2e-3
    g_ref = q32.grad.view(B, T, nh, 3, 64)
    g_got = dq.float().cpu().view(B, T, nh, 3, 64)
    for i, name in enumerate("qkv"):
        err = rel(g_got[:, :, :, i], g_ref[:, :, :, i])
        assert err < 4e-3, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 64, 256), (256, 64, 1024), (16, 24, 40)])
def test_bgemm_transposed_a(hip, M, N, K):
    from ddnm_amd import ops
    gen = torch.Generator().manual_seed(4)
    A = torch.randn(3, K, M, generator=gen)
    Bm = torch.randn(3, K, N, generator=gen)
    C = torch.empty(3, M, N, device="cuda")
    ops.bgemm(A.cuda(), Bm.cuda(), C, M, N, K, lda=M, ldb=N, ldc=N, transb=False, transa=True, batch=3,
              sA=(K * M, 0), sB=(K * N, 0), sC=(M * N, 0))
    torch.cuda.synchronize()
    assert rel(C.cpu(), torch.bmm(A.transpose(1, 2), Bm)) < 2e-6


@pytest.mark.gpu
def test_guided_sampler_small(hip):
    """ddnm_diffusion with a class-conditional ADM net + cls_fn (reference quirks: class 951 for every image,
    guidance evaluated on the INITIAL noise) against the oracle loop with the oracle cond_fn."""
    from ddnm_amd.functions.svd_ddnm import class_num, ddnm_diffusion
    from ddnm_amd.guided_diffusion.classifier import make_cond_fn
    from ddnm_amd.guided_diffusion.unet import create_model
    from oracle import schedule, unet_adm
    from tests.helpers import engine_operator
    cfg, sd = cases.adm_net("small")                       # class-conditional, 32 px
    cc = weights.classifier_config(**KINDS["small"])
    csd = weights.classifier_state_dict(cc)
    cfg.time_travel.T_sampling = 6
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, 6)
    orc = cases.make_operator("cs_walshhadamard", 32)
    y = orc.A(x_orig)
    betas = cases.betas()
    # oracle loop (functions/svd_ddnm.py:41-65 with cls_fn)
    net = unet_adm.Net(sd, cfg)
    xt, x0 = x_T.clone(), None
    cls = torch.full((2,), class_num, dtype=torch.long)
    times = schedule.jump_times(6, 1, 1)
    skip = 1000 // 6
    for k, (i, j) in enumerate(zip(times[:-1], times[1:])):
        i, j = i * skip, (j * skip if j >= 0 else -1)
        at, atn = schedule.alpha_bar(betas, i), schedule.alpha_bar(betas, j)
        t = torch.ones(2) * i
        et = net(xt, t, cls)[:, :3]
        et = et - (1 - at).sqrt() * OC.cond_fn(csd, cc, x_T, t, cls)
        x0 = (xt - et * (1 - at).sqrt()) / at.sqrt()
        x0h = x0 - orc.A_pinv(orc.A(x0.reshape(2, -1)) - y).reshape(x0.shape)
        xt = atn.sqrt() * x0h + (1 - atn).sqrt() * 0.85 * tape[k] + (1 - atn).sqrt() * ((1 - 0.85 ** 2) ** 0.5) * et
    model = create_model(**vars(cfg.model))
    model.load_state_dict(sd)
    clf = _engine(cc, csd)
    xs, _ = ddnm_diffusion(x_T.cuda(), model, betas.cuda(), 0.85, engine_operator("cs_walshhadamard", 32), y.cuda(),
                           cls_fn=make_cond_fn(clf, 1.0), classes=None, config=cfg, noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    assert rel(xs[0], xt) < 2e-4


@pytest.mark.gpu
def test_guidance_stream_equals_serial_order(hip, monkeypatch):
    """The guidance term is evaluated on the initial noise (svd_ddnm.py:49-52), so the engine runs it on a second HIP
    stream one reverse step ahead of the UNet (ddnm_amd/functions/svd_ddnm.py::_GuidanceAhead).  Same launches, same
    inputs: the restored images are bit-identical to the serial order (DDNM_CLS_OVERLAP=0), with time travel (the
    schedule revisits timesteps) and through DDNM+ as well."""
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion, ddnm_plus_diffusion
    from ddnm_amd.guided_diffusion.classifier import make_cond_fn
    from ddnm_amd.guided_diffusion.unet import create_model
    from oracle import schedule
    from tests.helpers import engine_operator
    cfg, sd = cases.adm_net("small")
    cc = weights.classifier_config(**KINDS["small"])
    clf = _engine(cc, weights.classifier_state_dict(cc))
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 8, 2, 2
    n_it = len(schedule.jump_times(8, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 3, n_it)
    op = engine_operator("colorization", 32)
    y = cases.make_operator("colorization", 32).A(x_orig).cuda()
    model = create_model(**vars(cfg.model))
    model.load_state_dict(sd)
    outs = {}
    monkeypatch.setenv("DDNM_CLS_GROUP", "1")       # step-by-step evaluation: the launches of the serial order
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("DDNM_CLS_OVERLAP", mode)
        xs, x0s = ddnm_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y, cls_fn=make_cond_fn(clf, 2.0),
                                 classes=None, config=cfg, noise=[n.cuda() for n in tape], return_cpu=False)
        xp, _ = ddnm_plus_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y, 0.1, cls_fn=make_cond_fn(clf, 2.0),
                                    classes=None, config=cfg, noise=[n.cuda() for n in tape], return_cpu=False)
        torch.cuda.synchronize()
        outs.setdefault(mode, []).append((xs[0].clone(), x0s[0].clone(), xp[0].clone()))
    a, b, c = outs["1"][0], outs["0"][0], outs["1"][1]
    for u, v, w in zip(a, b, c):
        assert bool(torch.isfinite(u).all())
        assert torch.equal(u, v) and torch.equal(u, w)
    # default mode: the guidance terms of two consecutive steps in ONE pass over [x; x] (round 4).  Same arithmetic per
    # sample; a batch of 2n picks other split-K plans than n, i.e. another fp32 summation order
    monkeypatch.delenv("DDNM_CLS_GROUP")
    monkeypatch.setenv("DDNM_CLS_OVERLAP", "1")
    xs, x0s = ddnm_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y, cls_fn=make_cond_fn(clf, 2.0),
                             classes=None, config=cfg, noise=[n.cuda() for n in tape], return_cpu=False)
    xp, _ = ddnm_plus_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y, 0.1, cls_fn=make_cond_fn(clf, 2.0),
                                classes=None, config=cfg, noise=[n.cuda() for n in tape], return_cpu=False)
    torch.cuda.synchronize()
    for got, want in ((xs[0], b[0]), (x0s[0], b[1]), (xp[0], b[2])):
        assert rel(got, want) < 2e-5, rel(got, want)
