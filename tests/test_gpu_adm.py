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
