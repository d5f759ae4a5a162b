"""DDNM+ (sigma_y > 0): oracle restatement pinned to the reference's ddnm_plus_diffusion goldens (CPU),
and the HIP engine against both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import cases, sampler, schedule, unet_celeba
from tests.helpers import engine_operator, rel

PLUS_OPS = ["sr_averagepooling", "colorization", "inpainting", "cs_walshhadamard", "denoising"]


BLUR_OPS = ["deblur_uni", "deblur_gauss"]           # the blur operators that define Lambda (svd_operators.py:1016-1091)


def _case(name, golden_dir):
    g = np.load(f"{golden_dir}/" + ("ddnm_plus_deblur.npz" if name in BLUR_OPS else "ddnm_plus_small.npz"))
    cfg, sd = cases.celeba_net("small")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
    y = torch.from_numpy(g[f"{name}_y"])
    return cfg, sd, x_T, tape, y, torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_x0"])


@pytest.mark.parametrize("name", PLUS_OPS)
def test_oracle_ddnm_plus_golden(name, golden_dir):
    cfg, sd, x_T, tape, y, gx, gx0 = _case(name, golden_dir)
    op = cases.make_operator(name, cfg.data.image_size)
    x, x0 = sampler.ddnm_plus_diffusion(x_T.clone(), unet_celeba.Net(sd, cfg), cases.betas(), 0.85, op, y, 0.2, tape,
                                        T_sampling=20, travel_length=2, travel_repeat=2)
    assert rel(x, gx) < 2e-5 and rel(x0, gx0) < 2e-5


@pytest.mark.parametrize("name", BLUR_OPS)
def test_oracle_deblur_lambda_golden(name, golden_dir):
    """Operator-level Lambda / Lambda_noise of the reference's Deblurring in three threshold regimes, and the
    whole DDNM+ loop."""
    g = np.load(f"{golden_dir}/ddnm_plus_deblur.npz")
    cfg, sd, x_T, tape, y, gx, gx0 = _case(name, golden_dir)
    d = cfg.data.image_size
    op = cases.make_operator(name, d)
    gen = torch.Generator().manual_seed(3)
    v, e = torch.randn(2, 3 * d * d, generator=gen), torch.randn(2, 3 * d * d, generator=gen)
    for tag, tn, sy in (("hi", 990, 0.4), ("mid", 500, 0.4), ("lo", 10, 0.4)):
        atn = schedule.alpha_bar(cases.betas(), tn)
        a, st = atn.sqrt(), (1 - atn).sqrt()
        assert rel(op.Lambda(v.clone(), a, sy, st, 0.85), torch.from_numpy(g[f"{name}_lambda_{tag}"])) < 1e-5
        assert rel(op.Lambda_noise(v.clone(), a, sy, st, 0.85, e.clone()),
                   torch.from_numpy(g[f"{name}_lambda_noise_{tag}"])) < 1e-5
    x, x0 = sampler.ddnm_plus_diffusion(x_T.clone(), unet_celeba.Net(sd, cfg), cases.betas(), 0.85, op, y, 0.2, tape,
                                        T_sampling=20, travel_length=2, travel_repeat=2)
    assert rel(x, gx) < 1e-4 and rel(x0, gx0) < 1e-4


def test_spectral_coefficients_regimes():
    from ddnm_amd.functions.svd_operators import spectral_coefficients as sc
    from oracle.operators import _coef
    for s in (0.0, 0.25, 0.577, 1.0):
        for (a, sy, st) in [(0.9, 0.2, 0.1), (0.9, 0.2, 0.5), (0.1, 0.4, 0.99), (1.0, 0.2, 0.0), (0.5, 0.0, 0.3)]:
            assert sc(s, a, sy, st, 0.85) == _coef(s, a, sy, st, 0.85)
    lam, d1, d2 = sc(0.5, 0.9, 0.2, 0.1, 0.85)         # sigma_t < a*sigma_y/s = 0.36
    assert abs(lam - 0.5 * 0.1 * (1 - 0.85 ** 2) ** 0.5 / 0.9 / 0.2) < 1e-12 and d1 == 0.1 * 0.85 and d2 == 0.0
    lam, d1, d2 = sc(0.5, 0.9, 0.2, 0.5, 0.85)         # above the threshold
    assert lam == 1.0 and abs(d1 - (0.25 - 0.81 * 0.04 / 0.25) ** 0.5) < 1e-12 and d2 == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", PLUS_OPS + BLUR_OPS)
def test_engine_lambda_and_lambda_noise(hip, name):
    """Operator-level parity of Lambda / Lambda_noise in the three regimes of the threshold a*sigma_y/s."""
    d, B = 32, 2
    orc, eng = cases.make_operator(name, d), engine_operator(name, d)
    g = torch.Generator().manual_seed(3)
    v, e = torch.randn(B, 3 * d * d, generator=g), torch.randn(B, 3 * d * d, generator=g)
    betas = cases.betas()
    for tn, sy in [(990, 0.4), (500, 0.4), (10, 0.4), (-1, 0.4), (500, 0.05)]:
        atn = schedule.alpha_bar(betas, tn)
        a, st = atn.sqrt(), (1 - atn).sqrt()
        lo, le = orc.Lambda(v.clone(), a, sy, st, 0.85), eng.Lambda(v.cuda(), a, sy, st, 0.85)
        no, ne = orc.Lambda_noise(v.clone(), a, sy, st, 0.85, e.clone()), eng.Lambda_noise(v.cuda(), a, sy, st, 0.85, e.cuda())
        torch.cuda.synchronize()
        tol = 1e-5 if name in BLUR_OPS else 3e-6           # blur: four 32-term GEMMs instead of site-local sums
        assert rel(le.reshape(B, -1), lo.reshape(B, -1)) < tol, (name, tn, sy)
        if no.abs().max() > 0:
            assert rel(ne.reshape(B, -1), no.reshape(B, -1)) < tol, (name, tn, sy)
        else:
            assert ne.abs().max().item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", PLUS_OPS + BLUR_OPS)
def test_engine_ddnm_plus_vs_reference_golden(hip, name, golden_dir):
    from ddnm_amd.functions.svd_ddnm import ddnm_plus_diffusion
    from ddnm_amd.guided_diffusion.models import Model
    cfg, sd, x_T, tape, y, gx, gx0 = _case(name, golden_dir)
    model = Model(cfg)
    model.load_state_dict(sd)
    op = engine_operator(name, cfg.data.image_size)
    xs, x0s = ddnm_plus_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y.cuda(), 0.2, cls_fn=None,
                                  classes=None, config=cfg, noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    # blur goldens come from another host's LAPACK basis (see tests/test_deblur.py): reproducible to ~1e-3 only
    tol = 5e-3 if name in BLUR_OPS else 2e-4
    assert rel(xs[0], gx) < tol and rel(x0s[0], gx0) < tol
    if name in BLUR_OPS:        # tight check against the oracle evaluated on THIS host
        orc = cases.make_operator(name, cfg.data.image_size)
        from oracle import unet_celeba as U
        x, x0 = sampler.ddnm_plus_diffusion(x_T.clone(), U.Net(sd, cfg), cases.betas(), 0.85, orc, y, 0.2, tape,
                                            T_sampling=20, travel_length=2, travel_repeat=2)
        assert rel(xs[0], x) < 3e-4 and rel(x0s[0], x0) < 3e-4


@pytest.mark.gpu
def test_srconv_has_no_lambda(hip):
    from ddnm_amd.functions.svd_operators import SRConv, bicubic_kernel
    op = SRConv(bicubic_kernel(4), 3, 32, "cuda", stride=4)
    with pytest.raises(NotImplementedError):          # same as the reference (svd_operators.py:93-97)
        op.Lambda(torch.zeros(1, 3 * 32 * 32, device="cuda"), 0.9, 0.2, 0.3, 0.85)
