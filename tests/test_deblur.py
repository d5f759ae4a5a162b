"""Separable blur operators (deblur_uni / deblur_gauss / deblur_aniso): oracle vs reference goldens (CPU),
HIP engine vs both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import cases, sampler, unet_celeba
from tests.helpers import engine_operator, rel

NAMES = ["deblur_uni", "deblur_gauss", "deblur_aniso"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_deblur_golden(name, golden_dir):
    g = np.load(f"{golden_dir}/deblur.npz")
    op = cases.make_operator(name, 64)
    x = cases.operator_input(64, 2)
    y = op.A(x)
    assert rel(y, torch.from_numpy(g[f"{name}_y"])) < 1e-6
    assert rel(op.A_pinv(y), torch.from_numpy(g[f"{name}_pinv"])) < 1e-5


def test_oracle_deblur_sampler_golden(golden_dir):
    g = np.load(f"{golden_dir}/deblur.npz")
    cfg, sd = cases.celeba_net("small")
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, 12)
    op = cases.make_operator("deblur_gauss", cfg.data.image_size)
    x, x0 = sampler.ddnm_diffusion(x_T.clone(), unet_celeba.Net(sd, cfg), cases.betas(), 0.85, op, op.A(x_orig), tape,
                                   T_sampling=12)
    assert rel(x, torch.from_numpy(g["sampler_gauss_x"])) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("d", [32, 64, 256])
def test_engine_deblur_operator(hip, name, d, golden_dir):
    orc, eng = cases.make_operator(name, d), engine_operator(name, d)
    x = cases.operator_input(d, 2)
    y_o = orc.A(x)
    y_e = eng.A(x.cuda())
    p_o, p_e = orc.A_pinv(y_o), eng.A_pinv(y_o.cuda())
    torch.cuda.synchronize()
    assert rel(y_e, y_o) < 5e-6 and rel(p_e, p_o) < 5e-5
    if d == 64:
        # The golden was produced on another host CPU: LAPACK's SVD basis of (near-)degenerate singular
        # subspaces and the tie order of the descending sort differ between machines, and the reference's
        # 3x tiling quirk makes A depend on them -- the reference itself is only reproducible to ~1e-4 here.
        g = np.load(f"{golden_dir}/deblur.npz")
        assert rel(y_e, torch.from_numpy(g[f"{name}_y"])) < 1e-3


@pytest.mark.gpu
def test_engine_deblur_sampler_vs_reference_golden(hip, golden_dir):
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from ddnm_amd.guided_diffusion.models import Model
    g = np.load(f"{golden_dir}/deblur.npz")
    cfg, sd = cases.celeba_net("small")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 12, 1, 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, 12)
    y = cases.make_operator("deblur_gauss", cfg.data.image_size).A(x_orig)
    model = Model(cfg)
    model.load_state_dict(sd)
    op = engine_operator("deblur_gauss", cfg.data.image_size)
    xs, x0s = ddnm_diffusion(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y.cuda(), cls_fn=None, classes=None,
                             config=cfg, noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    assert rel(xs[0], torch.from_numpy(g["sampler_gauss_x"])) < 5e-3      # see the SVD note above
