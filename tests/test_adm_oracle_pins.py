"""CPU tests: ADM (ImageNet) UNet restatement against goldens produced by the real
`script_util.create_model(...)` model of the reference (fp32, seeded weights)."""
import json

import numpy as np
import pytest
import torch

from oracle import cases, schedule, sampler, unet_adm, weights
from tests.helpers import rel


def test_adm_state_dict_keys_golden(golden_dir):
    keys = json.load(open(f"{golden_dir}/adm_state_dict_keys.json"))
    mine = weights.adm_shapes(weights.adm_config())
    assert [[k, list(v)] for k, v in mine.items()] == keys
    assert sum(int(np.prod(v)) for v in mine.values()) == 552_814_086      # 552.81 M


def test_engine_plan_consumes_reference_state_dict(golden_dir):
    from ddnm_amd.guided_diffusion.unet import create_model
    keys = json.load(open(f"{golden_dir}/adm_state_dict_keys.json"))
    cfg = weights.adm_config()
    m = create_model(**vars(cfg.model), )
    assert [[k, list(v)] for k, v in m.state_dict_shapes().items()] == keys
    cc = weights.adm_config(class_cond=True)
    mc = create_model(**vars(cc.model))
    assert "label_emb.weight" in mc.state_dict_shapes() and len(mc.state_dict_shapes()) == 567


@pytest.mark.parametrize("kind,batch", [("small", 2), ("mid", 2)])
def test_adm_forward_golden(kind, batch, golden_dir):
    g = np.load(f"{golden_dir}/adm_forward.npz")
    cfg, sd = cases.adm_net(kind)
    x, t, y = cases.adm_forward_inputs(cfg, batch)
    e = unet_adm.Net(sd, cfg)(x, t, y)
    assert torch.equal(e, torch.from_numpy(g[f"{kind}_eps"])), "restatement must be bit-identical on CPU"


@pytest.mark.parametrize("name", ["colorization", "inpainting"])
def test_adm_sampler_golden(name, golden_dir):
    g = np.load(f"{golden_dir}/adm_forward.npz")
    cfg, sd = cases.adm_net("mid")
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
    op = cases.make_operator(name, cfg.data.image_size)
    y = op.A(x_orig)
    x, x0 = sampler.ddnm_diffusion(x_T.clone(), unet_adm.Net(sd, cfg), cases.betas(), 0.85, op, y, tape,
                                   T_sampling=20, travel_length=2, travel_repeat=2)
    assert x.shape[1] == 3                         # learn_sigma head (6 channels) reduced to eps (svd_ddnm.py:54-55)
    assert rel(x, torch.from_numpy(g[f"mid_{name}_x"])) < 1e-5
    assert rel(x0, torch.from_numpy(g[f"mid_{name}_x0"])) < 1e-5
