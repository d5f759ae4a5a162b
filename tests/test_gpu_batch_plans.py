"""The launch plans bench.py times are the launch plans that are parity-checked.

`plan16` (ddnm_amd/csrc/conv16.hip) picks tile size, split-K factor and workgroup order from the batch size, and the
classifier's kernels do the same, so a forward at B = 1 executes other plans than the per-GPU batches of BASELINE
configs[2..4] (4 / 4 / 8 images).  The reference loop is batch-agnostic (functions/svd_ddnm.py:36-65,
guided_diffusion/unet.py:635-664): every image of a batch must come out as it does alone.

  * full 552 M-parameter ADM UNet, `convert_to_fp16()`, B = 4 and B = 8: image 0 against the reference's fp32 golden
    (adm_forward.npz `full_eps`, <= 3e-3) and every image against the same image run at B = 1 (<= 2e-3: both are
    fp16-class roundings of the same fp32 result; the plans differ in fp32 summation order);
  * full classifier, `convert_to_fp16()`, `cond_fn` at B = 8 against B = 1 per image, image 0 against the reference's
    autograd golden;
  * (tests/test_gpu_full_configs.py: `c3b4` = configs[2] at B = 4 through the sampling loop vs the reference).
"""
import numpy as np
import pytest
import torch

from tests.helpers import rel

pytestmark = pytest.mark.gpu

T_ROWS = [430.0, 990.0, 0.0, 10.0, 500.0, 250.0, 750.0, 100.0]


def _adm_batch(cfg, B):
    from oracle import cases
    x1, _, _ = cases.adm_forward_inputs(cfg, 1)          # image 0 = the input of the reference golden
    g = torch.Generator().manual_seed(cases.SEED + 77)
    r = cfg.model.image_size
    x = torch.cat([x1, torch.randn(B - 1, 3, r, r, generator=g)], 0) if B > 1 else x1
    return x, torch.tensor(T_ROWS[:B])


@pytest.fixture(scope="module")
def adm_full_fp16():
    from ddnm_amd.guided_diffusion.unet import create_model
    from oracle import cases
    cfg, sd = cases.adm_net("full")
    m = create_model(**vars(cfg.model))
    m.load_state_dict(sd)
    m.convert_to_fp16()
    x8, t8 = _adm_batch(cfg, 8)
    alone = []
    for i in range(8):
        alone.append(m(x8[i:i + 1].cuda(), t8[i:i + 1].cuda()).float().cpu())
    torch.cuda.synchronize()
    return cfg, m, x8, t8, alone


@pytest.mark.parametrize("B", [4, 8])
def test_full_adm_fp16_forward_at_benchmarked_batch(hip, adm_full_fp16, B, golden_dir):
    cfg, m, x8, t8, alone = adm_full_fp16
    e = m(x8[:B].cuda(), t8[:B].cuda()).float().cpu()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(e).all())
    ref0 = torch.from_numpy(np.load(f"{golden_dir}/adm_forward.npz")["full_eps"])
    err0 = rel(e[:1, :, ::4, ::4], ref0)
    assert 1e-7 < err0 < 3e-3, err0
    errs = [rel(e[i], alone[i][0]) for i in range(B)]
    print(f"[ADM fp16 B={B}] image 0 vs reference golden {err0:.2e}; per image vs B=1: " +
          ", ".join(f"{v:.1e}" for v in errs))
    assert max(errs) < 2e-3, errs


def test_full_adm_fp16_batch1_matches_golden(hip, adm_full_fp16, golden_dir):
    """The B = 1 run the batched ones are compared with is itself pinned to the reference."""
    ref0 = torch.from_numpy(np.load(f"{golden_dir}/adm_forward.npz")["full_eps"])
    assert rel(adm_full_fp16[4][0][:, :, ::4, ::4], ref0) < 3e-3


def test_full_classifier_fp16_cond_fn_batch8_vs_batch1(hip, golden_dir):
    from ddnm_amd.guided_diffusion.classifier import create_classifier
    from oracle import weights
    cc = weights.classifier_config()
    clf = create_classifier(**{k: v for k, v in vars(cc).items() if k != "classifier_scale"})
    clf.load_state_dict(weights.classifier_state_dict(cc))
    clf.convert_to_fp16()
    g = torch.Generator().manual_seed(97)
    r = cc.image_size
    x = torch.randn(8, 3, r, r, generator=g)
    # image 0 = the input of the reference golden (tests/test_classifier.py::_inputs)
    from tests.test_classifier import _inputs
    x0, t0, y0 = _inputs(r)
    x[0] = x0[0]
    t = torch.tensor(T_ROWS)
    t[0] = t0[0]
    y = torch.tensor([int(y0[0]), 3, 17, 999, 951, 0, 500, 42])
    lg8 = clf(x.cuda(), t.cuda()).float().cpu()
    gr8 = clf.log_prob_grad(x.cuda(), t.cuda(), y.cuda()).float().cpu()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(gr8).all())
    le, ge = [], []
    for i in range(8):
        lg1 = clf(x[i:i + 1].cuda(), t[i:i + 1].cuda()).float().cpu()
        gr1 = clf.log_prob_grad(x[i:i + 1].cuda(), t[i:i + 1].cuda(), y[i:i + 1].cuda()).float().cpu()
        le.append(rel(lg8[i], lg1[0]))
        ge.append(rel(gr8[i], gr1[0]))
    print("[classifier fp16 B=8 vs B=1] logits " + ", ".join(f"{v:.1e}" for v in le) + "; gradient " +
          ", ".join(f"{v:.1e}" for v in ge))
    assert max(le) < 2e-3, le
    assert max(ge) < 5e-3, ge
    gold = np.load(f"{golden_dir}/classifier.npz")
    assert rel(lg8[:1], torch.from_numpy(gold["full_logits"])[:1]) < 3e-3
    assert rel(gr8[:1, :, ::4, ::4], torch.from_numpy(gold["full_grad"])[:1]) < 1e-2


def test_micro_batched_forward_beyond_one_launch(hip, adm_full_fp16):
    """"You may increase the batch_size to accelerate evaluation" (reference README.md:89): a batch whose activations
    exceed the 2 GiB one convolution launch can address is split into micro-batches INSIDE forward() (round 4 refused
    it).  ADM fp16 at B = 40 and the celeba `Model` at B = 72 equal the per-chunk forwards bit for bit; the classifier's
    cond_fn at B = 70 likewise."""
    cfg, m, x8, t8, _ = adm_full_fp16
    mb = m.max_forward_batch
    assert mb == 32            # half of the 2 GiB a launch addresses: 256 x 256 x 256 fp16 channels per image
    g = torch.Generator().manual_seed(5)
    r = cfg.model.image_size
    x = torch.randn(40, 3, r, r, generator=g).cuda()
    t = (torch.rand(40, generator=g) * 999).cuda()
    whole = m(x, t)
    parts = torch.cat([m(x[i:i + mb], t[i:i + mb]) for i in range(0, 40, mb)], 0)
    torch.cuda.synchronize()
    assert whole.shape[0] == 40 and bool(torch.isfinite(whole).all()) and torch.equal(whole, parts)
    del whole, parts
    from ddnm_amd.guided_diffusion.models import Model
    from oracle import cases
    ccfg, _ = cases.celeba_net("full")
    cm = Model(ccfg)
    cm.load_state_dict(cm.random_state_dict(3))
    mb = cm.max_forward_batch
    assert mb == 32
    x = torch.randn(72, 3, 256, 256, generator=g).cuda()
    t = torch.full((72,), 430.0).cuda()
    whole = cm(x, t)
    parts = torch.cat([cm(x[i:i + mb], t[i:i + mb]) for i in range(0, 72, mb)], 0)
    torch.cuda.synchronize()
    assert whole.shape[0] == 72 and bool(torch.isfinite(whole).all()) and torch.equal(whole, parts)
    del whole, parts, cm
    from ddnm_amd.guided_diffusion.classifier import create_classifier
    from oracle import weights
    cc = weights.classifier_config()
    clf = create_classifier(**{k: v for k, v in vars(cc).items() if k != "classifier_scale"})
    clf.load_state_dict(weights.classifier_state_dict(cc))
    clf.convert_to_fp16()
    mb = clf.max_group_batch
    assert mb == 64
    x = torch.randn(70, 3, 256, 256, generator=g).cuda()
    t = torch.full((70,), 250.0).cuda()
    y = torch.randint(0, 1000, (70,), generator=g).cuda()
    whole = clf.log_prob_grad(x, t, y)
    parts = torch.cat([clf.log_prob_grad(x[i:i + mb], t[i:i + mb], y[i:i + mb]) for i in range(0, 70, mb)], 0)
    torch.cuda.synchronize()
    assert whole.shape == x.shape and bool(torch.isfinite(whole).all()) and torch.equal(whole, parts)
