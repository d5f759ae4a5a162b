"""Canary suite: every kernel of the hot path runs on tensors carved out of POISONED slabs.

`torch.empty / empty_like / zeros` are replaced, for the duration of a test, by an allocator that over-allocates every
tensor by a guard region on each side and fills guards AND payload with a poison pattern (NaN for floating point).
After the forward / sampler steps:
  * every guard must still hold the poison bit pattern  -> no kernel wrote outside its output;
  * the results must be finite                           -> no kernel READ outside its inputs or read scratch it never
                                                            wrote (a poisoned value anywhere on the data path ends up as
                                                            NaN in the output).
Shapes: the full celeba_hq Model, the full ADM UNet (fp32 kernels, fp16-activation path and first-generation fp16
path), the full classifier forward + input-gradient backward, at B = 1 and B = 3 (odd: ragged split-K / tile plans),
plus sampler steps of every BASELINE operator.  Rules out (or would have reproduced) the unexplained
`Memory access fault by GPU node-2` of round 1's last GPU call."""
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 4096           # elements on each side


class Poisoned:
    def __init__(self):
        self.slabs = []
        self._empty, self._empty_like, self._zeros = torch.empty, torch.empty_like, torch.zeros

    def _poison(self, dtype):
        return float("nan") if dtype.is_floating_point else 0x5A

    def alloc(self, shape, dtype, device, zero=False):
        n = 1
        for s in shape:
            n *= int(s)
        slab = self._empty(n + 2 * GUARD, dtype=dtype, device=device)
        slab.fill_(self._poison(dtype))
        view = slab[GUARD:GUARD + n]
        if zero:
            view.zero_()
        self.slabs.append((slab, n))
        return view.view(*shape) if len(shape) else view.view(())

    def empty(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dtype = dtype or torch.get_default_dtype()
        if device is None or torch.device(device).type != "cuda" or kw.get("pin_memory"):
            return self._empty(*size, dtype=dtype, device=device, **kw)
        return self.alloc(size, dtype, device)

    def zeros(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dtype = dtype or torch.get_default_dtype()
        if device is None or torch.device(device).type != "cuda":
            return self._zeros(*size, dtype=dtype, device=device, **kw)
        return self.alloc(size, dtype, device, zero=True)

    def empty_like(self, t, **kw):
        if not t.is_cuda or kw:
            return self._empty_like(t, **kw)
        return self.alloc(tuple(t.shape), t.dtype, t.device)

    def check_guards(self):
        bad = 0
        for slab, n in self.slabs:
            for g in (slab[:GUARD], slab[GUARD + n:]):
                ok = torch.isnan(g).all() if slab.dtype.is_floating_point else (g == 0x5A).all()
                bad += 0 if bool(ok) else 1
        return bad, len(self.slabs)


@pytest.fixture
def poisoned(monkeypatch):
    from ddnm_amd import ops
    p = Poisoned()
    ops._conv_ws.clear()               # cached scratch must be re-allocated under the guard allocator
    ops._f16_scratch_buf.clear()
    monkeypatch.setattr(torch, "empty", p.empty)
    monkeypatch.setattr(torch, "empty_like", p.empty_like)
    monkeypatch.setattr(torch, "zeros", p.zeros)
    yield p
    monkeypatch.undo()
    ops._conv_ws.clear()
    ops._f16_scratch_buf.clear()


def _finish(p, *outs, min_slabs=20):
    torch.cuda.synchronize()
    for o in outs:
        assert bool(torch.isfinite(o.float()).all()), "a poisoned (out-of-bounds or never written) value reached the output"
    bad, n = p.check_guards()
    assert n >= min_slabs and bad == 0, f"{bad} of {2 * n} guard regions were overwritten ({n} guarded tensors)"


@pytest.mark.parametrize("B", [1, 3])
def test_canary_celeba_model(hip, poisoned, B):
    from ddnm_amd.guided_diffusion.models import Model
    from oracle import cases
    cfg, _ = cases.celeba_net("full")
    m = Model(cfg)
    m.load_state_dict(m.random_state_dict(3))
    x = torch.randn(B, 3, 256, 256, device="cuda")
    _finish(poisoned, m(x, torch.full((B,), 430.0, device="cuda")))


@pytest.mark.parametrize("B,mode", [(1, "fp32"), (3, "fp32"), (1, "h16"), (3, "h16"), (3, "gen1"), (8, "h16")])
def test_canary_adm_unet(hip, poisoned, monkeypatch, B, mode):
    from ddnm_amd.guided_diffusion.unet import create_model
    from oracle import cases, weights
    cfg = weights.adm_config(class_cond=True)
    m = create_model(**vars(cfg.model))
    m.load_state_dict(m.random_state_dict(5))
    if mode != "fp32":
        m.convert_to_fp16()
    monkeypatch.setenv("DDNM_ADM_GEN1", "1" if mode == "gen1" else "0")
    x = torch.randn(B, 3, 256, 256, device="cuda")
    y = torch.randint(0, 1000, (B,), device="cuda")
    _finish(poisoned, m(x, torch.full((B,), 770.0, device="cuda"), y))


@pytest.mark.parametrize("B,fp16", [(1, False), (3, False), (3, "gen1"), (1, "h16"), (3, "h16"), (8, "h16")])
def test_canary_classifier_forward_backward(hip, poisoned, monkeypatch, B, fp16):
    """fp32 engine, first-generation fp16-operand engine (DDNM_CLS_GEN1=1) and the fp16-activation engine (round 5:
    conv16 forward / data-gradient launches, fused attention forward + backward, fp16 GroupNorm backward)."""
    from ddnm_amd.guided_diffusion.classifier import create_classifier, make_cond_fn
    from oracle import weights
    monkeypatch.setenv("DDNM_CLS_GEN1", "1" if fp16 == "gen1" else "0")
    cc = weights.classifier_config()
    clf = create_classifier(**{k: v for k, v in vars(cc).items() if k != "classifier_scale"})
    clf.load_state_dict(weights.classifier_state_dict(cc))
    if fp16:
        clf.convert_to_fp16()
    x = torch.randn(B, 3, 256, 256, device="cuda")
    grad = make_cond_fn(clf, 1.0)(x, torch.full((B,), 250.0, device="cuda"), torch.full((B,), 951, device="cuda"))
    _finish(poisoned, grad)


@pytest.mark.parametrize("deg", ["sr_averagepooling", "sr_bicubic", "colorization", "inpainting", "cs_walshhadamard",
                                 "denoising"])
def test_canary_sampler_steps(hip, poisoned, deg, golden_dir):
    """Projection + DDIM-update kernels of every BASELINE operator at 256 x 256, B = 3, with a stub noise predictor."""
    import types
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from oracle import cases
    from tests.helpers import engine_operator, real_mask
    ns = types.SimpleNamespace
    cfg = ns(diffusion=ns(num_diffusion_timesteps=1000), time_travel=ns(T_sampling=10, travel_length=2, travel_repeat=2))
    op = engine_operator(deg, 256, real_mask(golden_dir) if deg == "inpainting" else None)
    x_orig = torch.rand(3, 3, 256, 256, device="cuda") * 2 - 1
    y = op.A(x_orig)

    def model(xt, t):
        return torch.tanh(xt) * 0.5 + 0.01 * t.view(-1, 1, 1, 1) / 1000

    xs, x0s = ddnm_diffusion(torch.randn(3, 3, 256, 256, device="cuda"), model, cases.betas().cuda(), 0.85, op, y, cls_fn=None,
                             classes=None, config=cfg, return_cpu=False)
    _finish(poisoned, xs[0], x0s[0], y, min_slabs=3)
