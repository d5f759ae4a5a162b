"""UNet forward parity on the GPU: HIP engine vs the oracle restatement (CPU, same seeded
weights/inputs) and vs golden outputs of the REAL reference model (tests/golden)."""
import numpy as np
import pytest
import torch

from tests.helpers import rel

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5      # single-forward rel-L2 bar for the fp32 kernels


def build_engine(cfg, sd):
    from ddnm_amd.guided_diffusion.models import Model
    m = Model(cfg)
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("kind,batch", [("small", 2), ("mid", 2), ("full", 1)])
def test_forward_matches_reference_golden(hip, kind, batch, golden_dir):
    from oracle import cases
    cfg, sd = cases.celeba_net(kind)
    x, t = cases.forward_inputs(cfg, batch)
    eng = build_engine(cfg, sd)
    e = eng(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    e = e.cpu()
    g = np.load(f"{golden_dir}/celeba_forward.npz")
    ref = torch.from_numpy(g[f"{kind}_eps"])
    got = e if kind != "full" else e[..., ::4, ::4]
    assert got.shape == ref.shape
    assert rel(got, ref) < FWD_TOL
    mean, std, asum = g[f"{kind}_stats"]
    assert abs(e.double().abs().sum().item() - asum) / asum < 1e-5


@pytest.mark.parametrize("kind,batch", [("small", 3), ("mid", 1)])
def test_forward_matches_oracle(hip, kind, batch):
    from oracle import cases, unet_celeba
    cfg, sd = cases.celeba_net(kind)
    x, t = cases.forward_inputs(cfg, batch, seed=99)
    ref = unet_celeba.Net(sd, cfg)(x, t)
    e = build_engine(cfg, sd)(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert rel(e, ref) < FWD_TOL


def test_forward_batch_independence_and_determinism(hip):
    """Images of a batch are independent trajectories: B=4 == 4 x B=1 up to fp32
    summation order (the split-K plan of the low-resolution layers depends on the launch's batch size),
    and a repeated launch of the same shape is bit-for-bit reproducible (no atomics anywhere)."""
    from oracle import cases
    cfg, sd = cases.celeba_net("small")
    x, t = cases.forward_inputs(cfg, 4)
    eng = build_engine(cfg, sd)
    full = eng(x.cuda(), t.cuda()).cpu()
    again = eng(x.cuda(), t.cuda()).cpu()
    assert torch.equal(full, again)
    for i in range(4):
        one = eng(x[i:i + 1].cuda(), t[i:i + 1].cuda()).cpu()
        assert rel(one, full[i:i + 1]) < 2e-6


def test_model_requires_weights_and_gpu(hip):
    from oracle import cases
    from ddnm_amd.guided_diffusion.models import Model
    cfg, _ = cases.celeba_net("small")
    with pytest.raises(RuntimeError):
        Model(cfg)(torch.zeros(1, 3, 32, 32, device="cuda"), torch.zeros(1, device="cuda"))


def test_celeba_model_graph_replay_equals_eager(hip):
    """`Model.enable_graphs()` (hipGraph replay, ddnm_amd/graph.py) on the split-fp16 engine: the operand-bound tensors of
    the range guard (ops.amax_bound, the finalize launch's amax output) are allocated inside the capture like every other
    intermediate, so a replay on new inputs must equal the eager forward bit for bit."""
    import torch
    from oracle import cases
    from ddnm_amd.guided_diffusion.models import Model
    cfg = cases.weights.celeba_config(resolution=64, ch=128, ch_mult=(1, 2, 2), attn_resolutions=(16,))
    m = Model(cfg, device="cuda")
    m.load_state_dict(m.random_state_dict(seed=3))
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = [torch.randn(2, 3, 64, 64, device="cuda", generator=g) * s for s in (1.0, 40.0)]
    ts = [torch.tensor([500.0, 20.0], device="cuda"), torch.tensor([990.0, 0.0], device="cuda")]
    eager = [m(x, t).clone() for x, t in zip(xs, ts)]
    m.enable_graphs()
    for _ in range(2):
        for x, t, e in zip(xs, ts, eager):
            out = m(x, t)
            torch.cuda.synchronize()
            assert torch.equal(out, e)
    m.disable_graphs()


@pytest.mark.gpu
def test_auto_graphs_small_batches_equal_eager(hip):
    """`auto_graphs(2)` (what the runner switches on, guided_diffusion/diffusion.py::Diffusion.sample): forwards of one or
    two images are replayed from a captured hipGraph and equal the eager forward bit for bit, larger batches stay eager;
    a full small restoration through ddnm_diffusion gives the same image either way (celeba `Model` and the ADM UNet)."""
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from ddnm_amd.guided_diffusion.models import Model
    from ddnm_amd.guided_diffusion.unet import create_model
    from oracle import cases, schedule
    from tests.helpers import engine_operator
    cfg, sd = cases.celeba_net("small")
    m = Model(cfg)
    m.load_state_dict(sd)
    d = cfg.data.image_size
    g = torch.Generator().manual_seed(7)
    for B in (1, 2, 3):
        x, t = torch.randn(B, 3, d, d, generator=g).cuda(), torch.full((B,), 430.0).cuda()
        want = m(x, t).clone()
        m.auto_graphs(2)
        got = m(x, t)
        got2 = m(x * 0.5, t)                 # a replay with other inputs ...
        got3 = m(x, t)                       # ... and back
        captured = m._auto_graphs is not None and any(k[0][0] == B for k in m._auto_graphs.entries)
        m.auto_graphs(0)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got3, want) and not torch.equal(got2, want)
        assert captured == (B <= 2)
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 6, 1, 1
    n_it = len(schedule.jump_times(6, 1, 1)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 1, n_it)
    op = engine_operator("colorization", d)
    y = cases.make_operator("colorization", d).A(x_orig).cuda()
    outs = []
    for mb in (0, 2):
        m.auto_graphs(mb)
        xs, _ = ddnm_diffusion(x_T.cuda(), m, cases.betas().cuda(), 0.85, op, y, cls_fn=None, classes=None, config=cfg,
                               noise=[n.cuda() for n in tape], return_cpu=False)
        outs.append(xs[0].clone())
    assert torch.equal(outs[0], outs[1])
    acfg, asd = cases.adm_net("small")
    am = create_model(**vars(acfg.model))
    am.load_state_dict(asd)
    am.convert_to_fp16()
    r = acfg.data.image_size
    x, t = torch.randn(2, 3, r, r, generator=g).cuda(), torch.full((2,), 250.0).cuda()
    yy = torch.tensor([3, 951]).cuda() if acfg.model.class_cond else None
    want = am(x, t, yy).clone()
    am.auto_graphs(2)
    got = am(x, t, yy)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
