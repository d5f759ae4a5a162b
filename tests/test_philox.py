"""In-kernel noise (ddnm_amd/csrc/philox.h, ops.PhiloxNoise): the reference draws `torch.randn_like(x)` per loop iteration
(functions/svd_ddnm.py:65,74); the engine can draw the same distribution INSIDE the step kernels from a counter-based
generator.  Oracle = numpy restatement (oracle/philox.py) pinned to the Random123 known-answer vectors of Philox4x32-10."""
import numpy as np
import pytest
import torch

from oracle import philox as P

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_oracle_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    for ctr, key, want in KAT:
        got = P.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert [int(v) for v in got] == list(want)


def _moments(x):
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    m, s = x.mean(), x.std()
    z = (x - m) / s
    return m, s, (z ** 3).mean(), (z ** 4).mean()


def test_oracle_normal_moments_and_streams():
    x = P.randn(1234, 2, 3 * 128 * 128, 7)
    m, s, sk, ku = _moments(x)
    assert abs(m) < 1e-2 and abs(s - 1) < 1e-2 and abs(sk) < 3e-2 and abs(ku - 3) < 6e-2
    # different iteration / image / seed: independent streams; same counter: same values
    y = P.randn(1234, 2, 3 * 128 * 128, 8)
    assert abs(np.corrcoef(x.reshape(-1), y.reshape(-1))[0, 1]) < 1e-2
    assert abs(np.corrcoef(x[0], x[1])[0, 1]) < 2e-2
    assert np.array_equal(P.randn(1234, 1, 4096, 7, image_base=1)[0], x[1][:4096])


@pytest.mark.gpu
def test_engine_draw_matches_oracle_and_is_normal(hip):
    from scipy import stats
    from ddnm_amd import ops
    like = torch.empty(2, 3, 64, 64, device="cuda")
    ph = ops.PhiloxNoise(seed=(77 << 32) | 1234, image_base=5)
    got = ph.tensor(9, like).cpu().numpy().reshape(2, -1)
    want = P.randn((77 << 32) | 1234, 2, 3 * 64 * 64, 9, image_base=5)
    assert np.abs(got - want).max() < 3e-5            # same counters, same bits; libm vs numpy log / sincos
    big = ops.PhiloxNoise(seed=99).tensor(3, torch.empty(8, 3, 256, 256, device="cuda")).cpu().numpy().reshape(-1)
    assert np.isfinite(big).all()
    m, s, sk, ku = _moments(big)
    assert abs(m) < 2e-3 and abs(s - 1) < 2e-3 and abs(sk) < 5e-3 and abs(ku - 3) < 1e-2, (m, s, sk, ku)
    d = stats.kstest(big[::3], "norm").statistic       # 524288 samples: the 1 % critical value is 2.25e-3
    assert d < 2.25e-3, d
    assert abs(big).max() < 6.7                       # |z| <= sqrt(-2 ln 2^-33)
    # an image's noise depends on its GLOBAL index only: images 2, 3 of a batch of four = a batch of two starting at 2
    a = ops.PhiloxNoise(seed=5, image_base=0).tensor(1, torch.empty(4, 3, 32, 32, device="cuda"))
    b = ops.PhiloxNoise(seed=5, image_base=2).tensor(1, torch.empty(2, 3, 32, 32, device="cuda"))
    assert torch.equal(a[2:], b)


@pytest.mark.gpu
@pytest.mark.parametrize("deg,d", [("sr_averagepooling", 64), ("colorization", 64), ("inpainting", 64), ("denoising", 64),
                                   ("sr_bicubic", 64), ("cs_walshhadamard", 64), ("sr_bicubic", 256)])
def test_in_kernel_draw_equals_the_tensor_path(hip, deg, d):
    """A step with noise = NULL + ddnm_step_scalars::rng_* gives the bits of the same step fed with the materialised draw."""
    from ddnm_amd import ops
    from oracle import cases
    from tests.helpers import engine_operator
    B = 3
    op = engine_operator(deg, d)
    g = torch.Generator().manual_seed(3)
    x_orig = (torch.rand(B, 3, d, d, generator=g) * 2 - 1)
    y = cases.make_operator(deg, d).A(x_orig).cuda().reshape(B, -1).float().contiguous()
    xt, et = torch.randn(B, 3, d, d, generator=g).cuda(), torch.randn(B, 3, d, d, generator=g).cuda()
    ph = ops.PhiloxNoise(seed=4242, image_base=7)
    outs = []
    for in_kernel in (True, False):
        s = ops.step_scalars(torch.tensor(0.5), torch.tensor(0.6), 0.85)
        x0, out = torch.empty_like(xt), torch.empty_like(xt)
        if in_kernel:
            ph.stamp(s, 13)
            op.ddnm_step(xt, et, None, y, s, x0, out)
        else:
            op.ddnm_step(xt, et, ph.tensor(13, xt), y, s, x0, out)
        torch.cuda.synchronize()
        outs.append((x0.clone(), out.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool(torch.isfinite(outs[0][1]).all())


@pytest.mark.gpu
def test_sampling_run_with_in_kernel_noise_equals_its_tape(hip):
    """ddnm_diffusion(noise=PhiloxNoise) -- time travel included (the re-noise steps materialise their draw) -- equals the
    run on the explicit tape of the same draws, and two shards of the batch reproduce the unsharded run."""
    from ddnm_amd import ops
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from ddnm_amd.guided_diffusion.models import Model
    from oracle import cases, schedule
    from tests.helpers import engine_operator
    cfg, sd = cases.celeba_net("small")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 8, 2, 2
    n_it = len(schedule.jump_times(8, 2, 2)) - 1
    d = cfg.data.image_size
    m = Model(cfg)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x_orig = torch.rand(4, 3, d, d, generator=g) * 2 - 1
    op = engine_operator("sr_averagepooling", d)
    y = cases.make_operator("sr_averagepooling", d).A(x_orig).cuda()
    ph = ops.PhiloxNoise(seed=31337)
    x_T = ph.tensor(ops.PhiloxNoise.XT_ITER, torch.empty(4, 3, d, d, device="cuda"))
    run = lambda x, yy, nz: ddnm_diffusion(x, m, cases.betas().cuda(), 0.85, op, yy, cls_fn=None, classes=None, config=cfg,  # noqa: E731
                                           noise=nz, return_cpu=False)[0][0].clone()
    a = run(x_T, y, ph)
    tape = [ph.tensor(k, x_T) for k in range(n_it)]
    b = run(x_T, y, tape)
    assert torch.equal(a, b)
    lo = run(x_T[:2].contiguous(), y[:2], ops.PhiloxNoise(seed=31337, image_base=0))
    hi = run(x_T[2:].contiguous(), y[2:], ops.PhiloxNoise(seed=31337, image_base=2))
    torch.cuda.synchronize()
    both = torch.cat([lo, hi], 0)            # same noise per image; the batch size may pick other split-K plans (fp32 order)
    assert ((both - a).double().norm() / a.double().norm()).item() < 1e-5
    # un-pinned runs draw different noise from call to call, reproducibly under torch.cuda.manual_seed
    torch.cuda.manual_seed(1)
    from ddnm_amd.functions import svd_ddnm
    svd_ddnm._PHILOX_CALLS[0] = 0
    c1, c2 = run(x_T, y, None), run(x_T, y, None)
    svd_ddnm._PHILOX_CALLS[0] = 0
    c3 = run(x_T, y, None)
    assert not torch.equal(c1, c2) and torch.equal(c1, c3)
