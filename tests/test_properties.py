"""Property tests (hypothesis) of the host-side logic: engine vs oracle restatement on random parameters, and the
structural invariants both must satisfy.  CPU only."""
import math

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from ddnm_amd import dist as ddist
from ddnm_amd.functions.svd_ddnm import get_schedule_jump
from ddnm_amd.hq_demo import get_schedule_jump as hq_schedule
from ddnm_amd.hq_demo import space_timesteps, tile_plan
from oracle import hq_demo as H
from oracle import schedule


@settings(max_examples=200, deadline=None, derandomize=True)
@given(st.integers(256, 1300), st.integers(256, 1300))
def test_tile_plan_covers_image_and_strips_are_restored(h, w):
    plan = tile_plan(h, w)
    assert plan == H.tile_plan(h, w)
    assert len(plan) == (math.ceil(h / 128) - 1) * (math.ceil(w / 128) - 1)
    done = np.zeros((h, w), dtype=bool)
    for h0, w0, left, top in plan:
        assert 0 <= h0 <= h - 256 and 0 <= w0 <= w - 256
        if left:            # the pasted strips only contain pixels an earlier tile has written
            assert done[h0:h0 + 256, w0:w0 + left].all()
        if top:
            assert done[h0:h0 + top, w0:w0 + 256].all()
        done[h0:h0 + 256, w0:w0 + 256] = True
    assert done.all()


@settings(max_examples=200, deadline=None, derandomize=True)
@given(st.integers(2, 120), st.integers(1, 12), st.integers(1, 4))
def test_ddnm_schedule_matches_oracle(t_sampling, travel_length, travel_repeat):
    times = get_schedule_jump(t_sampling, travel_length, travel_repeat)
    assert times == schedule.jump_times(t_sampling, travel_length, travel_repeat)
    assert times[0] == t_sampling - 1 and times[-1] == -1
    assert all(abs(a - b) == 1 for a, b in zip(times[:-1], times[1:]))
    assert sum(1 for a, b in zip(times[:-1], times[1:]) if b < a) >= t_sampling      # every step is denoised at least once


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.integers(4, 120), st.integers(1, 3), st.integers(1, 12), st.integers(1, 4), st.integers(1, 4), st.integers(1, 3))
def test_hq_schedule_matches_oracle(t_T, n_sample, jump_length, jump_n_sample, jump2_length, jump2_n_sample):
    kw = dict(t_T=t_T, n_sample=n_sample, jump_length=jump_length, jump_n_sample=jump_n_sample,
              jump2_length=jump2_length, jump2_n_sample=jump2_n_sample)
    ts = hq_schedule(**kw)
    assert ts == H.schedule_jump(**kw)
    assert ts[0] == t_T - 1 and ts[-1] == -1 and max(ts) <= t_T - 1 + max(jump_length, jump2_length, 1)


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.integers(8, 1000), st.lists(st.integers(1, 40), min_size=1, max_size=4))
def test_space_timesteps_matches_oracle(steps, counts):
    spec = ",".join(str(c) for c in counts)
    try:
        want = H.space_timesteps(steps, spec)
    except ValueError:
        want = None
    if want is None:
        try:
            space_timesteps(steps, spec)
            assert False, "engine accepted a respacing the oracle rejects"
        except ValueError:
            return
    got = sorted(space_timesteps(steps, spec))
    assert got == want and got[0] == 0
    if not (len(counts) == 1 and counts[0] > steps):          # (the oversampling special case ends AT `steps`)
        assert got[-1] <= steps - 1


@settings(max_examples=200, deadline=None, derandomize=True)
@given(st.integers(0, 500), st.integers(1, 16))
def test_shard_range_partitions(n, world):
    ranges = [ddist.shard_range(n, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
