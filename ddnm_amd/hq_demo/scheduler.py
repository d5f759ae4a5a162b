"""Jump schedule of the hq_demo sampler (hq_demo/guided_diffusion/scheduler.py:69-147): host-side integers."""


def get_schedule_jump(t_T, n_sample, jump_length, jump_n_sample, jump2_length=1, jump2_n_sample=1, jump3_length=1,
                      jump3_n_sample=1, start_resampling=100000000):
    """Times t_T-1 ... -1 with `jump_n_sample - 1` re-ascents of `jump_length` at every multiple of `jump_length`
    below t_T - jump_length; two finer nested levels (jump2 / jump3) re-arm whenever a coarser level jumps, and
    `n_sample > 1` inserts one-step up/down resampling at every time below t_T - 2."""
    levels = ((jump_length, jump_n_sample), (jump2_length, jump2_n_sample), (jump3_length, jump3_n_sample))

    def fresh(level):
        length, n = levels[level]
        return dict.fromkeys(range(0, t_T - length, length), n - 1)

    remaining = [fresh(0), fresh(1), fresh(2)]
    ts, t = [], t_T
    while t >= 1:
        t -= 1
        ts.append(t)
        if t + 1 < t_T - 1 and t <= start_resampling:
            for _ in range(n_sample - 1):
                ts.append(t + 1)
                t += 1
                if t >= 0:
                    t -= 1
                    ts.append(t)
        for level in (2, 1, 0):                          # finest level first, like the reference's if-chain
            length = levels[level][0]
            if remaining[level].get(t, 0) > 0 and t <= start_resampling - length:
                remaining[level][t] -= 1
                ts.extend(range(t + 1, t + length + 1))
                t += length
                for finer in range(level + 1, 3):
                    remaining[finer] = fresh(finer)
    ts.append(-1)
    _check_times(ts, -1, t_T)
    return ts


def _check_times(times, t_0, t_T):
    """scheduler.py:47-62."""
    assert times[0] > times[1], (times[0], times[1])
    assert times[-1] == -1, times[-1]
    for a, b in zip(times[:-1], times[1:]):
        assert abs(a - b) == 1, (a, b)
    for t in times:
        assert t_0 <= t <= t_T, (t, t_0, t_T)
