"""Restated schedules of the DDNM sampler (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows
  * get_beta_schedule      guided_diffusion/diffusion.py:46-76  (float64 -> .float(), :98)
  * compute_alpha          functions/svd_ddnm.py:10-13          (fp32 cumprod over [0, beta])
  * get_schedule_jump      functions/svd_ddnm.py:167-190        (RePaint-style time travel)
  * _check_times           functions/svd_ddnm.py:192-206
"""
import numpy as np
import torch


def beta_schedule(kind, beta_start, beta_end, n):
    if kind == "linear":
        b = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif kind == "quad":
        b = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif kind == "const":
        b = beta_end * np.ones(n, dtype=np.float64)
    elif kind == "jsd":
        b = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif kind == "sigmoid":
        s = np.linspace(-6, 6, n)
        b = 1 / (np.exp(-s) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(kind)
    return torch.from_numpy(b).float()


def alpha_bar(betas, t):
    """alpha_bar_t as a python float32 value; t = -1 gives exactly 1.0."""
    full = torch.cat([torch.zeros(1), betas.float().cpu()], dim=0)
    return (1 - full).cumprod(dim=0)[t + 1]


def jump_times(T_sampling, travel_length, travel_repeat):
    pending = {j: travel_repeat - 1 for j in range(0, T_sampling - travel_length, travel_length)}
    t, out = T_sampling, []
    while t >= 1:
        t -= 1
        out.append(t)
        if pending.get(t, 0) > 0:
            pending[t] -= 1
            for _ in range(travel_length):
                t += 1
                out.append(t)
    out.append(-1)
    # invariants of _check_times
    assert out[0] > out[1] and out[-1] == -1
    assert all(abs(a - b) == 1 for a, b in zip(out[:-1], out[1:]))
    assert all(-1 <= v <= T_sampling for v in out)
    return out
