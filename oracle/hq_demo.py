"""CPU restatement of the reference's `hq_demo` sampler (arbitrary-size restoration with the mask-shift trick).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, in the reference tree:
  * hq_demo/guided_diffusion/scheduler.py:69-147      RePaint-style jump schedule
  * hq_demo/guided_diffusion/respace.py:23-122        timestep respacing (betas of the retained steps, index map)
  * hq_demo/guided_diffusion/gaussian_diffusion.py:148-212   posterior coefficients, `_undo`
  * ...:246-404   p_mean_variance with the "DDNM core" (Eq. 17/19) and the mask-shift paste
  * ...:430-487   p_sample (classifier-conditioned mean, sqrt(gamma_t) noise)
  * ...:548-756   p_sample_loop_progressive: degradation set-up, 256x256 tiles with 128-px shifts
Pinned by tests/golden/hq_demo.npz, produced by tests/golden/make_golden_hq.py from the imported reference.

Reference behaviours kept on purpose:
  * the state `x_t` is drawn ONCE; every tile after the first starts its reverse process from the previous
    tile's final sample (gaussian_diffusion.py:575-578 are outside the tile loops);
  * in the low-noise branch lambda_t = sigma_t / a_t * sigma_y (sic, :334), gamma_t = 0;
  * classifier guidance adds variance * grad with variance = gamma_t (:402, :417-427);
  * `model_var_values` (learned sigma) are computed and ignored.
"""
import math

import numpy as np
import torch

TILE, SHIFT = 256, 128


# --------------------------------------------------------------------------------------- schedule / respacing
def schedule_jump(t_T, n_sample, jump_length, jump_n_sample, jump2_length=1, jump2_n_sample=1, jump3_length=1,
                  jump3_n_sample=1, start_resampling=100000000):
    """scheduler.py:69-147.  Three nested jump levels; a jump on level k re-arms the levels above it."""
    lengths = [jump_length, jump2_length, jump3_length]
    counts = [jump_n_sample, jump2_n_sample, jump3_n_sample]

    def armed(k):
        return {j: counts[k] - 1 for j in range(0, t_T - lengths[k], lengths[k])}

    left = [armed(0), armed(1), armed(2)]
    t, ts = t_T, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if t + 1 < t_T - 1 and t <= start_resampling:
            for _ in range(n_sample - 1):
                t += 1
                ts.append(t)
                if t >= 0:
                    t -= 1
                    ts.append(t)
        for k in (2, 1, 0):
            if left[k].get(t, 0) > 0 and t <= start_resampling - lengths[k]:
                left[k][t] -= 1
                for _ in range(lengths[k]):
                    t += 1
                    ts.append(t)
                for hi in range(k + 1, 3):
                    left[hi] = armed(hi)
    ts.append(-1)
    assert ts[0] > ts[1] and all(abs(a - b) == 1 for a, b in zip(ts[:-1], ts[1:]))        # _check_times :47-62
    return ts


def space_timesteps(num_timesteps, section_counts):
    """respace.py:23-77 for the "N" / "a,b,c" forms used by the configs."""
    counts = [int(v) for v in str(section_counts).split(",")]
    if len(counts) == 1 and counts[0] > num_timesteps:                  # more steps than the process has (:52-53)
        return sorted(set(np.linspace(start=0, stop=num_timesteps, num=counts[0])))
    size_per, extra = divmod(num_timesteps, len(counts))
    start, steps = 0, []
    for i, c in enumerate(counts):
        size = size_per + (1 if i < extra else 0)
        if size < c:
            raise ValueError(f"cannot divide section of {size} steps into {c}")
        stride = 1 if c <= 1 else (size - 1) / (c - 1)
        cur = 0.0
        for _ in range(c):                       # running float sum, rounded half-to-even, exactly like :66-70
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class Posterior:
    """float64 tables of gaussian_diffusion.py:148-212 for the RESPACED process (respace.py:93-104) with the
    linear schedule scaled by 1000/steps (:81-92, use_scale=True in script_util.py:272)."""

    def __init__(self, steps=1000, timestep_respacing="100"):
        scale = 1000 / steps
        base = np.linspace(scale * 0.0001, scale * 0.02, steps, dtype=np.float64)
        acp = np.cumprod(1.0 - base)
        self.timestep_map = space_timesteps(steps, timestep_respacing)
        last, betas = 1.0, []
        for i in self.timestep_map:
            betas.append(1 - acp[i] / last)
            last = acp[i]
        self.betas = betas = np.array(betas)
        ac = np.cumprod(1.0 - betas)
        ac_prev = np.append(1.0, ac[:-1])
        self.sqrt_recip = np.sqrt(1.0 / ac)
        self.sqrt_recipm1 = np.sqrt(1.0 / ac - 1)
        self.variance = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)


# --------------------------------------------------------------------------------------- degradations
def mean_upsample(x, scale):
    return x.repeat_interleave(scale, dim=2).repeat_interleave(scale, dim=3)            # MeanUpsample :65-69


def color2gray(x):
    g = x[:, 0] * (1 / 3) + x[:, 1] * (1 / 3) + x[:, 2] * (1 / 3)                       # :54-57
    return g[:, None].repeat(1, 3, 1, 1)


def gray2color(x):
    coef = 1 / 3
    return torch.stack([x[:, 0] * coef / (3 * coef ** 2)] * 3, 1)                       # :59-63


def degradation(deg, scale, gt_shape, mask=None):
    """(A, Ap, A_temp) of gaussian_diffusion.py:592-641: A / Ap act on 256x256 tiles, A_temp on the whole image."""
    def pool(h, w):
        return torch.nn.AdaptiveAvgPool2d((h // scale, w // scale))

    def up(z):
        return mean_upsample(z, scale)
    if deg == "sr_averagepooling":
        return pool(TILE, TILE), up, pool(gt_shape[2], gt_shape[3])
    if deg == "colorization":
        return color2gray, gray2color, color2gray
    if deg == "sr_color":
        p_tile, p_full = pool(TILE, TILE), pool(gt_shape[2], gt_shape[3])
        return (lambda z: color2gray(p_tile(z))), (lambda z: up(gray2color(z))), (lambda z: color2gray(p_full(z)))
    if deg == "inpainting":                      # face256 only in the reference (:600)
        return (lambda z: z * mask), (lambda z: z * mask), (lambda z: z * mask)
    if deg == "mask_color_sr":                   # face256 only (:606)
        p_tile = pool(TILE, TILE)
        A = lambda z: p_tile(color2gray(z * mask))                                       # noqa: E731
        return A, (lambda z: gray2color(up(z)) * mask), A
    raise NotImplementedError("degradation type not supported")


def tile_plan(H, W):
    """Tile origins of gaussian_diffusion.py:664-689 and the already-restored strips pasted into x0_hat
    (:341-377): (h0, w0, left_cols, top_rows) per tile, row-major; 0 = no strip."""
    if H < TILE or W < TILE:
        raise ValueError("Please set a larger SR scale")
    nh, nw = math.ceil(H / SHIFT) - 1, math.ceil(W / SHIFT) - 1
    plan = []
    for i in range(nh):
        last_h = i == nh - 1 and H % SHIFT != 0
        h0 = H - TILE if last_h else SHIFT * i
        for j in range(nw):
            last_w = j == nw - 1 and W % SHIFT != 0
            w0 = W - TILE if last_w else SHIFT * j
            left = 0 if j == 0 else (TILE - W % SHIFT if last_w else SHIFT)
            top = 0 if i == 0 else (TILE - H % SHIFT if last_h else SHIFT)
            plan.append((h0, w0, left, top))
    return plan


# --------------------------------------------------------------------------------------- sampler
def restore(model, gt, deg, scale, sigma_y, resize_y, x_init, tape, *, classes=None, cond_fn=None, mask=None,
            timestep_respacing="100", schedule=None, steps=1000, clip_denoised=True, trace=None):
    """p_sample_loop_progressive (:548-756).  `model(x, t_original, y)` returns [B, 6, 256, 256] (or 3 channels);
    `cond_fn(x, t_original, y)` returns the scaled classifier gradient; `x_init` is the single initial draw and
    `tape` yields every later Gaussian draw (p_sample :474, _undo :205) in order.  Returns the full-size result."""
    post = Posterior(steps, timestep_respacing)
    schedule = dict(t_T=100, n_sample=1, jump_length=10, jump_n_sample=3) if schedule is None else schedule
    f32 = lambda v: torch.tensor(v, dtype=torch.float64).float()                         # noqa: E731  (_extract_into_tensor :758-771)
    if 256 % scale != 0:
        raise ValueError("Please set a SR scale divisible by 256")
    if resize_y:
        gt = mean_upsample(gt, scale)
    A, Ap, A_temp = degradation(deg, scale, gt.shape, mask)
    y_full = A_temp(gt)
    apy_full = Ap(y_full)
    H, W = apy_full.shape[2], apy_full.shape[3]
    final = torch.zeros_like(apy_full)
    noise = iter(tape)
    x = x_init
    times = schedule_jump(**schedule)
    tmap = torch.tensor(post.timestep_map)
    for (h0, w0, left, top) in tile_plan(H, W):
        apy = apy_full[:, :, h0:h0 + TILE, w0:w0 + TILE]
        x0_hat = None
        for t_last, t_cur in zip(times[:-1], times[1:]):
            if t_cur < t_last:
                t = t_last
                tt = tmap[torch.full((x.shape[0],), t)]
                out = model(x, tt, classes)
                eps = out[:, :3]
                x0 = f32(post.sqrt_recip[t]) * x - f32(post.sqrt_recipm1[t]) * eps       # :404-410
                if clip_denoised:
                    x0 = x0.clamp(-1, 1)
                var = f32(post.variance[t])
                sigma_t, a_t = torch.sqrt(var), f32(post.coef1[t])
                if sigma_t >= a_t * sigma_y:                                             # Eq. 19, :329-335
                    lam, gamma = 1, var - (a_t * 1 * sigma_y) ** 2
                else:
                    lam, gamma = sigma_t / a_t * sigma_y, 0.0
                x0_hat = lam * apy + x0 - lam * Ap(A(x0))                                # Eq. 17, :339
                if left:
                    x0_hat[:, :, :, :left] = final[:, :, h0:h0 + TILE, w0:w0 + left]
                if top:
                    x0_hat[:, :, :top, :] = final[:, :, h0:h0 + top, w0:w0 + TILE]
                mean = f32(post.coef1[t]) * x0_hat + f32(post.coef2[t]) * x              # :214-229
                if cond_fn is not None:
                    mean = mean.float() + gamma * cond_fn(x, tt, classes).float()        # :412-427
                nz = next(noise)
                x = mean + (0.0 if t == 0 else 1.0) * torch.sqrt(torch.ones(1) * gamma) * nz
                if trace is not None:
                    trace.append((x0_hat.clone(), x.clone()))
            else:
                beta = f32(post.betas[t_last + 1])                                       # undo, t_shift = 1 (:729-735)
                x = torch.sqrt(1 - beta) * x + torch.sqrt(beta) * next(noise)
        final[:, :, h0:h0 + TILE, w0:w0 + TILE] = x0_hat
    return final, y_full, apy_full
