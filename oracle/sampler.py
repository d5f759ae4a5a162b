"""Restated DDNM reverse loops (TEST INFRASTRUCTURE, see oracle/__init__.py).

  * ddnm_diffusion  follows functions/svd_ddnm.py:19-78   (sigma_y = 0, SVD path)
  * simplified_ddnm follows guided_diffusion/diffusion.py:333-397 (+ lambdas :245-292)

Reference quirks reproduced on purpose:
timesteps are float tensors i = 990, 980, ...; the time-travel branch re-noises
the UN-projected x0 prediction; the simplified path uses sigma_t = sqrt(1 - abar'^2)
and multiplies the whole noise term by gamma_t.

`noise` is an explicit tape: one N(0, I) tensor per loop iteration, consumed in
order (the reference draws them with torch.randn_like, svd_ddnm.py:65,74).
"""
import torch

from . import schedule


def ddnm_diffusion(x, model, betas, eta, A_funcs, y, noise, T_sampling=100, travel_length=1,
                   travel_repeat=1, num_timesteps=1000, record=None):
    """Returns (x_0, last x0_t).  `record(k, name, tensor)` is an optional probe."""
    skip = num_timesteps // T_sampling
    n = x.shape[0]
    times = schedule.jump_times(T_sampling, travel_length, travel_repeat)
    x0_last, xt = None, x
    tape = iter(noise)
    with torch.no_grad():
        for k, (i, j) in enumerate(zip(times[:-1], times[1:])):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = schedule.alpha_bar(betas, j)
            if j < i:
                t = torch.ones(n) * i
                at = schedule.alpha_bar(betas, i)
                et = model(xt, t)
                if et.shape[1] == 6:
                    et = et[:, :3]
                x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
                resid = A_funcs.A(x0_t.reshape(n, -1)) - y.reshape(n, -1)
                x0_hat = x0_t - A_funcs.A_pinv(resid).reshape(*x0_t.shape)
                c1 = (1 - at_next).sqrt() * eta
                c2 = (1 - at_next).sqrt() * ((1 - eta ** 2) ** 0.5)
                xt = at_next.sqrt() * x0_hat + c1 * next(tape) + c2 * et
                x0_last = x0_t
                if record is not None:
                    record(k, "et", et), record(k, "x0_t", x0_t), record(k, "xt_next", xt)
            else:
                xt = at_next.sqrt() * x0_last + next(tape) * (1 - at_next).sqrt()
    return xt, x0_last


def simplified_ddnm(x, model, betas, eta, A, Ap, y, sigma_y, noise, T_sampling=100, travel_length=1,
                    travel_repeat=1, num_timesteps=1000):
    skip = num_timesteps // T_sampling
    n = x.shape[0]
    times = schedule.jump_times(T_sampling, travel_length, travel_repeat)
    x0_last, xt = None, x
    tape = iter(noise)
    with torch.no_grad():
        for i, j in zip(times[:-1], times[1:]):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = schedule.alpha_bar(betas, j)
            if j < i:
                t = torch.ones(n) * i
                at = schedule.alpha_bar(betas, i)
                sigma_t = (1 - at_next ** 2).sqrt()
                et = model(xt, t)
                if et.shape[1] == 6:
                    et = et[:, :3]
                x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
                if sigma_t >= at_next * sigma_y:
                    lambda_t = 1.0
                    gamma_t = (sigma_t ** 2 - (at_next * sigma_y) ** 2).sqrt()
                else:
                    lambda_t = sigma_t / (at_next * sigma_y)
                    gamma_t = 0.0
                x0_hat = x0_t - lambda_t * Ap(A(x0_t) - y)
                c1 = (1 - at_next).sqrt() * eta
                c2 = (1 - at_next).sqrt() * ((1 - eta ** 2) ** 0.5)
                xt = at_next.sqrt() * x0_hat + gamma_t * (c1 * next(tape) + c2 * et)
                x0_last = x0_t
            else:
                xt = at_next.sqrt() * x0_last + next(tape) * (1 - at_next).sqrt()
    return xt, x0_last


def mean_upsample(x, scale):
    """guided_diffusion/diffusion.py:27-31."""
    return x.repeat_interleave(scale, dim=2).repeat_interleave(scale, dim=3)


def psnr(x, x_orig):
    """guided_diffusion/diffusion.py:599-602 on [-1,1]-range tensors (per image)."""
    a = torch.clamp((x + 1) / 2, 0, 1)
    b = torch.clamp((x_orig + 1) / 2, 0, 1)
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return 10 * torch.log10(1 / mse)


def ddnm_plus_diffusion(x, model, betas, eta, A_funcs, y, sigma_y, noise, T_sampling=100, travel_length=1,
                        travel_repeat=1, num_timesteps=1000):
    """functions/svd_ddnm.py:80-164 (Eq. 17 / Eq. 51): spectral lambda_t on the correction, Lambda_noise on
    the stochastic term.  `sigma_y` is the already doubled value (diffusion.py:524)."""
    skip = num_timesteps // T_sampling
    n = x.shape[0]
    times = schedule.jump_times(T_sampling, travel_length, travel_repeat)
    x0_last, xt = None, x
    tape = iter(noise)
    with torch.no_grad():
        for i, j in zip(times[:-1], times[1:]):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = schedule.alpha_bar(betas, j)
            if j < i:
                t = torch.ones(n) * i
                at = schedule.alpha_bar(betas, i)
                et = model(xt, t)
                if et.shape[1] == 6:
                    et = et[:, :3]
                x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
                sigma_t = (1 - at_next).sqrt()
                a = at_next.sqrt()
                corr = A_funcs.A_pinv(A_funcs.A(x0_t.reshape(n, -1)) - y.reshape(n, -1)).reshape(n, -1)
                x0_hat = x0_t - A_funcs.Lambda(corr, a, sigma_y, sigma_t, eta).reshape(*x0_t.shape)
                nz = A_funcs.Lambda_noise(next(tape).reshape(n, -1), a, sigma_y, sigma_t, eta, et.reshape(n, -1))
                xt = a * x0_hat + nz.reshape(*x0_t.shape)
                x0_last = x0_t
            else:
                xt = at_next.sqrt() * x0_last + next(tape) * (1 - at_next).sqrt()
    return xt, x0_last
