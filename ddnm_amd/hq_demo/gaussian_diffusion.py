"""hq_demo sampler on the HIP engine: DDPM posterior sampling with the DDNM projection (Eq. 17/19) over 256x256
tiles shifted by 128 px, each tile inheriting the already-restored strips of its left / upper neighbours
(mask-shift trick).  Drop-in for hq_demo/guided_diffusion/gaussian_diffusion.py::GaussianDiffusion.p_sample_loop
(:489-546) + respace.py::SpacedDiffusion as used by hq_demo/main.py:155-165.

Per reverse step on one tile batch [B,3,256,256] (all HBM-bound, 4 B/elem/operand):
    model (fp16-torso ADM UNet, ddnm_amd.guided_diffusion.unet)      -> eps = out[:, :3]
    ddnm_hq_x0_f32        x0 = clamp(c_recip x_t - c_recipm1 eps)
    A, A^+ kernels        A^+ A x0 (avg-pool / replicate, grey / colour, mask)
    ddnm_hq_project_f32   x0_hat = lambda A^+y + x0 - lambda A^+A x0
    ddnm_copy_rect_f32    paste restored strips of the big image into x0_hat (<= 2 launches)
    ddnm_hq_sample_f32    x_{t-1} = coef1 x0_hat + coef2 x_t [+ gamma grad] + sqrt(gamma) noise
Coefficient tables are float64 on the host, rounded to fp32 per step like `_extract_into_tensor` (:758-771).

Reference behaviours kept (see oracle/hq_demo.py): one initial x_T for ALL tiles; lambda_t = sigma_t/a_t*sigma_y in
the low-noise branch; guidance scaled by gamma_t; learned variances ignored.
"""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check
from .respace import respaced_betas
from .scheduler import get_schedule_jump

TILE, SHIFT = 256, 128
_THIRD = (ctypes.c_float * 3)(1 / 3, 1 / 3, 1 / 3)


def _p(t):
    return None if t is None else t.data_ptr()


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, use_scale):
    """gaussian_diffusion.py:71-92 (linear only, like the reference)."""
    if schedule_name != "linear":
        raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
    scale = 1000 / num_diffusion_timesteps if use_scale else 1
    return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)


def tile_plan(H, W):
    """[(h0, w0, left, top)]: tile origin and the width / height of the strips already restored by the left / upper
    neighbours (gaussian_diffusion.py:664-689 and :341-377).  The last tile of a direction whose size is not a
    multiple of 128 is aligned to the image border and therefore overlaps 256 - size % 128 restored pixels."""
    if H < TILE or W < TILE:
        raise ValueError("Please set a larger SR scale")
    rows, cols = math.ceil(H / SHIFT) - 1, math.ceil(W / SHIFT) - 1
    plan = []
    for i in range(rows):
        ragged_h = i == rows - 1 and H % SHIFT != 0
        for j in range(cols):
            ragged_w = j == cols - 1 and W % SHIFT != 0
            plan.append((H - TILE if ragged_h else SHIFT * i, W - TILE if ragged_w else SHIFT * j,
                         0 if j == 0 else (TILE - W % SHIFT if ragged_w else SHIFT),
                         0 if i == 0 else (TILE - H % SHIFT if ragged_h else SHIFT)))
    return plan


# ------------------------------------------------------------------------------------------------ image-space ops
def _new(x, c, h, w):
    return torch.empty(x.shape[0], c, h, w, dtype=torch.float32, device=x.device)


def avg_pool(x, r):
    B, C, H, W = x.shape
    y = _new(x, C, H // r, W // r)
    check(_lib.lib().ddnm_op_avgpool_f32(_p(x), _p(y), B * C, H, W, r, ops._stream()), "ddnm_op_avgpool_f32")
    return y


def mean_upsample(y, r):
    """MeanUpsample (gaussian_diffusion.py:65-69): nearest replication."""
    B, C, h, w = y.shape
    x = _new(y, C, h * r, w * r)
    check(_lib.lib().ddnm_op_upsample_f32(_p(y), _p(x), B * C, h * r, w * r, r, ops._stream()), "ddnm_op_upsample_f32")
    return x


def color2gray(x):
    """[B,3,H,W] -> [B,1,H,W] (the reference keeps three identical channels, :54-57)."""
    B, _, H, W = x.shape
    y = _new(x, 1, H, W)
    check(_lib.lib().ddnm_op_color_A_f32(_p(x), _p(y), B, H * W, _THIRD, ops._stream()), "ddnm_op_color_A_f32")
    return y


def gray2color(y):
    B, _, H, W = y.shape
    x = _new(y, 3, H, W)
    check(_lib.lib().ddnm_op_color_pinv_f32(_p(y), _p(x), B, H * W, _THIRD, ops._stream()), "ddnm_op_color_pinv_f32")
    return x


def apply_mask(x, mask_plane):
    out = torch.empty_like(x)
    hw = x.shape[2] * x.shape[3]
    check(_lib.lib().ddnm_mask_mix_f32(_p(x), None, _p(mask_plane), 1, hw, _p(out), x.numel(), 1.0, 0.0, 0.0, 0.0,
                                       ops._stream()), "ddnm_mask_mix_f32")
    return out


def copy_rect(src, sy, sx, dst, dy, dx, h, w):
    planes = src.shape[0] * src.shape[1]
    check(_lib.lib().ddnm_copy_rect_f32(_p(src), src.shape[2], src.shape[3], sy, sx, _p(dst), dst.shape[2], dst.shape[3],
                                        dy, dx, planes, h, w, ops._stream()), "ddnm_copy_rect_f32")


def degradation(deg, scale, mask=None, face=False):
    """(A, Ap) acting on NCHW fp32 tensors of any size, gaussian_diffusion.py:592-641."""
    if deg == "sr_averagepooling":
        return (lambda z: avg_pool(z, scale)), (lambda z: mean_upsample(z, scale))
    if deg == "colorization":
        return color2gray, gray2color
    if deg == "sr_color":
        return (lambda z: color2gray(avg_pool(z, scale))), (lambda z: mean_upsample(gray2color(z), scale))
    if deg in ("inpainting", "mask_color_sr") and face:
        if mask is None:
            raise ValueError("this degradation needs model_kwargs['gt_keep_mask']")
        m = mask.reshape(-1, mask.shape[-2], mask.shape[-1])[0].float().contiguous()
        if deg == "inpainting":
            return (lambda z: apply_mask(z, m)), (lambda z: apply_mask(z, m))
        return (lambda z: avg_pool(color2gray(apply_mask(z, m)), scale)), \
               (lambda z: apply_mask(gray2color(mean_upsample(z, scale)), m))
    raise NotImplementedError("degradation type not supported")


def _to_image(t):
    """tensor2im (:40-47)."""
    from PIL import Image
    img, _ = ops.finalize_psnr(t[None].contiguous() if t.dim() == 3 else t.contiguous())
    arr = (img[0].permute(1, 2, 0).cpu().numpy() * 255)
    if arr.shape[2] == 1:
        arr = np.repeat(arr, 3, axis=2)
    return Image.fromarray(arr.astype("uint8"))


def save_image(img, save_dir, idx):
    os.makedirs(save_dir, exist_ok=True)
    _to_image(img).save(os.path.join(save_dir, f"{int(idx):05d}.png"))


# ------------------------------------------------------------------------------------------------ the process
class SpacedDiffusion:
    """Respaced Gaussian diffusion (respace.py:80-122 over gaussian_diffusion.py:131-212) + the tiled DDNM sampler."""

    def __init__(self, use_timesteps, betas, conf=None, **unused):
        self.conf = conf
        self.original_num_steps = len(betas)
        self.use_timesteps = set(use_timesteps)
        new_betas, self.timestep_map = respaced_betas(betas, self.use_timesteps)
        self.betas = betas = np.asarray(new_betas, dtype=np.float64)
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = len(betas)
        ac = np.cumprod(1.0 - betas)
        ac_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod, self.alphas_cumprod_prev = ac, ac_prev
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        self.posterior_variance = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_mean_coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)

    # ---- one reverse step on a tile: p_mean_variance (:246-404) + p_sample (:430-487)
    def _p_sample(self, model, cond_fn, x, t, st, model_kwargs, noise, out):
        f32 = np.float32
        B, chw = x.shape[0], x[0].numel()
        L = _lib.lib()
        ts = torch.full((B,), self.timestep_map[t], dtype=torch.long, device=x.device)       # _WrappedModel :117-122
        eps = model(x, ts, **model_kwargs)
        ep, es = ops._et_args(eps)
        x0 = st["x0"]
        check(L.ddnm_hq_x0_f32(_p(x), ep, es, _p(x0), B, chw, float(f32(self.sqrt_recip_alphas_cumprod[t])),
                               float(f32(self.sqrt_recipm1_alphas_cumprod[t])), int(st["clip"]), ops._stream()),
              "ddnm_hq_x0_f32")
        var = f32(self.posterior_variance[t])
        sigma_t, a_t = np.sqrt(var), f32(self.posterior_mean_coef1[t])
        sigma_y = st["sigma_y"]
        if sigma_t >= a_t * f32(sigma_y):                                                     # Eq. 19 (:329-335)
            lam, gamma = f32(1.0), f32(var - (a_t * f32(sigma_y)) ** 2)
        else:
            lam, gamma = f32(f32(sigma_t / a_t) * f32(sigma_y)), f32(0.0)
        apax0 = st["Ap"](st["A"](x0))
        x0_hat = st["x0_hat"]
        check(L.ddnm_hq_project_f32(_p(x0), _p(st["apy"]), _p(apax0), _p(x0_hat), x0.numel(), float(lam), ops._stream()),
              "ddnm_hq_project_f32")
        h0, w0, left, top = st["tile"]
        if left:                                                                              # mask-shift (:341-377)
            copy_rect(st["final"], h0, w0, x0_hat, 0, 0, TILE, left)
        if top:
            copy_rect(st["final"], h0, w0, x0_hat, 0, 0, top, TILE)
        if t % 25 == 0 and st["save_path"] is not None:                                       # :379-383
            save_image(x0_hat[0], os.path.join("results", st["save_path"], st["tile_name"]), t)
        grad = None
        if cond_fn is not None:
            grad = cond_fn(x, ts, **model_kwargs).float().contiguous()                        # :412-427
        noise_scale = f32(0.0) if t == 0 else np.sqrt(gamma)
        check(L.ddnm_hq_sample_f32(_p(x0_hat), _p(x), _p(grad), _p(noise), _p(out), x.numel(),
                                   float(f32(self.posterior_mean_coef1[t])), float(f32(self.posterior_mean_coef2[t])),
                                   float(gamma), float(noise_scale), ops._stream()), "ddnm_hq_sample_f32")
        return out

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=True, return_all=False, conf=None, noise_tape=None):
        """Restores model_kwargs['gt'] (the measurement when `resize_y`) at its full size; returns the dict
        {'sample': full-size result, 'x0_t': last tile, 'gt': ...} if `return_all` else the full-size result.
        `noise_tape` (test hook) replaces the Gaussian draws after the initial one."""
        conf = self.conf if conf is None else conf
        conf = {} if conf is None else conf
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not used by hq_demo/main.py")
        device = torch.device("cuda") if device is None else torch.device(device)
        kw = dict(model_kwargs or {})
        gt, scale = kw["gt"].to(device).float().contiguous(), int(kw["scale"])
        face = conf.get("name") == "face256"
        t_shift = conf.get("inpa_inj_time_shift") or 1
        if 256 % scale != 0:
            raise ValueError("Please set a SR scale divisible by 256")
        if gt.shape[2] != 256 and face:
            raise ValueError("Only support output size 256x256 for face images")
        if kw.get("resize_y"):
            gt = mean_upsample(gt, scale)
        A, Ap = degradation(kw["deg"], scale, kw.get("gt_keep_mask"), face)
        y_full = A(gt)
        apy_full = Ap(y_full)
        H, W = apy_full.shape[2], apy_full.shape[3]
        plan = tile_plan(H, W)
        save_path = kw.get("save_path")
        if save_path is not None:
            save_image(apy_full[0], os.path.join("results", save_path, "Apy"), 0)
            save_image(y_full[0], os.path.join("results", save_path, "y"), 0)
        final = torch.zeros_like(apy_full)
        B = shape[0]
        x = (torch.randn(*shape, device=device) if noise is None else noise.to(device)).float().contiguous()
        bufs = [torch.empty_like(x), torch.empty_like(x)]
        st = {"A": A, "Ap": Ap, "sigma_y": float(kw.get("sigma_y", 0.0)), "clip": clip_denoised, "final": final,
              "x0": torch.empty_like(x), "x0_hat": torch.empty_like(x), "apy": torch.empty_like(x),
              "save_path": save_path}
        sched = dict(conf.get("schedule_jump_params") or dict(t_T=self.num_timesteps, n_sample=1, jump_length=10,
                                                               jump_n_sample=3))
        times = get_schedule_jump(**sched)
        tape = None if noise_tape is None else iter(noise_tape)
        rows, cols = math.ceil(H / SHIFT) - 1, math.ceil(W / SHIFT) - 1
        kw.update(A=A, Ap=Ap, H_target=H, W_target=W, shift_h_total=rows, shift_w_total=cols)
        bar = None
        if progress:
            from tqdm.auto import tqdm
            bar = tqdm(total=len(plan), desc="total shifts")
        k = 0
        with torch.no_grad():
            for n, tile in enumerate(plan):
                h0, w0 = tile[0], tile[1]
                copy_rect(apy_full, h0, w0, st["apy"], 0, 0, TILE, TILE)
                st["tile"], st["tile_name"] = tile, f"{n // cols}_{n % cols}"
                kw.update(shift_h=n // cols, shift_w=n % cols, Apy=st["apy"], x_temp=final)
                for t_last, t_cur in zip(times[:-1], times[1:]):
                    nz = torch.randn_like(x) if tape is None else next(tape).to(device)
                    out = bufs[k & 1]
                    k += 1
                    if t_cur < t_last:
                        x = self._p_sample(model, cond_fn, x, t_last, st, kw, nz, out)
                    else:                                                   # `undo` with t_shift = 1 (:197-206,729-735)
                        beta = np.float32(self.betas[t_last + t_shift])
                        check(_lib.lib().ddnm_axpby_f32(_p(x), _p(nz), _p(out), x.numel(),
                                                        float(np.sqrt(np.float32(1) - beta)), float(np.sqrt(beta)),
                                                        ops._stream()), "ddnm_axpby_f32")
                        x = out
                copy_rect(st["x0_hat"], 0, 0, final, h0, w0, TILE, TILE)                      # :737-746
                if bar is not None:
                    bar.update(1)
        if bar is not None:
            bar.close()
        if save_path is not None:
            save_image(final[0], os.path.join("results", save_path, "final"), 0)
        result = {"sample": final, "x0_t": st["x0_hat"], "gt": kw.get("gt"), "y": y_full, "Apy": apy_full}
        return result if return_all else final

    p_sample_loop_progressive = p_sample_loop
