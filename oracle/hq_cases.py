"""Seeded cases of the `hq_demo` sampler shared by the golden generator (real reference), the oracle tests (CPU)
and the engine tests (GPU).  TEST INFRASTRUCTURE ONLY.

The denoiser is a 5-level ADM UNet with 32 base channels at the fixed 256x256 tile size (the tile size is
hard-coded in the reference, hq_demo/guided_diffusion/gaussian_diffusion.py:594,664-689), the classifier a
128-wide EncoderUNetModel (the GroupNorm backward kernels need >= 4 channels per group); weights come from oracle.weights with the usual seeds.
"""
import torch

from . import cases, weights

# scheduler.get_schedule_jump keyword sets (confs/inet256.yml:57-61 is "paper")
SCHEDULES = {
    "paper": dict(t_T=100, n_sample=1, jump_length=10, jump_n_sample=3),
    "short": dict(t_T=8, n_sample=1, jump_length=2, jump_n_sample=2),
    "resample": dict(t_T=12, n_sample=2, jump_length=4, jump_n_sample=2),
    "nested": dict(t_T=20, n_sample=1, jump_length=8, jump_n_sample=2, jump2_length=2, jump2_n_sample=2),
}
RESPACINGS = {"100": (1000, "100"), "8": (1000, "8"), "sections": (1000, "10,15,20")}

_MODEL = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,1,2,2,4", attention_resolutions="16",
              num_head_channels=32, learn_sigma=True, class_cond=True, use_scale_shift_norm=True, resblock_updown=True)
_SHORT = dict(timestep_respacing="8", schedule=SCHEDULES["short"])

CASES = {
    # 72x80 measurement -> 288x320 result: 2x2 tiles, both directions end with an irregular shift
    "sr4_tiles": dict(deg="sr_averagepooling", scale=4, resize_y=True, sigma_y=0.0, gt_hw=(72, 80), **_SHORT),
    # noisy SR with classifier guidance, 1x2 regular tiles; sigma_y large enough to visit both Eq. 19 branches
    "sr2_noisy_guided": dict(deg="sr_averagepooling", scale=2, resize_y=True, sigma_y=0.5, gt_hw=(128, 192),
                             classifier=True, **_SHORT),
    "colorization": dict(deg="colorization", scale=4, resize_y=False, sigma_y=0.0, gt_hw=(256, 256), **_SHORT),
    "sr_color": dict(deg="sr_color", scale=4, resize_y=False, sigma_y=0.0, gt_hw=(256, 384), **_SHORT),
}
# face256-only degradations (gaussian_diffusion.py:600-622): unconditional model, keep-mask from the data loader
FACE_CASES = {
    "face_inpainting": dict(deg="inpainting", scale=4, resize_y=False, sigma_y=0.0, gt_hw=(256, 256), face=True, **_SHORT),
    "face_mask_color_sr": dict(deg="mask_color_sr", scale=4, resize_y=False, sigma_y=0.0, gt_hw=(256, 256), face=True,
                               **_SHORT),
}
CASES_ALL = dict(CASES, **FACE_CASES)
for _i, _c in enumerate(CASES_ALL.values()):
    _c.setdefault("classifier", False)
    _c["class"] = 950
    _c["seed"] = cases.SEED + 40 + _i


def model_config(case):
    if case.get("face"):
        return weights.adm_config(**dict(_MODEL, class_cond=False))
    return weights.adm_config(**_MODEL)


def keep_mask(case):
    """[1, 3, 256, 256] float keep-mask in {0, 1} (the loader's `gt_keep_mask`, image_datasets.py:167-181)."""
    m = cases.random_mask(256).float()
    return m.reshape(1, 1, 256, 256).repeat(1, 3, 1, 1).contiguous()


def classifier_config(case):
    return weights.classifier_config(image_size=256, classifier_width=128, classifier_depth=1)


def conf_dict(case):
    """The keys of hq_demo/confs/inet256.yml that the sampler reads, for the reduced model."""
    d = dict(name="inet256", diffusion_steps=1000, noise_schedule="linear", use_kl=False, predict_xstart=False,
             rescale_timesteps=False, rescale_learned_sigmas=False, num_heads=4, num_heads_upsample=-1, dropout=0.0,
             use_checkpoint=False, use_new_attention_order=False, use_fp16=False, clip_denoised=True, use_ddim=False,
             classifier_scale=1.0, classifier_use_fp16=False, classifier_width=128, classifier_depth=1,
             classifier_attention_resolutions="32,16,8", classifier_use_scale_shift_norm=True,
             classifier_resblock_updown=True, classifier_pool="attention", show_progress=False,
             timestep_respacing=case["timestep_respacing"], schedule_jump_params=dict(case["schedule"]))
    d.update(_MODEL)
    if case.get("face"):
        d.update(name="face256", class_cond=False)
    return d


def n_draws(case, n_tiles):
    """Gaussian draws consumed after the initial one: one per schedule transition per tile."""
    from . import hq_demo
    return n_tiles * (len(hq_demo.schedule_jump(**case["schedule"])) - 1)


def inputs(case):
    """(gt in [-1,1], x_init, noise tape)."""
    from . import hq_demo
    g = torch.Generator().manual_seed(case["seed"])
    h, w = case["gt_hw"]
    gt = torch.rand(1, 3, h, w, generator=g) * 2 - 1
    s = case["scale"] if case["resize_y"] else 1
    n_tiles = len(hq_demo.tile_plan(h * s, w * s))
    x_init = torch.randn(1, 3, 256, 256, generator=g)
    tape = [torch.randn(1, 3, 256, 256, generator=g) for _ in range(n_draws(case, n_tiles))]
    return gt, x_init, tape
