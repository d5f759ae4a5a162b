"""Seeded synthetic test cases shared by tests/golden/make_golden.py (which runs
the REAL reference on them) and by the parity tests (which run the oracle
restatement and the HIP engine on them).  TEST INFRASTRUCTURE.

All randomness comes from CPU `torch.Generator`s (mt19937), so the same inputs
are reproduced bit-for-bit on every box with this torch build
(seeds fixed at 1234).
"""
import torch

from . import operators as O
from . import schedule, weights

SEED = 1234

SMALL_NET = dict(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32)
MID_NET = dict(ch=128, ch_mult=(1, 1, 2), num_res_blocks=2, attn_resolutions=(16,), resolution=64)
FULL_NET = dict()   # configs/celeba_hq.yml


def celeba_net(kind):
    cfg = weights.celeba_config(**{"small": SMALL_NET, "mid": MID_NET, "full": FULL_NET}[kind])
    return cfg, weights.celeba_state_dict(cfg, SEED)


def forward_inputs(cfg, batch, seed=SEED + 1):
    g = torch.Generator().manual_seed(seed)
    r = cfg.data.image_size
    x = torch.randn(batch, cfg.data.channels, r, r, generator=g)
    t = torch.tensor([430.0, 990.0, 0.0, 10.0, 770.0, 250.0, 120.0, 640.0][:batch])
    return x, t


def operator_input(img_dim, batch, seed=SEED + 2):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, img_dim, img_dim, generator=g)


def random_mask(img_dim, seed=SEED + 3):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(img_dim, img_dim, generator=g) > 0.26).long()


def wh_perm(img_dim, seed=SEED + 4):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(img_dim ** 2, generator=g)


def make_operator(name, img_dim, mask=None, ratio=4):
    """Oracle operator objects for the --deg names of guided_diffusion/diffusion.py:451-523."""
    if name == "sr_averagepooling":
        return O.SuperResolution(3, img_dim, ratio)
    if name == "sr_bicubic":
        k = O.bicubic_kernel(4)
        return O.SRConv(k / k.sum(), 3, img_dim, stride=4)
    if name == "colorization":
        return O.Colorization(img_dim)
    if name == "inpainting":
        mask = random_mask(img_dim) if mask is None else mask
        return O.Inpainting(3, img_dim, O.Inpainting.missing_from_mask(mask))
    if name == "cs_walshhadamard":
        return O.WalshHadamardCS(3, img_dim, 4, wh_perm(img_dim))
    if name == "denoising":
        return O.Denoising(3, img_dim)
    if name == "deblur_uni":
        return O.Deblurring(torch.Tensor([1 / 9] * 9), 3, img_dim)
    if name == "deblur_gauss":
        k = O.gaussian_taps(10, 2)
        return O.Deblurring(k / k.sum(), 3, img_dim)
    if name == "deblur_aniso":
        k2, k1 = O.gaussian_taps(20, 4), O.gaussian_taps(1, 4)
        return O.Deblurring2D(k1 / k1.sum(), k2 / k2.sum(), 3, img_dim)
    if name == "cs_blockbased":
        return O.CS(3, img_dim, 0.25, O.gauss_matrix(SEED + 21))
    raise ValueError(name)


def sampler_case(cfg, batch, n_iters, seed=SEED + 5):
    """x_orig in [-1,1], x_T, and the noise tape (one tensor per loop iteration)."""
    g = torch.Generator().manual_seed(seed)
    r = cfg.data.image_size
    x_orig = torch.rand(batch, 3, r, r, generator=g) * 2 - 1
    x_T = torch.randn(batch, 3, r, r, generator=g)
    tape = [torch.randn(batch, 3, r, r, generator=g) for _ in range(n_iters)]
    return x_orig, x_T, tape


def betas():
    return schedule.beta_schedule("linear", 1e-4, 0.02, 1000)


# ---- ADM (ImageNet) nets ------------------------------------------------------------------------
ADM_SMALL = dict(image_size=32, num_channels=128, num_res_blocks=1, channel_mult="1,2", attention_resolutions="16",
                 class_cond=True)
ADM_MID = dict(image_size=64, num_channels=128, num_res_blocks=2, channel_mult="1,1,2", attention_resolutions="32,16")
ADM_FULL = dict()      # configs/imagenet_256.yml (552.81 M parameters)


def adm_net(kind):
    cfg = weights.adm_config(**{"small": ADM_SMALL, "mid": ADM_MID, "full": ADM_FULL}[kind])
    return cfg, weights.adm_state_dict(cfg, SEED)


def adm_forward_inputs(cfg, batch, seed=SEED + 6):
    g = torch.Generator().manual_seed(seed)
    r = cfg.model.image_size
    x = torch.randn(batch, 3, r, r, generator=g)
    t = torch.tensor([430.0, 990.0, 0.0, 10.0][:batch])
    y = torch.tensor([951, 3, 17, 999][:batch]) if cfg.model.class_cond else None
    return x, t, y


# ---- simplified path (guided_diffusion/diffusion.py:211-415; batch size 1) -----------------------------------
# sr_averagepooling / inpainting / mask_color_sr are tied to 256x256 there (AdaptiveAvgPool2d(256 // scale) :253, the
# 256x256 exp/inp_masks/mask.npy :257), so those run a 5-level net at 256 px; colorization / denoising run at 32 px.
SIMPLIFIED_CASES = [
    dict(name="colorization", deg="colorization", res=32, sigma_y=0.0, T=10, travel=(2, 2)),
    dict(name="denoising_noisy", deg="denoising", res=32, sigma_y=0.3, T=10, travel=(2, 2)),
    dict(name="sr_averagepooling", deg="sr_averagepooling", res=256, sigma_y=0.0, T=6, travel=(2, 2)),
    dict(name="inpainting_noisy", deg="inpainting", res=256, sigma_y=0.3, T=6, travel=(2, 2)),
    dict(name="mask_color_sr", deg="mask_color_sr", res=256, sigma_y=0.0, T=6, travel=(2, 2)),
]


def simplified_net(res):
    if res == 32:
        return celeba_net("small")
    cfg = weights.celeba_config(ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=1, attn_resolutions=(16,), resolution=res)
    return cfg, weights.celeba_state_dict(cfg, SEED)


# ---- full BASELINE configurations (tests/golden/full_*.npz) ---------------------------------
# c2b8: configs[1] itself (celeba_hq Model, sr_bicubic 4x, B=8, T=100);  c3 / c4 / c5: one image of configs[2..4] on
# the full 552 M-parameter ADM net (colorization T=100; inpainting with exp/inp_masks/mask.npy, travel l=10 r=3 =
# 460 loop iterations / 280 NFE; class-conditional + classifier guidance + cs_walshhadamard 0.25).
FULL_CASES = {
    "c2b8": dict(net="celeba", deg="sr_bicubic", batch=8, T=100, travel=(1, 1), class_cond=False, record=(10, 50, 90)),
    "c3": dict(net="adm", deg="colorization", batch=1, T=100, travel=(1, 1), class_cond=False, record=(10, 50, 90)),
    "c4": dict(net="adm", deg="inpainting", batch=1, T=100, travel=(10, 3), class_cond=False, record=(9, 229, 449)),
    "c5": dict(net="adm", deg="cs_walshhadamard", batch=1, T=100, travel=(1, 1), class_cond=True, record=(10, 50, 90)),
    # configs[2] at the per-GPU batch bench.py times (B = 4 selects other conv16 launch plans than B = 1), 10 of the
    # 100 steps: ~40 evaluations of the full net through the reference on CPU
    "c3b4": dict(net="adm", deg="colorization", batch=4, T=10, travel=(1, 1), class_cond=False, record=(1, 5, 8)),
    # configs[3] / configs[4] at their per-GPU benchmarked batches (B = 4 / B = 8), 10 steps; c4b4 keeps a time-travel
    # schedule (l = 2, r = 2: 26 iterations) and the repository mask, c5b8 the classifier guidance at B = 8
    "c4b4": dict(net="adm", deg="inpainting", batch=4, T=10, travel=(2, 2), class_cond=False, record=(1, 13, 24)),
    "c5b8": dict(net="adm", deg="cs_walshhadamard", batch=8, T=10, travel=(1, 1), class_cond=True, record=(1, 5, 8)),
    # evaluation.sh:18: `--config celeba_hq.yml --deg sr_averagepooling --deg_scale 16 --sigma_y 0.2 --add_noise`: the
    # runner doubles sigma_y (diffusion.py:524) and takes ddnm_plus_diffusion (:590); 16x16 patches (n = 256 singular
    # vectors per patch: the large-ratio branch of the SuperResolution spectral surface)
    "c2sr16": dict(net="celeba", deg="sr_averagepooling", ratio=16, sigma_y=0.4, batch=2, T=20, travel=(1, 1),
                   class_cond=False, record=(2, 10, 17)),
}


def full_case(name):
    """(case dict, cfg, state_dict, x_orig, x_T, tape) of one full BASELINE configuration."""
    c = FULL_CASES[name]
    if c["net"] == "celeba":
        cfg, sd = celeba_net("full")
    else:
        cfg = weights.adm_config(class_cond=c["class_cond"])
        sd = weights.adm_state_dict(cfg, SEED)
    tt = cfg.time_travel
    tt.T_sampling, tt.travel_length, tt.travel_repeat = c["T"], c["travel"][0], c["travel"][1]
    n_it = len(schedule.jump_times(c["T"], *c["travel"])) - 1
    x_orig, x_T, tape = sampler_case(cfg, c["batch"], n_it)
    return c, cfg, sd, x_orig, x_T, tape
