"""Shared helpers of the GPU parity tests."""
import numpy as np
import torch


def real_mask(golden_dir):
    """exp/inp_masks/mask.npy of the reference, stored bit-packed in tests/golden/inp_mask.npz."""
    g = np.load(f"{golden_dir}/inp_mask.npz")
    shape = tuple(g["shape"])
    bits = np.unpackbits(g["packed"])[: shape[0] * shape[1]]
    return torch.from_numpy(bits.reshape(shape).astype(np.int64))


def engine_operator(name, d, mask=None, device="cuda", ratio=4):
    """Engine (HIP) operator for a --deg name, built from the same seeded ingredients as
    oracle.cases.make_operator."""
    from ddnm_amd.functions import svd_operators as E
    from oracle import cases
    if name == "sr_averagepooling":
        return E.SuperResolution(3, d, ratio, device)
    if name == "sr_averagepooling_x16":              # evaluation.sh:18 (`--deg_scale 16`): 256 entries per site
        return E.SuperResolution(3, d, 16, device)
    if name == "sr_bicubic":
        return E.SRConv(E.bicubic_kernel(4), 3, d, device, stride=4)
    if name == "colorization":
        return E.Colorization(d, device)
    if name == "inpainting":
        mask = cases.random_mask(d) if mask is None else mask
        r = torch.nonzero(mask.reshape(-1) == 0).long().reshape(-1) * 3
        return E.Inpainting(3, d, torch.cat([r, r + 1, r + 2], dim=0), device)
    if name == "cs_walshhadamard":
        return E.WalshHadamardCS(3, d, 4, cases.wh_perm(d), device)
    if name == "denoising":
        return E.Denoising(3, d, device)
    if name in ("deblur_uni", "deblur_gauss", "deblur_aniso"):
        cfg = cases.weights.celeba_config(resolution=d)
        return E.build_operator(name, 0, cfg, device)
    if name == "cs_blockbased":
        from oracle import operators as O
        return E.CS(3, d, 0.25, device, gauss=O.gauss_matrix(cases.SEED + 21))
    raise ValueError(name)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
