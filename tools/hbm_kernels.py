#!/usr/bin/env python
"""Target of the HBM-fringe passes (`hbm_frac = bytes / (t * 6.29 TB/s)` per HBM-bound kernel):
launches every HBM-bound kernel of the hot path a few times at the benchmarked shapes -- the sampler steps of the six
BASELINE operators at B = 8 (stub noise prediction), re-noise, finalize + PSNR, the FWHT passes, the fp16 and fp32
GroupNorm backward, the 3-channel output convolution, the FiLM row-streaming projection, GroupNorm apply / pooling -- after
a marker launch (`finalize_psnr` on an 8 x 8 image).  Run it under `rocprofv3 --kernel-trace` and under
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes); tools/pmc_hbm_summary.py merges them.  It also writes the
ALGORITHMIC bytes per launch (every operand once) to gpurun_out/hbm_kernels_algorithmic.json."""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ddnm_amd import _lib, ops  # noqa: E402
from ddnm_amd.functions.svd_operators import build_operator  # noqa: E402

dev = "cuda"
B, R = 8, 256
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.make_config()
g = torch.Generator().manual_seed(1)
x_orig = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev)
xt = torch.randn(B, 3, R, R, device=dev)
et = torch.randn(B, 3, R, R, device=dev)
noise = torch.randn(B, 3, R, R, device=dev)
x0, out = torch.empty_like(xt), torch.empty_like(xt)
s = ops.step_scalars(torch.tensor(0.5), torch.tensor(0.6), 0.85)
E = B * 3 * R * R                      # elements of one sampler-state tensor
alg = {}

mask = bench.real_inpainting_mask()                    # exp/inp_masks/mask.npy of the reference (bit-packed copy)
opsrc = {"sr_averagepooling": 4, "colorization": 0, "denoising": 0, "sr_bicubic": 4, "cs_walshhadamard": 0.25}
operators = {}
for deg, scale in opsrc.items():
    operators[deg] = build_operator(deg, scale, cfg, dev, perm=torch.randperm(R * R, generator=g).to(dev) if deg == "cs_walshhadamard" else None)
from ddnm_amd.functions.svd_operators import Inpainting  # noqa: E402
mk = mask.reshape(-1)
r = torch.nonzero(mk == 0).long().reshape(-1) * 3
operators["inpainting"] = Inpainting(3, R, torch.cat([r, r + 1, r + 2], 0), dev)
ys = {k: op.A(x_orig) for k, op in operators.items()}

# GroupNorm backward operands (classifier, 128 channels at 256^2)
C = 128
L = _lib.lib()
xa16 = torch.randn(B, R, R, C, device=dev).half()
da16 = torch.randn(B, R, R, C, device=dev).half()
ad16 = torch.randn(B, R, R, C, device=dev).half()
ws = ops.GroupNormWorkspace(dev, B, C, B * ops.gn_nchunk(R * R, C) * 64)
keep16 = {}
ops.group_norm_affine(xa16, None, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-5, ws, keep=keep16)
nchunk = L.ddnm_gn_bwd_nchunk(R * R, C)
partial = torch.empty(B * nchunk * 64, dtype=torch.float64, device=dev)
coef = torch.empty(B * 64, device=dev)
dx16 = torch.empty_like(xa16)
xa32, da32, ad32 = xa16[:2].float(), da16[:2].float(), ad16[:2].float()
keep32 = {}
ops.group_norm_affine(xa32, None, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-5, ws, keep=keep32)
dx32 = torch.empty_like(xa32)
EA = B * R * R * C
# celeba output convolution 128 -> 3 (fp32 NHWC in, NCHW out), FiLM projection rows (ADM: 51712 x 1024 fp32)
h32 = torch.randn(B, R, R, C, device=dev)
w_out = ops.pack_conv_weight(torch.randn(3, C, 3, 3, device=dev) * 0.03)
gn_out = (torch.rand(B * C, device=dev) + 0.5, torch.randn(B * C, device=dev) * 0.1)
emb = torch.randn(4, 1024, device=dev)
Wf, bf = torch.randn(51712, 1024, device=dev) * 0.03, torch.zeros(51712, device=dev)

torch.cuda.synchronize()
a = torch.rand(1, 3, 8, 8, device=dev)
ops.finalize_psnr(a, a.clone())            # marker
torch.cuda.synchronize()
for _ in range(REP):
    for deg, op in operators.items():
        y = ys[deg].reshape(B, -1).float().contiguous()
        op.ddnm_step(xt, et, noise, y, s, x0, out)
    ops.renoise(x0, noise, 0.8, 0.6, out=out)
    ops.finalize_psnr(out, x_orig)
    _lib.check(L.ddnm_gn_bwd_h16(xa16.data_ptr(), da16.data_ptr(), 0, keep16["scale"].data_ptr(), keep16["shift"].data_ptr(),
                                 keep16["mean_rstd"].data_ptr(), 1, ad16.data_ptr(), 0, B, R, R, C, 32, partial.data_ptr(),
                                 nchunk, coef.data_ptr(), dx16.data_ptr(), ops._stream()), "gn_bwd_h16")
    _lib.check(L.ddnm_gn_bwd_f32(xa32.data_ptr(), da32.data_ptr(), 0, keep32["scale"].data_ptr(), keep32["shift"].data_ptr(),
                                 keep32["mean_rstd"].data_ptr(), 1, ad32.data_ptr(), 0, 2, R, R, C, 32, partial.data_ptr(),
                                 nchunk, coef.data_ptr(), dx32.data_ptr(), ops._stream()), "gn_bwd_f32")
    ops.conv2d(h32, w_out, 3, 3, gn=gn_out, gn_silu=True, bias=None, out_nchw=True)
    ops.linear(emb, Wf, bf, silu_in=True)
    ops.gn_apply16(xa16, None, (gn_out[0], gn_out[1]), True)
    ops.gn_apply16(xa16, None, (gn_out[0], gn_out[1]), True, pool=True)
torch.cuda.synchronize()

f4 = 4.0
alg = {
    "step_sr4_kernel": (5 * E + E / 16) * f4,            # read xt, et, noise (+ y), write x0, xt'
    "step_color_kernel": (5 * E + E / 3) * f4,
    "step_inpaint_kernel": (5 * E + E) * f4,              # + y (kept pixels) and the rank table
    "step_denoise_kernel": (6 * E) * f4,
    "step_x0_kernel": 3 * E * f4,
    "step_combine_kernel": 5 * E * f4,                    # x0, proj, noise, et -> xt' (cs_walshhadamard: + A^+ y)
    "renoise_kernel": 3 * E * f4,
    "finalize_psnr_kernel": 3 * E * f4,
    "fwht_rows": 2 * E * f4,
    "fwht_cols": 2 * E * f4,
    "gn_bwd_reduce_kernelIDF16": 2 * EA * 2.0,
    "gn_bwd_apply_kernelIDF16": 4 * EA * 2.0,
    "gn_bwd_reduce_kernelIf": 2 * (EA / 4) * 4.0,
    "gn_bwd_apply_kernelIf": 4 * (EA / 4) * 4.0,
    "conv3x3_small_cout_f32_kernel": EA * 4.0 + E * 4.0,
    "linear_rows_kernel": 51712 * 1024 * 4.0,
    "gn_apply_h16_kernel": None,                           # two forms (plain 4 B/element, pooled 2.5): reported without a fraction
}
os.makedirs("/root/repo/gpurun_out", exist_ok=True)
json.dump(alg, open("/root/repo/gpurun_out/hbm_kernels_algorithmic.json", "w"), indent=1)
print("done")
