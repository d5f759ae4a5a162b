#!/usr/bin/env python
"""Development probe: ADM (imagenet_256) UNet forward time at B=4, fp32 kernels vs fp16-operand torso,
with the per-variant convolution timing table (HIP events)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops  # noqa: E402
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = dict(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
          learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m = create_model(**kw)
m.load_state_dict(m.random_state_dict(1))
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
FLOPS = 2242.87e9 * B
for mode in ("fp32", "fp16"):
    if mode == "fp16":
        m.convert_to_fp16()
    for _ in range(2):
        m(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        m(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mode}: {dt * 1e3:.1f} ms / forward (B={B})  -> {FLOPS / dt / 1e12:.1f} TFLOP/s whole-forward, "
          f"{B / (dt * 100):.3f} img/s at 100 steps")
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    m(x, t)
    ops.set_kernel_timer(None)
    for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
        print(f"    {k:30s} {v['launches']:4d} launches {v['ms']:8.2f} ms  {v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TFLOP/s")
