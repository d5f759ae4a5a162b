#!/usr/bin/env python
"""ADM forward wall time at B=4 (fp16 torso), fp16-activation path vs first-generation path (DDNM_ADM_GEN1=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

B = int(os.environ.get("B", "4"))
m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
outs = {}
for mode in ("h16", "gen1", "h16"):
    os.environ["DDNM_ADM_GEN1"] = "1" if mode == "gen1" else "0"
    for _ in range(3):
        e = m(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        e = m(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    outs[mode] = e
    print(f"{mode}: {dt * 1e3:.2f} ms / forward at B={B}  ->  {B * 2242.87e9 / dt / 1e12:.0f} TFLOP/s", flush=True)
d = (outs["h16"].double() - outs["gen1"].double()).norm() / outs["gen1"].double().norm()
print(f"rel-L2 h16 vs gen1: {d.item():.3e}")
