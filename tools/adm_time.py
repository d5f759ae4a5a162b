#!/usr/bin/env python
"""ADM forward wall time (fp16 torso): eager fp16-activation path, first-generation path (DDNM_ADM_GEN1=1), and the
captured-graph replays with one / two half-batch streams (ddnm_amd/graph.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

B = int(os.environ.get("B", "4"))
MODES = (os.environ.get("MODES") or "h16,gen1,graph1,graph2,h16").split(",")
m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
outs = {}
for mode in MODES:
    os.environ["DDNM_ADM_GEN1"] = "1" if mode == "gen1" else "0"
    m.disable_graphs()
    if mode.startswith("graph"):
        m.enable_graphs(two_streams=(mode == "graph2"))
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
ref = outs[MODES[0]].double()
for k, v in outs.items():
    print(f"rel-L2 {k} vs {MODES[0]}: {((v.double() - ref).norm() / ref.norm()).item():.3e}")
