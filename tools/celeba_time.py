#!/usr/bin/env python
"""celeba_hq Model forward wall time at B = 8 (fp32): eager vs captured graph with one / two half-batch streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ddnm_amd.guided_diffusion.models import Model  # noqa: E402

B = int(os.environ.get("B", "8"))
m = Model(bench.make_config())
m.load_state_dict(m.random_state_dict(1234))
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
outs = {}
for mode in ("eager", "graph1", "graph2", "eager"):
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
    print(f"{mode}: {dt * 1e3:.2f} ms / forward at B={B}  ->  {B * 498.35e9 / dt / 1e12:.1f} TFLOP/s", flush=True)
ref = outs["eager"].double()
for k, v in outs.items():
    print(f"rel-L2 {k} vs eager: {((v.double() - ref).norm() / ref.norm()).item():.3e}")
