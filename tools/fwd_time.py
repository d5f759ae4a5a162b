#!/usr/bin/env python
"""Wall time of the celeba UNet forward at B=8 (HIP events, 20 forwards) -- quick A/B of environment knobs (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ddnm_amd.guided_diffusion.models import Model  # noqa: E402

m = Model(bench.make_config())
m.load_state_dict(m.random_state_dict(1234))
x = torch.randn(8, 3, 256, 256, device="cuda")
t = torch.full((8,), 500.0, device="cuda")
for _ in range(3):
    m(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    m(x, t)
e1.record()
torch.cuda.synchronize()
print(f"forward B=8: {e0.elapsed_time(e1) / 20:.3f} ms")
