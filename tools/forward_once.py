#!/usr/bin/env python
"""Two celeba UNet forwards at B=8 (the bench workload's model) -- a short target for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ddnm_amd.guided_diffusion.models import Model  # noqa: E402

cfg = bench.make_config()
m = Model(cfg)
m.load_state_dict(m.random_state_dict(1234))
x = torch.randn(8, 3, 256, 256, device="cuda")
t = torch.full((8,), 500.0, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    m(x, t)
torch.cuda.synchronize()
print("done")
