#!/usr/bin/env python
"""Celeba UNet forwards at B=8 (the bench workload's model) -- a short target for rocprofv3 passes.  Set-up and one
warm-up forward come first, then the marker launch (`finalize_psnr_kernel`) the summary tools cut at."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ddnm_amd import ops  # noqa: E402
from ddnm_amd.guided_diffusion.models import Model  # noqa: E402

cfg = bench.make_config()
m = Model(cfg)
m.load_state_dict(m.random_state_dict(1234))
x = torch.randn(8, 3, 256, 256, device="cuda")
t = torch.full((8,), 500.0, device="cuda")
m(x, t)
torch.cuda.synchronize()
a = torch.rand(1, 3, 8, 8, device="cuda")
ops.finalize_psnr(a, a.clone())            # marker: everything after this launch is "the forwards"
torch.cuda.synchronize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    m(x, t)
torch.cuda.synchronize()
print("done")
