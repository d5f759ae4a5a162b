#!/usr/bin/env python
"""Per-layer convolution timing of one celeba UNet forward at B=8 (development tool): which layer shapes sit
below the dominant kernel's best rate."""
import os
import sys
from collections import OrderedDict

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
for _ in range(2):
    m(x, t)
timer = ops.KernelTimer()
ops.set_kernel_timer(timer)
REP = 5
for _ in range(REP):
    m(x, t)
ops.set_kernel_timer(None)
torch.cuda.synchronize()
agg = OrderedDict()
for (variant, flops, e0, e1), shp in zip(timer.records, timer.shapes):
    r = agg.setdefault((variant,) + shp, [0, 0.0, 0.0])
    r[0] += 1
    r[1] += flops
    r[2] += e0.elapsed_time(e1)
tot = sum(r[2] for r in agg.values()) / REP
print(f"conv time per forward: {tot:.2f} ms")
print(f"{'kernel':28s} {'B,H,W,Cin,Cout,k,s,ups,skip,gn,res':44s} {'n':>3s} {'us':>8s} {'TF':>7s} {'ms/fwd':>7s}")
for key, (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{key[0]:28s} {str(key[1:]):44s} {n // REP:3d} {ms / n * 1e3:8.1f} {fl / ms / 1e9:7.1f} {ms / REP:7.3f}")
