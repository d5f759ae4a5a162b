#!/usr/bin/env python
"""Development probe: time of the 128 -> 3 output convolution (celeba conv_out) at B = 8."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops
x = torch.randn(8, 256, 256, 128, device="cuda")
w = ops.pack_conv_weight(torch.randn(3, 128, 3, 3, device="cuda") * 0.03)
b = torch.randn(3, device="cuda")
sc, sh = torch.randn(8, 128, device="cuda"), torch.randn(8, 128, device="cuda")
f = lambda: ops.conv2d(x, w, 3, 3, bias=b, gn=(sc, sh), gn_silu=True, out_nchw=True)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"conv_out 128->3 @256^2 B=8: {us:.1f} us  ({268.4e6 / us / 1e6:.2f} TB/s of input)")
