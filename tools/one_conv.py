#!/usr/bin/env python
"""One convolution shape, a few launches (development tool: target for rocprofv3 --pmc passes).
usage: one_conv.py [Cin Cout H gn res f16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]] + [128, 128, 256, 0, 0, 0][len(sys.argv) - 1:]
Cin, Cout, H, gn, res, f16 = a
B = 8 if not f16 else 4
x = torch.randn(B, H, H, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
w32 = ops.pack_conv_weight(w)
w16 = ops.pack_conv_weight_f16(w) if f16 else None
sc, sh = torch.randn(B, Cin, device="cuda"), torch.randn(B, Cin, device="cuda")
r = torch.randn(B, H, H, Cout, device="cuda") if res else None
out = torch.empty(B, H, H, Cout, device="cuda")
for _ in range(6):
    ops.conv2d(x, w32, Cout, 3, gn=(sc, sh) if gn else None, res=r, out=out, weight_f16=w16)
torch.cuda.synchronize()
print("done")
