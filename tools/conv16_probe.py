#!/usr/bin/env python
"""Development probe: fp16-operand 3x3 convolution on ADM layer shapes (B=4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops  # noqa: E402

SHAPES = [  # name, B, C0, C1, Cout, H, gn, res
    ("warm", 4, 256, 0, 256, 256, 1, 1),
    ("256_256_256_gn_res", 4, 256, 0, 256, 256, 1, 1),
    ("256_256_256_plain", 4, 256, 0, 256, 256, 0, 0),
    ("512cat_256_256_gn", 4, 256, 256, 256, 256, 1, 0),
    ("256_256_128_gn_res", 4, 256, 0, 256, 128, 1, 1),
    ("512_512_128_gn_res", 4, 512, 0, 512, 128, 1, 1),
    ("512_512_64_gn_res", 4, 512, 0, 512, 64, 1, 1),
    ("1024_512_64_gn", 4, 512, 512, 512, 64, 1, 0),
    ("512_512_32_gn_res", 4, 512, 0, 512, 32, 1, 1),
    ("1024_1024_32_gn_res", 4, 1024, 0, 1024, 32, 1, 1),
    ("1024_1024_16_gn_res", 4, 1024, 0, 1024, 16, 1, 1),
]
dev = "cuda"
for name, B, C0, C1, Cout, H, gn, res in SHAPES:
    a = torch.randn(B, H, H, C0, device=dev)
    b = torch.randn(B, H, H, C1, device=dev) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) * 0.02
    w32, w16 = ops.pack_conv_weight(w), ops.pack_conv_weight_f16(w)
    bias = torch.randn(Cout, device=dev)
    g = (torch.randn(B, C0 + C1, device=dev), torch.randn(B, C0 + C1, device=dev)) if gn else None
    r = torch.randn(B, H, H, Cout, device=dev) if res else None
    out = torch.empty(B, H, H, Cout, device=dev)
    flops = 2.0 * B * H * H * Cout * 9 * (C0 + C1)
    row = []
    for wf in (w16, None):
        for _ in range(2):
            ops.conv2d(a, w32, Cout, 3, src1=b, bias=bias, gn=g, res=r, out=out, weight_f16=wf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(a, w32, Cout, 3, src1=b, bias=bias, gn=g, res=r, out=out, weight_f16=wf)
        e1.record()
        torch.cuda.synchronize()
        row.append(flops / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    print(f"{name:24s} f16 {row[0]:7.1f}   f32 {row[1]:6.1f}  TFLOP/s", flush=True)
