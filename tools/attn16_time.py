#!/usr/bin/env python
"""Time ddnm_attn16_d64 (ops.attn16) on the ADM attention shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops  # noqa: E402

print("B    T     C       us   TFLOP/s")
for B, T, C in [(4, 1024, 512), (4, 256, 1024), (4, 64, 1024), (8, 1024, 512), (8, 256, 1024), (8, 64, 1024), (1, 1024, 512)]:
    H = int(T ** 0.5)
    qkv = torch.randn(B, H, H, 3 * C, device="cuda").half()
    for _ in range(3):
        ops.attn16(qkv, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attn16(qkv, C)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{B:<4d} {T:<5d} {C:<5d} {us:8.1f} {4.0 * B * T * T * C / us / 1e6:8.1f}", flush=True)
