#!/usr/bin/env python
"""Time ddnm_linear_f32 on the ADM FiLM projection shape (N = 51712 rows, K = 1024, B = 4 / 8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import ops  # noqa: E402

for B, K, N in [(4, 1024, 51712), (8, 1024, 51712), (8, 512, 11776), (4, 1024, 3000)]:
    x = torch.randn(B, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.02
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.linear(x, W, b, silu_in=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear(x, W, b, silu_in=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B} K={K} N={N}: {us:7.1f} us  {N * K * 4 / us / 1e6:6.2f} TB/s of weights")
