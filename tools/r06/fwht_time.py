#!/usr/bin/env python
"""Column-FWHT kernels: result against a torch butterfly reference and launch time (development tool, round 6)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ddnm_amd import _lib, ops  # noqa: E402


def fwht_ref(x):                     # natural-order, unnormalised, over the last axis (fp64)
    n = x.shape[-1]
    x = x.double().clone()
    h = 1
    while h < n:
        x = x.view(*x.shape[:-1], n // (2 * h), 2, h)
        a, b = x[..., 0, :], x[..., 1, :]
        x = torch.stack([a + b, a - b], -2).reshape(*x.shape[:-3], n)
        h *= 2
    return x


def main():
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n, planes in ((256, 24), (128, 24), (64, 12), (32, 6)):
        g = torch.Generator(device="cuda").manual_seed(n)
        x = torch.randn(planes, n, n, device="cuda", generator=g)
        mask = (torch.rand(3, n, n, device="cuda", generator=g) < 0.25).float()
        out, scratch = torch.empty_like(x), torch.empty_like(x)
        assert L.ddnm_fwht2d_f32(x.data_ptr(), out.data_ptr(), planes, n, st) == 0
        want = fwht_ref(fwht_ref(x).transpose(1, 2)).transpose(1, 2) / n
        e1 = ((out.double() - want).norm() / want.norm()).item()
        assert L.ddnm_fwht2d_masked_f32(x.data_ptr(), mask.data_ptr(), 3, out.data_ptr(), planes, n, scratch.data_ptr(), st) == 0
        m = mask.repeat(planes // 3, 1, 1).double()
        w2 = fwht_ref(fwht_ref(want * m).transpose(1, 2)).transpose(1, 2) / n
        e2 = ((out.double() - w2).norm() / w2.norm()).item()
        torch.cuda.synchronize()
        ts = []
        for fn in (lambda: L.ddnm_fwht2d_f32(x.data_ptr(), out.data_ptr(), planes, n, st),
                   lambda: L.ddnm_fwht2d_masked_f32(x.data_ptr(), mask.data_ptr(), 3, out.data_ptr(), planes, n, scratch.data_ptr(), st)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 20)
        mb = planes * n * n * 4 / 1e6
        print(f"n={n:3d} planes={planes:2d} ({mb:5.1f} MB): fwht2d err {e1:.1e} {ts[0]:6.1f} us (rows + cols)   masked err {e2:.1e} {ts[1]:6.1f} us (rows + cols-mask-cols + rows)")


if __name__ == "__main__":
    main()
