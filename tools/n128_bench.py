#!/usr/bin/env python
"""Launch-form timing of the Cout = 128 convolution (conv16_n128_kernel) and of the HBM-bound backward kernels at the
classifier's 256^2 / 128^2 shapes (development tool): B=${B:-32}."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import _lib, ops  # noqa: E402

B = int(os.environ.get("B", "32"))
dev = "cuda"


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Cin in ((256, 128), (128, 128), (256, 64)):
    x = torch.randn(B, H, H, Cin, device=dev).half()
    w = ops.pack_conv_weight16((torch.randn(128, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5))
    bias = torch.randn(128, device=dev)
    r = torch.randn(B, H, H, 128, device=dev).half()
    gn = (torch.randn(B * Cin, device=dev) * 0.5 + 1, torch.randn(B * Cin, device=dev) * 0.1)
    fl = 2.0 * B * H * H * 128 * 9 * Cin
    rows = [("plain, no stats", lambda: ops.conv16(x, w, 128, 3, bias=bias, emit_stats=False)),
            ("plain + stats", lambda: ops.conv16(x, w, 128, 3, bias=bias)),
            ("res + stats", lambda: ops.conv16(x, w, 128, 3, bias=bias, res=r)),
            ("gn fused + stats", lambda: ops.conv16(x, w, 128, 3, bias=bias, gn=gn)),
            ("gn fused + res + stats", lambda: ops.conv16(x, w, 128, 3, bias=bias, gn=gn, res=r)),
            ("gn pre-pass + conv + res + stats", lambda: ops.conv16(ops.gn_apply16(x, None, gn, True), w, 128, 3, bias=bias, res=r)),
            ("gn pre-pass alone", lambda: ops.gn_apply16(x, None, gn, True))]
    for name, fn in rows:
        us = timeit(fn)
        print(f"B={B} {Cin}->128 @{H}^2  {name:36s} {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s", flush=True)

# GroupNorm backward (fp16) at the 256^2 level
H, C = 256, 128
L = _lib.lib()
x = torch.randn(B, H, H, C, device=dev).half()
dA = torch.randn(B, H, H, C, device=dev).half()
add = torch.randn(B, H, H, C, device=dev).half()
ws = ops.GroupNormWorkspace(dev, B, C, B * ops.gn_nchunk(H * H, C) * 64)
keep = {}
ops.group_norm_affine(x, None, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-5, ws, keep=keep)
nchunk = L.ddnm_gn_bwd_nchunk(H * H, C)
partial = torch.empty(B * nchunk * 64, dtype=torch.float64, device=dev)
coef = torch.empty(B * 64, device=dev)
dx = torch.empty_like(x)


def gnb():
    _lib.check(L.ddnm_gn_bwd_h16(x.data_ptr(), dA.data_ptr(), 0, keep["scale"].data_ptr(), keep["shift"].data_ptr(),
                                 keep["mean_rstd"].data_ptr(), 1, add.data_ptr(), 0, B, H, H, C, 32, partial.data_ptr(), nchunk,
                                 coef.data_ptr(), dx.data_ptr(), ops._stream()), "gn_bwd_h16")


us = timeit(gnb)
el = B * H * H * C
print(f"gn_bwd_h16 (reduce + finalize + apply) B={B} {C}ch @{H}^2: {us:.1f} us  = {12.0 * el / us / 1e6:.2f} TB/s of 12 B/element")
wd = ops.pack_conv_weight16(torch.randn(3, 128, 3, 3, device=dev) * 0.03)
us = timeit(lambda: ops.conv16_out(x, wd, 3))
print(f"conv16_out 128->3 @256^2 B={B}: {us:.1f} us = {2.0 * el / us / 1e6:.2f} TB/s of the input read")
