#!/usr/bin/env python
"""Development probe: TFLOP/s of ddnm_conv16 on the ADM layer shapes (B = 4), next to the first-generation fp16 kernel."""
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnm_amd import ops  # noqa: E402

B = int(os.environ.get("B", "4"))
SHAPES = [  # name, Cin, Cout, H, k, res, ups, skip
    ("256->256 @256", 256, 256, 256, 3, True, False, 0),
    ("512->256 @256 +skip", 512, 256, 256, 3, False, False, 0),
    ("256->256 @128", 256, 256, 128, 3, True, False, 0),
    ("512->512 @128", 512, 512, 128, 3, True, False, 0),
    ("512->512 @64", 512, 512, 64, 3, True, False, 0),
    ("1024->512 @64", 1024, 512, 64, 3, False, False, 0),
    ("512->512 @32", 512, 512, 32, 3, True, False, 0),
    ("1024->1024 @32", 1024, 1024, 32, 3, True, False, 0),
    ("1024->1024 @16", 1024, 1024, 16, 3, True, False, 0),
    ("2048->1024 @16", 2048, 1024, 16, 3, False, False, 0),
    ("256->256 up @256", 256, 256, 256, 3, False, True, 0),
    ("qkv 512->1536 @32", 512, 1536, 32, 1, False, False, 0),
    ("proj 1024->1024 @16", 1024, 1024, 16, 1, True, False, 0),
    ("skip1x1 512->256 @256", 512, 256, 256, 1, False, False, 0),
]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    print(f"{'shape':26s} {'conv16 us':>10s} {'TF/s':>8s} | {'gen1 us':>9s} {'TF/s':>8s}")
    for name, cin, cout, H, k, res, ups, _ in SHAPES:
        Hs = H // 2 if ups else H
        x16 = torch.randn(B, Hs, Hs, cin, device=dev).half()
        w = torch.randn(cout, cin, k, k, device=dev) * (k * k * cin) ** -0.5
        w16 = ops.pack_conv_weight16(w)
        bias = torch.randn(cout, device=dev)
        r16 = torch.randn(B, H, H, cout, device=dev).half() if res else None
        flops = 2.0 * B * H * H * cout * k * k * cin
        t2 = timeit(lambda: ops.conv16(x16, w16, cout, k, bias=bias, res=r16, ups=ups))
        # first generation: fp32 activations, fp16 operands (GroupNorm pre-pass output as the operand)
        t1 = float("nan")
        try:
            x32 = x16.float()
            wp32, wp16 = ops.pack_conv_weight(w), ops.pack_conv_weight_f16(w)
            r32 = r16.float() if res else None
            t1 = timeit(lambda: ops.conv2d(x32, wp32, cout, k, bias=bias, res=r32, ups=ups, emit_stats=True, weight_f16=wp16))
        except Exception as e:      # noqa: BLE001
            print("   gen1 failed:", e)
        print(f"{name:26s} {t2 * 1e3:10.1f} {flops / t2 / 1e9:8.1f} | {t1 * 1e3:9.1f} {flops / t1 / 1e9:8.1f}", flush=True)


if __name__ == "__main__":
    main()
