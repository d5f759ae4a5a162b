#!/usr/bin/env python
"""Development probe: ablation builds of the fp16-operand conv kernel (WRONG results on purpose)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnm_amd._lib import ConvDesc  # noqa: E402

CSRC = os.path.join(ROOT, "ddnm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")      # git-ignored, but travels with gpurun (prebuilt here)
VARIANTS = {"base": [], "no_res_fold": ["-DDDNM_PROBE_NO_RES_FOLD"]}
BUILD_ONLY = "--build-only" in sys.argv
_args = [a for a in sys.argv[1:] if a != "--build-only"]
if _args:
    VARIANTS = {"base": [], **{f"v{i}": a.split() for i, a in enumerate(_args)}}
os.makedirs(OUT, exist_ok=True)
libs = {}
for n, fl in VARIANTS.items():
    so = os.path.join(OUT, f"libprobe16_{n}.so")
    if not os.path.exists(so) or BUILD_ONLY:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + fl +
                       [os.path.join(CSRC, "conv_igemm_f16.hip"), "-o", so], check=True)
    if BUILD_ONLY:
        continue
    lib = ctypes.CDLL(so)
    lib.ddnm_conv3x3_f16_f32.restype = ctypes.c_int32
    lib.ddnm_conv3x3_f16_f32.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_void_p]
    libs[n] = lib
if BUILD_ONLY:
    sys.exit(0)
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
wsb = torch.empty(64 << 20, device=dev)
for name, B, C, H, gn, res in [("warm", 4, 256, 256, 1, 1), ("256@256 gn res", 4, 256, 256, 1, 1), ("256@256 plain", 4, 256, 256, 0, 0),
                               ("256@256 res", 4, 256, 256, 0, 1), ("512@128 gn res", 4, 512, 128, 1, 1), ("512@128 res", 4, 512, 128, 0, 1),
                               ("512@64 gn res", 4, 512, 64, 1, 1), ("1024@32 res", 4, 1024, 32, 0, 1)]:
    a = torch.randn(B, H, H, C, device=dev)
    w = (torch.randn(C, 9, C, device=dev) * 0.02).half()
    bias = torch.randn(C, device=dev)
    sc, sh = torch.randn(B, C, device=dev), torch.randn(B, C, device=dev)
    r = torch.randn(B, H, H, C, device=dev)
    out = torch.empty(B, H, H, C, device=dev)
    d = ConvDesc()
    d.src0, d.weight, d.bias, d.out = a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.res = r.data_ptr() if res else None
    d.gn_scale, d.gn_shift = (sc.data_ptr(), sh.data_ptr()) if gn else (None, None)
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, H, H, C, 0, C
    d.ksize, d.stride, d.pad, d.Ho, d.Wo, d.gn_silu = 3, 1, 1, H, H, 1
    d.workspace, d.workspace_floats = wsb.data_ptr(), wsb.numel()
    flops = 2.0 * B * H * H * C * 9 * C
    row = []
    for n, lib in libs.items():
        for _ in range(2):
            assert lib.ddnm_conv3x3_f16_f32(ctypes.byref(d), stream) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.ddnm_conv3x3_f16_f32(ctypes.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        row.append(flops / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    print(f"{name:18s} " + " ".join(f"{k}={v:6.1f}" for k, v in zip(libs, row)), flush=True)
