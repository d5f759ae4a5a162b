#!/usr/bin/env python
"""Probe harness (development tool, not product code): times the convolution kernel on the layer
shapes that dominate the celeba UNet, for several builds of the library (ablation builds are
compiled with -DDDNM_PROBE_* and produce WRONG results on purpose).

    python tools/conv_probe.py            # on the GPU box; builds variants with hipcc first
"""
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

VARIANTS = {
    "base": [],
    "old": None,                                 # a library built beforehand from another source (tools/_build/libprobe_old.so)
    "prio_mfma": ["-DDDNM_PROBE_SETPRIO_MFMA"],
    "prio_half": ["-DDDNM_PROBE_SETPRIO_HALF"],
    "nostore": ["-DDDNM_PROBE_NO_STORE"],
    "dephase1": ["-DDDNM_PROBE_DEPHASE=1"],      # blocks 256..511 skip half of their first tile's K (3.1 % less work at 4096 tiles)
    "dephase2": ["-DDDNM_PROBE_DEPHASE=2"],      # every other group of 8 blocks among the first 512
    "nogn": ["-DDDNM_PROBE_NO_GN"],
    "nopin": ["-DDDNM_PROBE_NO_PIN"],
    "nofrag": ["-DDDNM_PROBE_NO_FRAG"], "nosync": ["-DDDNM_PROBE_NO_SYNC"], "nowl": ["-DDDNM_PROBE_NO_WL"],
    "all3": ["-DDDNM_PROBE_NO_FRAG", "-DDDNM_PROBE_NO_SYNC", "-DDDNM_PROBE_NO_WL"],
    "all3ns": ["-DDDNM_PROBE_NO_FRAG", "-DDDNM_PROBE_NO_SYNC", "-DDDNM_PROBE_NO_WL", "-DDDNM_PROBE_NO_STORE", "-DDDNM_PROBE_NO_GN"],
    "occ1": ["-DDDNM_PROBE_LDS_PAD=8192"],            # round-2 schedule: the compiler sinks the weight prefetch to the end of a tap
}
if os.environ.get("ONLY"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["ONLY"].split(",")}

# (name, B, C0, C1, Cout, H (input, pre-upsample), k, stride, ups, gn, res, tile)
SHAPES = [
    ("warmup", 8, 128, 0, 128, 256, 3, 1, 0, 1, 1, 0),
    ("c128_128_256_gn_res", 8, 128, 0, 128, 256, 3, 1, 0, 1, 1, 0),
    ("c128_128_256_plain", 8, 128, 0, 128, 256, 3, 1, 0, 0, 0, 0),
    ("c128_128_256_gn_only", 8, 128, 0, 128, 256, 3, 1, 0, 1, 0, 0),
    ("c128_128_256_res_only", 8, 128, 0, 128, 256, 3, 1, 0, 0, 1, 0),
    ("c256cat_128_256_gn", 8, 128, 128, 128, 256, 3, 1, 0, 1, 0, 0),
    ("c256cat_128_256_1x1", 8, 128, 128, 128, 256, 1, 1, 0, 0, 0, 0),
    ("c128_128_128_gn_res", 8, 128, 0, 128, 128, 3, 1, 0, 1, 1, 0),
    ("c256_256_64_gn_res", 8, 256, 0, 256, 64, 3, 1, 0, 1, 1, 0),
    ("c256_256_32_gn_res", 8, 256, 0, 256, 32, 3, 1, 0, 1, 1, 0),
    ("c512_512_16_gn_res", 8, 512, 0, 512, 16, 3, 1, 0, 1, 1, 0),
    ("c512_512_8_gn_res", 8, 512, 0, 512, 8, 3, 1, 0, 1, 1, 0),
    ("c256_256_32_plain", 8, 256, 0, 256, 32, 3, 1, 0, 0, 0, 0),
    ("c768_256_32_gn", 8, 512, 256, 256, 32, 3, 1, 0, 1, 0, 0),
    ("c1024_512_16_gn", 8, 512, 512, 512, 16, 3, 1, 0, 1, 0, 0),
    ("c1024_512_8_gn", 8, 512, 512, 512, 8, 3, 1, 0, 1, 0, 0),
    ("c128_3_256_out", 8, 128, 0, 3, 256, 3, 1, 0, 1, 0, 0),
    ("c128_128_up256", 8, 128, 0, 128, 128, 3, 1, 1, 0, 0, 0),
    ("c128_128_down", 8, 128, 0, 128, 256, 3, 2, 0, 0, 0, 0),
]


def build(name, flags):
    so = os.path.join(OUT, f"libprobe_{name}.so")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + (flags or []) + \
          [os.path.join(CSRC, "conv_igemm_f32.hip"), "-o", so]
    if flags is None:
        assert os.path.exists(so), so
    elif not os.path.exists(so) or "--rebuild" in sys.argv:
        subprocess.run(cmd, check=True)
    if "--build-only" in sys.argv:
        return None
    lib = ctypes.CDLL(so)
    lib.ddnm_conv2d_f32.restype = ctypes.c_int32
    lib.ddnm_conv2d_f32.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_void_p]
    return lib


def main():
    os.makedirs(OUT, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
    global SHAPES
    if os.environ.get("SHAPES") == "big":
        SHAPES = [s for s in SHAPES if s[0] in ("warmup", "c128_128_256_gn_res", "c128_128_256_plain", "c128_128_256_gn_only",
                                               "c128_128_256_res_only", "c256cat_128_256_gn", "c128_128_128_gn_res",
                                               "c256_256_64_gn_res")]
    libs = {n: build(n, VARIANTS[n]) for n in only}
    if "--build-only" in sys.argv:
        return
    dev = "cuda"
    ws = torch.empty(64 << 20, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':26s} " + " ".join(f"{n:>10s}" for n in libs) + "   (TFLOP/s)")
    for (name, B, C0, C1, Cout, H, k, stride, ups, gn, res, tile) in SHAPES:
        Hin = 2 * H if ups else H
        Ho = Hin // stride
        zero = os.environ.get("ZERO") == "1"         # zero-filled operands: DVFS / power check (MI355X_MICROARCH.md)
        a = torch.randn(B, H, H, C0, device=dev)
        b = torch.randn(B, H, H, C1, device=dev) if C1 else None
        cpad = (Cout + 127) // 128 * 128
        w = torch.randn(cpad, k * k, C0 + C1, device=dev) * 0.02
        # COLD=1: rotate through > 512 MB of weight copies so that no launch finds its weights in L2 / MALL (as inside a forward)
        ncopy = min(64, int(512e6 // (w.numel() * 4)) + 1) if os.environ.get("COLD") == "1" else 1
        wcopies = [w] + [w.clone() for _ in range(ncopy - 1)]
        if zero:
            a.zero_()
            for wc in wcopies:
                wc.zero_()
            if b is not None:
                b.zero_()
        bias = torch.randn(Cout, device=dev)
        sc = torch.randn(B, C0 + C1, device=dev)
        sh = torch.randn(B, C0 + C1, device=dev)
        r = torch.randn(B, Ho, Ho, Cout, device=dev) if res else None
        out = torch.empty(B, Ho, Ho, Cout, device=dev)
        d = ConvDesc()
        d.src0, d.src1, d.weight, d.bias = a.data_ptr(), (b.data_ptr() if C1 else None), w.data_ptr(), bias.data_ptr()
        d.res = r.data_ptr() if res else None
        d.gn_scale, d.gn_shift = (sc.data_ptr(), sh.data_ptr()) if gn else (None, None)
        d.out = out.data_ptr()
        d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, Hin, Hin, C0, C1, Cout
        d.ksize, d.stride, d.pad, d.Ho, d.Wo = k, stride, (0 if stride == 2 else k // 2), Ho, Ho
        d.ups, d.gn_silu, d.out_nchw, d.badd_stride, d.tile = ups, 1, 0, 0, (int(os.environ.get('TILE', '0')) if (k == 3 and stride == 1 and Cout % 128 == 0) else tile)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
        flops = 2.0 * B * Ho * Ho * Cout * k * k * (C0 + C1)
        row = []
        for n, lib in libs.items():
            for _ in range(2):
                rc = lib.ddnm_conv2d_f32(ctypes.byref(d), stream)
                assert rc == 0, rc
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for it in range(reps):
                d.weight = wcopies[it % len(wcopies)].data_ptr()
                lib.ddnm_conv2d_f32(ctypes.byref(d), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row.append(flops / (ms * 1e-3) / 1e12)
        print(f"{name:26s} " + " ".join(f"{v:10.1f}" for v in row) + "   us: " +
              " ".join(f"{flops / (v * 1e12) * 1e6:8.1f}" for v in row), flush=True)


if __name__ == "__main__":
    main()
