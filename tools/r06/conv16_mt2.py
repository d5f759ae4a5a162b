#!/usr/bin/env python
"""Experiment (round 6): would the 128-pixel tile of conv16 (`<9,2,4>`: 64 accumulator registers per lane, room for a deferred
epilogue) carry the 256^2 / 128^2 layers of the ADM UNet as well as the 256-pixel tile (`<9,4,4>`)?  Builds a COPY of the
library whose plan picks MT = 2 for every 3x3 launch and times the layer shapes on both.    python tools/r06/conv16_mt2.py"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tools", "_build")


def build():
    src = os.path.join(ROOT, "ddnm_amd", "csrc")
    tmp = os.path.join(OUT, "mt2_src", "ddnm_amd", "csrc")
    shutil.rmtree(os.path.join(OUT, "mt2_src"), ignore_errors=True)
    os.makedirs(tmp)
    os.makedirs(os.path.join(OUT, "mt2_src", "include"))
    shutil.copy(os.path.join(ROOT, "include", "ddnm_hip.h"), os.path.join(OUT, "mt2_src", "include"))
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), tmp)
    p = os.path.join(tmp, "conv16.hip")
    s = open(p).read()
    old = "    int best_mt = tiles_of[4] > 0 ? 4 : (tiles_of[2] > 0 ? 2 : 0);"
    assert old in s
    s = s.replace(old, "    int best_mt = (tiles_of[4] > 0 && pl->taps != 9) ? 4 : (tiles_of[2] > 0 ? 2 : 0);")
    open(p, "w").write(s)
    so = os.path.join(OUT, "libddnm_mt2.so")
    from ddnm_amd import build as B
    srcs = [os.path.join(tmp, f) for f in B.SOURCES]
    subprocess.run(["hipcc"] + [f for f in B.FLAGS] + srcs + ["-o", so], check=True)
    return so


def main():
    so = build()
    if "--build-only" in sys.argv:
        return
    import torch
    from ddnm_amd import _lib, ops
    import conv16_bench as cb
    base = _lib.lib()
    alt = ctypes.CDLL(so)
    for name, (restype, argtypes) in _lib.PROTOTYPES.items():
        fn = getattr(alt, name)
        fn.restype, fn.argtypes = restype, argtypes
    dev = "cuda"
    B = 4
    print(f"{'shape (B = 4)':26s} {'<9,4,4> us':>11s} {'TF/s':>8s} | {'<9,2,4> us':>11s} {'TF/s':>8s}")
    for name, cin, cout, H, k, res, ups, _ in cb.SHAPES:
        if k != 3 or H < 64:
            continue
        Hs = H // 2 if ups else H
        x16 = torch.randn(B, Hs, Hs, cin, device=dev).half()
        w = torch.randn(cout, cin, k, k, device=dev) * (k * k * cin) ** -0.5
        w16 = ops.pack_conv_weight16(w)
        bias = torch.randn(cout, device=dev)
        r16 = torch.randn(B, H, H, cout, device=dev).half() if res else None
        sc, sh = torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.2
        flops = 2.0 * B * H * H * cout * k * k * cin
        ts, outs = [], []
        for L in (base, alt):
            _lib._lib = L
            f = lambda: ops.conv16(x16, w16, cout, k, bias=bias, res=r16, ups=ups, gn=(sc, sh))      # noqa: E731
            outs.append(f().t.clone())
            ts.append(cb.timeit(f))
        _lib._lib = base
        same = torch.equal(outs[0], outs[1])
        print(f"{name:26s} {ts[0] * 1e3:11.1f} {flops / ts[0] / 1e9:8.1f} | {ts[1] * 1e3:11.1f} {flops / ts[1] / 1e9:8.1f}   {'==' if same else 'differs'}", flush=True)


if __name__ == "__main__":
    main()
