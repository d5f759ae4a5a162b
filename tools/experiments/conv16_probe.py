#!/usr/bin/env python
"""Development probe: several builds of conv16.hip side by side in ONE session (box-to-box variation is 2-4 %), over the
ADM layer shapes.  Ablation builds (-DDDNM_P16_NO_*) give WRONG results on purpose and separate the per-launch fixed cost
(dispatch, prologue, epilogue) from the main-loop rate; "old" is a library built beforehand from another source
(tools/_build/libp16_old.so).  Environment: ONLY=a,b (variants), SHAPES=low|one|mid (shape lists), GN=1 (fused GroupNorm),
COLD=1 (rotate through > 512 MB of weight copies: weights come from HBM as inside a forward -- without it the low-resolution
layers look 15-20 % faster than they are and a deeper weight pipeline shows no gain)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnm_amd._lib import Conv16Desc  # noqa: E402

CSRC = os.path.join(ROOT, "ddnm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")
VARIANTS = {"base": [], "no_epi": ["-DDDNM_P16_NO_EPI"], "no_main": ["-DDDNM_P16_NO_MAIN"],
            "ne_nowl": ["-DDDNM_P16_NO_EPI", "-DDDNM_P16_NO_WLOAD"], "ne_nosync": ["-DDDNM_P16_NO_EPI", "-DDDNM_P16_NO_SYNC"],
            "ne_nofrag": ["-DDDNM_P16_NO_EPI", "-DDDNM_P16_NO_FRAG"],
            "ne_all3": ["-DDDNM_P16_NO_EPI", "-DDDNM_P16_NO_FRAG", "-DDDNM_P16_NO_SYNC", "-DDDNM_P16_NO_WLOAD"],
            "burst": ["-DDDNM_P16_ILV=0"], "early": ["-DDDNM_P16_LATE_DMA=0"], "late_res": ["-DDDNM_P16_EARLY_RES=0"], "old": None, "maxm0": ["-DDDNM_P16_WMAJOR_MAXM=0"], "maxm16": ["-DDDNM_P16_WMAJOR_MAXM=16"],
            "maxm64": ["-DDDNM_P16_WMAJOR_MAXM=64"],
            "mt1off": ["-DDDNM_P16_MT1=0"], "nowl": ["-DDDNM_P16_NO_WLOAD"], "nosync": ["-DDDNM_P16_NO_SYNC"], "nofrag": ["-DDDNM_P16_NO_FRAG"], "ks1": ["-DDDNM_P16_KSCAP=1"], "ks2": ["-DDDNM_P16_KSCAP=2"], "ks4": ["-DDDNM_P16_KSCAP=4"]}
if os.environ.get("ONLY"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["ONLY"].split(",")}
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
for i, a in enumerate(extra):
    VARIANTS[f"x{i}"] = a.split(",")
BUILD_ONLY = "--build-only" in sys.argv
os.makedirs(OUT, exist_ok=True)
libs = {}
for n, fl in VARIANTS.items():
    so = os.path.join(OUT, f"libp16_{n}.so")
    if fl is None:          # a previously built library of another source (A/B inside one session)
        assert os.path.exists(so), so
    elif not os.path.exists(so) or BUILD_ONLY:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + fl +
                       [os.path.join(CSRC, "conv16.hip"), "-o", so], check=True)
    if BUILD_ONLY:
        continue
    lib = ctypes.CDLL(so)
    lib.ddnm_conv16.restype = ctypes.c_int32
    lib.ddnm_conv16.argtypes = [ctypes.POINTER(Conv16Desc), ctypes.c_void_p]
    libs[n] = lib
if BUILD_ONLY:
    sys.exit(0)
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
B = 4
flush = torch.empty(1 << 28, device=dev) if os.environ.get("COLD") == "1" else None      # 1 GiB
print("shape                 " + " ".join(f"{n:>10s}" for n in libs) + "   (us)")
LOW = [("warm", 256, 256, 256, 3, 1), ("512->512@64 res", 512, 512, 64, 3, 1), ("512->1024@32", 512, 1024, 32, 3, 0),
       ("1024->1024@32 res", 1024, 1024, 32, 3, 1), ("1024->1024@16 res", 1024, 1024, 16, 3, 1),
       ("2048->1024@16", 2048, 1024, 16, 3, 0), ("1536->1024@32", 1536, 1024, 32, 3, 0)]
MID = [("warm", 256, 256, 256, 3, 1), ("256->256@128 res", 256, 256, 128, 3, 1), ("256->256@128", 256, 256, 128, 3, 0),
       ("512->256@128", 512, 256, 128, 3, 0), ("768->256@128", 768, 256, 128, 3, 0), ("256->512@128", 256, 512, 128, 3, 0)]
ONE = [("warm", 256, 256, 256, 3, 1),
       ("9216->1024@8 1x1", 9216, 1024, 8, 1, 0), ("1024->3072@32 1x1", 1024, 3072, 32, 1, 0),
       ("1024->1024@16 1x1", 1024, 1024, 16, 1, 0), ("1024->3072@16 1x1", 1024, 3072, 16, 1, 0),
       ("1024->1024@8 1x1", 1024, 1024, 8, 1, 0), ("1024->3072@8 1x1", 1024, 3072, 8, 1, 0),
       ("512->1536@32 1x1", 512, 1536, 32, 1, 0), ("512->512@32 1x1", 512, 512, 32, 1, 0),
       ("18432->1024@8 1x1", 18432, 1024, 8, 1, 0), ("4608->512@8 1x1", 4608, 512, 8, 1, 0)]
for name, Cin, Cout, H, k, res in {"low": LOW, "one": ONE, "mid": MID}.get(os.environ.get("SHAPES"), None) or [("warm", 256, 256, 256, 3, 1), ("64->256@256 res", 64, 256, 256, 3, 1), ("128->256@256 res", 128, 256, 256, 3, 1),
                                   ("256->256@256 res", 256, 256, 256, 3, 1), ("256->256@256", 256, 256, 256, 3, 0),
                                   ("512->256@256", 512, 256, 256, 3, 0), ("1024->256@256", 1024, 256, 256, 3, 0),
                                   ("256->256@128 res", 256, 256, 128, 3, 1), ("512->512@128 res", 512, 512, 128, 3, 1),
                                   ("512->512@64 res", 512, 512, 64, 3, 1), ("256->256@256 1x1", 256, 256, 256, 1, 0),
                                   ("1024->256@256 1x1", 1024, 256, 256, 1, 0)]:
    x = torch.randn(B, H, H, Cin, device=dev).half()
    w = (torch.randn(Cout, k * k, Cin, device=dev) * 0.02).half()
    # COLD=1: rotate through copies of the weights (> 512 MB together) so that no launch finds them in L2 / MALL,
    # like a layer inside a forward pass
    ncopy = max(1, int(512e6 // (w.numel() * 2)) + 1) if os.environ.get("COLD") == "1" else 1
    wcopies = [w] + [w.clone() for _ in range(min(ncopy, 64) - 1)]
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(B, H, H, Cout, device=dev).half()
    out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.float16)
    stats = torch.empty(B * max(H * H // 128, 64) * Cout * 2, device=dev)
    d = Conv16Desc()
    d.src, d.weight, d.bias, d.out, d.stats_out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), stats.data_ptr()
    d.res = r.data_ptr() if res else None
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize = B, H, H, Cin, Cout, k
    if os.environ.get("GN") == "1" and k == 3:
        gsc, gsh = torch.randn(B, Cin, device=dev), torch.randn(B, Cin, device=dev)
        d.gn_scale, d.gn_shift, d.gn_silu = gsc.data_ptr(), gsh.data_ptr(), 1
    row = []
    for n, lib in libs.items():
        lib.ddnm_conv16_workspace_floats.restype = ctypes.c_int64
        lib.ddnm_conv16_workspace_floats.argtypes = [ctypes.POINTER(Conv16Desc)]
        lib.ddnm_conv16_stats_tiles.restype = ctypes.c_int32
        lib.ddnm_conv16_stats_tiles.argtypes = [ctypes.POINTER(Conv16Desc)]
        d.stats_out = stats.data_ptr() if lib.ddnm_conv16_stats_tiles(ctypes.byref(d)) > 0 else None
        need = lib.ddnm_conv16_workspace_floats(ctypes.byref(d))
        if need > 0:
            ws = torch.empty(need, device=dev)
            d.workspace, d.workspace_floats = ws.data_ptr(), need
        else:
            d.workspace, d.workspace_floats = None, 0
        for _ in range(2):
            assert lib.ddnm_conv16(ctypes.byref(d), stream) == 0
        if len(wcopies) > 1:
            flush.zero_()               # evict the freshly cloned weights from L2 / MALL
        torch.cuda.synchronize()
        if os.environ.get("PREF") == "1":
            # weight-prefetch experiment: an unrelated kernel READS the (cold) weight copy right before the convolution
            # that uses it (HBM -> MALL / some L2s); only the convolutions are timed
            tot = 0.0
            for it in range(10):
                wc = wcopies[it % len(wcopies)]
                d.weight = wc.data_ptr()
                _ = wc.view(torch.int16).sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.ddnm_conv16(ctypes.byref(d), stream)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            row.append(tot / 10 * 1e3)
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(10):
            d.weight = wcopies[it % len(wcopies)].data_ptr()
            lib.ddnm_conv16(ctypes.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * B * H * H * Cout * k * k * Cin
    print(f"{name:22s} " + " ".join(f"{v:10.1f}" for v in row) + f"   base {fl / row[0] / 1e6:7.1f} TF/s", flush=True)
