#!/usr/bin/env python
"""A/B of the persistent split-fp16 3x3 kernel against the one-tile kernel (round 6): bit-identity of output and
GroupNorm partials, and launch time, per layer shape.   python tools/r06/persist_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import s16_probe as sp  # noqa: E402
from ddnm_amd import ops  # noqa: E402

# (name, B, C0, C1, Cout, H (input, pre-upsample), ups, gn, res, skip)
SHAPES = [s for s in sp.SHAPES if s[0] in ("c128_128_256_gn_res", "c128_128_256_plain", "c256cat_128_256_gn", "c128_128_128_gn_res", "c128_128_up256")] + [
    ("c256_256_128_plain", 8, 256, 0, 256, 128, 0, 0, 0, 0),       # two channel tiles per pixel tile
    ("c128_256_up128", 8, 128, 0, 256, 64, 1, 0, 0, 0),
    ("c128_128_256_b3", 3, 128, 0, 128, 256, 0, 1, 1, 0),           # 768 tiles: 3 per workgroup
    ("c64_128_256_b2", 2, 64, 0, 128, 256, 0, 1, 1, 0),             # two chunks: FIRST directly followed by LAST
    ("c96cat_128_256_b2", 2, 64, 32, 128, 256, 0, 1, 0, 0),
    ("c128_128_256_gn_skip", 8, 128, 0, 128, 256, 0, 1, 0, 1),      # fused 1x1 shortcut (64 raw channels)
    ("c128_256_128_gn_skip_b4", 4, 128, 0, 256, 128, 0, 1, 0, 1),    # two channel tiles
    ("c128_128_256_b1", 1, 128, 0, 128, 256, 0, 1, 1, 0),           # 256 tiles: not eligible, both paths the same kernel
]


def call(t, one_tile):
    ws = [t["w"]] + ([t["wsk"]] if t["wsk"] is not None else [])
    scale = ops.s16_weight_scale(*ws)
    s16 = (ops.pack_conv_weight_s16(t["w"], scale), scale, ops.pack_conv_weight_s16(t["wsk"], scale) if t["wsk"] is not None else None)
    w32 = ops.pack_conv_weight(t["w"])
    wsk32 = ops.pack_skip_weight(t["wsk"]) if t["wsk"] is not None else None
    gn = None if t["sc"] is None else (t["sc"], t["sh"])
    badd = t["badd"]

    def f():
        return ops.conv2d(t["a"], w32, t["Cout"], 3, src1=t["b"], bias=t["bias"], badd=badd, badd_stride=badd.shape[1], res=t["r"], gn=gn,
                          gn_silu=True, ups=bool(t["ups"]), emit_stats=True, weight_s16=s16, one_tile=one_tile, raw_amax=t["amax"],
                          skip=None if t["sk"] is None else (t["sk"], None), skip_weight=wsk32)
    return f


def main():
    bad = 0
    for s in SHAPES:
        name, B, C0, C1, Cout = s[:5]
        t = sp.make(*s)
        t["badd"] = torch.randn(B, Cout, device="cuda")
        t["amax"] = ops.amax_bound(t["sk"]) if s[9] else (None if s[7] else ops.amax_bound(t["a"], t["b"]))
        Ho = t["Ho"]
        flops = 2.0 * B * Ho * Ho * Cout * (9 * (C0 + C1) + (64 if s[9] else 0))
        res, us = {}, {}
        for one in (True, False):
            f = call(t, one)
            a = f()
            torch.cuda.synchronize()
            res[one] = (a.t.clone(), a.stats.clone(), a.tiles)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            us[one] = e0.elapsed_time(e1) * 50
        same_o = torch.equal(res[True][0], res[False][0])
        same_s = torch.equal(res[True][1], res[False][1]) and res[True][2] == res[False][2]
        d = (res[True][0] - res[False][0]).abs().max().item()
        nan = not torch.isfinite(res[False][0]).all().item()
        bad += (not same_o) or (not same_s)
        print(f"{name:24s} one-tile {us[True]:8.1f} us  persistent {us[False]:8.1f} us  ({us[True] / us[False]:.3f}x, {flops / us[False] / 1e6:6.1f} TF)  "
              f"out {'==' if same_o else f'DIFFERS max {d:.3e}'}  stats {'==' if same_s else 'DIFFER'}{'  NON-FINITE' if nan else ''}", flush=True)
        del t, res
        torch.cuda.empty_cache()
    print("FAILED" if bad else "all bit-identical")


if __name__ == "__main__":
    main()
