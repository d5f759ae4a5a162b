#!/usr/bin/env python
"""Probe harness (development tool, not product code) for the split-fp16 3x3 kernel (ddnm_conv3x3_s16_f32):

  * does the fp16 MFMA honour subnormal inputs (tools/experiments/mfma_denorm.hip),
  * accuracy of the split kernel AND of the fp32-MFMA kernel against an fp64 evaluation of the same layer
    (GroupNorm affine + swish prologue, concat, x2 upsample, fused 1x1 shortcut, bias, residual),
  * speed of both on the layer shapes of the celeba UNet.

    python tools/s16_probe.py [acc] [time]        # on the GPU box, product library

(The build-variant ablations of rounds 3-5 -- `vtime`, -DDDNM_PROBE_* -- left with the probe sites they compiled: the product
sources carry no wrong-result switches since round 6; `git show a920112:tools/s16_probe.py` has them.)
"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnm_amd import ops  # noqa: E402

dev = "cuda"

# (name, B, C0, C1, Cout, H (input, pre-upsample), ups, gn, res, skip)
SHAPES = [
    ("c128_128_256_gn_res", 8, 128, 0, 128, 256, 0, 1, 1, 0),
    ("c128_128_256_plain", 8, 128, 0, 128, 256, 0, 0, 0, 0),
    ("c256cat_128_256_gn", 8, 128, 128, 128, 256, 0, 1, 0, 0),
    ("c128_128_256_gn_skip", 8, 128, 0, 128, 256, 0, 1, 0, 1),
    ("c128_128_128_gn_res", 8, 128, 0, 128, 128, 0, 1, 1, 0),
    ("c128_256_64_gn", 8, 128, 0, 256, 64, 0, 1, 0, 0),
    ("c256_256_64_gn_res", 8, 256, 0, 256, 64, 0, 1, 1, 0),
    ("c256_256_32_gn_res", 8, 256, 0, 256, 32, 0, 1, 1, 0),
    ("c512_512_16_gn_res", 8, 512, 0, 512, 16, 0, 1, 1, 0),
    ("c768_256_32_gn_skip", 8, 512, 256, 256, 32, 0, 1, 0, 1),
    ("c1024_512_16_gn", 8, 512, 512, 512, 16, 0, 1, 0, 0),
    ("c512_256_32_gn", 8, 256, 256, 256, 32, 0, 1, 0, 0),
    ("c256_512_16_gn", 8, 256, 0, 512, 16, 0, 1, 0, 0),
    ("c512_512_16_b4", 4, 512, 0, 512, 16, 0, 1, 1, 0),
    ("c128_128_up256", 8, 128, 0, 128, 128, 1, 0, 0, 0),
]


def denorm():
    so = os.path.join(ROOT, "tools", "_build", "libmfma_denorm.so")
    if not os.path.exists(so):
        print("denorm probe: library not built")
        return
    lib = ctypes.CDLL(so)
    lib.mfma_denorm_probe.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
    out = torch.zeros(2, device=dev)
    for e in (-15, -20, -24):
        lib.mfma_denorm_probe(out.data_ptr(), ctypes.c_float(2.0 ** e), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o = out.tolist()
        print(f"denorm probe 2^{e}: mfma sum / (16 * 2^{e}) = {o[0] / (16 * 2.0 ** e):.3f}   cvt round trip / 2^{e} = "
              f"{o[1] / 2.0 ** e:.3f}")


def make(name, B, C0, C1, Cout, H, ups, gn, res, skip, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    Ho = 2 * H if ups else H
    a = rn(B, H, H, C0) * 1.5
    b = rn(B, H, H, C1) * 1.5 if C1 else None
    w = rn(Cout, C0 + C1, 3, 3) * (1.0 / (3.0 * (C0 + C1) ** 0.5))
    bias = rn(Cout)
    sc, sh = (rn(B, C0 + C1) * 0.3 + 1.0, rn(B, C0 + C1) * 0.3) if gn else (None, None)
    r = rn(B, Ho, Ho, Cout) * 2.0 if res else None
    sk = rn(B, H, H, 64) * 2.0 if skip else None
    wsk = rn(Cout, 64, 1, 1) * 0.1 if skip else None
    return dict(a=a, b=b, w=w, bias=bias, sc=sc, sh=sh, r=r, sk=sk, wsk=wsk, Ho=Ho, ups=ups, Cout=Cout)


def ref64(t):
    x = t["a"] if t["b"] is None else torch.cat([t["a"], t["b"]], 3)
    x = x.double()
    if t["sc"] is not None:
        x = x * t["sc"].double()[:, None, None, :] + t["sh"].double()[:, None, None, :]
        x = x * torch.sigmoid(x)
    x = x.permute(0, 3, 1, 2)
    if t["ups"]:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, t["w"].double(), t["bias"].double(), padding=1)
    if t["sk"] is not None:
        y = y + F.conv2d(t["sk"].double().permute(0, 3, 1, 2), t["wsk"].double())
    y = y.permute(0, 2, 3, 1)
    if t["r"] is not None:
        y = y + t["r"].double()
    return y


def run(t, split):
    w32 = ops.pack_conv_weight(t["w"])
    wsk32 = ops.pack_skip_weight(t["wsk"]) if t["wsk"] is not None else None
    s16 = None
    if split:
        ws = [t["w"]] + ([t["wsk"]] if t["wsk"] is not None else [])
        scale = ops.s16_weight_scale(*ws)
        s16 = (ops.pack_conv_weight_s16(t["w"], scale), scale,
               ops.pack_conv_weight_s16(t["wsk"], scale) if t["wsk"] is not None else None)
    gn = None if t["sc"] is None else (t["sc"], t["sh"])

    def call():
        return ops.conv2d(t["a"], w32, t["Cout"], 3, src1=t["b"], bias=t["bias"], res=t["r"], gn=gn, gn_silu=True,
                          ups=bool(t["ups"]), emit_stats=True, weight_s16=s16,
                          skip=None if t["sk"] is None else (t["sk"], None), skip_weight=wsk32)
    return call


def main():
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["denorm", "acc", "time"]
    if "denorm" in what:
        denorm()
    print("activation pre-scale of this build:", ops._s16_act_scale())
    shapes = SHAPES
    if os.environ.get("SHAPES"):
        shapes = [s for s in SHAPES if s[0] in os.environ["SHAPES"].split(",")]
    for s in shapes:
        name = s[0]
        t = make(*s)
        line = f"{name:24s}"
        if "acc" in what:
            bsmall = 2
            ts = {k: (v[:bsmall] if torch.is_tensor(v) and v.dim() >= 2 and k not in ("w", "wsk") else v) for k, v in t.items()}
            y = ref64(ts)
            for split in (False, True):
                act = run(ts, split)()
                o = act.t.double()
                rel = ((o - y).norm() / y.norm()).item()
                mx = ((o - y).abs().max() / y.abs().max()).item()
                # the emitted GroupNorm partials against the tensor they describe
                st = act.stats.view(bsmall, act.tiles, -1, 2).double().sum(1) if act.stats is not None else None
                serr = 0.0
                if st is not None:
                    s1 = o.sum((1, 2))
                    s2 = (o * o).sum((1, 2))
                    serr = max(((st[..., 0] - s1).abs().max() / s1.abs().max()).item(),
                               ((st[..., 1] - s2).abs().max() / s2.abs().max()).item())
                line += f"  {'s16' if split else 'f32'}: rel {rel:.2e} max {mx:.2e} stats {serr:.1e}"
            del y
        if "time" in what:
            B, C0, C1, Cout, H, ups = s[1], s[2], s[3], s[4], s[5], s[6]
            Ho = 2 * H if ups else H
            flops = 2.0 * B * Ho * Ho * Cout * (9 * (C0 + C1) + (64 if s[9] else 0))
            for split in (False, True):
                call = run(t, split)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                line += f"  {'s16' if split else 'f32'}: {ms * 1e3:8.1f} us {flops / ms / 1e9:7.1f} TF"
        print(line, flush=True)
        del t
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
