#!/usr/bin/env python
"""A/B of the 256-pixel 3x3 launches on the product library: conv16_kernel<9,4,4> (DDNM_CONV16B=0) against the
two-workgroups-per-CU kernel of csrc/conv16b.hip (=1; DDNM_CONV16B_DEPHASE=0 without the first-round stagger).  The switches
are read once per process, so the script re-runs itself per setting.  GN=1 fuses the GroupNorm affine + swish."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("256->256@256 res", 256, 0, 256, 256, 1), ("256->256@256", 256, 0, 256, 256, 0), ("512->256@256 cat", 256, 256, 256, 256, 0),
          ("128->256@256... 64->256", 64, 0, 256, 256, 1), ("256->256@128 res", 256, 0, 256, 128, 1), ("512->512@128 res", 512, 0, 512, 128, 1),
          ("768->512@128 cat", 512, 256, 512, 128, 0), ("256->512@128", 256, 0, 512, 128, 0)]


def child():
    import torch
    from ddnm_amd import ops
    B = int(os.environ.get("B", "4"))
    gn = os.environ.get("GN", "1") == "1"
    out = []
    for name, C0, C1, Cout, H, res in SHAPES:
        g = torch.Generator().manual_seed(5)
        x0 = torch.randn(B, H, H, C0, generator=g).half().cuda()
        x1 = torch.randn(B, H, H, C1, generator=g).half().cuda() if C1 else None
        w = ops.pack_conv_weight16((torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.02).cuda())
        bias = torch.randn(Cout, generator=g).cuda()
        r = torch.randn(B, H, H, Cout, generator=g).half().cuda() if res else None
        sc = (1 + 0.1 * torch.randn(B, C0 + C1, generator=g)).cuda()
        sh = (0.1 * torch.randn(B, C0 + C1, generator=g)).cuda()
        kw = dict(src1=x1, bias=bias, res=r, gn=(sc, sh) if gn else None)
        for _ in range(3):
            o = ops.conv16(x0, w, Cout, 3, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o = ops.conv16(x0, w, Cout, 3, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * B * H * H * Cout * 9 * (C0 + C1)
        out.append((name, us, fl / us / 1e6, float(o.t.float().abs().sum()), float(o.stats.sum()) if o.stats is not None else 0.0))
    for row in out:
        print("%-26s %8.1f us %7.1f TF/s  chk %.6e %.6e" % row, flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD") == "1":
        child()
    else:
        for label, env in (("old <9,4,4>", {"DDNM_CONV16B": "0"}), ("conv16b", {"DDNM_CONV16B": "1"}),
                           ("conv16b no stagger", {"DDNM_CONV16B": "1", "DDNM_CONV16B_DEPHASE": "0"}),
                           ("conv16b no-epilogue probe (wrong results)", {"DDNM_CONV16B": "1", "DDNM_P16B_WCONTIG": "2"})):
            print("==", label, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, AB_CHILD="1", **env))
