#!/usr/bin/env python
"""Which ATen ops (= torch-launched kernels) one classifier-guidance evaluation and one guided reverse step still issue
(development tool: the product path should show allocations / views only)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier, make_cond_fn  # noqa: E402

B = int(os.environ.get("B", "2"))
kw = classifier_defaults()
kw["image_size"] = 256
clf = create_classifier(**kw)
g = torch.Generator().manual_seed(4321)
clf.load_state_dict({k: (torch.randn(v, generator=g) * (1.0 / max(1, int(torch.tensor(v[1:]).prod()))) ** 0.5 if len(v) > 1
                         else (1.0 + 0.1 * torch.randn(v, generator=g) if k.endswith("weight") else 0.05 * torch.randn(v, generator=g)))
                     for k, v in clf.state_dict_shapes().items()})
clf.convert_to_fp16()
fn = make_cond_fn(clf, 1.0)
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
y = torch.full((B,), 951, dtype=torch.long, device="cuda")
fn(x, t, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fn(x, t, y)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and (getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)) > 0]
print("ATen ops with device time in one guidance evaluation:")
for e in sorted(rows, key=lambda e: -e.count):
    print(f"  {e.key:40s} x{e.count}")
print("kernels launched by torch (names containing 'at::' / 'elementwise'):")
for e in prof.key_averages():
    if "at::" in e.key or "elementwise" in e.key:
        print(f"  {e.key[:100]:100s} x{e.count}")
