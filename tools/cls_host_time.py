#!/usr/bin/env python
"""Host time to ENQUEUE one classifier-guidance evaluation vs its GPU time (is the guidance chain host-bound?), and the same
for one ADM UNet forward (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier, make_cond_fn  # noqa: E402
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

B = int(os.environ.get("B", "32"))
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
for _ in range(2):
    fn(x, t, y)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    fn(x, t, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"classifier B={B}: host enqueue {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms", flush=True)
m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True, class_cond=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
xb, tb, yb = x[:8].contiguous(), t[:8].contiguous(), y[:8].contiguous()
for _ in range(2):
    m(xb, tb, yb)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    m(xb, tb, yb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"ADM UNet B=8: host enqueue {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms", flush=True)
