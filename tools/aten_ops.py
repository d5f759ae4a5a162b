#!/usr/bin/env python
"""Which ATen ops (= torch-launched kernels) one ADM forward still issues (development tool)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
x = torch.randn(4, 3, 256, 256, device="cuda")
t = torch.full((4,), 500.0, device="cuda")
m(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    m(x, t)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=12, max_src_column_width=90))
