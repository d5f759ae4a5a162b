#!/usr/bin/env python
"""ADM fp16-torso forwards at B=4 for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
x = torch.randn(4, 3, 256, 256, device="cuda")
t = torch.full((4,), 500.0, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    m(x, t)
torch.cuda.synchronize()
print("done")
