#!/usr/bin/env python
"""Experiment: two independent half-batch ADM chains (B/2 images each) as hipGraph replays on two HIP streams, started a
fraction of a forward apart, against ONE chain at the full batch (eager and graph).  Images of a batch are independent
trajectories, so this only re-orders launches; the question is whether the latency-bound low-resolution section of one
chain fills under the chip-filling high-resolution section of the other.  (Two host threads driving eager chains are
host-bound: 19 ... 27 ms per 4 images against 11.6.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd.guided_diffusion.unet import create_model  # noqa: E402

B = int(os.environ.get("B", "4"))
N = int(os.environ.get("N", "20"))
m = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=64,
                 learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True)
m.load_state_dict(m.random_state_dict(1))
m.convert_to_fp16()
x = torch.randn(B, 3, 256, 256, device="cuda")
t = torch.full((B,), 500.0, device="cuda")


def timed(fn, n=N):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


one = timed(lambda: m(x, t))
print(f"one eager chain, B={B}: {one * 1e3:.2f} ms per {B} images")


def capture(xx, tt):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(xx, tt)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(xx, tt)
    return g, out


gfull, _ = capture(x, t)
print(f"one graph chain, B={B}: {timed(gfull.replay) * 1e3:.2f} ms per {B} images")
h = B // 2
halves = [capture(x[i * h:(i + 1) * h].contiguous(), t[i * h:(i + 1) * h].contiguous()) for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
clock_hz = 2.0e9
for off_ms in (0.0, 2.0, 4.0, 6.0):
    def both():
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                halves[i][0].replay()
    # the offset is applied once: chain 1 sleeps before its first replay, both then free-run
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[1]):
        if off_ms:
            torch.cuda._sleep(int(off_ms * 1e-3 * clock_hz))
    for _ in range(3):
        both()
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[1]):
        if off_ms:
            torch.cuda._sleep(int(off_ms * 1e-3 * clock_hz))
    t0 = time.perf_counter()
    for _ in range(N):
        both()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"two graph chains of B={h}, offset {off_ms} ms: {dt * 1e3:.2f} ms per {B} images  (x{one / dt:.3f} vs eager)")
