"""RCCL on the MI355X with a world of ONE rank: every collective of ddnm_amd/dist.py (the image all_gather that replaces the
reference's nn.DataParallel gather, guided_diffusion/diffusion.py:140,164,180; the scalar all_reduce of the PSNR sum, :602;
the broadcast of rank 0's decisions; the device-pinned barrier) executes on the `nccl` (= RCCL) backend with device tensors,
and `bench.py --gpus 1` started by torchrun reports `backend: "nccl (RCCL)"`, `ranks_seen: 1`.  Only one GPU is reachable
from the build container, so this is the execution of the multi-GPU code path that CAN be had before an 8-GPU node runs
it; every N > 1 throughput number stays unmeasured."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["DDNM_ROOT"])
import torch.distributed as dist
from ddnm_amd import dist as ddist
rank, local_rank, world = ddist.init()                 # RANK / WORLD_SIZE exported by the test: backend nccl = RCCL
assert (rank, world) == (0, 1) and dist.is_initialized() and ddist.backend_name() == "nccl", ddist.backend_name()
dev = torch.device("cuda", torch.cuda.current_device())
g = torch.Generator(device="cpu").manual_seed(3)
x = torch.randn(8, 3, 256, 256, generator=g).to(dev)   # the headline workload's gather payload: 8 images, 6.3 MB
calls = []
orig = dist.all_gather
dist.all_gather = lambda *a, **k: (calls.append(a[1].device.type), orig(*a, **k))[1]
full = ddist.gather_images(x)
dist.all_gather = orig
torch.cuda.synchronize()
assert calls == ["cuda"], calls                         # the collective ran, on a DEVICE tensor
assert full.data_ptr() != x.data_ptr() and torch.equal(full, x)
ragged = ddist.gather_images(x[:5], n_total=5)
assert torch.equal(ragged, x[:5])
assert ddist.reduce_scalar(2.25, dev, "sum") == 2.25 and ddist.reduce_scalar(7.0, dev, "max") == 7.0
assert ddist.broadcast_flag(True) is True and ddist.broadcast_flag(False) is False
ddist.barrier()                                         # dist.barrier(device_ids=[...]) on RCCL
# the sampler's own use: one restoration, gathered
ddist.shutdown()
assert not dist.is_initialized()
print("RCCL_WORLD1_OK")
'''


def test_every_collective_runs_on_rccl_with_one_rank(tmp_path):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DDNM_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DDNM_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", _WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_under_torchrun_with_one_rank_reports_rccl(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DDNM_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra-workloads", "--no-side-path", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]               # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["backend"] == "nccl (RCCL)" and line["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["steps"] == 1
