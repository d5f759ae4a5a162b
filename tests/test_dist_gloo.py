"""N>1 path on CPU: world_size-2 `gloo` run of the sharding + single-gather logic (ddnm_amd/dist.py).
The per-image work is replaced by a deterministic stand-in (the HIP engine needs a GPU); what is
under test is that images and noise tapes are partitioned by index and that the one collective of
the path returns every image, in order, on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ddnm_amd import dist as ddist
    r, lr, w = ddist.init(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n_images, 3, 8, 8, generator=g)
    tape = [torch.randn(n_images, 3, 8, 8, generator=g) for _ in range(3)]
    xs, ts = ddist.shard_batch(rank, world, x, tape)
    lo, hi = ddist.shard_range(n_images, rank, world)
    assert xs.shape[0] == hi - lo and all(t.shape[0] == hi - lo for t in ts)
    local = xs * 2 + ts[0] - ts[2]                  # stand-in for one restored shard
    full = ddist.gather_images(local, n_total=n_images)
    total = ddist.reduce_sum(float(local.sum()), "cpu")
    ddist.barrier()
    torch.save({"full": full, "total": total}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [8, 5])
def test_two_rank_shard_and_gather(tmp_path, n_images):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_images, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n_images, 3, 8, 8, generator=g)
    tape = [torch.randn(n_images, 3, 8, 8, generator=g) for _ in range(3)]
    want = x * 2 + tape[0] - tape[2]
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(got["full"], want)
        assert abs(got["total"] - float(want.sum())) < 1e-3


def test_shard_range_partitions_exactly():
    from ddnm_amd.dist import shard_range
    for n in (1, 7, 8, 32, 33):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _world1_worker(rank, port, out_dir):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ddnm_amd import dist as ddist
    r, lr, w = ddist.init(backend="gloo")           # a launcher-started world of ONE still gets a process group ...
    assert (r, w) == (0, 1) and dist.is_initialized() and ddist.backend_name() == "gloo"
    x = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    calls = []
    orig = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), orig(*a, **k))[1]
    full = ddist.gather_images(x)                   # ... and its collectives really run (the 8-GPU code path)
    dist.all_gather = orig
    assert calls == ["all_gather"] and torch.equal(full, x)
    assert ddist.reduce_scalar(3.5, "cpu", "sum") == 3.5 and ddist.broadcast_flag(True) is True
    ddist.barrier()
    ddist.shutdown()
    assert not dist.is_initialized()
    assert ddist.gather_images(x) is x              # without a group: the identity, no collective
    open(os.path.join(out_dir, "ok"), "w").write("ok")


def test_world_of_one_under_a_launcher_runs_its_collectives(tmp_path):
    mp.spawn(_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok")
