"""Multi-GPU driver pieces: one process per GPU, images sharded by index, one RCCL gather.

The reference scales with `torch.nn.DataParallel` (guided_diffusion/diffusion.py:140,164), i.e. a
parameter broadcast + scatter/gather inside EVERY model call.  Images of a batch are independent
trajectories, so here each rank keeps its own replica of the packed weights,
runs the whole reverse loop on its slice of the batch and of the noise tape, and the only
communication is ONE `all_gather` of the restored images (786 KB per image) over xGMI at the end.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# A process group of ONE rank still runs its collectives when the process was started by a distributed launcher (RANK and
# WORLD_SIZE in the environment: `torchrun --nproc-per-node 1`) or when a caller passes force=True: the RCCL code path of
# an 8-GPU job is then exactly the one a 1-GPU box can execute and test (tests/test_gpu_rccl.py).
_COLLECTIVES_AT_WORLD_1 = False


def _active(force=False):
    return dist.is_initialized() and (dist.get_world_size() > 1 or force or _COLLECTIVES_AT_WORLD_1)


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (backend "nccl" == RCCL on ROCm).  A world of one is
    initialised too when a launcher started the process (RANK / WORLD_SIZE exported) or `force` is set."""
    global _COLLECTIVES_AT_WORLD_1
    rank, local_rank, world = env_world()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world == 1 and (force or launched):
        _COLLECTIVES_AT_WORLD_1 = True
    if (world > 1 or _COLLECTIVES_AT_WORLD_1) and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; DDNM_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests on a 1-GPU box)
            backend = os.environ.get("DDNM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            # a world of one needs no agreed port: take a free one, so that two single-GPU jobs on one host (schedulers
            # that export RANK=0 / WORLD_SIZE=1) cannot collide on 29500
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_PORT", "29500")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:      # noqa: BLE001
            if world > 1:
                raise
            # a single rank needs no communication at all: a box without a usable RCCL / TCP store still runs
            import warnings
            warnings.warn(f"torch.distributed init failed for a world of one ({e!r}): running without a process group")
            _COLLECTIVES_AT_WORLD_1 = False
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of `n_items` images owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(rank, world, *tensors):
    """Slice every tensor / list-of-tensors (noise tape) along the image axis."""
    out = []
    for t in tensors:
        if isinstance(t, (list, tuple)):
            lo, hi = shard_range(t[0].shape[0], rank, world)
            out.append([x[lo:hi] for x in t])
        else:
            lo, hi = shard_range(t.shape[0], rank, world)
            out.append(t[lo:hi])
    return out


def gather_images(x_local, n_total=None, force=False):
    """The path's single collective: all ranks receive the full restored batch, in image order."""
    if not _active(force):
        return x_local
    world = dist.get_world_size()
    n_total = x_local.shape[0] * world if n_total is None else n_total
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = x_local
    if x_local.shape[0] < max_n:     # all_gather needs equal shapes
        pad = torch.cat([x_local, x_local.new_zeros(max_n - x_local.shape[0], *x_local.shape[1:])], 0)
    pad = pad.contiguous()
    if dist.get_backend() == "gloo" and pad.is_cuda:      # gloo gathers host tensors only
        host = pad.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        parts = [p.to(pad.device) for p in parts]
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def reduce_scalar(value, device, op="sum", force=False):
    """Scalar sum / max / min over ranks (gloo reduces host tensors, RCCL device tensors)."""
    if _active(force) and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _active(force):
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return t.item()


def reduce_sum(value, device):
    """Scalar sum over ranks (PSNR accumulation, diffusion.py:602)."""
    return reduce_scalar(value, device, "sum")


def world_state():
    """(rank, world) of the INITIALISED process group, (0, 1) without one.  A process started under torchrun that never
    called init() must not shard: gather_images / reduce_sum would silently keep its slice only."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if int(os.environ.get("WORLD_SIZE", 1)) > 1:
        raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call ddnm_amd.dist.init() first")
    return 0, 1


def backend_name():
    return dist.get_backend() if dist.is_initialized() else "none"


def barrier(force=False):
    if _active(force):
        if dist.get_backend() == "nccl":          # pin the RCCL barrier to this rank's GPU
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def broadcast_flag(flag, force=False):
    """Rank 0's boolean, on every rank (a collective: also orders rank 0's side effects before the others go on)."""
    if not _active(force):
        return bool(flag)
    dev = "cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=0)
    return bool(t.item())


def shutdown():
    global _COLLECTIVES_AT_WORLD_1
    if dist.is_initialized():
        dist.destroy_process_group()
    _COLLECTIVES_AT_WORLD_1 = False
