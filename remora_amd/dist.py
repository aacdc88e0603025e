"""Multi-GPU layer: reads / chunk ranges shard embarrassingly (one process per GPU, weights
replicated, no data-path collective); the only exchange is ONE all-reduce(sum) of the
per-label call counts at the end of a run — the distributed form of the label tally in
src/remora/validate.py:42-45 / get_label_counts src/remora/data_chunks.py:1074-1082.
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
The payload is int64[num_out] (<= 128 B): latency-bound, link bandwidth irrelevant."""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n, rank, world):
    """Contiguous [start, stop) of `n` units for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(backend=None, set_device=True):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world <= 1:
        return rank, world, local
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and set_device:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_counts(counts):
    """Sum per-label counts over all ranks.  `counts`: int64 torch tensor (on the GPU for
    nccl/RCCL, CPU for gloo) or numpy array; returns the same type, reduced in place for tensors."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return counts
    if isinstance(counts, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(counts, np.int64))
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()
    if counts.is_cuda and dist.get_backend() != "nccl":  # gloo (tests): reduce through the host
        t = counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        counts.copy_(t)
        return counts
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def allreduce_max_float(x):
    """max over ranks of a python float (used for the max-over-ranks step time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
