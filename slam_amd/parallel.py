"""Sharding of independent frames / sequences over ranks (one process per GPU).

The tracking hot path has no exchange step: every (sequence, frame) is independent, so ranks own
contiguous slices of the unit list and the only communication is the gather of the small result
arrays (and the max / sum reductions the benchmark reports).  Works on any torch.distributed
backend: "nccl" (= RCCL on ROCm) with device tensors, "gloo" with CPU tensors.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Rank g owns units [g*n/W, (g+1)*n/W): contiguous, sizes differ by at most one."""
    lo = (n_units * rank) // world
    hi = (n_units * (rank + 1)) // world
    return lo, hi


def gather_records(local, device=None):
    """All-gather a rank-local 1-D numpy record array (any dtype); every rank gets the
    concatenation in rank order.  Without a process group: returns `local` unchanged (a group of one rank still
    runs the collectives, on `device`)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.concatenate([p[:s].cpu().numpy() for p, s in zip(parts, sizes)])
    return out.view(local.dtype)


def reduce_max_sum(elapsed, count, device=None):
    """(max over ranks of elapsed, sum over ranks of count) — what the benchmark reports."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed), float(count)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    c = torch.tensor([float(count)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())


def reduce_max_vector(values, device=None):
    """Element-wise max over ranks of a list of floats (the per-block times of the benchmark)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.cpu().tolist()]
