"""Multi-GPU plumbing for the image-sharded path (one process per GPU).

Images are independent units: every image is coded with state reset to {0,0,0,255} and a
zeroed index (qoi.h:393-400, 533-537), so a batch shards across ranks with NO data-path
collective.  The only communication is the end-of-run gathering of a few counters
(all_reduce over RCCL on GPUs — backend "nccl" on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device=None):
    """Initialise the default process group when WORLD_SIZE > 1 (rendezvous on 127.0.0.1)."""
    import torch.distributed as dist
    rank, world, _ = env_world()
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_frames(rank: int, world: int, frames_per_rank: int) -> List[int]:
    """Weak scaling: rank r owns synthetic frame ids r*F .. r*F+F-1 (disjoint across ranks)."""
    return list(range(rank * frames_per_rank, (rank + 1) * frames_per_rank))


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Strong scaling helper: contiguous block partition of n_items over ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def reduce_counters(elapsed_s: float, counters: Sequence[float], device="cpu") -> Tuple[float, List[float]]:
    """MAX of the elapsed time and SUM of the counters over all ranks (identity if world == 1)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor(list(counters), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.cpu().numpy()]


def reduce_min(value: float, device="cpu") -> float:
    """MIN of a per-rank figure over all ranks (the fastest rank's elapsed time: skew = max - min)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
