"""Multi-GPU layer: one process per GPU, reads sharded contiguously, no data-path collective.

The path is embarrassingly parallel over reads; the only exchange is the final
per-barcode call histogram, an int64[nY+1] all-reduce (RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests).
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous read range [lo, hi) of `rank`: GPU g gets reads [g*ceil(n/G), (g+1)*ceil(n/G))."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1.  backend defaults to
    "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo"."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    if backend == "nccl" and torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} has no GPU (visible: {torch.cuda.device_count()})")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def reduce_counts(counts):
    """In-place SUM all-reduce of the per-barcode call histogram (int64 tensor, any device the
    process group supports).  No-op for a single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (timing)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
