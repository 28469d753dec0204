"""
Multi-GPU sharding of independent frames.

Frames are the units: no kernel couples two frames (reference
nn/basic.py:679-787 builds one graph per structure). So N GPUs = N processes,
each owning a contiguous block of frames and its own model copy; the only
exchange is ONE all-reduce of the 8-byte batch energy (RCCL over xGMI when the
backend is "nccl"; gloo on CPU for tests). The reference has no inference-side
collective; its only collective is the training gradient all-reduce
(train/distribute_utils.py:56-81): see `tensoralloy_amd/train.py::allreduce_mean`.
"""
from __future__ import annotations

import os
from typing import Tuple


def world_from_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_units` owned by `rank`; sizes differ by <= 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def allreduce_sum_(tensor, async_op=False):
    """In-place sum over the default process group (no-op for world size 1)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=async_op)
