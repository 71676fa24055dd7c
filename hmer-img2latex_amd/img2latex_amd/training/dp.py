"""Data-parallel pieces of the training step (no reference counterpart: the reference is
single-process; parity is defined against its single-process batch).

One process per GPU.  Each rank runs forward/backward on its contiguous shard of the batch
and produces the gradient of the SUM of its per-token losses; ONE all-reduce (RCCL over
xGMI when the backend is "nccl") of a single flat fp32 buffer -- all gradients plus two
trailing slots (loss sum, non-PAD token count) -- followed by a division by the GLOBAL
count reproduces the reference's mean-over-all-non-PAD-tokens loss exactly (averaging
per-shard means would not: shards hold different token counts)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_batch(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of a batch of n that rank owns: contiguous, sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_of(group=None) -> int:
    """This process's data-parallel rank (0 when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group)
    return 0


def all_reduce_gradients(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the flat buffer [grads..., loss_sum, count]; a no-op when
    torch.distributed is not initialised (single GPU)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None) -> None:
    """Replicas must start from identical weights."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
