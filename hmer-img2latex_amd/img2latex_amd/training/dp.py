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


class OverlappedAllReduce:
    """The ONE logical gradient all-reduce of a step, issued in two pieces so that it hides behind the backward pass:
    ``flat[:split]`` (the decoder's and the encoder FC's gradients: 46.1 of 46.5 MB at the primary dims) is final
    once the FC backward has been enqueued, ~1 ms of conv backward before the optimizer needs it, so it starts there
    (``start_early``, asynchronous: RCCL's own stream waits for the work enqueued so far and runs beside the conv
    kernels); ``flat[split:]`` (conv gradients + [loss sum, count]: 0.4 MB, latency-bound) follows at the end
    (``finish``), which also makes the current stream wait for the early piece.  Sums are element-wise, so the result
    is bit-identical to a single all-reduce of the whole buffer.  A no-op without an initialised process group."""

    def __init__(self, flat: torch.Tensor, split: int, group=None):
        self.flat, self.split, self.group = flat, int(split), group
        self._work = None

    def _active(self) -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start_early(self) -> None:
        """Exactly one finish() (TrainStep.apply) must follow each start_early() (TrainStep.forward_backward): the
        buffer is being reduced in place.  A second start_early() without it -- a forward_backward() repeated after an
        exception, or a caller only inspecting gradients -- first waits for the pending piece, so that the next
        backward does not overwrite memory a collective is still reading; the buffer then holds the REDUCED early
        gradients of the abandoned step, which the new backward overwrites.  Gradient accumulation (trainer.py:345-383) is
        TrainStep.micro_step: it switches the early piece off and reduces the accumulated buffer once, at the update."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._active() and 0 < self.split < self.flat.numel():
            import torch.distributed as dist
            self._work = dist.all_reduce(self.flat[: self.split], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        if not self._active():
            return
        import torch.distributed as dist
        if self._work is None:                      # start_early was not called (or there is nothing to split)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, group=self.group)
        self._work.wait()                           # the current stream waits for the early piece
        self._work = None


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None) -> None:
    """Replicas must start from identical weights."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
