"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient arena.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for
tests).  The arena is cut into contiguous buckets; backward fills it from the last layer to the
first, so buckets are reduced in reverse arena order, each as soon as its lowest offset has
been written.  With async_op=True the collective runs on the communicator's own stream behind
an event on the compute stream, i.e. it overlaps the rest of backward; finish() makes the compute
stream wait for all of them before the optimizer reads the gradients.  The sum is scaled by
1/world inside the optimizer kernel (grad_scale), so no extra pass touches the arena.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def make_buckets(boundaries: Sequence[int], total: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Cut [0, total) at layer boundaries into buckets of about `bucket_elems` elements.

    boundaries: ascending start offsets of the layers (first must be 0).  Buckets are returned
    in REVERSE arena order (the order backward completes them); every bucket is a whole number
    of layers and the buckets tile [0, total) exactly.
    """
    if not boundaries or boundaries[0] != 0:
        raise ValueError("boundaries must start at 0")
    cuts = list(boundaries) + [total]
    buckets: List[Tuple[int, int]] = []
    end = total
    i = len(boundaries) - 1
    while i >= 0:
        start = cuts[i]
        # grow the bucket downwards until it is big enough
        while i > 0 and end - start < bucket_elems:
            i -= 1
            start = cuts[i]
        buckets.append((start, end))
        end = start
        i -= 1
    return buckets


class GradBucketer:
    """Issues one all-reduce per bucket as backward passes the bucket's lowest layer."""

    def __init__(self, arena: torch.Tensor, boundaries: Sequence[int],
                 bucket_bytes: int = 32 << 20, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.arena = arena
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # RCCL (and gloo on CPU tensors) overlap with backward through async ops.  gloo on GPU
        # tensors exists only to rehearse the N > 1 path on a one-GPU box; its device-tensor
        # collectives stall from the second round on (observed on ROCm), so that combination
        # stages each bucket through host memory and reduces it synchronously.
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._staged = backend == "gloo" and arena.is_cuda
        elems = max(1, bucket_bytes // arena.element_size())
        self.buckets = make_buckets(boundaries, arena.numel(), elems)
        self._ready_at = {start: (start, end) for (start, end) in self.buckets}
        self._pending: List = []

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def layer_done(self, layer_start: int) -> None:
        """Call after the gradient of the layer starting at `layer_start` (and of every layer
        above it) has been enqueued on the current stream."""
        if self.world == 1:
            return
        b = self._ready_at.get(layer_start)
        if b is None:
            return
        start, end = b
        view = self.arena[start:end]
        if self._staged:
            host = view.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            view.copy_(host)
            return
        self._pending.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))

    def finish(self) -> None:
        """Make the current stream (or the host, on CPU) wait for every issued bucket."""
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def reduce_all_now(self) -> None:
        """Reduce every bucket (in backward order) without overlap; used by tests."""
        for (start, _end) in self.buckets:
            self.layer_done(start)
        self.finish()
