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


def make_buckets(boundaries: Sequence[int], total: int, bucket_elems: int,
                 tail_elems: Optional[int] = None) -> List[Tuple[int, int]]:
    """Cut [0, total) at layer boundaries into buckets of about `bucket_elems` elements.

    boundaries: ascending start offsets of the layers (first must be 0).  Buckets are returned
    in REVERSE arena order (the order backward completes them); every bucket is a whole number
    of layers and the buckets tile [0, total) exactly.

    tail_elems: the bucket backward completes LAST (the first layers of the network) cannot
    overlap anything: its all-reduce is exposed in full before the optimizer.  When given, that
    bucket is cut down to about tail_elems elements and the rest of it becomes a bucket of its
    own, which is reduced while the first layers are still being differentiated.
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
    if tail_elems is not None and buckets:
        lo, hi = buckets[-1]
        inner = [c for c in boundaries if lo < c < hi and c - lo >= tail_elems]
        if inner:
            cut = inner[0]
            buckets[-1] = (cut, hi)
            buckets.append((lo, cut))
    return buckets


class GradBucketer:
    """Issues one all-reduce per bucket as backward passes the bucket's lowest layer."""

    def __init__(self, arena: torch.Tensor, boundaries: Sequence[int],
                 bucket_bytes: int = 32 << 20, group=None, tail_bytes: Optional[int] = None):
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
        if tail_bytes is None:
            tail_bytes = bucket_bytes // 8
        self.buckets = make_buckets(boundaries, arena.numel(), elems,
                                    max(1, tail_bytes // arena.element_size()))
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


class NativeComm:
    """The library's own RCCL communicator (pe_comm_*): the route for hosts without
    torch.distributed.  rank 0 makes `unique_id()` and the launcher hands the 128 bytes to the
    other ranks; every rank then constructs NativeComm(rank, world, id) with its GPU current."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes

        from ._ffi import check, lib

        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        self._lib, self._check = lib(), check
        self.rank, self.world = rank, world
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._h = ctypes.c_void_p()
        check(self._lib.pe_comm_create(rank, world, ctypes.addressof(buf), ctypes.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from ._ffi import check, lib

        buf = (ctypes.c_uint8 * 128)()
        check(lib().pe_comm_unique_id(ctypes.addressof(buf)))
        return bytes(buf)

    def all_reduce_(self, t: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """t = scale * sum over ranks of t, in place, asynchronous on torch's current stream."""
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("all_reduce_: contiguous float32 / bfloat16 GPU tensor")
        dt = 0 if t.dtype == torch.float32 else 1
        self._check(self._lib.pe_allreduce_bucket(self._h, t.data_ptr(), t.numel(), dt, float(scale),
                                                  torch.cuda.current_stream(t.device).cuda_stream))
        return t

    def close(self) -> None:
        if self._h:
            self._lib.pe_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
