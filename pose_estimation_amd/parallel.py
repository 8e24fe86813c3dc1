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


class _HostStagedExchange:
    """gloo has no device collectives that survive more than one round on this stack, so a
    rehearsal of the N > 1 path on one GPU sums buckets on the host.  It keeps the ordering the
    RCCL path has: a bucket is reduced BEHIND an event recorded on the stream that produced it,
    on a stream of its own, while the caller keeps enqueueing backward kernels; finish() makes
    the caller's stream wait for every bucket.  One worker thread issues the collectives in
    submission order, which is the same on every rank."""

    def __init__(self, dist, group, device):
        import queue
        import threading

        self.dist, self.group, self.device = dist, group, device
        self.stream = torch.cuda.Stream(device=device)
        self.q: "queue.Queue" = queue.Queue()
        self.done: List[torch.cuda.Event] = []
        self.error: Optional[BaseException] = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self) -> None:
        torch.cuda.set_device(self.device)
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                view, ready = item
                if self.error is None:
                    ready.synchronize()                      # the producer's kernels are done
                    with torch.cuda.stream(self.stream):
                        host = view.to("cpu", non_blocking=False)
                        self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                        view.copy_(host, non_blocking=False)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    self.done.append(ev)
            except BaseException as exc:  # noqa: BLE001 - re-raised on the caller's thread
                self.error = exc
            finally:
                self.q.task_done()

    def submit(self, view: torch.Tensor) -> None:
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self.q.put((view, ready))

    def finish(self) -> None:
        self.q.join()
        if self.error is not None:
            err, self.error = self.error, None
            raise err
        cur = torch.cuda.current_stream(self.device)
        for ev in self.done:
            cur.wait_event(ev)
        self.done.clear()

    def close(self) -> None:
        self.q.put(None)


class GradBucketer:
    """Issues one all-reduce per bucket as backward passes the bucket's lowest layer.

    Backends, all behind the same layer_done() / finish() pair:
      * torch.distributed "nccl" (RCCL): async collectives on the communicator's stream;
      * comm=NativeComm: the library's own RCCL communicator (pe_allreduce_bucket) on a stream
        of this object, ordered by events -- no torch.distributed involved;
      * torch.distributed "gloo" with a GPU arena: host-staged rehearsal (_HostStagedExchange);
      * torch.distributed "gloo" with a CPU arena: async collectives (CPU tests).
    transport="bf16" sends each bucket as bf16 (half the bytes on the links): the float32
    bucket is packed into a staging buffer, summed in bf16 and unpacked over the bucket.
    The sum is never scaled here: the optimizer applies grad_scale = 1/world.
    """

    def __init__(self, arena: torch.Tensor, boundaries: Sequence[int],
                 bucket_bytes: int = 32 << 20, group=None, tail_bytes: Optional[int] = None,
                 comm: Optional["NativeComm"] = None, transport: str = "f32",
                 exchange_when_alone: bool = False):
        """exchange_when_alone: issue the collectives even in a one-rank process group (a sum
        over one rank changes nothing; this is how the RCCL path is exercised on one GPU)."""
        import torch.distributed as dist

        if transport not in ("f32", "bf16"):
            raise ValueError("transport must be 'f32' or 'bf16'")
        if transport == "bf16" and not arena.is_cuda:
            raise ValueError("bf16 transport packs on the GPU: the arena must be a GPU tensor")
        self.dist = dist
        self.group = group
        self.arena = arena
        self.comm = comm
        self.transport = transport
        if comm is not None:
            self.world = comm.world
            backend = "native"
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.backend = backend
        self.exchange_when_alone = bool(exchange_when_alone)
        # enabled = False: layer_done() and finish() do nothing, every rank keeps its local
        # gradients.  A measurement switch (the step without its exchange); the ranks'
        # parameters drift apart while it is off.
        self.enabled = True
        self._staged = (_HostStagedExchange(dist, group, arena.device)
                        if backend == "gloo" and arena.is_cuda else None)
        self._comm_stream = torch.cuda.Stream(device=arena.device) if comm is not None else None
        elems = max(1, bucket_bytes // arena.element_size())
        if tail_bytes is None:
            tail_bytes = bucket_bytes // 8
        self.buckets = make_buckets(boundaries, arena.numel(), elems,
                                    max(1, tail_bytes // arena.element_size()))
        self._ready_at = {start: (start, end) for (start, end) in self.buckets}
        self._pending: List = []
        # bf16 transport: one staging buffer per bucket (68 MB for the 34 M-parameter arena), so
        # buckets in flight never share memory
        self._stage = {}
        if transport == "bf16":
            for (start, end) in self.buckets:
                self._stage[start] = torch.empty(end - start, dtype=torch.bfloat16,
                                                 device=arena.device)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def describe(self) -> dict:
        """What one step puts on the links: for logs and benchmark lines."""
        per = 2 if self.transport == "bf16" else self.arena.element_size()
        sizes = [(end - start) * per for (start, end) in self.buckets]
        return {"backend": self.backend, "world": self.world, "transport": self.transport,
                "buckets": len(self.buckets), "bucket_bytes": sizes,
                "bytes_per_step": int(sum(sizes))}

    # ---- one bucket ------------------------------------------------------------------
    def _pack(self, view: torch.Tensor, stage: torch.Tensor) -> None:
        from ._ffi import check, lib

        check(lib().pe_grad_pack_bf16(view.data_ptr(), stage.data_ptr(), view.numel(),
                                      torch.cuda.current_stream(view.device).cuda_stream))

    def _unpack(self, stage: torch.Tensor, view: torch.Tensor) -> None:
        from ._ffi import check, lib

        check(lib().pe_grad_unpack_bf16(stage.data_ptr(), view.data_ptr(), view.numel(),
                                        torch.cuda.current_stream(view.device).cuda_stream))

    def layer_done(self, layer_start: int) -> None:
        """Call after the gradient of the layer starting at `layer_start` (and of every layer
        above it) has been enqueued on the current stream."""
        if not self.enabled:
            return
        if self.world == 1 and self.comm is None and not self.exchange_when_alone:
            return
        b = self._ready_at.get(layer_start)
        if b is None:
            return
        start, end = b
        view = self.arena[start:end]
        stage = self._stage.get(start)
        payload = view
        if stage is not None:
            self._pack(view, stage)        # on the producing stream, right behind the gradients
            payload = stage
        if self.comm is not None:
            cur = torch.cuda.current_stream(view.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            self._comm_stream.wait_event(ready)
            with torch.cuda.stream(self._comm_stream):
                self.comm.all_reduce_(payload)
                done = torch.cuda.Event()
                done.record(self._comm_stream)
            self._pending.append((done, stage, view))
        elif self._staged is not None:
            self._staged.submit(payload)
            self._pending.append((None, stage, view))
        else:
            work = self.dist.all_reduce(payload, op=self.dist.ReduceOp.SUM, group=self.group,
                                        async_op=True)
            self._pending.append((work, stage, view))

    def finish(self) -> None:
        """Make the current stream (or the host, on CPU) wait for every issued bucket; with
        bf16 transport the summed buckets are unpacked over the float32 arena here."""
        if self._staged is not None:
            self._staged.finish()
        for handle, stage, view in self._pending:
            if handle is None:
                pass
            elif isinstance(handle, torch.cuda.Event):
                torch.cuda.current_stream(view.device).wait_event(handle)
            else:
                handle.wait()
            if stage is not None:
                self._unpack(stage, view)
        self._pending.clear()

    def reduce_all_now(self) -> None:
        """Reduce every bucket (in backward order) without overlap; used by tests."""
        for (start, _end) in self.buckets:
            self.layer_done(start)
        self.finish()

    def close(self) -> None:
        if self._staged is not None:
            self._staged.close()
            self._staged = None


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
        """t = sum over ranks of scale * t, in place, asynchronous on torch's current stream
        (the scale rides inside the reduction; 1.0 is a plain sum)."""
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
