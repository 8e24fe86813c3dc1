"""Host-side replay of a recorded pass.

A training step of a branched network is thousands of small launches, and walking the Python
graph to issue them costs the host more than the GPU needs to run them.  With static shapes
and persistent buffers the walk always produces the same calls, so it is done once:

  * the library records every launch it enqueues between pe_tape_begin and pe_tape_end with a
    copy of its arguments (csrc/pe_common.h, pe::launch), and pe_tape_replay enqueues a range
    of them again at the cost of one hipLaunchKernel each;
  * what the pass does through torch between those launches (zero fills, copies) is seen by a
    TorchDispatchMode and kept as (stream, op, arguments) to be issued again at the same place;
  * the pass allocates from a memory pool of its own (torch.cuda.MemPool), which the recording
    keeps: memory freed inside the pass is reused inside the pass, as in an eager step, but
    never handed to anyone else, so the addresses stored in the tape stay valid.  The pool
    replays the allocator's decisions as they were taken while recording, so what a side
    stream reads must be kept alive by the pass until the streams have joined (reuse ordered
    only by an event the host happened to see completed would not be ordered in a replay).

The reference re-ran one compiled TensorFlow graph per step (`train.py`, absent from the
snapshot; test.ipynb:1060 drives it): this is that idea for the library's launches, without a
HIP graph (replaying thousands of kernel nodes measured slower than enqueueing them).
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

from ._ffi import check, lib

aten = torch.ops.aten

# results that are fresh memory filled with a constant: replayed as a fill of that memory
_FILL_FACTORIES = {
    aten.zeros.default: 0, aten.zeros_like.default: 0, aten.ones.default: 1,
    aten.ones_like.default: 1, aten.new_zeros.default: 0, aten.new_ones.default: 1,
}
_FULL_FACTORIES = {aten.full.default: 1, aten.full_like.default: 1, aten.new_full.default: 2}
# results that are fresh memory holding a copy of the first argument
_COPIES = {aten.clone.default, aten._to_copy.default}
# fresh, uninitialised memory
_ALLOCS = {aten.empty.memory_format, aten.empty_like.default, aten.empty_strided.default,
           aten.new_empty.default, aten.new_empty_strided.default}
_IGNORED = {aten.record_stream.default}
# views whose schema does not say so
_VIEWS = {aten._unsafe_view.default}


class NotReplayable(RuntimeError):
    """The pass did something a recorded call list cannot repeat."""


def wait_stream(waiter: torch.cuda.Stream, waited: torch.cuda.Stream) -> None:
    """waiter.wait_stream(waited) through the library, so that a recording tape sees it."""
    check(lib().pe_stream_wait_stream(waiter.cuda_stream, waited.cuda_stream))


def recording() -> bool:
    """True while the library records this thread's launches."""
    return lib().pe_tape_length() >= 0


def _cuda_tensors(tree) -> List[torch.Tensor]:
    return [t for t in tree_flatten(tree)[0] if isinstance(t, torch.Tensor) and t.is_cuda]


class _Recorder(TorchDispatchMode):
    def __init__(self, owner: "RecordedPass"):
        super().__init__()
        self.owner = owner

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        self.owner._note(func, args, kwargs, out)
        return out


class RecordedPass:
    """fn(*inputs) recorded once (it runs for real while being recorded) and replayed by
    __call__ with new inputs copied into the recorded input buffers.  The outputs live in the
    recording's own memory and are overwritten by the next replay.  fn must issue the same
    calls whatever its inputs hold (no data-dependent host decisions, no .item()), must not
    pass the library values that change from step to step, and replays must be issued with the
    stream current that was current while recording."""

    def __init__(self, fn: Callable, inputs: Sequence[Optional[torch.Tensor]],
                 device: torch.device, scratch: Optional[Callable[[], list]] = None):
        self.device = torch.device(device)
        self.keep: list = []           # scratch that lives outside the pass but is on the tape
        self.static_in = [None if t is None else t.clone() for t in inputs]
        self.items: list = []          # (None, first, count) | (stream, func, args, kwargs)
        self.pool = torch.cuda.MemPool()
        self._mark = 0
        self._tape = ctypes.c_void_p()
        self._lib = lib()
        self.main = torch.cuda.current_stream(self.device)
        check(self._lib.pe_tape_begin())
        try:
            with torch.cuda.use_mem_pool(self.pool, device=self.device), _Recorder(self):
                self.static_out = fn(*self.static_in)
            self._close_segment()
            # The tape also addresses scratch the pass did not allocate: the module-level
            # weight-gradient workspaces and the owner's batch-norm workspaces.  Those are
            # replaced (and the old buffer freed) when a later, larger user makes them grow;
            # holding the tensors that were current while recording keeps every recorded
            # address valid for as long as this recording lives.
            from . import nn as _nn
            self.keep = list(_nn.live_scratch(self.device))
            if scratch is not None:
                self.keep += list(scratch())
        finally:
            check(self._lib.pe_tape_end(ctypes.byref(self._tape)))
        self.launches = sum(it[2] for it in self.items if it[0] is None)
        self.torch_ops = sum(1 for it in self.items if it[0] is not None)

    # ------------------------------------------------------------------ recording
    def _close_segment(self) -> None:
        n = int(self._lib.pe_tape_length())
        if n > self._mark:
            self.items.append((None, self._mark, n - self._mark))
            self._mark = n

    def _note(self, func, args, kwargs, out) -> None:
        outs = _cuda_tensors(out)
        if func in _IGNORED or func in _ALLOCS or func in _VIEWS:
            return
        ins = _cuda_tensors((args, kwargs))
        if not outs and not ins:
            return                                     # host-side arithmetic
        if any(r.alias_info is not None and not r.alias_info.is_write
               for r in func._schema.returns):
            return                                     # a view of existing memory
        stream = torch.cuda.current_stream(self.device)
        if func._schema.is_mutable:                    # in-place and out= variants
            item = (stream, func, args, kwargs)
        elif func in _FILL_FACTORIES:
            item = (stream, aten.fill_.Scalar, (out, _FILL_FACTORIES[func]), {})
        elif func in _FULL_FACTORIES:
            item = (stream, aten.fill_.Scalar, (out, args[_FULL_FACTORIES[func]]), {})
        elif func in _COPIES:
            item = (stream, aten.copy_.default, (out, args[0]), {})
        else:
            raise NotReplayable(
                f"{func} allocates its result (or reads the device from the host) inside a "
                "recorded pass; use an in-place or out= form, or keep it outside the pass")
        self._close_segment()
        self.items.append(item)

    # ------------------------------------------------------------------ replay
    def __call__(self, *inputs: Optional[torch.Tensor]):
        cur = entry = torch.cuda.current_stream(self.device)
        if entry.cuda_stream != self.main.cuda_stream:
            raise RuntimeError("a recorded pass replays on the stream it was recorded on")
        for dst, src in zip(self.static_in, inputs):
            if dst is not None and src is not dst:
                dst.copy_(src)
        replay, tape = self._lib.pe_tape_replay, self._tape
        try:
            for st, a, b, *rest in self.items:
                if st is None:
                    rc = replay(tape, a, b)
                    if rc:
                        check(rc)
                else:
                    if st.cuda_stream != cur.cuda_stream:
                        torch.cuda.set_stream(st)
                        cur = st
                    a(*b, **rest[0])
        finally:
            if cur.cuda_stream != entry.cuda_stream:
                torch.cuda.set_stream(entry)
        return self.static_out

    def close(self) -> None:
        if self._tape:
            self._lib.pe_tape_destroy(self._tape)
            self._tape = ctypes.c_void_p()
        self.items.clear()
        self.static_out = None
        self.keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_recorded(cache: dict, net, fn: Callable, inputs: Sequence[Optional[torch.Tensor]]):
    """fn(*inputs) through the RecordedPass cached for these input shapes.  The first call
    runs fn twice: once with the running statistics held still, for whatever is set up on
    first use (workspace growth, kernel attributes), then for real while being recorded."""
    key = (fn.__name__,) + tuple(None if t is None else (tuple(t.shape), t.dtype) for t in inputs)
    rp = cache.get(key)
    if rp is not None:
        return rp(*inputs)
    momentum, net.bn_momentum = net.bn_momentum, 0.0
    try:
        fn(*inputs)
    finally:
        net.bn_momentum = momentum
    rp = cache[key] = RecordedPass(fn, inputs, net.device,
                                   scratch=lambda: list(getattr(net, "_ws", {}).values()))
    return rp.static_out
