"""Training graphs on the HIP kernels for networks that are not a plain chain of blocks
(multi-branch, multi-head): parameters in flat arenas, forward layers that record what their
backward needs on a tape, a reverse sweep that accumulates activation gradients, bucketed
data-parallel gradient exchange and the fused optimizer.

The reference's training code (train.py, model/base_model.py; PoseEstimation.pyproj:30,34) is
absent; what is visible is `Trainer(config)`, Adam(0.9, 0.999, 1e-8) with exponential decay
(test.ipynb:400, 1001-1008) and Horovod-style data parallelism (test.ipynb:12): local batch
statistics, summed gradients averaged over ranks.

A layer is conv + batch norm (+ residual) (+ ReLU / ReLU6), or a head (1x1 conv + bias to
float32 NCHW maps).  In training mode every layer is
    conv with fused batch statistics -> finalize -> normalise/residual/activation
and pushes (input, pre-norm output, output) on the tape; backward pops the tape: batch-norm
backward, weight gradient on a side stream, backward-data into the input's gradient (added to
what other consumers already left there).  In inference mode the same graph code runs the
folded-BN kernels and records nothing.
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import nn as pnn
from . import ops
from . import replay
from .parallel import GradBucketer

BF16 = torch.bfloat16
BN_EPS = 1e-5
NARROW = (8, 16, 32)


@dataclass
class Layer:
    name: str
    kind: str            # "conv" | "dw" (depthwise 3x3) | "stem3" (3x3/2 on the space-to-depth
                         # image) | "head" (1x1 conv + bias to float32 NCHW maps)
    cin: int
    cout: int
    k: int = 1
    stride: int = 1
    relu: int = 1        # activation after BN (+ residual): 0 none, 1 ReLU, 2 ReLU6
    dilation: int = 1    # depthwise layers only

    @property
    def pad(self) -> int:
        return self.k // 2

    @property
    def narrow(self) -> bool:
        return self.kind in ("conv", "head") and self.cin in NARROW

    @property
    def kpad(self) -> int:
        """A head's loss gradient is zero-padded to this many channels for the MFMA passes."""
        return (self.cout + 63) // 64 * 64

    def packed_shape(self) -> Tuple[int, ...]:
        if self.kind == "stem3":
            return (self.cout, 2, 64)
        if self.kind == "dw":
            return (9, self.cout)
        if self.narrow:
            P = 64 // self.cin
            return (self.cout, self.k, (self.k + P - 1) // P, 64)
        return (self.cout, self.k, self.k, self.cin)

    def pack(self, w_oihw: torch.Tensor) -> torch.Tensor:
        """Framework-layout float32 weight -> packed layout, still float32."""
        w = w_oihw.float()
        if self.kind == "stem3":
            return pnn.pack_stem3x3_weight(w, torch.float32)
        if self.kind == "dw":                       # [C,1,3,3] -> [9,C]
            return w.reshape(self.cout, 9).t().contiguous()
        if self.narrow:
            return pnn.pack_conv_weight_narrow(w, torch.float32)
        return pnn.pack_conv_weight(w, torch.float32)

    def param_mask(self) -> Optional[torch.Tensor]:
        """1 where the packed layout holds a parameter, None when it is dense."""
        if self.kind == "stem3" or self.narrow:
            return self.pack(torch.ones(self.cout, 3 if self.kind == "stem3" else self.cin,
                                        self.k, self.k))
        return None


class Var:
    """An activation and, during backward, the gradient accumulated for it so far."""
    __slots__ = ("t", "grad", "needs_grad", "consumers", "prod", "dz", "ngrad", "prod_res")

    def __init__(self, t: torch.Tensor, needs_grad: bool = True):
        self.t = t
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self.consumers = 0      # layers that read this activation in the current forward pass
        self.prod = None        # (layer, z, act) when it is act(bn(z)) with no residual
        self.dz = None          # gradient of that z, when the consumer's backward-data made it
        self.ngrad = 0          # gradient contributions received so far in backward
        self.prod_res = None    # (layer, z) when it is relu(bn(z) + residual)


def _align8(n: int) -> int:
    return (n + 7) // 8 * 8


class TrainGraph:
    """Parameter arenas + tape for a list of layers given in forward execution order."""

    def __init__(self, device, layers: Sequence[Layer], bn_momentum: float = 0.1):
        self.device = torch.device(device)
        self.layers: List[Layer] = list(layers)
        self.by_name: Dict[str, Layer] = {L.name: L for L in self.layers}
        if len(self.by_name) != len(self.layers):
            raise ValueError("layer names must be unique")
        self.bn_momentum = bn_momentum
        dev = self.device
        # ---- arena layout: every tensor starts on an 8-element boundary
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.layer_starts: List[int] = []
        off = 0
        for L in self.layers:
            self.layer_starts.append(off)
            shp = L.packed_shape()
            n = 1
            for s in shp:
                n *= s
            self.offsets[L.name + ".weight"] = (off, shp)
            off += _align8(n)
            for nm in ((".bias",) if L.kind == "head" else (".bn.gamma", ".bn.beta")):
                self.offsets[L.name + nm] = (off, (L.cout,))
                off += _align8(L.cout)
        self.n_params = off
        self._views: Dict[tuple, torch.Tensor] = {}
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.work = torch.zeros(off, dtype=BF16, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.opt_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.opt_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_count = 0
        bn_layers = [L for L in self.layers if L.kind != "head"]
        self.bn_mean = {L.name: torch.zeros(L.cout, device=dev) for L in bn_layers}
        self.bn_var = {L.name: torch.ones(L.cout, device=dev) for L in bn_layers}
        self.bn_buf = {L.name: torch.empty(4, L.cout, dtype=torch.float32, device=dev)
                       for L in bn_layers}
        self.bn_rows = {k: tuple(v.unbind(0)) for k, v in self.bn_buf.items()}
        self.scale: Dict[str, torch.Tensor] = {}
        self.shift: Dict[str, torch.Tensor] = {}
        # re-packed weights for the backward-data passes (the stem has none: images need no gradient)
        self.wt: Dict[str, torch.Tensor] = {}
        self.masks: Dict[str, torch.Tensor] = {}
        for L in self.layers:
            if L.kind == "conv":
                shp = pnn.bwd_data_weight_shape(L.cin, L.cout, L.k, L.stride, L.pad)
                self.wt[L.name] = torch.zeros(shp, dtype=BF16, device=dev)
            elif L.kind == "head":
                self.wt[L.name] = torch.zeros(L.cin, 1, L.kpad, dtype=BF16, device=dev)
            m = L.param_mask()
            if m is not None:
                self.masks[L.name] = m.to(dev)
        self._tplan: Optional[pnn.TransposePlan] = None
        self._ws: Dict[int, torch.Tensor] = {}
        self.tape: list = []
        self.training = False
        self.bucketer: Optional[GradBucketer] = None
        self._done_names: set = set()
        self._next_layer = len(self.layers) - 1
        # weight gradients depend only on a layer's input and its output gradient and nothing
        # reads them before the optimizer: they run beside the backward-data chain
        self.wgrad_stream: Optional[torch.cuda.Stream] = torch.cuda.Stream(device=dev)
        self._keep: list = []
        # backward-data GEMMs mask their result and reduce it for the batch norm below them
        # whenever that layer's output has no other reader (PE_FUSE_BN_BWD=0: separate passes)
        self.fuse_bn_bwd = os.environ.get("PE_FUSE_BN_BWD", "1") != "0"
        # ... and for residual layers whenever the last reader of their output is such a GEMM
        # (PE_FUSE_RES_BWD=0: the separate reduction pass)
        self.fuse_res_bwd = self.fuse_bn_bwd and os.environ.get("PE_FUSE_RES_BWD", "1") != "0"
        self.freeze_bn = False
        # PE_BN_FUSED=1: finalize + element-wise pass of every batch norm as ONE launch (an
        # in-launch hand-off between workgroups; the hand-off buffer belongs to the main stream,
        # on which all of these calls are issued).  Bit-identical results and a launch less per
        # layer and direction, but measured no faster: the fold's latency is paid inside the
        # launch instead of before it (ResNet-50 step 9.25 vs 8.93 ms, HRNet 37.8 vs 37.2 ms,
        # MobileNetV2 16.13 vs 16.17 ms), so the separate finalize launch stays the default.
        self.bn_sync = pnn.bn_sync(dev) if os.environ.get("PE_BN_FUSED", "0") == "1" else None
        self._syncs: Dict[int, torch.Tensor] = {}
        # ... for GEMMs whose reduction is at least this long.  While the masked 128 x 128 tile
        # still spilled registers a short reduction could not carry the heavier epilogue and the
        # threshold stood at 256; since it runs three workgroups per CU, fusing everywhere
        # measures faster (16 against 256: HRNet-W32 32.33 against 32.64 ms, MobileNetV2 15.58
        # against 15.80 ms)
        self.fuse_min_k = int(os.environ.get("PE_FUSE_MIN_K", "16"))
        self._pending_wg: list = []
        # set by a network that runs branches of its forward pass on streams of their own
        self.multi_stream = False
        self._step_graphs: Dict[tuple, "StepGraph"] = {}
        self._recorded: Dict[tuple, "replay.RecordedPass"] = {}

    # ------------------------------------------------------------------ parameters
    def view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        key = (arena.data_ptr(), arena.dtype, name)
        v = self._views.get(key)
        if v is None:
            off, shp = self.offsets[name]
            n = 1
            for s in shp:
                n *= s
            v = arena[off:off + n].view(shp)
            self._views[key] = v
        return v

    def w(self, L: Layer) -> torch.Tensor:
        return self.view(self.work, L.name + ".weight")

    def _g(self, name: str) -> torch.Tensor:
        return self.view(self.grads, name)

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Framework-layout state dict ([Cout,Cin,k,k] weights, .bn.gamma/.bn.beta/.bn.mean/
        .bn.var vectors, .bias for heads) -> arenas."""
        for L in self.layers:
            self.view(self.master, L.name + ".weight").copy_(
                L.pack(state[L.name + ".weight"]).to(self.device))
            if L.kind == "head":
                self.view(self.master, L.name + ".bias").copy_(state[L.name + ".bias"])
            else:
                self.view(self.master, L.name + ".bn.gamma").copy_(state[L.name + ".bn.gamma"])
                self.view(self.master, L.name + ".bn.beta").copy_(state[L.name + ".bn.beta"])
                self.bn_mean[L.name].copy_(state[L.name + ".bn.mean"])
                self.bn_var[L.name].copy_(state[L.name + ".bn.var"])
        self.work.copy_(self.master)  # one rounding to bf16
        self.fold_bn()
        self.refresh_transposes()

    def fold_bn(self) -> None:
        """Inference epilogue constants from the running statistics: every layer in one launch,
        written in place."""
        if getattr(self, "_fold_plan", None) is None:
            plan = pnn.BnFoldPlan(self.device)
            for L in self.layers:
                if L.kind == "head":
                    continue
                C = self.bn_var[L.name].numel()
                self.scale[L.name] = torch.empty(C, dtype=torch.float32, device=self.device)
                self.shift[L.name] = torch.empty(C, dtype=torch.float32, device=self.device)
                plan.add(self.view(self.master, L.name + ".bn.gamma"),
                         self.view(self.master, L.name + ".bn.beta"), self.bn_mean[L.name],
                         self.bn_var[L.name], self.scale[L.name], self.shift[L.name])
            self._fold_plan = plan
        self._fold_plan.run(BN_EPS)

    def freeze_batch_norm(self, on: bool = True) -> None:
        """Frozen statistics: every batch norm normalises with its running mean and variance,
        which no longer move, and backward treats them as constants (gamma and beta still
        train).  With them N ranks x B/N images give the gradient of one rank x B images."""
        self.freeze_bn = bool(on)

    def _refresh_frozen_rows(self) -> None:
        if getattr(self, "_freeze_plan", None) is None:
            plan = pnn.BnFoldPlan(self.device)
            for L in self.layers:
                if L.kind == "head":
                    continue
                r = self.bn_rows[L.name]
                plan.add(self.view(self.master, L.name + ".bn.gamma"),
                         self.view(self.master, L.name + ".bn.beta"), self.bn_mean[L.name],
                         self.bn_var[L.name], r[2], r[3], mean=r[0], rstd=r[1])
            self._freeze_plan = plan
        self._freeze_plan.run(BN_EPS)

    def refresh_transposes(self) -> None:
        """Re-packs every weight for the backward-data GEMMs (one launch for all layers)."""
        if self._tplan is None:
            plan = pnn.TransposePlan(self.device)
            for L in self.layers:
                if L.kind == "conv":
                    pnn.plan_bwd_data_weight(plan, self.w(L), self.wt[L.name], L.cin, L.cout, L.k,
                                             L.stride, L.pad)
                elif L.kind == "head":
                    pnn.plan_bwd_data_weight(plan, self.w(L), self.wt[L.name], L.cin, L.cout, 1, 1, 0)
            self._tplan = plan
        self._tplan.run()

    def enable_data_parallel(self, bucket_bytes: int = 32 << 20, group=None, comm=None,
                             transport: str = "f32", exchange_when_alone: bool = False) -> None:
        """Call after torch.distributed is initialised (or pass comm = a parallel.NativeComm to
        exchange gradients through the library's own RCCL communicator); parameters must already
        be identical on every rank (same seed / same loaded state, or broadcast_parameters()).
        transport "bf16" halves the bytes on the links (gradients summed in bf16)."""
        self.bucketer = GradBucketer(self.grads, self.layer_starts, bucket_bytes, group,
                                     comm=comm, transport=transport,
                                     exchange_when_alone=exchange_when_alone)

    def broadcast_parameters(self, src: int = 0) -> None:
        import torch.distributed as dist

        dist.broadcast(self.master, src)
        for t in list(self.bn_mean.values()) + list(self.bn_var.values()):
            dist.broadcast(t, src)
        self.work.copy_(self.master)
        self.refresh_transposes()

    def _sync(self) -> Optional[torch.Tensor]:
        """The one-launch batch norm's hand-off buffer of the current stream (launches that share
        a buffer must be issued from one stream), or None when that variant is off."""
        if self.bn_sync is None:
            return None
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self._syncs.get(key)
        if buf is None:
            buf = self._syncs[key] = pnn.bn_sync(self.device)
        return buf

    def _workspace_for(self, M: int, C: int) -> torch.Tensor:
        """Scratch for batch-norm partial sums: one buffer per stream, since a producer's rows
        must survive until the finalize behind it on the same stream has read them and branches
        on streams of their own run side by side."""
        need = pnn.lib().pe_bn_workspace_bytes(M, C)
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(max(need, 1 << 24), dtype=torch.uint8,
                                             device=self.device)
        return ws

    # ------------------------------------------------------------------ forward layers
    def begin(self, training: bool) -> None:
        self.training = training
        self.tape = []
        if training and self.freeze_bn:
            self._refresh_frozen_rows()

    def unit(self, L: Layer, x: Var, residual: Optional[Var] = None,
             relu: Optional[int] = None) -> Var:
        """act( bn(conv(x)) + residual ); `relu` overrides the layer's own activation."""
        act = L.relu if relu is None else int(relu)
        kw = {"kw": L.k} if L.narrow else {}
        res_t = residual.t if residual is not None else None
        if not self.training:
            if L.kind == "stem3":
                y = pnn.stem3x3s2(x.t, self.w(L), self.scale[L.name], self.shift[L.name], relu=act)
            elif L.kind == "dw":
                y = pnn.depthwise3x3(x.t, self.w(L), self.scale[L.name], self.shift[L.name],
                                     L.stride, L.dilation, act)
                if residual is not None:
                    raise ValueError("a depthwise layer takes no residual")
            else:
                y = pnn.conv2d(x.t, self.w(L), self.scale[L.name], self.shift[L.name], res_t,
                               L.stride, L.pad, act, **kw)
            return Var(y, False)
        B, H, W, _ = x.t.shape
        x.consumers += 1
        if residual is not None:
            residual.consumers += 1
        gamma = self.view(self.master, L.name + ".bn.gamma")
        beta = self.view(self.master, L.name + ".bn.beta")
        if L.kind == "dw":
            # bandwidth-bound conv without an epilogue, then one reduction pass for the statistics
            z = pnn.depthwise3x3(x.t, self.w(L), None, None, L.stride, L.dilation, relu=0)
            M = z.numel() // L.cout
            if self.freeze_bn:
                r = self.bn_rows[L.name]
                out = Var(pnn.bn_apply(z, r[2], r[3], res_t, act))
            elif self.bn_sync is not None:
                out = Var(pnn.bn_stats_apply(z, gamma, beta, BN_EPS, self.bn_momentum,
                                             self.bn_mean[L.name], self.bn_var[L.name],
                                             self._workspace_for(M, L.cout),
                                             self.bn_rows[L.name], self._sync(), res_t, act))
            else:
                _, _, scale, shift = pnn.bn_stats(z, gamma, beta, BN_EPS, self.bn_momentum,
                                                  self.bn_mean[L.name], self.bn_var[L.name],
                                                  self._workspace_for(M, L.cout),
                                                  out=self.bn_buf[L.name])
                out = Var(pnn.bn_apply(z, scale, shift, res_t, act))
            if act and residual is None:
                out.prod = (L, z, act)
            self.tape.append(("unit", L, x, residual, act, z, out))
            return out
        if L.kind == "stem3":
            M = B * H * W
            ws = self._workspace_for(M, L.cout)
            z, rows = pnn.stem3x3s2_bnstats(x.t, self.w(L), ws)
        else:
            Ho = (H + 2 * L.pad - L.k) // L.stride + 1
            Wo = (W + 2 * L.pad - L.k) // L.stride + 1
            M = B * Ho * Wo
            ws = self._workspace_for(M, L.cout)
            z, rows = pnn.conv2d_bnstats(x.t, self.w(L), ws, L.stride, L.pad, **kw)
        if self.freeze_bn:
            r = self.bn_rows[L.name]
            out = Var(pnn.bn_apply(z, r[2], r[3], res_t, act))
        elif self.bn_sync is not None:
            out = Var(pnn.bn_finalize_apply(ws, rows, z, gamma, beta, BN_EPS,
                                            self.bn_momentum, self.bn_mean[L.name],
                                            self.bn_var[L.name], self.bn_rows[L.name],
                                            self._sync(), res_t, act))
        else:
            _, _, scale, shift = pnn.bn_finalize(
                ws, rows, M, L.cout, gamma, beta, BN_EPS, self.bn_momentum,
                self.bn_mean[L.name], self.bn_var[L.name], out=self.bn_rows[L.name])
            out = Var(pnn.bn_apply(z, scale, shift, res_t, act))
        if act and residual is None:
            out.prod = (L, z, act)
        elif act == 1 and residual is not None:
            out.prod_res = (L, z)
        self.tape.append(("unit", L, x, residual, act, z, out))
        return out

    def fuse(self, base: Var, lows: Sequence[Var], log2_factors: Sequence[int],
             relu: bool = True) -> Var:
        """relu?( base + sum of nearest-upsampled lows )."""
        y = pnn.fuse_upsample_add(base.t, [v.t for v in lows], list(log2_factors), relu=relu)
        out = Var(y, self.training)
        if self.training:
            base.consumers += 1
            for v in lows:
                v.consumers += 1
            self.tape.append(("fuse", base, list(lows), list(log2_factors), relu, out))
        return out

    def head(self, L: Layer, x: Var) -> torch.Tensor:
        """1x1 conv + bias to float32 NCHW maps."""
        kw = {"kw": 1} if L.narrow else {}
        y = pnn.conv2d(x.t, self.w(L), None, self.view(self.master, L.name + ".bias"),
                       out_f32_nchw=True, **kw)
        if self.training:
            x.consumers += 1
            self.tape.append(("head", L, x))
        return y

    # ------------------------------------------------------------------ backward
    @contextmanager
    def _beside(self, *tensors: torch.Tensor):
        """Launches inside run on the weight-gradient stream, after everything queued on the
        current stream so far.  `tensors` are what those launches read: they are kept alive until
        backward has joined the two streams again, so the allocator cannot hand their memory to
        a later launch on the main stream while the side stream still reads it (this also holds
        under HIP-graph capture, where per-tensor stream bookkeeping is not available)."""
        side = self.wgrad_stream
        if side is None:
            yield
            return
        replay.wait_stream(side, torch.cuda.current_stream(self.device))
        self._keep.extend(tensors)
        with torch.cuda.stream(side):
            yield

    def _done(self, L: Layer) -> None:
        """Marks a layer's gradients as enqueued.  Buckets are announced in descending arena
        order whatever order a branched graph finishes its layers in."""
        if self.bucketer is None:
            return
        self._done_names.add(L.name)
        while self._next_layer >= 0 and self.layers[self._next_layer].name in self._done_names:
            self.bucketer.layer_done(self.layer_starts[self._next_layer])
            self._next_layer -= 1

    def _acc(self, v: Var, g: torch.Tensor) -> None:
        if not v.needs_grad:
            return
        v.grad = g if v.grad is None else pnn.add_bf16(v.grad, g)
        v.ngrad += 1

    WGRAD_BATCH = 8   # weight gradients handed to the side stream per fork
    WGRAD_TAIL = int(os.environ.get("PE_WGRAD_TAIL", "10"))   # tape records that count as the end

    def _weight_grad(self, L: Layer, x: torch.Tensor, dz: torch.Tensor, cout=None) -> None:
        """Queues a layer's weight gradient.  Forking to the side stream costs the host an event
        record, a wait and two stream switches; done per layer that is a third of the time it
        takes to enqueue a step of a 300-layer network, so the forks are batched."""
        self._pending_wg.append((L, x, dz, cout))
        # the last layers of the pass fork in pairs: whatever is still queued when the main chain
        # ends runs behind it, with nothing beside it
        limit = self.WGRAD_BATCH if len(self.tape) >= self.WGRAD_TAIL else 2
        if len(self._pending_wg) >= limit:
            self._flush_weight_grads()

    def _flush_weight_grads(self) -> None:
        if not self._pending_wg:
            return
        pending, self._pending_wg = self._pending_wg, []
        with self._beside(*[t for (_, x, dz, _) in pending for t in (x, dz)]):
            # the implicit-GEMM layers of the batch share one launch (and one fold launch)
            convs = [(x, dz, self._g(L.name + ".weight"), L.k, L.stride, L.pad, cout)
                     for L, x, dz, cout in pending if L.kind not in ("stem3", "dw")]
            if convs:
                pnn.conv2d_bwd_weight_batched(convs)
            for L, x, dz, cout in pending:
                gw = self._g(L.name + ".weight")
                if L.kind == "stem3":
                    pnn.stem3x3s2_bwd_weight(x, dz, gw)
                elif L.kind == "dw":
                    pnn.depthwise3x3_bwd_weight(x, dz, gw, L.stride, L.dilation)
                m = self.masks.get(L.name)
                if m is not None:
                    pnn.mul_(gw, m)
                self._done(L)

    def _data_grad(self, L: Layer, x: Var, dz: torch.Tensor) -> None:
        if not x.needs_grad:
            return
        hw = tuple(x.t.shape[1:3])
        x.ngrad += 1
        if L.kind == "dw":
            dx = pnn.depthwise3x3_bwd_data(dz, self.w(L), hw, L.stride, L.dilation)
            x.grad = dx if x.grad is None else pnn.add_bf16(x.grad, dx)
            return
        if (self.fuse_res_bwd and x.prod_res is not None and x.ngrad == x.consumers
                and L.k * L.k * L.cout >= self.fuse_min_k
                and not (L.k == 1 and L.stride == 2)
                and hw[0] % L.stride == 0 and hw[1] % L.stride == 0):
            # x = relu(bn(z) + identity) and this is the last of its readers to be differentiated:
            # the GEMM adds what the others left, masks the sum with the stored activation and
            # reduces batch-norm backward's sums, so the residual layer below gets its dz (and
            # the identity path its masked gradient) without a reduction pass over the gradient
            Lb, zb = x.prod_res
            buf = self.bn_rows[Lb.name]
            ws = self._workspace_for(zb.numel() // Lb.cout, Lb.cout)
            g, rows = pnn.conv2d_bwd_data_resmask(dz, self.wt[L.name], hw, L.k, L.stride, L.pad,
                                                  x.grad, x.t, zb, ws)
            x.dz = pnn.bn_bwd_from_partials(
                g, zb, rows, self.view(self.master, Lb.name + ".bn.gamma"), buf[0], buf[1],
                self._g(Lb.name + ".bn.gamma"), self._g(Lb.name + ".bn.beta"), ws,
                frozen=self.freeze_bn, sync=self._sync())
            x.grad = g
            return
        if (self.fuse_bn_bwd and x.prod is not None and x.consumers == 1 and x.grad is None
                and L.k * L.k * L.cout >= self.fuse_min_k
                and not (L.k == 1 and L.stride == 2)
                and hw[0] % L.stride == 0 and hw[1] % L.stride == 0):
            # x = act(bn(z)) read by this layer alone: the GEMM masks its result with the
            # activation rebuilt from z and reduces batch-norm backward's sums, so the layer
            # below gets its dz without reading the gradient again
            Lb, zb, actb = x.prod
            buf = self.bn_rows[Lb.name]
            ws = self._workspace_for(zb.numel() // Lb.cout, Lb.cout)
            g, rows = pnn.conv2d_bwd_data_bnmask(dz, self.wt[L.name], hw, L.k, L.stride, L.pad,
                                                 zb, buf[2], buf[3], ws, relu=actb)
            x.dz = pnn.bn_bwd_from_partials(
                g, zb, rows, self.view(self.master, Lb.name + ".bn.gamma"), buf[0], buf[1],
                self._g(Lb.name + ".bn.gamma"), self._g(Lb.name + ".bn.beta"), ws,
                frozen=self.freeze_bn, sync=self._sync())
            x.grad = g
            return
        fused = x.grad is not None and not (L.k == 1 and L.stride == 2)
        dx = pnn.conv2d_bwd_data(dz, self.wt[L.name], hw, L.k, L.stride, L.pad,
                                 residual=x.grad if fused else None)
        if x.grad is not None and not fused:
            dx = pnn.add_bf16(x.grad, dx)
        x.grad = dx

    def backward(self, head_grads: Dict[str, torch.Tensor]) -> None:
        """head_grads: {head layer name: float32 [B,K,H,W] dLoss/dMaps}.  Fills self.grads."""
        self.grads.zero_()
        self._done_names = set()
        self._next_layer = len(self.layers) - 1
        # Backward stays on one stream (plus the weight-gradient stream) even when forward forked
        # its branches: with the weight gradients already filling the gaps beside the main chain,
        # running every record on the stream that made it measured no faster (34.4 against
        # 34.5 ms on the HRNet-W32 step) and slower once the streams outnumber the four
        # hardware queues (36.9 ms).
        while self.tape:
            rec = self.tape.pop()
            if rec[0] == "head":
                _, L, x = rec
                dyp = pnn.final_grad_prepare(head_grads[L.name], L.kpad, self._g(L.name + ".bias"))
                self._weight_grad(L, x.t, dyp, cout=L.cout)
                if x.needs_grad:
                    hw = tuple(x.t.shape[1:3])
                    x.grad = pnn.conv2d_bwd_data(dyp, self.wt[L.name], hw, 1, 1, 0,
                                                 residual=x.grad)
                    x.ngrad += 1
            elif rec[0] == "fuse":
                _, base, lows, facs, relu, out = rec
                if out.grad is None:
                    raise RuntimeError("a fusion output received no gradient")
                dbase, dlows = pnn.fuse_upsample_add_bwd(out.grad, out.t if relu else None, facs)
                self._acc(base, dbase)
                for v, g in zip(lows, dlows):
                    self._acc(v, g)
                out.grad = None
            else:
                _, L, x, residual, act, z, out = rec
                if out.grad is None:
                    raise RuntimeError(f"layer {L.name}: output received no gradient")
                if out.dz is not None:       # made by the consumer's backward-data GEMM
                    dz = out.dz
                    out.dz = None
                    if residual is not None:     # out.grad is the masked gradient here
                        self._acc(residual, out.grad)
                    out.grad = None
                    self._weight_grad(L, x.t, dz)
                    if L.kind != "stem3":
                        self._data_grad(L, x, dz)
                    continue
                buf = self.bn_rows[L.name]
                masked = residual is not None and act != 0
                # without a residual the activation mask is rebuilt from z (scale and shift of the
                # forward pass are still in the layer's rows): one tensor read less per pass
                rebuild = act != 0 and residual is None
                dz, g = pnn.bn_bwd(out.grad, out.t if (act and not rebuild) else None, z,
                                   self.view(self.master, L.name + ".bn.gamma"), buf[0], buf[1],
                                   self._g(L.name + ".bn.gamma"), self._g(L.name + ".bn.beta"),
                                   self._workspace_for(z.numel() // L.cout, L.cout), masked,
                                   relu=act or 1, act_scale=buf[2] if rebuild else None,
                                   act_shift=buf[3] if rebuild else None, frozen=self.freeze_bn,
                                   sync=self._sync())
                if residual is not None:
                    self._acc(residual, g if masked else out.grad)
                self._weight_grad(L, x.t, dz)
                if L.kind != "stem3":
                    self._data_grad(L, x, dz)
                out.grad = None
        self._flush_weight_grads()
        if self.wgrad_stream is not None:   # the optimizer reads every gradient
            replay.wait_stream(torch.cuda.current_stream(self.device), self.wgrad_stream)
        self._keep.clear()

    # ------------------------------------------------------------------ HIP-graph replay
    def replay(self, fn, *inputs: Optional[torch.Tensor]):
        """Runs fn(*inputs) (a forward + loss + backward pass made of launches on the current
        stream and the weight-gradient stream) as one HIP graph: captured the first time a given
        set of input shapes is seen, replayed afterwards with the inputs copied into the
        buffers it was captured on.  A branched network issues thousands of small launches per
        step and the host cannot enqueue them as fast as the GPU runs them; a replay is one
        launch.  Returns fn's outputs, which live in the graph's own memory and are overwritten
        by the next replay.  Not available with a gradient exchange attached (the collectives
        are issued by the host)."""
        if self.bucketer is not None:
            raise RuntimeError("HIP-graph replay and the bucketed gradient exchange are exclusive")
        key = (fn.__name__,) + tuple(None if t is None else (tuple(t.shape), t.dtype) for t in inputs)
        g = self._step_graphs.get(key)
        if g is None:
            g = self._step_graphs[key] = StepGraph(self, fn, inputs)
        return g(*inputs)

    def replay_host(self, fn, *inputs: Optional[torch.Tensor]):
        """Runs fn(*inputs) from a recorded call list (replay.RecordedPass): the launches of the
        first call are kept by the library and enqueued again without the Python walk that
        produced them.  Same contract as replay(): static shapes, outputs overwritten by the
        next call, not with a gradient exchange attached; and not with the one-launch batch
        norm, whose hand-off counter is an argument that moves from launch to launch."""
        if self.bucketer is not None:
            raise RuntimeError("host replay and the bucketed gradient exchange are exclusive")
        if self.bn_sync is not None:
            raise RuntimeError("host replay and PE_BN_FUSED=1 are exclusive")
        return replay.run_recorded(self._recorded, self, fn, inputs)

    # ------------------------------------------------------------------ optimizer
    def optimizer_step(self, lr: float, optimizer: str = "adam", beta1: float = 0.9,
                       beta2: float = 0.999, eps: float = 1e-8, momentum: float = 0.9) -> None:
        scale = 1.0
        if self.bucketer is not None:
            self.bucketer.finish()
            scale = self.bucketer.grad_scale
        self.step_count += 1
        if optimizer == "adam":
            ops.adam_step(self.master, self.opt_m, self.opt_v, self.grads, self.step_count, lr,
                          beta1, beta2, eps, p_bf16=self.work, tf1=True, grad_scale=scale)
        elif optimizer == "sgd":
            ops.sgd_step(self.master, self.opt_m, self.grads, lr, momentum, p_bf16=self.work,
                         grad_scale=scale)
        else:
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.refresh_transposes()


class StepGraph:
    """One captured pass of a TrainGraph (see TrainGraph.replay)."""

    def __init__(self, net: TrainGraph, fn, example_inputs):
        dev = net.device
        self.static_in = [None if t is None else t.clone() for t in example_inputs]
        # One pass outside capture: one-time set-up (workspace growth, kernel attributes, the
        # allocator's pools) must not happen while capturing.  Running statistics stay frozen.
        momentum, net.bn_momentum = net.bn_momentum, 0.0
        cur = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            fn(*self.static_in)
        cur.wait_stream(warm)
        net.bn_momentum = momentum
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        return self.static_out
