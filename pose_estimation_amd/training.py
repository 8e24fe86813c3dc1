"""Training graph of the pose network on the HIP kernels: forward with batch statistics,
fused target-render + masked-MSE loss, hand-written backward, bucketed data-parallel gradient
exchange, fused Adam / SGD over the flat arenas.

The reference's training loop (train.py / model/base_model.py, PoseEstimation.pyproj:30,34) is
absent; its visible contract is `Trainer(config)`, Adam(0.9, 0.999, 1e-8), lr 1e-3 with
exponential decay 0.95 / 10 000 steps (test.ipynb:400, 1001-1008) and Horovod-style data
parallelism (test.ipynb:12): local batch statistics, summed gradients averaged over ranks.
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch

from . import nn as pnn
from . import ops
from . import replay
from .model import BF16, BN_EPS, ConvSpec, PoseResNet
from .parallel import GradBucketer

FINAL_KPAD = 64  # the K-joint loss gradient is zero-padded to this many channels for the MFMA passes


class TrainablePoseResNet(PoseResNet):
    def __init__(self, device, num_joints: int = 17, layers=(3, 4, 6, 3), width: int = 64,
                 head_ch: int = 256, bn_momentum: float = 0.1, sigma: float = 2.0,
                 heatmap_stride: float = 4.0):
        super().__init__(device, num_joints, layers, width, head_ch)
        dev = self.device
        self.bn_momentum = bn_momentum
        self.sigma = sigma
        self.radius = int(round(3 * sigma))
        self.heatmap_stride = heatmap_stride
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.opt_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.opt_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.step_count = 0
        # transposed weights for the backward-data passes
        self.wt: Dict[str, torch.Tensor] = {}
        for c in self.convs:
            if c.kind == "conv":
                self.wt[c.name] = torch.zeros(c.cin, c.k * c.k, c.cout, dtype=BF16, device=dev)
            elif c.kind == "deconv":
                self.wt[c.name] = torch.zeros(c.cin, 16, c.cout, dtype=BF16, device=dev)
            elif c.kind == "final":
                self.wt[c.name] = torch.zeros(c.cin, 1, FINAL_KPAD, dtype=BF16, device=dev)
        self.bn_buf = {c.name: torch.empty(4, c.cout, dtype=torch.float32, device=dev)
                       for c in self.convs if c.bn}
        self.bn_rows = {k: tuple(v.unbind(0)) for k, v in self.bn_buf.items()}
        self._ws: Dict[int, torch.Tensor] = {}     # batch-norm workspace, one per stream
        self._tplan = None
        stem = self.specs[0]
        self.stem_mask = pnn.pack_stem_weight(torch.ones(stem.cout, 3, 7, 7)).float().to(dev)
        # layer start offsets in arena order (a layer = weight + its BN / bias vectors)
        self.layer_starts: List[int] = [self.offsets[c.name + ".weight"][0] for c in self.convs]
        self.bucketer: Optional[GradBucketer] = None
        self.saved: dict = {}
        self._wt_pending = False
        self._recorded: Dict[tuple, "replay.RecordedPass"] = {}
        self._keep: list = []
        self._pending_wg: list = []
        # the projection of a block's identity path runs beside the block's main chain: forward
        # on the weight-gradient stream (idle then), backward on a stream of its own
        self.down_beside = os.environ.get("PE_TRAIN_DOWN_BESIDE", "1") != "0"
        self.down_stream = torch.cuda.Stream(device=dev)
        self._down_unjoined = False
        # backward-data GEMMs mask their result and reduce it for the batch norm below them
        # (PE_FUSE_BN_BWD=0: the separate reduction pass, for comparison)
        self.fuse_bn_bwd = os.environ.get("PE_FUSE_BN_BWD", "1") != "0"
        # ... and the residual layers: conv1's backward-data GEMM of a block makes the whole
        # gradient of the block input, masks it with the stored activation and reduces it for the
        # conv3 batch norm of the block below (PE_FUSE_RES_BWD=0: the separate reduction pass)
        self.fuse_res_bwd = self.fuse_bn_bwd and os.environ.get("PE_FUSE_RES_BWD", "1") != "0"
        # Weight gradients run on their own stream: they depend only on a layer's input and its
        # output gradient, nothing downstream reads them before the optimizer, and at batch 64
        # many backward kernels leave CUs idle, so they fill in beside the backward-data chain.
        self.wgrad_stream: Optional[torch.cuda.Stream] = torch.cuda.Stream(device=dev)
        self.freeze_bn = False
        self._freeze_plan = None
        # PE_BN_FUSED=1: finalize + element-wise pass of every batch norm as ONE launch (an
        # in-launch hand-off between workgroups; the hand-off buffer belongs to the main stream,
        # on which all of these calls are issued).  Bit-identical results and a launch less per
        # layer and direction, but measured no faster: the fold's latency is paid inside the
        # launch instead of before it (ResNet-50 step 9.25 vs 8.93 ms, HRNet 37.8 vs 37.2 ms,
        # MobileNetV2 16.13 vs 16.17 ms), so the separate finalize launch stays the default.
        self.bn_sync = pnn.bn_sync(dev) if os.environ.get("PE_BN_FUSED", "0") == "1" else None

    # ------------------------------------------------------------------ set-up
    def load_state(self, state) -> None:
        super().load_state(state)
        self.refresh_transposes()

    def enable_data_parallel(self, bucket_bytes: int = 32 << 20, group=None, comm=None,
                             transport: str = "f32", exchange_when_alone: bool = False) -> None:
        """Call after torch.distributed is initialised (or pass comm = a parallel.NativeComm to
        exchange gradients through the library's own RCCL communicator); parameters must already
        be identical on every rank (same seed / same loaded state, or broadcast_parameters()).
        transport "bf16" halves the bytes on the links (gradients summed in bf16)."""
        self.bucketer = GradBucketer(self.grads, self.layer_starts, bucket_bytes, group,
                                     comm=comm, transport=transport,
                                     exchange_when_alone=exchange_when_alone)

    def freeze_batch_norm(self, on: bool = True) -> None:
        """Frozen statistics (fine-tuning; also what makes N ranks x B/N images equal one rank
        x B images exactly): every batch norm normalises with its RUNNING mean and variance,
        which no longer move, and backward treats them as constants.  gamma and beta still
        train.  The per-layer rows (mean, rstd, scale, shift) are refreshed from the current
        parameters at the start of every forward pass, all layers in one launch."""
        self.freeze_bn = bool(on)

    def _refresh_frozen_rows(self) -> None:
        if self._freeze_plan is None:
            plan = pnn.BnFoldPlan(self.device)
            for c in self.convs:
                if c.bn:
                    r = self.bn_rows[c.name]
                    plan.add(self._gamma(c), self._beta(c), self.bn_mean[c.name],
                             self.bn_var[c.name], r[2], r[3], mean=r[0], rstd=r[1])
            self._freeze_plan = plan
        self._freeze_plan.run(BN_EPS)

    def broadcast_parameters(self, src: int = 0) -> None:
        import torch.distributed as dist

        dist.broadcast(self.master, src)
        for t in list(self.bn_mean.values()) + list(self.bn_var.values()):
            dist.broadcast(t, src)
        self.work.copy_(self.master)
        self.refresh_transposes()

    def _join_transposes(self) -> None:
        """The re-pack a training step left running on the side stream is done before anything
        on the current stream touches the backward-data weights."""
        if self._wt_pending or replay.recording():   # a recorded pass always waits
            replay.wait_stream(torch.cuda.current_stream(self.device), self.wgrad_stream)
            self._wt_pending = False

    def refresh_transposes(self) -> None:
        """Re-packs every weight for the backward-data GEMMs (one launch for all layers)."""
        self._join_transposes()
        if self._tplan is None:
            plan = pnn.TransposePlan(self.work.device)
            for c in self.convs:
                w = self.w(c)
                if c.kind == "conv":
                    plan.add(w.view(c.cout, c.k * c.k, c.cin), self.wt[c.name])
                elif c.kind == "deconv":
                    for ph in range(4):
                        plan.add(w[ph], self.wt[c.name], t0=4 * ph)
                elif c.kind == "final":
                    plan.add(w.view(c.cout, 1, c.cin), self.wt[c.name])
            self._tplan = plan
        self._tplan.run()

    def _tplan_run(self) -> None:
        if self._tplan is None:
            self.refresh_transposes()
        else:
            self._tplan.run()

    def _workspace(self, z: torch.Tensor) -> torch.Tensor:
        C = z.shape[3]
        return self._workspace_for(z.numel() // C, C)

    def _workspace_for(self, M: int, C: int) -> torch.Tensor:
        """Scratch for batch-norm partial sums, one buffer per stream: a conv's rows must survive
        until the finalize behind it on the same stream has read them."""
        need = pnn.lib().pe_bn_workspace_bytes(M, C)
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(max(need, 1 << 24), dtype=torch.uint8,
                                             device=self.device)
        return ws

    def _gamma(self, c: ConvSpec):
        return self.view(self.master, c.name + ".bn.gamma")

    def _beta(self, c: ConvSpec):
        return self.view(self.master, c.name + ".bn.beta")

    def _g(self, name: str) -> torch.Tensor:
        return self.view(self.grads, name)

    # ------------------------------------------------------------------ forward (batch stats)
    def _bn(self, c: ConvSpec, z: torch.Tensor, rows: int, residual=None, relu=True,
            ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Normalises z with the batch statistics its conv left in the workspace `ws` as `rows`
        partial rows.  The workspace is shared by the stream's layers, so this must run before
        the next conv on the same stream."""
        C = z.shape[3]
        if ws is None:
            ws = self._workspace_for(z.numel() // C, C)
        if self.freeze_bn:     # rows hold the running statistics (see _refresh_frozen_rows)
            r = self.bn_rows[c.name]
            return pnn.bn_apply(z, r[2], r[3], residual, relu)
        if self.bn_sync is not None:
            return pnn.bn_finalize_apply(ws, rows, z, self._gamma(c), self._beta(c), BN_EPS,
                                         self.bn_momentum, self.bn_mean[c.name],
                                         self.bn_var[c.name], self.bn_rows[c.name], self.bn_sync,
                                         residual, relu)
        _, _, scale, shift = pnn.bn_finalize(ws, rows, z.numel() // C, C, self._gamma(c),
                                             self._beta(c), BN_EPS, self.bn_momentum,
                                             self.bn_mean[c.name], self.bn_var[c.name],
                                             out=self.bn_rows[c.name])
        return pnn.bn_apply(z, scale, shift, residual, relu)

    def _conv_bn(self, c: ConvSpec, x: torch.Tensor, stride=1, pad=0, residual=None, relu=True):
        B, H, W, _ = x.shape
        k = c.k
        M = B * ((H + 2 * pad - k) // stride + 1) * ((W + 2 * pad - k) // stride + 1)
        ws = self._workspace_for(M, c.cout)
        z, rows = pnn.conv2d_bnstats(x, self.w(c), ws, stride, pad)
        return z, self._bn(c, z, rows, residual, relu, ws)

    def block_forward(self, b, x: torch.Tensor) -> dict:
        """One bottleneck block with batch statistics; returns everything backward needs."""
        zd = None
        side = self.wgrad_stream if (b.down and self.down_beside and self.bn_sync is None) else None
        if side is not None:
            # the projection of the identity path depends on the block input alone: it runs on the
            # weight-gradient stream (idle during forward) beside conv1 and conv2
            main = torch.cuda.current_stream(self.device)
            replay.wait_stream(side, main)
            with torch.cuda.stream(side):
                zd, idn = self._conv_bn(b.down, x, stride=b.down.stride, relu=False)
        z1, a1 = self._conv_bn(b.c1, x)
        z2, a2 = self._conv_bn(b.c2, a1, stride=b.c2.stride, pad=1)
        if side is not None:
            replay.wait_stream(main, side)
        elif b.down:
            zd, idn = self._conv_bn(b.down, x, stride=b.down.stride, relu=False)
        else:
            idn = x
        z3, out = self._conv_bn(b.c3, a2, residual=idn, relu=True)
        return dict(x=x, z1=z1, a1=a1, z2=z2, a2=a2, z3=z3, zd=zd, out=out)

    def block_backward(self, b, s: dict, dout, flush: bool = True, below=None):
        """Backward of block_forward: accumulates the block's parameter gradients into the
        gradient arena and returns the gradient with respect to the block input.  flush=False
        leaves the block's weight gradients queued for a batched launch with later layers'.

        below = (conv3 spec, z3) of the block whose output is this block's input: the block's last
        backward-data GEMM then makes that output's whole gradient (its own result plus the
        identity path's), masks it with the stored activation and leaves conv3's batch-norm sums
        behind; the return value is (g, rows) and is what the block below takes as `dout`."""
        self._join_transposes()
        x = s["x"]
        hw_in = tuple(x.shape[1:3])
        hw_mid = tuple(s["a1"].shape[1:3])
        if isinstance(dout, tuple):      # masked and summed by the block above (see `below`)
            g3, rows3 = dout
            buf = self.bn_rows[b.c3.name]
            dz3 = pnn.bn_bwd_from_partials(g3, s["z3"], rows3, self._gamma(b.c3), buf[0], buf[1],
                                           self._g(b.c3.name + ".bn.gamma"),
                                           self._g(b.c3.name + ".bn.beta"),
                                           self._workspace(s["z3"]), frozen=self.freeze_bn,
                                           sync=self.bn_sync)
        else:
            dz3, g3 = self._bn_bwd(b.c3, dout, s["out"], s["z3"], want_masked=True)
        dside = self.down_stream if (b.down and self.down_beside and self.bn_sync is None) else None
        if dside is not None:
            # the projection's backward needs g3 alone and is needed by the block's last GEMM
            main = torch.cuda.current_stream(self.device)
            replay.wait_stream(dside, main)
            with torch.cuda.stream(dside):
                dzd, _ = self._bn_bwd(b.down, g3, None, s["zd"])
                res = pnn.conv2d_bwd_data(dzd, self.wt[b.down.name], hw_in, 1, b.down.stride, 0)
            self._down_unjoined = True     # weight-gradient launches wait for it (see _beside)
            self._weight_grad(b.down, x, dzd, 1, b.down.stride, 0)
            self._weight_grad(b.c3, s["a2"], dz3, 1, 1, 0)
        elif b.down:
            dzd, _ = self._bn_bwd(b.down, g3, None, s["zd"])
            # the arena holds `down` after conv3, so it is finished before conv3 is marked
            self._weight_grad(b.down, x, dzd, 1, b.down.stride, 0)
            self._weight_grad(b.c3, s["a2"], dz3, 1, 1, 0)
            res = pnn.conv2d_bwd_data(dzd, self.wt[b.down.name], hw_in, 1, b.down.stride, 0)
        else:
            self._weight_grad(b.c3, s["a2"], dz3, 1, 1, 0)
            res = g3
        if self.fuse_bn_bwd:
            dz2 = self._bwd_data_bn(b.c2, s["z2"], dz3, self.wt[b.c3.name], 1, 1, 0)
        else:
            da2 = pnn.conv2d_bwd_data(dz3, self.wt[b.c3.name], tuple(s["a2"].shape[1:3]), 1, 1, 0)
            dz2, _ = self._bn_bwd(b.c2, da2, s["a2"], s["z2"], rebuild=True)
        self._weight_grad(b.c2, s["a1"], dz2, 3, b.c2.stride, 1)
        if self.fuse_bn_bwd and hw_mid[0] % b.c2.stride == 0 and hw_mid[1] % b.c2.stride == 0:
            dz1 = self._bwd_data_bn(b.c1, s["z1"], dz2, self.wt[b.c2.name], 3, b.c2.stride, 1)
        else:
            da1 = pnn.conv2d_bwd_data(dz2, self.wt[b.c2.name], hw_mid, 3, b.c2.stride, 1)
            dz1, _ = self._bn_bwd(b.c1, da1, s["a1"], s["z1"], rebuild=True)
        self._weight_grad(b.c1, x, dz1, 1, 1, 0)
        if flush:
            self._flush_weight_grads()
        if dside is not None:
            replay.wait_stream(main, dside)
            self._down_unjoined = False
        if below is not None:
            _, z3_below = below
            return pnn.conv2d_bwd_data_resmask(dz1, self.wt[b.c1.name], hw_in, 1, 1, 0, res, x,
                                               z3_below, self._workspace(z3_below))
        return pnn.conv2d_bwd_data(dz1, self.wt[b.c1.name], hw_in, 1, 1, 0, residual=res)

    def deconv_forward(self, d, x: torch.Tensor) -> dict:
        B, H, W, _ = x.shape
        z, rows = pnn.deconv4x4s2_bnstats(x, self.w(d), self._workspace_for(4 * B * H * W, d.cout))
        return dict(x=x, z=z, a=self._bn(d, z, rows))

    def deconv_backward(self, d, s: dict, da: torch.Tensor) -> torch.Tensor:
        self._join_transposes()
        dz, _ = self._bn_bwd(d, da, s["a"], s["z"], rebuild=True)
        self._flush_weight_grads()
        with self._beside(s["x"], dz):
            pnn.deconv4x4s2_bwd_weight(s["x"], dz, self._g(d.name + ".weight"))
            self._done(d)
        return pnn.deconv4x4s2_bwd_data(dz, self.wt[d.name])

    def forward_train(self, images_u8: Optional[torch.Tensor],
                      x_s2d: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward with batch statistics.  Give uint8 images at network resolution, or x_s2d, the
        normalised space-to-depth image an input pipeline (augment.augment_u8) already made."""
        stem, blocks, deconvs, final = self.specs
        sv: dict = {"blocks": [], "deconvs": []}
        xs = x_s2d if x_s2d is not None else pnn.preprocess_u8(images_u8)
        if self.freeze_bn:
            self._refresh_frozen_rows()
        z0, rows = pnn.stem7x7s2_bnstats(xs, self.w(stem),
                                         self._workspace_for(xs.numel() // 16, stem.cout))
        a0 = self._bn(stem, z0, rows)
        x, pool_idx = pnn.maxpool3x3s2(a0, want_index=True)
        sv.update(xs=xs, z0=z0, a0=a0, pool_idx=pool_idx)
        for b in blocks:
            s = self.block_forward(b, x)
            sv["blocks"].append(s)
            x = s["out"]
        for d in deconvs:
            s = self.deconv_forward(d, x)
            sv["deconvs"].append(s)
            x = s["a"]
        sv["feat"] = x
        hm = pnn.conv2d(x, self.w(final), None, self.view(self.master, final.name + ".bias"),
                        out_f32_nchw=True)
        self.saved = sv
        return hm

    # ------------------------------------------------------------------ backward
    def _bn_bwd(self, c: ConvSpec, da, a, z, want_masked=False, rebuild=False):
        """rebuild: the layer has a ReLU and no residual, so its mask is rebuilt from z with the
        scale and shift its forward pass left in the layer's rows, instead of reading `a`."""
        buf = self.bn_rows[c.name]
        return pnn.bn_bwd(da, None if rebuild else a, z, self._gamma(c), buf[0], buf[1],
                          self._g(c.name + ".bn.gamma"), self._g(c.name + ".bn.beta"),
                          self._workspace(z), want_masked,
                          act_scale=buf[2] if rebuild else None,
                          act_shift=buf[3] if rebuild else None, frozen=self.freeze_bn,
                          sync=self.bn_sync)

    def _bwd_data_bn(self, below: ConvSpec, z_below: torch.Tensor, dy: torch.Tensor,
                     wt: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
        """Backward-data of a conv whose input is the ReLU output of layer `below` (no residual,
        no other consumer), straight to dz of `below`: the GEMM masks its result with the
        activation rebuilt from z_below and leaves batch-norm backward's sums behind, so the
        gradient is not read again to reduce it."""
        buf = self.bn_rows[below.name]
        ws = self._workspace(z_below)
        g, rows = pnn.conv2d_bwd_data_bnmask(dy, wt, tuple(z_below.shape[1:3]), k, stride, pad,
                                             z_below, buf[2], buf[3], ws)
        return pnn.bn_bwd_from_partials(g, z_below, rows, self._gamma(below), buf[0], buf[1],
                                        self._g(below.name + ".bn.gamma"),
                                        self._g(below.name + ".bn.beta"), ws,
                                        frozen=self.freeze_bn, sync=self.bn_sync)

    def _deconv_bwd_data_bn(self, below: ConvSpec, z_below: torch.Tensor, dy: torch.Tensor,
                            wt: torch.Tensor) -> torch.Tensor:
        buf = self.bn_rows[below.name]
        ws = self._workspace(z_below)
        g, rows = pnn.deconv4x4s2_bwd_data_bnmask(dy, wt, z_below, buf[2], buf[3], ws)
        return pnn.bn_bwd_from_partials(g, z_below, rows, self._gamma(below), buf[0], buf[1],
                                        self._g(below.name + ".bn.gamma"),
                                        self._g(below.name + ".bn.beta"), ws,
                                        frozen=self.freeze_bn, sync=self.bn_sync)

    WGRAD_BATCH = 8   # conv weight gradients per batched launch (and per fork to the side stream)

    def _weight_grad(self, c: ConvSpec, x: torch.Tensor, dz: torch.Tensor, k: int, stride: int,
                     pad: int, cout: Optional[int] = None) -> None:
        """Queues a conv layer's weight gradient.  At batch 64 one layer's gradient is a few
        microseconds of work inside some 19 us of launch, fill, drain and fold latency, so the
        layers whose gradients are due together go out as one batched launch on the side
        stream (nothing reads them before the optimizer).  Layers are marked done, in the order
        they were queued, once that launch is enqueued."""
        self._pending_wg.append((c, x, dz, k, stride, pad, cout))
        if len(self._pending_wg) >= self.WGRAD_BATCH:
            self._flush_weight_grads()

    def _flush_weight_grads(self) -> None:
        if not self._pending_wg:
            return
        pending, self._pending_wg = self._pending_wg, []
        with self._beside(*[t for (_, x, dz, *_r) in pending for t in (x, dz)]):
            pnn.conv2d_bwd_weight_batched(
                [(x, dz, self._g(c.name + ".weight"), k, stride, pad, cout)
                 for (c, x, dz, k, stride, pad, cout) in pending])
            for c, *_r in pending:
                self._done(c)

    def _done(self, c: ConvSpec) -> None:
        if self.bucketer is not None:
            self.bucketer.layer_done(self.offsets[c.name + ".weight"][0])

    @contextmanager
    def _beside(self, *tensors: torch.Tensor):
        """Launches inside run on the weight-gradient stream, after everything queued on the
        current stream so far.  `tensors` are what those launches read: they stay alive until
        backward has joined the two streams, so the allocator cannot hand their memory to a later
        launch on the main stream while the side stream still reads it."""
        side = self.wgrad_stream
        if side is None:
            yield
            return
        replay.wait_stream(side, torch.cuda.current_stream(self.device))
        if self._down_unjoined:    # a projection's dz and batch-norm gradients are still on their way
            replay.wait_stream(side, self.down_stream)
        # kept alive until backward has joined the streams: cheaper for the host than telling the
        # allocator per tensor (record_stream; 8.71 against 8.80 ms per step), and it is what a
        # recorded pass needs, which replays the allocator's decisions as well
        self._keep.extend(tensors)
        with torch.cuda.stream(side):
            yield

    def backward(self, grad_hm: torch.Tensor) -> None:
        """grad_hm: float32 [B,K,H,W] = dLoss/dHeatmaps.  Fills self.grads (pre-zeroed here)."""
        stem, blocks, deconvs, final = self.specs
        sv = self.saved
        self._join_transposes()
        self.grads.zero_()
        # final 1x1 conv + bias
        dyp = pnn.final_grad_prepare(grad_hm, FINAL_KPAD, self._g(final.name + ".bias"))
        self._weight_grad(final, sv["feat"], dyp, 1, 1, 0, cout=final.cout)
        H, W = sv["feat"].shape[1:3]
        # deconv head.  With the fused batch-norm backward every backward-data GEMM of the chain
        # hands the layer below its dz directly (mask and sums in the GEMM epilogue).
        dz = None
        if self.fuse_bn_bwd and deconvs:
            dz = self._bwd_data_bn(deconvs[-1], sv["deconvs"][-1]["z"], dyp, self.wt[final.name],
                                   1, 1, 0)
        else:
            dx = pnn.conv2d_bwd_data(dyp, self.wt[final.name], (H, W), 1, 1, 0)
        for i in reversed(range(len(deconvs))):
            d, s = deconvs[i], sv["deconvs"][i]
            if dz is None:
                dz, _ = self._bn_bwd(d, dx, s["a"], s["z"], rebuild=True)
            self._flush_weight_grads()      # layers are marked done in arena order
            with self._beside(s["x"], dz):
                pnn.deconv4x4s2_bwd_weight(s["x"], dz, self._g(d.name + ".weight"))
                self._done(d)
            if self.fuse_bn_bwd and i > 0:
                dz = self._deconv_bwd_data_bn(deconvs[i - 1], sv["deconvs"][i - 1]["z"], dz,
                                              self.wt[d.name])
            else:
                dx = pnn.deconv4x4s2_bwd_data(dz, self.wt[d.name])
                dz = None
        # bottleneck blocks
        for i in reversed(range(len(blocks))):
            # the block's last GEMM also starts the batch-norm backward of the block below it
            below = ((blocks[i - 1].c3, sv["blocks"][i - 1]["z3"])
                     if i > 0 and self.fuse_res_bwd else None)
            dx = self.block_backward(blocks[i], sv["blocks"][i], dx, flush=False, below=below)
        # stem.  The blocks' last weight gradients go out first, so that they run beside the
        # pooling and batch-norm backward instead of behind them.
        self._flush_weight_grads()
        da0 = pnn.maxpool3x3s2_bwd(sv["pool_idx"], dx)
        dz0, _ = self._bn_bwd(stem, da0, sv["a0"], sv["z0"], rebuild=True)
        gw = self._g(stem.name + ".weight")
        with self._beside(sv["xs"], dz0):
            pnn.stem7x7s2_bwd_weight(sv["xs"], dz0, gw)
            pnn.mul_(gw, self.stem_mask)
            self._done(stem)
        if self.wgrad_stream is not None:   # the optimizer reads every gradient
            replay.wait_stream(torch.cuda.current_stream(self.device), self.wgrad_stream)
        self._keep.clear()

    # ------------------------------------------------------------------ one optimisation step
    def _pass(self, images_u8, x_s2d, joints, mask):
        """Forward with batch statistics, loss, backward: fills the gradient arena."""
        hm = self.forward_train(images_u8, x_s2d)
        loss, per_joint, grad, _ = ops.mse_fwd_bwd(hm, joints=joints, mask=mask,
                                                  stride=self.heatmap_stride, sigma=self.sigma,
                                                  radius=self.radius)
        self.backward(grad)
        self.saved = {}
        return loss, per_joint

    def replay_host(self, fn, *inputs: Optional[torch.Tensor]):
        """fn(*inputs) from a recorded call list (replay.RecordedPass): the first call's launches
        are kept by the library and enqueued again without the Python code that produced them.
        Static shapes; outputs are overwritten by the next call; not with a gradient exchange
        attached (its collectives are issued by the host) nor with the one-launch batch norm
        (its hand-off counter is an argument that moves from launch to launch)."""
        if self.bucketer is not None:
            raise RuntimeError("host replay and the bucketed gradient exchange are exclusive")
        if self.bn_sync is not None:
            raise RuntimeError("host replay and PE_BN_FUSED=1 are exclusive")
        return replay.run_recorded(self._recorded, self, fn, inputs)

    def train_step(self, images_u8: torch.Tensor, joints: torch.Tensor,
                   mask: Optional[torch.Tensor] = None, lr: float = 1e-3, optimizer: str = "adam",
                   beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                   momentum: float = 0.9, x_s2d: Optional[torch.Tensor] = None,
                   host_replay: bool = False):
        """images uint8 [B,H,W,3] (or None with x_s2d given, see forward_train); joints float32
        [B,K,3] (x, y, v) in input pixels; mask uint8 [B,H/4,W/4] or None.  host_replay: enqueue
        the forward/loss/backward pass from the call list recorded on the first step.
        Returns (loss [1] float32 device tensor, per_joint [K])."""
        if host_replay:
            loss, per_joint = self.replay_host(self._pass, images_u8, x_s2d, joints, mask)
        else:
            loss, per_joint = self._pass(images_u8, x_s2d, joints, mask)
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
        # the backward-data weights are not needed before the next backward pass: their re-pack
        # runs on the side stream, beside the next step's forward pass
        side = self.wgrad_stream if os.environ.get("PE_TRANSPOSE_SIDE", "1") != "0" else None
        if side is not None:
            self._join_transposes()
            replay.wait_stream(side, torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._tplan_run()
            self._wt_pending = True
        else:
            self.refresh_transposes()
        return loss, per_joint
