"""MobileNetV2 pose network (inference) on the HIP kernels: the backbone family the reference
names (nets/mobilenet_v2.py, model/mobilenet_pose.py; PoseEstimation.pyproj:31-32, both absent)
at the resolution its notebook uses: 320x320 images, stride-8 label maps of 40x40
(test.ipynb:919, 1010).

Architecture (the public MobileNetV2-1.0 definition, no reference source to follow): 3x3/2 stem
to 32 channels, 17 inverted-residual blocks (t, c, n, s) = (1,16,1,1) (6,24,2,2) (6,32,3,2)
(6,64,4,2) (6,96,3,1) (6,160,3,2) (6,320,1,1), every conv followed by BN and ReLU6 except the
linear 1x1 projections.  Once the running stride reaches `output_stride` (8) later strides
become dilation, so the feature map stays 40x40 at 320x320.  Heads: a 1x1 conv to K heat maps
and, optionally, a 1x1 conv to 4 channels per keypoint pair (vector maps); float32 NCHW outputs.

Layout: activations NHWC bf16 at their true channel counts (multiples of 8; a count that is
not would be padded up with zero weights, scales and shifts, so the padding stays zero through
ReLU6 and the linear projections).  The pointwise convs run on the MFMA engine
(pe_conv2d_fwd_bf16), which walks a K run of 24, 96, 144 ... channels in steps of 64 with a
part-filled last step, or packs 16 / 32 channels as multi-pixel runs; the depthwise convs run on
the bandwidth-bound pe_depthwise3x3_fwd_bf16.  Nothing is padded to the GEMM's K step in
memory: these networks are bound by tensor traffic, not by the matrix cores.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import nn as pnn
from . import ops
from .graph import Layer, TrainGraph, Var

BF16 = torch.bfloat16
BN_EPS = 1e-5
MBV2_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1),
                (6, 160, 3, 2), (6, 320, 1, 1)]


def pad8(c: int) -> int:
    """Channel counts are kept as they are, rounded up to whole 16-byte chunks (every count of
    the public MobileNetV2 definition is a multiple of 8 already)."""
    return (c + 7) // 8 * 8


@dataclass
class MBBlock:
    name: str
    cin: int
    cout: int
    expand: int          # expansion factor t (1 = no expansion conv)
    stride: int
    dilation: int
    residual: bool

    @property
    def mid(self) -> int:
        return self.cin * self.expand


def mobilenet_v2_specs(output_stride: int = 8, setting=None, stem_ch: int = 32) -> List[MBBlock]:
    blocks: List[MBBlock] = []
    cin, cur, dil = stem_ch, 2, 1
    for si, (t, c, n, s) in enumerate(setting or MBV2_SETTING):
        for i in range(n):
            stride = s if i == 0 else 1
            d = dil
            if stride == 2 and cur >= output_stride:
                stride, dil = 1, dil * 2   # the stride turns into dilation of the layers after it
            elif stride == 2:
                cur *= 2
            blocks.append(MBBlock(f"block{len(blocks)}", cin, c, t, stride, d,
                                  stride == 1 and cin == c))
            cin = c
    return blocks


def detection_specs(cin: int, channels: int = 128, expand: int = 2, levels: int = 2) -> List[MBBlock]:
    """The detection branch's extra feature levels: one stride-2 inverted-residual block per
    level on top of the stride-8 trunk output, so a 320x320 image gives the 20x20 and 10x10 maps
    the reference's anchors live on (feat_1 / feat_2, test.ipynb:432-464).  Block shapes are
    this build's own (the reference's model file is absent)."""
    out: List[MBBlock] = []
    for i in range(levels):
        out.append(MBBlock(f"det{i + 1}", cin, channels, expand, 2, 1, False))
        cin = channels
    return out


BOX_FIELDS = 5   # per anchor: objectness logit + 4 box regression values


def detection_macs(H: int = 320, W: int = 320, cin: int = 320, output_stride: int = 8,
                   anchors_per_cell: int = 3) -> int:
    """Multiply-accumulates of the detection branch's forward at the true channel counts."""
    h, w = H // output_stride, W // output_stride
    macs = 0
    for b in detection_specs(cin):
        if b.expand != 1:
            macs += h * w * b.cin * b.mid
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        macs += h * w * 9 * b.mid + h * w * b.mid * b.cout
        macs += h * w * b.cout * anchors_per_cell * BOX_FIELDS
    return macs


def forward_macs(H: int = 320, W: int = 320, num_joints: int = 17, num_pairs: int = 0,
                 output_stride: int = 8, setting=None) -> int:
    """Multiply-accumulates of one forward at the true (unpadded) channel counts."""
    blocks = mobilenet_v2_specs(output_stride, setting)
    h, w = H // 2, W // 2
    macs = h * w * 27 * 32
    for b in blocks:
        if b.expand != 1:
            macs += h * w * b.cin * b.mid
        h, w = (h - 1) // b.stride + 1, (w - 1) // b.stride + 1
        macs += h * w * 9 * b.mid + h * w * b.mid * b.cout
    macs += h * w * blocks[-1].cout * (num_joints + 4 * num_pairs)
    return macs


def init_state(blocks: List[MBBlock], num_joints: int = 17, num_pairs: int = 0,
               seed: int = 3, stem_ch: int = 32, det_blocks: Optional[List[MBBlock]] = None,
               anchors_per_cell: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init state dict in framework layouts (CPU float32): He-normal convs (fan-out),
    BN gamma 1 / beta 0 / running (0, 1), heads N(0, 0.01) with zero bias."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, groups=1):
        fan_out = cout * k * k // groups
        st[name + ".weight"] = torch.randn((cout, cin // groups, k, k), generator=g) * (2.0 / fan_out) ** 0.5
        st[name + ".bn.gamma"] = torch.ones(cout)
        st[name + ".bn.beta"] = torch.zeros(cout)
        st[name + ".bn.mean"] = torch.zeros(cout)
        st[name + ".bn.var"] = torch.ones(cout)

    conv("stem", stem_ch, 3, 3)
    for b in blocks:
        if b.expand != 1:
            conv(b.name + ".expand", b.mid, b.cin, 1)
        conv(b.name + ".dw", b.mid, b.mid, 3, groups=b.mid)
        conv(b.name + ".project", b.cout, b.mid, 1)
    cf = blocks[-1].cout
    st["heat.weight"] = torch.randn((num_joints, cf, 1, 1), generator=g) * 0.01
    st["heat.bias"] = torch.zeros(num_joints)
    if num_pairs:
        st["vec.weight"] = torch.randn((4 * num_pairs, cf, 1, 1), generator=g) * 0.01
        st["vec.bias"] = torch.zeros(4 * num_pairs)
    for i, b in enumerate(det_blocks or []):
        if b.expand != 1:
            conv(b.name + ".expand", b.mid, b.cin, 1)
        conv(b.name + ".dw", b.mid, b.mid, 3, groups=b.mid)
        conv(b.name + ".project", b.cout, b.mid, 1)
        n = anchors_per_cell * BOX_FIELDS
        st[f"box{i + 1}.weight"] = torch.randn((n, b.cout, 1, 1), generator=g) * 0.01
        st[f"box{i + 1}.bias"] = torch.zeros(n)
    return st


class PoseMobileNetV2:
    """MobileNetV2 backbone + heat-map (and vector-map) heads on one GPU, inference only."""

    def __init__(self, device, num_joints: int = 17, num_pairs: int = 0, output_stride: int = 8,
                 setting=None):
        self.device = torch.device(device)
        self.num_joints = num_joints
        self.num_pairs = num_pairs
        self.blocks = mobilenet_v2_specs(output_stride, setting)
        self.output_stride = output_stride
        self.p: Dict[str, torch.Tensor] = {}
        self.kw: Dict[str, dict] = {}

    # ------------------------------------------------------------------ parameters
    def _fold(self, st, name, cout_pad):
        """Queues the layer's folded constants (padded channels: gamma = 0 -> scale = shift = 0);
        load_state computes all of them with one pe_bn_fold_f32 launch."""
        def up(key, fill):
            t = torch.full((cout_pad,), fill, dtype=torch.float32)
            v = st[name + key].float().reshape(-1)
            t[:v.numel()] = v
            return t.to(self.device)

        self.p[name + ".scale"] = torch.empty(cout_pad, dtype=torch.float32, device=self.device)
        self.p[name + ".shift"] = torch.empty(cout_pad, dtype=torch.float32, device=self.device)
        src = (up(".bn.gamma", 0.0), up(".bn.beta", 0.0), up(".bn.mean", 0.0), up(".bn.var", 1.0))
        self._fold_src.append(src)     # kept alive until the launch has read them
        self._fold_plan.add(*src, self.p[name + ".scale"], self.p[name + ".shift"])

    def _pointwise(self, st, name, cin, cout, fold=True):
        w = torch.zeros(pad8(cout) if fold else cout, pad8(cin), 1, 1)
        w[:cout, :cin] = st[name + ".weight"]
        if pad8(cin) in (8, 16, 32):      # multi-pixel K runs: the engine's narrow packing
            self.p[name + ".weight"] = pnn.pack_conv_weight_narrow(w).to(self.device)
            self.kw[name] = {"kw": 1}
        else:
            self.p[name + ".weight"] = pnn.pack_conv_weight(w).to(self.device)
            self.kw[name] = {}
        if fold:
            self._fold(st, name, pad8(cout))
        else:
            self.p[name + ".bias"] = st[name + ".bias"].float().to(self.device)

    def load_state(self, st: Dict[str, torch.Tensor]) -> None:
        self.p.clear()
        self._fold_plan = pnn.BnFoldPlan(self.device)
        self._fold_src = []
        sc = pad8(st["stem.weight"].shape[0])
        self.p["stem.weight"] = torch.zeros(sc, 2, 64, dtype=BF16)
        self.p["stem.weight"][:st["stem.weight"].shape[0]] = pnn.pack_stem3x3_weight(st["stem.weight"])
        self.p["stem.weight"] = self.p["stem.weight"].to(self.device)
        self._fold(st, "stem", sc)
        for b in self.blocks:
            if b.expand != 1:
                self._pointwise(st, b.name + ".expand", b.cin, b.mid)
            w = torch.zeros(9, pad8(b.mid))
            w[:, :b.mid] = st[b.name + ".dw.weight"][:, 0].reshape(b.mid, 9).t()
            self.p[b.name + ".dw.weight"] = w.to(BF16).contiguous().to(self.device)
            self._fold(st, b.name + ".dw", pad8(b.mid))
            self._pointwise(st, b.name + ".project", b.mid, b.cout)
        cf = self.blocks[-1].cout
        self._pointwise(st, "heat", cf, self.num_joints, fold=False)
        if self.num_pairs:
            self._pointwise(st, "vec", cf, 4 * self.num_pairs, fold=False)
        self._fold_plan.run(BN_EPS)
        torch.cuda.synchronize(self.device)
        self._fold_plan, self._fold_src = None, []

    # ------------------------------------------------------------------ inference
    def forward_features(self, x_s2d: torch.Tensor) -> torch.Tensor:
        p = self.p
        x = pnn.stem3x3s2(x_s2d, p["stem.weight"], p["stem.scale"], p["stem.shift"], relu=2)
        for b in self.blocks:
            h = x
            if b.expand != 1:
                n = b.name + ".expand"
                h = pnn.conv2d(h, p[n + ".weight"], p[n + ".scale"], p[n + ".shift"], relu=2,
                               **self.kw[n])
            n = b.name + ".dw"
            h = pnn.depthwise3x3(h, p[n + ".weight"], p[n + ".scale"], p[n + ".shift"],
                                 stride=b.stride, dilation=b.dilation, relu=2)
            n = b.name + ".project"
            x = pnn.conv2d(h, p[n + ".weight"], p[n + ".scale"], p[n + ".shift"],
                           residual=x if b.residual else None, relu=0, **self.kw[n])
        return x

    def predict(self, images_u8: torch.Tensor):
        """uint8 [B,H,W,3] -> heat maps float32 [B,K,H/8,W/8] (and vector maps [B,4P,H/8,W/8]
        when the model has a vector-map head: returns (heat, vec))."""
        if not self.p:
            raise RuntimeError("load_state() has not been called")
        f = self.forward_features(pnn.preprocess_u8(images_u8))
        heat = pnn.conv2d(f, self.p["heat.weight"], None, self.p["heat.bias"], out_f32_nchw=True,
                          **self.kw["heat"])
        if not self.num_pairs:
            return heat
        vec = pnn.conv2d(f, self.p["vec.weight"], None, self.p["vec.bias"], out_f32_nchw=True,
                         **self.kw["vec"])
        return heat, vec

    def decode(self, heatmaps: torch.Tensor):
        return ops.decode(heatmaps)


# --------------------------------------------------------------------------------------------
# Training (and inference from the same arenas) on the tape-based graph
# --------------------------------------------------------------------------------------------
# keypoint pairs of the 17-joint COCO skeleton (limbs), used when a model asks for vector maps
# without naming its pairs
COCO_PAIRS = [(15, 13), (13, 11), (16, 14), (14, 12), (11, 12), (5, 11), (6, 12), (5, 6), (5, 7),
              (6, 8), (7, 9), (8, 10), (1, 2), (0, 1), (0, 2), (1, 3), (2, 4), (3, 5), (4, 6)]


def _block_layers(b: MBBlock) -> List[Layer]:
    layers = []
    if b.expand != 1:
        layers.append(Layer(b.name + ".expand", "conv", pad8(b.cin), pad8(b.mid), 1, 1, 2))
    layers.append(Layer(b.name + ".dw", "dw", pad8(b.mid), pad8(b.mid), 3, b.stride, 2,
                        b.dilation))
    layers.append(Layer(b.name + ".project", "conv", pad8(b.mid), pad8(b.cout), 1, 1, 0))
    return layers


def _graph_layers(blocks: List[MBBlock], num_joints: int, num_pairs: int, stem_ch: int = 32,
                  det_blocks: Optional[List[MBBlock]] = None, anchors_per_cell: int = 3):
    layers = [Layer("stem", "stem3", 3, pad8(stem_ch), 3, 2, 2)]
    for b in blocks:
        layers.extend(_block_layers(b))
    cf = pad8(blocks[-1].cout)
    layers.append(Layer("heat", "head", cf, num_joints))
    if num_pairs:
        layers.append(Layer("vec", "head", cf, 4 * num_pairs))
    for i, b in enumerate(det_blocks or []):     # forward execution order: level by level
        layers.extend(_block_layers(b))
        layers.append(Layer(f"box{i + 1}", "head", pad8(b.cout), anchors_per_cell * BOX_FIELDS))
    return layers


def box_outputs(maps: List[torch.Tensor], anchors_per_cell: int = 3):
    """Box-head maps, float32 [B, A*5, h, w] per level (channel = anchor * 5 + field) ->
    (logit [B,N], reg [B,N,4]) with anchors ordered level, row, column, anchor: the order of the
    reference's `classes` / `regs` labels (test.ipynb:423-464, 504-507)."""
    rows = []
    for m in maps:
        B, _, h, w = m.shape
        rows.append(m.view(B, anchors_per_cell, BOX_FIELDS, h, w).permute(0, 3, 4, 1, 2)
                    .reshape(B, h * w * anchors_per_cell, BOX_FIELDS))
    flat = torch.cat(rows, dim=1)
    return flat[..., 0].contiguous(), flat[..., 1:].contiguous()


def box_grads(g_logit: torch.Tensor, g_reg: torch.Tensor, shapes, anchors_per_cell: int = 3):
    """Inverse of box_outputs for gradients: -> one float32 [B, A*5, h, w] map per level."""
    flat = torch.cat([g_logit.unsqueeze(-1), g_reg], dim=-1)
    out, at = [], 0
    for (h, w) in shapes:
        n = h * w * anchors_per_cell
        part = flat[:, at:at + n].reshape(-1, h, w, anchors_per_cell, BOX_FIELDS)
        out.append(part.permute(0, 3, 4, 1, 2).reshape(-1, anchors_per_cell * BOX_FIELDS, h, w)
                   .contiguous())
        at += n
    return out


def pad_state(layers, st: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """True-size framework-layout state -> the channel-padded sizes of the graph's layers.
    Padded weights are zero and padded batch norms have gamma = beta = 0, so padded channels
    hold exact zeros in every activation and receive exact zeros as gradients."""
    out: Dict[str, torch.Tensor] = {}
    for L in layers:
        w = st[L.name + ".weight"]
        if L.kind == "stem3":
            shape = (L.cout, 3, 3, 3)
        elif L.kind == "dw":
            shape = (L.cout, 1, 3, 3)
        else:
            shape = (L.cout, L.cin, 1, 1)
        wp = torch.zeros(shape)
        wp[tuple(slice(0, n) for n in w.shape)] = w
        out[L.name + ".weight"] = wp
        if L.kind == "head":
            out[L.name + ".bias"] = st[L.name + ".bias"]
            continue
        for nm, fill in ((".bn.gamma", 0.0), (".bn.beta", 0.0), (".bn.mean", 0.0), (".bn.var", 1.0)):
            v = torch.full((L.cout,), fill)
            v[:st[L.name + nm].numel()] = st[L.name + nm]
            out[L.name + nm] = v
    return out


class TrainablePoseMobileNetV2(TrainGraph):
    """MobileNetV2 backbone + heat-map (and vector-map) heads with the parameter arenas on one
    GPU: forward with batch statistics, label rendering fused into the two losses, backward
    (pointwise convs on the MFMA engine, depthwise convs on bandwidth kernels), fused optimizer;
    `predict` runs the same graph with folded BN.  Labels follow the reference's notebook:
    stride-8 maps, 3x3 supports (test.ipynb:606-653, 919)."""

    def __init__(self, device, num_joints: int = 17, pairs=None, output_stride: int = 8,
                 setting=None, bn_momentum: float = 0.1, sigma: float = 1.0, radius: int = 1,
                 detection: bool = False, anchors_per_cell: int = 3, box_loss_weight: float = 1.0):
        """detection: adds the reference's third output (`labels['bboxes']`, test.ipynb:432):
        two stride-2 feature levels on top of the trunk, each with a 1x1 box head that predicts
        an objectness logit and four regression values per anchor."""
        self.blocks = mobilenet_v2_specs(output_stride, setting)
        self.pairs = list(pairs) if pairs else []
        self.num_joints = num_joints
        self.num_pairs = len(self.pairs)
        self.det_blocks = detection_specs(self.blocks[-1].cout) if detection else []
        self.anchors_per_cell = anchors_per_cell
        self.box_loss_weight = box_loss_weight
        super().__init__(device, _graph_layers(self.blocks, num_joints, self.num_pairs,
                                               det_blocks=self.det_blocks,
                                               anchors_per_cell=anchors_per_cell), bn_momentum)
        self.output_stride = output_stride
        self.sigma = sigma
        self.radius = radius

    def load_state(self, st: Dict[str, torch.Tensor]) -> None:
        super().load_state(pad_state(self.layers, st))

    def block(self, b: MBBlock, x: Var) -> Var:
        """One inverted-residual block: 1x1 expand, depthwise 3x3, linear 1x1 projection."""
        L = self.by_name
        h = x
        if b.expand != 1:
            h = self.unit(L[b.name + ".expand"], h)
        h = self.unit(L[b.name + ".dw"], h)
        return self.unit(L[b.name + ".project"], h, residual=x if b.residual else None)

    def forward(self, x_s2d: torch.Tensor, training: bool):
        """-> heat, or (heat, vec) with a vector-map head; with the detection branch a third /
        second entry follows: the list of box-head maps, one float32 [B, A*5, h, w] per level."""
        self.begin(training)
        L = self.by_name
        x = self.unit(L["stem"], Var(x_s2d, False))
        for b in self.blocks:
            x = self.block(b, x)
        outs = [self.head(L["heat"], x)]
        if self.num_pairs:
            outs.append(self.head(L["vec"], x))
        if self.det_blocks:
            f, maps = x, []
            for i, b in enumerate(self.det_blocks):
                f = self.block(b, f)
                maps.append(self.head(L[f"box{i + 1}"], f))
            outs.append(maps)
        return outs[0] if len(outs) == 1 else tuple(outs)

    def forward_train(self, images_u8: Optional[torch.Tensor],
                      x_s2d: Optional[torch.Tensor] = None):
        return self.forward(x_s2d if x_s2d is not None else pnn.preprocess_u8(images_u8), True)

    def predict(self, images_u8: torch.Tensor):
        """uint8 [B,H,W,3] -> heat maps (and vector maps, and with the detection branch
        (logit [B,N], reg [B,N,4]) over all anchors), running BN statistics (call fold_bn()
        after training steps)."""
        out = self.forward(pnn.preprocess_u8(images_u8), False)
        if not self.det_blocks:
            return out
        return tuple(out[:-1]) + (box_outputs(out[-1], self.anchors_per_cell),)

    def _pass_u8(self, images_u8, joints, mask, det_labels=None):
        return self._pass(pnn.preprocess_u8(images_u8), joints, mask, det_labels)

    def _pass(self, x_s2d, joints, mask, det_labels=None):
        """Forward with batch statistics, the losses of every head, backward: fills the gradient
        arena.  det_labels: (classes [B,N], weights [B,N], regs [B,N,4]) per-anchor targets as
        detection.anchor_assign makes them; the detection branch needs them."""
        out = self.forward(x_s2d, True)
        heat = out[0] if isinstance(out, tuple) else out
        s = float(self.output_stride)
        loss, per_joint, grad, _ = ops.mse_fwd_bwd(heat, joints=joints, mask=mask, stride=s,
                                                  sigma=self.sigma, radius=self.radius)
        grads = {"heat": grad}
        if self.num_pairs:
            vloss, _, vgrad = ops.vecmap_loss_fwd_bwd(out[1], joints, self.pairs, s, self.radius,
                                                      mask)
            grads["vec"] = vgrad
            loss.add_(vloss)      # in place: a recorded pass cannot repeat an allocation
        if self.det_blocks:
            if det_labels is None:
                raise ValueError("this network has a detection branch: pass det_labels")
            from . import detection

            maps = out[-1]
            logit, reg = box_outputs(maps, self.anchors_per_cell)
            classes, weights, regs = det_labels
            loss4, g_logit, g_reg = detection.box_loss_fwd_bwd(logit, reg, classes, weights, regs)
            if self.box_loss_weight != 1.0:
                g_logit, g_reg = g_logit * self.box_loss_weight, g_reg * self.box_loss_weight
            gm = box_grads(g_logit, g_reg, [tuple(m.shape[2:]) for m in maps],
                           self.anchors_per_cell)
            for i, g in enumerate(gm):
                grads[f"box{i + 1}"] = g
            loss = loss + loss4[0:1] * self.box_loss_weight
            self.last_box_loss = loss4
        self.backward(grads)
        return loss, per_joint

    def train_step(self, images_u8, joints: torch.Tensor, mask: Optional[torch.Tensor] = None,
                   lr: float = 1e-3, optimizer: str = "adam", beta1: float = 0.9,
                   beta2: float = 0.999, eps: float = 1e-8, momentum: float = 0.9,
                   x_s2d: Optional[torch.Tensor] = None, hip_graph: bool = False,
                   det_labels=None, host_replay: bool = False):
        """One optimisation step on heat-map loss + vector-map loss (+ objectness and box loss
        with the detection branch, which needs det_labels = (classes, weights, regs)); returns
        (loss [1] device tensor, per_joint [K]).  hip_graph: replay the forward/loss/backward
        pass as one captured HIP graph (see TrainGraph.replay); host_replay: enqueue it from the
        call list recorded on the first step (TrainGraph.replay_host)."""
        fn, x = (self._pass, x_s2d) if x_s2d is not None else (self._pass_u8, images_u8)
        if hip_graph or host_replay:
            if self.det_blocks:
                raise ValueError("replay is not wired for the detection branch")
            run = self.replay_host if host_replay else self.replay
            loss, per_joint = run(fn, x, joints, mask)
        else:
            loss, per_joint = fn(x, joints, mask, det_labels)
        self.optimizer_step(lr, optimizer, beta1, beta2, eps, momentum)
        return loss, per_joint
