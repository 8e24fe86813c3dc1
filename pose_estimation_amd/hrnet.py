"""HRNet-W32-style multi-branch pose network (inference) on the HIP kernels, for the
high-resolution configuration: 384x288 images, 96x72 heat maps.

The reference's own network is MobileNetV2 (nets/mobilenet_v2.py, model/mobilenet_pose.py;
PoseEstimation.pyproj:31-32, absent); this backbone is not in it.  Architecture = the public
HRNet-W32 pose definition: two 3x3/2 stem convs to 64 channels, four bottlenecks (256 channels),
then three stages of parallel branches at strides 4/8/16/32 with widths 32/64/128/256
(1, 4 and 3 exchange modules; four basic blocks per branch per module), and a 1x1 head on the
stride-4 branch.  Every conv is followed by BN; ReLU follows except before a residual or
fusion sum, which applies it after the sum.

How it maps onto the kernels:
  * every conv + folded BN (+ residual) (+ ReLU) is one pe_conv2d_fwd_bf16 launch; the
    32-channel branch uses the narrow-input weight packing (two pixels per 64-element K run)
    instead of padding the branch to 64 channels;
  * a fusion output i sums, in this order: its own branch, the stride-2 3x3 chains from the
    higher-resolution branches (each chain's last conv takes the running sum as its residual),
    then the 1x1-projected lower-resolution branches, which one pe_fuse_upsample_add_bf16
    launch upsamples (nearest), adds and rectifies.
Activations are NHWC bf16; BN is folded into per-channel float32 scale and shift.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import nn as pnn
from . import ops
from . import replay
from .graph import Layer, TrainGraph, Var

BF16 = torch.bfloat16
BN_EPS = 1e-5

# The next bottleneck's conv1 inside the fused conv2 + conv3 launch of layer1
# (pe_conv3x3_c64_pw_next_fwd_bf16).  PE_HRNET_FUSE_NEXT=0: conv1 as its own launch.
FUSE_NEXT = os.environ.get("PE_HRNET_FUSE_NEXT", "1") != "0"


@dataclass
class Unit:
    """conv k x k (pad k//2) + BN (+ ReLU)."""
    name: str
    cin: int
    cout: int
    k: int
    stride: int = 1
    relu: bool = True


@dataclass
class Bottleneck:
    c1: Unit
    c2: Unit
    c3: Unit
    down: Optional[Unit]


@dataclass
class BasicBlock:
    c1: Unit
    c2: Unit


@dataclass
class Module:
    branches: List[List[BasicBlock]]
    # fuse[i][j]: None when i == j, [1x1 unit] when j > i (then upsampled by 2^(j-i)),
    # a chain of (i - j) stride-2 3x3 units when j < i
    fuse: List[List[Optional[List[Unit]]]] = field(default_factory=list)


@dataclass
class HRNetSpec:
    stem: List[Unit]
    layer1: List[Bottleneck]
    transitions: List[List[Optional[Unit]]]   # per stage, per branch: None = pass through
    stages: List[List[Module]]
    widths: List[int]
    num_joints: int


def hrnet_specs(num_joints: int = 17, width: int = 32, modules=(1, 4, 3), blocks: int = 4,
                layer1_blocks: int = 4) -> HRNetSpec:
    widths = [width, 2 * width, 4 * width, 8 * width]
    stem = [Unit("stem.conv1", 3, 64, 3, 2), Unit("stem.conv2", 64, 64, 3, 2)]
    layer1 = []
    cin = 64
    for i in range(layer1_blocks):
        n = f"layer1.{i}"
        layer1.append(Bottleneck(Unit(n + ".conv1", cin, 64, 1), Unit(n + ".conv2", 64, 64, 3),
                                 Unit(n + ".conv3", 64, 256, 1, relu=False),
                                 Unit(n + ".down", cin, 256, 1, relu=False) if cin != 256 else None))
        cin = 256
    transitions, stages = [], []
    prev = [256]
    for si, nmod in enumerate(modules):
        nb = si + 2
        cur = widths[:nb]
        tr: List[Optional[Unit]] = []
        for i in range(nb):
            if i < len(prev):
                tr.append(Unit(f"transition{si + 1}.{i}", prev[i], cur[i], 3)
                          if prev[i] != cur[i] else None)
            else:
                tr.append(Unit(f"transition{si + 1}.{i}", prev[-1], cur[i], 3, 2))
        transitions.append(tr)
        mods = []
        for m in range(nmod):
            last = si == len(modules) - 1 and m == nmod - 1   # only the stride-4 output is used
            name = f"stage{si + 2}.{m}"
            br = [[BasicBlock(Unit(f"{name}.branch{i}.{k}.conv1", cur[i], cur[i], 3),
                              Unit(f"{name}.branch{i}.{k}.conv2", cur[i], cur[i], 3, relu=False))
                   for k in range(blocks)] for i in range(nb)]
            fuse: List[List[Optional[List[Unit]]]] = []
            for i in range(1 if last else nb):
                row: List[Optional[List[Unit]]] = []
                for j in range(nb):
                    if j == i:
                        row.append(None)
                    elif j > i:
                        row.append([Unit(f"{name}.fuse{i}.{j}", cur[j], cur[i], 1, relu=False)])
                    else:
                        chain = []
                        for k in range(i - j):
                            end = k == i - j - 1
                            chain.append(Unit(f"{name}.fuse{i}.{j}.{k}", cur[j],
                                              cur[i] if end else cur[j], 3, 2, relu=not end))
                        row.append(chain)
                fuse.append(row)
            mods.append(Module(br, fuse))
        stages.append(mods)
        prev = cur
    return HRNetSpec(stem, layer1, transitions, stages, widths, num_joints)


def all_units(spec: HRNetSpec) -> List[Unit]:
    out = list(spec.stem)
    for b in spec.layer1:
        out += [b.c1, b.c2, b.c3] + ([b.down] if b.down else [])
    for tr, mods in zip(spec.transitions, spec.stages):
        out += [t for t in tr if t is not None]
        for m in mods:
            for br in m.branches:
                for blk in br:
                    out += [blk.c1, blk.c2]
            for row in m.fuse:
                for chain in row:
                    out += chain or []
    return out


def forward_macs(spec: HRNetSpec, H: int = 384, W: int = 288) -> int:
    """Multiply-accumulates of one forward, from the layer list (true channel counts)."""
    res = {}

    def run(u: Unit, h: int, w: int):
        ho, wo = (h - 1) // u.stride + 1, (w - 1) // u.stride + 1
        res["macs"] = res.get("macs", 0) + ho * wo * u.k * u.k * u.cin * u.cout
        return ho, wo

    h, w = H, W
    for u in spec.stem:
        h, w = run(u, h, w)
    for b in spec.layer1:
        for u in (b.c1, b.c2, b.c3) + ((b.down,) if b.down else ()):
            run(u, h, w)
    sizes = [(h, w)]
    for tr, mods in zip(spec.transitions, spec.stages):
        new = []
        for i, t in enumerate(tr):
            src = sizes[i] if i < len(sizes) else sizes[-1]
            new.append(run(t, *src) if t is not None else src)
        sizes = new
        for m in mods:
            for i, br in enumerate(m.branches):
                for blk in br:
                    run(blk.c1, *sizes[i])
                    run(blk.c2, *sizes[i])
            for row in m.fuse:
                for j, chain in enumerate(row):
                    s = sizes[j]
                    for u in chain or []:
                        s = run(u, *s)
    res["macs"] += sizes[0][0] * sizes[0][1] * spec.widths[0] * spec.num_joints
    return res["macs"]


def train_step_bytes(spec: HRNetSpec, batch: int = 64, H: int = 384, W: int = 288):
    """Algorithmic HBM bytes of one training step (the denominator for measured step traffic):
    flops.conv_train_bytes over the layer list, with the second conv of every residual unit and
    the last conv of every fusion chain priced as residual layers (they read what they are
    added to and write the sum; backward reads the other consumer's gradient), the uint8
    images and their space-to-depth copy, the float32 heat maps through the loss and the
    optimizer's 30 bytes per parameter.  Returns (total, {group: bytes})."""
    from .flops import conv_train_bytes

    g = {"convs": 0}
    params = [0]

    def run(u: Unit, h: int, w: int, residual: bool = False):
        ho, wo = (h - 1) // u.stride + 1, (w - 1) // u.stride + 1
        we = u.k * u.k * u.cin * u.cout
        g["convs"] += conv_train_bytes(batch * h * w, max(u.cin, 8), batch * ho * wo, u.cout, we,
                                       residual=residual)
        params[0] += we + 2 * u.cout
        return ho, wo

    h, w = H, W
    g["input"] = batch * H * W * 3 + 2 * batch * (H // 2) * (W // 2) * 16
    for u in spec.stem:
        h, w = run(u, h, w)
    for b in spec.layer1:
        run(b.c1, h, w)
        run(b.c2, h, w)
        run(b.c3, h, w, residual=True)
        if b.down:
            run(b.down, h, w)
    sizes = [(h, w)]
    for tr, mods in zip(spec.transitions, spec.stages):
        new = []
        for i, t in enumerate(tr):
            src = sizes[i] if i < len(sizes) else sizes[-1]
            new.append(run(t, *src) if t is not None else src)
        sizes = new
        for m in mods:
            for i, br in enumerate(m.branches):
                for blk in br:
                    run(blk.c1, *sizes[i])
                    run(blk.c2, *sizes[i], residual=True)
            for row in m.fuse:
                for j, chain in enumerate(row):
                    s_ = sizes[j]
                    for n, u in enumerate(chain or []):
                        s_ = run(u, *s_, residual=(n == len(chain) - 1))
    mo = batch * sizes[0][0] * sizes[0][1]
    we = spec.widths[0] * spec.num_joints
    g["final+loss"] = conv_train_bytes(mo, spec.widths[0], 0, 0, we, bn=False) \
        + 4 * (4 * mo * spec.num_joints)
    g["optimizer"] = 30 * (params[0] + we + spec.num_joints)
    return sum(g.values()), g


def init_state(spec: HRNetSpec, seed: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init state dict in framework layouts (CPU float32): He-normal convs (fan-out),
    BN gamma 1 / beta 0 / running (0, 1), head N(0, 0.001) with zero bias."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for u in all_units(spec):
        st[u.name + ".weight"] = torch.randn((u.cout, u.cin, u.k, u.k), generator=g) \
            * (2.0 / (u.cout * u.k * u.k)) ** 0.5
        st[u.name + ".bn.gamma"] = torch.ones(u.cout)
        st[u.name + ".bn.beta"] = torch.zeros(u.cout)
        st[u.name + ".bn.mean"] = torch.zeros(u.cout)
        st[u.name + ".bn.var"] = torch.ones(u.cout)
    st["final.weight"] = torch.randn((spec.num_joints, spec.widths[0], 1, 1), generator=g) * 0.001
    st["final.bias"] = torch.zeros(spec.num_joints)
    return st


class PoseHRNet:
    """HRNet-W32-style pose network on one GPU, inference only."""

    def __init__(self, device, num_joints: int = 17, width: int = 32, modules=(1, 4, 3),
                 blocks: int = 4, layer1_blocks: int = 4):
        self.device = torch.device(device)
        self.spec = hrnet_specs(num_joints, width, modules, blocks, layer1_blocks)
        self.p: Dict[str, torch.Tensor] = {}
        # The branches of an exchange module are independent chains of small launches (the
        # low-resolution ones fill a fraction of the chip): each branch but the first runs on a
        # stream of its own and the module joins them before its fusion layers.
        # PE_HRNET_BRANCH_STREAMS=0: one stream, for comparison.
        self._side: List[torch.cuda.Stream] = []
        self.branch_streams = os.environ.get("PE_HRNET_BRANCH_STREAMS", "1") != "0"

    def _side_streams(self, n: int) -> List[torch.cuda.Stream]:
        while len(self._side) < n:
            self._side.append(torch.cuda.Stream(self.device))
        return self._side[:n]

    # ------------------------------------------------------------------ parameters
    def _pack(self, w: torch.Tensor) -> torch.Tensor:
        cin = w.shape[1]
        if cin in (8, 16, 32):
            return pnn.pack_conv_weight_narrow(w)
        if cin % 64 != 0:
            raise ValueError(f"input channels {cin}: need 8, 16, 32 or a multiple of 64")
        return pnn.pack_conv_weight(w)

    def load_state(self, st: Dict[str, torch.Tensor]) -> None:
        self.p.clear()
        plan, keep = pnn.BnFoldPlan(self.device), []
        for u in all_units(self.spec):
            w = st[u.name + ".weight"]
            if u.name == "stem.conv1":
                pw = pnn.pack_stem3x3_weight(w)
            else:
                pw = self._pack(w)
            self.p[u.name + ".weight"] = pw.to(self.device)
            src = tuple(st[u.name + k].float().contiguous().to(self.device)
                        for k in (".bn.gamma", ".bn.beta", ".bn.mean", ".bn.var"))
            keep.append(src)
            C = src[0].numel()
            self.p[u.name + ".scale"] = torch.empty(C, dtype=torch.float32, device=self.device)
            self.p[u.name + ".shift"] = torch.empty(C, dtype=torch.float32, device=self.device)
            plan.add(*src, self.p[u.name + ".scale"], self.p[u.name + ".shift"])
        plan.run(BN_EPS)               # every layer's folded constants in one launch
        torch.cuda.synchronize(self.device)
        self.p["final.weight"] = self._pack(st["final.weight"]).to(self.device)
        self.p["final.bias"] = st["final.bias"].float().to(self.device)

    # ------------------------------------------------------------------ inference
    def _unit(self, u: Unit, x: torch.Tensor, residual=None, relu=None) -> torch.Tensor:
        p = self.p
        return pnn.conv2d(x, p[u.name + ".weight"], p[u.name + ".scale"], p[u.name + ".shift"],
                          residual, u.stride, u.k // 2, u.relu if relu is None else relu, kw=u.k)

    def _branch(self, x: torch.Tensor, br) -> torch.Tensor:
        for blk in br:
            x = self._unit(blk.c2, self._unit(blk.c1, x), residual=x, relu=True)
        return x

    def _module(self, m: Module, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        if self.branch_streams and len(xs) > 1:
            # Every side stream starts behind everything queued on the main stream and the main
            # stream resumes behind every side stream, so a tensor freed after this module is
            # reused only by work ordered after its last reader, whichever stream made it.
            main = torch.cuda.current_stream(self.device)
            side = self._side_streams(len(xs) - 1)
            ys = [None] * len(xs)
            for i in range(len(xs) - 1, 0, -1):     # the smallest maps first: they take longest
                side[i - 1].wait_stream(main)       # per byte and hide under branch 0
                with torch.cuda.stream(side[i - 1]):
                    ys[i] = self._branch(xs[i], m.branches[i])
            ys[0] = self._branch(xs[0], m.branches[0])
            for st in side:
                main.wait_stream(st)
            # the fusion rows read every branch but are independent of each other
            outs = [None] * len(m.fuse)
            for i in range(len(m.fuse) - 1, 0, -1):
                if i - 1 < len(side):
                    side[i - 1].wait_stream(main)
                    with torch.cuda.stream(side[i - 1]):
                        outs[i] = self._fuse_row(m, i, ys)
            for i in range(len(m.fuse)):
                if outs[i] is None:
                    outs[i] = self._fuse_row(m, i, ys)
            for st in side:
                main.wait_stream(st)
            return outs
        ys = [self._branch(x, br) for x, br in zip(xs, m.branches)]
        return [self._fuse_row(m, i, ys) for i in range(len(m.fuse))]

    def _fuse_row(self, m: Module, i: int, ys: List[torch.Tensor]) -> torch.Tensor:
        row = m.fuse[i]
        lows = [(j, row[j]) for j in range(len(row)) if j > i]
        s = ys[i]
        for j in range(i):
            t = ys[j]
            chain = row[j]
            for u in chain[:-1]:
                t = self._unit(u, t)
            # the last conv of the chain adds the running sum; it also rectifies when
            # nothing else follows
            s = self._unit(chain[-1], t, residual=s, relu=(j == i - 1 and not lows))
        if lows:
            proj = [self._unit(chain[0], ys[j]) for j, chain in lows]
            s = pnn.fuse_upsample_add(s, proj, [j - i for j, _ in lows], relu=True)
        return s

    def forward_features(self, x_s2d: torch.Tensor) -> torch.Tensor:
        sp, p = self.spec, self.p
        u = sp.stem[0]
        x = pnn.stem3x3s2(x_s2d, p[u.name + ".weight"], p[u.name + ".scale"],
                          p[u.name + ".shift"], relu=1)
        x = self._unit(sp.stem[1], x)
        c1_out = None    # conv1 of the current block, when the previous block's launch made it
        for bi, b in enumerate(sp.layer1):
            c2, c3 = b.c2.name, b.c3.name
            B, H, W, C = x.shape
            have_c1, c1_out = c1_out, None
            nxt = sp.layer1[bi + 1].c1 if bi + 1 < len(sp.layer1) else None
            fused = (b.c2.k == 3 and b.c2.stride == 1 and b.c3.k == 1 and b.c3.stride == 1
                     and pnn.c2c3_fusable(B, H, W, p[c2 + ".weight"], p[c3 + ".weight"],
                                          p[c2 + ".scale"], p[c2 + ".shift"]))
            d = b.down
            if fused and d is not None and d.k == 1 and d.stride == 1 and C == 64 and not d.relu:
                # conv2 + conv3 + the skip path's projection + ReLU in one launch
                o1 = have_c1 if have_c1 is not None else self._unit(b.c1, x)
                x = pnn.conv3x3_c64_pw_proj(o1, p[c2 + ".weight"], p[c2 + ".scale"],
                                            p[c2 + ".shift"], p[c3 + ".weight"], p[c3 + ".scale"],
                                            p[c3 + ".shift"], x, p[d.name + ".weight"],
                                            p[d.name + ".scale"], p[d.name + ".shift"])
                continue
            idn = self._unit(d, x) if d else x
            o1 = have_c1 if have_c1 is not None else self._unit(b.c1, x)
            if (fused and FUSE_NEXT and nxt is not None and nxt.k == 1 and nxt.stride == 1
                    and nxt.cin == 256 and nxt.cout == 64 and nxt.relu):
                # ... and the next block's conv1 on the tile just finished
                x, c1_out = pnn.conv3x3_c64_pw_next(
                    o1, p[c2 + ".weight"], p[c2 + ".scale"], p[c2 + ".shift"], p[c3 + ".weight"],
                    p[c3 + ".scale"], p[c3 + ".shift"], idn, p[nxt.name + ".weight"],
                    p[nxt.name + ".scale"], p[nxt.name + ".shift"])
            elif fused:
                # conv2 + conv3 + residual + ReLU in one launch (bit-identical to the two)
                x = pnn.conv3x3_c64_pw(o1, p[c2 + ".weight"], p[c2 + ".scale"], p[c2 + ".shift"],
                                       p[c3 + ".weight"], p[c3 + ".scale"], p[c3 + ".shift"], idn, True)
            else:
                x = self._unit(b.c3, self._unit(b.c2, o1), residual=idn, relu=True)
        xs = [x]
        for tr, mods in zip(sp.transitions, sp.stages):
            xs = [(xs[i] if t is None else self._unit(t, xs[i])) if i < len(xs)
                  else self._unit(t, xs[-1]) for i, t in enumerate(tr)]
            for m in mods:
                xs = self._module(m, xs)
        return xs[0]

    def predict(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] -> heat maps float32 [B,K,H/4,W/4]."""
        if not self.p:
            raise RuntimeError("load_state() has not been called")
        f = self.forward_features(pnn.preprocess_u8(images_u8))
        return pnn.conv2d(f, self.p["final.weight"], None, self.p["final.bias"],
                          out_f32_nchw=True, kw=1)

    def decode(self, heatmaps: torch.Tensor):
        return ops.decode(heatmaps)


# --------------------------------------------------------------------------------------------
# Training (and inference from the same arenas) on the tape-based graph
# --------------------------------------------------------------------------------------------
def _graph_layers(spec: HRNetSpec):
    layers = []
    for u in all_units(spec):
        kind = "stem3" if u.name == "stem.conv1" else "conv"
        layers.append(Layer(u.name, kind, u.cin, u.cout, u.k, u.stride, 1 if u.relu else 0))
    layers.append(Layer("final", "head", spec.widths[0], spec.num_joints, 1, 1, 0))
    return layers


class TrainablePoseHRNet(TrainGraph):
    """HRNet-W32-style pose network with its parameter arenas on one GPU: forward with batch
    statistics, fused target-render + masked-MSE loss, backward through the branches and
    exchange modules, fused optimizer; `predict` runs the same graph with folded BN."""

    def __init__(self, device, num_joints: int = 17, width: int = 32, modules=(1, 4, 3),
                 blocks: int = 4, layer1_blocks: int = 4, bn_momentum: float = 0.1,
                 sigma: float = 3.0, heatmap_stride: float = 4.0):
        self.spec = hrnet_specs(num_joints, width, modules, blocks, layer1_blocks)
        super().__init__(device, _graph_layers(self.spec), bn_momentum)
        self.num_joints = num_joints
        self.sigma = sigma
        self.radius = int(round(3 * sigma))
        self.heatmap_stride = heatmap_stride
        self._side: List[torch.cuda.Stream] = []
        self.branch_streams = os.environ.get("PE_HRNET_BRANCH_STREAMS", "1") != "0"
        # training too: the branches of a module on streams of their own in forward (the step is
        # bound by the GPU, 36.8 -> 34.5 ms; backward stays on one stream, see
        # TrainGraph.backward).  Activations made on a side stream are read by backward on the
        # main stream and freed there; the side streams start every module behind the main
        # stream, so their next allocation is ordered after those readers.  The one-launch batch
        # norm's hand-off buffer belongs to one stream, so not with it.
        self.multi_stream = (self.branch_streams
                             and os.environ.get("PE_HRNET_TRAIN_STREAMS", "1") != "0")

    def _u(self, u: Unit, x, residual=None, relu=None):
        return self.unit(self.by_name[u.name], x, residual,
                         None if relu is None else (1 if relu else 0))

    def _branch(self, x, br):
        for blk in br:
            x = self._u(blk.c2, self._u(blk.c1, x), residual=x, relu=True)
        return x

    def _module(self, m: Module, xs):
        if (not self.training or self.multi_stream) and self.branch_streams and len(xs) > 1:
            # the branches on streams of their own, as in PoseHRNet._module
            main = torch.cuda.current_stream(self.device)
            nside = len(xs) - 1
            while len(self._side) < nside:
                self._side.append(torch.cuda.Stream(self.device))
            side = self._side[:nside]
            ys = [None] * len(xs)
            for st in side:
                replay.wait_stream(st, main)
            for i in range(len(xs) - 1, 0, -1):
                with torch.cuda.stream(side[min(i, nside) - 1]):
                    ys[i] = self._branch(xs[i], m.branches[i])
            ys[0] = self._branch(xs[0], m.branches[0])
            for st in side:
                replay.wait_stream(main, st)
        else:
            ys = [self._branch(x, br) for x, br in zip(xs, m.branches)]
        outs = []
        for i, row in enumerate(m.fuse):
            lows = [(j, row[j]) for j in range(len(row)) if j > i]
            s = ys[i]
            for j in range(i):
                t = ys[j]
                chain = row[j]
                for u in chain[:-1]:
                    t = self._u(u, t)
                s = self._u(chain[-1], t, residual=s, relu=(j == i - 1 and not lows))
            if lows:
                proj = [self._u(chain[0], ys[j]) for j, chain in lows]
                s = self.fuse(s, proj, [j - i for j, _ in lows], relu=True)
            outs.append(s)
        return outs

    def forward(self, x_s2d: torch.Tensor, training: bool) -> torch.Tensor:
        """Space-to-depth image -> float32 heat maps [B,K,H/4,W/4]."""
        sp = self.spec
        self.begin(training)
        x = Var(x_s2d, False)
        x = self._u(sp.stem[0], x)
        x = self._u(sp.stem[1], x)
        for b in sp.layer1:
            idn = self._u(b.down, x) if b.down else x
            x = self._u(b.c3, self._u(b.c2, self._u(b.c1, x)), residual=idn, relu=True)
        xs = [x]
        for tr, mods in zip(sp.transitions, sp.stages):
            xs = [(xs[i] if t is None else self._u(t, xs[i])) if i < len(xs)
                  else self._u(t, xs[-1]) for i, t in enumerate(tr)]
            for m in mods:
                xs = self._module(m, xs)
        return self.head(self.by_name["final"], xs[0])

    def forward_train(self, images_u8: Optional[torch.Tensor],
                      x_s2d: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.forward(x_s2d if x_s2d is not None else pnn.preprocess_u8(images_u8), True)

    def predict(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] -> float32 heat maps (running BN statistics; call fold_bn() after
        training steps)."""
        return self.forward(pnn.preprocess_u8(images_u8), False)

    def _pass_u8(self, images_u8, joints, mask):
        return self._pass(pnn.preprocess_u8(images_u8), joints, mask)

    def _pass(self, x_s2d, joints, mask):
        """Forward with batch statistics, loss, backward: fills the gradient arena."""
        hm = self.forward(x_s2d, True)
        loss, per_joint, grad, _ = ops.mse_fwd_bwd(hm, joints=joints, mask=mask,
                                                  stride=self.heatmap_stride, sigma=self.sigma,
                                                  radius=self.radius)
        self.backward({"final": grad})
        return loss, per_joint

    def train_step(self, images_u8, joints: torch.Tensor, mask: Optional[torch.Tensor] = None,
                   lr: float = 1e-3, optimizer: str = "adam", beta1: float = 0.9,
                   beta2: float = 0.999, eps: float = 1e-8, momentum: float = 0.9,
                   x_s2d: Optional[torch.Tensor] = None, hip_graph: bool = False,
                   host_replay: bool = False):
        """One optimisation step; returns (loss [1] device tensor, per_joint [K]).  hip_graph:
        replay the forward/loss/backward pass as one captured HIP graph (see TrainGraph.replay);
        host_replay: enqueue it from the call list recorded on the first step
        (TrainGraph.replay_host)."""
        fn, x = (self._pass, x_s2d) if x_s2d is not None else (self._pass_u8, images_u8)
        if host_replay:
            loss, per_joint = self.replay_host(fn, x, joints, mask)
        elif hip_graph:
            loss, per_joint = self.replay(fn, x, joints, mask)
        else:
            loss, per_joint = fn(x, joints, mask)
        self.optimizer_step(lr, optimizer, beta1, beta2, eps, momentum)
        return loss, per_joint
