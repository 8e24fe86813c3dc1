"""ResNet-50 "simple baseline" pose network (3 deconv head, K heat maps) on the HIP kernels.

The reference's network files (nets/mobilenet_v2.py, model/mobilenet_pose.py;
PoseEstimation.pyproj:31-32) are absent; the architecture here is the one the build's
north star names: ResNet-50 v1 bottlenecks with the stride on the 3x3, three
ConvTranspose(4x4, s2, p1, 256) + BN + ReLU, and a 1x1 conv to K joints.

Parameters live in one flat float32 arena (master copy) with a bf16 working copy of the same
layout; every weight view is in the packed layout its kernel reads.  torch is used for
buffers, views and the one-off BN folding only.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import nn as pnn
from . import ops

BF16 = torch.bfloat16
BN_EPS = 1e-5


@dataclass
class ConvSpec:
    name: str
    kind: str            # "stem" | "conv" | "deconv" | "final"
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    bn: bool             # followed by batch norm (else plain bias)
    relu: bool


@dataclass
class BlockSpec:
    c1: ConvSpec
    c2: ConvSpec
    c3: ConvSpec
    down: Optional[ConvSpec]


def resnet50_pose_specs(num_joints: int = 17, layers=(3, 4, 6, 3), width: int = 64,
                        head_ch: int = 256):
    """(stem, blocks, deconvs, final) layer specs.  `width` scales every channel count
    (64 = ResNet-50); tests use narrower nets."""
    stem = ConvSpec("stem", "stem", 3, width, 7, 2, 3, True, True)
    blocks: List[BlockSpec] = []
    cin = width
    for li, n in enumerate(layers):
        mid = width * (2 ** li)
        cout = mid * 4
        for bi in range(n):
            s = 2 if (bi == 0 and li > 0) else 1
            p = f"layer{li + 1}.{bi}"
            c1 = ConvSpec(p + ".conv1", "conv", cin, mid, 1, 1, 0, True, True)
            c2 = ConvSpec(p + ".conv2", "conv", mid, mid, 3, s, 1, True, True)
            c3 = ConvSpec(p + ".conv3", "conv", mid, cout, 1, 1, 0, True, True)  # relu after add
            down = None
            if bi == 0:
                down = ConvSpec(p + ".down", "conv", cin, cout, 1, s, 0, True, False)
            blocks.append(BlockSpec(c1, c2, c3, down))
            cin = cout
    deconvs = []
    for i in range(3):
        deconvs.append(ConvSpec(f"deconv{i + 1}", "deconv", cin, head_ch, 4, 2, 1, True, True))
        cin = head_ch
    final = ConvSpec("final", "final", cin, num_joints, 1, 1, 0, False, False)
    return stem, blocks, deconvs, final


def all_convs(specs) -> List[ConvSpec]:
    stem, blocks, deconvs, final = specs
    out = [stem]
    for b in blocks:
        out += [b.c1, b.c2, b.c3] + ([b.down] if b.down else [])
    return out + deconvs + [final]


def framework_shape(c: ConvSpec) -> Tuple[int, ...]:
    if c.kind == "deconv":
        return (c.cin, c.cout, 4, 4)          # ConvTranspose2d layout
    return (c.cout, c.cin, c.k, c.k)


def packed_shape(c: ConvSpec) -> Tuple[int, ...]:
    if c.kind == "stem":
        return (c.cout, 4, 64)
    if c.kind == "deconv":
        return (4, c.cout, 4, c.cin)
    return (c.cout, c.k, c.k, c.cin)


def pack_weight(c: ConvSpec, w: torch.Tensor) -> torch.Tensor:
    """Framework-layout float32 weight -> packed layout (still float32)."""
    if c.kind == "stem":
        return pnn.pack_stem_weight(w.float()).float()
    if c.kind == "deconv":
        return pnn.pack_deconv_weight(w.float()).float()
    return w.permute(0, 2, 3, 1).contiguous()


def init_state(specs, seed: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init state dict in framework layouts (CPU float32): He-normal convs, BN gamma 1 /
    beta 0 / running (0, 1), deconv and final conv N(0, 0.001), zero bias."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for c in all_convs(specs):
        shape = framework_shape(c)
        if c.kind in ("deconv", "final"):
            w = torch.randn(shape, generator=g) * 0.001
        else:
            fan_out = c.cout * c.k * c.k
            w = torch.randn(shape, generator=g) * (2.0 / fan_out) ** 0.5
        st[c.name + ".weight"] = w
        if c.bn:
            st[c.name + ".bn.gamma"] = torch.ones(c.cout)
            st[c.name + ".bn.beta"] = torch.zeros(c.cout)
            st[c.name + ".bn.mean"] = torch.zeros(c.cout)
            st[c.name + ".bn.var"] = torch.ones(c.cout)
        else:
            st[c.name + ".bias"] = torch.zeros(c.cout)
    return st


def randomize_bn(state: Dict[str, torch.Tensor], seed: int = 5) -> None:
    """Give every BN non-trivial statistics (used by tests so folding is exercised)."""
    g = torch.Generator().manual_seed(seed)
    for k in list(state):
        n = state[k].numel()
        if k.endswith(".bn.gamma"):
            state[k] = torch.rand(n, generator=g) + 0.5
        elif k.endswith(".bn.beta") or k.endswith(".bias"):
            state[k] = torch.randn(n, generator=g) * 0.1
        elif k.endswith(".bn.mean"):
            state[k] = torch.randn(n, generator=g) * 0.1
        elif k.endswith(".bn.var"):
            state[k] = torch.rand(n, generator=g) + 0.5


def _align8(n: int) -> int:
    return (n + 7) // 8 * 8


class PoseResNet:
    """The network with its parameter arenas on one GPU."""

    def __init__(self, device, num_joints: int = 17, layers=(3, 4, 6, 3), width: int = 64,
                 head_ch: int = 256):
        self.device = torch.device(device)
        self.num_joints = num_joints
        self.specs = resnet50_pose_specs(num_joints, layers, width, head_ch)
        self.convs = all_convs(self.specs)
        # ---- arena layout: every tensor starts on an 8-element boundary
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for c in self.convs:
            shp = packed_shape(c)
            self.offsets[c.name + ".weight"] = (off, shp)
            off += _align8(int(torch.tensor(shp).prod()))
            names = (".bn.gamma", ".bn.beta") if c.bn else (".bias",)
            for nm in names:
                self.offsets[c.name + nm] = (off, (c.cout,))
                off += _align8(c.cout)
        self.n_params = off
        self._views: Dict[tuple, torch.Tensor] = {}
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.work = torch.zeros(off, dtype=BF16, device=self.device)
        self.bn_mean = {c.name: torch.zeros(c.cout, device=self.device) for c in self.convs if c.bn}
        self.bn_var = {c.name: torch.ones(c.cout, device=self.device) for c in self.convs if c.bn}
        self.scale: Dict[str, Optional[torch.Tensor]] = {}
        self.shift: Dict[str, Optional[torch.Tensor]] = {}
        self._fold_plan = None

    # ------------------------------------------------------------------ parameters
    def view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        """The named parameter inside an arena laid out like the master.  Views are cached per
        (arena storage, dtype): a training step asks for several hundred of them."""
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

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Copy a framework-layout state dict into the arenas (packing every weight)."""
        for c in self.convs:
            w = pack_weight(c, state[c.name + ".weight"])
            self.view(self.master, c.name + ".weight").copy_(w.to(self.device))
            if c.bn:
                self.view(self.master, c.name + ".bn.gamma").copy_(state[c.name + ".bn.gamma"])
                self.view(self.master, c.name + ".bn.beta").copy_(state[c.name + ".bn.beta"])
                self.bn_mean[c.name].copy_(state[c.name + ".bn.mean"])
                self.bn_var[c.name].copy_(state[c.name + ".bn.var"])
            else:
                self.view(self.master, c.name + ".bias").copy_(state[c.name + ".bias"])
        self.work.copy_(self.master)  # one rounding to bf16
        self.fold_bn()

    def fold_bn(self) -> None:
        """Inference epilogue constants: scale = gamma/sqrt(var+eps), shift = beta - mean*scale,
        for every layer in one launch, written in place (the tensors in self.scale / self.shift
        keep their addresses)."""
        if self._fold_plan is None:
            plan = pnn.BnFoldPlan(self.device)
            for c in self.convs:
                if c.bn:
                    self.scale[c.name] = torch.empty(c.cout, dtype=torch.float32, device=self.device)
                    self.shift[c.name] = torch.empty(c.cout, dtype=torch.float32, device=self.device)
                    plan.add(self.view(self.master, c.name + ".bn.gamma"),
                             self.view(self.master, c.name + ".bn.beta"), self.bn_mean[c.name],
                             self.bn_var[c.name], self.scale[c.name], self.shift[c.name])
                else:
                    self.scale[c.name] = None
                    self.shift[c.name] = self.view(self.master, c.name + ".bias")
            self._fold_plan = plan
        self._fold_plan.run(BN_EPS)

    def w(self, c: ConvSpec) -> torch.Tensor:
        return self.view(self.work, c.name + ".weight")

    # PE_RESNET_DOWN_STREAM=0: the skip-path projections stay on the main stream (comparison)
    down_stream = os.environ.get("PE_RESNET_DOWN_STREAM", "1") != "0"
    _side = None

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    # ------------------------------------------------------------------ inference
    def _conv(self, c: ConvSpec, x, residual=None, relu=None, **kw):
        return pnn.conv2d(x, self.w(c), self.scale[c.name], self.shift[c.name], residual,
                          c.stride, c.pad, c.relu if relu is None else relu, **kw)

    # Depth-first front: the stem, the pool and the first `front_blocks` bottlenecks run over
    # `front_chunk` images at a time, one chunk after the other, so that what a layer writes is
    # still in the last-level cache when the next layer reads it (at batch 256 the tensors of
    # layer1 are 100-400 MB each, more than the cache holds).  0 = whole batch at once.
    front_chunk = int(os.environ.get("PE_RESNET_CHUNK", "0"))
    front_blocks = int(os.environ.get("PE_RESNET_CHUNK_BLOCKS", "3"))

    def _front_shape(self, x_s2d: torch.Tensor, nblocks: int) -> Tuple[int, int, int, int]:
        """Shape of the activation behind the first `nblocks` bottlenecks."""
        B, h, w, _ = x_s2d.shape
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1           # 3x3 stride-2 pool, pad 1
        for b in self.specs[1][:nblocks]:
            if b.c2.stride == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return B, h, w, self.specs[1][nblocks - 1].c3.cout

    # The stem and the max pool as ONE launch (pe_stem7x7s2_pool_fwd_bf16): the stem's
    # half-resolution map (402 MB at batch 256) is never written or read back.  Bit-identical
    # to the two launches.  PE_RESNET_STEM_POOL=0: two launches (comparison).
    fuse_stem_pool = os.environ.get("PE_RESNET_STEM_POOL", "1") != "0"

    def _front(self, x_s2d: torch.Tensor, nblocks: int, out: Optional[torch.Tensor]) -> torch.Tensor:
        stem, blocks, _, _ = self.specs
        if (self.fuse_stem_pool and stem.cout == 64 and x_s2d.shape[1] % 2 == 0
                and x_s2d.shape[2] % 2 == 0):
            x = pnn.stem7x7s2_pool(x_s2d, self.w(stem), self.scale[stem.name],
                                   self.shift[stem.name])
        else:
            x = pnn.stem7x7s2(x_s2d, self.w(stem), self.scale[stem.name], self.shift[stem.name],
                              True)
            x = pnn.maxpool3x3s2(x)
        return self._blocks(x, blocks[:nblocks], out)

    # Chained bottlenecks (PE_RESNET_CHAIN=1): in the stages whose 3x3 conv has 64 output
    # channels (layer1), conv2 -> conv3 -> the next block's conv1 run as ONE launch over the same
    # pixel tiles (pe_conv2d_fwd_chain_bf16).  Bit-identical, and measured slower than a launch
    # per conv at batch 256 (389 against 375 us per bottleneck tail, 5.21 against 5.07 ms per
    # forward): 96 resident workgroups per XCD write 12 MB between a tile's store and its
    # re-read, three times the L2, so the re-read is served from memory as before, and the
    # 64-channel sub-tiles of the chain store 128-byte runs.  Off by default.
    chain = os.environ.get("PE_RESNET_CHAIN", "0") == "1"

    def _stage(self, c: ConvSpec, residual=None, relu=None, out=None) -> dict:
        return {"w": self.w(c), "scale": self.scale[c.name], "shift": self.shift[c.name],
                "residual": residual, "stride": c.stride, "pad": c.pad,
                "relu": c.relu if relu is None else relu, "out": out}

    def _chainable(self, b: BlockSpec) -> bool:
        return self.chain and b.c2.cout == 64 and b.c2.cin == 64 and b.c2.stride == 1

    # conv2 (3x3 64 -> 64) and conv3 (1x1 64 -> 256, residual, ReLU) of a first-stage bottleneck
    # as ONE launch (pe_conv3x3_c64_pw_fwd_bf16): the 64-channel tensor between them (100 MB at
    # batch 256) is never written or read back.  Bit-identical to the two launches; taken when
    # the layer has at least `pnn.PATCH3X3_MIN_TILES` 16 x 16 tiles (one persistent workgroup
    # per CU).  PE_RESNET_FUSE_C2C3=0 or PE_CONV_C2C3=0: two launches (comparison).
    fuse_c2c3 = os.environ.get("PE_RESNET_FUSE_C2C3", "1") != "0"
    # ... and, in a stage's first block, the 1x1 projection of the skip path too
    # (pe_conv3x3_c64_pw_proj_fwd_bf16).  PE_RESNET_FUSE_DOWN=0: the projection as its own launch.
    fuse_down = os.environ.get("PE_RESNET_FUSE_DOWN", "1") != "0"
    # ... and, when another bottleneck of the stage follows, that block's conv1 (1x1 256 -> 64) on
    # the tile just finished (pe_conv3x3_c64_pw_next_fwd_bf16): the 256-channel output is written
    # (the next block adds it back) but not read again by a conv1 launch (403 MB at batch 256).
    # PE_RESNET_FUSE_NEXT=0: conv1 as its own launch.
    fuse_next = os.environ.get("PE_RESNET_FUSE_NEXT", "1") != "0"

    def _next_c1_fusable(self, nxt: Optional[BlockSpec]) -> bool:
        if not self.fuse_next or nxt is None:
            return False
        c = nxt.c1
        return (c.k == 1 and c.stride == 1 and c.pad == 0 and c.cin == 256 and c.cout == 64
                and bool(c.relu))

    def _c2c3_fusable(self, b: BlockSpec, x: torch.Tensor) -> bool:
        c1, c2, c3 = b.c1, b.c2, b.c3
        B, H, W, _ = x.shape            # the block's input; conv1 keeps its size
        return (self.fuse_c2c3 and c1.stride == 1 and c2.k == 3 and c2.stride == 1 and c2.pad == 1
                and c2.relu and c3.k == 1 and c3.stride == 1
                and pnn.c2c3_fusable(B, H, W, self.w(c2), self.w(c3), self.scale[c2.name],
                                     self.shift[c2.name]))

    def _blocks(self, x: torch.Tensor, blocks, out: Optional[torch.Tensor]) -> torch.Tensor:
        side = self._side_stream() if self.down_stream else None
        c1_out = None   # conv1 of the current block, when the previous block's chain made it
        for bi, b in enumerate(blocks):
            last = bi == len(blocks) - 1
            if self._chainable(b):
                main = torch.cuda.current_stream(self.device)
                if b.down and side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        idn = self._conv(b.down, x)
                elif b.down:
                    idn = self._conv(b.down, x)
                else:
                    idn = x
                o1 = c1_out if c1_out is not None else self._conv(b.c1, x)
                if b.down and side is not None:
                    main.wait_stream(side)
                nxt = blocks[bi + 1] if not last else None
                stages = [self._stage(b.c2),
                          self._stage(b.c3, residual=idn, relu=True, out=out if last else None)]
                if nxt is not None and nxt.c1.stride == 1 and nxt.c1.k == 1:
                    stages.append(self._stage(nxt.c1))
                outs = pnn.conv2d_chain(o1, stages)
                x = outs[1]
                c1_out = outs[2] if len(outs) == 3 else None
                continue
            have_c1, c1_out = c1_out, None
            kw = {"out": out} if (out is not None and bi == len(blocks) - 1) else {}
            if self._c2c3_fusable(b, x):
                main = torch.cuda.current_stream(self.device)
                if (self.fuse_down and b.down and b.down.k == 1 and b.down.stride == 1
                        and b.down.cin == 64 and not b.down.relu):
                    # the skip path's projection is computed inside the same launch as well:
                    # its 256-channel tensor (403 MB at batch 256) is never written or read
                    o1 = have_c1 if have_c1 is not None else self._conv(b.c1, x)
                    d = b.down
                    x = pnn.conv3x3_c64_pw_proj(o1, self.w(b.c2), self.scale[b.c2.name],
                                                self.shift[b.c2.name], self.w(b.c3),
                                                self.scale[b.c3.name], self.shift[b.c3.name], x,
                                                self.w(d), self.scale[d.name], self.shift[d.name], **kw)
                    continue
                if b.down and side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        idn = self._conv(b.down, x)
                    o1 = have_c1 if have_c1 is not None else self._conv(b.c1, x)
                    main.wait_stream(side)
                else:
                    o1 = have_c1 if have_c1 is not None else self._conv(b.c1, x)
                    idn = self._conv(b.down, x) if b.down else x
                nxt = blocks[bi + 1] if not last else None
                if self._next_c1_fusable(nxt):
                    n1 = nxt.c1
                    x, c1_out = pnn.conv3x3_c64_pw_next(
                        o1, self.w(b.c2), self.scale[b.c2.name], self.shift[b.c2.name], self.w(b.c3),
                        self.scale[b.c3.name], self.shift[b.c3.name], idn, self.w(n1),
                        self.scale[n1.name], self.shift[n1.name])
                    continue
                x = pnn.conv3x3_c64_pw(o1, self.w(b.c2), self.scale[b.c2.name], self.shift[b.c2.name],
                                       self.w(b.c3), self.scale[b.c3.name], self.shift[b.c3.name],
                                       idn, True, **kw)
                continue
            if b.down and side is not None:
                # the projection of the skip path runs beside conv1 / conv2 (it is bound by
                # memory traffic, conv2 by the matrix cores); both streams wait for each other,
                # so freed activations are reused only behind their last reader
                main = torch.cuda.current_stream(self.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    idn = self._conv(b.down, x)
                o = self._conv(b.c2, have_c1 if have_c1 is not None else self._conv(b.c1, x))
                main.wait_stream(side)
            else:
                o = self._conv(b.c2, have_c1 if have_c1 is not None else self._conv(b.c1, x))
                idn = self._conv(b.down, x) if b.down else x
            x = self._conv(b.c3, o, residual=idn, relu=True, **kw)
        return x

    def forward_features(self, x_s2d: torch.Tensor, skip_last_deconvs: int = 0) -> torch.Tensor:
        stem, blocks, deconvs, final = self.specs
        B = x_s2d.shape[0]
        ch, nb = self.front_chunk, min(self.front_blocks, len(blocks))
        if ch > 0 and nb > 0 and B > ch and B % ch == 0:
            x = torch.empty(self._front_shape(x_s2d, nb), dtype=BF16, device=x_s2d.device)
            for i in range(0, B, ch):
                self._front(x_s2d[i:i + ch], nb, x[i:i + ch])
            x = self._blocks(x, blocks[nb:], None)
        else:
            x = self._front(x_s2d, len(blocks), None)
        for d in deconvs[:len(deconvs) - skip_last_deconvs]:
            x = pnn.deconv4x4s2(x, self.w(d), self.scale[d.name], self.shift[d.name], True)
        return x

    # PE_RESNET_FUSE_HEAD=1: the last transposed conv and the final 1x1 conv as ONE launch
    # (pe_deconv4x4s2_head_fwd_bf16): the 256-channel map between them (402 MB at batch 256) is
    # never written or read back.  Bit-identical to the two launches, and measured no faster at
    # batch 256 (5.05 against 5.02 ms per forward): a workgroup has to walk both 128-channel
    # tiles of its pixels one after the other, so the second pass re-reads its input tile 16 K
    # steps after the first, when the L2 has long dropped it (two workgroups side by side share
    # it while it is hot), and the epilogues no longer hide behind neighbours' K loops.  With
    # the head's epilogue switched off altogether the launch still saved only 46-70 us of the
    # 500 the two launches take, so it stays off by default.
    fuse_head = os.environ.get("PE_RESNET_FUSE_HEAD", "0") == "1"

    def _head_fusable(self) -> bool:
        deconvs, final = self.specs[2], self.specs[3]
        return (self.fuse_head and len(deconvs) > 0 and deconvs[-1].cout == 256
                and deconvs[-1].cin % 64 == 0 and final.k == 1 and final.cout <= 32)

    def predict(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] -> float32 heat maps [B,K,H/4,W/4] (inference, folded BN)."""
        deconvs, final = self.specs[2], self.specs[3]
        x = pnn.preprocess_u8(images_u8)
        if self._head_fusable():
            x = self.forward_features(x, skip_last_deconvs=1)
            d = deconvs[-1]
            return pnn.deconv4x4s2_head(x, self.w(d), self.scale[d.name], self.shift[d.name],
                                        self.w(final), self.shift[final.name], relu=True)
        x = self.forward_features(x)
        return self._conv(final, x, out_f32_nchw=True)

    def predict_and_decode(self, images_u8: torch.Tensor):
        return ops.decode(self.predict(images_u8))
