"""torch-CPU float32 restatement of the conv / deconv layers of the pose network.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference's network files
(nets/mobilenet_v2.py, model/mobilenet_pose.py, model/base_model.py;
PoseEstimation.pyproj:30-32) are absent and its arithmetic lived in TensorFlow 1.x
kernels, so these functions restate the public layer definitions with
torch.nn.functional on the CPU.  Inputs are float32 tensors holding bf16-representable
values (the kernels' operands are bf16); accumulation is float32; the result is returned
in float32 *before* the final bf16 rounding so tests can apply it explicitly.
All tensors here are NHWC to match the kernels; weights are in framework layouts.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def bf16r(t: torch.Tensor) -> torch.Tensor:
    """Round a float32 tensor to the nearest bf16 value (kept as float32)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def conv2d(x_nhwc, w_oihw, scale=None, shift=None, residual=None, stride=1, pad=0, relu=False,
           acc64=False):
    """acc64=True accumulates the dot products in float64 (then rounds to float32): a second
    accumulation order, used to measure how sensitive a deep bf16 network is to that order."""
    if acc64:
        y = F.conv2d(_nchw(x_nhwc).double(), w_oihw.double(), None, stride=stride,
                     padding=pad).float()
    else:
        y = F.conv2d(_nchw(x_nhwc), w_oihw, None, stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = _nhwc(y)
    if residual is not None:
        y = y + residual
    if relu:
        y = torch.relu(y)
    return y


def deconv4x4s2(x_nhwc, w_iohw, scale=None, shift=None, relu=False, acc64=False):
    if acc64:
        y = F.conv_transpose2d(_nchw(x_nhwc).double(), w_iohw.double(), None, stride=2,
                               padding=1).float()
    else:
        y = F.conv_transpose2d(_nchw(x_nhwc), w_iohw, None, stride=2, padding=1)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = _nhwc(y)
    if relu:
        y = torch.relu(y)
    return y


def preprocess_u8(img_u8: np.ndarray, mean, std) -> np.ndarray:
    """uint8 [B,H,W,3] -> float32 (x/255 - mean) * (1/std), evaluated in float32 in that
    order, then rounded to bf16 (returned as float32 NHWC [B,H,W,3])."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)
    inv = np.float32(1.0) / np.asarray(std, np.float32)
    y = ((x - m) * inv).astype(np.float32)
    return bf16r(torch.from_numpy(y)).numpy()


def space_to_depth16(x_nhwc3: np.ndarray) -> np.ndarray:
    """[B,H,W,3] -> [B,H/2,W/2,16] with channel (ry*2+rx)*3+c and four zero channels."""
    B, H, W, C = x_nhwc3.shape
    out = np.zeros((B, H // 2, W // 2, 16), x_nhwc3.dtype)
    for ry in range(2):
        for rx in range(2):
            s = (ry * 2 + rx) * 3
            out[..., s:s + 3] = x_nhwc3[:, ry::2, rx::2, :]
    return out


def maxpool3x3s2(x_nhwc):
    return _nhwc(F.max_pool2d(_nchw(x_nhwc), 3, 2, 1))


def pose_resnet_forward(specs, state, img_u8: np.ndarray, mean, std, bn_eps: float = 1e-5,
                        acc64: bool = False):
    """Inference forward of the ResNet + 3-deconv pose net on the CPU, float32 accumulation,
    with operands rounded to bf16 at the points where the kernels store bf16: the weights, the
    preprocessed image and every layer output except the final heat maps.

    specs: (stem, blocks, deconvs, final) from pose_estimation_amd.model.resnet50_pose_specs
    (plain dataclasses; no device code runs).  state: framework-layout float32 tensors.
    Returns float32 heat maps [B,K,H/4,W/4].

    A deep bf16 network amplifies the choice of float32 accumulation order (an order change
    flips an occasional bf16 rounding, and later layers magnify it), so an end-to-end bound
    is stated against the spread between acc64=False and acc64=True of this same function.
    """
    stem, blocks, deconvs, final = specs

    def fold(c):
        if not c.bn:
            return None, state[c.name + ".bias"]
        sc = state[c.name + ".bn.gamma"] / torch.sqrt(state[c.name + ".bn.var"] + bn_eps)
        return sc, state[c.name + ".bn.beta"] - state[c.name + ".bn.mean"] * sc

    def conv(c, x, residual=None, relu=None):
        sc, sh = fold(c)
        w = bf16r(state[c.name + ".weight"])
        return bf16r(conv2d(x, w, sc, sh, residual, c.stride, c.pad,
                            c.relu if relu is None else relu, acc64=acc64))

    x = torch.from_numpy(preprocess_u8(img_u8, mean, std))
    x = conv(stem, x)
    x = maxpool3x3s2(x)
    for b in blocks:
        o = conv(b.c1, x)
        o = conv(b.c2, o)
        idn = conv(b.down, x) if b.down else x
        x = conv(b.c3, o, residual=idn, relu=True)
    for d in deconvs:
        sc, sh = fold(d)
        x = bf16r(deconv4x4s2(x, bf16r(state[d.name + ".weight"]), sc, sh, relu=True,
                              acc64=acc64))
    sc, sh = fold(final)
    y = conv2d(x, bf16r(state[final.name + ".weight"]), sc, sh, acc64=acc64)
    return y.permute(0, 3, 1, 2).contiguous()


def depthwise3x3(x_nhwc, w_c133, scale=None, shift=None, stride=1, dilation=1, relu6=True,
                 acc64=False):
    """Depthwise 3x3, pad = dilation; w in torch layout [C,1,3,3]."""
    C = x_nhwc.shape[3]
    if acc64:
        y = F.conv2d(_nchw(x_nhwc).double(), w_c133.double(), None, stride=stride,
                     padding=dilation, dilation=dilation, groups=C).float()
    else:
        y = F.conv2d(_nchw(x_nhwc), w_c133, None, stride=stride, padding=dilation,
                     dilation=dilation, groups=C)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = _nhwc(y)
    return torch.clamp(y, 0.0, 6.0) if relu6 else y


def mobilenet_pose_forward(blocks, state, img_u8: np.ndarray, mean, std, bn_eps: float = 1e-5,
                           acc64: bool = False, with_vec: bool = False):
    """Inference forward of the MobileNetV2 pose net on the CPU (true channel counts, no
    padding), float32 accumulation, operands rounded to bf16 where the kernels store bf16: the
    weights, the preprocessed image and every layer output except the heads' float32 maps.

    blocks: pose_estimation_amd.mobilenet.mobilenet_v2_specs(...) (plain dataclasses).
    Returns heat maps [B,K,h,w] (and vector maps when with_vec)."""
    def fold(name):
        sc = state[name + ".bn.gamma"] / torch.sqrt(state[name + ".bn.var"] + bn_eps)
        return sc, state[name + ".bn.beta"] - state[name + ".bn.mean"] * sc

    def relu6(t):
        return torch.clamp(t, 0.0, 6.0)

    x = torch.from_numpy(preprocess_u8(img_u8, mean, std))
    sc, sh = fold("stem")
    x = bf16r(relu6(conv2d(x, bf16r(state["stem.weight"]), sc, sh, None, 2, 1, acc64=acc64)))
    for b in blocks:
        h = x
        if b.expand != 1:
            sc, sh = fold(b.name + ".expand")
            h = bf16r(relu6(conv2d(h, bf16r(state[b.name + ".expand.weight"]), sc, sh,
                                   acc64=acc64)))
        sc, sh = fold(b.name + ".dw")
        h = bf16r(depthwise3x3(h, bf16r(state[b.name + ".dw.weight"]), sc, sh, b.stride,
                               b.dilation, True, acc64=acc64))
        sc, sh = fold(b.name + ".project")
        x = bf16r(conv2d(h, bf16r(state[b.name + ".project.weight"]), sc, sh,
                         x if b.residual else None, acc64=acc64))
    heat = conv2d(x, bf16r(state["heat.weight"]), None, state["heat.bias"], acc64=acc64)
    heat = heat.permute(0, 3, 1, 2).contiguous()
    if not with_vec:
        return heat
    vec = conv2d(x, bf16r(state["vec.weight"]), None, state["vec.bias"], acc64=acc64)
    return heat, vec.permute(0, 3, 1, 2).contiguous()


def hrnet_pose_forward(spec, state, img_u8: np.ndarray, mean, std, bn_eps: float = 1e-5,
                       acc64: bool = False):
    """Inference forward of the HRNet-style multi-branch pose net on the CPU, float32
    accumulation, operands rounded to bf16 where the kernels store bf16 (weights, the
    preprocessed image, every conv output, every fusion output).

    spec: pose_estimation_amd.hrnet.hrnet_specs(...) (plain dataclasses).  A fusion output sums
    its own branch, then the strided chains from higher resolutions (rounded after each), then
    the nearest-upsampled 1x1 projections of the lower resolutions in one float32 sum.
    Returns float32 heat maps [B,K,H/4,W/4]."""
    def unit(u, x, residual=None, relu=None):
        sc = state[u.name + ".bn.gamma"] / torch.sqrt(state[u.name + ".bn.var"] + bn_eps)
        sh = state[u.name + ".bn.beta"] - state[u.name + ".bn.mean"] * sc
        return bf16r(conv2d(x, bf16r(state[u.name + ".weight"]), sc, sh, residual, u.stride,
                            u.k // 2, u.relu if relu is None else relu, acc64=acc64))

    def up(t, f):
        return t.repeat_interleave(1 << f, dim=1).repeat_interleave(1 << f, dim=2)

    x = torch.from_numpy(preprocess_u8(img_u8, mean, std))
    for u in spec.stem:
        x = unit(u, x)
    for b in spec.layer1:
        idn = unit(b.down, x) if b.down else x
        x = unit(b.c3, unit(b.c2, unit(b.c1, x)), residual=idn, relu=True)
    xs = [x]
    for tr, mods in zip(spec.transitions, spec.stages):
        xs = [(xs[i] if t is None else unit(t, xs[i])) if i < len(xs) else unit(t, xs[-1])
              for i, t in enumerate(tr)]
        for m in mods:
            ys = []
            for x, br in zip(xs, m.branches):
                for blk in br:
                    x = unit(blk.c2, unit(blk.c1, x), residual=x, relu=True)
                ys.append(x)
            outs = []
            for i, row in enumerate(m.fuse):
                lows = [j for j in range(len(row)) if j > i]
                s = ys[i]
                for j in range(i):
                    t = ys[j]
                    for u in row[j][:-1]:
                        t = unit(u, t)
                    s = unit(row[j][-1], t, residual=s, relu=(j == i - 1 and not lows))
                if lows:
                    for j in lows:
                        s = s + up(unit(row[j][0], ys[j]), j - i)
                    s = bf16r(torch.relu(s))
                outs.append(s)
            xs = outs
    y = conv2d(xs[0], bf16r(state["final.weight"]), None, state["final.bias"], acc64=acc64)
    return y.permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
# Training step: forward with batch statistics, masked MSE against rendered targets, autograd.
# ----------------------------------------------------------------------------------------------
class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of a value in forward and of its gradient in backward: the points where
    the kernels store an activation / an activation gradient in bf16."""

    @staticmethod
    def forward(ctx, x):
        return bf16r(x)

    @staticmethod
    def backward(ctx, g):
        return bf16r(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 rounding in forward only (weights: the bf16 working copy feeds the kernels, the
    float32 weight gradient is not rounded)."""

    @staticmethod
    def forward(ctx, x):
        return bf16r(x)

    @staticmethod
    def backward(ctx, g):
        return g


def bottleneck_train_reference(b, state, x_nhwc, dout_nhwc, bn_eps: float = 1e-5):
    """Forward (batch statistics) + autograd backward of ONE bottleneck block on the CPU with
    bf16 rounding at the kernels' storage points.  x, dout: float32 NHWC holding bf16 values.
    Returns (out NHWC, dx NHWC, {name: grad in framework layout})."""
    rb, rw = _RoundBoth.apply, _RoundFwd.apply
    P = {}
    for c in (b.c1, b.c2, b.c3) + ((b.down,) if b.down else ()):
        for nm in (".weight", ".bn.gamma", ".bn.beta"):
            P[c.name + nm] = state[c.name + nm].clone().requires_grad_(True)

    def bn(c, z, residual=None, relu=True):
        y = F.batch_norm(z, None, None, P[c.name + ".bn.gamma"], P[c.name + ".bn.beta"],
                         training=True, momentum=0.0, eps=bn_eps)
        if residual is not None:
            y = y + residual
        return rb(torch.relu(y) if relu else y)

    def conv(c, t):
        return rb(F.conv2d(t, rw(P[c.name + ".weight"]), None, stride=c.stride, padding=c.pad))

    x = _nchw(x_nhwc).requires_grad_(True)
    o = bn(b.c1, conv(b.c1, x))
    o = bn(b.c2, conv(b.c2, o))
    z3 = conv(b.c3, o)
    idn = bn(b.down, conv(b.down, x), relu=False) if b.down else x
    out = bn(b.c3, z3, residual=idn, relu=True)
    out.backward(_nchw(dout_nhwc))
    return _nhwc(out.detach()), _nhwc(x.grad), {k: p.grad for k, p in P.items()}


def deconv_train_reference(d, state, x_nhwc, da_nhwc, bn_eps: float = 1e-5):
    """Same for one ConvTranspose(4x4, s2, p1) + BN + ReLU layer."""
    rb, rw = _RoundBoth.apply, _RoundFwd.apply
    P = {d.name + nm: state[d.name + nm].clone().requires_grad_(True)
         for nm in (".weight", ".bn.gamma", ".bn.beta")}
    x = _nchw(x_nhwc).requires_grad_(True)
    z = rb(F.conv_transpose2d(x, rw(P[d.name + ".weight"]), None, stride=2, padding=1))
    y = F.batch_norm(z, None, None, P[d.name + ".bn.gamma"], P[d.name + ".bn.beta"],
                     training=True, momentum=0.0, eps=bn_eps)
    a = rb(torch.relu(y))
    a.backward(_nchw(da_nhwc))
    return _nhwc(a.detach()), _nhwc(x.grad), {k: p.grad for k, p in P.items()}


def pose_resnet_train_reference(specs, state, img_u8, targets, weights, mask, mean, std,
                                bn_eps: float = 1e-5, round_bf16: bool = True,
                                frozen_bn: bool = False):
    """One forward/backward of the training graph on the CPU with torch autograd.

    targets [B,K,H,W], weights [B,K], mask [B,H,W] or None: as produced by the NumPy oracle
    (render_targets).  Loss = 0.5/(BKHW) * sum m (w (p - t))^2.  With round_bf16 the graph rounds
    to bf16 where the kernels do (weights, every stored activation and activation gradient);
    without it the graph is plain float32.  frozen_bn: every batch norm uses the state's running
    mean / variance as constants (eval-mode normalisation; gamma and beta still receive
    gradients).  Returns (loss, heatmaps, {name: grad}) with grads in framework layouts for every
    weight / gamma / beta / bias.
    """
    stem, blocks, deconvs, final = specs
    rb = _RoundBoth.apply if round_bf16 else (lambda t: t)
    rw = _RoundFwd.apply if round_bf16 else (lambda t: t)
    P = {}
    for k, v in state.items():
        if k.endswith(".bn.mean") or k.endswith(".bn.var"):
            continue
        P[k] = v.clone().double().float().requires_grad_(True)

    def bn(c, z, residual=None, relu=True):
        if frozen_bn:
            y = F.batch_norm(z, state[c.name + ".bn.mean"].float(), state[c.name + ".bn.var"].float(),
                             P[c.name + ".bn.gamma"], P[c.name + ".bn.beta"], training=False,
                             eps=bn_eps)
        else:
            y = F.batch_norm(z, None, None, P[c.name + ".bn.gamma"], P[c.name + ".bn.beta"],
                             training=True, momentum=0.0, eps=bn_eps)
        if residual is not None:
            y = y + residual
        if relu:
            y = torch.relu(y)
        return rb(y)

    def conv(c, x):
        return rb(F.conv2d(x, rw(P[c.name + ".weight"]), None, stride=c.stride, padding=c.pad))

    x = torch.from_numpy(preprocess_u8(img_u8, mean, std)).permute(0, 3, 1, 2).contiguous()
    x = bn(stem, conv(stem, x))
    x = F.max_pool2d(x, 3, 2, 1)
    for b in blocks:
        o = bn(b.c1, conv(b.c1, x))
        o = bn(b.c2, conv(b.c2, o))
        z3 = conv(b.c3, o)
        idn = bn(b.down, conv(b.down, x), relu=False) if b.down else x
        x = bn(b.c3, z3, residual=idn, relu=True)
    for d in deconvs:
        z = rb(F.conv_transpose2d(x, rw(P[d.name + ".weight"]), None, stride=2, padding=1))
        x = bn(d, z)
    hm = F.conv2d(x, rw(P[final.name + ".weight"]), P[final.name + ".bias"])
    t = torch.as_tensor(targets)
    w = torch.as_tensor(weights)[:, :, None, None]
    e = (w * (hm - t)) ** 2
    if mask is not None:
        e = e * torch.as_tensor(mask).bool()[:, None].float()
    loss = 0.5 * e.mean()
    loss.backward()
    return loss.item(), hm.detach(), {k: p.grad for k, p in P.items()}


def _hrnet_module_autograd(m, xs, unit, rb):
    """One exchange module (branches, then fusion) in NCHW autograd tensors."""
    def up(t, f):
        return t.repeat_interleave(1 << f, dim=2).repeat_interleave(1 << f, dim=3)

    ys = []
    for x, br in zip(xs, m.branches):
        for blk in br:
            x = unit(blk.c2, unit(blk.c1, x), residual=x, relu=True)
        ys.append(x)
    outs = []
    for i, row in enumerate(m.fuse):
        lows = [j for j in range(len(row)) if j > i]
        s = ys[i]
        for j in range(i):
            t = ys[j]
            for u in row[j][:-1]:
                t = unit(u, t)
            s = unit(row[j][-1], t, residual=s, relu=(j == i - 1 and not lows))
        if lows:
            for j in lows:
                s = s + up(unit(row[j][0], ys[j]), j - i)
            s = rb(torch.relu(s))
        outs.append(s)
    return outs


def hrnet_module_train_reference(m, state, xs_nhwc, douts_nhwc, bn_eps: float = 1e-5,
                                 acc64: bool = False):
    """Forward (batch statistics) + autograd backward of ONE exchange module on the CPU with
    bf16 rounding at the kernels' storage points.  xs / douts: float32 NHWC lists (one per
    branch / per fusion output) holding bf16 values.
    acc64: the convolutions (forward and, through autograd, backward) accumulate in float64:
    a second accumulation order over the same operands and rounding points, whose distance
    from the float32 run measures how far summation order alone moves the results.
    Returns ([out NHWC], [dx NHWC], {name: grad in framework layout})."""
    rb, rw = _RoundBoth.apply, _RoundFwd.apply
    P = {}

    def unit(u, x, residual=None, relu=None):
        for nm in (".weight", ".bn.gamma", ".bn.beta"):
            if u.name + nm not in P:
                P[u.name + nm] = state[u.name + nm].clone().float().requires_grad_(True)
        w = rw(P[u.name + ".weight"])
        if acc64:
            z = rb(F.conv2d(x.double(), w.double(), None, stride=u.stride,
                            padding=u.k // 2).float())
        else:
            z = rb(F.conv2d(x, w, None, stride=u.stride, padding=u.k // 2))
        y = F.batch_norm(z, None, None, P[u.name + ".bn.gamma"], P[u.name + ".bn.beta"],
                         training=True, momentum=0.0, eps=bn_eps)
        if residual is not None:
            y = y + residual
        if u.relu if relu is None else relu:
            y = torch.relu(y)
        return rb(y)

    xs = [_nchw(x).requires_grad_(True) for x in xs_nhwc]
    outs = _hrnet_module_autograd(m, xs, unit, rb)
    torch.autograd.backward(outs, [_nchw(d) for d in douts_nhwc])
    return ([_nhwc(o.detach()) for o in outs], [_nhwc(x.grad) for x in xs],
            {k: p.grad for k, p in P.items()})


def hrnet_train_reference(spec, state, img_u8, targets, weights, mask, mean, std,
                          bn_eps: float = 1e-5, round_bf16: bool = True, x_nhwc=None,
                          dout_nchw=None):
    """One forward/backward of the HRNet-style training graph on the CPU with torch autograd:
    batch-statistics BN after every conv, the same branch / exchange wiring as
    hrnet_pose_forward, bf16 rounding (round_bf16) where the kernels store bf16 (weights, conv
    outputs, activations, fusion outputs and the gradients of all of those).

    Default: image -> heat maps -> masked MSE (as pose_resnet_train_reference); returns
    (loss, heatmaps, {name: grad}).  With x_nhwc and dout_nchw given, the stem is skipped: the
    graph starts at layer1 from x_nhwc and the heat maps are seeded with dout_nchw instead of a
    loss (loss is returned as 0)."""
    rb = _RoundBoth.apply if round_bf16 else (lambda t: t)
    rw = _RoundFwd.apply if round_bf16 else (lambda t: t)
    P = {}
    for k, v in state.items():
        if k.endswith(".bn.mean") or k.endswith(".bn.var"):
            continue
        P[k] = v.clone().float().requires_grad_(True)

    def unit(u, x, residual=None, relu=None):
        z = rb(F.conv2d(x, rw(P[u.name + ".weight"]), None, stride=u.stride, padding=u.k // 2))
        y = F.batch_norm(z, None, None, P[u.name + ".bn.gamma"], P[u.name + ".bn.beta"],
                         training=True, momentum=0.0, eps=bn_eps)
        if residual is not None:
            y = y + residual
        if u.relu if relu is None else relu:
            y = torch.relu(y)
        return rb(y)

    if x_nhwc is None:
        x = torch.from_numpy(preprocess_u8(img_u8, mean, std)).permute(0, 3, 1, 2).contiguous()
        for u in spec.stem:
            x = unit(u, x)
    else:
        x = _nchw(x_nhwc)
    for b in spec.layer1:
        idn = unit(b.down, x) if b.down else x
        x = unit(b.c3, unit(b.c2, unit(b.c1, x)), residual=idn, relu=True)
    xs = [x]
    for tr, mods in zip(spec.transitions, spec.stages):
        xs = [(xs[i] if t is None else unit(t, xs[i])) if i < len(xs) else unit(t, xs[-1])
              for i, t in enumerate(tr)]
        for m in mods:
            xs = _hrnet_module_autograd(m, xs, unit, rb)
    hm = F.conv2d(xs[0], rw(P["final.weight"]), P["final.bias"])
    if dout_nchw is not None:
        hm.backward(dout_nchw)
        return 0.0, hm.detach(), {k: p.grad for k, p in P.items()}
    t = torch.as_tensor(targets)
    w = torch.as_tensor(weights)[:, :, None, None]
    e = (w * (hm - t)) ** 2
    if mask is not None:
        e = e * torch.as_tensor(mask).bool()[:, None].float()
    loss = 0.5 * e.mean()
    loss.backward()
    return loss.item(), hm.detach(), {k: p.grad for k, p in P.items()}


# ----------------------------------------------------------------------------------------------
# MobileNetV2 training: inverted-residual blocks with batch-statistics BN and ReLU6.
# ----------------------------------------------------------------------------------------------
class _Relu6Stored(torch.autograd.Function):
    """ReLU6 whose output is stored in bf16 and whose backward masks with that stored value
    (0 < a < 6), as a kernel that keeps only the bf16 activation must; the gradient is rounded
    to bf16 as well."""

    @staticmethod
    def forward(ctx, y):
        a = bf16r(torch.clamp(y, 0.0, 6.0))
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return bf16r(g) * ((a > 0) & (a < 6)).to(g.dtype)


def _mb_block_autograd(b, P, x, round_bf16: bool, bn_eps: float):
    rb = _RoundBoth.apply if round_bf16 else (lambda t: t)
    rw = _RoundFwd.apply if round_bf16 else (lambda t: t)

    def bn(name, z, relu6, residual=None):
        y = F.batch_norm(z, None, None, P[name + ".bn.gamma"], P[name + ".bn.beta"],
                         training=True, momentum=0.0, eps=bn_eps)
        if residual is not None:
            y = y + residual
        if not relu6:
            return rb(y)
        return _Relu6Stored.apply(y) if round_bf16 else torch.clamp(y, 0.0, 6.0)

    h = x
    if b.expand != 1:
        n = b.name + ".expand"
        h = bn(n, rb(F.conv2d(h, rw(P[n + ".weight"]))), True)
    n = b.name + ".dw"
    h = bn(n, rb(F.conv2d(h, rw(P[n + ".weight"]), None, stride=b.stride, padding=b.dilation,
                          dilation=b.dilation, groups=b.mid)), True)
    n = b.name + ".project"
    return bn(n, rb(F.conv2d(h, rw(P[n + ".weight"]))), False, residual=x if b.residual else None)


def _grad_params(state):
    return {k: v.clone().float().requires_grad_(True) for k, v in state.items()
            if not (k.endswith(".bn.mean") or k.endswith(".bn.var"))}


def mobilenet_block_train_reference(b, state, x_nhwc, dout_nhwc, bn_eps: float = 1e-5):
    """Forward (batch statistics) + autograd backward of ONE inverted-residual block on the CPU
    with bf16 rounding at the kernels' storage points.  Returns (out NHWC, dx NHWC, {name: grad})."""
    P = {k: v for k, v in _grad_params(state).items() if k.startswith(b.name + ".")}
    x = _nchw(x_nhwc).requires_grad_(True)
    out = _mb_block_autograd(b, P, x, True, bn_eps)
    out.backward(_nchw(dout_nhwc))
    return _nhwc(out.detach()), _nhwc(x.grad), {k: p.grad for k, p in P.items() if p.grad is not None}


def mobilenet_train_reference(blocks, state, img_u8, mean, std, dheat, dvec=None,
                              bn_eps: float = 1e-5, round_bf16: bool = True, det_blocks=None,
                              dboxes=None):
    """Forward with batch statistics and autograd backward of the whole MobileNetV2 pose network,
    the heads seeded with given gradients (float32 [B,K,H,W] / [B,4P,H,W]) instead of a loss.
    Returns (heat, vec or None, {name: grad in framework layout}).
    det_blocks / dboxes: the detection branch (one stride-2 block + one 1x1 box head per level on
    top of the trunk output) and the seeds of its heads; the box-head maps are then returned as
    a fourth value."""
    rb = _RoundBoth.apply if round_bf16 else (lambda t: t)
    rw = _RoundFwd.apply if round_bf16 else (lambda t: t)
    P = _grad_params(state)
    x = torch.from_numpy(preprocess_u8(img_u8, mean, std)).permute(0, 3, 1, 2).contiguous()
    z = rb(F.conv2d(x, rw(P["stem.weight"]), None, stride=2, padding=1))
    y = F.batch_norm(z, None, None, P["stem.bn.gamma"], P["stem.bn.beta"], training=True,
                     momentum=0.0, eps=bn_eps)
    x = _Relu6Stored.apply(y) if round_bf16 else torch.clamp(y, 0.0, 6.0)
    for b in blocks:
        x = _mb_block_autograd(b, P, x, round_bf16, bn_eps)
    heat = F.conv2d(x, rw(P["heat.weight"]), P["heat.bias"])
    outs, seeds = [heat], [torch.as_tensor(dheat)]
    vec = None
    if dvec is not None:
        vec = F.conv2d(x, rw(P["vec.weight"]), P["vec.bias"])
        outs.append(vec)
        seeds.append(torch.as_tensor(dvec))
    boxes = []
    if det_blocks:
        f = x
        for i, b in enumerate(det_blocks):
            f = _mb_block_autograd(b, P, f, round_bf16, bn_eps)
            m = F.conv2d(f, rw(P[f"box{i + 1}.weight"]), P[f"box{i + 1}.bias"])
            boxes.append(m)
            outs.append(m)
            seeds.append(torch.as_tensor(dboxes[i]))
    torch.autograd.backward(outs, seeds)
    grads = {k: p.grad for k, p in P.items()}
    vec_out = None if vec is None else vec.detach()
    if det_blocks:
        return heat.detach(), vec_out, grads, [m.detach() for m in boxes]
    return heat.detach(), vec_out, grads
