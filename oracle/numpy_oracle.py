"""NumPy restatement of the bandwidth-bound units of the pose hot path.

TEST INFRASTRUCTURE ONLY (see the package docstring): the product never
imports this file.  PARITY UNPINNED against the reference's implementation,
because the reference snapshot holds none; the absent homes of these units
are named by PoseEstimation.pyproj:28,30,34,37 (inference.py,
model/base_model.py, train.py, utils/ops.py).  What the reference does pin is
tested in tests/test_reference_fixture.py.

Every function fixes the *order of float32 operations* so that the HIP kernels,
which follow the same order with FMA contraction disabled, can be compared
bit for bit wherever the arithmetic is elementwise.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even on the float32 bit pattern; finite data)
# --------------------------------------------------------------------------
def f32_to_bf16_bits(x) -> np.ndarray:
    """float32 -> bf16 bit pattern (uint16), round-to-nearest-even.  NaN stays NaN."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        r = np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)
    return r


def bf16_bits_to_f32(b) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(F32)


def bf16_round(x) -> np.ndarray:
    """float32 value rounded to the nearest bf16, returned as float32."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# --------------------------------------------------------------------------
# a1  heat-map peak decode
# --------------------------------------------------------------------------
def decode_heatmaps(hm, refine: bool = True):
    """Per (b, k) map: first arg-max, confidence, quarter-pixel refined (x, y).

    hm: logical [B, K, H, W] float32 with any strides (an NHWC array viewed
    through .transpose(0, 3, 1, 2) is fine).  Finite inputs assumed.

    idx  = lowest flat index y*W + x of the maximum (numpy.argmax semantics)
    conf = that maximum
    x, y = idx % W, idx // W; when 1 < x < W-1 and 1 < y < H-1 each coordinate
           moves 0.25 towards the larger of its two neighbours (no move on a
           tie); when conf <= 0 both coordinates are reported as 0.
    The reference would hold this in inference.py / utils/ops.py
    (PoseEstimation.pyproj:28,37; absent); its notebook only ever calls
    heatmaps[:, :, kp].nonzero() (test.ipynb:931).
    """
    hm = np.asarray(hm, dtype=F32)
    B, K, H, W = hm.shape
    flat = np.ascontiguousarray(hm).reshape(B, K, H * W)
    idx64 = flat.argmax(axis=2)
    conf = np.take_along_axis(flat, idx64[..., None], axis=2)[..., 0].astype(F32)
    x = idx64 % W
    y = idx64 // W
    xf = x.astype(F32)
    yf = y.astype(F32)
    if refine and H >= 4 and W >= 4:
        inner = (x > 1) & (x < W - 1) & (y > 1) & (y < H - 1)
        xc = np.clip(x, 1, W - 2)
        yc = np.clip(y, 1, H - 2)
        bi = np.arange(B)[:, None]
        ki = np.arange(K)[None, :]
        full = np.ascontiguousarray(hm)
        dx = full[bi, ki, yc, xc + 1] - full[bi, ki, yc, xc - 1]
        dy = full[bi, ki, yc + 1, xc] - full[bi, ki, yc - 1, xc]
        xf = xf + np.where(inner, F32(0.25) * np.sign(dx), F32(0)).astype(F32)
        yf = yf + np.where(inner, F32(0.25) * np.sign(dy), F32(0)).astype(F32)
    pos = conf > 0
    xy = np.stack([np.where(pos, xf, F32(0)), np.where(pos, yf, F32(0))], axis=-1).astype(F32)
    return xy, conf, idx64.astype(np.int32)


def decode_heatmaps_nan_skipped(hm, refine: bool = True):
    """The decode rule for maps that hold NaN, where decode_heatmaps (numpy.argmax: the first
    NaN wins) and an ordered compare part ways.  This is the rule pe_decode_f32 documents:

      * a NaN never wins: idx is the lowest flat index of the largest non-NaN element that is
        above -inf; a map with no such element (all NaN and/or -inf) answers idx 0 and
        conf = hm[0] (which may itself be NaN), coordinates 0;
      * in the quarter-pixel step a NaN neighbour is neither greater nor less than the other
        one, so that axis does not move;
      * conf <= 0 or NaN conf zeroes the coordinates.

    Plain Python loops: for small test cases only.  On maps without NaN and without +-inf
    neighbours it equals decode_heatmaps (tests/test_oracle.py)."""
    hm = np.asarray(hm, dtype=F32)
    B, K, H, W = hm.shape
    xy = np.zeros((B, K, 2), F32)
    conf = np.zeros((B, K), F32)
    idx = np.zeros((B, K), np.int32)
    for b in range(B):
        for k in range(K):
            m = hm[b, k]
            best, bi = F32(-np.inf), -1
            for i, v in enumerate(m.reshape(-1)):
                if v > best:           # False for NaN; strict, so the first maximum is kept
                    best, bi = v, i
            if bi < 0:
                bi, best = 0, m.reshape(-1)[0]
            y, x = divmod(bi, W)
            fx, fy = F32(x), F32(y)
            if refine and 1 < x < W - 1 and 1 < y < H - 1:
                r, l, d, u = m[y, x + 1], m[y, x - 1], m[y + 1, x], m[y - 1, x]
                fx = F32(fx + F32(0.25) * F32(int(r > l) - int(r < l)))
                fy = F32(fy + F32(0.25) * F32(int(d > u) - int(d < u)))
            if not (best > 0):
                fx = fy = F32(0)
            xy[b, k] = (fx, fy)
            conf[b, k] = best
            idx[b, k] = bi
    return xy, conf, idx


# --------------------------------------------------------------------------
# a2  Gaussian target rendering
# --------------------------------------------------------------------------
def gaussian_table(sigma: float, radius: int) -> np.ndarray:
    """g[i] = float32(exp(-i^2 / (2 sigma^2))) for i = 0..radius, exp in float64.

    sigma is first rounded to float32 (it crosses the C ABI as a float).
    """
    s = float(F32(sigma))
    i = np.arange(int(radius) + 1, dtype=np.float64)
    return np.exp(-(i * i) / (2.0 * s * s)).astype(F32)


def joint_centres(joints, stride: float):
    """mu = floor(coord / stride + 0.5), evaluated in float32."""
    j = np.asarray(joints, dtype=F32)
    st = F32(stride)
    mx = np.floor(j[..., 0] / st + F32(0.5)).astype(F32)
    my = np.floor(j[..., 1] / st + F32(0.5)).astype(F32)
    return mx, my


def render_targets(joints, H: int, W: int, stride: float, sigma: float, radius: int):
    """Gaussian heat-map targets and per-joint weights.

    joints [B, K, 3] float32 = (x, y, v) in input pixels.  Returns
    target [B, K, H, W] float32 and weight [B, K] float32.

    weight = 0 when v == 0 (or v is not > 0), when a coordinate is not finite,
    or when the centre cell mu lies off the map; else 1.  (The reference's label
    of test.ipynb:606-653 has a keypoint whose centre is one column past the
    right edge: its vector-map entries exist in the clipped window but its heat
    map is empty, so a joint is drawn only when its centre cell is on the map.)
    target = g[|x - mu_x|] * g[|y - mu_y|] (one float32 product) inside the
    window, 0 elsewhere, and all zeros when weight == 0.
    With radius = 1 the support is the 3x3 block the reference's label maps
    show (test.ipynb:606-614).  The reference would hold this in
    dataset/pose_data.py / utils/ops.py (PoseEstimation.pyproj:27,37; absent).
    """
    j = np.asarray(joints, dtype=F32)
    B, K, _ = j.shape
    r = int(radius)
    g = gaussian_table(sigma, r)
    mx, my = joint_centres(j, stride)
    finite = np.isfinite(j[..., 0]) & np.isfinite(j[..., 1])
    with np.errstate(invalid="ignore"):
        on = (mx < W) & (mx >= 0) & (my < H) & (my >= 0)
        vis = j[..., 2] > 0
    wgt = (finite & on & vis)
    target = np.zeros((B, K, H, W), dtype=F32)
    xs = np.arange(W)
    ys = np.arange(H)
    for b in range(B):
        for k in range(K):
            if not wgt[b, k]:
                continue
            cx = int(mx[b, k])
            cy = int(my[b, k])
            ax = np.abs(xs - cx)
            ay = np.abs(ys - cy)
            gx = np.where(ax <= r, g[np.minimum(ax, r)], F32(0)).astype(F32)
            gy = np.where(ay <= r, g[np.minimum(ay, r)], F32(0)).astype(F32)
            target[b, k] = gy[:, None] * gx[None, :]
    return target, wgt.astype(F32)


# --------------------------------------------------------------------------
# input pipeline: affine resample (resize / scale / crop / flip) + normalise
# --------------------------------------------------------------------------
def augment_u8(img_u8, params, out_hw, mean, std):
    """uint8 [B,Hs,Ws,3], params [B,5] = (sx, sy, tx, ty, flip) -> float32 [B,H,W,3] holding
    bf16-representable values (the kernel writes the same values in space-to-depth order).

    Output pixel (x, y) reads source (u, v) = ((xf+.5)*sx + tx - .5, (y+.5)*sy + ty - .5) with
    xf = W-1-x when flipped, bilinear: top = p00 + fx*(p01-p00), bot likewise, val = top +
    fy*(bot-top); taps outside the frame are 0; then (val/255 - mean) * (1/std).  Every step is
    one float32 operation in this order.  Follows the configuration the reference prints
    (test.ipynb:995-1010); its own resampling code (dataset/pose_data.py) is absent.
    """
    img = np.asarray(img_u8)
    B, Hs, Ws, _ = img.shape
    H, W = out_hw
    prm = np.asarray(params, dtype=F32)
    m = np.asarray(mean, F32)
    inv = (F32(1.0) / np.asarray(std, F32)).astype(F32)
    out = np.zeros((B, H, W, 3), F32)
    ys = np.arange(H, dtype=F32)
    xs = np.arange(W, dtype=F32)
    for b in range(B):
        sx, sy, tx, ty, flip = (F32(v) for v in prm[b])
        xf = (F32(W - 1) - xs) if flip != 0 else xs
        u = (((xf + F32(0.5)) * sx).astype(F32) + tx).astype(F32) - F32(0.5)
        v = (((ys + F32(0.5)) * sy).astype(F32) + ty).astype(F32) - F32(0.5)
        uf, vf = np.floor(u), np.floor(v)
        fx = (u - uf).astype(F32)[None, :, None]
        fy = (v - vf).astype(F32)[:, None, None]
        x0 = np.clip(uf, -1e9, 1e9).astype(np.int64)
        y0 = np.clip(vf, -1e9, 1e9).astype(np.int64)

        def tap(yy, xx):
            ok = ((yy >= 0) & (yy < Hs))[:, None] & ((xx >= 0) & (xx < Ws))[None, :]
            g = img[b][np.clip(yy, 0, Hs - 1)[:, None], np.clip(xx, 0, Ws - 1)[None, :]].astype(F32)
            return np.where(ok[..., None], g, F32(0))

        p00, p01 = tap(y0, x0), tap(y0, x0 + 1)
        p10, p11 = tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
        top = (p00 + (fx * (p01 - p00)).astype(F32)).astype(F32)
        bot = (p10 + (fx * (p11 - p10)).astype(F32)).astype(F32)
        val = (top + (fy * (bot - top)).astype(F32)).astype(F32)
        out[b] = (((val / F32(255.0)).astype(F32) - m).astype(F32) * inv).astype(F32)
    return bf16_round(out)


def augment_mask(mask, params, src_stride: float, out_hw, stride: float, fill: int = 1):
    """Ignore mask [B,Hms,Wms] over the source frame (cells of src_stride pixels) -> uint8
    [B,Hm,Wm] over the output image (cells of `stride` pixels) under augment_u8's map, nearest
    sampling at the cell centre: xc = (x+.5)*stride - .5 (mirrored to (Wm*stride - 1) - xc when
    flipped), u = (xc+.5)*sx + tx - .5, source cell floor((u+.5)/src_stride); outside -> fill.
    float32, one operation at a time.  The reference's `masks` label (test.ipynb:949) went through
    the same tf.data map as the image; its code (dataset/pose_data.py) is absent."""
    m = np.asarray(mask).astype(np.uint8)
    prm = np.asarray(params, dtype=F32)
    B, Hms, Wms = m.shape
    Hm, Wm = out_hw
    st, ss = F32(stride), F32(src_stride)
    out = np.empty((B, Hm, Wm), np.uint8)
    xs = np.arange(Wm, dtype=F32)
    ys = np.arange(Hm, dtype=F32)
    Wpx = (F32(Wm) * st).astype(F32)
    for b in range(B):
        sx, sy, tx, ty, flip = (F32(v) for v in prm[b])
        xc = ((xs + F32(0.5)) * st).astype(F32) - F32(0.5)
        yc = ((ys + F32(0.5)) * st).astype(F32) - F32(0.5)
        xf = ((Wpx - F32(1.0)).astype(F32) - xc).astype(F32) if flip != 0 else xc
        u = (((xf + F32(0.5)) * sx).astype(F32) + tx).astype(F32) - F32(0.5)
        v = (((yc + F32(0.5)) * sy).astype(F32) + ty).astype(F32) - F32(0.5)
        cu = np.floor(((u + F32(0.5)).astype(F32) / ss).astype(F32))
        cv = np.floor(((v + F32(0.5)).astype(F32) / ss).astype(F32))
        okx = (cu >= 0) & (cu < Wms)
        oky = (cv >= 0) & (cv < Hms)
        ix = np.clip(cu, 0, Wms - 1).astype(np.int64)
        iy = np.clip(cv, 0, Hms - 1).astype(np.int64)
        g = m[b][iy[:, None], ix[None, :]]
        out[b] = np.where(oky[:, None] & okx[None, :], g, np.uint8(fill))
    return out


def transform_joints(joints, params, out_w: int, flip_perm=None):
    """Inverse of augment_u8's map for joints [B,K,3]: xo = (x + .5 - tx)/sx - .5 (W-1-xo when
    flipped), yo likewise; when flipped, output joint k is source joint flip_perm[k]."""
    j = np.asarray(joints, dtype=F32)
    prm = np.asarray(params, dtype=F32)
    B, K, _ = j.shape
    out = np.empty_like(j)
    for b in range(B):
        sx, sy, tx, ty, flip = (F32(v) for v in prm[b])
        src = j[b][list(flip_perm)] if (flip != 0 and flip_perm is not None) else j[b]
        xo = (((src[:, 0] + F32(0.5)) - tx).astype(F32) / sx).astype(F32) - F32(0.5)
        yo = (((src[:, 1] + F32(0.5)) - ty).astype(F32) / sy).astype(F32) - F32(0.5)
        if flip != 0:
            xo = (F32(out_w - 1) - xo).astype(F32)
        out[b, :, 0], out[b, :, 1], out[b, :, 2] = xo, yo, src[:, 2]
    return out


# --------------------------------------------------------------------------
# vector maps: per keypoint pair, the integer cell offset to the partner joint
# --------------------------------------------------------------------------
VEC_MAX_CENTRE = 1 << 20  # centres further out than this are treated as unlabelled


def render_vecmaps(joints, pairs, H: int, W: int, stride: float, radius: int):
    """Vector-map labels.  joints [B, K, 3] float32 (x, y, v); pairs = [(k1, k2), ...].

    Returns vec [B, 4*P, H, W] float32 and support (bool, same shape).  For pair
    i = (k1, k2) with centres c1, c2 (joint_centres):
      channel 4i   at cells of k1's window: c2.x - x     channel 4i+1: c2.y - y
      channel 4i+2 at cells of k2's window: c1.x - x     channel 4i+3: c1.y - y
    The window is the (2r+1)^2 block around the joint's centre clipped to the
    map; a centre may lie off the map (the entries in the clipped window stay,
    and offsets still point at the off-map centre).  Both joints must be
    labelled (v > 0) and finite, else all four channels are empty.
    Follows the reference's label printed at test.ipynb:634-653 (8 channels for
    pairs [[0, 1], [2, 1]], 3x3 windows, channel use read off the zero pattern:
    an offset component that is 0 is missing from the non-zero list) and the
    channel indexing of its drawing cell (test.ipynb: vecmaps[y, x, 4*i],
    vecmaps[y, x, 4*i + 1] scaled by the stride give the line to the partner).
    """
    j = np.asarray(joints, dtype=F32)
    B, K, _ = j.shape
    r = int(radius)
    P = len(pairs)
    mx, my = joint_centres(j, stride)
    with np.errstate(invalid="ignore"):
        ok = (np.isfinite(j[..., 0]) & np.isfinite(j[..., 1]) & (j[..., 2] > 0)
              & (np.abs(mx) < VEC_MAX_CENTRE) & (np.abs(my) < VEC_MAX_CENTRE))
    vec = np.zeros((B, 4 * P, H, W), dtype=F32)
    sup = np.zeros((B, 4 * P, H, W), dtype=bool)
    xs = np.arange(W, dtype=np.int64)
    ys = np.arange(H, dtype=np.int64)
    for b in range(B):
        for i, (k1, k2) in enumerate(pairs):
            if not (ok[b, k1] and ok[b, k2]):
                continue
            for half, (ks, kd) in enumerate(((k1, k2), (k2, k1))):
                sx, sy = int(mx[b, ks]), int(my[b, ks])
                dx, dy = int(mx[b, kd]), int(my[b, kd])
                win = (np.abs(ys - sy) <= r)[:, None] & (np.abs(xs - sx) <= r)[None, :]
                c = 4 * i + 2 * half
                vec[b, c] = np.where(win, (dx - xs)[None, :], 0).astype(F32)
                vec[b, c + 1] = np.where(win, (dy - ys)[:, None], 0).astype(F32)
                sup[b, c] = win
                sup[b, c + 1] = win
    return vec, sup


def vecmap_loss_fwd_bwd(pred, joints, pairs, stride: float, radius: int, mask=None,
                        pred_is_bf16: bool = False):
    """L = 0.5/n * sum over supported cells of m (p - t)^2 and dL/dp, n = B*4P*H*W.

    pred [B, 4P, H, W] float32 (or bf16 bits); t and the support come from
    render_vecmaps; cells outside the support carry no loss and no gradient
    (the offset is undefined there).  Returns loss, per_channel [4P] (mean of
    per_channel == loss) and grad.  Float32 order: d = p - t; q = d*d;
    grad = inv_n * d with inv_n = float32(1/n).
    """
    if pred_is_bf16:
        p = bf16_bits_to_f32(pred)
    else:
        p = np.asarray(pred, dtype=F32)
    B, C, H, W = p.shape
    t, sup = render_vecmaps(joints, pairs, H, W, stride, radius)
    n = B * C * H * W
    inv_n = F32(1.0 / n)
    d = (p - t).astype(F32)
    q = (d * d).astype(F32)
    g = (inv_n * d).astype(F32)
    keep = sup
    if mask is not None:
        keep = keep & np.asarray(mask).astype(bool)[:, None, :, :]
    q = np.where(keep, q, F32(0))
    g = np.where(keep, g, F32(0)).astype(F32)
    sums = q.astype(np.float64).sum(axis=(0, 2, 3))
    per_channel = (sums * (0.5 / (B * H * W))).astype(F32)
    loss = F32(sums.sum() * (0.5 / n))
    grad = f32_to_bf16_bits(g) if pred_is_bf16 else g
    return loss, per_channel, grad


# --------------------------------------------------------------------------
# a3  masked per-joint MSE loss and gradient
# --------------------------------------------------------------------------
def mse_fwd_bwd(pred, target, weight, mask=None, pred_is_bf16: bool = False):
    """L = 0.5/(B K H W) * sum m (w (p - t))^2 and dL/dp.

    pred   [B, K, H, W] float32, or uint16 bf16 bit patterns when pred_is_bf16
    target [B, K, H, W] float32
    weight [B, K] float32
    mask   [B, H, W] uint8 (0 = ignore) or None
    Returns loss (float32 scalar), per_joint [K] float32 with
    mean(per_joint) == loss, and grad (float32, or bf16 bits when
    pred_is_bf16).

    Float32 order: d = p - t; e = w*d; q = e*e (summed in float64 here; the
    kernels sum in float32 trees, so the loss is compared with a tolerance);
    c = (w*w)*inv_n with inv_n = float32(1/(B K H W)); grad = c*d where m else 0
    (elementwise, compared bit for bit).
    The reference would hold the loss in model/base_model.py
    (PoseEstimation.pyproj:30; absent); a per-pixel `masks` label exists at
    test.ipynb:949.
    """
    if pred_is_bf16:
        p = bf16_bits_to_f32(pred)
    else:
        p = np.asarray(pred, dtype=F32)
    t = np.asarray(target, dtype=F32)
    w = np.asarray(weight, dtype=F32)
    B, K, H, W = p.shape
    n = B * K * H * W
    inv_n = F32(1.0 / n)
    d = (p - t).astype(F32)
    e = (w[:, :, None, None] * d).astype(F32)
    q = (e * e).astype(F32)
    c = ((w * w).astype(F32) * inv_n).astype(F32)
    g = (c[:, :, None, None] * d).astype(F32)
    if mask is not None:
        m = np.asarray(mask).astype(bool)[:, None, :, :]
        q = np.where(m, q, F32(0))
        g = np.where(m, g, F32(0)).astype(F32)
    sums = q.astype(np.float64).sum(axis=(0, 2, 3))
    per_joint = (sums * (0.5 / (B * H * W))).astype(F32)
    loss = F32(sums.sum() * (0.5 / n))
    grad = f32_to_bf16_bits(g) if pred_is_bf16 else g
    return loss, per_joint, grad


# --------------------------------------------------------------------------
# a6  optimizer steps over a flat float32 arena
# --------------------------------------------------------------------------
def adam_scalars(lr: float, beta1: float, beta2: float, step: int, tf1: bool = True):
    """Host-side scalars (s, a) for the update p -= s * m / (sqrt(v) * a + eps).

    tf1=True  : s = lr*sqrt(1-b2^t)/(1-b1^t), a = 1   (tf.train.AdamOptimizer,
                the reference's optimizer by name: test.ipynb:1007-1008)
    tf1=False : s = lr/(1-b1^t), a = 1/sqrt(1-b2^t)   (eps on sqrt(v_hat))
    """
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if tf1:
        return F32(lr * np.sqrt(bc2) / bc1), F32(1.0)
    return F32(lr / bc1), F32(1.0 / np.sqrt(bc2))


def adam_step(p, m, v, g, lr, beta1, beta2, eps, step, tf1=True, grad_scale=1.0):
    """One Adam step, float32 elementwise in a fixed order.  Returns new
    (p, m, v, p_bf16_bits)."""
    p = np.asarray(p, dtype=F32)
    m = np.asarray(m, dtype=F32)
    v = np.asarray(v, dtype=F32)
    g = (np.asarray(g, dtype=F32) * F32(grad_scale)).astype(F32)
    b1 = F32(beta1)
    b2 = F32(beta2)
    omb1 = F32(1.0 - float(b1))  # formed from the float32 betas, as the C ABI receives them
    omb2 = F32(1.0 - float(b2))
    s, a = adam_scalars(lr, beta1, beta2, step, tf1)
    m2 = ((b1 * m).astype(F32) + (omb1 * g).astype(F32)).astype(F32)
    v2 = ((b2 * v).astype(F32) + ((omb2 * g).astype(F32) * g).astype(F32)).astype(F32)
    den = ((np.sqrt(v2).astype(F32) * a).astype(F32) + F32(eps)).astype(F32)
    upd = ((s * m2).astype(F32) / den).astype(F32)
    p2 = (p - upd).astype(F32)
    return p2, m2, v2, f32_to_bf16_bits(p2)


def sgd_step(p, mom, g, lr, momentum, grad_scale=1.0):
    """Momentum SGD: mom = momentum*mom + g; p -= lr*mom (float32, fixed order)."""
    p = np.asarray(p, dtype=F32)
    mom = np.asarray(mom, dtype=F32)
    g = (np.asarray(g, dtype=F32) * F32(grad_scale)).astype(F32)
    mom2 = ((F32(momentum) * mom).astype(F32) + g).astype(F32)
    p2 = (p - (F32(lr) * mom2).astype(F32)).astype(F32)
    return p2, mom2, f32_to_bf16_bits(p2)


# --------------------------------------------------------------------------- detection branch
def ssd_anchors(levels=((20, 20), (10, 10)), scales=(0.35, 0.75), ratios=(0.5, 1.0, 2.0)):
    """Anchor boxes (ymin, xmin, ymax, xmax) in normalised image coordinates, float32 [A, 4].

    Order: level by level, then row, column, aspect ratio, which is the [20, 20, 3] + [10, 10, 3]
    = 1500 layout the reference's labels are reshaped with (test.ipynb:423-464).  Box height is
    scale / sqrt(ratio) and width scale * sqrt(ratio), centred on the cell centre.  The scale and
    ratio values themselves are not visible in the reference.
    """
    out = []
    for (fh, fw), s in zip(levels, scales):
        for y in range(fh):
            for x in range(fw):
                cy = F32((F32(y) + F32(0.5)) / F32(fh))
                cx = F32((F32(x) + F32(0.5)) / F32(fw))
                for r in ratios:
                    sq = F32(np.sqrt(F32(r)))
                    hh = F32(F32(s) / sq) * F32(0.5)
                    hw = F32(F32(s) * sq) * F32(0.5)
                    out.append((cy - hh, cx - hw, cy + hh, cx + hw))
    return np.asarray(out, dtype=F32)


def boxes_from_joints(joints, H: int, W: int, margin: float = 0.05):
    """One person box per image from its labelled keypoints (v > 0): normalised corners
    (ymin, xmin, ymax, xmax) = clamp(extent / size -+ margin, 0, 1), the format the reference's
    `regs` rows print (test.ipynb:571-572); all zeros and class 0 when nothing is labelled.
    The reference read its boxes from annotation files (dataset/pose_data.py,
    PoseEstimation.pyproj:27; absent): this stands in for them on synthetic batches.
    joints [B,K,3] float32 -> (boxes [B,1,4] float32, box_classes [B,1] int32)."""
    j = np.asarray(joints, dtype=F32)
    B = j.shape[0]
    boxes = np.zeros((B, 1, 4), F32)
    cls = np.zeros((B, 1), np.int32)
    m, h, w = F32(margin), F32(H), F32(W)
    for b in range(B):
        vis = j[b, :, 2] > 0
        if not vis.any():
            continue
        x, y = j[b, vis, 0], j[b, vis, 1]
        box = np.array([y.min() / h - m, x.min() / w - m, y.max() / h + m, x.max() / w + m], F32)
        boxes[b, 0] = np.minimum(np.maximum(box, F32(0)), F32(1))
        cls[b, 0] = 1
    return boxes, cls


def box_iou(a, g) -> np.float32:
    """IoU of two corner boxes in float32, operations in the order the kernel uses."""
    iy0, ix0 = max(a[0], g[0]), max(a[1], g[1])
    iy1, ix1 = min(a[2], g[2]), min(a[3], g[3])
    ih = max(F32(iy1 - iy0), F32(0))
    iw = max(F32(ix1 - ix0), F32(0))
    inter = F32(ih * iw)
    area_a = F32(F32(a[2] - a[0]) * F32(a[3] - a[1]))
    area_g = F32(F32(g[2] - g[0]) * F32(g[3] - g[1]))
    union = F32(F32(area_a + area_g) - inter)
    return F32(inter / union) if union > 0 else F32(0)


def anchor_assign(anchors, boxes, box_classes, pos_iou: float = 2.0, ignore_iou: float = 0.5):
    """Per-anchor detection targets.

    anchors [A,4]; boxes [B,G,4] corner boxes; box_classes [B,G] int (0 = padding slot).
    Returns classes [B,A] f32, weights [B,A] f32, regs [B,A,4] f32, match [B,A] int32.

    Every labelled box claims the anchor it overlaps most (lowest anchor index on ties; a box
    that overlaps no anchor claims none; the lowest box index wins a contested anchor).  With
    pos_iou <= 1, anchors whose best overlap reaches pos_iou also take their best box (lowest
    box index on ties).  A matched anchor gets that box's class, weight 1 and the box itself as
    regression target: the reference's labels show exactly one positive anchor per person and
    `regs` rows that are corner boxes in [0, 1] (test.ipynb:504-572).  Unmatched anchors are
    background (class 0, regs 0) with weight 1, or weight 0 when their best overlap reaches
    ignore_iou.
    """
    anchors = np.asarray(anchors, F32)
    boxes = np.asarray(boxes, F32)
    box_classes = np.asarray(box_classes)
    B, G = box_classes.shape
    A = anchors.shape[0]
    classes = np.zeros((B, A), F32)
    weights = np.ones((B, A), F32)
    regs = np.zeros((B, A, 4), F32)
    match = np.full((B, A), -1, np.int32)
    for b in range(B):
        best_iou = np.zeros(A, F32)
        best_g = np.full(A, -1, np.int32)
        claim = np.full(G, -1, np.int32)
        for g in range(G):
            if box_classes[b, g] <= 0:
                continue
            top = F32(0)
            for a in range(A):
                v = box_iou(anchors[a], boxes[b, g])
                if v > best_iou[a]:
                    best_iou[a], best_g[a] = v, g
                if v > top:
                    top, claim[g] = v, a
        for a in range(A):
            if pos_iou <= 1.0 and best_g[a] >= 0 and best_iou[a] >= F32(pos_iou):
                match[b, a] = best_g[a]
            elif best_iou[a] >= F32(ignore_iou):
                weights[b, a] = 0.0
        for g in range(G - 1, -1, -1):
            if claim[g] >= 0:
                match[b, claim[g]] = g
        for a in range(A):
            g = match[b, a]
            if g >= 0:
                classes[b, a] = F32(box_classes[b, g])
                weights[b, a] = 1.0
                regs[b, a] = boxes[b, g]
    return classes, weights, regs, match


def box_loss_fwd_bwd(logit, reg_pred, classes, weights, regs, reg_weight: float = 1.0):
    """Detection loss and gradients in float64 (the kernel sums in float32: compare with a
    tolerance).  Objectness: weighted sigmoid cross-entropy against (classes > 0); boxes: Huber
    (delta 1) on the four coordinates of positive anchors; both divided by max(1, positives).
    Returns (total, cls, reg, grad_logit, grad_reg)."""
    x = np.asarray(logit, np.float64)
    t = (np.asarray(classes) > 0).astype(np.float64)
    w = np.asarray(weights, np.float64)
    pos = t * (w > 0)
    n = max(1.0, float(pos.sum()))
    bce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    cls = float((w * bce).sum() / n)
    g_logit = w * (1.0 / (1.0 + np.exp(-x)) - t) / n
    d = np.asarray(reg_pred, np.float64) - np.asarray(regs, np.float64)
    hub = np.where(np.abs(d) < 1.0, 0.5 * d * d, np.abs(d) - 0.5)
    reg = float((pos[..., None] * hub).sum() / n)
    g_reg = pos[..., None] * np.clip(d, -1.0, 1.0) * (reg_weight / n)
    return cls + reg_weight * reg, cls, reg, g_logit.astype(F32), g_reg.astype(F32)
