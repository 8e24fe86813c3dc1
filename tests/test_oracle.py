"""Known-answer and cross-check tests of the CPU oracle itself (no GPU).

The reference holds no implementation to pin these against (see oracle/__init__.py); each
oracle function is instead checked against an independent statement of the same formula:
plain Python loops, float64 closed forms, finite differences, torch-CPU optimizers.
"""
import math

import numpy as np
import pytest
import torch

import oracle
from pose_estimation_amd import synthetic

F32 = np.float32


# ------------------------------------------------------------------ decode
def decode_loops(hm, refine=True):
    B, K, H, W = hm.shape
    xy = np.zeros((B, K, 2), F32)
    conf = np.zeros((B, K), F32)
    idx = np.zeros((B, K), np.int32)
    for b in range(B):
        for k in range(K):
            best, bi = None, 0
            for i, v in enumerate(hm[b, k].reshape(-1).tolist()):
                if best is None or v > best:
                    best, bi = v, i
            y, x = divmod(bi, W)
            fx, fy = float(x), float(y)
            if refine and 1 < x < W - 1 and 1 < y < H - 1:
                dx = float(hm[b, k, y, x + 1]) - float(hm[b, k, y, x - 1])
                dy = float(hm[b, k, y + 1, x]) - float(hm[b, k, y - 1, x])
                fx += 0.25 * ((dx > 0) - (dx < 0))
                fy += 0.25 * ((dy > 0) - (dy < 0))
            if not best > 0:
                fx = fy = 0.0
            xy[b, k] = (fx, fy)
            conf[b, k] = best
            idx[b, k] = bi
    return xy, conf, idx


def test_decode_matches_loops_on_random_blobs():
    hm = synthetic.decode_heatmaps(3, 5, 16, 12, seed=7)
    got = oracle.decode_heatmaps(hm)
    want = decode_loops(hm)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_decode_known_answers():
    H, W = 8, 6
    hm = np.zeros((1, 8, H, W), F32)
    # 0: all zeros -> idx 0, conf 0, coords zeroed
    # 1: all negative -> argmax is the least negative, coords zeroed
    hm[0, 1] = -1.0
    hm[0, 1, 3, 2] = -0.5
    # 2: tie -> lowest flat index
    hm[0, 2, 5, 4] = 1.0
    hm[0, 2, 2, 3] = 1.0
    # 3: interior peak with larger right and upper neighbours -> (+0.25, -0.25)
    hm[0, 3, 4, 3] = 1.0
    hm[0, 3, 4, 4] = 0.5
    hm[0, 3, 3, 3] = 0.7
    # 4: peak at x == 1 (not refined although neighbours exist)
    hm[0, 4, 4, 1] = 1.0
    hm[0, 4, 4, 2] = 0.5
    # 5: peak on the last row / last column corner
    hm[0, 5, H - 1, W - 1] = 2.0
    # 6: constant positive map -> idx 0
    hm[0, 6] = 0.25
    # 7: equal neighbours -> no move
    hm[0, 7, 3, 3] = 1.0
    hm[0, 7, 3, 2] = hm[0, 7, 3, 4] = 0.4
    xy, conf, idx = oracle.decode_heatmaps(hm)
    assert idx[0].tolist() == [0, 3 * W + 2, 2 * W + 3, 4 * W + 3, 4 * W + 1,
                               (H - 1) * W + W - 1, 0, 3 * W + 3]
    assert conf[0].tolist() == [0.0, -0.5, 1.0, 1.0, 1.0, 2.0, 0.25, 1.0]
    assert xy[0].tolist() == [[0, 0], [0, 0], [3, 2], [3.25, 3.75], [1, 4], [W - 1, H - 1],
                              [0, 0], [3, 3]]
    xy_nr, _, _ = oracle.decode_heatmaps(hm, refine=False)
    assert xy_nr[0, 3].tolist() == [3, 4]


def test_decode_nan_rule_equals_plain_decode_without_nan_and_skips_nan():
    hm = synthetic.decode_heatmaps(2, 4, 12, 10, seed=9)
    for a, b in zip(oracle.decode_heatmaps_nan_skipped(hm), oracle.decode_heatmaps(hm)):
        assert np.array_equal(a, b)
    m = np.zeros((1, 3, 8, 8), np.float32)
    m[0, 0, 3, 4] = 0.5
    m[0, 0, 1, 1] = np.nan                 # a NaN elsewhere never wins
    m[0, 1, 3, 4] = 0.5
    m[0, 1, 3, 5] = np.nan                 # NaN neighbour: x stays, y moves towards 0.2
    m[0, 1, 4, 4] = 0.2
    m[0, 2] = np.nan                       # nothing but NaN: idx 0, conf NaN, xy 0
    xy, conf, idx = oracle.decode_heatmaps_nan_skipped(m)
    assert idx.tolist() == [[28, 28, 0]]
    assert xy[0, 0].tolist() == [4.0, 3.0] and xy[0, 1].tolist() == [4.0, 3.25]
    assert conf[0, 0] == 0.5 and np.isnan(conf[0, 2]) and xy[0, 2].tolist() == [0.0, 0.0]


def test_decode_accepts_nhwc_views():
    hm = synthetic.decode_heatmaps(2, 3, 10, 8, seed=3)
    nhwc = np.ascontiguousarray(hm.transpose(0, 2, 3, 1))
    a = oracle.decode_heatmaps(hm)
    b = oracle.decode_heatmaps(nhwc.transpose(0, 3, 1, 2))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ------------------------------------------------------------------ render
def test_gaussian_table_against_closed_form():
    for sigma, r in [(2.0, 6), (3.0, 9), (1.0, 3)]:
        g = oracle.gaussian_table(sigma, r)
        ref = [math.exp(-(i * i) / (2 * sigma * sigma)) for i in range(r + 1)]
        assert g.dtype == F32 and g[0] == 1.0
        assert np.allclose(g, ref, rtol=1e-7, atol=0)


def test_render_against_float64_closed_form():
    H, W, stride, sigma, r = 64, 48, 4.0, 2.0, 6
    j = synthetic.joints(4, 17, 256, 192, seed=2)
    t, w = oracle.render_targets(j, H, W, stride, sigma, r)
    assert t.shape == (4, 17, H, W) and w.shape == (4, 17)
    ys, xs = np.mgrid[0:H, 0:W]
    for b in range(4):
        for k in range(17):
            x, y, v = j[b, k]
            cx = math.floor(x / stride + 0.5)
            cy = math.floor(y / stride + 0.5)
            if v == 0:
                assert w[b, k] == 0 and not t[b, k].any()
                continue
            assert w[b, k] == 1
            ref = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma * sigma))
            ref[(abs(xs - cx) > r) | (abs(ys - cy) > r)] = 0
            # one float32 product of two rounded factors: < 2 ulp of the closed form
            assert np.allclose(t[b, k], ref, rtol=3e-7, atol=0)
            assert t[b, k].max() <= 1.0


def test_render_edge_cases():
    H, W, stride, sigma, r = 16, 12, 4.0, 1.0, 3
    j = np.array([[
        [20.0, 24.0, 2],            # 0 centre exactly on a cell: (5, 6)
        [21.9, 25.9, 1],            # 1 sub-pixel: 5.475+.5 -> 5, 6.475+.5 -> 6
        [22.0, 26.0, 1],            # 2 exactly .5 -> rounds up: (6, 7)
        [0.0, 0.0, 1],              # 3 window clipped at the top-left corner
        [45.9, 61.9, 1],            # 4 centre (11, 15): the last cell, clipped bottom-right
        [47.9, 63.9, 1],            # 5 centre (12, 16): one cell off the map -> weight 0, as the
                                    #   reference leaves such a joint's heat map empty
        [-2.1, 10.0, 1],            # 6 centre -1: off the map although its window is not
        [60.0, 10.0, 1],            # 7 centre 15 -> weight 0
        [20.0, 24.0, 0],            # 8 invisible
        [float("nan"), 5.0, 1],     # 9 not finite
        [20.0, float("inf"), 1],    # 10 not finite
        [20.0, 24.0, -1],           # 11 negative visibility counts as not visible
    ]], F32)
    t, w = oracle.render_targets(j, H, W, stride, sigma, r)
    assert w[0].tolist() == [1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    g = oracle.gaussian_table(sigma, r)
    assert t[0, 0, 6, 5] == 1.0 and t[0, 0, 6, 8] == g[3] and t[0, 0, 6, 9] == 0
    assert t[0, 0, 9, 8] == F32(g[3] * g[3])
    assert np.argmax(t[0, 1]) == 6 * W + 5
    assert np.argmax(t[0, 2]) == 7 * W + 6
    assert t[0, 3, 0, 0] == 1.0 and np.count_nonzero(t[0, 3]) == 16
    assert t[0, 4, H - 1, W - 1] == 1.0 and np.count_nonzero(t[0, 4]) == 16
    for k in range(5, 12):
        assert not t[0, k].any()


# ------------------------------------------------------------------ vector maps
def test_vecmaps_against_python_loops():
    H, W, stride, r = 16, 12, 4.0, 2
    pairs = [(0, 1), (2, 1), (3, 4)]
    j = np.array([[[20.0, 24.0, 2], [9.0, 40.0, 1], [47.9, 63.9, 1], [20.0, 24.0, 0], [5.0, 5.0, 1]],
                  [[-30.0, 8.0, 1], [2.0, 2.0, 1], [float("nan"), 1.0, 1], [1.0, 1.0, 1],
                   [46.0, 60.0, 2]]], F32)
    vec, sup = oracle.render_vecmaps(j, pairs, H, W, stride, r)
    assert vec.shape == (2, 12, H, W) and sup.shape == vec.shape
    for b in range(2):
        for i, (k1, k2) in enumerate(pairs):
            ok = all(j[b, k, 2] > 0 and np.isfinite(j[b, k, :2]).all() for k in (k1, k2))
            for half, (ks, kd) in enumerate(((k1, k2), (k2, k1))):
                for axis in range(2):
                    c = 4 * i + 2 * half + axis
                    for y in range(H):
                        for x in range(W):
                            want, inside = 0.0, False
                            if ok:
                                sx = math.floor(j[b, ks, 0] / stride + 0.5)
                                sy = math.floor(j[b, ks, 1] / stride + 0.5)
                                dx = math.floor(j[b, kd, 0] / stride + 0.5)
                                dy = math.floor(j[b, kd, 1] / stride + 0.5)
                                inside = abs(x - sx) <= r and abs(y - sy) <= r
                                if inside:
                                    want = (dy - y) if axis else (dx - x)
                            assert vec[b, c, y, x] == want and sup[b, c, y, x] == inside
    assert not vec[0, 8:12].any()       # joint 3 is unlabelled: the whole pair is empty
    assert not vec[1, 4:8].any()        # joint 2 is not finite
    assert vec[1, 2:4].any() and not vec[1, 0:2].any()   # joint 0's window is wholly off the map


def test_vecmap_loss_against_float64_and_finite_differences():
    rng = np.random.default_rng(3)
    B, H, W, stride, r = 2, 8, 8, 4.0, 1
    pairs = [(0, 1), (1, 2)]
    j = np.array([[[10.0, 12.0, 1], [20.0, 6.0, 1], [30.5, 30.5, 2]],
                  [[4.0, 4.0, 1], [31.9, 0.0, 1], [8.0, 8.0, 0]]], F32)
    p = rng.normal(size=(B, 8, H, W)).astype(F32)
    m = (rng.uniform(size=(B, H, W)) > 0.2).astype(np.uint8)
    t, sup = oracle.render_vecmaps(j, pairs, H, W, stride, r)
    loss, pc, g = oracle.vecmap_loss_fwd_bwd(p, j, pairs, stride, r, m)

    def loss64(pp):
        keep = sup & (m[:, None] != 0)
        return 0.5 * (((pp.astype(np.float64) - t) ** 2) * keep).sum() / pp.size

    assert float(loss) == pytest.approx(loss64(p), rel=1e-6)
    assert float(pc.mean()) == pytest.approx(float(loss), rel=1e-6)
    assert not g[~sup].any() and g[sup & (m[:, None] != 0)].any()
    for idx in [(0, 0, 3, 2), (0, 5, 1, 5), (1, 2, 0, 7), (1, 0, 1, 1), (0, 7, 7, 7)]:
        e = np.zeros_like(p, dtype=np.float64)
        e[idx] = 1e-3
        fd = (loss64(p + e) - loss64(p - e)) / 2e-3
        assert g[idx] == pytest.approx(fd, rel=1e-3, abs=1e-9)
    # bf16 predictions: gradient rounds to bf16, loss uses the widened values
    pb = oracle.f32_to_bf16_bits(p)
    lb, _, gb = oracle.vecmap_loss_fwd_bwd(pb, j, pairs, stride, r, m, pred_is_bf16=True)
    assert float(lb) == pytest.approx(loss64(oracle.bf16_bits_to_f32(pb)), rel=1e-6)
    assert gb.dtype == np.uint16


# ------------------------------------------------------------------ loss
def _loss64(p, t, w, m):
    B, K, H, W = p.shape
    e = w[:, :, None, None].astype(np.float64) * (p.astype(np.float64) - t.astype(np.float64))
    q = e * e
    if m is not None:
        q = q * (m[:, None] != 0)
    return 0.5 * q.sum() / (B * K * H * W)


def test_mse_loss_and_gradient_against_float64_and_finite_differences():
    rng = np.random.default_rng(0)
    B, K, H, W = 2, 3, 8, 8
    p = rng.normal(size=(B, K, H, W)).astype(F32)
    t = rng.uniform(size=(B, K, H, W)).astype(F32)
    w = np.array([[1, 0, 1], [1, 1, 0.5]], F32)
    m = (rng.uniform(size=(B, H, W)) > 0.3).astype(np.uint8)
    loss, pj, g = oracle.mse_fwd_bwd(p, t, w, m)
    assert abs(loss - _loss64(p, t, w, m)) <= 1e-6 * abs(loss)
    assert abs(pj.mean() - loss) <= 1e-6 * abs(loss)
    # weight-0 joints and masked pixels: zero loss contribution and zero gradient
    assert not g[0, 1].any() and pj[1] == pytest.approx(_loss64(p[:, 1:2], t[:, 1:2], w[:, 1:2], m))
    assert not g[:, :, m[0] == 0][0].any()
    # finite differences in float64
    eps = 1e-3
    for (b, k, y, x) in [(0, 0, 1, 2), (1, 2, 7, 7), (1, 1, 0, 0), (0, 2, 3, 3)]:
        pp = p.astype(np.float64).copy()
        pm = pp.copy()
        pp[b, k, y, x] += eps
        pm[b, k, y, x] -= eps
        fd = (_loss64(pp, t, w, m) - _loss64(pm, t, w, m)) / (2 * eps)
        assert g[b, k, y, x] == pytest.approx(fd, rel=1e-4, abs=1e-9)


def test_mse_bf16_path_rounds_gradient_to_nearest_even():
    rng = np.random.default_rng(1)
    B, K, H, W = 1, 2, 4, 8
    p32 = rng.normal(size=(B, K, H, W)).astype(F32)
    pb = oracle.f32_to_bf16_bits(p32)
    t = rng.uniform(size=(B, K, H, W)).astype(F32)
    w = np.ones((B, K), F32)
    loss_b, _, gb = oracle.mse_fwd_bwd(pb, t, w, None, pred_is_bf16=True)
    loss_f, _, gf = oracle.mse_fwd_bwd(oracle.bf16_bits_to_f32(pb), t, w, None)
    assert gb.dtype == np.uint16 and loss_b == loss_f
    assert np.array_equal(gb, oracle.f32_to_bf16_bits(gf))
    tb = torch.from_numpy(gf).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(gb, tb)  # same rounding as torch's float32 -> bfloat16 cast


def test_bf16_round_trip_helpers():
    x = np.array([1.0, 1.00390625, 1.0078125, -3.14159, 1e-20, 65504.0, 0.0], F32)
    tb = torch.from_numpy(x).to(torch.bfloat16)
    assert np.array_equal(oracle.bf16_round(x), tb.to(torch.float32).numpy())


# ------------------------------------------------------------------ optimizers
def test_adam_tracks_torch_adam():
    rng = np.random.default_rng(3)
    n = 1000
    p0 = rng.normal(size=n).astype(F32)
    grads = [rng.normal(size=n).astype(F32) for _ in range(3)]
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    p, m, v = p0.copy(), np.zeros(n, F32), np.zeros(n, F32)
    for step, g in enumerate(grads, start=1):
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v, pb = oracle.adam_step(p, m, v, g, 1e-3, 0.9, 0.999, 1e-8, step, tf1=False)
        # same formula, different float32 operation order: a few ulp of |p| ~ 1
        assert np.allclose(p, tp.detach().numpy(), rtol=0, atol=1e-6)
        assert np.array_equal(pb, oracle.f32_to_bf16_bits(p))


def test_adam_tf1_form_first_step_is_lr_sized():
    # t = 1: m = (1-b1) g, v = (1-b2) g^2, lr_t = lr*sqrt(1-b2)/(1-b1)  =>  |dp| ~= lr
    g = np.array([0.5, -2.0, 1e-3], F32)
    p, m, v, _ = oracle.adam_step(np.zeros(3, F32), np.zeros(3, F32), np.zeros(3, F32), g,
                                  1e-3, 0.9, 0.999, 1e-8, 1, tf1=True)
    assert np.allclose(p, -1e-3 * np.sign(g), rtol=1e-3)  # eps is 3e-4 of sqrt(v) at g = 1e-3
    assert np.allclose(m, 0.1 * g, rtol=1e-6) and np.allclose(v, 1e-3 * g * g, rtol=1e-4)  # 1 - float32(0.999) is 1.3e-5 off 1e-3


def test_sgd_momentum_tracks_torch_sgd():
    rng = np.random.default_rng(4)
    n = 257
    p0 = rng.normal(size=n).astype(F32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=0.1, momentum=0.9)
    p, mom = p0.copy(), np.zeros(n, F32)
    for _ in range(3):
        g = rng.normal(size=n).astype(F32)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, mom, _ = oracle.sgd_step(p, mom, g, 0.1, 0.9)
        assert np.allclose(p, tp.detach().numpy(), rtol=0, atol=1e-6)
