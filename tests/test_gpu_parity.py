"""GPU parity: every HIP entry point, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Integer results and element-wise float32 results must match bit for bit;
reductions carry a stated tolerance."""
import numpy as np
import pytest
import torch

import oracle
import pose_estimation_amd as pe
from pose_estimation_amd import ops, synthetic

pytestmark = pytest.mark.gpu
F32 = np.float32


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_decode_equal(got, want):
    xy, conf, idx = (t.cpu().numpy() for t in got)
    oxy, oconf, oidx = want
    assert np.array_equal(idx, oidx), "arg-max indices differ"
    assert np.array_equal(conf.view(np.uint32), oconf.view(np.uint32))
    assert np.array_equal(xy, oxy)


# ------------------------------------------------------------------ decode
@pytest.mark.parametrize("shape", [(4, 17, 64, 48), (2, 17, 96, 72), (3, 3, 40, 40), (5, 2, 8, 8),
                                   (1, 1, 4, 4), (2, 5, 128, 96), (3, 4, 7, 5), (2, 2, 150, 100)])
def test_decode_contiguous_matches_oracle(shape):
    hm = synthetic.decode_heatmaps(*shape, seed=sum(shape))
    assert_decode_equal(pe.decode(dev(hm)), oracle.decode_heatmaps(hm))
    assert_decode_equal(pe.decode(dev(hm), refine=False), oracle.decode_heatmaps(hm, refine=False))


def adversarial_maps(H, W):
    rng = np.random.default_rng(11)
    maps = []
    z = np.zeros((H, W), F32)
    maps.append(z.copy())                                   # all zero
    maps.append(z - 1.0)                                    # all negative, constant
    m = z - rng.uniform(0.1, 1.0, (H, W)).astype(F32)       # all negative, distinct
    maps.append(m)
    maps.append(z + 0.5)                                    # constant positive
    for (y, x) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (1, 1), (H - 2, W - 2),
                   (2, 2), (H // 2, 1), (1, W // 2), (H // 2, W - 2), (H - 2, W // 2)]:
        m = rng.uniform(0, 0.5, (H, W)).astype(F32)
        m[y, x] = 1.0
        maps.append(m)
    for (a, b) in [((3, 5), (3, 6)), ((H - 1, W - 1), (0, 0)), ((10, 3), (9, W - 1)),
                   ((5, 5), (5, 5 + 64 * 4 % W))]:
        m = z.copy()                                        # exact ties: lowest index wins
        m[a] = 0.75
        m[b] = 0.75
        maps.append(m)
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):                               # every refinement sign case
            m = z.copy()
            m[6, 7] = 1.0
            m[6, 8] = 0.5 + 0.1 * sx
            m[6, 6] = 0.5
            m[7, 7] = 0.5 + 0.1 * sy
            m[5, 7] = 0.5
            maps.append(m)
    m = z.copy()
    m[4, 4] = np.float32(1e-38)                             # denormal-scale neighbour difference
    m[4, 5] = np.float32(1e-45)
    maps.append(m)
    m = z.copy()
    m[:] = -0.0
    maps.append(m)                                          # negative zeros
    m = rng.uniform(-1, 1, (H, W)).astype(F32)
    m[H - 1, W - 1] = m.max()                               # duplicate max at the very end
    maps.append(m)
    m = np.full((H, W), -np.inf, F32)                       # nothing above -inf
    maps.append(m)
    return np.stack(maps)


@pytest.mark.parametrize("hw", [(64, 48), (16, 12), (96, 72)])
def test_decode_adversarial_maps(hw):
    H, W = hw
    maps = adversarial_maps(H, W)
    hm = maps[None]  # [1, K, H, W]
    assert_decode_equal(pe.decode(dev(hm)), oracle.decode_heatmaps(hm))
    hm2 = np.stack([maps, maps[::-1]])  # [2, K, H, W]
    assert_decode_equal(pe.decode(dev(hm2)), oracle.decode_heatmaps(hm2))


@pytest.mark.parametrize("shape", [(4, 17, 64, 48), (2, 3, 40, 40), (3, 5, 9, 7), (2, 64, 16, 16)])
def test_decode_nhwc_and_strided_views(shape):
    B, K, H, W = shape
    hm = synthetic.decode_heatmaps(B, K, H, W, seed=5)
    want = oracle.decode_heatmaps(hm)
    nhwc = dev(hm.transpose(0, 2, 3, 1))
    assert nhwc.is_contiguous()
    assert_decode_equal(pe.decode(nhwc, layout="NHWC"), want)
    # NCHW view of channels-last memory
    assert_decode_equal(pe.decode(nhwc.permute(0, 3, 1, 2)), want)
    # padded rows and a channel slice: generic strides
    padded = torch.zeros(B, K + 2, H, W + 3, device="cuda")
    padded[:, 1:K + 1, :, :W] = dev(hm)
    assert_decode_equal(pe.decode(padded[:, 1:K + 1, :, :W]), want)


def test_decode_empty_batch():
    xy, conf, idx = pe.decode(torch.zeros(0, 17, 64, 48, device="cuda"))
    assert xy.shape == (0, 17, 2) and conf.shape == (0, 17) and idx.shape == (0, 17)


def test_decode_full_size_properties():
    """BASELINE config 2 shape: properties that need no CPU pass over 214 MB."""
    B, K, H, W = 1024, 17, 64, 48
    g = torch.Generator(device="cuda").manual_seed(0)
    hm = torch.rand(B, K, H, W, device="cuda", generator=g) * 0.5
    pos = torch.randint(0, H * W, (B, K), device="cuda", generator=g)
    hm.view(B, K, -1).scatter_(2, pos[..., None], 1.0)  # plant a unique maximum
    xy, conf, idx = pe.decode(hm)
    assert torch.equal(idx.long(), pos)
    assert torch.all(conf == 1.0)
    assert torch.all((xy[..., 0] - (pos % W).float()).abs() <= 0.25)
    assert torch.all((xy[..., 1] - (pos // W).float()).abs() <= 0.25)
    # idempotence on a second call and agreement with the NHWC path on the same data
    xy2, conf2, idx2 = pe.decode(hm)
    assert torch.equal(idx, idx2) and torch.equal(xy, xy2) and torch.equal(conf, conf2)
    sub = hm[:64]
    a = pe.decode(sub)
    b = pe.decode(sub.permute(0, 2, 3, 1).contiguous(), layout="NHWC")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # a slice against the oracle
    assert_decode_equal(pe.decode(hm[:8]), oracle.decode_heatmaps(hm[:8].cpu().numpy()))


# ------------------------------------------------------------------ render
@pytest.mark.parametrize("cfg", [(4, 17, 64, 48, 4.0, 2.0, 6), (2, 17, 96, 72, 4.0, 3.0, 9),
                                 (3, 3, 40, 40, 8.0, 1.0, 1), (2, 5, 16, 12, 4.0, 1.0, 3),
                                 (2, 4, 10, 7, 4.0, 1.5, 4), (1, 2, 64, 48, 4.0, 10.0, 31)])
def test_render_matches_oracle_bit_for_bit(cfg):
    B, K, H, W, stride, sigma, r = cfg
    j = synthetic.joints(B, K, int(H * stride), int(W * stride), seed=2)
    # push some joints to and over the borders
    j[0, 0, :2] = (0.0, 0.0)
    j[0, 1, :2] = (W * stride - 0.1, H * stride - 0.1)
    j[-1, 0, :2] = (-stride * (r - 0.4), 10.0)
    j[-1, 1, :2] = (-stride * (r + 1.0), 10.0)
    j[:, :2, 2] = 1
    want_t, want_w = oracle.render_targets(j, H, W, stride, sigma, r)
    t, w = pe.render_targets(dev(j), H, W, stride, sigma, r)
    assert np.array_equal(w.cpu().numpy(), want_w)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), want_t.view(np.uint32))
    t2, w2 = pe.render_targets(dev(j), H, W, stride, sigma, r, layout="NHWC")
    assert np.array_equal(t2.permute(0, 3, 1, 2).cpu().numpy(), want_t)
    assert np.array_equal(w2.cpu().numpy(), want_w)


def test_render_edge_cases_match_oracle():
    H, W, stride, sigma, r = 16, 12, 4.0, 1.0, 3
    j = np.array([[[20.0, 24.0, 2], [21.9, 25.9, 1], [22.0, 26.0, 1], [0.0, 0.0, 1],
                   [47.9, 63.9, 1], [-12.1, 10.0, 1], [-16.1, 10.0, 1], [60.0, 10.0, 1],
                   [20.0, 24.0, 0], [float("nan"), 5.0, 1], [20.0, float("inf"), 1],
                   [20.0, 24.0, -1], [1e30, 1e30, 1], [-1e30, 3.0, 1]]], F32)
    want_t, want_w = oracle.render_targets(j, H, W, stride, sigma, r)
    t, w = pe.render_targets(dev(j), H, W, stride, sigma, r)
    assert np.array_equal(w.cpu().numpy(), want_w)
    assert np.array_equal(t.cpu().numpy(), want_t)


def test_render_then_decode_round_trip_full_size():
    """Config-4 sized batch: decode(render(j)) returns the rounded centres of visible joints."""
    B, K, H, W, stride = 64, 17, 64, 48, 4.0
    j = synthetic.joints(B, K, 256, 192, seed=2)
    t, w = pe.render_targets(dev(j), H, W, stride, 2.0, 6)
    xy, conf, idx = pe.decode(t, refine=False)
    mx, my = oracle.numpy_oracle.joint_centres(j, stride)
    on = w.cpu().numpy() > 0
    inside = on & (mx >= 0) & (mx < W) & (my >= 0) & (my < H)
    got = xy.cpu().numpy()
    assert np.array_equal(got[inside][:, 0], mx[inside])
    assert np.array_equal(got[inside][:, 1], my[inside])
    assert np.all(conf.cpu().numpy()[inside] == 1.0)
    assert np.all(conf.cpu().numpy()[~on] == 0.0)
    assert float(t.max()) == 1.0 and float(t.min()) == 0.0


# ------------------------------------------------------------------ vector maps
COCO_PAIRS = [(5, 7), (7, 9), (6, 8), (8, 10), (11, 13), (13, 15), (12, 14), (14, 16), (5, 6),
              (11, 12), (5, 11), (6, 12), (0, 5), (0, 6), (1, 2), (3, 4)]


def _vec_joints(B, K, H, W, stride, seed=2):
    j = synthetic.joints(B, K, int(H * stride), int(W * stride), seed=seed)
    j[0, 5] = (W * stride + 1.0, 20.0, 2)         # centre one cell past the right edge
    j[0, 7] = (-3 * stride, -3 * stride, 1)       # window wholly off the map
    j[0, 6] = (float("nan"), 3.0, 1)
    j[-1, 11] = (1e30, 5.0, 1)                    # beyond the representable centre range
    return j


@pytest.mark.parametrize("cfg", [(4, 17, 64, 48, 4.0, 2), (2, 17, 96, 72, 4.0, 3),
                                 (3, 17, 40, 40, 8.0, 1), (2, 17, 10, 7, 4.0, 4)])
def test_vecmap_render_matches_oracle_bit_for_bit(cfg):
    B, K, H, W, stride, r = cfg
    j = _vec_joints(B, K, H, W, stride)
    want, _ = oracle.render_vecmaps(j, COCO_PAIRS, H, W, stride, r)
    got = ops.render_vecmaps(dev(j), COCO_PAIRS, H, W, stride, r)
    assert np.array_equal(got.cpu().numpy(), want)
    nhwc = ops.render_vecmaps(dev(j), COCO_PAIRS, H, W, stride, r, layout="NHWC")
    assert np.array_equal(nhwc.permute(0, 3, 1, 2).cpu().numpy(), want)


@pytest.mark.parametrize("use_mask", [False, True])
@pytest.mark.parametrize("bf16", [False, True])
def test_vecmap_loss_matches_oracle(use_mask, bf16):
    B, K, H, W, stride, r = 3, 17, 64, 48, 4.0, 2
    j = _vec_joints(B, K, H, W, stride)
    rng = np.random.default_rng(5)
    t, _ = oracle.render_vecmaps(j, COCO_PAIRS, H, W, stride, r)
    p = (t + rng.normal(0, 1.0, t.shape)).astype(F32)
    m = synthetic.ignore_mask(B, H, W, seed=4) if use_mask else None
    if bf16:
        pb = oracle.f32_to_bf16_bits(p)
        want_l, want_pc, want_g = oracle.vecmap_loss_fwd_bwd(pb, j, COCO_PAIRS, stride, r, m,
                                                             pred_is_bf16=True)
        pt = torch.from_numpy(pb.view(np.int16)).cuda().view(torch.bfloat16)
    else:
        want_l, want_pc, want_g = oracle.vecmap_loss_fwd_bwd(p, j, COCO_PAIRS, stride, r, m)
        pt = dev(p)
    loss, pc, g = ops.vecmap_loss_fwd_bwd(pt, dev(j), COCO_PAIRS, stride, r,
                                          mask=dev(m) if use_mask else None)
    assert loss.item() == pytest.approx(float(want_l), rel=2e-6)   # float32 tree vs float64 sum
    assert np.allclose(pc.cpu().numpy(), want_pc, rtol=2e-6, atol=1e-12)
    if bf16:
        assert np.array_equal(g.view(torch.int16).cpu().numpy().view(np.uint16), want_g)
    else:
        assert np.array_equal(g.cpu().numpy(), want_g)
    l2, _, none = ops.vecmap_loss_fwd_bwd(pt, dev(j), COCO_PAIRS, stride, r,
                                          mask=dev(m) if use_mask else None, need_grad=False)
    assert none is None and torch.equal(l2, loss)      # fixed reduction order: bit-stable


def test_vecmap_full_size_properties():
    """Batch 64 x 17 joints, 16 pairs: cell + offset lands on the partner's centre everywhere,
    and a prediction equal to the label has zero loss and zero gradient."""
    B, K, H, W, stride, r = 64, 17, 64, 48, 4.0, 2
    j = synthetic.joints(B, K, 256, 192, seed=2)
    vec = ops.render_vecmaps(dev(j), COCO_PAIRS, H, W, stride, r)
    mx, my = oracle.numpy_oracle.joint_centres(j, stride)
    v = vec.cpu().numpy()
    xs = np.arange(W)[None, :]
    ys = np.arange(H)[:, None]
    for i, (k1, k2) in enumerate(COCO_PAIRS[:4]):
        for b in range(0, B, 7):
            if not (j[b, k1, 2] > 0 and j[b, k2, 2] > 0):
                assert not v[b, 4 * i:4 * i + 4].any()
                continue
            win = (np.abs(xs - mx[b, k1]) <= r) & (np.abs(ys - my[b, k1]) <= r)
            assert np.all((v[b, 4 * i] + xs)[win] == mx[b, k2])
            assert np.all((v[b, 4 * i + 1] + ys)[win] == my[b, k2])
            assert not v[b, 4 * i][~win].any()
    loss, pc, g = ops.vecmap_loss_fwd_bwd(vec, dev(j), COCO_PAIRS, stride, r)
    assert loss.item() == 0.0 and not g.any()


def test_vecmap_argument_errors():
    j = dev(synthetic.joints(1, 17, 256, 192, seed=2))
    with pytest.raises(pe._ffi.PeError):
        ops.render_vecmaps(j, [(0, 17)], 64, 48, 4.0, 1)          # joint index out of range
    with pytest.raises(pe._ffi.PeError):
        ops.render_vecmaps(j, [(0, 1)] * 33, 64, 48, 4.0, 1)      # more than 32 pairs
    with pytest.raises(RuntimeError):
        ops.vecmap_loss_fwd_bwd(torch.zeros(1, 4, 64, 48), j, [(0, 1)], 4.0, 1)   # CPU tensor


# ------------------------------------------------------------------ loss
def _loss_inputs(B, K, H, W, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.normal(0, 0.3, size=(B, K, H, W)).astype(F32)
    j = synthetic.joints(B, K, H * 4, W * 4, seed=seed + 2)
    m = synthetic.ignore_mask(B, H, W, seed=seed + 4)
    return p, j, m


@pytest.mark.parametrize("shape", [(2, 17, 64, 48), (3, 5, 16, 16), (64, 17, 64, 48)])
@pytest.mark.parametrize("use_mask", [False, True])
def test_mse_unfused_f32(shape, use_mask):
    B, K, H, W = shape
    p, j, m = _loss_inputs(B, K, H, W)
    t, w = oracle.render_targets(j, H, W, 4.0, 2.0, 6)
    w = (w * np.random.default_rng(9).choice([1.0, 0.5], size=w.shape)).astype(F32)
    mask = m if use_mask else None
    oloss, opj, og = oracle.mse_fwd_bwd(p, t, w, mask)
    loss, pj, g, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w), dev(m) if use_mask else None)
    assert np.array_equal(g.cpu().numpy().view(np.uint32), og.view(np.uint32))
    # float32 tree sum vs float64 sum of the same float32 terms
    assert loss.item() == pytest.approx(float(oloss), rel=2e-6)
    assert np.allclose(pj.cpu().numpy(), opj, rtol=2e-6, atol=0)
    # run-to-run bit stability
    loss2, pj2, g2, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w), dev(m) if use_mask else None)
    assert loss2.item() == loss.item() and torch.equal(pj, pj2) and torch.equal(g, g2)


@pytest.mark.parametrize("use_mask", [False, True])
def test_mse_fused_f32_equals_unfused(use_mask):
    B, K, H, W = 8, 17, 64, 48
    p, j, m = _loss_inputs(B, K, H, W, seed=3)
    t, w = oracle.render_targets(j, H, W, 4.0, 2.0, 6)
    mask = m if use_mask else None
    oloss, opj, og = oracle.mse_fwd_bwd(p, t, w, mask)
    loss, pj, g, wk = pe.mse_fwd_bwd(dev(p), joints=dev(j), mask=dev(m) if use_mask else None,
                                     stride=4.0, sigma=2.0, radius=6)
    assert np.array_equal(wk.cpu().numpy(), w)
    assert np.array_equal(g.cpu().numpy().view(np.uint32), og.view(np.uint32))
    assert loss.item() == pytest.approx(float(oloss), rel=2e-6)
    loss_u, pj_u, g_u, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w), dev(m) if use_mask else None)
    assert loss_u.item() == loss.item() and torch.equal(pj, pj_u) and torch.equal(g, g_u)


@pytest.mark.parametrize("fused", [False, True])
def test_mse_bf16(fused):
    B, K, H, W = 4, 17, 64, 48
    p, j, m = _loss_inputs(B, K, H, W, seed=5)
    pb = oracle.f32_to_bf16_bits(p)
    t, w = oracle.render_targets(j, H, W, 4.0, 2.0, 6)
    oloss, opj, og = oracle.mse_fwd_bwd(pb, t, w, m, pred_is_bf16=True)
    pt = torch.from_numpy(pb.view(np.int16)).cuda().view(torch.bfloat16)
    if fused:
        loss, pj, g, _ = pe.mse_fwd_bwd(pt, joints=dev(j), mask=dev(m), stride=4.0, sigma=2.0,
                                        radius=6)
    else:
        loss, pj, g, _ = pe.mse_fwd_bwd(pt, dev(t), dev(w), dev(m))
    assert g.dtype == torch.bfloat16
    gb = g.view(torch.int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(gb, og)
    assert loss.item() == pytest.approx(float(oloss), rel=2e-6)
    assert np.allclose(pj.cpu().numpy(), opj, rtol=2e-6)


def test_mse_loss_only_and_zero_weight():
    B, K, H, W = 2, 4, 16, 16
    p, j, m = _loss_inputs(B, K, H, W, seed=7)
    t, w = oracle.render_targets(j, H, W, 4.0, 1.0, 3)
    w[:] = 0
    loss, pj, g, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w), need_grad=False)
    assert g is None and loss.item() == 0.0 and not pj.any()
    loss, pj, g, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w))
    assert not g.any()
    # linearity property: scaling (p - t) by 2 scales the loss by 4 and the gradient by 2
    w[:] = 1
    l1, _, g1, _ = pe.mse_fwd_bwd(dev(p), dev(t), dev(w))
    l2, _, g2, _ = pe.mse_fwd_bwd(dev(t + 2 * (p - t)), dev(t), dev(w))
    assert l2.item() == pytest.approx(4 * l1.item(), rel=1e-5)
    assert torch.allclose(g2, 2 * g1, rtol=1e-5, atol=1e-9)


# ------------------------------------------------------------------ optimizers
@pytest.mark.parametrize("n", [1000, 4096 + 3, 1, 34_000_000 // 64])
@pytest.mark.parametrize("tf1", [True, False])
def test_adam_bit_exact_three_steps(n, tf1):
    rng = np.random.default_rng(n)
    p = rng.normal(size=n).astype(F32)
    m = np.zeros(n, F32)
    v = np.zeros(n, F32)
    tp, tm, tv = dev(p), dev(m), dev(v)
    tb = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        g = (rng.normal(size=n) * 10.0 ** rng.integers(-4, 2)).astype(F32)
        p, m, v, pb = oracle.adam_step(p, m, v, g, 1e-3, 0.9, 0.999, 1e-8, step, tf1=tf1,
                                       grad_scale=0.125)
        pe.adam_step(tp, tm, tv, dev(g), step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                     p_bf16=tb, tf1=tf1, grad_scale=0.125)
        assert np.array_equal(tm.cpu().numpy().view(np.uint32), m.view(np.uint32))
        assert np.array_equal(tv.cpu().numpy().view(np.uint32), v.view(np.uint32))
        assert np.array_equal(tp.cpu().numpy().view(np.uint32), p.view(np.uint32))
        assert np.array_equal(tb.view(torch.int16).cpu().numpy().view(np.uint16), pb)


def test_sgd_bit_exact():
    rng = np.random.default_rng(5)
    n = 10_007
    p = rng.normal(size=n).astype(F32)
    mom = np.zeros(n, F32)
    tp, tm = dev(p), dev(mom)
    tb = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        g = rng.normal(size=n).astype(F32)
        p, mom, pb = oracle.sgd_step(p, mom, g, 0.05, 0.9)
        pe.sgd_step(tp, tm, dev(g), 0.05, 0.9, p_bf16=tb)
        assert np.array_equal(tp.cpu().numpy().view(np.uint32), p.view(np.uint32))
        assert np.array_equal(tm.cpu().numpy().view(np.uint32), mom.view(np.uint32))
        assert np.array_equal(tb.view(torch.int16).cpu().numpy().view(np.uint16), pb)
