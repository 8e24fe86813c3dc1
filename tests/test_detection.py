"""Detection branch labels (anchors, per-anchor targets) and loss.

CPU: the NumPy restatement against what the reference's notebook shows of its `bboxes` labels
(1500 anchors in [20,20,3] + [10,10,3] order, one positive anchor per person on the middle
aspect ratio of the 20x20 level, `regs` = the person's corner box; test.ipynb:423-572) and
against hand-built cases.  GPU: pe_anchor_assign_f32 bit for bit, pe_box_loss_fwd_bwd within
float32 summation tolerance.  Anchor scales/ratios and the matching rule beyond those facts are
not visible in the reference: parity unpinned.
"""
import numpy as np
import pytest
import torch

import oracle
from pose_estimation_amd import detection as det

F32 = np.float32
NOTEBOOK_BOX = (0.54014605, 0.2080292, 0.9726278, 0.7080292)   # test.ipynb:571-572


def random_boxes(B, G, seed, empty=0.3):
    rng = np.random.default_rng(seed)
    cy, cx = rng.uniform(0.1, 0.9, (B, G)), rng.uniform(0.1, 0.9, (B, G))
    h, w = rng.uniform(0.05, 0.7, (B, G)), rng.uniform(0.05, 0.7, (B, G))
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(F32)
    cls = (rng.uniform(size=(B, G)) > empty).astype(np.int32) * rng.integers(1, 4, (B, G)).astype(np.int32)
    return boxes, cls


def test_anchor_layout_matches_the_notebook():
    a = oracle.ssd_anchors()
    assert a.shape == (1500, 4) and 20 * 20 * 3 + 10 * 10 * 3 == 1500       # test.ipynb:423-453
    lvl1 = a[:1200].reshape(20, 20, 3, 4)
    lvl2 = a[1200:].reshape(10, 10, 3, 4)
    cy = (lvl1[..., 0] + lvl1[..., 2]) / 2
    cx = (lvl1[..., 1] + lvl1[..., 3]) / 2
    assert np.allclose(cy[:, 0, 0], (np.arange(20) + 0.5) / 20, atol=1e-6)
    assert np.allclose(cx[0, :, 2], (np.arange(20) + 0.5) / 20, atol=1e-6)
    h, w = lvl2[..., 2] - lvl2[..., 0], lvl2[..., 3] - lvl2[..., 1]
    assert np.allclose(h[..., 1], 0.75, atol=1e-6) and np.allclose(w[..., 1], 0.75, atol=1e-6)
    assert np.allclose(w[..., 2] / h[..., 2], 2.0, rtol=1e-5) and np.allclose(h * w, 0.75 ** 2, rtol=1e-5)
    # the host-side generator states the same boxes
    assert np.array_equal(det.make_anchors(), a)


def test_notebook_person_gets_one_positive_anchor_with_its_box_as_regs():
    a = oracle.ssd_anchors()
    boxes = np.zeros((1, 4, 4), F32)
    boxes[0, 1] = NOTEBOOK_BOX
    cls = np.array([[0, 1, 0, 0]], np.int32)
    classes, weights, regs, match = oracle.anchor_assign(a, boxes, cls)
    pos = np.nonzero(classes[0])[0]
    assert len(pos) == 1 and pos[0] < 1200 and pos[0] % 3 == 1      # feat_1, middle ratio
    sel = regs[0][regs[0][:, 2] - regs[0][:, 0] > 0]                  # the notebook's expression
    assert sel.shape == (1, 4) and np.array_equal(sel[0], np.asarray(NOTEBOOK_BOX, F32))
    assert match[0, pos[0]] == 1 and weights[0, pos[0]] == 1.0
    # neighbours that overlap the person at >= 0.5 are ignored, not negatives
    assert (weights[0] == 0).sum() > 0 and np.all(classes[0][weights[0] == 0] == 0)
    # two people -> two positives (test.ipynb:538-541 shows two cells)
    boxes[0, 3] = (0.2, 0.1, 0.5, 0.35)
    cls[0, 3] = 1
    classes, _, _, _ = oracle.anchor_assign(a, boxes, cls)
    assert (classes[0] > 0).sum() == 2


def test_assignment_rules_on_hand_built_cases():
    a = np.array([[0, 0, .5, .5], [0, .5, .5, 1], [.5, 0, 1, .5], [.5, .5, 1, 1]], F32)
    # box 0 equals anchor 2; box 1 overlaps anchors 0 and 1 equally -> lowest anchor; slot 2 empty
    boxes = np.array([[[.5, 0, 1, .5], [0, .25, .5, .75], [0, 0, 1, 1]]], F32)
    cls = np.array([[2, 3, 0]], np.int32)
    classes, weights, regs, match = oracle.anchor_assign(a, boxes, cls, ignore_iou=0.3)
    assert match[0].tolist() == [1, -1, 0, -1] and classes[0].tolist() == [3, 0, 2, 0]
    assert weights[0].tolist() == [1, 0, 1, 1]                      # anchor 1: IoU 1/3 >= 0.3, unmatched
    assert np.array_equal(regs[0, 2], boxes[0, 0]) and not regs[0, 1].any()
    # threshold matching on top of the claims
    _, _, _, m2 = oracle.anchor_assign(a, boxes, cls, pos_iou=0.3)
    assert m2[0].tolist() == [1, 1, 0, -1]
    # two boxes contest one anchor: the lower box index wins
    boxes = np.array([[[0, 0, .5, .5], [0, 0, .5, .5]]], F32)
    _, _, _, m3 = oracle.anchor_assign(a, boxes, np.array([[1, 1]], np.int32))
    assert m3[0].tolist() == [0, -1, -1, -1]
    # no labelled boxes: all background, weight 1
    c4, w4, r4, m4 = oracle.anchor_assign(a, boxes, np.zeros((1, 2), np.int32))
    assert not c4.any() and w4.all() and not r4.any() and (m4 == -1).all()


def test_box_loss_oracle_against_torch_autograd():
    rng = np.random.default_rng(5)
    a = oracle.ssd_anchors()
    boxes, cls = random_boxes(3, 5, 1)
    classes, weights, regs, _ = oracle.anchor_assign(a, boxes, cls)
    logit = rng.normal(size=classes.shape).astype(F32) * 3
    pred = (regs + rng.normal(size=regs.shape).astype(F32) * 0.8).astype(F32)
    total, c, r, gl, gr = oracle.box_loss_fwd_bwd(logit, pred, classes, weights, regs, 2.0)
    x = torch.tensor(logit, dtype=torch.float64, requires_grad=True)
    p = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    t = torch.tensor((classes > 0).astype(np.float64))
    w = torch.tensor(weights.astype(np.float64))
    n = max(1.0, float((t * (w > 0)).sum()))
    lc = (torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none") * w).sum() / n
    hub = torch.nn.functional.huber_loss(p, torch.tensor(regs.astype(np.float64)), reduction="none")
    lr = (hub * (t * (w > 0))[..., None]).sum() / n
    (lc + 2.0 * lr).backward()
    assert total == pytest.approx(float((lc + 2.0 * lr).detach()), rel=1e-12)
    assert np.allclose(gl, x.grad.numpy(), atol=1e-7) and np.allclose(gr, p.grad.numpy(), atol=1e-7)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(4, 6, 2.0, 0.5, None), (3, 64, 0.5, 0.4, None), (2, 1, 2.0, 0.5, None),
                                 (2, 7, 0.45, 0.45, ((40, 40), (20, 20), (10, 10)))])
def test_hip_anchor_assign_matches_oracle_bit_for_bit(cfg):
    B, G, pos_iou, ign, levels = cfg
    if levels is None:
        a = det.make_anchors()
    else:
        a = det.make_anchors(levels, (0.15, 0.35, 0.75))         # 6300 anchors: several sweeps
    boxes, cls = random_boxes(B, G, seed=G)
    boxes[0, 0] = a[7]                                            # an exact hit
    cls[0, 0] = 2
    if G > 2:
        boxes[1, 1] = boxes[1, 0]                                 # a contested anchor
        cls[1, :2] = 1
        cls[B - 1, :] = 0                                         # an image without people
    want = oracle.anchor_assign(a, boxes, cls, pos_iou, ign)
    got = det.anchor_assign(torch.from_numpy(a).cuda(), torch.from_numpy(boxes).cuda(),
                            torch.from_numpy(cls).cuda(), pos_iou, ign)
    for g, w, name in zip(got, want, ("classes", "weights", "regs", "match")):
        assert np.array_equal(g.cpu().numpy(), w), name


@pytest.mark.gpu
def test_hip_box_loss_matches_oracle():
    rng = np.random.default_rng(9)
    a = det.make_anchors()
    boxes, cls = random_boxes(16, 8, 3)
    classes, weights, regs, _ = oracle.anchor_assign(a, boxes, cls)
    logit = rng.normal(size=classes.shape).astype(F32) * 4
    pred = (regs + rng.normal(size=regs.shape).astype(F32) * 0.8).astype(F32)
    total, c, r, gl, gr = oracle.box_loss_fwd_bwd(logit, pred, classes, weights, regs, 1.5)
    dev = [torch.from_numpy(x).cuda() for x in (logit, pred, classes, weights, regs)]
    loss4, g_logit, g_reg = det.box_loss_fwd_bwd(*dev, reg_weight=1.5)
    l4 = loss4.cpu().numpy()
    # float32 sums over 16 x 1500 anchors against float64: 1e-5 relative
    assert l4[0] == pytest.approx(total, rel=1e-5) and l4[1] == pytest.approx(c, rel=1e-5)
    assert l4[2] == pytest.approx(r, rel=1e-5) and l4[3] == float((classes > 0).sum())
    assert np.allclose(g_logit.cpu().numpy(), gl, rtol=1e-5, atol=1e-9)
    assert np.allclose(g_reg.cpu().numpy(), gr, rtol=1e-5, atol=1e-9)
    again, _, _ = det.box_loss_fwd_bwd(*dev, reg_weight=1.5)
    assert torch.equal(again, loss4)                               # ordered sums
    # no positives anywhere: N = 1, box term 0
    z = torch.zeros_like(dev[2])
    l0, _, g0 = det.box_loss_fwd_bwd(dev[0], dev[1], z, dev[3], dev[4])
    assert float(l0[2]) == 0.0 and float(l0[3]) == 0.0 and not g0.any()


def test_boxes_from_joints_oracle_known_answers():
    j = np.zeros((3, 4, 3), np.float32)
    j[0] = [[32, 64, 2], [96, 160, 1], [300, 300, 0], [64, 128, 2]]   # third joint unlabelled
    j[1, :, 2] = 0                                                    # nothing labelled
    j[2] = [[0, 0, 1], [319, 319, 1], [5, 5, 1], [9, 9, 1]]           # clipped on every side
    boxes, cls = oracle.boxes_from_joints(j, 320, 320, 0.05)
    assert cls.tolist() == [[1], [0], [1]]
    assert np.allclose(boxes[0, 0], [0.2 - 0.05, 0.1 - 0.05, 0.5 + 0.05, 0.3 + 0.05], atol=1e-7)
    assert boxes[1, 0].tolist() == [0, 0, 0, 0] and boxes[2, 0].tolist() == [0, 0, 1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,H,W,margin", [(64, 17, 320, 320, 0.05), (5, 3, 256, 192, 0.0),
                                            (130, 17, 384, 288, 0.1), (1, 1, 64, 64, 0.05)])
def test_hip_boxes_from_joints_matches_oracle_bit_for_bit(B, K, H, W, margin):
    import torch

    from pose_estimation_amd import detection, synthetic

    j = synthetic.joints(B, K, H, W, seed=B + K)
    j[B // 2, :, 2] = 0                                  # an image with nothing labelled
    got_b, got_c = detection.boxes_from_joints(torch.from_numpy(j).cuda(), H, W, margin)
    want_b, want_c = oracle.boxes_from_joints(j, H, W, margin)
    assert np.array_equal(got_b.cpu().numpy(), want_b)
    assert np.array_equal(got_c.cpu().numpy(), want_c)


@pytest.mark.gpu
def test_detection_argument_errors():
    a = torch.from_numpy(det.make_anchors()).cuda()
    boxes = torch.zeros((1, 65, 4), device="cuda")
    with pytest.raises(Exception, match="at most 64"):
        det.anchor_assign(a, boxes, torch.zeros((1, 65), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        det.anchor_assign(a, boxes[:, :4], torch.zeros((1, 4), dtype=torch.int64, device="cuda"))
