"""Training of the MobileNetV2 pose network (the reference's own backbone family,
nets/mobilenet_v2.py / model/mobilenet_pose.py, PoseEstimation.pyproj:31-32, both absent) on the
tape graph, against torch-CPU autograd of the same graph with bf16 rounding at the kernels'
storage points.  Parity unpinned beyond the label layout the notebook shows.

Single inverted-residual blocks, fed identical inputs and output gradients: outputs within one
bf16 ulp of the largest value, gradients 2e-2 relative L2 per tensor.  Whole network: relative L2
per tensor against max(3e-2, 3 x the float32-vs-bf16 spread of the CPU graph itself), as for the
other networks.  The depthwise and pointwise kernels are pinned bit for bit in test_gpu_bwd.py.
"""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as to
from pose_estimation_amd import mobilenet as mb
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import ops
from pose_estimation_amd import synthetic
from pose_estimation_amd.graph import Var
from pose_estimation_amd.model import randomize_bn

pytestmark = pytest.mark.gpu
SMALL = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 2, 2), (6, 64, 2, 2), (6, 96, 1, 1)]
PAIRS = mb.COCO_PAIRS[:5]


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def make_net(seed, setting=SMALL, pairs=PAIRS):
    net = mb.TrainablePoseMobileNetV2("cuda:0", num_joints=17, pairs=pairs, setting=setting)
    state = mb.init_state(net.blocks, 17, len(pairs), seed=seed)
    randomize_bn(state, seed=seed + 1)
    for k in ("heat.weight", "vec.weight"):
        if k in state:
            state[k] = state[k] * 30.0
    net.load_state(state)
    return net, state


def true_slice(net, name, got, ref_shape):
    """The true-size part of a channel-padded gradient, in the framework layout of `ref`."""
    L = net.by_name[name.rsplit(".", 1)[0] if not name.endswith((".bn.gamma", ".bn.beta"))
                    else name.rsplit(".", 2)[0]]
    if not name.endswith(".weight"):
        return got[:ref_shape[0]]
    if L.kind == "dw":                                  # [9, Cpad] -> [C,1,3,3]
        return got[:, :ref_shape[0]].t().reshape(ref_shape)
    if L.kind == "stem3":
        return None
    return got[:ref_shape[0], 0, 0, :ref_shape[1]].reshape(ref_shape)


def check_grads(net, rg, bound, fg=None):
    for nm, ref in rg.items():
        got = net.view(net.grads, nm).cpu()
        if nm == "stem.weight":
            L = net.by_name["stem"]
            pad = torch.zeros(L.cout, 3, 3, 3)
            pad[:ref.shape[0]] = ref
            g2, r2 = got * net.masks["stem"].cpu(), L.pack(pad)
            f2 = None
            if fg is not None:
                padf = torch.zeros(L.cout, 3, 3, 3)
                padf[:ref.shape[0]] = fg[nm]
                f2 = L.pack(padf)
        else:
            g2, r2 = true_slice(net, nm, got, tuple(ref.shape)), ref
            f2 = fg[nm] if fg is not None else None
            # nothing leaks into the padding
            assert got.abs().sum().item() == pytest.approx(g2.abs().sum().item(), rel=1e-6), nm
        e = rel_l2(g2, r2)
        s = rel_l2(f2, r2) if f2 is not None else 0.0
        print(f"   {nm:28s} rel L2 {e:.2e}   spread {s:.2e}")
        assert e <= max(bound, 3 * s), nm


def test_single_blocks_against_autograd():
    _single_blocks(*make_net(seed=21))


def test_single_blocks_with_padded_channel_counts():
    """Widths that are not whole 16-byte chunks (12, 20, 36 channels and their expansions) are
    padded to the next multiple of 8 with zero weights and switched-off norms: the padding holds
    exact zeros in every activation and gradient, and the true part matches autograd."""
    net, state = make_net(seed=23, setting=[(1, 12, 1, 1), (6, 20, 2, 2), (6, 36, 1, 1)])
    assert [(b.cin, b.mid, b.cout) for b in net.blocks][-1] == (20, 120, 36)
    assert net.by_name["block3.project"].cout == 40 and net.by_name["block1.expand"].cin == 16
    _single_blocks(net, state)


def _single_blocks(net, state):
    g = torch.Generator().manual_seed(5)
    B, H, W = 3, 16, 12
    for b in net.blocks:
        cin_p = mb.pad8(b.cin)
        x = torch.zeros(B, H, W, cin_p)
        x[..., :b.cin] = to.bf16r(torch.randn((B, H, W, b.cin), generator=g))
        net.grads.zero_()
        net.begin(True)
        xv = Var(x.to(torch.bfloat16).cuda(), True)
        out = net.block(b, xv)
        Ho, Wo = out.t.shape[1:3]
        dout = torch.zeros(B, Ho, Wo, mb.pad8(b.cout))
        dout[..., :b.cout] = to.bf16r(torch.randn((B, Ho, Wo, b.cout), generator=g))
        out.grad = dout.to(torch.bfloat16).cuda()
        net.backward({})
        torch.cuda.synchronize()
        rout, rdx, rg = to.mobilenet_block_train_reference(b, state, x[..., :b.cin],
                                                           dout[..., :b.cout])
        o = out.t.float().cpu()
        assert not o[..., b.cout:].any()                       # padded channels stay zero
        err = (o[..., :b.cout] - rout).abs().max().item()
        assert err <= 8e-3 * rout.abs().max().item() + 1e-6, b.name
        dx = xv.grad.float().cpu()
        assert not dx[..., b.cin:].any()
        e = rel_l2(dx[..., :b.cin], rdx)
        print(f"{b.name}: out err {err:.2e}, dx rel L2 {e:.2e}")
        assert e <= 2e-2, b.name
        check_grads(net, rg, 2e-2)


def test_whole_network_against_autograd():
    B, H, W = 4, 128, 96
    net, state = make_net(seed=3)
    img = synthetic.images_u8(B, H, W, seed=1)
    joints = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=2)).cuda()
    heat, vec = net.forward_train(torch.from_numpy(img).cuda())
    _, _, gh, _ = ops.mse_fwd_bwd(heat, joints=joints, stride=8.0, sigma=1.0, radius=1)
    _, _, gv = ops.vecmap_loss_fwd_bwd(vec, joints, PAIRS, 8.0, 1)
    gh, gv = gh * 1e3, gv * 1e3      # O(1e-3 .. 1) seeds: keeps bf16 gradients out of denormals
    net.backward({"heat": gh, "vec": gv})
    torch.cuda.synchronize()
    args = (net.blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, gh.cpu(), gv.cpu())
    rheat, rvec, rg = to.mobilenet_train_reference(*args)
    fheat, fvec, fg = to.mobilenet_train_reference(*args, round_bf16=False)
    e_h, s_h = rel_l2(heat.cpu(), rheat), rel_l2(fheat, rheat)
    e_v, s_v = rel_l2(vec.cpu(), rvec), rel_l2(fvec, rvec)
    print(f"heat {e_h:.3e} (spread {s_h:.3e})  vec {e_v:.3e} (spread {s_v:.3e})")
    assert e_h <= max(2e-2, 3 * s_h) and e_v <= max(2e-2, 3 * s_v)
    check_grads(net, rg, 3e-2, fg)


def test_training_reduces_the_loss_and_is_reproducible():
    B, H, W = 8, 96, 96
    losses = []
    for rep in range(2):
        net, state = make_net(seed=7)
        it = torch.from_numpy(synthetic.images_u8(B, H, W, seed=1)).cuda()
        jt = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=2)).cuda()
        run = []
        for _ in range(12):
            loss, _ = net.train_step(it, jt, lr=1e-3)
            run.append(loss.item())
        losses.append(run)
    print("losses", losses[0])
    assert losses[0][-1] < 0.8 * losses[0][0]
    assert losses[0] == losses[1]
    # channel padding stays exactly zero through the optimizer
    for L in net.layers:
        if L.kind == "conv":
            b = next(bb for bb in net.blocks if L.name.startswith(bb.name + "."))
            cin, cout = (b.cin, b.mid) if L.name.endswith(".expand") else (b.mid, b.cout)
            w = net.view(net.master, L.name + ".weight")
            assert not w[cout:].any() and not w[:, :, :, cin:].any(), L.name


def test_inference_mode_matches_the_inference_network():
    net, state = make_net(seed=17)
    ref = mb.PoseMobileNetV2("cuda:0", num_joints=17, num_pairs=len(PAIRS), setting=SMALL)
    ref.load_state(state)
    it = torch.from_numpy(synthetic.images_u8(2, 128, 96, seed=1)).cuda()
    (a, av), (b, bv) = net.predict(it), ref.predict(it)
    assert rel_l2(a.cpu(), b.cpu()) <= 1e-2 and rel_l2(av.cpu(), bv.cpu()) <= 1e-2


def test_full_network_step_runs_at_320x320():
    net = mb.TrainablePoseMobileNetV2("cuda:0", num_joints=17, pairs=mb.COCO_PAIRS)
    net.load_state(mb.init_state(net.blocks, 17, len(mb.COCO_PAIRS), seed=5))
    img = torch.from_numpy(synthetic.images_u8(4, 320, 320, seed=1)).cuda()
    joints = torch.from_numpy(synthetic.joints(4, 17, 320, 320, seed=2)).cuda()
    l0, _ = net.train_step(img, joints, lr=1e-3)
    l1, _ = net.train_step(img, joints, lr=1e-3)
    assert np.isfinite(l0.item()) and np.isfinite(l1.item())
    assert torch.isfinite(net.master).all()
    heat, vec = net.predict(img)
    assert tuple(heat.shape) == (4, 17, 40, 40) and tuple(vec.shape) == (4, 76, 40, 40)


def test_trainer_with_the_mobilenet_backbone(tmp_path):
    """The reference's configuration shape: MobileNetV2, stride-8 labels with 3x3 supports,
    heat maps + vector maps in the batch dict, checkpoint round trip."""
    from pose_estimation_amd import Trainer

    cfg = {"model_config": {"backbone": "mobilenet_v2", "heatmap_stride": 8, "sigma": 1.0,
                            "radius": 1, "pairs": [list(p) for p in PAIRS]},
           "train_config": {"batch_size": 4, "preprocess": {"image_resize": [96, 64]}}}
    tr = Trainer(cfg, device="cuda:0")
    tr.build_model(seed=5)
    feats, labels = next(iter(tr.get_features_labels_data(num_batches=1)))
    assert tuple(labels["heatmaps"].shape) == (4, 12, 8, 17)
    assert tuple(labels["vecmaps"].shape) == (4, 12, 8, 4 * len(PAIRS))
    losses = [tr.train_step(feats, labels).item() for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    heat, vec = tr.predict(feats["images"])
    assert tuple(heat.shape) == (4, 17, 12, 8) and tuple(vec.shape) == (4, 4 * len(PAIRS), 12, 8)
    path = str(tmp_path / "mb.pt")
    tr.save(path)
    tr2 = Trainer(cfg, device="cuda:0")
    tr2.build_model(seed=99)
    tr2.load(path)
    assert torch.equal(tr2.net.master, tr.net.master)
    h2, v2 = tr2.predict(feats["images"])
    assert torch.equal(h2, heat) and torch.equal(v2, vec)
    # one more step on both: identical trajectories after the round trip
    assert tr.train_step(feats, labels).item() == tr2.train_step(feats, labels).item()


def make_det_net(seed, B, H, W):
    net = mb.TrainablePoseMobileNetV2("cuda:0", num_joints=17, pairs=PAIRS, setting=SMALL,
                                      detection=True)
    state = mb.init_state(net.blocks, 17, len(PAIRS), seed=seed, det_blocks=net.det_blocks)
    randomize_bn(state, seed=seed + 1)
    for k in ("heat.weight", "vec.weight", "box1.weight", "box2.weight"):
        state[k] = state[k] * 30.0
    net.load_state(state)
    img = synthetic.images_u8(B, H, W, seed=1)
    joints = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=2)).cuda()
    return net, state, img, joints


def test_detection_branch_against_autograd():
    """The reference's third output (`labels['bboxes']`, test.ipynb:432): two stride-2 feature
    levels on the trunk output, each with a box head.  All four heads seeded at once, so the
    trunk output collects gradients from three consumers.  Same bounds as the whole-network test."""
    B, H, W = 8, 128, 128
    net, state, img, joints = make_det_net(31, B, H, W)
    heat, vec, maps = net.forward_train(torch.from_numpy(img).cuda())
    assert [tuple(m.shape) for m in maps] == [(B, 15, 8, 8), (B, 15, 4, 4)]
    _, _, gh, _ = ops.mse_fwd_bwd(heat, joints=joints, stride=8.0, sigma=1.0, radius=1)
    _, _, gv = ops.vecmap_loss_fwd_bwd(vec, joints, PAIRS, 8.0, 1)
    gh, gv = gh * 1e3, gv * 1e3
    g = torch.Generator().manual_seed(9)
    dboxes = [torch.randn(tuple(m.shape), generator=g) * 1e-2 for m in maps]
    net.backward({"heat": gh, "vec": gv, "box1": dboxes[0].cuda(), "box2": dboxes[1].cuda()})
    torch.cuda.synchronize()
    args = (net.blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, gh.cpu(), gv.cpu())
    kw = dict(det_blocks=net.det_blocks, dboxes=dboxes)
    rheat, rvec, rg, rmaps = to.mobilenet_train_reference(*args, **kw)
    fheat, fvec, fg, fmaps = to.mobilenet_train_reference(*args, round_bf16=False, **kw)
    for got, ref, f32, nm in [(heat, rheat, fheat, "heat"), (vec, rvec, fvec, "vec"),
                              (maps[0], rmaps[0], fmaps[0], "box1"),
                              (maps[1], rmaps[1], fmaps[1], "box2")]:
        e, s = rel_l2(got.cpu(), ref), rel_l2(f32, ref)
        print(f"{nm}: {e:.3e} (spread {s:.3e})")
        assert e <= max(2e-2, 3 * s), nm
    assert any(k.startswith("det2.") for k in rg) and "box2.weight" in rg
    check_grads(net, rg, 3e-2, fg)


def test_detection_training_steps():
    """End to end with labels made on the GPU: person boxes -> anchor targets -> objectness + box
    loss beside the heat-map and vector-map losses.  The box loss falls and the run repeats bit
    for bit."""
    from pose_estimation_amd import detection

    B, H, W = 8, 128, 128
    runs = []
    for rep in range(2):
        net, state, img, joints = make_det_net(41, B, H, W)
        anchors = torch.from_numpy(detection.make_anchors(((8, 8), (4, 4)))).cuda()
        assert anchors.shape[0] == 8 * 8 * 3 + 4 * 4 * 3
        boxes = torch.tensor([[[0.2, 0.25, 0.8, 0.7]]], device="cuda").repeat(B, 1, 1)
        boxes = boxes + 0.01 * torch.arange(B, device="cuda").view(B, 1, 1)
        cls = torch.ones((B, 1), dtype=torch.int32, device="cuda")
        classes, weights, regs, _ = detection.anchor_assign(anchors, boxes.contiguous(), cls)
        it = torch.from_numpy(img).cuda()
        hist = []
        for _ in range(10):
            loss, _ = net.train_step(it, joints, lr=2e-3, det_labels=(classes, weights, regs))
            hist.append((loss.item(), net.last_box_loss[0].item()))
        runs.append(hist)
        net.fold_bn()
        heat, vec, (logit, reg) = net.predict(it)
        assert tuple(logit.shape) == (B, anchors.shape[0]) and tuple(reg.shape) == (B, anchors.shape[0], 4)
        assert torch.isfinite(logit).all() and torch.isfinite(reg).all()
    assert runs[0] == runs[1]
    box = [h[1] for h in runs[0]]
    print("box loss:", " ".join(f"{v:.4f}" for v in box))
    assert box[-1] < box[0]
    with pytest.raises(ValueError):
        net.train_step(it, joints)          # detection branch without its labels


def test_trainer_with_the_reference_label_set():
    """All three label groups of the reference's batches (heatmaps, vecmaps, bboxes:
    test.ipynb:432, 623, 662) at its resolution: 320x320 -> 40x40 maps and 1500 anchors."""
    from pose_estimation_amd import Trainer, detection

    cfg = {"model_config": {"backbone": "mobilenet_v2", "heatmap_stride": 8, "sigma": 1.0,
                            "radius": 1, "pairs": [list(p) for p in PAIRS], "detection": True},
           "train_config": {"batch_size": 4, "preprocess": {"image_resize": [320, 320]}}}
    tr = Trainer(cfg, device="cuda:0")
    tr.build_model(seed=5)
    feats, labels = next(iter(tr.get_features_labels_data(num_batches=1)))
    bb = labels["bboxes"]
    assert tuple(bb["classes"].shape) == (4, 1500) and tuple(bb["regs"].shape) == (4, 1500, 4)
    assert bb["classes"].sum(dim=1).tolist() == [1.0] * 4      # one positive anchor per person
    # the boxes are the extent of the labelled joints (+ 5 % of the image), restated in NumPy
    j = labels["joints"].cpu().numpy()
    for b in range(4):
        vis = j[b, :, 2] > 0
        want = np.array([j[b, vis, 1].min() / 320 - 0.05, j[b, vis, 0].min() / 320 - 0.05,
                         j[b, vis, 1].max() / 320 + 0.05, j[b, vis, 0].max() / 320 + 0.05],
                        np.float32).clip(0, 1)
        assert np.allclose(bb["boxes"][b, 0].cpu().numpy(), want, atol=1e-6)
    # and the positive anchor's regression row is that box (test.ipynb:571-572)
    pos = bb["classes"][0].argmax().item()
    assert torch.allclose(bb["regs"][0, pos], bb["boxes"][0, 0])
    # the per-level groups the notebook indexes (test.ipynb:538-541, 571-572): feat_1 = the
    # 20x20x3 anchors, feat_2 = the 10x10x3, views of the flat arrays
    f1, f2 = bb["feat_1"], bb["feat_2"]
    assert tuple(f1["classes"].shape) == (4, 1200) and tuple(f2["weights"].shape) == (4, 300)
    assert tuple(f1["regs"].shape) == (4, 1200, 4) and tuple(f2["regs"].shape) == (4, 300, 4)
    assert f1["classes"][0].reshape(20, 20, 3).shape == (20, 20, 3)        # the notebook's reshape
    assert torch.equal(torch.cat([f1["classes"], f2["classes"]], 1), bb["classes"])
    assert torch.equal(torch.cat([f1["regs"], f2["regs"]], 1), bb["regs"])
    assert f1["weights"].data_ptr() == bb["weights"].data_ptr()
    regs1 = f1["regs"][0].cpu().numpy()                                     # test.ipynb:571-572
    rows = regs1[regs1[:, 2] - regs1[:, 0] > 0]
    both = np.concatenate([rows, f2["regs"][0].cpu().numpy()[f2["classes"][0].cpu().numpy() > 0]])
    assert len(both) == 1 and np.array_equal(both[0], bb["boxes"][0, 0].cpu().numpy())
    losses = [tr.train_step(feats, labels).item() for _ in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    heat, vec, (logit, reg) = tr.predict(feats["images"])
    assert tuple(heat.shape) == (4, 17, 40, 40) and tuple(logit.shape) == (4, 1500)
    # frames of another size with augmentation: boxes are rebuilt from the transformed joints
    cfg["train_config"]["augmentation"] = {"flip_left_right": True, "scale_range": [0.8, 1.2]}
    tr2 = Trainer(cfg, device="cuda:0")
    tr2.build_model(seed=5)
    big = torch.from_numpy(synthetic.images_u8(4, 480, 640, seed=3)).cuda()
    jb = torch.from_numpy(synthetic.joints(4, 17, 480, 640, seed=4)).cuda()
    assert np.isfinite(tr2.train_step(big, jb).item())
    assert detection.boxes_from_joints(torch.zeros(2, 17, 3).cuda(), 320, 320)[1].tolist() == [[0], [0]]
