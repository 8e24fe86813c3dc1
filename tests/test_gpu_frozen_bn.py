"""Training with frozen batch-norm statistics (running mean / variance as constants; gamma and
beta still train) against torch-CPU autograd of the same graph with eval-mode normalisation.
Bounds as in test_gpu_train.py: relative L2 per parameter tensor, at most 3x the float32-vs-bf16
spread of the CPU graph itself, floor 3e-2."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle as to
from pose_estimation_amd import model as pm
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import ops, synthetic
from pose_estimation_amd.training import TrainablePoseResNet

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_frozen_statistics_forward_backward_against_autograd():
    B, H, W = 4, 64, 64
    net = TrainablePoseResNet("cuda:0", num_joints=17, layers=(1, 1, 1, 1))
    state = pm.init_state(net.specs, seed=3)
    pm.randomize_bn(state, seed=4)
    for k in state:
        if k.startswith("deconv") and k.endswith(".weight"):
            state[k] = state[k] * 30.0
        if k == "final.weight":
            state[k] = state[k] * 100.0
    net.load_state(state)
    net.freeze_batch_norm(True)
    img = synthetic.images_u8(B, H, W, seed=1)
    joints = synthetic.joints(B, 17, H, W, seed=2)
    mask = synthetic.ignore_mask(B, H // 4, W // 4, seed=4)
    rm0 = net.bn_mean["stem"].clone()

    hm = net.forward_train(torch.from_numpy(img).cuda())
    loss, _, grad, _ = ops.mse_fwd_bwd(hm, joints=torch.from_numpy(joints).cuda(),
                                       mask=torch.from_numpy(mask).cuda(), stride=4.0, sigma=2.0,
                                       radius=6)
    net.backward(grad)
    torch.cuda.synchronize()
    assert torch.equal(net.bn_mean["stem"], rm0), "frozen running statistics moved"

    t, w = oracle.render_targets(joints, H // 4, W // 4, 4.0, 2.0, 6)
    args = (net.specs, state, img, t, w, mask, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    rl, rhm, rg = to.pose_resnet_train_reference(*args, frozen_bn=True)
    fl, fhm, fg = to.pose_resnet_train_reference(*args, frozen_bn=True, round_bf16=False)
    bl, _, bg = to.pose_resnet_train_reference(*args)      # batch statistics: a different graph
    e_hm, s_hm = rel_l2(hm.cpu(), rhm), rel_l2(fhm, rhm)
    print(f"heatmaps: gpu-vs-bf16ref {e_hm:.3e}, f32ref-vs-bf16ref {s_hm:.3e}")
    assert e_hm <= max(2e-2, 3 * s_hm)
    assert abs(loss.item() - rl) <= max(2e-2, 3 * abs(fl - rl) / abs(rl)) * abs(rl)
    far = 0
    for c in net.convs:
        names = [c.name + ".weight"] + ([c.name + ".bn.gamma", c.name + ".bn.beta"] if c.bn
                                         else [c.name + ".bias"])
        for nm in names:
            got = net.view(net.grads, nm).cpu()
            ref, f32, bat = rg[nm], fg[nm], bg[nm]
            if nm.endswith(".weight"):
                ref, f32, bat = (pm.pack_weight(c, x) for x in (ref, f32, bat))
            ref, f32, bat = (x.reshape(got.shape) for x in (ref, f32, bat))
            e, s = rel_l2(got, ref), rel_l2(f32, ref)
            print(f"{nm:28s} gpu-vs-bf16ref {e:.3e}   f32ref-vs-bf16ref {s:.3e}   "
                  f"batch-stat graph {rel_l2(bat, ref):.2e}")
            assert e <= max(3e-2, 3 * s), nm
            far += rel_l2(bat, ref) > 0.1
    # the test can tell the two graphs apart: most tensors of the batch-statistics graph are
    # far from the frozen one
    assert far > len(net.convs)
    # an optimizer step in frozen mode leaves the running statistics alone
    net.train_step(torch.from_numpy(img).cuda(), torch.from_numpy(joints).cuda(), lr=1e-3)
    torch.cuda.synchronize()
    assert torch.equal(net.bn_mean["stem"], rm0)
    assert np.isfinite(net.master.cpu().numpy()).all()
