"""The full training step of the benchmark configuration: ResNet-50 (3, 4, 6, 3) + three
transposed convs, 256x192 images, 17 joints, batch 64 -- the network `bench.py --workload train`
times, checked here instead of only timed.

A CPU autograd pass of this size takes minutes, so parity with the CPU graph is asserted block
by block in test_gpu_train.py; this test pins what only the full-size step can show:
  * the loss + gradient kernel on the GPU's own batch-64 heat maps equals the NumPy oracle
    (gradient bit for bit, loss to 2e-6);
  * every parameter tensor of every one of the 57 layers receives a finite, non-zero gradient;
  * three optimizer steps lower the loss, keep the bf16 working weights equal to the rounded
    float32 master weights, and repeat bit for bit on a second run (fixed reduction orders).
"""
import numpy as np
import pytest
import torch

import oracle
from pose_estimation_amd import model as pm
from pose_estimation_amd import ops, synthetic
from pose_estimation_amd.training import TrainablePoseResNet

pytestmark = pytest.mark.gpu

B, H, W, K = 64, 256, 192, 17


def _net(seed=3):
    net = TrainablePoseResNet("cuda:0", num_joints=K, layers=(3, 4, 6, 3))
    state = pm.init_state(net.specs, seed=seed)
    for k in state:  # heat maps of O(0.1): a loss the optimizer can visibly lower in three steps
        if k.startswith("deconv") and k.endswith(".weight"):
            state[k] = state[k] * 30.0
        if k == "final.weight":
            state[k] = state[k] * 100.0
    net.load_state(state)
    return net


def _batch():
    img = torch.from_numpy(synthetic.images_u8(B, H, W, seed=1)).cuda()
    j_np = synthetic.joints(B, K, H, W, seed=2)
    m_np = synthetic.ignore_mask(B, H // 4, W // 4, seed=4)
    return img, j_np, m_np


def test_full_resnet50_batch64_step():
    img, j_np, m_np = _batch()
    joints = torch.from_numpy(j_np).cuda()
    mask = torch.from_numpy(m_np).cuda()

    net = _net()
    assert len(net.convs) == 57 and 33.9e6 < net.n_params < 34.2e6
    hm = net.forward_train(img)
    assert tuple(hm.shape) == (B, K, H // 4, W // 4) and hm.dtype == torch.float32
    loss, pj, grad, wgt = ops.mse_fwd_bwd(hm, joints=joints, mask=mask, stride=4.0, sigma=2.0,
                                          radius=6)
    net.backward(grad)
    torch.cuda.synchronize()

    # loss + gradient kernel on the GPU's own heat maps against the oracle
    t, w = oracle.render_targets(j_np, H // 4, W // 4, 4.0, 2.0, 6)
    ol, opj, og = oracle.mse_fwd_bwd(hm.cpu().numpy(), t, w, m_np)
    assert np.array_equal(wgt.cpu().numpy(), w)
    assert np.array_equal(grad.cpu().numpy(), og)
    assert loss.item() == pytest.approx(float(ol), rel=2e-6)
    assert np.allclose(pj.cpu().numpy(), opj, rtol=2e-6, atol=0)

    # every layer's gradient arrived
    grads = net.grads.cpu()
    assert torch.isfinite(grads).all()
    for c in net.convs:
        names = [c.name + ".weight"] + ([c.name + ".bn.gamma", c.name + ".bn.beta"] if c.bn
                                         else [c.name + ".bias"])
        for nm in names:
            n = net.view(net.grads, nm).float().norm().item()
            assert np.isfinite(n) and n > 0.0, f"{nm}: gradient norm {n}"

    # three steps, twice, from the same state
    runs, masters = [], []
    for rep in range(2):
        net = _net()
        run = []
        for _ in range(3):
            l, _ = net.train_step(img, joints, mask, lr=1e-3)
            run.append(l.item())
        torch.cuda.synchronize()
        assert torch.equal(net.work.float(), net.master.to(torch.bfloat16).float())
        assert torch.isfinite(net.master).all()
        runs.append(run)
        masters.append(net.master.clone())
        del net
    print("losses", runs[0])
    assert runs[0][0] == pytest.approx(float(ol), rel=2e-6)   # same state, same batch as above
    assert all(np.isfinite(runs[0])) and runs[0][-1] < runs[0][0]
    assert runs[0] == runs[1], "the step is not reproducible bit for bit"
    assert torch.equal(masters[0], masters[1])


def test_projection_beside_the_main_chain_changes_nothing():
    """The identity path's projection runs on side streams (forward beside conv1 and conv2,
    backward beside conv3..conv1); the same kernels on one stream must give the same bits."""
    img, j_np, m_np = _batch()
    joints = torch.from_numpy(j_np).cuda()
    mask = torch.from_numpy(m_np).cuda()
    out = []
    for beside in (True, False):
        net = _net()
        net.down_beside = beside
        losses = [net.train_step(img, joints, mask, lr=1e-3)[0].item() for _ in range(2)]
        torch.cuda.synchronize()
        out.append((losses, net.master.clone(), {k: v.clone() for k, v in net.bn_mean.items()}))
        del net
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])
    for k in out[0][2]:
        assert torch.equal(out[0][2][k], out[1][2][k]), k
