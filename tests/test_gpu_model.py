"""End-to-end inference parity: the HIP network against the torch-CPU restatement that rounds
to bf16 at the same points (weights, preprocessed image, every layer output).

Tolerance.  Each layer on its own is held to 4e-3 * max|ref| (and to bit-exactness on
integer operands) in test_gpu_conv.py.  End to end, a deep bf16 network amplifies the float32
accumulation ORDER: the CPU restatement run with float64 instead of float32 accumulation
(same operands, same rounding points) already moves the heat maps by 0.6 % (4-block net) to
1.3 % (ResNet-50) of max|heatmap|, because an order change flips an occasional bf16 rounding
and later layers magnify it.  The MFMA accumulation order is a third order, so the bound
stated here is: max|gpu - ref64| <= max(3 * spread, 4e-3 * max|ref|), where spread =
max|ref32 - ref64|, and relative L2 error <= 3 * the restatements' own relative L2 spread.
The arg-max of every map must agree wherever the reference's top-2 gap exceeds the bound.
"""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as to
import oracle
import pose_estimation_amd as pe
from pose_estimation_amd import model as pm
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 4e-3


def run_case(layers, B, H, W, seed):
    net = pm.PoseResNet("cuda:0", num_joints=17, layers=layers)
    state = pm.init_state(net.specs, seed=seed)
    pm.randomize_bn(state, seed=seed + 1)
    # head weights at a scale that keeps activations O(1) so relative error is meaningful
    for k in state:
        if k.startswith("deconv") and k.endswith(".weight"):
            state[k] = state[k] * 30.0
        if k == "final.weight":
            state[k] = state[k] * 100.0
    net.load_state(state)
    img = synthetic.images_u8(B, H, W, seed=1)
    got = net.predict(torch.from_numpy(img).cuda())
    ref32 = to.pose_resnet_forward(net.specs, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    ref = to.pose_resnet_forward(net.specs, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD,
                                 acc64=True)
    assert got.shape == ref.shape == (B, 17, H // 4, W // 4)
    return got, ref, ref32


def check(got, ref, ref32):
    g = got.cpu()
    scale = ref.abs().max().item()
    spread = (ref32 - ref).abs().max().item()
    spread_l2 = ((ref32 - ref).norm() / ref.norm()).item()
    err = (g - ref).abs().max().item()
    err_l2 = ((g - ref).norm() / ref.norm()).item()
    print(f"max|ref| {scale:.3e}  spread {spread:.3e} ({spread_l2:.2e} L2)  "
          f"gpu err {err:.3e} ({err_l2:.2e} L2)")
    assert scale > 1e-3, "degenerate reference output"
    bound = max(3 * spread, TOL * scale)
    assert err <= bound, f"max err {err:.3e} vs bound {bound:.3e}"
    assert err_l2 <= max(3 * spread_l2, TOL)
    TOLB = bound / scale
    # decode both; indices must agree unless the reference's best two values are within the
    # tolerance (then either is a legitimate arg-max of in-tolerance heat maps)
    xy, conf, idx = pe.decode(got)
    oxy, oconf, oidx = oracle.decode_heatmaps(ref.numpy())
    flat = ref.reshape(ref.shape[0], ref.shape[1], -1).numpy()
    top2 = np.sort(flat, axis=2)[:, :, -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2 * TOLB * scale
    same = idx.cpu().numpy() == oidx
    assert same[clear].all()
    # and the decode of the GPU heat maps is bit-exact against the oracle decode of the same maps
    gxy, gconf, gidx = oracle.decode_heatmaps(g.numpy())
    assert np.array_equal(idx.cpu().numpy(), gidx) and np.array_equal(xy.cpu().numpy(), gxy)


def test_small_resnet_pose_forward():
    check(*run_case((1, 1, 1, 1), B=3, H=64, W=64, seed=3))


def test_resnet50_pose_forward_reduced_resolution():
    check(*run_case((3, 4, 6, 3), B=2, H=128, W=96, seed=4))


def test_high_resolution_384x288_gives_96x72_maps():
    got, ref, ref32 = run_case((1, 1, 1, 1), B=1, H=384, W=288, seed=6)
    assert got.shape == (1, 17, 96, 72)
    check(got, ref, ref32)


def test_resnet50_pose_forward_full_resolution():
    check(*run_case((3, 4, 6, 3), B=1, H=256, W=192, seed=8))


def test_predict_is_deterministic_and_batch_independent():
    net = pm.PoseResNet("cuda:0", layers=(1, 1, 1, 1))
    state = pm.init_state(net.specs, seed=9)
    pm.randomize_bn(state, seed=10)
    net.load_state(state)
    img = torch.from_numpy(synthetic.images_u8(5, 64, 48, seed=2)).cuda()
    a = net.predict(img)
    b = net.predict(img)
    assert torch.equal(a, b)
    c = net.predict(img[1:3].contiguous())
    assert torch.equal(a[1:3], c)  # batch sharding changes nothing: inference needs no collective


def test_depth_first_front_is_bit_identical():
    """PE_RESNET_CHUNK: the stem, the pool and the first bottlenecks run over a few images at a
    time; every image's result is what the whole-batch pass gives."""
    net = pm.PoseResNet("cuda:0", layers=(2, 1, 1, 1))
    state = pm.init_state(net.specs, seed=11)
    pm.randomize_bn(state, seed=12)
    net.load_state(state)
    img = torch.from_numpy(synthetic.images_u8(6, 64, 48, seed=3)).cuda()
    ref = net.predict(img)
    for chunk, nblocks in ((2, 2), (3, 3), (1, 5)):
        net.front_chunk, net.front_blocks = chunk, nblocks
        assert torch.equal(net.predict(img), ref)
    net.front_chunk = 0


def test_chained_bottlenecks_are_bit_identical():
    """PE_RESNET_CHAIN: conv2 -> conv3 -> next conv1 of the 64-channel bottlenecks in one launch."""
    net = pm.PoseResNet("cuda:0", layers=(3, 1, 1, 1))
    state = pm.init_state(net.specs, seed=13)
    pm.randomize_bn(state, seed=14)
    net.load_state(state)
    img = torch.from_numpy(synthetic.images_u8(3, 64, 48, seed=4)).cuda()
    net.chain = False
    ref = net.predict(img)
    net.chain = True
    assert torch.equal(net.predict(img), ref)
    net.front_chunk, net.front_blocks = 1, 2     # the chain's last block writes into the slice
    assert torch.equal(net.predict(img), ref)
    net.chain, net.front_chunk = False, 0
