"""GPU parity of the implicit-GEMM conv / deconv / stem kernels against the torch-CPU float32
restatement (oracle/torch_oracle.py).

Two kinds of comparison:
  * small-integer operands, where every product and partial sum is exact in float32, so the
    bf16 result must match bit for bit whatever the accumulation order (asymmetric weights:
    a transposed or mis-indexed tile cannot pass);
  * random bf16 operands with the tolerance |got - ref| <= 4e-3 * max|ref| (one bf16 rounding
    of the result is 2^-9 relative; fp32 accumulation-order noise is far below that).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as to
from pose_estimation_amd import nn as pnn

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
TOL = 4e-3


def rand_bf16(shape, gen, scale=1.0):
    return to.bf16r(torch.randn(shape, generator=gen) * scale)


def int_tensor(shape, gen, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=gen).to(torch.float32)


def close(got_bf16_gpu, ref_f32, tol=TOL):
    got = got_bf16_gpu.float().cpu()
    err = (got - ref_f32).abs().max().item()
    ref_max = ref_f32.abs().max().item()
    assert err <= tol * ref_max + 1e-6, f"max err {err} vs max |ref| {ref_max}"


CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad
    (2, 16, 12, 64, 64, 1, 1, 0),
    (2, 16, 12, 64, 256, 1, 1, 0),
    (2, 16, 12, 256, 64, 1, 1, 0),
    (2, 16, 12, 64, 64, 3, 1, 1),
    (3, 16, 12, 128, 128, 3, 2, 1),
    (2, 16, 12, 256, 512, 1, 2, 0),
    (1, 8, 6, 512, 512, 3, 1, 1),
    (5, 7, 5, 64, 192, 3, 1, 1),      # odd sizes, partial tiles in M and N
    (1, 64, 48, 64, 68, 3, 1, 1),     # N not a multiple of 16
    (2, 9, 11, 128, 36, 3, 2, 1),     # odd H, W with stride 2; N < 64
    (2, 16, 12, 96, 24, 1, 1, 0),     # K per tap not a multiple of 64: part-filled last K step
    (2, 16, 12, 24, 144, 1, 1, 0),    # a single, part-filled K step
    (3, 10, 10, 160, 960, 1, 1, 0),
    (2, 9, 7, 96, 40, 3, 1, 1),       # ... and with more than one tap
    (2, 12, 10, 144, 64, 3, 2, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_integer_operands_exact(case):
    B, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = int_tensor((B, H, W, Cin), g, -2, 2)
    w = int_tensor((Cout, Cin, K, K), g, -1, 1)
    ref = to.bf16r(to.conv2d(x, w, stride=s, pad=p))
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), stride=s, pad=p)
    assert got.shape == ref.shape
    assert torch.equal(got.float().cpu(), ref)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_random_operands_with_epilogue(case):
    B, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(1 + hash(case) % 2**31)
    x = rand_bf16((B, H, W, Cin), g)
    w = rand_bf16((Cout, Cin, K, K), g, (2.0 / (Cin * K * K)) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    Ho = (H + 2 * p - K) // s + 1
    Wo = (W + 2 * p - K) // s + 1
    res = rand_bf16((B, Ho, Wo, Cout), g)
    ref = to.conv2d(x, w, scale, shift, res, s, p, relu=True)
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), scale.cuda(), shift.cuda(),
                     res.to(BF16).cuda(), s, p, relu=True)
    close(got, ref)
    # no epilogue at all
    ref2 = to.conv2d(x, w, stride=s, pad=p)
    got2 = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), stride=s, pad=p)
    close(got2, ref2)


def test_conv_f32_nchw_output_with_bias():
    """The final 1x1 conv: 256 -> 17 channels, float32 NCHW heat maps."""
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 3, 64, 48, 256, 17
    x = rand_bf16((B, H, W, Cin), g)
    w = rand_bf16((Cout, Cin, 1, 1), g, 0.05)
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = to.conv2d(x, w, None, bias).permute(0, 3, 1, 2).contiguous()
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), None, bias.cuda(),
                     out_f32_nchw=True)
    assert got.dtype == torch.float32 and got.shape == (B, Cout, H, W)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item()  # float32 output: accumulation order only
    # integer operands: exact
    xi = int_tensor((B, H, W, Cin), g, -2, 2)
    wi = int_tensor((Cout, Cin, 1, 1), g, -1, 1)
    refi = to.conv2d(xi, wi).permute(0, 3, 1, 2).contiguous()
    goti = pnn.conv2d(xi.to(BF16).cuda(), pnn.pack_conv_weight(wi).cuda(), out_f32_nchw=True)
    assert torch.equal(goti.cpu(), refi)


@pytest.mark.parametrize("case", [(2, 8, 6, 64, 64), (1, 8, 6, 2048, 256), (3, 16, 12, 256, 256),
                                  (2, 5, 7, 128, 72)])
def test_deconv_phases(case):
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(7 + Cin)
    xi = int_tensor((B, H, W, Cin), g, -2, 2)
    wi = int_tensor((Cin, Cout, 4, 4), g, -1, 1)
    ref = to.bf16r(to.deconv4x4s2(xi, wi))
    got = pnn.deconv4x4s2(xi.to(BF16).cuda(), pnn.pack_deconv_weight(wi).cuda())
    assert got.shape == (B, 2 * H, 2 * W, Cout)
    assert torch.equal(got.float().cpu(), ref)
    x = rand_bf16((B, H, W, Cin), g)
    w = rand_bf16((Cin, Cout, 4, 4), g, (1.0 / (Cin * 4)) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = to.deconv4x4s2(x, w, scale, shift, relu=True)
    got = pnn.deconv4x4s2(x.to(BF16).cuda(), pnn.pack_deconv_weight(w).cuda(), scale.cuda(),
                          shift.cuda(), relu=True)
    close(got, ref)


@pytest.mark.parametrize("case", [(3, 16, 12, 256, 17, 256), (2, 5, 7, 128, 17, 256),
                                  (1, 32, 24, 256, 3, 256), (5, 9, 6, 64, 32, 256),
                                  (2, 8, 6, 2048, 16, 256), (3, 7, 9, 64, 17, 128),
                                  (2, 6, 5, 128, 5, 512)])
def test_deconv_with_fused_pointwise_head(case):
    """pe_deconv4x4s2_head_fwd_bf16: the last transposed conv of the pose head and the final 1x1
    conv in one launch.  Bit for bit what the two launches give (the tile written to LDS holds
    the values the deconv would store, and the head's MFMAs walk the channels in the same
    order), and against the oracle: exact on integer operands."""
    B, H, W, Cin, K, Cout = case
    g = torch.Generator().manual_seed(31 + Cin + K)
    x = rand_bf16((B, H, W, Cin), g)
    w = rand_bf16((Cin, Cout, 4, 4), g, (1.0 / (Cin * 4)) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    hw = rand_bf16((K, Cout, 1, 1), g, 0.05)
    bias = torch.randn(K, generator=g) * 0.1
    xg, wg = x.to(BF16).cuda(), pnn.pack_deconv_weight(w).cuda()
    hwg = pnn.pack_conv_weight(hw).cuda()
    mid = pnn.deconv4x4s2(xg, wg, scale.cuda(), shift.cuda(), relu=True)
    want = pnn.conv2d(mid, hwg, None, bias.cuda(), out_f32_nchw=True)
    got = pnn.deconv4x4s2_head(xg, wg, scale.cuda(), shift.cuda(), hwg, bias.cuda(), relu=True)
    assert got.shape == want.shape == (B, K, 2 * H, 2 * W)
    assert torch.equal(got, want)
    # without scale / shift / bias / ReLU, integer operands: exact against the oracle
    xi = int_tensor((B, H, W, Cin), g, -2, 2)
    wi = int_tensor((Cin, Cout, 4, 4), g, -1, 1)
    hi = int_tensor((K, Cout, 1, 1), g, -1, 1)
    midr = to.bf16r(to.deconv4x4s2(xi, wi))
    ref = to.conv2d(midr, hi).permute(0, 3, 1, 2).contiguous()
    goti = pnn.deconv4x4s2_head(xi.to(BF16).cuda(), pnn.pack_deconv_weight(wi).cuda(), None, None,
                                pnn.pack_conv_weight(hi).cuda(), None, relu=False)
    assert torch.equal(goti.cpu(), ref)


def test_deconv_head_rejects_what_it_does_not_cover():
    g = torch.Generator().manual_seed(3)
    x = rand_bf16((1, 4, 4, 64), g).to(BF16).cuda()
    w64 = pnn.pack_deconv_weight(rand_bf16((64, 64, 4, 4), g)).cuda()
    hw = pnn.pack_conv_weight(rand_bf16((17, 64, 1, 1), g)).cuda()
    with pytest.raises(RuntimeError, match="output channels"):
        pnn.deconv4x4s2_head(x, w64, None, None, hw)


def test_preprocess_and_stem():
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 64, 48
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    xs = pnn.preprocess_u8(img.cuda())
    x3 = to.preprocess_u8(img.numpy(), pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    want = to.space_to_depth16(x3)
    assert xs.shape == (B, H // 2, W // 2, 16)
    assert np.array_equal(xs.float().cpu().numpy(), want)  # bit-exact preprocessing
    w = rand_bf16((64, 3, 7, 7), g, 0.1)
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.1
    ref = to.conv2d(torch.from_numpy(x3), w, scale, shift, None, 2, 3, relu=True)
    got = pnn.stem7x7s2(xs, pnn.pack_stem_weight(w).cuda(), scale.cuda(), shift.cuda(), relu=True)
    assert got.shape == (B, H // 2, W // 2, 64)
    close(got, ref)
    # integer image / weights through the stem alone: exact
    xi = int_tensor((B, H, W, 3), g, -2, 2)
    wi = int_tensor((64, 3, 7, 7), g, -1, 1)
    refi = to.bf16r(to.conv2d(xi, wi, stride=2, pad=3))
    xsi = torch.from_numpy(to.space_to_depth16(xi.numpy())).to(BF16).cuda()
    goti = pnn.stem7x7s2(xsi, pnn.pack_stem_weight(wi).cuda(), relu=False)
    assert torch.equal(goti.float().cpu(), refi)


@pytest.mark.parametrize("case", [(2, 64, 48), (5, 16, 16), (1, 7, 5), (3, 33, 50), (300, 16, 12)])
def test_conv3x3_64_channels_resident_filter_kernel(case):
    """pe_conv3x3_c64_fwd_bf16 against the engine on the same operands, bit for bit (whole and
    part-filled 16 x 16 tiles, more tiles than CUs), and against the oracle: exact on integer
    operands."""
    B, H, W = case
    g = torch.Generator().manual_seed(5 + H + W)
    x = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    w = rand_bf16((64, 64, 3, 3), g, (2.0 / 576) ** 0.5)
    wp = pnn.pack_conv_weight(w).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    for sc, sh, relu in ((scale, shift, 1), (None, None, 0), (None, shift, 2)):
        want = pnn.conv2d(x, wp, sc, sh, None, 1, 1, relu)
        got = pnn.conv3x3_c64(x, wp, sc, sh, relu)
        assert torch.equal(got, want), (sc is not None, relu)
    if B <= 5:
        xi = int_tensor((B, H, W, 64), g, -2, 2)
        wi = int_tensor((64, 64, 3, 3), g, -1, 1)
        ref = to.bf16r(to.conv2d(xi, wi, stride=1, pad=1))
        goti = pnn.conv3x3_c64(xi.to(BF16).cuda(), pnn.pack_conv_weight(wi).cuda(), relu=0)
        assert torch.equal(goti.float().cpu(), ref)


@pytest.mark.parametrize("case", [(2, 96, 72), (5, 16, 16), (1, 7, 5), (3, 33, 50), (300, 16, 12)])
def test_conv3x3_32_channels_resident_filter_kernel(case):
    """pe_conv3x3_c32_fwd_bf16 against the engine on the same operands (narrow weight packing),
    bit for bit, with and without scale / shift, residual and the activations; exact against
    the oracle on integer operands."""
    B, H, W = case
    g = torch.Generator().manual_seed(31 + H + W)
    x = rand_bf16((B, H, W, 32), g).to(BF16).cuda()
    w = rand_bf16((32, 32, 3, 3), g, (2.0 / 288) ** 0.5)
    wp = pnn.pack_conv_weight_narrow(w).cuda()
    scale = (torch.rand(32, generator=g) + 0.5).cuda()
    shift = (torch.randn(32, generator=g) * 0.1).cuda()
    res = rand_bf16((B, H, W, 32), g).to(BF16).cuda()
    for sc, sh, r, relu in ((scale, shift, None, 1), (scale, shift, res, 1), (None, None, res, 0),
                            (None, shift, None, 2), (None, None, None, 0)):
        got = pnn.conv3x3_c32(x, wp, sc, sh, r, relu)
        d = pnn.ConvDesc(B, H, W, 32, 32, 3, 3, 1, 1, relu, 0)
        want = torch.empty_like(got)
        pnn.check(pnn.lib().pe_conv2d_fwd_bf16(ctypes.addressof(d), x.data_ptr(), wp.data_ptr(),
                                               pnn._ptr(sc), pnn._ptr(sh), pnn._ptr(r),
                                               want.data_ptr(), pnn._stream(x)))
        assert torch.equal(got, want), (sc is not None, r is not None, relu)
    if B <= 5:
        xi = int_tensor((B, H, W, 32), g, -2, 2)
        wi = int_tensor((32, 32, 3, 3), g, -1, 1)
        ref = to.bf16r(to.conv2d(xi, wi, stride=1, pad=1))
        goti = pnn.conv3x3_c32(xi.to(BF16).cuda(), pnn.pack_conv_weight_narrow(wi).cuda(), relu=0)
        assert torch.equal(goti.float().cpu(), ref)


@pytest.mark.parametrize("case", [(2, 64, 48), (5, 16, 16), (1, 7, 5), (3, 33, 50), (300, 16, 12)])
def test_conv3x3_64_channels_with_the_pointwise_layer_in_the_same_launch(case):
    """pe_conv3x3_c64_pw_fwd_bf16 (conv2 + conv3 of a first-stage bottleneck, residual and ReLU)
    against the two engine launches it replaces, bit for bit, with and without the last ReLU;
    exact against the oracle on integer operands."""
    B, H, W = case
    g = torch.Generator().manual_seed(9 + H + W)
    x = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    w = pnn.pack_conv_weight(rand_bf16((64, 64, 3, 3), g, (2.0 / 576) ** 0.5)).cuda()
    w3 = pnn.pack_conv_weight(rand_bf16((256, 64, 1, 1), g, (2.0 / 64) ** 0.5)).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    scale3 = (torch.rand(256, generator=g) + 0.5).cuda()
    shift3 = (torch.randn(256, generator=g) * 0.1).cuda()
    res = rand_bf16((B, H, W, 256), g).to(BF16).cuda()
    for sc3, sh3, relu3 in ((scale3, shift3, True), (None, None, False), (None, shift3, True)):
        mid = pnn.conv2d(x, w, scale, shift, None, 1, 1, True)
        want = pnn.conv2d(mid, w3, sc3, sh3, res, 1, 0, relu3)
        got = pnn.conv3x3_c64_pw(x, w, scale, shift, w3, sc3, sh3, res, relu3)
        assert torch.equal(got, want), (sc3 is not None, relu3)
    if B <= 5:
        xi = int_tensor((B, H, W, 64), g, -2, 2)
        wi = int_tensor((64, 64, 3, 3), g, -1, 1)
        w3i = int_tensor((256, 64, 1, 1), g, -1, 1)
        ri = int_tensor((B, H, W, 256), g, -3, 3)
        one = torch.ones(64).cuda()
        zero = torch.zeros(64).cuda()
        mid = torch.relu(to.bf16r(to.conv2d(xi, wi, stride=1, pad=1)))
        ref = to.bf16r(torch.relu(to.conv2d(mid, w3i, stride=1, pad=0) + ri))
        goti = pnn.conv3x3_c64_pw(xi.to(BF16).cuda(), pnn.pack_conv_weight(wi).cuda(), one, zero,
                                  pnn.pack_conv_weight(w3i).cuda(), None, None, ri.to(BF16).cuda())
        assert torch.equal(goti.float().cpu(), ref)


@pytest.mark.parametrize("case", [(2, 64, 48), (5, 16, 16), (1, 7, 5), (3, 33, 50), (300, 16, 12)])
def test_conv3x3_pointwise_and_the_next_blocks_pointwise_in_the_same_launch(case):
    """pe_conv3x3_c64_pw_next_fwd_bf16 (conv2 + conv3 of a bottleneck and conv1 of the next one)
    against the three engine launches it replaces, both outputs bit for bit; exact against the
    oracle on integer operands."""
    B, H, W = case
    g = torch.Generator().manual_seed(33 + H + W)
    x = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    w = pnn.pack_conv_weight(rand_bf16((64, 64, 3, 3), g, (2.0 / 576) ** 0.5)).cuda()
    w3 = pnn.pack_conv_weight(rand_bf16((256, 64, 1, 1), g, (2.0 / 64) ** 0.5)).cuda()
    w1 = pnn.pack_conv_weight(rand_bf16((64, 256, 1, 1), g, (2.0 / 256) ** 0.5)).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    sc3, sh3 = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
    sc1, sh1 = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    res = rand_bf16((B, H, W, 256), g).to(BF16).cuda()
    for a3, b3, a1, b1 in ((sc3, sh3, sc1, sh1), (None, None, None, None), (None, sh3, sc1, None)):
        mid = pnn.conv2d(x, w, scale, shift, None, 1, 1, True)
        want = pnn.conv2d(mid, w3, a3, b3, res, 1, 0, True)
        want1 = pnn.conv2d(want, w1, a1, b1, None, 1, 0, True)
        got, got1 = pnn.conv3x3_c64_pw_next(x, w, scale, shift, w3, a3, b3, res, w1, a1, b1)
        assert torch.equal(got, want), (a3 is not None, b3 is not None)
        assert torch.equal(got1, want1), (a1 is not None, b1 is not None)
    if B <= 5:
        xi = int_tensor((B, H, W, 64), g, -2, 2)
        wi = int_tensor((64, 64, 3, 3), g, -1, 1)
        w3i = int_tensor((256, 64, 1, 1), g, -1, 1)
        w1i = int_tensor((64, 256, 1, 1), g, -1, 1)
        ri = int_tensor((B, H, W, 256), g, -3, 3)
        one = torch.ones(64).cuda()
        zero = torch.zeros(64).cuda()
        mid = torch.relu(to.bf16r(to.conv2d(xi, wi, stride=1, pad=1)))
        ref = to.bf16r(torch.relu(to.conv2d(mid, w3i, stride=1, pad=0) + ri))
        ref1 = to.bf16r(torch.relu(to.conv2d(ref, w1i, stride=1, pad=0)))
        goti, got1i = pnn.conv3x3_c64_pw_next(
            xi.to(BF16).cuda(), pnn.pack_conv_weight(wi).cuda(), one, zero,
            pnn.pack_conv_weight(w3i).cuda(), None, None, ri.to(BF16).cuda(),
            pnn.pack_conv_weight(w1i).cuda(), None, None)
        assert torch.equal(goti.float().cpu(), ref)
        assert torch.equal(got1i.float().cpu(), ref1)


@pytest.mark.parametrize("case", [(2, 64, 48), (5, 16, 16), (1, 7, 5), (3, 33, 50), (300, 16, 12)])
def test_conv3x3_pointwise_with_the_projected_residual_in_the_same_launch(case):
    """pe_conv3x3_c64_pw_proj_fwd_bf16 (conv2 + conv3 + the projection of the skip path of a
    stage's first bottleneck) against the three engine launches it replaces, bit for bit."""
    B, H, W = case
    g = torch.Generator().manual_seed(21 + H + W)
    x = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    x0 = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    w = pnn.pack_conv_weight(rand_bf16((64, 64, 3, 3), g, (2.0 / 576) ** 0.5)).cuda()
    w3 = pnn.pack_conv_weight(rand_bf16((256, 64, 1, 1), g, (2.0 / 64) ** 0.5)).cuda()
    wd = pnn.pack_conv_weight(rand_bf16((256, 64, 1, 1), g, (2.0 / 64) ** 0.5)).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    sc3, sh3 = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
    scd, shd = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
    for a3, b3, ad, bd in ((sc3, sh3, scd, shd), (None, None, None, None), (None, sh3, scd, None)):
        idn = pnn.conv2d(x0, wd, ad, bd, None, 1, 0, False)
        mid = pnn.conv2d(x, w, scale, shift, None, 1, 1, True)
        want = pnn.conv2d(mid, w3, a3, b3, idn, 1, 0, True)
        got = pnn.conv3x3_c64_pw_proj(x, w, scale, shift, w3, a3, b3, x0, wd, ad, bd)
        assert torch.equal(got, want), (a3 is not None, ad is not None)


def test_conv3x3_pointwise_rejects_what_it_does_not_cover():
    x = torch.zeros((1, 16, 16, 64), dtype=BF16).cuda()
    w = torch.zeros((64, 3, 3, 64), dtype=BF16).cuda()
    s = torch.ones(64).cuda()
    w3 = torch.zeros((128, 1, 1, 64), dtype=BF16).cuda()
    res = torch.zeros((1, 16, 16, 128), dtype=BF16).cuda()
    with pytest.raises(RuntimeError, match="256 outputs"):
        pnn.conv3x3_c64_pw(x, w, s, s, w3, None, None, res)
    w3 = torch.zeros((256, 1, 1, 64), dtype=BF16).cuda()
    res = torch.zeros((1, 16, 16, 256), dtype=BF16).cuda()
    with pytest.raises(RuntimeError, match="scale / shift"):
        pnn.conv3x3_c64_pw(x, w, None, None, w3, None, None, res)


@pytest.mark.parametrize("case", [(2, 64, 48), (3, 128, 96), (1, 16, 12), (2, 40, 72), (1, 8, 4),
                                  (5, 36, 68)])
def test_stem_and_pool_in_one_launch(case):
    """pe_stem7x7s2_pool_fwd_bf16 against the two launches it replaces, bit for bit (sizes that
    are and are not whole numbers of 4 x 16 pooled tiles), and against the oracle on integer
    operands."""
    B, H, W = case                      # image size; the stem works at H/2 x W/2
    g = torch.Generator().manual_seed(17 + H + W)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    xs = pnn.preprocess_u8(img.cuda())
    w = rand_bf16((64, 3, 7, 7), g, 0.1)
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.3
    wp = pnn.pack_stem_weight(w).cuda()
    for sc, sh in ((scale.cuda(), shift.cuda()), (None, None), (None, shift.cuda())):
        want = pnn.maxpool3x3s2(pnn.stem7x7s2(xs, wp, sc, sh, relu=True))
        got = pnn.stem7x7s2_pool(xs, wp, sc, sh)
        assert got.shape == want.shape == (B, H // 4, W // 4, 64)
        assert torch.equal(got, want)
    xi = int_tensor((B, H, W, 3), g, -2, 2)
    wi = int_tensor((64, 3, 7, 7), g, -1, 1)
    ref = to.maxpool3x3s2(to.bf16r(torch.relu(to.conv2d(xi, wi, stride=2, pad=3))))
    xsi = torch.from_numpy(to.space_to_depth16(xi.numpy())).to(BF16).cuda()
    goti = pnn.stem7x7s2_pool(xsi, pnn.pack_stem_weight(wi).cuda())
    assert torch.equal(goti.float().cpu(), ref)


def test_maxpool():
    g = torch.Generator().manual_seed(13)
    x = rand_bf16((3, 16, 12, 64), g)
    ref = to.maxpool3x3s2(x)
    got = pnn.maxpool3x3s2(x.to(BF16).cuda())
    assert torch.equal(got.float().cpu(), ref)


# 64 output channels and at least 2048 pixel tiles of 128 select the 256-pixel x 64-channel
# workgroup (four waves along the pixels); smaller layers never reach it.  Integer operands:
# exact, so the comparison is bit for bit.  The last tile is part-filled (266 112 pixels).
@pytest.mark.parametrize("K,cin,bnstats", [(3, 64, False), (1, 128, False), (3, 64, True)])
def test_conv_256_pixel_tile_for_64_channel_layers(K, cin, bnstats):
    B, H, W, Cout, p = 22, 126, 96, 64, K // 2
    g = torch.Generator().manual_seed(77 + K + cin)
    x = int_tensor((B, H, W, cin), g, -2, 2)
    w = int_tensor((Cout, cin, K, K), g, -1, 1)
    xg, wg = x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda()
    if bnstats:
        ws = pnn.bn_workspace(B * H * W, Cout, "cuda:0")
        got, rows = pnn.conv2d_bnstats(xg, wg, ws, stride=1, pad=p)
        tile = 128 if os.environ.get("PE_IGEMM_T256", "4") == "0" else 256   # one partial row per tile
        assert rows == (B * H * W + tile - 1) // tile
        part = ws.view(torch.float32)[:rows * 2 * Cout].view(rows, 2, Cout).double().sum(0).cpu()
    else:
        got = pnn.conv2d(xg, wg, stride=1, pad=p)
    for b in (0, 9, B - 1):
        ref = to.bf16r(to.conv2d(x[b:b + 1], w, stride=1, pad=p))
        assert torch.equal(got[b:b + 1].float().cpu(), ref)
    if bnstats:
        z = got.double().cpu().reshape(-1, Cout)
        assert torch.allclose(part[0], z.sum(0), rtol=1e-6, atol=1e-3)
        assert torch.allclose(part[1], (z * z).sum(0), rtol=1e-6, atol=1e-3)


# pe_conv2d_fwd_chain_bf16: conv2 (3x3) -> conv3 (1x1 + residual + ReLU) -> next conv1 (1x1) of a
# bottleneck in one launch over the same pixel tiles.  Every stage's output must be, bit for
# bit, what the stage gives as a launch of its own; the first stage is also held against the
# oracle.  2 x 37 x 29 pixels: the last 256-pixel tile is part-filled.
@pytest.mark.parametrize("nstage,c_last", [(3, 64), (3, 128), (2, 0), (1, 0)])
def test_conv_chain_equals_separate_launches(nstage, c_last):
    B, H, W = 2, 37, 29
    g = torch.Generator().manual_seed(5 + nstage + c_last)
    x = rand_bf16((B, H, W, 64), g).to(BF16).cuda()
    chans = [(64, 64, 3), (64, 256, 1), (256, c_last, 1)][:nstage]
    stages, ws = [], []
    for (ci, co, k) in chans:
        wf = rand_bf16((co, ci, k, k), g, (2.0 / (ci * k * k)) ** 0.5)
        ws.append(wf)
        stages.append({"w": pnn.pack_conv_weight(wf).cuda(),
                       "scale": (torch.rand(co, generator=g) + 0.5).cuda(),
                       "shift": (torch.randn(co, generator=g) * 0.1).cuda(),
                       "pad": k // 2, "relu": True})
    if nstage >= 2:
        stages[1]["residual"] = rand_bf16((B, H, W, 256), g).to(BF16).cuda()
    outs = pnn.conv2d_chain(x, stages)
    cur = x
    for st, got in zip(stages, outs):
        want = pnn.conv2d(cur, st["w"], st["scale"], st["shift"], st.get("residual"), 1,
                          st["pad"], True)
        assert torch.equal(got, want)
        cur = want
    ref = to.conv2d(x.float().cpu(), ws[0], stages[0]["scale"].cpu(), stages[0]["shift"].cpu(),
                    None, 1, 1, relu=True)
    close(outs[0], ref)


def test_conv_chain_rejects_a_stage_that_is_not_pointwise():
    g = torch.Generator().manual_seed(9)
    x = rand_bf16((1, 8, 8, 64), g).to(BF16).cuda()
    w3 = pnn.pack_conv_weight(rand_bf16((64, 64, 3, 3), g)).cuda()
    with pytest.raises(RuntimeError):
        pnn.conv2d_chain(x, [{"w": w3, "pad": 1}, {"w": w3, "pad": 1}])


def test_conv_256_pixel_tile_with_full_epilogue():
    """The 256 x 64 workgroup with scale, shift, residual and ReLU (a MobileNetV2 projection at
    batch 256 has this shape class); random operands, the usual tolerance."""
    B, H, W, cin, Cout = 22, 126, 96, 128, 64
    g = torch.Generator().manual_seed(91)
    x = rand_bf16((B, H, W, cin), g)
    w = rand_bf16((Cout, cin, 1, 1), g, (2.0 / cin) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    res = rand_bf16((B, H, W, Cout), g)
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), scale.cuda(), shift.cuda(),
                     res.to(BF16).cuda(), 1, 0, relu=True)
    for b in (0, 13, B - 1):
        ref = to.conv2d(x[b:b + 1], w, scale, shift, res[b:b + 1], 1, 0, relu=True)
        close(got[b:b + 1], ref)
