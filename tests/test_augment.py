"""Input pipeline (resize / scale jitter / crop / flip + normalise, and the joint transform).

CPU: the NumPy restatement against independent statements (identity map = plain preprocessing,
integer shifts and mirrors as array slicing, a marker pixel tracked through the joint map).
GPU: pe_augment_u8_s2d_bf16 / pe_transform_joints_f32 / pe_augment_mask_u8 bit for bit against
the restatement.
The reference pins only the configuration (test.ipynb:995-1010), not results: parity unpinned.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle as to
from pose_estimation_amd import augment as aug
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import synthetic

MEAN, STD = pnn.IMAGENET_MEAN, pnn.IMAGENET_STD
F32 = np.float32


def test_identity_map_is_plain_preprocessing():
    img = synthetic.images_u8(2, 16, 12, seed=3)
    p = aug.resize_params(2, (16, 12), (16, 12))
    assert np.array_equal(oracle.augment_u8(img, p, (16, 12), MEAN, STD),
                          to.preprocess_u8(img, MEAN, STD))


def test_integer_shift_crop_and_flip_are_slices():
    img = synthetic.images_u8(1, 20, 24, seed=4)
    ref = to.preprocess_u8(img, MEAN, STD)
    p = np.array([[1.0, 1.0, 3.0, 5.0, 0.0]], F32)          # window at (x 3, y 5), 8 x 6
    got = oracle.augment_u8(img, p, (6, 8), MEAN, STD)
    assert np.array_equal(got, ref[:, 5:11, 3:11])
    p[0, 4] = 1.0
    assert np.array_equal(oracle.augment_u8(img, p, (6, 8), MEAN, STD), ref[:, 5:11, 3:11][:, :, ::-1])
    # outside the frame is black (0 before normalisation)
    p = np.array([[1.0, 1.0, -2.0, -1.0, 0.0]], F32)
    got = oracle.augment_u8(img, p, (4, 4), MEAN, STD)
    black = to.preprocess_u8(np.zeros((1, 1, 1, 3), np.uint8), MEAN, STD)[0, 0, 0]
    assert np.array_equal(got[0, 0, 0], black) and np.array_equal(got[0, 1, 2], ref[0, 0, 0])


def test_downscale_by_two_averages_four_pixels():
    img = synthetic.images_u8(1, 8, 8, seed=5)
    p = aug.resize_params(1, (8, 8), (4, 4))                 # sx = sy = 2: samples at pixel corners
    got = oracle.augment_u8(img, p, (4, 4), MEAN, STD)
    blk = img[0].astype(np.float64).reshape(4, 2, 4, 2, 3).mean(axis=(1, 3))
    want = (blk / 255.0 - np.asarray(MEAN)) / np.asarray(STD)
    assert np.allclose(got[0], want, atol=2e-2)              # one bf16 rounding of values up to ~2.6


def test_joint_map_follows_the_image_map():
    H, W, Hs, Ws = 32, 32, 48, 64
    prm = aug.random_params(6, (Hs, Ws), (H, W), seed=7)
    rng = np.random.default_rng(1)
    for b in range(6):
        # a joint sitting exactly on a source pixel lands where that pixel's value goes
        ox, oy = rng.integers(4, W - 4), rng.integers(4, H - 4)
        sx, sy, tx, ty, flip = prm[b]
        xf = W - 1 - ox if flip else ox
        u, v = (xf + 0.5) * sx + tx - 0.5, (oy + 0.5) * sy + ty - 0.5
        j = np.array([[[u, v, 2]]], F32)
        out = oracle.transform_joints(j, prm[b:b + 1], W)
        assert out[0, 0, 0] == pytest.approx(ox, abs=1e-3) and out[0, 0, 1] == pytest.approx(oy, abs=1e-3)
    # flipping exchanges left and right joints
    j = synthetic.joints(1, 17, Hs, Ws, seed=2)
    p = np.array([[2.0, 1.5, 0.0, 0.0, 1.0]], F32)
    out = oracle.transform_joints(j, p, W, aug.COCO_FLIP_PERM)
    assert out[0, 5, 2] == j[0, 6, 2] and out[0, 0, 2] == j[0, 0, 2]
    assert out[0, 5, 0] == pytest.approx(W - 1 - ((j[0, 6, 0] + 0.5) / 2.0 - 0.5), abs=1e-4)


def test_mask_map_follows_the_image_map():
    # identity map at equal cell sizes copies the mask; a mirror reverses its columns
    m = synthetic.ignore_mask(2, 8, 12, seed=3, p_ignore=0.4)
    ident = aug.resize_params(2, (32, 48), (32, 48))
    assert np.array_equal(oracle.augment_mask(m, ident, 4.0, (8, 12), 4.0), m)
    flip = ident.copy()
    flip[:, 4] = 1.0
    assert np.array_equal(oracle.augment_mask(m, flip, 4.0, (8, 12), 4.0), m[:, :, ::-1])
    # a crop shifted by two cells reads two cells to the right; what leaves the frame is `fill`
    shift = ident.copy()
    shift[:, 2] = 8.0
    out = oracle.augment_mask(m, shift, 4.0, (8, 12), 4.0, fill=7)
    assert np.array_equal(out[:, :, :10], m[:, :, 2:]) and np.all(out[:, :, 10:] == 7)
    # the mask lands where the image lands: a 2x downscale of a frame whose ignored region is the
    # left half keeps the left half ignored
    half = np.ones((1, 16, 16), np.uint8)
    half[:, :, :8] = 0
    down = aug.resize_params(1, (64, 64), (32, 32))
    out = oracle.augment_mask(half, down, 4.0, (8, 8), 4.0)
    assert np.all(out[:, :, :4] == 0) and np.all(out[:, :, 4:] == 1)


def test_random_params_respect_the_configured_ranges():
    prm = aug.random_params(2000, (480, 640), (320, 320), (0.5, 2.0), True, True, seed=3)
    z = (640 / 320) / prm[:, 0]
    assert z.min() >= 0.5 - 1e-5 and z.max() <= 2.0 + 1e-5 and 0.4 < prm[:, 4].mean() < 0.6
    assert np.allclose((480 / 320) / prm[:, 1], z, rtol=1e-5)
    zoomed = z > 1.0                                           # window lies inside the frame
    assert np.all(prm[zoomed, 2] >= -1e-3) and np.all(prm[zoomed, 2] + prm[zoomed, 0] * 320 <= 640 + 1e-2)
    fixed = aug.random_params(4, (480, 640), (320, 320), (1.0, 1.0), False, False, seed=3)
    assert np.allclose(fixed, aug.resize_params(4, (480, 640), (320, 320)), atol=1e-4)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(3, 48, 64, 32, 32), (2, 120, 160, 64, 48), (2, 37, 53, 320, 320),
                                 (1, 480, 640, 256, 192)])
def test_hip_augment_matches_oracle_bit_for_bit(cfg):
    B, Hs, Ws, H, W = cfg
    img = synthetic.images_u8(B, Hs, Ws, seed=9)
    prm = aug.random_params(B, (Hs, Ws), (H, W), seed=11)
    prm[0] = aug.resize_params(1, (Hs, Ws), (H, W))[0]
    got = aug.augment_u8(torch.from_numpy(img).cuda(), torch.from_numpy(prm).cuda(), (H, W))
    want = to.space_to_depth16(oracle.augment_u8(img, prm, (H, W), MEAN, STD))
    assert got.shape == (B, H // 2, W // 2, 16)
    assert np.array_equal(got.float().cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(4, 480, 640, 8, 256, 192, 4), (3, 96, 64, 4, 64, 96, 4),
                                 (2, 37, 53, 1, 320, 320, 8), (5, 64, 64, 16, 32, 32, 1)])
def test_hip_mask_warp_matches_oracle_bit_for_bit(cfg):
    B, Hs, Ws, ss, H, W, st = cfg
    Hms, Wms = -(-Hs // ss), -(-Ws // ss)
    m = synthetic.ignore_mask(B, Hms, Wms, seed=B, p_ignore=0.3)
    prm = aug.random_params(B, (Hs, Ws), (H, W), (0.5, 2.0), True, True, seed=11)
    for fill in (1, 0, 255):
        got = aug.augment_mask(torch.from_numpy(m).cuda(), torch.from_numpy(prm).cuda(), float(ss),
                               (H // st, W // st), float(st), fill=fill)
        want = oracle.augment_mask(m, prm, float(ss), (H // st, W // st), float(st), fill=fill)
        assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="strides"):
        aug.augment_mask(torch.from_numpy(m).cuda(), torch.from_numpy(prm).cuda(), 0.0,
                         (H // st, W // st), float(st))


@pytest.mark.gpu
def test_hip_identity_augment_equals_preprocess_kernel():
    img = torch.from_numpy(synthetic.images_u8(2, 64, 48, seed=1)).cuda()
    p = torch.from_numpy(aug.resize_params(2, (64, 48), (64, 48))).cuda()
    assert torch.equal(aug.augment_u8(img, p, (64, 48)), pnn.preprocess_u8(img))


@pytest.mark.gpu
def test_hip_joint_transform_matches_oracle():
    B, K = 16, 17
    j = synthetic.joints(B, K, 480, 640, seed=2)
    prm = aug.random_params(B, (480, 640), (256, 192), seed=5)
    got = aug.transform_joints(torch.from_numpy(j).cuda(), torch.from_numpy(prm).cuda(), 192,
                               aug.COCO_FLIP_PERM)
    assert np.array_equal(got.cpu().numpy(), oracle.transform_joints(j, prm, 192, aug.COCO_FLIP_PERM))
    got = aug.transform_joints(torch.from_numpy(j).cuda(), torch.from_numpy(prm).cuda(), 192)
    assert np.array_equal(got.cpu().numpy(), oracle.transform_joints(j, prm, 192))
