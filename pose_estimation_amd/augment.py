"""GPU-side input pipeline: resize, scale jitter, random crop and left-right flip of uint8
frames, fused with normalisation into the bf16 space-to-depth image the stem reads, plus the
matching joint transform.

The reference configures this in its train_config (test.ipynb:995-1010): augmentation
{flip_left_right: True, random_crop: True, scale_range: [0.5, 2.0]} and preprocess
{image_resize: [320, 320]}; the code (dataset/pose_data.py) is absent, so the parameterisation
below is this build's: one affine map per image, drawn on the host, applied on the GPU.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ._ffi import check, lib
from .nn import BF16, IMAGENET_MEAN, IMAGENET_STD
from .ops import _need_gpu, _stream, on_device

# COCO order: nose, eyes, ears, shoulders, elbows, wrists, hips, knees, ankles
COCO_FLIP_PERM = (0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15)


def resize_params(B: int, src_hw: Tuple[int, int], out_hw: Tuple[int, int]) -> np.ndarray:
    """Plain resize of the whole frame: float32 [B,5] = (sx, sy, tx, ty, flip)."""
    p = np.zeros((B, 5), np.float32)
    p[:, 0] = src_hw[1] / out_hw[1]
    p[:, 1] = src_hw[0] / out_hw[0]
    return p


def random_params(B: int, src_hw: Tuple[int, int], out_hw: Tuple[int, int],
                  scale_range: Sequence[float] = (0.5, 2.0), flip_left_right: bool = True,
                  random_crop: bool = True, seed: int = 0) -> np.ndarray:
    """Per-image affine maps in the reference's configuration vocabulary: the frame is zoomed by
    z ~ U(scale_range) relative to a plain resize (z > 1 enlarges the subject, so the output
    window covers 1/z of the frame), the window is placed uniformly inside the frame when
    random_crop (centred otherwise; for z < 1 the frame floats inside the window), and the
    result is mirrored with probability 1/2 when flip_left_right."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Hs, Ws = src_hw
    H, W = out_hw
    z = rng.uniform(scale_range[0], scale_range[1], size=B)
    sx = (Ws / W) / z
    sy = (Hs / H) / z
    span_x, span_y = sx * W, sy * H              # extent of the window in source pixels
    fx = rng.uniform(size=B) if random_crop else np.full(B, 0.5)
    fy = rng.uniform(size=B) if random_crop else np.full(B, 0.5)
    tx = (Ws - span_x) * fx
    ty = (Hs - span_y) * fy
    flip = (rng.uniform(size=B) < 0.5) if flip_left_right else np.zeros(B, bool)
    return np.stack([sx, sy, tx, ty, flip.astype(np.float64)], axis=1).astype(np.float32)


def augment_u8(img: torch.Tensor, params: torch.Tensor, out_hw: Tuple[int, int],
               mean=IMAGENET_MEAN, std=IMAGENET_STD, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [B,Hs,Ws,3] + params float32 [B,5] (device) -> bf16 space-to-depth [B,H/2,W/2,16]."""
    _need_gpu(img, params, out)
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3 or not img.is_contiguous():
        raise ValueError("img must be contiguous uint8 [B,Hs,Ws,3]")
    B, Hs, Ws, _ = img.shape
    if params.dtype != torch.float32 or tuple(params.shape) != (B, 5) or not params.is_contiguous():
        raise ValueError("params must be contiguous float32 [B,5]")
    H, W = out_hw
    if out is None:
        out = torch.empty((B, H // 2, W // 2, 16), dtype=BF16, device=img.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    with on_device(img):
        check(lib().pe_augment_u8_s2d_bf16(img.data_ptr(), B, Hs, Ws, params.data_ptr(), H, W,
                                           ctypes.addressof(m), ctypes.addressof(s),
                                           out.data_ptr(), _stream(img)))
    return out


def transform_joints(joints: torch.Tensor, params: torch.Tensor, out_w: int,
                     flip_perm: Optional[Sequence[int]] = None) -> torch.Tensor:
    """joints float32 [B,K,3] in source pixels -> the same joints in output pixels."""
    _need_gpu(joints, params)
    if joints.dtype != torch.float32 or joints.dim() != 3 or joints.shape[2] != 3:
        raise ValueError("joints must be float32 [B,K,3]")
    joints = joints.contiguous()
    B, K, _ = joints.shape
    if params.dtype != torch.float32 or tuple(params.shape) != (B, 5) or not params.is_contiguous():
        raise ValueError("params must be contiguous float32 [B,5]")
    out = torch.empty_like(joints)
    perm = None
    if flip_perm is not None:
        if len(flip_perm) != K:
            raise ValueError("flip_perm must name one source joint per joint")
        perm = (ctypes.c_int32 * K)(*flip_perm)
    with on_device(joints):
        check(lib().pe_transform_joints_f32(joints.data_ptr(), B, K, params.data_ptr(), int(out_w),
                                            ctypes.addressof(perm) if perm is not None else None,
                                            out.data_ptr(), _stream(joints)))
    return out


def augment_mask(mask: torch.Tensor, params: torch.Tensor, src_stride: float,
                 out_hw: Tuple[int, int], stride: float, fill: int = 1) -> torch.Tensor:
    """Ignore mask uint8/bool [B,Hms,Wms] over the source frame (cells of src_stride pixels) ->
    uint8 [B,Hm,Wm] over the output image (cells of `stride` pixels) under the map of `params`;
    cells whose centre leaves the source frame get `fill` (1 = trained as background)."""
    _need_gpu(mask, params)
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if mask.dtype != torch.uint8 or mask.dim() != 3:
        raise ValueError("mask must be uint8 or bool [B,Hms,Wms]")
    mask = mask.contiguous()
    B, Hms, Wms = mask.shape
    if params.dtype != torch.float32 or tuple(params.shape) != (B, 5) or not params.is_contiguous():
        raise ValueError("params must be contiguous float32 [B,5]")
    Hm, Wm = out_hw
    out = torch.empty((B, Hm, Wm), dtype=torch.uint8, device=mask.device)
    with on_device(mask):
        check(lib().pe_augment_mask_u8(mask.data_ptr(), B, Hms, Wms, float(src_stride),
                                       params.data_ptr(), int(Hm), int(Wm), float(stride),
                                       int(fill), out.data_ptr(), _stream(mask)))
    return out
