"""torch-CPU float32 restatement of the conv / deconv layers of the pose network.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference's network files
(nets/mobilenet_v2.py, model/mobilenet_pose.py, model/base_model.py;
PoseEstimation.pyproj:30-32) are absent and its arithmetic lived in TensorFlow 1.x
kernels, so these functions restate the public layer definitions with
torch.nn.functional on the CPU.  Inputs are float32 tensors holding bf16-representable
values (the kernels' operands are bf16); accumulation is float32; the result is returned
in float32 *before* the final bf16 rounding so tests can apply it explicitly.
All tensors here are NHWC to match the kernels; weights are in framework layouts.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def bf16r(t: torch.Tensor) -> torch.Tensor:
    """Round a float32 tensor to the nearest bf16 value (kept as float32)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def conv2d(x_nhwc, w_oihw, scale=None, shift=None, residual=None, stride=1, pad=0, relu=False):
    y = F.conv2d(_nchw(x_nhwc), w_oihw, None, stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = _nhwc(y)
    if residual is not None:
        y = y + residual
    if relu:
        y = torch.relu(y)
    return y


def deconv4x4s2(x_nhwc, w_iohw, scale=None, shift=None, relu=False):
    y = F.conv_transpose2d(_nchw(x_nhwc), w_iohw, None, stride=2, padding=1)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = _nhwc(y)
    if relu:
        y = torch.relu(y)
    return y


def preprocess_u8(img_u8: np.ndarray, mean, std) -> np.ndarray:
    """uint8 [B,H,W,3] -> float32 (x/255 - mean) * (1/std), evaluated in float32 in that
    order, then rounded to bf16 (returned as float32 NHWC [B,H,W,3])."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)
    inv = np.float32(1.0) / np.asarray(std, np.float32)
    y = ((x - m) * inv).astype(np.float32)
    return bf16r(torch.from_numpy(y)).numpy()


def space_to_depth16(x_nhwc3: np.ndarray) -> np.ndarray:
    """[B,H,W,3] -> [B,H/2,W/2,16] with channel (ry*2+rx)*3+c and four zero channels."""
    B, H, W, C = x_nhwc3.shape
    out = np.zeros((B, H // 2, W // 2, 16), x_nhwc3.dtype)
    for ry in range(2):
        for rx in range(2):
            s = (ry * 2 + rx) * 3
            out[..., s:s + 3] = x_nhwc3[:, ry::2, rx::2, :]
    return out


def maxpool3x3s2(x_nhwc):
    return _nhwc(F.max_pool2d(_nchw(x_nhwc), 3, 2, 1))
