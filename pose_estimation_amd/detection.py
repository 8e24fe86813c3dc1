"""Detection branch labels on the GPU: anchor boxes, per-anchor class / weight / box targets
and the objectness + box loss.

The reference makes these labels in utils/bboxes.py (absent).  Its notebook shows the layout:
1500 = 20*20*3 + 10*10*3 anchors per image, flattened level by level as [rows, cols, 3]
(test.ipynb:423-464), labels `classes`, `weights`, `regs` (test.ipynb:504-507), one positive
anchor per person (test.ipynb:538-541) and corner-box `regs` rows (test.ipynb:571-572).  Anchor
scales and aspect ratios are not visible there; the defaults below are this build's.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ._ffi import check, lib
from .ops import _need_gpu, _stream, on_device

DEFAULT_LEVELS = ((20, 20), (10, 10))
DEFAULT_SCALES = (0.35, 0.75)
DEFAULT_RATIOS = (0.5, 1.0, 2.0)


def make_anchors(levels: Sequence[Tuple[int, int]] = DEFAULT_LEVELS,
                 scales: Sequence[float] = DEFAULT_SCALES,
                 ratios: Sequence[float] = DEFAULT_RATIOS) -> np.ndarray:
    """float32 [A,4] corner boxes (ymin, xmin, ymax, xmax) in normalised coordinates, ordered
    level, row, column, ratio.  Height = scale / sqrt(ratio), width = scale * sqrt(ratio)."""
    if len(levels) != len(scales):
        raise ValueError("one scale per feature level")
    f = np.float32
    sq = np.sqrt(np.asarray(ratios, f))
    parts = []
    for (fh, fw), s in zip(levels, scales):
        cy = (np.arange(fh, dtype=f) + f(0.5)) / f(fh)
        cx = (np.arange(fw, dtype=f) + f(0.5)) / f(fw)
        hh = (f(s) / sq) * f(0.5)
        hw = (f(s) * sq) * f(0.5)
        cy = cy[:, None, None]
        cx = cx[None, :, None]
        box = np.stack(np.broadcast_arrays(cy - hh, cx - hw, cy + hh, cx + hw), axis=-1)
        parts.append(box.reshape(-1, 4))
    return np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=f)


def anchor_assign(anchors: torch.Tensor, boxes: torch.Tensor, box_classes: torch.Tensor,
                  pos_iou: float = 2.0, ignore_iou: float = 0.5):
    """anchors [A,4] f32, boxes [B,G,4] f32, box_classes [B,G] int32 (<= 0: empty slot), all on
    the GPU -> (classes [B,A], weights [B,A], regs [B,A,4], match [B,A] int32).

    The default pos_iou (> 1) keeps only each box's best anchor positive, which is what the
    reference's labels show; pass e.g. 0.5 for SSD-style threshold matching on top of that."""
    _need_gpu(anchors, boxes, box_classes)
    if anchors.dtype != torch.float32 or anchors.dim() != 2 or anchors.shape[1] != 4 \
            or not anchors.is_contiguous():
        raise ValueError("anchors must be contiguous float32 [A,4]")
    if boxes.dtype != torch.float32 or boxes.dim() != 3 or boxes.shape[2] != 4 \
            or not boxes.is_contiguous():
        raise ValueError("boxes must be contiguous float32 [B,G,4]")
    B, G, _ = boxes.shape
    if box_classes.dtype != torch.int32 or tuple(box_classes.shape) != (B, G) \
            or not box_classes.is_contiguous():
        raise ValueError("box_classes must be contiguous int32 [B,G]")
    A = anchors.shape[0]
    dev = anchors.device
    classes = torch.empty((B, A), dtype=torch.float32, device=dev)
    weights = torch.empty((B, A), dtype=torch.float32, device=dev)
    regs = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    match = torch.empty((B, A), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().pe_anchor_assign_f32(anchors.data_ptr(), A, boxes.data_ptr(),
                                         box_classes.data_ptr(), B, G, float(pos_iou),
                                         float(ignore_iou), classes.data_ptr(), weights.data_ptr(),
                                         regs.data_ptr(), match.data_ptr(), _stream(anchors)))
    return classes, weights, regs, match


def box_loss_fwd_bwd(logit: torch.Tensor, reg_pred: torch.Tensor, classes: torch.Tensor,
                     weights: torch.Tensor, regs: torch.Tensor, reg_weight: float = 1.0,
                     want_grad: bool = True):
    """Returns (loss4, grad_logit, grad_reg): loss4 = device float32 [total, cls, reg, positives]."""
    _need_gpu(logit, reg_pred, classes, weights, regs)
    B, A = logit.shape
    for t, shape, name in ((logit, (B, A), "logit"), (classes, (B, A), "classes"),
                           (weights, (B, A), "weights"), (reg_pred, (B, A, 4), "reg_pred"),
                           (regs, (B, A, 4), "regs")):
        if t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous float32 {shape}")
    dev = logit.device
    loss4 = torch.empty(4, dtype=torch.float32, device=dev)
    g_logit = torch.empty_like(logit) if want_grad else None
    g_reg = torch.empty_like(reg_pred) if want_grad else None
    ws = torch.empty(lib().pe_box_loss_workspace_bytes(B), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().pe_box_loss_fwd_bwd(logit.data_ptr(), reg_pred.data_ptr(), classes.data_ptr(),
                                        weights.data_ptr(), regs.data_ptr(), B, A,
                                        float(reg_weight), loss4.data_ptr(),
                                        g_logit.data_ptr() if want_grad else None,
                                        g_reg.data_ptr() if want_grad else None, ws.data_ptr(),
                                        _stream(logit)))
    return loss4, g_logit, g_reg


def boxes_from_joints(joints: torch.Tensor, H: int, W: int, margin: float = 0.05):
    """One person box per image from its labelled keypoints, for batches that carry keypoints
    but no box annotation (synthetic data): the extent of the joints with v > 0, grown by
    `margin` of the image on every side and clipped to it (pe_boxes_from_joints_f32).
    joints [B,K,3] float32 (x, y in input pixels, v) -> (boxes [B,1,4] float32 normalised
    (ymin, xmin, ymax, xmax), box_classes [B,1] int32: 1, or 0 when no joint is labelled)."""
    _need_gpu(joints)
    if joints.dtype != torch.float32 or joints.dim() != 3 or joints.shape[2] != 3 \
            or not joints.is_contiguous():
        raise ValueError("joints must be contiguous float32 [B,K,3]")
    B, K, _ = joints.shape
    boxes = torch.empty((B, 1, 4), dtype=torch.float32, device=joints.device)
    cls = torch.empty((B, 1), dtype=torch.int32, device=joints.device)
    with torch.cuda.device(joints.device):
        check(lib().pe_boxes_from_joints_f32(joints.data_ptr(), B, K, int(H), int(W),
                                             float(margin), boxes.data_ptr(), cls.data_ptr(),
                                             _stream(joints)))
    return boxes, cls
