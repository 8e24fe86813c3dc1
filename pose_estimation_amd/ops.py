"""Python wrappers over the C ABI.  torch tensors are used only as device buffers
and for the current HIP stream; all arithmetic happens in libpe_hip.so.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _ffi
from ._ffi import PE_DTYPE_BF16, PE_DTYPE_F32, check, lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(t: torch.Tensor) -> int:
    """Raw handle of torch's current stream on the tensor's device.  The training loop makes
    hundreds of launches per step, so this avoids building a torch.cuda.Stream object."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(t: torch.Tensor):
    """Context manager that makes the tensor's GPU current for a launch; free when it already
    is (the usual case: one process per GPU)."""
    idx = t.device.index
    if idx is None or (_raw_device is not None and idx == _raw_device()):
        return _NO_SWITCH
    return torch.cuda.device(idx)


def _need_gpu(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "pose_estimation_amd ops run only on GPU tensors (no CPU fallback); got a "
                f"{t.device} tensor"
            )


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------
# a1 decode
# ---------------------------------------------------------------------------
def decode(heatmaps: torch.Tensor, refine: bool = True, layout: str = "NCHW",
           out: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Heat-map peak decode -> (xy [B,K,2] f32, conf [B,K] f32, idx [B,K] i32).

    heatmaps: float32 [B,K,H,W] (layout="NCHW") or [B,H,W,K] (layout="NHWC", the
    reference's label layout, test.ipynb:931), any strides (no copy is made).
    out: optional preallocated contiguous (xy, conf, idx) to write into.
    """
    _need_gpu(heatmaps)
    if heatmaps.dtype != torch.float32 or heatmaps.dim() != 4:
        raise ValueError("decode expects a 4-D float32 tensor")
    if layout == "NCHW":
        B, K, H, W = heatmaps.shape
        sB, sK, sH, sW = heatmaps.stride()
    elif layout == "NHWC":
        B, H, W, K = heatmaps.shape
        sB, sH, sW, sK = heatmaps.stride()
    else:
        raise ValueError(f"unknown layout {layout!r}")
    dev = heatmaps.device
    if out is None:
        xy = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
        conf = torch.empty((B, K), dtype=torch.float32, device=dev)
        idx = torch.empty((B, K), dtype=torch.int32, device=dev)
    else:
        xy, conf, idx = out
        _need_gpu(xy, conf, idx)
        ok = (xy.dtype == torch.float32 and tuple(xy.shape) == (B, K, 2) and xy.is_contiguous()
              and conf.dtype == torch.float32 and tuple(conf.shape) == (B, K)
              and conf.is_contiguous() and idx.dtype == torch.int32
              and tuple(idx.shape) == (B, K) and idx.is_contiguous())
        if not ok:
            raise ValueError("out must be contiguous (f32 [B,K,2], f32 [B,K], i32 [B,K])")
    with torch.cuda.device(dev):
        check(lib().pe_decode_f32(heatmaps.data_ptr(), B, K, H, W, sB, sK, sH, sW,
                                  1 if refine else 0, xy.data_ptr(), conf.data_ptr(),
                                  idx.data_ptr(), _stream(heatmaps)))
    return xy, conf, idx


# ---------------------------------------------------------------------------
# a2 target render
# ---------------------------------------------------------------------------
def gaussian_table(sigma: float, radius: int) -> np.ndarray:
    """The host-computed Gaussian table the kernels use (float32, radius+1 entries)."""
    out = np.zeros(radius + 1, dtype=np.float32)
    check(lib().pe_gaussian_table_f32(float(sigma), int(radius), out.ctypes.data))
    return out


def render_targets(joints: torch.Tensor, H: int, W: int, stride: float, sigma: float,
                   radius: Optional[int] = None, layout: str = "NCHW",
                   out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gaussian targets from joints [B,K,3] (x, y, v) -> (target, weight [B,K]).

    target is [B,K,H,W] (layout="NCHW") or [B,H,W,K] (layout="NHWC", the reference's
    label layout).  radius defaults to 3*sigma; radius=1 gives the reference's 3x3 support.
    """
    _need_gpu(joints, out)
    if joints.dtype != torch.float32 or joints.dim() != 3 or joints.shape[2] != 3:
        raise ValueError("joints must be float32 [B,K,3]")
    joints = joints.contiguous()
    B, K, _ = joints.shape
    if radius is None:
        radius = int(round(3 * sigma))
    dev = joints.device
    if layout == "NCHW":
        shape = (B, K, H, W)
    elif layout == "NHWC":
        shape = (B, H, W, K)
    else:
        raise ValueError(f"unknown layout {layout!r}")
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    elif tuple(out.shape) != shape or out.dtype != torch.float32:
        raise ValueError(f"out must be float32 {shape}")
    if layout == "NCHW":
        tB, tK, tH, tW = out.stride()
    else:
        tB, tH, tW, tK = out.stride()
    weight = torch.empty((B, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().pe_render_targets_f32(joints.data_ptr(), B, K, H, W, float(stride),
                                          float(sigma), int(radius), out.data_ptr(),
                                          tB, tK, tH, tW, weight.data_ptr(), _stream(joints)))
    return out, weight


# ---------------------------------------------------------------------------
# vector maps (keypoint-pair offset fields): labels and loss
# ---------------------------------------------------------------------------
def _pairs_array(pairs) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
    if arr.shape[0] == 0:
        raise ValueError("pairs must hold at least one (k1, k2)")
    return arr


def render_vecmaps(joints: torch.Tensor, pairs, H: int, W: int, stride: float, radius: int = 1,
                   layout: str = "NCHW", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Vector-map labels from joints [B,K,3]: [B,4P,H,W] ("NCHW") or [B,H,W,4P] ("NHWC", the
    reference's layout).  Channel 4i, 4i+1 = (x, y) cell offset to the second joint of pair i
    inside the first joint's window; 4i+2, 4i+3 the reverse."""
    _need_gpu(joints, out)
    if joints.dtype != torch.float32 or joints.dim() != 3 or joints.shape[2] != 3:
        raise ValueError("joints must be float32 [B,K,3]")
    joints = joints.contiguous()
    B, K, _ = joints.shape
    pa = _pairs_array(pairs)
    C = 4 * pa.shape[0]
    if layout == "NCHW":
        shape = (B, C, H, W)
    elif layout == "NHWC":
        shape = (B, H, W, C)
    else:
        raise ValueError(f"unknown layout {layout!r}")
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=joints.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32:
        raise ValueError(f"out must be float32 {shape}")
    if layout == "NCHW":
        sB, sC, sH, sW = out.stride()
    else:
        sB, sH, sW, sC = out.stride()
    with on_device(joints):
        check(lib().pe_render_vecmaps_f32(joints.data_ptr(), B, K, pa.ctypes.data, pa.shape[0], H, W,
                                          float(stride), int(radius), out.data_ptr(), sB, sC, sH,
                                          sW, _stream(joints)))
    return out


def vecmap_loss_fwd_bwd(pred: torch.Tensor, joints: torch.Tensor, pairs, stride: float,
                        radius: int = 1, mask: Optional[torch.Tensor] = None,
                        need_grad: bool = True, workspace: Optional[torch.Tensor] = None):
    """Squared-error loss of a predicted vector map against the label rebuilt from the joints,
    over window cells only.  pred [B,4P,H,W] float32 or bfloat16 contiguous.
    Returns (loss [1], per_channel [4P], grad or None)."""
    _need_gpu(pred, joints, mask, workspace)
    if pred.dim() != 4 or not pred.is_contiguous():
        raise ValueError("pred must be a contiguous [B,4P,H,W] tensor")
    if pred.dtype == torch.float32:
        dt = PE_DTYPE_F32
    elif pred.dtype == torch.bfloat16:
        dt = PE_DTYPE_BF16
    else:
        raise ValueError("pred must be float32 or bfloat16")
    B, C, H, W = pred.shape
    pa = _pairs_array(pairs)
    if C != 4 * pa.shape[0]:
        raise ValueError(f"pred has {C} channels, expected {4 * pa.shape[0]}")
    if joints.dtype != torch.float32 or joints.dim() != 3 or joints.shape[0] != B or \
            joints.shape[2] != 3:
        raise ValueError("joints must be float32 [B,K,3]")
    joints = joints.contiguous()
    if mask is not None:
        if tuple(mask.shape) != (B, H, W):
            raise ValueError(f"mask must be [B,H,W] = {(B, H, W)}, got {tuple(mask.shape)}")
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.dtype != torch.uint8:
            raise ValueError("mask must be uint8 or bool")
        mask = mask.contiguous()
    dev = pred.device
    if workspace is None:
        workspace = torch.empty(lib().pe_vecmap_workspace_bytes(B, pa.shape[0]), dtype=torch.uint8,
                                device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    per_channel = torch.empty(C, dtype=torch.float32, device=dev)
    grad = torch.empty_like(pred) if need_grad else None
    with torch.cuda.device(dev):
        check(lib().pe_vecmap_loss_fwd_bwd(pred.data_ptr(), dt, joints.data_ptr(), _ptr(mask), B,
                                           joints.shape[1], pa.ctypes.data, pa.shape[0], H, W,
                                           float(stride), int(radius), loss.data_ptr(),
                                           per_channel.data_ptr(), _ptr(grad),
                                           workspace.data_ptr(), _stream(pred)))
    return loss, per_channel, grad


# ---------------------------------------------------------------------------
# a3 masked MSE loss + gradient
# ---------------------------------------------------------------------------
def mse_workspace(B: int, K: int, device) -> torch.Tensor:
    nbytes = lib().pe_mse_workspace_bytes(B, K)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def mse_fwd_bwd(pred: torch.Tensor, target: Optional[torch.Tensor] = None,
                weight: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, *,
                joints: Optional[torch.Tensor] = None, stride: float = 4.0, sigma: float = 2.0,
                radius: Optional[int] = None, need_grad: bool = True,
                workspace: Optional[torch.Tensor] = None):
    """Masked per-joint MSE.  Returns (loss [1] f32, per_joint [K] f32, grad or None, weight).

    pred [B,K,H,W] float32 or bfloat16, contiguous.  Give either `target` (+ optional
    `weight` [B,K]) or `joints` [B,K,3] (target rendered on the fly, never materialised;
    the returned weight is then computed by the kernel).  mask: [B,H,W] uint8/bool.
    """
    _need_gpu(pred, target, weight, mask, joints, workspace)
    if pred.dim() != 4 or not pred.is_contiguous():
        raise ValueError("pred must be a contiguous [B,K,H,W] tensor")
    if pred.dtype == torch.float32:
        dt = PE_DTYPE_F32
    elif pred.dtype == torch.bfloat16:
        dt = PE_DTYPE_BF16
    else:
        raise ValueError("pred must be float32 or bfloat16")
    if (target is None) == (joints is None):
        raise ValueError("give exactly one of target and joints")
    B, K, H, W = pred.shape
    dev = pred.device
    if target is not None:
        if target.dtype != torch.float32 or tuple(target.shape) != (B, K, H, W):
            raise ValueError("target must be float32 [B,K,H,W]")
        target = target.contiguous()
        if weight is not None:
            if weight.dtype != torch.float32 or tuple(weight.shape) != (B, K):
                raise ValueError("weight must be float32 [B,K]")
            weight = weight.contiguous()
    else:
        if joints.dtype != torch.float32 or tuple(joints.shape) != (B, K, 3):
            raise ValueError("joints must be float32 [B,K,3]")
        joints = joints.contiguous()
        weight = torch.empty((B, K), dtype=torch.float32, device=dev)
        if radius is None:
            radius = int(round(3 * sigma))
    if mask is not None:
        if tuple(mask.shape) != (B, H, W):
            raise ValueError(f"mask must be [B,H,W] = {(B, H, W)}, got {tuple(mask.shape)}")
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.dtype != torch.uint8:
            raise ValueError("mask must be uint8 or bool")
        mask = mask.contiguous()
    if workspace is None:
        workspace = mse_workspace(B, K, dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    per_joint = torch.empty(K, dtype=torch.float32, device=dev)
    grad = torch.empty_like(pred) if need_grad else None
    with torch.cuda.device(dev):
        check(lib().pe_mse_fwd_bwd(pred.data_ptr(), dt, _ptr(target), _ptr(joints), _ptr(weight),
                                   _ptr(mask), B, K, H, W, float(stride), float(sigma),
                                   int(radius or 0), loss.data_ptr(), per_joint.data_ptr(),
                                   _ptr(grad), workspace.data_ptr(), _stream(pred)))
    return loss, per_joint, grad, weight


# ---------------------------------------------------------------------------
# a6 optimizer steps
# ---------------------------------------------------------------------------
def adam_scalars(lr: float, beta1: float, beta2: float, step: int, tf1: bool = True):
    """(step_size, denom_scale) for pe_adam_step.  tf1=True is tf.train.AdamOptimizer's form
    (the reference's optimizer by name, test.ipynb:1007-1008)."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if tf1:
        return lr * math.sqrt(bc2) / bc1, 1.0
    return lr / bc1, 1.0 / math.sqrt(bc2)


def adam_step(p: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor,
              step: int, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8, p_bf16: Optional[torch.Tensor] = None, tf1: bool = True,
              grad_scale: float = 1.0) -> None:
    """In-place Adam over flat float32 arenas (p, m, v updated; p_bf16 receives the copy)."""
    _need_gpu(p, m, v, grad, p_bf16)
    for t in (p, m, v, grad):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adam arenas must be contiguous float32 of equal length")
    if p_bf16 is not None and (p_bf16.dtype != torch.bfloat16 or p_bf16.numel() != p.numel()
                               or not p_bf16.is_contiguous()):
        raise ValueError("p_bf16 must be contiguous bfloat16 of the same length")
    s, a = adam_scalars(lr, beta1, beta2, step, tf1)
    with on_device(p):
        check(lib().pe_adam_step(p.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(),
                                 _ptr(p_bf16), p.numel(), beta1, beta2, eps, s, a, grad_scale,
                                 _stream(p)))


def sgd_step(p: torch.Tensor, mom: torch.Tensor, grad: torch.Tensor, lr: float,
             momentum: float = 0.9, p_bf16: Optional[torch.Tensor] = None,
             grad_scale: float = 1.0) -> None:
    """In-place momentum SGD over flat float32 arenas."""
    _need_gpu(p, mom, grad, p_bf16)
    for t in (p, mom, grad):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("sgd arenas must be contiguous float32 of equal length")
    with on_device(p):
        check(lib().pe_sgd_step(p.data_ptr(), mom.data_ptr(), grad.data_ptr(), _ptr(p_bf16),
                                p.numel(), lr, momentum, grad_scale, _stream(p)))
