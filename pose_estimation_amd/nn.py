"""Layer-level wrappers over the conv / deconv C ABI plus the weight packers.

Activations are NHWC bfloat16 torch tensors (buffers only); packed weights are bfloat16 with
the reduction axis contiguous.  The packers take weights in the usual framework layouts
(conv: [Cout,Cin,KH,KW]; transposed conv: [Cin,Cout,KH,KW]) so a checkpoint of the reference
architecture can be loaded tensor by tensor.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import torch

from ._ffi import (PE_CONV_MAX_CHAIN, PE_WGRAD_MAX_BATCH, ConvDesc, ConvStage, WgradJob, check,
                   lib)
from .ops import _need_gpu, _ptr, _stream, on_device

BF16 = torch.bfloat16


# --------------------------------------------------------------------------- packers
def pack_conv_weight(w_oihw: torch.Tensor, dtype=BF16) -> torch.Tensor:
    """[Cout,Cin,KH,KW] -> [Cout,KH,KW,Cin] (bf16 unless another dtype is asked for)."""
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_conv_weight_narrow(w_oihw: torch.Tensor, dtype=BF16) -> torch.Tensor:
    """[Cout,Cin,KH,KW] with Cin of 8, 16 or 32 -> [Cout,KH,ceil(KW/P),P*Cin], P = 64/Cin:
    run c of a filter row holds taps kw = c*P .. c*P+P-1 pixel by pixel, zero past the filter."""
    cout, cin, kh, kw = w_oihw.shape
    if cin not in (8, 16, 32):
        raise ValueError("narrow packing is for 8, 16 or 32 input channels")
    P = 64 // cin
    kwc = (kw + P - 1) // P
    out = torch.zeros(cout, kh, kwc * P, cin, dtype=w_oihw.dtype, device=w_oihw.device)
    out[:, :, :kw, :] = w_oihw.permute(0, 2, 3, 1)
    return out.reshape(cout, kh, kwc, P * cin).contiguous().to(dtype)


def deconv_phase_taps(py: int, a: int):
    """Kernel index and input offset of tap `a` (0/1) of output parity `py` for a 4x4,
    stride-2, pad-1 transposed conv: out[2Y+py] += in[Y+d] * w[k]."""
    if py == 0:
        return (1, 0) if a == 0 else (3, -1)
    return (2, 0) if a == 0 else (0, 1)


def pack_deconv_weight(w_iohw: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d weight [Cin,Cout,4,4] -> bf16 [4 phases, Cout, 4 taps, Cin]
    (phase = 2*py+px, tap = 2*a+b)."""
    cin, cout, kh, kw = w_iohw.shape
    assert kh == 4 and kw == 4
    out = torch.empty(4, cout, 4, cin, dtype=w_iohw.dtype, device=w_iohw.device)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    k_y, _ = deconv_phase_taps(py, a)
                    k_x, _ = deconv_phase_taps(px, b)
                    out[2 * py + px, :, 2 * a + b, :] = w_iohw[:, :, k_y, k_x].t()
    return out.contiguous().to(BF16)


def pack_stem_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """7x7 stride-2 pad-3 stem weight [Cout,3,7,7] -> bf16 [Cout, 4 rows, 4 cols * 16] for the
    equivalent 4x4 stride-1 conv over the 2x2 space-to-depth image (channel = (ry*2+rx)*3+c)."""
    cout = w_oihw.shape[0]
    assert tuple(w_oihw.shape[1:]) == (3, 7, 7)
    out = torch.zeros(cout, 4, 4, 16, dtype=w_oihw.dtype, device=w_oihw.device)
    for dY in range(-2, 2):
        for ry in range(2):
            kh = 2 * dY + ry + 3
            if not 0 <= kh < 7:
                continue
            for dX in range(-2, 2):
                for rx in range(2):
                    kw = 2 * dX + rx + 3
                    if not 0 <= kw < 7:
                        continue
                    s = (ry * 2 + rx) * 3
                    out[:, dY + 2, dX + 2, s:s + 3] = w_oihw[:, :, kh, kw]
    return out.reshape(cout, 4, 64).contiguous().to(BF16)


# --------------------------------------------------------------------------- layers
def _check_act(x: torch.Tensor, name: str = "x"):
    _need_gpu(x)
    if x.dtype != BF16 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"{name} must be a contiguous NHWC bfloat16 tensor")


# 3x3 stride-1 pad-1 layers with 64 input and 64 output channels go through the resident-filter
# kernel (pe_conv3x3_c64_fwd_bf16, bit-identical to the engine) from 1024 16 x 16 tiles up.  One
# persistent workgroup per CU loads the 72 KB filter once and holds the CU's whole LDS.  Alone it
# beats the engine at every size tried (192 tiles: 8.1 against 12.6 us; 3072: 57 against 104 us),
# but a launch that owns every CU cannot run beside the other branches of a multi-branch network:
# with the threshold at 128 tiles the HRNet-style forward, whose 64-channel branch has 576, went
# from 5.65 to 6.03 ms.  PE_CONV_PATCH3X3=0: always the engine (comparison).
PATCH3X3 = os.environ.get("PE_CONV_PATCH3X3", "1") != "0"
# The 32-channel twin (pe_conv3x3_c32_fwd_bf16: 3x3 stride-1 pad-1, 32 -> 32 channels, residual
# allowed) needs 60 KB of LDS, so it shares a CU with other streams' launches.
# PE_CONV_PATCH3X3_C32=0: always the engine.
PATCH3X3_C32 = os.environ.get("PE_CONV_PATCH3X3_C32", "1") != "0"
PATCH3X3_C32_MIN_TILES = int(os.environ.get("PE_CONV_PATCH3X3_C32_MIN_TILES", "512"))
PATCH3X3_MIN_TILES = int(os.environ.get("PE_CONV_PATCH3X3_MIN_TILES", "1024"))


def conv2d(x: torch.Tensor, w: torch.Tensor, scale: Optional[torch.Tensor] = None,
           shift: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           stride: int = 1, pad: int = 0, relu: bool = False, out_f32_nchw: bool = False,
           out: Optional[torch.Tensor] = None, kw: Optional[int] = None) -> torch.Tensor:
    """y = relu?( conv(x, w) * scale + shift + residual ).  x [B,H,W,Cin], w [Cout,KH,KW,Cin].
    With 8, 16 or 32 input channels w is the pack_conv_weight_narrow layout and `kw` gives the
    filter width."""
    _check_act(x)
    _need_gpu(w, scale, shift, residual, out)
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    if Cin in (8, 16, 32):
        P = 64 // Cin
        if kw is None or Cin2 != 64 or KW != (kw + P - 1) // P:
            raise ValueError("narrow conv: w must be [Cout,KH,ceil(kw/P),64] and kw given")
        KW = kw
    elif Cin2 != Cin:
        raise ValueError("w must be contiguous bfloat16 [Cout,KH,KW,Cin]")
    if w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("w must be contiguous bfloat16 [Cout,KH,KW,Cin]")
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if (PATCH3X3 and Cin == 64 and Cout == 64 and KH == 3 and KW == 3 and stride == 1 and pad == 1
            and residual is None and not out_f32_nchw
            and B * ((H + 15) // 16) * ((W + 15) // 16) >= PATCH3X3_MIN_TILES):
        return conv3x3_c64(x, w, scale, shift, int(relu), out)
    if (PATCH3X3_C32 and Cin == 32 and Cout == 32 and KH == 3 and KW == 3 and stride == 1
            and pad == 1 and not out_f32_nchw
            and B * ((H + 15) // 16) * ((W + 15) // 16) >= PATCH3X3_C32_MIN_TILES):
        return conv3x3_c32(x, w, scale, shift, residual, int(relu), out)
    if out is None:
        if out_f32_nchw:
            out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((B, Ho, Wo, Cout), dtype=BF16, device=x.device)
    if residual is not None and (residual.shape != out.shape or residual.dtype != BF16):
        raise ValueError("residual must match the output")
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), int(out_f32_nchw))
    with on_device(x):
        check(lib().pe_conv2d_fwd_bf16(ctypes.addressof(d), x.data_ptr(), w.data_ptr(), _ptr(scale),
                                       _ptr(shift), _ptr(residual), out.data_ptr(), _stream(x)))
    return out


def conv2d_chain(x: torch.Tensor, stages) -> list:
    """Up to three convolutions over the same pixels in one launch (pe_conv2d_fwd_chain_bf16).

    stages: dicts with w [Cout,KH,KW,Cin] and optionally scale, shift, residual, stride, pad,
    relu, out; every stage after the first must be a 1x1 stride-1 conv over the previous
    stage's output.  Returns the list of outputs, each what conv2d would give for that stage."""
    _check_act(x)
    if not 1 <= len(stages) <= PE_CONV_MAX_CHAIN:
        raise ValueError(f"1 to {PE_CONV_MAX_CHAIN} stages")
    arr = (ConvStage * len(stages))()
    outs, keep = [], []
    B, H, W, Cin = x.shape
    for i, st in enumerate(stages):
        w = st["w"]
        scale, shift, residual = st.get("scale"), st.get("shift"), st.get("residual")
        stride, pad, relu = int(st.get("stride", 1)), int(st.get("pad", 0)), bool(st.get("relu", False))
        _need_gpu(w, scale, shift, residual, st.get("out"))
        Cout, KH, KW, Cin2 = w.shape
        if Cin2 != Cin or w.dtype != BF16 or not w.is_contiguous():
            raise ValueError(f"stage {i}: w must be contiguous bfloat16 [Cout,KH,KW,{Cin}]")
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        out = st.get("out")
        if out is None:
            out = torch.empty((B, Ho, Wo, Cout), dtype=BF16, device=x.device)
        elif tuple(out.shape) != (B, Ho, Wo, Cout) or out.dtype != BF16 or not out.is_contiguous():
            raise ValueError(f"stage {i}: out must be contiguous bfloat16 {(B, Ho, Wo, Cout)}")
        if residual is not None and (residual.shape != out.shape or residual.dtype != BF16):
            raise ValueError(f"stage {i}: residual must match the output")
        arr[i].desc = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), 0)
        arr[i].w, arr[i].scale, arr[i].shift = w.data_ptr(), _ptr(scale), _ptr(shift)
        arr[i].residual, arr[i].y = _ptr(residual), out.data_ptr()
        keep += [w, scale, shift, residual]
        outs.append(out)
        H, W, Cin = Ho, Wo, Cout
    with on_device(x):
        check(lib().pe_conv2d_fwd_chain_bf16(x.data_ptr(), ctypes.addressof(arr), len(stages),
                                             _stream(x)))
    return outs


def fuse_upsample_add(base: Optional[torch.Tensor], lows, log2_factors, relu: bool = True,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu?( base + sum_i nearest_upsample(lows[i], 2**log2_factors[i]) ), NHWC bf16, summed in
    float32.  base may be None when there is at least one low-resolution map."""
    lows = list(lows)
    if base is None and not lows:
        raise ValueError("nothing to add")
    if base is not None:
        _check_act(base, "base")
        B, H, W, C = base.shape
    else:
        _check_act(lows[0], "lows[0]")
        B, h, w, C = lows[0].shape
        H, W = h << log2_factors[0], w << log2_factors[0]
    for t, f in zip(lows, log2_factors):
        _check_act(t, "low")
        if tuple(t.shape) != (B, H >> f, W >> f, C):
            raise ValueError(f"low-resolution map {tuple(t.shape)} does not match {(B, H >> f, W >> f, C)}")
    if out is None:
        out = torch.empty((B, H, W, C), dtype=BF16, device=lows[0].device if base is None else base.device)
    n = len(lows)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in lows])
    facs = (ctypes.c_int32 * max(n, 1))(*[int(f) for f in log2_factors[:n]])
    with on_device(out):
        check(lib().pe_fuse_upsample_add_bf16(_ptr(base), n, ctypes.addressof(ptrs),
                                              ctypes.addressof(facs), B, H, W, C, int(relu),
                                              out.data_ptr(), _stream(out)))
    return out


def deconv4x4s2(x: torch.Tensor, w_phases: torch.Tensor, scale=None, shift=None,
                relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ConvTranspose2d(k=4, s=2, p=1): x [B,H,W,Cin] -> [B,2H,2W,Cout]."""
    _check_act(x)
    B, H, W, Cin = x.shape
    _, Cout, _, _ = w_phases.shape
    if tuple(w_phases.shape) != (4, Cout, 4, Cin) or w_phases.dtype != BF16:
        raise ValueError("w_phases must be bfloat16 [4,Cout,4,Cin]")
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_deconv4x4s2_fwd_bf16(B, H, W, Cin, Cout, int(relu), x.data_ptr(),
                                            w_phases.data_ptr(), _ptr(scale), _ptr(shift),
                                            out.data_ptr(), _stream(x)))
    return out


def deconv4x4s2_head(x: torch.Tensor, w_phases: torch.Tensor, scale, shift, head_w: torch.Tensor,
                     head_bias: Optional[torch.Tensor] = None, relu: bool = True,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """deconv4x4s2 (+ scale, shift, ReLU) followed by a 1x1 conv head_w [K,1,1,Cout] (+ bias) in
    one launch: x [B,H,W,Cin] -> float32 heat maps [B,K,2H,2W].  The deconv's output is never
    written; the result equals deconv4x4s2 followed by conv2d(out_f32_nchw=True) bit for bit."""
    _check_act(x)
    _need_gpu(w_phases, scale, shift, head_w, head_bias, out)
    B, H, W, Cin = x.shape
    _, Cout, _, _ = w_phases.shape
    if tuple(w_phases.shape) != (4, Cout, 4, Cin) or w_phases.dtype != BF16:
        raise ValueError("w_phases must be bfloat16 [4,Cout,4,Cin]")
    K = head_w.shape[0]
    if head_w.dtype != BF16 or head_w.numel() != K * Cout or not head_w.is_contiguous():
        raise ValueError("head_w must be contiguous bfloat16 [K,1,1,Cout]")
    if out is None:
        out = torch.empty((B, K, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    with on_device(x):
        check(lib().pe_deconv4x4s2_head_fwd_bf16(B, H, W, Cin, Cout, int(relu), x.data_ptr(),
                                                 w_phases.data_ptr(), _ptr(scale), _ptr(shift),
                                                 head_w.data_ptr(), _ptr(head_bias), K,
                                                 out.data_ptr(), _stream(x)))
    return out


def stem7x7s2(x_s2d: torch.Tensor, w: torch.Tensor, scale=None, shift=None, relu: bool = True,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """7x7/2 stem on the space-to-depth image: x_s2d [B,H/2,W/2,16] -> [B,H/2,W/2,Cout]."""
    _check_act(x_s2d, "x_s2d")
    B, H2, W2, C = x_s2d.shape
    Cout = w.shape[0]
    if C != 16 or tuple(w.shape) != (Cout, 4, 64) or w.dtype != BF16:
        raise ValueError("stem expects x_s2d [...,16] and w bfloat16 [Cout,4,64]")
    if out is None:
        out = torch.empty((B, H2, W2, Cout), dtype=BF16, device=x_s2d.device)
    with on_device(x_s2d):
        check(lib().pe_stem7x7s2_fwd_bf16(B, H2, W2, Cout, int(relu), x_s2d.data_ptr(),
                                          w.data_ptr(), _ptr(scale), _ptr(shift), out.data_ptr(),
                                          _stream(x_s2d)))
    return out


def conv3x3_c64(x: torch.Tensor, w: torch.Tensor, scale=None, shift=None, relu: int = 1,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 stride-1 pad-1 conv, 64 -> 64 channels, through the resident-filter kernel
    (pe_conv3x3_c64_fwd_bf16): equals conv2d(x, w, scale, shift, None, 1, 1, relu) bit for bit."""
    _check_act(x)
    _need_gpu(w, scale, shift, out)
    B, H, W, C = x.shape
    if C != 64 or tuple(w.shape) != (64, 3, 3, 64) or w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("conv3x3_c64 expects x [...,64] and w contiguous bfloat16 [64,3,3,64]")
    if out is None:
        out = torch.empty((B, H, W, 64), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_conv3x3_c64_fwd_bf16(B, H, W, x.data_ptr(), w.data_ptr(), _ptr(scale),
                                            _ptr(shift), int(relu), out.data_ptr(), _stream(x)))
    return out


# conv2 (3x3 64 -> 64, folded BN, ReLU) and conv3 (1x1 64 -> 256, folded BN, residual, ReLU) of a
# bottleneck in one launch (conv3x3_c64_pw), under the same tile-count rule as the 3x3 kernel.
# PE_CONV_C2C3=0: always two launches (comparison).
C2C3 = os.environ.get("PE_CONV_C2C3", "1") != "0"


def c2c3_fusable(B: int, H: int, W: int, w2: torch.Tensor, w3: torch.Tensor, scale2, shift2) -> bool:
    """Whether conv3x3_c64_pw covers this pair of layers on a [B,H,W,64] bf16 input."""
    return (C2C3 and tuple(w2.shape) == (64, 3, 3, 64) and w3.shape[0] == 256
            and w3.numel() == 256 * 64 and scale2 is not None and shift2 is not None
            and B * ((H + 15) // 16) * ((W + 15) // 16) >= PATCH3X3_MIN_TILES)


def conv3x3_c32(x: torch.Tensor, w_narrow: torch.Tensor, scale=None, shift=None, residual=None,
                relu: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 stride-1 pad-1 conv, 32 -> 32 channels (+ scale, shift, residual, ReLU), through the
    resident-filter kernel (pe_conv3x3_c32_fwd_bf16); w_narrow is pack_conv_weight_narrow's
    [32,3,2,64].  Equals conv2d(x, w_narrow, scale, shift, residual, 1, 1, relu, kw=3) bit for
    bit."""
    _check_act(x)
    _need_gpu(w_narrow, scale, shift, residual, out)
    B, H, W, C = x.shape
    if (C != 32 or tuple(w_narrow.shape) != (32, 3, 2, 64) or w_narrow.dtype != BF16
            or not w_narrow.is_contiguous()):
        raise ValueError("conv3x3_c32 expects x [...,32] and w contiguous bfloat16 [32,3,2,64]")
    if residual is not None and (tuple(residual.shape) != (B, H, W, 32) or residual.dtype != BF16
                                 or not residual.is_contiguous()):
        raise ValueError("residual must be contiguous bfloat16 [B,H,W,32]")
    if out is None:
        out = torch.empty((B, H, W, 32), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_conv3x3_c32_fwd_bf16(B, H, W, x.data_ptr(), w_narrow.data_ptr(), _ptr(scale),
                                            _ptr(shift), _ptr(residual), int(relu), out.data_ptr(),
                                            _stream(x)))
    return out


def conv3x3_c64_pw(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                   w3: torch.Tensor, scale3, shift3, residual: torch.Tensor, relu3: bool = True,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3_c64(x, w, scale, shift, relu=1) followed by the pointwise layer w3 [256,1,1,64]
    (+ scale3, shift3, residual, ReLU) in one launch (pe_conv3x3_c64_pw_fwd_bf16): the
    64-channel tensor between them is never written.  Equals
    conv2d(conv2d(x, w, scale, shift, None, 1, 1, True), w3, scale3, shift3, residual, relu=relu3)
    bit for bit."""
    _check_act(x)
    _need_gpu(w, scale, shift, w3, scale3, shift3, residual, out)
    B, H, W, C = x.shape
    if C != 64 or tuple(w.shape) != (64, 3, 3, 64) or w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("conv3x3_c64_pw expects x [...,64] and w contiguous bfloat16 [64,3,3,64]")
    N3 = w3.shape[0]
    if w3.numel() != N3 * 64 or w3.dtype != BF16 or not w3.is_contiguous():
        raise ValueError("w3 must be contiguous bfloat16 [Cout3,1,1,64]")
    if tuple(residual.shape) != (B, H, W, N3) or residual.dtype != BF16 or not residual.is_contiguous():
        raise ValueError("residual must be contiguous bfloat16 [B,H,W,Cout3]")
    if out is None:
        out = torch.empty((B, H, W, N3), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_conv3x3_c64_pw_fwd_bf16(B, H, W, x.data_ptr(), w.data_ptr(), _ptr(scale),
                                               _ptr(shift), 1, w3.data_ptr(), N3, _ptr(scale3),
                                               _ptr(shift3), residual.data_ptr(), int(relu3),
                                               out.data_ptr(), _stream(x)))
    return out


def conv3x3_c64_pw_next(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                        w3: torch.Tensor, scale3, shift3, residual: torch.Tensor,
                        w_next: torch.Tensor, scale_next, shift_next,
                        out: Optional[torch.Tensor] = None, out_next: Optional[torch.Tensor] = None):
    """conv3x3_c64_pw(..., relu3=True) and, in the same launch, the NEXT bottleneck's first
    pointwise layer w_next [64,1,1,256] (+ scale_next, shift_next, ReLU) on its result
    (pe_conv3x3_c64_pw_next_fwd_bf16).  Returns (y [B,H,W,256], y_next [B,H,W,64]); equals
    y = conv3x3_c64_pw(...), y_next = conv2d(y, w_next, scale_next, shift_next, relu=True) bit
    for bit."""
    _check_act(x)
    _need_gpu(w, scale, shift, w3, scale3, shift3, residual, w_next, scale_next, shift_next, out,
              out_next)
    B, H, W, C = x.shape
    if C != 64 or tuple(w.shape) != (64, 3, 3, 64) or w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("conv3x3_c64_pw_next expects x [...,64] and w contiguous bfloat16 [64,3,3,64]")
    N3 = w3.shape[0]
    if w3.numel() != N3 * 64 or w3.dtype != BF16 or not w3.is_contiguous():
        raise ValueError("w3 must be contiguous bfloat16 [Cout3,1,1,64]")
    N1 = w_next.shape[0]
    if w_next.numel() != N1 * N3 or w_next.dtype != BF16 or not w_next.is_contiguous():
        raise ValueError("w_next must be contiguous bfloat16 [Cout_next,1,1,Cout3]")
    if tuple(residual.shape) != (B, H, W, N3) or residual.dtype != BF16 or not residual.is_contiguous():
        raise ValueError("residual must be contiguous bfloat16 [B,H,W,Cout3]")
    if out is None:
        out = torch.empty((B, H, W, N3), dtype=BF16, device=x.device)
    if out_next is None:
        out_next = torch.empty((B, H, W, N1), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_conv3x3_c64_pw_next_fwd_bf16(
            B, H, W, x.data_ptr(), w.data_ptr(), _ptr(scale), _ptr(shift), 1, w3.data_ptr(), N3,
            _ptr(scale3), _ptr(shift3), residual.data_ptr(), out.data_ptr(), w_next.data_ptr(), N1,
            _ptr(scale_next), _ptr(shift_next), out_next.data_ptr(), _stream(x)))
    return out, out_next


def conv3x3_c64_pw_proj(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                        w3: torch.Tensor, scale3, shift3, x0: torch.Tensor, wd: torch.Tensor,
                        scaled, shiftd, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3_c64_pw whose residual is the pointwise projection of x0 [B,H,W,64] by wd
    [256,1,1,64] (+ scaled, shiftd, no ReLU), computed inside the same launch
    (pe_conv3x3_c64_pw_proj_fwd_bf16).  Equals
    conv3x3_c64_pw(x, ..., residual=conv2d(x0, wd, scaled, shiftd)) bit for bit."""
    _check_act(x)
    _check_act(x0, "x0")
    _need_gpu(w, scale, shift, w3, scale3, shift3, wd, scaled, shiftd, out)
    B, H, W, C = x.shape
    if C != 64 or tuple(w.shape) != (64, 3, 3, 64) or w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("conv3x3_c64_pw_proj expects x [...,64] and w contiguous bfloat16 [64,3,3,64]")
    N3 = w3.shape[0]
    for name, t in (("w3", w3), ("wd", wd)):
        if t.shape[0] != N3 or t.numel() != N3 * 64 or t.dtype != BF16 or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous bfloat16 [Cout3,1,1,64]")
    if tuple(x0.shape) != (B, H, W, 64):
        raise ValueError("x0 must be [B,H,W,64]")
    if out is None:
        out = torch.empty((B, H, W, N3), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_conv3x3_c64_pw_proj_fwd_bf16(
            B, H, W, x.data_ptr(), w.data_ptr(), _ptr(scale), _ptr(shift), 1, w3.data_ptr(), N3,
            _ptr(scale3), _ptr(shift3), x0.data_ptr(), wd.data_ptr(), _ptr(scaled), _ptr(shiftd),
            out.data_ptr(), _stream(x)))
    return out


def stem7x7s2_pool(x_s2d: torch.Tensor, w: torch.Tensor, scale=None, shift=None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stem7x7s2 (+ scale, shift, ReLU) followed by maxpool3x3s2 in one launch:
    x_s2d [B,H/2,W/2,16] -> [B,H/4,W/4,64].  The stem's own output is never written; the result
    equals maxpool3x3s2(stem7x7s2(x_s2d, ..., relu=True)) bit for bit."""
    _check_act(x_s2d, "x_s2d")
    _need_gpu(w, scale, shift, out)
    B, H2, W2, C = x_s2d.shape
    Cout = w.shape[0]
    if C != 16 or tuple(w.shape) != (Cout, 4, 64) or w.dtype != BF16:
        raise ValueError("stem expects x_s2d [...,16] and w bfloat16 [Cout,4,64]")
    if out is None:
        out = torch.empty((B, H2 // 2, W2 // 2, Cout), dtype=BF16, device=x_s2d.device)
    with on_device(x_s2d):
        check(lib().pe_stem7x7s2_pool_fwd_bf16(B, H2, W2, Cout, x_s2d.data_ptr(), w.data_ptr(),
                                               _ptr(scale), _ptr(shift), out.data_ptr(),
                                               _stream(x_s2d)))
    return out


def conv2d_bnstats(x: torch.Tensor, w: torch.Tensor, ws: torch.Tensor, stride: int = 1,
                   pad: int = 0, out: Optional[torch.Tensor] = None, kw: Optional[int] = None):
    """z = conv(x, w) with the per-channel sums of z and z*z left in `ws` as one partial row per
    GEMM tile; returns (z, rows) for bn_finalize.  Narrow inputs (8, 16, 32 channels): as conv2d."""
    _check_act(x)
    _need_gpu(w, ws, out)
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    if Cin in (8, 16, 32):
        P = 64 // Cin
        if kw is None or Cin2 != 64 or KW != (kw + P - 1) // P:
            raise ValueError("narrow conv: w must be [Cout,KH,ceil(kw/P),64] and kw given")
        KW = kw
    elif Cin2 != Cin:
        raise ValueError("w must be contiguous bfloat16 [Cout,KH,KW,Cin]")
    if w.dtype != BF16 or not w.is_contiguous():
        raise ValueError("w must be contiguous bfloat16 [Cout,KH,KW,Cin]")
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=BF16, device=x.device)
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0)
    rows = ctypes.c_int32(0)
    with on_device(x):
        check(lib().pe_conv2d_fwd_bnstats_bf16(ctypes.addressof(d), x.data_ptr(), w.data_ptr(),
                                               out.data_ptr(), ws.data_ptr(), ws.numel(),
                                               ctypes.byref(rows), _stream(x)))
    return out, rows.value


def deconv4x4s2_bnstats(x: torch.Tensor, w_phases: torch.Tensor, ws: torch.Tensor,
                        out: Optional[torch.Tensor] = None):
    """deconv4x4s2 without an epilogue, plus the statistics partials (see conv2d_bnstats)."""
    _check_act(x)
    _need_gpu(w_phases, ws, out)
    B, H, W, Cin = x.shape
    _, Cout, _, _ = w_phases.shape
    if tuple(w_phases.shape) != (4, Cout, 4, Cin) or w_phases.dtype != BF16:
        raise ValueError("w_phases must be bfloat16 [4,Cout,4,Cin]")
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=BF16, device=x.device)
    rows = ctypes.c_int32(0)
    with on_device(x):
        check(lib().pe_deconv4x4s2_fwd_bnstats_bf16(B, H, W, Cin, Cout, x.data_ptr(),
                                                    w_phases.data_ptr(), out.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), ctypes.byref(rows),
                                                    _stream(x)))
    return out, rows.value


def stem7x7s2_bnstats(x_s2d: torch.Tensor, w: torch.Tensor, ws: torch.Tensor,
                      out: Optional[torch.Tensor] = None):
    """stem7x7s2 without an epilogue, plus the statistics partials (see conv2d_bnstats)."""
    _check_act(x_s2d, "x_s2d")
    _need_gpu(w, ws, out)
    B, H2, W2, C = x_s2d.shape
    Cout = w.shape[0]
    if C != 16 or tuple(w.shape) != (Cout, 4, 64) or w.dtype != BF16:
        raise ValueError("stem expects x_s2d [...,16] and w bfloat16 [Cout,4,64]")
    if out is None:
        out = torch.empty((B, H2, W2, Cout), dtype=BF16, device=x_s2d.device)
    rows = ctypes.c_int32(0)
    with on_device(x_s2d):
        check(lib().pe_stem7x7s2_fwd_bnstats_bf16(B, H2, W2, Cout, x_s2d.data_ptr(), w.data_ptr(),
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  ctypes.byref(rows), _stream(x_s2d)))
    return out, rows.value


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def pack_stem3x3_weight(w_oihw: torch.Tensor, dtype=BF16) -> torch.Tensor:
    """3x3 stride-2 pad-1 stem weight [Cout,3,3,3] -> bf16 [Cout, 2 rows, 4 cols * 16] for the
    equivalent 2-row conv over the 2x2 space-to-depth image (channel = (ry*2+rx)*3+c).  Kernel
    row ky reads s2d row offset (ky+1)//2 - 1 at parity (ky+1) % 2, and likewise for columns."""
    cout = w_oihw.shape[0]
    out = torch.zeros(cout, 2, 4, 16, dtype=torch.float32)
    for ky in range(3):
        r, ry = (ky + 1) // 2, (ky + 1) % 2
        for kx in range(3):
            q, rx = (kx + 1) // 2, (kx + 1) % 2
            s = (ry * 2 + rx) * 3
            out[:, r, q, s:s + 3] = w_oihw[:, :, ky, kx]
    return out.reshape(cout, 2, 64).contiguous().to(dtype)


def stem3x3s2(x_s2d: torch.Tensor, w: torch.Tensor, scale=None, shift=None, relu: int = 2,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3/2 stem on the space-to-depth image: x_s2d [B,H/2,W/2,16] -> [B,H/2,W/2,Cout]."""
    _check_act(x_s2d, "x_s2d")
    B, H2, W2, c16 = x_s2d.shape
    if c16 != 16 or tuple(w.shape[1:]) != (2, 64) or w.dtype != BF16:
        raise ValueError("stem3x3s2 expects x_s2d [B,H/2,W/2,16] and w bf16 [Cout,2,64]")
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, H2, W2, Cout), dtype=BF16, device=x_s2d.device)
    with on_device(x_s2d):
        check(lib().pe_stem3x3s2_fwd_bf16(B, H2, W2, Cout, int(relu), x_s2d.data_ptr(),
                                          w.data_ptr(), _ptr(scale), _ptr(shift), out.data_ptr(),
                                          _stream(x_s2d)))
    return out


def stem3x3s2_bnstats(x_s2d: torch.Tensor, w: torch.Tensor, ws: torch.Tensor,
                      out: Optional[torch.Tensor] = None):
    """stem3x3s2 without an epilogue, plus the statistics partials (see conv2d_bnstats)."""
    _check_act(x_s2d, "x_s2d")
    _need_gpu(w, ws, out)
    B, H2, W2, c16 = x_s2d.shape
    if c16 != 16 or tuple(w.shape[1:]) != (2, 64) or w.dtype != BF16:
        raise ValueError("stem3x3s2 expects x_s2d [B,H/2,W/2,16] and w bf16 [Cout,2,64]")
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, H2, W2, Cout), dtype=BF16, device=x_s2d.device)
    rows = ctypes.c_int32(0)
    with on_device(x_s2d):
        check(lib().pe_stem3x3s2_fwd_bnstats_bf16(B, H2, W2, Cout, x_s2d.data_ptr(), w.data_ptr(),
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  ctypes.byref(rows), _stream(x_s2d)))
    return out, rows.value


def depthwise3x3(x: torch.Tensor, w9c: torch.Tensor, scale=None, shift=None, stride: int = 1,
                 dilation: int = 1, relu: int = 2, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Depthwise 3x3 conv with pad = dilation.  x [B,H,W,C] bf16, w9c [9,C] bf16."""
    _check_act(x)
    B, H, W, C = x.shape
    if w9c.dtype != BF16 or tuple(w9c.shape) != (9, C) or not w9c.is_contiguous():
        raise ValueError("depthwise weight must be contiguous bfloat16 [9, C]")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=BF16, device=x.device)
    with on_device(x):
        check(lib().pe_depthwise3x3_fwd_bf16(x.data_ptr(), B, H, W, C, int(stride), int(dilation),
                                             w9c.data_ptr(), _ptr(scale), _ptr(shift), int(relu),
                                             out.data_ptr(), _stream(x)))
    return out


def depthwise3x3_bwd_data(dy: torch.Tensor, w9c: torch.Tensor, in_hw, stride: int = 1,
                          dilation: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx [B,H,W,C] of depthwise3x3 from dy [B,Ho,Wo,C] and the forward weight [9,C]."""
    _check_act(dy, "dy")
    B, Ho, Wo, C = dy.shape
    H, W = in_hw
    if (Ho, Wo) != ((H - 1) // stride + 1, (W - 1) // stride + 1):
        raise ValueError("dy does not match the input size")
    if w9c.dtype != BF16 or tuple(w9c.shape) != (9, C) or not w9c.is_contiguous():
        raise ValueError("depthwise weight must be contiguous bfloat16 [9, C]")
    if out is None:
        out = torch.empty((B, H, W, C), dtype=BF16, device=dy.device)
    with on_device(dy):
        check(lib().pe_depthwise3x3_bwd_data_bf16(dy.data_ptr(), B, H, W, C, int(stride),
                                                  int(dilation), w9c.data_ptr(), out.data_ptr(),
                                                  _stream(dy)))
    return out


_dw_ws: dict = {}


def depthwise3x3_bwd_weight(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, stride: int = 1,
                            dilation: int = 1) -> None:
    """dw (float32 [9,C], pre-zeroed) += weight gradient of depthwise3x3."""
    _check_act(x)
    _check_act(dy, "dy")
    B, H, W, C = x.shape
    if dw.dtype != torch.float32 or tuple(dw.shape) != (9, C) or not dw.is_contiguous():
        raise ValueError("dw must be contiguous float32 [9, C]")
    need = lib().pe_depthwise3x3_bwd_weight_workspace_bytes(C)
    key = _dev_key(x.device)
    ws = _dw_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _dw_ws[key] = ws
    with on_device(x):
        check(lib().pe_depthwise3x3_bwd_weight_bf16(x.data_ptr(), dy.data_ptr(), B, H, W, C,
                                                    int(stride), int(dilation), dw.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), _stream(x)))


def preprocess_u8(img: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [B,H,W,3] -> normalised bf16 space-to-depth [B,H/2,W/2,16]."""
    _need_gpu(img)
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3 or not img.is_contiguous():
        raise ValueError("img must be contiguous uint8 [B,H,W,3]")
    B, H, W, _ = img.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, 16), dtype=BF16, device=img.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    with on_device(img):
        check(lib().pe_preprocess_u8_s2d_bf16(img.data_ptr(), B, H, W, ctypes.addressof(m),
                                              ctypes.addressof(s), out.data_ptr(), _stream(img)))
    return out


def maxpool3x3s2(x: torch.Tensor, out: Optional[torch.Tensor] = None, want_index: bool = False):
    """3x3/2 max pool.  With want_index also returns the uint8 first-maximum positions the
    backward pass reads: (y, idx)."""
    _check_act(x)
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, C), dtype=BF16, device=x.device)
    idx = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=x.device) if want_index \
        else None
    with on_device(x):
        check(lib().pe_maxpool3x3s2_fwd_bf16(x.data_ptr(), B, H, W, C, out.data_ptr(), _ptr(idx),
                                             _stream(x)))
    return (out, idx) if want_index else out


# --------------------------------------------------------------------------- backward
def transpose_weights(w: torch.Tensor, out: torch.Tensor, t0: int = 0) -> torch.Tensor:
    """out[c, t0 + t, a] = w[a, t, c]; w is any packed weight viewed as [A,T,C] bf16 and out a
    contiguous bf16 [C, slots, apitch] with apitch >= A (columns past A are left untouched)."""
    _need_gpu(w, out)
    A, T, C = w.shape
    C2, slots, apitch = out.shape
    if C2 != C or w.dtype != BF16 or out.dtype != BF16 or not (w.is_contiguous() and out.is_contiguous()):
        raise ValueError("transpose_weights: w [A,T,C], out [C,slots,apitch], contiguous bf16")
    with on_device(w):
        check(lib().pe_transpose_weights_bf16(w.data_ptr(), A, T, C, out.data_ptr(), slots, apitch,
                                              t0, _stream(w)))
    return out


class TransposePlan:
    """A fixed list of weight transposes (see transpose_weights) executed as one launch.
    The tensors must stay alive and in place for the life of the plan."""

    _DTYPE = np.dtype([("in", np.uint64), ("out", np.uint64), ("A", np.int32), ("T", np.int32),
                       ("C", np.int32), ("slots", np.int32), ("apitch", np.int32),
                       ("t0", np.int32), ("tile_begin", np.int32), ("src_T", np.int16),
                       ("src_t0", np.int16)])

    def __init__(self, device):
        self.device = device
        self._jobs = []
        self._table = None
        self._tiles = 0

    def add(self, w: torch.Tensor, out: torch.Tensor, t0: int = 0, src_t0: int = 0,
            ntaps: Optional[int] = None) -> None:
        """out[c, t0 + t, a] = w[a, src_t0 + t, c] for t < ntaps (default: every tap of w)."""
        _need_gpu(w, out)
        A, Tsrc, C = w.shape
        T = Tsrc - src_t0 if ntaps is None else ntaps
        if w.dtype != BF16 or out.dtype != BF16 or not w.is_contiguous() or not out.is_contiguous():
            raise ValueError("transpose operands must be contiguous bfloat16")
        if T < 1 or src_t0 < 0 or src_t0 + T > Tsrc or Tsrc > 32767:
            raise ValueError("transpose tap window out of range")
        if out.shape[0] != C or out.shape[1] < t0 + T or out.shape[2] < A:
            raise ValueError(f"transpose output {tuple(out.shape)} too small for {tuple(w.shape)}")
        self._jobs.append((w.data_ptr(), out.data_ptr(), A, T, C, out.shape[1], out.shape[2], t0,
                           Tsrc, src_t0))
        self._table = None

    def run(self) -> None:
        if not self._jobs:
            return
        if self._table is None:
            rec = np.zeros(len(self._jobs), dtype=self._DTYPE)
            tiles = 0
            for i, (src, dst, A, T, C, slots, apitch, t0, Tsrc, st0) in enumerate(self._jobs):
                rec[i] = (src, dst, A, T, C, slots, apitch, t0, tiles, Tsrc, st0)
                tiles += T * ((A + 63) // 64) * ((C + 63) // 64)
            self._tiles = tiles
            self._table = torch.from_numpy(rec.view(np.uint8)).to(self.device)
        with on_device(self):
            check(lib().pe_transpose_weights_batched_bf16(
                self._table.data_ptr(), len(self._jobs), self._tiles,
                torch.cuda.current_stream(self.device).cuda_stream))


class BnFoldPlan:
    """The batch-norm layers of a network whose folded constants (scale = gamma / sqrt(var + eps),
    shift = beta - mean * scale) are produced together by one launch (pe_bn_fold_f32).  The
    tensors must stay alive and in place for the life of the plan; run() reads their current
    values, so it is called again after the parameters or running statistics changed."""

    _DTYPE = np.dtype([("gamma", np.uint64), ("beta", np.uint64), ("running_mean", np.uint64),
                       ("running_var", np.uint64), ("scale", np.uint64), ("shift", np.uint64),
                       ("mean", np.uint64), ("rstd", np.uint64), ("C", np.int32),
                       ("reserved", np.int32)])

    def __init__(self, device):
        self.device = torch.device(device)
        self._jobs = []
        self._table = None

    def add(self, gamma, beta, running_mean, running_var, scale, shift, mean=None, rstd=None):
        ts = [gamma, beta, running_mean, running_var, scale, shift]
        opt = [t for t in (mean, rstd) if t is not None]
        _need_gpu(*ts, *opt)
        C = gamma.numel()
        for t in ts + opt:
            if t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous():
                raise ValueError("bn fold operands must be contiguous float32 of one length")
        self._jobs.append(tuple(t.data_ptr() for t in ts)
                          + (_ptr(mean) or 0, _ptr(rstd) or 0, C, 0))
        self._table = None

    def __len__(self) -> int:
        return len(self._jobs)

    def run(self, eps: float) -> None:
        if not self._jobs:
            return
        if self._table is None:
            rec = np.zeros(len(self._jobs), dtype=self._DTYPE)
            for i, job in enumerate(self._jobs):
                rec[i] = job
            self._table = torch.from_numpy(rec.view(np.uint8)).to(self.device)
        with torch.cuda.device(self.device):
            check(lib().pe_bn_fold_f32(self._table.data_ptr(), len(self._jobs), float(eps),
                                       torch.cuda.current_stream(self.device).cuda_stream))


_narrow_layouts: dict = {}


def narrow_bwd_layout(k: int, stride: int, pad: int, cout: int):
    """(slot_of_tap, sub_of_tap, nslots) of the backward-data weights of a k x k conv with 8, 16
    or 32 output channels (see pe_conv2d_bwd_data_narrow_layout)."""
    key = (k, stride, pad, cout)
    if key not in _narrow_layouts:
        slot = (ctypes.c_int32 * (k * k))()
        sub = (ctypes.c_int32 * (k * k))()
        n = ctypes.c_int32(0)
        check(lib().pe_conv2d_bwd_data_narrow_layout(k, k, stride, pad, cout,
                                                     ctypes.addressof(slot), ctypes.addressof(sub),
                                                     ctypes.byref(n)))
        _narrow_layouts[key] = (list(slot), list(sub), n.value)
    return _narrow_layouts[key]


def bwd_data_weight_shape(cin: int, cout: int, k: int, stride: int, pad: int):
    """Shape of the re-packed weight conv2d_bwd_data reads."""
    if cout in (8, 16, 32):
        return (cin, narrow_bwd_layout(k, stride, pad, cout)[2], 64)
    return (cin, k * k, cout)


def plan_bwd_data_weight(plan: "TransposePlan", w_packed: torch.Tensor, w_t: torch.Tensor,
                         cin: int, cout: int, k: int, stride: int, pad: int) -> None:
    """Adds to `plan` the jobs that re-pack the forward weight of a k x k conv (regular
    [Cout,k,k,Cin] or narrow [Cout,k,ceil(k/P),64] packing) into w_t (bwd_data_weight_shape,
    zero-initialised by the caller: sub-blocks no tap maps to must stay zero)."""
    if cin in (8, 16, 32):
        Pin = 64 // cin
        row = (k + Pin - 1) // Pin * Pin        # source taps per filter row, zero past k
        src = w_packed.view(cout, k * row, cin)
    else:
        row = k
        src = w_packed.view(cout, k * k, cin)
    if cout in (8, 16, 32):
        slot, sub, nslots = narrow_bwd_layout(k, stride, pad, cout)
        P = 64 // cout
        dst = w_t.view(cin, nslots * P, cout)
        for kh in range(k):
            for kw in range(k):
                plan.add(src, dst, t0=slot[kh * k + kw] * P + sub[kh * k + kw],
                         src_t0=kh * row + kw, ntaps=1)
    else:
        for kh in range(k):
            plan.add(src, w_t, t0=kh * k, src_t0=kh * row, ntaps=k)


def narrow_weight_mask(cout: int, cin: int, k: int) -> torch.Tensor:
    """float32 [Cout,k,ceil(k/P),64]: 1 where the narrow packing holds a parameter, 0 on the
    zero fill past the filter width."""
    return pack_conv_weight_narrow(torch.ones(cout, cin, k, k)).float()


def add_bf16(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + b for bf16 tensors of one shape (float32 sum, one rounding)."""
    _need_gpu(a, b, out)
    if a.shape != b.shape or a.dtype != BF16 or b.dtype != BF16 or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("add_bf16: contiguous bfloat16 tensors of one shape")
    if out is None:
        out = torch.empty_like(a)
    with on_device(a):
        check(lib().pe_add_bf16(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream(a)))
    return out


def fuse_upsample_add_bwd(dy: torch.Tensor, y: Optional[torch.Tensor], log2_factors,
                          want_base: bool = True):
    """Backward of fuse_upsample_add: returns (dbase, [dlow_i]).  y is the forward output when
    the forward applied a ReLU, else None (then dbase is dy itself)."""
    _check_act(dy, "dy")
    B, H, W, C = dy.shape
    if y is not None and (y.shape != dy.shape or y.dtype != BF16):
        raise ValueError("y must match dy")
    n = len(log2_factors)
    dlows = [torch.empty((B, H >> f, W >> f, C), dtype=BF16, device=dy.device) for f in log2_factors]
    dbase = torch.empty_like(dy) if (want_base and y is not None) else None
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in dlows])
    facs = (ctypes.c_int32 * max(n, 1))(*[int(f) for f in log2_factors])
    with on_device(dy):
        check(lib().pe_fuse_upsample_add_bwd_bf16(dy.data_ptr(), _ptr(y), n, ctypes.addressof(ptrs),
                                                  ctypes.addressof(facs), B, H, W, C, _ptr(dbase),
                                                  _stream(dy)))
    if want_base and y is None:
        dbase = dy
    return dbase, dlows


def conv2d_bwd_data(dy: torch.Tensor, w_t: torch.Tensor, in_hw, k: int, stride: int, pad: int,
                    residual: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx [B,H,W,Cin] of a k x k conv from dy [B,Ho,Wo,Cout] and w_t [Cin, k*k, Cout]."""
    _check_act(dy, "dy")
    B, Ho, Wo, Cout = dy.shape
    Cin = w_t.shape[0]
    H, W = in_hw
    if Cout in (8, 16, 32):
        nslots = narrow_bwd_layout(k, stride, pad, Cout)[2]
        if tuple(w_t.shape) != (Cin, nslots, 64) or w_t.dtype != BF16:
            raise ValueError(f"narrow w_t must be bfloat16 [Cin, {nslots}, 64]")
    elif tuple(w_t.shape) != (Cin, k * k, Cout) or w_t.dtype != BF16:
        raise ValueError("w_t must be bfloat16 [Cin, k*k, Cout]")
    if out is None:
        out = torch.empty((B, H, W, Cin), dtype=BF16, device=dy.device)
    d = ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad, 0, 0)
    with on_device(dy):
        check(lib().pe_conv2d_bwd_data_bf16(ctypes.addressof(d), dy.data_ptr(), w_t.data_ptr(),
                                            _ptr(residual), out.data_ptr(), _stream(dy)))
    return out


def conv2d_bwd_data_bnmask(dy: torch.Tensor, w_t: torch.Tensor, in_hw, k: int, stride: int,
                           pad: int, z: torch.Tensor, act_scale: torch.Tensor,
                           act_shift: torch.Tensor, ws: torch.Tensor, relu: int = 1,
                           out: Optional[torch.Tensor] = None):
    """conv2d_bwd_data whose result is the gradient of a = act(z * act_scale + act_shift): stores
    it masked by the activation and leaves the partial rows of batch-norm backward's two sums in
    `ws`; returns (g, rows) for bn_bwd_from_partials."""
    _check_act(dy, "dy")
    _check_act(z, "z")
    B, Ho, Wo, Cout = dy.shape
    Cin = w_t.shape[0]
    H, W = in_hw
    if tuple(z.shape) != (B, H, W, Cin):
        raise ValueError(f"z must be [B,H,W,Cin] = {(B, H, W, Cin)}")
    if Cout in (8, 16, 32):
        nslots = narrow_bwd_layout(k, stride, pad, Cout)[2]
        if tuple(w_t.shape) != (Cin, nslots, 64) or w_t.dtype != BF16:
            raise ValueError(f"narrow w_t must be bfloat16 [Cin, {nslots}, 64]")
    elif tuple(w_t.shape) != (Cin, k * k, Cout) or w_t.dtype != BF16:
        raise ValueError("w_t must be bfloat16 [Cin, k*k, Cout]")
    if out is None:
        out = torch.empty((B, H, W, Cin), dtype=BF16, device=dy.device)
    d = ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad, 0, 0)
    rows = ctypes.c_int32(0)
    with on_device(dy):
        check(lib().pe_conv2d_bwd_data_bnmask_bf16(
            ctypes.addressof(d), dy.data_ptr(), w_t.data_ptr(), z.data_ptr(), act_scale.data_ptr(),
            act_shift.data_ptr(), int(relu), out.data_ptr(), ws.data_ptr(), ws.numel(),
            ctypes.byref(rows), _stream(dy)))
    return out, rows.value


def conv2d_bwd_data_resmask(dy: torch.Tensor, w_t: torch.Tensor, in_hw, k: int, stride: int,
                            pad: int, residual: Optional[torch.Tensor], act: torch.Tensor,
                            z: torch.Tensor, ws: torch.Tensor,
                            out: Optional[torch.Tensor] = None):
    """conv2d_bwd_data (+ residual) whose result is the whole gradient of the residual-layer
    output `act` = relu(bn(z) + identity): stores it masked by act > 0 and leaves the partial rows
    of batch-norm backward's two sums in `ws`; returns (g, rows) for bn_bwd_from_partials."""
    _check_act(dy, "dy")
    _check_act(z, "z")
    _check_act(act, "act")
    B, Ho, Wo, Cout = dy.shape
    Cin = w_t.shape[0]
    H, W = in_hw
    for name, t in (("z", z), ("act", act), ("residual", residual)):
        if t is not None and tuple(t.shape) != (B, H, W, Cin):
            raise ValueError(f"{name} must be [B,H,W,Cin] = {(B, H, W, Cin)}")
    if residual is not None:
        _check_act(residual, "residual")
    if Cout in (8, 16, 32):
        nslots = narrow_bwd_layout(k, stride, pad, Cout)[2]
        if tuple(w_t.shape) != (Cin, nslots, 64) or w_t.dtype != BF16:
            raise ValueError(f"narrow w_t must be bfloat16 [Cin, {nslots}, 64]")
    elif tuple(w_t.shape) != (Cin, k * k, Cout) or w_t.dtype != BF16:
        raise ValueError("w_t must be bfloat16 [Cin, k*k, Cout]")
    if out is None:
        out = torch.empty((B, H, W, Cin), dtype=BF16, device=dy.device)
    d = ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad, 0, 0)
    rows = ctypes.c_int32(0)
    with on_device(dy):
        check(lib().pe_conv2d_bwd_data_resmask_bf16(
            ctypes.addressof(d), dy.data_ptr(), w_t.data_ptr(), _ptr(residual), act.data_ptr(),
            z.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), ctypes.byref(rows),
            _stream(dy)))
    return out, rows.value


def deconv4x4s2_bwd_data_bnmask(dy: torch.Tensor, w_t: torch.Tensor, z: torch.Tensor,
                                act_scale: torch.Tensor, act_shift: torch.Tensor,
                                ws: torch.Tensor, relu: int = 1,
                                out: Optional[torch.Tensor] = None):
    """deconv4x4s2_bwd_data with the fused activation mask and sums (see conv2d_bwd_data_bnmask)."""
    _check_act(dy, "dy")
    _check_act(z, "z")
    B, H2, W2, Cout = dy.shape
    Cin = w_t.shape[0]
    if tuple(z.shape) != (B, H2 // 2, W2 // 2, Cin):
        raise ValueError("z must be [B,H,W,Cin]")
    if out is None:
        out = torch.empty((B, H2 // 2, W2 // 2, Cin), dtype=BF16, device=dy.device)
    rows = ctypes.c_int32(0)
    with on_device(dy):
        check(lib().pe_deconv4x4s2_bwd_data_bnmask_bf16(
            B, H2 // 2, W2 // 2, Cin, Cout, dy.data_ptr(), w_t.data_ptr(), z.data_ptr(),
            act_scale.data_ptr(), act_shift.data_ptr(), int(relu), out.data_ptr(), ws.data_ptr(),
            ws.numel(), ctypes.byref(rows), _stream(dy)))
    return out, rows.value


def bn_bwd_from_partials(g: torch.Tensor, z: torch.Tensor, rows: int, gamma, mean, rstd, dgamma,
                         dbeta, ws: torch.Tensor, dz=None, frozen: bool = False,
                         sync=None) -> torch.Tensor:
    """Second half of the fused batch-norm backward: folds the `rows` partial rows a *_bnmask
    backward-data call left in `ws` and returns dz."""
    _check_act(g, "g")
    C = z.shape[3]
    M = z.numel() // C
    if dz is None:
        dz = torch.empty_like(z)
    with on_device(z):
        check(lib().pe_bn_bwd_from_partials_bf16(g.data_ptr(), z.data_ptr(), int(rows), M, C,
                                                 gamma.data_ptr(), mean.data_ptr(),
                                                 rstd.data_ptr(), dgamma.data_ptr(),
                                                 dbeta.data_ptr(), dz.data_ptr(), ws.data_ptr(),
                                                 int(bool(frozen)), _ptr(sync), _stream(z)))
    return dz


def deconv4x4s2_bwd_data(dy: torch.Tensor, w_t: torch.Tensor,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx [B,H,W,Cin] from dy [B,2H,2W,Cout] and w_t [Cin,16,Cout]."""
    _check_act(dy, "dy")
    B, H2, W2, Cout = dy.shape
    Cin = w_t.shape[0]
    if out is None:
        out = torch.empty((B, H2 // 2, W2 // 2, Cin), dtype=BF16, device=dy.device)
    with on_device(dy):
        check(lib().pe_deconv4x4s2_bwd_data_bf16(B, H2 // 2, W2 // 2, Cin, Cout, dy.data_ptr(),
                                                 w_t.data_ptr(), out.data_ptr(), _stream(dy)))
    return out


_wgrad_ws: dict = {}


def _dev_key(device) -> str:
    """One key per physical device: "cuda" means the current device."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def wgrad_workspace(weight_elems: int, device) -> torch.Tensor:
    """Scratch for the split partial sums of the backward-weight GEMM (cached per device,
    grown on demand; calls on one stream reuse it safely)."""
    need = lib().pe_wgrad_workspace_bytes(int(weight_elems))
    key = _dev_key(device)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


def wgrad_batched_workspace(total_weight_elems: int, device) -> torch.Tensor:
    """Scratch for a batched weight-gradient launch (the same cached buffer as the single
    launches use, grown on demand)."""
    need = lib().pe_wgrad_batched_workspace_bytes(int(total_weight_elems))
    key = _dev_key(device)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


def live_scratch(device) -> list:
    """The module-level scratch buffers of `device` as they are now.  A grown buffer replaces
    its predecessor, which is then freed; whoever has stored a buffer's ADDRESS (a recorded
    launch tape) must hold the tensor itself, and gets it from here."""
    key = _dev_key(device)
    return [d[key] for d in (_wgrad_ws, _dw_ws) if key in d]


def _wgrad_elems(Cin: int, cout: int, k: int) -> int:
    return cout * k * ((k + 64 // Cin - 1) // (64 // Cin)) * 64 if Cin in (8, 16, 32) \
        else cout * k * k * Cin


def conv2d_bwd_weight_batched(jobs) -> None:
    """jobs: sequence of (x, dy, dw, k, stride, pad, cout or None), each what one
    conv2d_bwd_weight call takes.  The weight gradients of up to PE_WGRAD_MAX_BATCH layers go
    out as ONE launch plus one fold launch (longer lists are cut into such batches)."""
    jobs = list(jobs)
    if os.environ.get("PE_WGRAD_BATCHED", "1") == "0":      # comparison switch: one launch each
        for x, dy, dw, k, stride, pad, cout in jobs:
            conv2d_bwd_weight(x, dy, dw, k, stride, pad, cout)
        return
    for at in range(0, len(jobs), PE_WGRAD_MAX_BATCH):
        part = jobs[at:at + PE_WGRAD_MAX_BATCH]
        arr = (WgradJob * len(part))()
        total = 0
        for j, (x, dy, dw, k, stride, pad, cout) in enumerate(part):
            _check_act(x)
            _check_act(dy, "dy")
            B, H, W, Cin = x.shape
            ldy = dy.shape[3]
            cout = ldy if cout is None else cout
            if dw.dtype != torch.float32 or dw.numel() != _wgrad_elems(Cin, cout, k) \
                    or not dw.is_contiguous():
                raise ValueError("dw must be contiguous float32 in the forward weight's packed layout")
            arr[j].desc = ConvDesc(B, H, W, Cin, cout, k, k, stride, pad, 0, 0)
            arr[j].x, arr[j].dy, arr[j].dw, arr[j].ldy = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ldy
            total += dw.numel()
        x0 = part[0][0]
        with on_device(x0):
            ws = wgrad_batched_workspace(total, x0.device)
            check(lib().pe_conv2d_bwd_weight_batched_bf16(ctypes.byref(arr), len(part), ws.data_ptr(),
                                                          ws.numel(), _stream(x0)))


def conv2d_bwd_weight(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, k: int, stride: int,
                      pad: int, cout: Optional[int] = None) -> None:
    """dw (float32 [Cout,k,k,Cin], pre-zeroed) += weight gradient.  dy may be zero-padded to more
    channels than `cout`."""
    _check_act(x)
    _check_act(dy, "dy")
    B, H, W, Cin = x.shape
    ldy = dy.shape[3]
    cout = ldy if cout is None else cout
    if dw.dtype != torch.float32 or dw.numel() != _wgrad_elems(Cin, cout, k) or not dw.is_contiguous():
        raise ValueError("dw must be contiguous float32 in the forward weight's packed layout")
    d = ConvDesc(B, H, W, Cin, cout, k, k, stride, pad, 0, 0)
    with on_device(x):
        ws = wgrad_workspace(dw.numel(), x.device)
        check(lib().pe_conv2d_bwd_weight_bf16(ctypes.addressof(d), x.data_ptr(), dy.data_ptr(), ldy,
                                              dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _stream(x)))


def deconv4x4s2_bwd_weight(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor) -> None:
    _check_act(x)
    _check_act(dy, "dy")
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    if dw.dtype != torch.float32 or dw.numel() != 16 * Cout * Cin:
        raise ValueError("dw must be float32 [4,Cout,4,Cin]")
    with on_device(x):
        ws = wgrad_batched_workspace(dw.numel(), x.device)   # four phases, one batched launch
        check(lib().pe_deconv4x4s2_bwd_weight_bf16(B, H, W, Cin, Cout, x.data_ptr(), dy.data_ptr(),
                                                   dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   _stream(x)))


def stem7x7s2_bwd_weight(x_s2d: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor) -> None:
    _check_act(x_s2d, "x_s2d")
    _check_act(dy, "dy")
    B, H2, W2, _ = x_s2d.shape
    Cout = dy.shape[3]
    with on_device(x_s2d):
        ws = wgrad_workspace(dw.numel(), x_s2d.device)
        check(lib().pe_stem7x7s2_bwd_weight_bf16(B, H2, W2, Cout, x_s2d.data_ptr(), dy.data_ptr(),
                                                 dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _stream(x_s2d)))


def stem3x3s2_bwd_weight(x_s2d: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor) -> None:
    """dw (float32 [Cout,2,64], pre-zeroed) += weight gradient of the 3x3/2 stem; entries of the
    packing that hold no parameter must be masked off by the caller."""
    _check_act(x_s2d, "x_s2d")
    _check_act(dy, "dy")
    B, H2, W2, _ = x_s2d.shape
    Cout = dy.shape[3]
    if dw.dtype != torch.float32 or dw.numel() != Cout * 128:
        raise ValueError("dw must be float32 [Cout,2,64]")
    with on_device(x_s2d):
        ws = wgrad_workspace(dw.numel(), x_s2d.device)
        check(lib().pe_stem3x3s2_bwd_weight_bf16(B, H2, W2, Cout, x_s2d.data_ptr(), dy.data_ptr(),
                                                 dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _stream(x_s2d)))


def maxpool3x3s2_bwd(idx: torch.Tensor, dy: torch.Tensor,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx [B,2Ho,2Wo,C] from the forward pass's idx and dy [B,Ho,Wo,C]."""
    _check_act(dy, "dy")
    _need_gpu(idx)
    if idx.dtype != torch.uint8 or tuple(idx.shape) != tuple(dy.shape) or not idx.is_contiguous():
        raise ValueError("idx must be a contiguous uint8 tensor shaped like dy")
    B, Ho, Wo, C = dy.shape
    if out is None:
        out = torch.empty((B, 2 * Ho, 2 * Wo, C), dtype=BF16, device=dy.device)
    with on_device(dy):
        check(lib().pe_maxpool3x3s2_bwd_bf16(idx.data_ptr(), dy.data_ptr(), B, 2 * Ho, 2 * Wo, C,
                                             out.data_ptr(), _stream(dy)))
    return out


def final_grad_prepare(g_nchw: torch.Tensor, kpad: int, dbias: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [B,K,H,W] -> bf16 [B,H,W,kpad] zero padded; dbias[K] (pre-zeroed) += pixel sums."""
    _need_gpu(g_nchw, dbias)
    B, K, H, W = g_nchw.shape
    if out is None:
        out = torch.empty((B, H, W, kpad), dtype=BF16, device=g_nchw.device)
    with on_device(g_nchw):
        check(lib().pe_final_grad_prepare_bf16(g_nchw.data_ptr(), B, K, H * W, kpad, out.data_ptr(),
                                               _ptr(dbias), _stream(g_nchw)))
    return out


def mul_(a: torch.Tensor, b: torch.Tensor) -> None:
    _need_gpu(a, b)
    with on_device(a):
        check(lib().pe_mul_f32(a.data_ptr(), b.data_ptr(), a.numel(), _stream(a)))


# --------------------------------------------------------------------------- batch norm
def bn_workspace(M: int, C: int, device) -> torch.Tensor:
    return torch.empty(lib().pe_bn_workspace_bytes(M, C), dtype=torch.uint8, device=device)


def bn_stats(z: torch.Tensor, gamma, beta, eps: float, momentum: float, running_mean,
             running_var, ws: torch.Tensor, out=None):
    """Batch statistics of z [B,H,W,C] -> (mean, rstd, scale, shift), each float32 [C]."""
    _check_act(z, "z")
    C = z.shape[3]
    M = z.numel() // C
    if out is None:
        out = torch.empty((4, C), dtype=torch.float32, device=z.device)
    with on_device(z):
        check(lib().pe_bn_stats_bf16(z.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), eps,
                                     momentum, _ptr(running_mean), _ptr(running_var),
                                     out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                     out[3].data_ptr(), ws.data_ptr(), _stream(z)))
    return out[0], out[1], out[2], out[3]


def bn_finalize(ws: torch.Tensor, rows: int, M: int, C: int, gamma, beta, eps: float,
                momentum: float, running_mean, running_var, out=None):
    """Second half of bn_stats for partial sums a *_bnstats conv left in `ws`.  `out` may be a
    float32 [4,C] tensor or a tuple of its four rows (mean, rstd, scale, shift)."""
    _need_gpu(ws, gamma, beta, running_mean, running_var)
    if out is None:
        out = torch.empty((4, C), dtype=torch.float32, device=ws.device)
    r = out if isinstance(out, tuple) else out.unbind(0)
    with on_device(ws):
        check(lib().pe_bn_finalize_bf16(ws.data_ptr(), int(rows), M, C, gamma.data_ptr(),
                                        beta.data_ptr(), eps, momentum, _ptr(running_mean),
                                        _ptr(running_var), r[0].data_ptr(), r[1].data_ptr(),
                                        r[2].data_ptr(), r[3].data_ptr(), _stream(ws)))
    return r[0], r[1], r[2], r[3]


def bn_sync(device) -> torch.Tensor:
    """The hand-off buffer of the one-launch batch-norm calls (an arrival counter that only
    grows; the library tracks its value).  One per stream on which such calls are issued."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    n = lib().pe_bn_sync_bytes()
    buf = torch.empty(n + 128, dtype=torch.uint8, device=dev)
    off = (-buf.data_ptr()) % 128
    sync = buf[off:off + n].view(torch.int32)
    with torch.cuda.device(dev):
        check(lib().pe_bn_sync_init(sync.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream))
    return sync


def bn_finalize_apply(ws: torch.Tensor, rows: int, z: torch.Tensor, gamma, beta, eps: float,
                      momentum: float, running_mean, running_var, out, sync: torch.Tensor,
                      residual=None, relu: int = 1, y: Optional[torch.Tensor] = None):
    """bn_finalize + bn_apply in one launch; `out` = the layer's (mean, rstd, scale, shift) rows.
    Returns y."""
    _check_act(z, "z")
    C = z.shape[3]
    M = z.numel() // C
    r = out if isinstance(out, tuple) else out.unbind(0)
    if y is None:
        y = torch.empty_like(z)
    with on_device(z):
        check(lib().pe_bn_finalize_apply_bf16(
            ws.data_ptr(), int(rows), z.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), eps,
            momentum, _ptr(running_mean), _ptr(running_var), r[0].data_ptr(), r[1].data_ptr(),
            r[2].data_ptr(), r[3].data_ptr(), _ptr(residual), int(relu), y.data_ptr(),
            sync.data_ptr(), _stream(z)))
    return y


def bn_stats_apply(z: torch.Tensor, gamma, beta, eps: float, momentum: float, running_mean,
                   running_var, ws: torch.Tensor, out, sync: torch.Tensor, residual=None,
                   relu: int = 1, y: Optional[torch.Tensor] = None):
    """bn_stats + bn_apply in two launches instead of three.  Returns y."""
    _check_act(z, "z")
    C = z.shape[3]
    M = z.numel() // C
    r = out if isinstance(out, tuple) else out.unbind(0)
    if y is None:
        y = torch.empty_like(z)
    with on_device(z):
        check(lib().pe_bn_stats_apply_bf16(
            z.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
            _ptr(running_mean), _ptr(running_var), r[0].data_ptr(), r[1].data_ptr(),
            r[2].data_ptr(), r[3].data_ptr(), _ptr(residual), int(relu), y.data_ptr(),
            ws.data_ptr(), sync.data_ptr(), _stream(z)))
    return y


def bn_apply(z: torch.Tensor, scale, shift, residual=None, relu: int = 1,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _check_act(z, "z")
    C = z.shape[3]
    M = z.numel() // C
    if out is None:
        out = torch.empty_like(z)
    with on_device(z):
        check(lib().pe_bn_apply_bf16(z.data_ptr(), M, C, scale.data_ptr(), shift.data_ptr(),
                                     _ptr(residual), int(relu), out.data_ptr(), _stream(z)))
    return out


def bn_bwd(da: torch.Tensor, a: Optional[torch.Tensor], z: torch.Tensor, gamma, mean, rstd,
           dgamma, dbeta, ws: torch.Tensor, want_masked: bool = False, dz=None, relu: int = 1,
           act_scale=None, act_shift=None, frozen: bool = False, sync=None):
    """Returns (dz, g) where g = da masked by the activation that produced `a` (relu 1: ReLU,
    2: ReLU6; no mask when a is None); g is None unless want_masked.  For a layer without a
    residual, a = None with act_scale / act_shift (what its bn_apply used) rebuilds the mask from
    z instead of reading the stored activation.  frozen: mean / rstd are running statistics."""
    _check_act(da, "da")
    C = z.shape[3]
    M = z.numel() // C
    if dz is None:
        dz = torch.empty_like(z)
    g = torch.empty_like(z) if want_masked else None
    with on_device(z):
        check(lib().pe_bn_bwd_bf16(da.data_ptr(), _ptr(a), int(relu), z.data_ptr(), M, C,
                                   gamma.data_ptr(),
                                   mean.data_ptr(), rstd.data_ptr(), _ptr(act_scale),
                                   _ptr(act_shift), dgamma.data_ptr(),
                                   dbeta.data_ptr(), dz.data_ptr(), _ptr(g), ws.data_ptr(),
                                   int(bool(frozen)), _ptr(sync), _stream(z)))
    return dz, g
