"""ctypes binding of libpe_hip.so (the C ABI declared in include/pe_api.h).

There is no CPU fallback: if the shared library is missing, loading fails with
an instruction to build it, and every op raises if its tensors are not on a GPU.
"""
from __future__ import annotations

import ctypes
from ctypes import c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpe_hip.so"

PE_OK = 0
PE_E_INVALID = -1
PE_E_HIP = -2
PE_E_UNSUPPORTED = -3
PE_DTYPE_F32 = 0
PE_DTYPE_BF16 = 1
PE_MAX_RADIUS = 31
PE_MSE_STATIC_MAX_MAPS = 65536


class PeError(RuntimeError):
    """A libpe_hip.so entry point returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libpe_hip status {code}: {message}")
        self.code = code
        self.message = message


# name -> (restype, argtypes); must list every symbol include/pe_api.h declares.
SIGNATURES = {
    "pe_version": (c_int32, []),
    "pe_last_error": (c_char_p, []),
    "pe_device_count": (c_int32, []),
    "pe_tape_begin": (c_int32, []),
    "pe_tape_length": (c_int64, []),
    "pe_tape_end": (c_int32, [c_void_p]),
    "pe_tape_replay": (c_int32, [c_void_p, c_int64, c_int64]),
    "pe_tape_destroy": (c_int32, [c_void_p]),
    "pe_stream_wait_stream": (c_int32, [c_void_p, c_void_p]),
    "pe_decode_f32": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64,
         c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_gaussian_table_f32": (c_int32, [c_float, c_int32, c_void_p]),
    "pe_render_targets_f32": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int32,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p],
    ),
    "pe_render_vecmaps_f32": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_float, c_int32,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p],
    ),
    "pe_vecmap_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "pe_vecmap_loss_fwd_bwd": (
        c_int32,
        [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32,
         c_int32, c_float, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_mse_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "pe_mse_fwd_bwd": (
        c_int32,
        [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int32,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_mse_fwd_bwd_static": (
        c_int32,
        [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int32,
         c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_adam_step": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
         c_float, c_float, c_float, c_float, c_float, c_float, c_void_p],
    ),
    "pe_sgd_step": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_void_p],
    ),
    "pe_conv2d_fwd_bf16": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_conv2d_fwd_chain_bf16": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "pe_deconv4x4s2_fwd_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p],
    ),
    "pe_stem7x7s2_fwd_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "pe_stem3x3s2_fwd_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "pe_depthwise3x3_fwd_bf16": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
         c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "pe_preprocess_u8_s2d_bf16": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_augment_u8_s2d_bf16": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "pe_augment_mask_u8": (
        c_int32, [c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_int32,
                  c_float, c_int32, c_void_p, c_void_p]),
    "pe_transform_joints_f32": (
        c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "pe_maxpool3x3s2_fwd_bf16": (
        c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "pe_maxpool3x3s2_bwd_bf16": (
        c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p],
    ),
    "pe_conv2d_bwd_data_bf16": (
        c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_deconv4x4s2_bwd_data_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_wgrad_workspace_bytes": (c_size_t, [c_int64]),
    "pe_conv2d_bwd_weight_bf16": (
        c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "pe_wgrad_batched_workspace_bytes": (c_size_t, [c_int64]),
    "pe_conv2d_bwd_weight_batched_bf16": (c_int32, [c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "pe_deconv4x4s2_bwd_weight_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_size_t, c_void_p],
    ),
    "pe_stem7x7s2_bwd_weight_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
         c_void_p],
    ),
    "pe_transpose_weights_bf16": (
        c_int32,
        [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p],
    ),
    "pe_transpose_weights_batched_bf16": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "pe_final_grad_prepare_bf16": (
        c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "pe_mul_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pe_stem3x3s2_bwd_weight_bf16": (
        c_int32,
        [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
         c_void_p],
    ),
    "pe_stem3x3s2_fwd_bnstats_bf16": (
        c_int32, [c_int32] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                  c_void_p]),
    "pe_conv2d_bwd_data_narrow_layout": (
        c_int32, [c_int32] * 5 + [c_void_p, c_void_p, c_void_p]),
    "pe_fuse_upsample_add_bwd_bf16": (
        c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                  c_int32, c_void_p, c_void_p]),
    "pe_add_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pe_comm_unique_id": (c_int32, [c_void_p]),
    "pe_comm_create": (c_int32, [c_int32, c_int32, c_void_p, c_void_p]),
    "pe_allreduce_bucket": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "pe_comm_destroy": (c_int32, [c_void_p]),
    "pe_grad_pack_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pe_grad_unpack_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pe_depthwise3x3_bwd_data_bf16": (
        c_int32, [c_void_p] + [c_int32] * 6 + [c_void_p, c_void_p, c_void_p]),
    "pe_depthwise3x3_bwd_weight_workspace_bytes": (c_size_t, [c_int32]),
    "pe_depthwise3x3_bwd_weight_bf16": (
        c_int32, [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "pe_fuse_upsample_add_bf16": (
        c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                  c_int32, c_void_p, c_void_p]),
    "pe_anchor_assign_f32": (
        c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float,
                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pe_conv3x3_c64_fwd_bf16": (
        c_int32, [c_int32] * 3 + [c_void_p] * 4 + [c_int32, c_void_p, c_void_p]),
    "pe_conv3x3_c32_fwd_bf16": (
        c_int32, [c_int32] * 3 + [c_void_p] * 5 + [c_int32, c_void_p, c_void_p]),
    "pe_conv3x3_c64_pw_fwd_bf16": (
        c_int32, [c_int32] * 3 + [c_void_p] * 4 + [c_int32, c_void_p, c_int32] + [c_void_p] * 3
        + [c_int32, c_void_p, c_void_p]),
    "pe_conv3x3_c64_pw_proj_fwd_bf16": (
        c_int32, [c_int32] * 3 + [c_void_p] * 4 + [c_int32, c_void_p, c_int32] + [c_void_p] * 8),
    "pe_conv3x3_c64_pw_next_fwd_bf16": (
        c_int32, [c_int32] * 3 + [c_void_p] * 4 + [c_int32, c_void_p, c_int32] + [c_void_p] * 5
        + [c_int32] + [c_void_p] * 4),
    "pe_stem7x7s2_pool_fwd_bf16": (
        c_int32, [c_int32] * 4 + [c_void_p] * 6),
    "pe_deconv4x4s2_head_fwd_bf16": (
        c_int32, [c_int32] * 6 + [c_void_p] * 6 + [c_int32, c_void_p, c_void_p]),
    "pe_boxes_from_joints_f32": (
        c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                  c_void_p]),
    "pe_box_loss_workspace_bytes": (c_size_t, [c_int32]),
    "pe_box_loss_fwd_bwd": (
        c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_float] + [c_void_p] * 5),
    "pe_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "pe_conv2d_fwd_bnstats_bf16": (
        c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pe_deconv4x4s2_fwd_bnstats_bf16": (
        c_int32, [c_int32] * 5 + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                  c_void_p]),
    "pe_stem7x7s2_fwd_bnstats_bf16": (
        c_int32, [c_int32] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                  c_void_p]),
    "pe_bn_finalize_bf16": (
        c_int32, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float]
        + [c_void_p] * 7),
    "pe_bn_stats_bf16": (
        c_int32,
        [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "pe_bn_apply_bf16": (
        c_int32,
        [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "pe_conv2d_bwd_data_bnmask_bf16": (
        c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                  c_void_p, c_size_t, c_void_p, c_void_p]),
    "pe_conv2d_bwd_data_resmask_bf16": (
        c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                  c_size_t, c_void_p, c_void_p]),
    "pe_deconv4x4s2_bwd_data_bnmask_bf16": (
        c_int32, [c_int32] * 5 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                  c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pe_bn_bwd_from_partials_bf16": (
        c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                  c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "pe_bn_bwd_bf16": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
         c_void_p],
    ),
    "pe_bn_sync_bytes": (c_size_t, []),
    "pe_bn_sync_init": (c_int32, [c_void_p, c_void_p]),
    "pe_bn_stats_apply_bf16": (
        c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float]
        + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pe_bn_finalize_apply_bf16": (
        c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float,
                  c_float] + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p, c_void_p]),
    "pe_bn_fold_f32": (c_int32, [c_void_p, c_int32, c_float, c_void_p]),
}


class ConvDesc(ctypes.Structure):
    """Mirror of pe_conv_desc (include/pe_api.h)."""

    _fields_ = [(n, c_int32) for n in ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad",
                                       "relu", "out_f32_nchw")]


class ConvStage(ctypes.Structure):
    """Mirror of pe_conv_stage (include/pe_api.h)."""

    _fields_ = [("desc", ConvDesc), ("w", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("residual", c_void_p), ("y", c_void_p)]


PE_CONV_MAX_CHAIN = 3
PE_WGRAD_MAX_BATCH = 8


class WgradJob(ctypes.Structure):
    """Mirror of pe_wgrad_job (include/pe_api.h)."""

    _fields_ = [("desc", ConvDesc), ("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p),
                ("ldy", c_int32), ("reserved", c_int32)]


_lib = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -m pose_estimation_amd.build` (needs hipcc); there is no CPU fallback."
            )
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status: int) -> None:
    if status != PE_OK:
        raise PeError(status, lib().pe_last_error().decode("utf-8", "replace"))
