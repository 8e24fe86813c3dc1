// Parameters of the backward-weight GEMM (pe_wgrad.hip).  The gather side (x, taps, strides,
// K runs spanning pixels) and the scatter side (dy addressed through OH/OW/os*/oo*) mirror
// IgemmParams, so a layer is described to both engines the same way.
#pragma once

#include "pe_igemm.h"

namespace pe {

// The pixel range is split until about this many workgroups exist; each split leaves a partial
// dW slab in the workspace and an ordered reduction folds them (bit-stable, no float atomics).
// Two workgroups per CU measured best on the training step (256: 14.2 ms, 512: 13.2, 1024: 13.3).
constexpr long kWgradTargetBlocks = 512;
// Weights with at least this many 128 x 128 output tiles aim for half as many workgroups: every
// split costs a slab of the weight's size.  Measured per step (tiles >= 8 vs never): ResNet-50
// 9.07 vs 9.26 ms, HRNet-style 37.7 vs 38.5 ms, MobileNetV2 16.44 vs 16.53 ms; with a threshold
// of 4 the HRNet step loses what it gained (38.6 ms).
constexpr long kWgradBigTiles = 8;
constexpr long kWgradMaxBlocks = 2048;   // upper bound the workspace is sized for
constexpr int kWgradMaxBatch = 8;        // weight gradients per batched launch
// Workgroups of a batched launch.  Two per CU measured best on the training steps, where the
// launch runs beside the main chain (ResNet-50 8.49 / 8.41 / 8.44 / 8.55 ms and HRNet-style
// 34.9 / 33.0 / 33.3 / 33.9 ms for 256 / 512 / 640 / 1024); alone, more is faster.
constexpr long kWgradBatchBlocks = 512;

struct WgradParams {
    const uint16_t* x;   // forward input, NHWC bf16, Cx channels per pixel
    const uint16_t* dy;  // output gradient, NHWC bf16, ldy channels per pixel
    float* dw;           // [Nw][T*Cin] float32, accumulated into (the caller zeroes it)
    float* slab;         // workspace: one partial [Nw][T*Cin] per pixel split (filled in)
    long slab_stride;    // elements between slabs (filled in)
    int B, Hi, Wi, Cx;
    int Ho, Wo, sy, sx;
    int N;               // channels of dy that take part (multiple of 8)
    int Nw;              // rows of dw that exist (<= N)
    int Cin, cshift, cmask, T;
    int OH, OW, osy, osx, ooy, oox, ldy;
    long m_per_split;    // filled in by wgrad_launch
    int tiles, splits;   // filled in by wgrad_launch: 128 x 128 output tiles, pixel splits
    int16_t tap_dy[kMaxTaps];
    int16_t tap_dx[kMaxTaps];
};

inline void wgrad_defaults(WgradParams& p) {
    p = WgradParams{};
    p.cshift = 31;
    p.cmask = 0x7fffffff;
    p.sy = p.sx = 1;
    p.osy = p.osx = 1;
}

// Workspace needed for a weight of P = rows * (taps * Cin) elements.
size_t wgrad_workspace_bytes(long P);

int32_t wgrad_launch(WgradParams& p, void* workspace, size_t workspace_bytes, hipStream_t st,
                     const char* what);

// Up to kWgradMaxBatch weight gradients in one launch plus one fold launch; the workspace
// (wgrad_batch_workspace_bytes of the weights' total size) is shared out among the jobs.
int32_t wgrad_launch_batch(WgradParams* ps, int n, void* workspace, size_t workspace_bytes,
                           hipStream_t st, const char* what);
size_t wgrad_batch_workspace_bytes(long P_total);

}  // namespace pe
