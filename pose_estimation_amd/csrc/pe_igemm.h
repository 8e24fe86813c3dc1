// Implicit-GEMM convolution engine for gfx950 (bf16 MFMA, fp32 accumulate).
//
// One kernel family serves every dense convolution on the path:
//   forward conv (1x1 / 3x3 / 7x7-as-space-to-depth, stride 1 or 2), the four output-parity
//   phases of the stride-2 transposed conv, and every backward-data pass (a backward-data pass
//   is itself a gather-conv over the output gradient with re-packed weights).
//
//   D[n][m] = sum_{tap, c} Wp[n][tap*Cin + c] * X[b, oy*sy + dy(tap), ox*sx + dx(tap), c]
//
//   m enumerates GEMM rows (b, oy, ox) over a Ho x Wo grid; the result lands at output pixel
//   (b, oy*osy + ooy, ox*osx + oox) of an OH x OW map, which is how a deconv phase scatters.
#pragma once

#include "pe_common.h"

namespace pe {

constexpr int kMaxTaps = 64;
constexpr int kMaxPhases = 4;

struct IgemmParams {
    const uint16_t* x;      // activations, NHWC bf16, Cx channels per pixel
    const uint16_t* w;      // packed weights [N][T*Cin] bf16, K contiguous
    void* y;                // output
    const float* scale;     // per-output-channel scale (null = 1)
    const float* shift;     // per-output-channel shift / bias (null = 0)
    const uint16_t* res;    // residual added before ReLU, laid out like a bf16 NHWC y (null = none)
    int B, Hi, Wi, Cx;      // input geometry
    int Ho, Wo;             // GEMM row grid
    int sy, sx;             // input step per GEMM row step
    int N;                  // output channels
    int Cin;                // K elements per tap, multiple of 64 (may span pixels: see cshift)
    int cshift;             // log2(Cx) when a tap's K run spans several pixels, else 31
    int cmask;              // Cx-1 in that case, else 0x7fffffff
    int T;                  // taps
    int wk_row;             // weight row length in elements (0 = T*Cin, taps packed densely)
    int OH, OW, osy, osx, ooy, oox;  // output scatter
    int ldc;                // output channels per pixel (bf16 NHWC) / unused for f32 NCHW
    int relu;               // 0: none   1: ReLU   2: ReLU6
    int out_mode;           // 0: bf16 NHWC   1: f32 NCHW [B][N][OH*OW]
    int16_t tap_dy[kMaxTaps];
    int16_t tap_dx[kMaxTaps];
    uint8_t tap_slot[kMaxTaps];  // position of each tap's Cin-block inside a weight row
    // Fused batch-norm statistics (training): when stat_partial is set, every output tile also
    // writes, per output channel, the sum and the sum of squares of its bf16-rounded values to
    // stat_partial[(phase * stat_ntm + tile_m) * 2 + {0, 1}][N]; needs no scale/shift/residual/
    // ReLU, bf16 NHWC output and equally sized phases.  stat_rows_out receives the row count.
    float* stat_partial;
    long stat_capacity_rows;
    int32_t* stat_rows_out;   // host pointer
    int stat_ntm;             // filled in by igemm_launch
    int nt_store;             // filled in by igemm_launch: non-temporal output stores
    // Fused batch-norm backward, first half (training, backward-data launches): when mask_z is
    // set the result is the gradient da of an activation a = act(z * mask_sc + mask_sh) whose
    // pre-norm tensor z is laid out like y.  The epilogue rebuilds a from z (same operations
    // and rounding as the forward normalise pass), stores g = (0 < a < mask_hi) ? da : 0 and
    // writes, per output channel and tile, sum g and sum g*z to stat_partial (same layout as the
    // forward statistics).  Needs bf16 NHWC output, no scale/shift/residual/ReLU, equal phases.
    const uint16_t* mask_z;
    const float* mask_sc;
    const float* mask_sh;
    float mask_hi;            // +inf: ReLU, 6: ReLU6
    // Residual layers: a = relu(bn(z) + identity) cannot be rebuilt from z, so the mask is read
    // from the stored activation (mask_a, same layout as the output; mask_sc / mask_sh unused),
    // and `res` may be set: the masked value is the GEMM's result plus the residual.
    const uint16_t* mask_a;
    // Fused pointwise head (igemm_head_launch only): the layer's bf16 output tile stays in LDS
    // and a 1x1 conv over ALL of its N channels (head_w [head_N][N] bf16, bias head_bias) is
    // applied to it; only that conv's float32 NCHW result [B][head_N][OH*OW] is written.
    const uint16_t* head_w;
    const float* head_bias;
    float* head_out;
    int head_N;
    int32_t tap_info[kMaxTaps];  // dy | dx << 8 | slot << 16, packed by igemm_launch for the kernel
    // Optional phases (gridDim.y): phase i uses taps [ph_t0[i], ph_t0[i] + ph_T[i]), the row grid
    // ph_Ho[i] x ph_Wo[i], output offset (ph_ooy[i], ph_oox[i]) and weights w + ph_woff[i].
    // nphase == 0: one phase taken from T, Ho, Wo, ooy, oox.
    int nphase;
    int16_t ph_t0[kMaxPhases], ph_T[kMaxPhases], ph_ooy[kMaxPhases], ph_oox[kMaxPhases];
    int ph_Ho[kMaxPhases], ph_Wo[kMaxPhases];
    long ph_woff[kMaxPhases];
};

// Neutral defaults: unit strides, identity scatter, densely packed taps, single-pixel K runs.
inline void igemm_defaults(IgemmParams& p) {
    p = IgemmParams{};
    p.cshift = 31;
    p.cmask = 0x7fffffff;
    p.osy = p.osx = 1;
    p.sy = p.sx = 1;
    for (int i = 0; i < kMaxTaps; ++i) p.tap_slot[i] = (uint8_t)i;
}

// Launches the engine (asynchronous).  Returns PE_OK / PE_E_*.
int32_t igemm_launch(const IgemmParams& p, hipStream_t st, const char* what);

// The layer described by p (N == 256, single-pixel K runs that are multiples of 64) followed by
// the 1x1 conv p.head_* over its output, in one launch; p.y is not written.
int32_t igemm_head_launch(const IgemmParams& p, hipStream_t st, const char* what);

// Up to kMaxChain layers over the same pixel tiles in one launch: stage s > 0 is a 1x1 stride-1
// layer whose input is the output of stage s - 1 (see igemm_chain_kernel).
constexpr int kMaxChain = 3;
struct IgemmChainArgs {
    IgemmParams st[kMaxChain];
    int nstage;
};
int32_t igemm_chain_launch(const IgemmParams* stages, int nstage, hipStream_t st, const char* what);

}  // namespace pe
