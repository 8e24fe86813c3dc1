// Parameters of the backward-weight GEMM (pe_wgrad.hip).  The gather side (x, taps, strides,
// K runs spanning pixels) and the scatter side (dy addressed through OH/OW/os*/oo*) mirror
// IgemmParams, so a layer is described to both engines the same way.
#pragma once

#include "pe_igemm.h"

namespace pe {

struct WgradParams {
    const uint16_t* x;   // forward input, NHWC bf16, Cx channels per pixel
    const uint16_t* dy;  // output gradient, NHWC bf16, ldy channels per pixel
    float* dw;           // [Nw][T*Cin] float32, accumulated with atomics (caller zeroes)
    int B, Hi, Wi, Cx;
    int Ho, Wo, sy, sx;
    int N;               // channels of dy that take part (multiple of 8)
    int Nw;              // rows of dw that exist (<= N)
    int Cin, cshift, cmask, T;
    int OH, OW, osy, osx, ooy, oox, ldy;
    long m_per_split;    // filled in by wgrad_launch
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

int32_t wgrad_launch(WgradParams& p, hipStream_t st, const char* what);

}  // namespace pe
