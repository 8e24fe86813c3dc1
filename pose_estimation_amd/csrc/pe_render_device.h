// Device helpers shared by the target-render kernels and the fused render+loss kernels, so
// both produce bit-identical targets.
#pragma once

#include "pe_common.h"

namespace pe {

struct JointWindow {
    int cx;   // window centre, heat-map cells
    int cy;
    bool on;  // weight != 0
};

// j -> (x, y, v) in input pixels.  Centre = floor(coord/stride + 0.5) in float32; the joint
// is off when it is not labelled (v > 0 fails), not finite, or its centre cell is off the map
// (the reference's label at test.ipynb:606-653 leaves the heat map of such a joint empty).
__device__ __forceinline__ JointWindow joint_window(const float* __restrict__ j, int H, int W,
                                                    float stride, int radius) {
    const float x = j[0], y = j[1], v = j[2];
    const float mx = floorf(x / stride + 0.5f);
    const float my = floorf(y / stride + 0.5f);
    const float r = (float)radius;
    JointWindow w;
    (void)r;
    w.on = (v > 0.0f) && isfinite(x) && isfinite(y) && (mx < (float)W) && (mx >= 0.0f) &&
           (my < (float)H) && (my >= 0.0f);
    w.cx = w.on ? (int)mx : 0;
    w.cy = w.on ? (int)my : 0;
    return w;
}

// Vector-map channel c = 4*pair + 2*half + axis of image b: cells of the source joint's window
// hold the integer offset (axis 0: x, 1: y) from the cell to the partner joint's centre.  half 0:
// source = first joint of the pair; half 1: source = second.  Centres may lie off the map
// (test.ipynb:634-653 keeps the entries of the clipped window); both joints must be labelled,
// finite and within kVecMaxCentre cells of the origin.
constexpr float kVecMaxCentre = 1048576.0f;

struct VecWindow {
    int sx, sy;  // source centre: the window is |x - sx| <= r, |y - sy| <= r
    int dx, dy;  // partner centre
    bool on;
};

__device__ __forceinline__ bool vec_centre(const float* __restrict__ j, float stride, int& cx,
                                           int& cy) {
    const float x = j[0], y = j[1], v = j[2];
    const float mx = floorf(x / stride + 0.5f);
    const float my = floorf(y / stride + 0.5f);
    const bool ok = (v > 0.0f) && isfinite(x) && isfinite(y) && fabsf(mx) < kVecMaxCentre &&
                    fabsf(my) < kVecMaxCentre;
    cx = ok ? (int)mx : 0;
    cy = ok ? (int)my : 0;
    return ok;
}

__device__ __forceinline__ VecWindow vec_window(const float* __restrict__ joints_b, int k1, int k2,
                                                int half, float stride) {
    int ax, ay, bx, by;
    const bool oka = vec_centre(joints_b + 3 * k1, stride, ax, ay);
    const bool okb = vec_centre(joints_b + 3 * k2, stride, bx, by);
    VecWindow w;
    w.on = oka && okb;
    w.sx = half ? bx : ax;
    w.sy = half ? by : ay;
    w.dx = half ? ax : bx;
    w.dy = half ? ay : by;
    return w;
}

// g lives in LDS; entries above `radius` are never read.
__device__ __forceinline__ float gauss_x(const float* g, int x, int cx, int radius) {
    const int ad = abs(x - cx);
    const float t = g[min(ad, radius)];
    return ad <= radius ? t : 0.0f;
}

}  // namespace pe
