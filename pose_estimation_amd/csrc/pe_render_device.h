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
// is off when it is not visible (v > 0 fails), not finite, or its window misses the map.
__device__ __forceinline__ JointWindow joint_window(const float* __restrict__ j, int H, int W,
                                                    float stride, int radius) {
    const float x = j[0], y = j[1], v = j[2];
    const float mx = floorf(x / stride + 0.5f);
    const float my = floorf(y / stride + 0.5f);
    const float r = (float)radius;
    JointWindow w;
    w.on = (v > 0.0f) && isfinite(x) && isfinite(y) && (mx - r < (float)W) && (mx + r >= 0.0f) &&
           (my - r < (float)H) && (my + r >= 0.0f);
    w.cx = w.on ? (int)mx : 0;
    w.cy = w.on ? (int)my : 0;
    return w;
}

// g lives in LDS; entries above `radius` are never read.
__device__ __forceinline__ float gauss_x(const float* g, int x, int cx, int radius) {
    const int ad = abs(x - cx);
    const float t = g[min(ad, radius)];
    return ad <= radius ? t : 0.0f;
}

}  // namespace pe
