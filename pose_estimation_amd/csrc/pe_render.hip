// a2. Gaussian target rendering.
//
// Bound: HBM write bandwidth (every map is written in full, zeros included).
// The Gaussian is separable and its (radius+1)-entry table is computed once on the host in
// double precision and travels in the kernel arguments, so the device evaluates no expf and
// the result is bit-identical to the CPU restatement: target = g[|dx|] * g[|dy|].
#include <math.h>

#include "pe_common.h"
#include "pe_render_device.h"

namespace pe {

// Contiguous [B,K,H,W] output, W % 4 == 0: one wave per map, 16-byte stores.
__global__ __launch_bounds__(256) void render_contig_kernel(const float* __restrict__ joints,
                                                            int nmaps, int H, int W,
                                                            float stride, int radius,
                                                            GaussTable tab,
                                                            float* __restrict__ target,
                                                            float* __restrict__ weight) {
    __shared__ float g[PE_MAX_RADIUS + 1];
    if (threadIdx.x <= PE_MAX_RADIUS) g[threadIdx.x] = tab.g[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= nmaps) return;  // wave-uniform, after the only barrier
    const JointWindow jw = joint_window(joints + map * 3, H, W, stride, radius);
    const int n4 = (H * W) >> 2;
    const int w4 = W >> 2;
    f32x4* dst = reinterpret_cast<f32x4*>(target + map * (long)(H * W));
    for (int c = lane; c < n4; c += 64) {
        const int y = c / w4;
        const int x0 = (c - y * w4) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        const int ady = abs(y - jw.cy);
        if (jw.on && ady <= radius && x0 + 3 >= jw.cx - radius && x0 <= jw.cx + radius) {
            const float gy = g[ady];
            o.x = gauss_x(g, x0 + 0, jw.cx, radius) * gy;
            o.y = gauss_x(g, x0 + 1, jw.cx, radius) * gy;
            o.z = gauss_x(g, x0 + 2, jw.cx, radius) * gy;
            o.w = gauss_x(g, x0 + 3, jw.cx, radius) * gy;
        }
        __builtin_nontemporal_store(o, dst + c);
    }
    if (lane == 0) weight[map] = jw.on ? 1.0f : 0.0f;
}

// Any output strides (e.g. the reference's NHWC label layout): one thread per element,
// threads ordered along the fastest-varying memory axis chosen by the host.
__global__ __launch_bounds__(256) void render_strided_kernel(const float* __restrict__ joints,
                                                             int B, int K, int H, int W,
                                                             float stride, int radius,
                                                             GaussTable tab, float* target,
                                                             long tB, long tK, long tH, long tW,
                                                             int k_fastest, float* weight) {
    __shared__ float g[PE_MAX_RADIUS + 1];
    if (threadIdx.x <= PE_MAX_RADIUS) g[threadIdx.x] = tab.g[threadIdx.x];
    __syncthreads();
    const long total = (long)B * K * H * W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int b, k, y, x;
        long r = e;
        if (k_fastest) {  // enumerate as [B,H,W,K]
            k = (int)(r % K); r /= K;
            x = (int)(r % W); r /= W;
            y = (int)(r % H); r /= H;
            b = (int)r;
        } else {          // enumerate as [B,K,H,W]
            x = (int)(r % W); r /= W;
            y = (int)(r % H); r /= H;
            k = (int)(r % K); r /= K;
            b = (int)r;
        }
        const long map = (long)b * K + k;
        const JointWindow jw = joint_window(joints + map * 3, H, W, stride, radius);
        float o = 0.f;
        const int ady = abs(y - jw.cy);
        if (jw.on && ady <= radius) o = gauss_x(g, x, jw.cx, radius) * g[ady];
        target[b * tB + k * tK + y * tH + x * tW] = o;
        if (y == 0 && x == 0) weight[map] = jw.on ? 1.0f : 0.0f;
    }
}

}  // namespace pe

extern "C" int32_t pe_render_targets_f32(const float* joints, int32_t B, int32_t K, int32_t H,
                                         int32_t W, float stride, float sigma, int32_t radius,
                                         float* target, int64_t tB, int64_t tK, int64_t tH,
                                         int64_t tW, float* weight, pe_stream_t stream) {
    using namespace pe;
    PE_REQUIRE(B >= 0 && K >= 0, PE_E_INVALID, "pe_render_targets_f32: negative batch or joints");
    PE_REQUIRE(H > 0 && W > 0, PE_E_INVALID, "pe_render_targets_f32: H and W must be positive");
    PE_REQUIRE(radius >= 0 && radius <= PE_MAX_RADIUS, PE_E_UNSUPPORTED,
               "pe_render_targets_f32: radius %d outside [0, %d]", radius, PE_MAX_RADIUS);
    PE_REQUIRE(sigma > 0.0f && isfinite(sigma), PE_E_INVALID,
               "pe_render_targets_f32: sigma must be positive and finite");
    PE_REQUIRE(stride > 0.0f && isfinite(stride), PE_E_INVALID,
               "pe_render_targets_f32: stride must be positive and finite");
    const int64_t nmaps = (int64_t)B * K;
    PE_REQUIRE(nmaps * H * W < ((int64_t)1 << 40), PE_E_UNSUPPORTED,
               "pe_render_targets_f32: tensor too large");
    PE_REQUIRE(nmaps < 0x7fffffff && (int64_t)H * W < 0x7fffffff, PE_E_UNSUPPORTED,
               "pe_render_targets_f32: B*K or H*W does not fit int32");
    if (nmaps == 0) return PE_OK;
    PE_REQUIRE(joints && target && weight, PE_E_INVALID, "pe_render_targets_f32: NULL pointer");
    PE_REQUIRE(tB >= 0 && tK >= 0 && tH >= 0 && tW >= 0, PE_E_INVALID,
               "pe_render_targets_f32: negative strides are not supported");
    GaussTable tab;
    fill_gauss_table(sigma, radius, &tab);
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)H * W;
    const bool contig = (tW == 1 && tH == W && tK == n && tB == (int64_t)K * n);
    if (contig && (W % 4) == 0 && aligned(target, 16)) {
        const unsigned grid = (unsigned)((nmaps + 3) / 4);
        pe::launch(render_contig_kernel, grid, 256, 0, st, joints, (int)nmaps, H, W, stride, radius, tab,
                                                   target, weight);
        return check_launch("pe_render_targets_f32(contiguous)");
    }
    const int64_t total = nmaps * n;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const int k_fastest = (tK < tW) ? 1 : 0;
    pe::launch(render_strided_kernel, (unsigned)blocks, 256, 0, st, joints, B, K, H, W, stride, radius,
                                                            tab, target, tB, tK, tH, tW,
                                                            k_fastest, weight);
    return check_launch("pe_render_targets_f32(strided)");
}
