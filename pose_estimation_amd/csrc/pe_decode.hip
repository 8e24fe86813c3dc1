// a1. Heat-map peak decode: per-(b,k) arg-max, confidence and quarter-pixel refinement.
//
// Bound: HBM read bandwidth (one pass over every map, 16 B written per map).
// Contiguous maps: one 64-lane wave owns one map, issues all of its 16-byte loads up front
// (H*W/256 per lane: 12 at 64x48), reduces (value, index) in registers, then butterflies
// across the wave.  Four waves (= four maps) per workgroup keep 8 waves per SIMD resident.
// Any other layout (e.g. the NHWC label layout) goes through the strided kernels below.
#include <limits.h>
#include <math.h>

#include "pe_common.h"

namespace pe {

struct DecodeOut {
    float* xy;
    float* conf;
    int32_t* idx;
};

// Shared tail: lane 0 of the wave owning `map` applies the refinement and stores the results.
// `at(y, x)` returns the map element; it is only called for in-range neighbours.
template <typename At>
__device__ __forceinline__ void decode_finish(long map, float best, int bidx, int H, int W,
                                              int refine, const DecodeOut& out, At at) {
    const int y = bidx / W;
    const int x = bidx - y * W;
    float fx = (float)x;
    float fy = (float)y;
    if (refine && x > 1 && x < W - 1 && y > 1 && y < H - 1) {
        const float r = at(y, x + 1), l = at(y, x - 1);
        const float d = at(y + 1, x), u = at(y - 1, x);
        fx += 0.25f * (float)((r > l) - (r < l));
        fy += 0.25f * (float)((d > u) - (d < u));
    }
    if (!(best > 0.0f)) {
        fx = 0.0f;
        fy = 0.0f;
    }
    out.xy[2 * map + 0] = fx;
    out.xy[2 * map + 1] = fy;
    out.conf[map] = best;
    out.idx[map] = bidx;
}

template <int NV>
__global__ __launch_bounds__(256) void decode_contig_kernel(const float* __restrict__ hm,
                                                            int nmaps, int H, int W, int refine,
                                                            DecodeOut out) {
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= nmaps) return;  // wave-uniform
    const int n = H * W;
    const int n4 = n >> 2;
    const float* base = hm + map * (long)n;
    const f32x4* src = reinterpret_cast<const f32x4*>(base);

    float best = -INFINITY;
    int bidx = INT_MAX;
    for (int c0 = 0; c0 < n4; c0 += 64 * NV) {
        f32x4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = c0 + j * 64 + lane;
            v[j] = (c < n4) ? __builtin_nontemporal_load(src + c)
                            : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = (c0 + j * 64 + lane) * 4;
            // ascending index inside a lane, so a strict compare keeps the first maximum
            if (v[j].x > best) { best = v[j].x; bidx = e; }
            if (v[j].y > best) { best = v[j].y; bidx = e + 1; }
            if (v[j].z > best) { best = v[j].z; bidx = e + 2; }
            if (v[j].w > best) { best = v[j].w; bidx = e + 3; }
        }
    }
    wave_argmax(best, bidx);
    if (lane == 0) {
        if (bidx == INT_MAX) {  // no element above -inf: numpy.argmax answers 0
            bidx = 0;
            best = base[0];
        }
        decode_finish(map, best, bidx, H, W, refine, out,
                      [&](int y, int x) { return base[y * W + x]; });
    }
}

// Any strides: rows in order, lanes across x, so per-lane indices still ascend.
__global__ __launch_bounds__(256) void decode_strided_kernel(const float* __restrict__ hm,
                                                             int nmaps, int K, int H, int W,
                                                             long sB, long sK, long sH, long sW,
                                                             int refine, DecodeOut out) {
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= nmaps) return;
    const long b = map / K;
    const long k = map - b * K;
    const float* base = hm + b * sB + k * sK;
    float best = -INFINITY;
    int bidx = INT_MAX;
    for (int y = 0; y < H; ++y) {
        const float* row = base + (long)y * sH;
        for (int x = lane; x < W; x += 64) {
            const float v = row[(long)x * sW];
            if (v > best) { best = v; bidx = y * W + x; }
        }
    }
    wave_argmax(best, bidx);
    if (lane == 0) {
        if (bidx == INT_MAX) {
            bidx = 0;
            best = base[0];
        }
        decode_finish(map, best, bidx, H, W, refine, out,
                      [&](int y, int x) { return base[(long)y * sH + (long)x * sW]; });
    }
}

// Channels-last maps ([B,H,W,K] memory, sK == 1): one workgroup per image reads the image
// in memory order (coalesced); thread t keeps the running maximum of channel (t % K) over the
// pixels it visits when 256 % K == 0 is not required: each thread walks elements
// t, t + T, t + 2T, ... with T = the largest multiple of K not above 256, so its channel
// never changes.  Partial results meet in LDS.
__global__ __launch_bounds__(256) void decode_nhwc_kernel(const float* __restrict__ hm, int K,
                                                          int H, int W, long sB, int refine,
                                                          DecodeOut out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = (256 / K) * K;  // active threads
    const int G = T / K;          // pixel lanes per channel
    float* sval = reinterpret_cast<float*>(smem);
    int* sidx = reinterpret_cast<int*>(smem + sizeof(float) * 256);
    const long b = blockIdx.x;
    const float* base = hm + b * sB;
    const int t = threadIdx.x;
    const int n = H * W;
    float best = -INFINITY;
    int bidx = INT_MAX;
    if (t < T) {
        const int k = t % K;
        const int p0 = t / K;
        // 4 independent loads in flight per thread
        int p = p0;
        for (; p + 3 * G < n; p += 4 * G) {
            const float v0 = base[(long)p * K + k];
            const float v1 = base[(long)(p + G) * K + k];
            const float v2 = base[(long)(p + 2 * G) * K + k];
            const float v3 = base[(long)(p + 3 * G) * K + k];
            if (v0 > best) { best = v0; bidx = p; }
            if (v1 > best) { best = v1; bidx = p + G; }
            if (v2 > best) { best = v2; bidx = p + 2 * G; }
            if (v3 > best) { best = v3; bidx = p + 3 * G; }
        }
        for (; p < n; p += G) {
            const float v = base[(long)p * K + k];
            if (v > best) { best = v; bidx = p; }
        }
    }
    sval[t] = best;
    sidx[t] = bidx;
    __syncthreads();
    if (t < K) {
        float v = sval[t];
        int i = sidx[t];
        for (int g = 1; g < G; ++g) argmax_combine(v, i, sval[g * K + t], sidx[g * K + t]);
        if (i == INT_MAX) {
            i = 0;
            v = base[t];
        }
        const long map = b * K + t;
        decode_finish(map, v, i, H, W, refine, out,
                      [&](int y, int x) { return base[((long)y * W + x) * K + t]; });
    }
}

}  // namespace pe

extern "C" int32_t pe_decode_f32(const float* hm, int32_t B, int32_t K, int32_t H, int32_t W,
                                 int64_t sB, int64_t sK, int64_t sH, int64_t sW, int32_t refine,
                                 float* xy, float* conf, int32_t* idx, pe_stream_t stream) {
    using namespace pe;
    PE_REQUIRE(B >= 0 && K >= 0, PE_E_INVALID, "pe_decode_f32: negative batch or joint count");
    PE_REQUIRE(H > 0 && W > 0, PE_E_INVALID, "pe_decode_f32: H and W must be positive");
    PE_REQUIRE((int64_t)H * W < (int64_t)INT_MAX, PE_E_UNSUPPORTED,
               "pe_decode_f32: H*W does not fit int32");
    const int64_t nmaps = (int64_t)B * K;
    PE_REQUIRE(nmaps < (int64_t)INT_MAX, PE_E_UNSUPPORTED, "pe_decode_f32: B*K does not fit int32");
    if (nmaps == 0) return PE_OK;
    PE_REQUIRE(hm && xy && conf && idx, PE_E_INVALID, "pe_decode_f32: NULL tensor pointer");
    PE_REQUIRE(sB >= 0 && sK >= 0 && sH >= 0 && sW >= 0, PE_E_INVALID,
               "pe_decode_f32: negative strides are not supported");

    const DecodeOut out{xy, conf, idx};
    hipStream_t st = as_stream(stream);
    const int n = H * W;
    const bool contig = (sW == 1 && sH == W && sK == (int64_t)n && sB == (int64_t)K * n);
    const unsigned grid = (unsigned)((nmaps + 3) / 4);
    if (contig && (n % 4) == 0 && aligned(hm, 16)) {
        const int per_lane = (n / 4 + 63) / 64;
        if (per_lane <= 4) {
            pe::launch(decode_contig_kernel<4>, grid, 256, 0, st, hm, (int)nmaps, H, W, refine, out);
        } else if (per_lane <= 8) {
            pe::launch(decode_contig_kernel<8>, grid, 256, 0, st, hm, (int)nmaps, H, W, refine, out);
        } else if (per_lane <= 12) {
            pe::launch(decode_contig_kernel<12>, grid, 256, 0, st, hm, (int)nmaps, H, W, refine, out);
        } else {
            pe::launch(decode_contig_kernel<16>, grid, 256, 0, st, hm, (int)nmaps, H, W, refine, out);
        }
        return check_launch("pe_decode_f32(contiguous)");
    }
    const bool nhwc = (sK == 1 && sW == K && sH == (int64_t)W * K && K <= 128);
    if (nhwc) {
        pe::launch(decode_nhwc_kernel, (unsigned)B, 256, 256 * 8, st, hm, K, H, W, sB, refine, out);
        return check_launch("pe_decode_f32(nhwc)");
    }
    pe::launch(decode_strided_kernel, grid, 256, 0, st, hm, (int)nmaps, K, H, W, sB, sK, sH, sW, refine,
                                                out);
    return check_launch("pe_decode_f32(strided)");
}
