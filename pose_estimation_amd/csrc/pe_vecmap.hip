// Vector-map labels (per keypoint pair, the integer cell offset to the partner joint) and their
// loss.  The reference's second dense label: test.ipynb:634-653 prints its support, and its
// drawing cell reads channels 4*i, 4*i+1 at the first joint's cells and 4*i+2, 4*i+3 at the
// second's.
//
// Bound: HBM bandwidth.  Render writes every map in full (zeros included), one wave per (b, c)
// map with 16-byte stores, or one thread per element for other layouts (the reference's NHWC).
// The loss never materialises the label: it reads pred, rebuilds the label in registers and
// writes the gradient (2 x 4 B per element, 2 x 2 B with bf16); cells outside a window carry no
// loss and no gradient.  Per-map partial sums are folded in a fixed order (bit-stable loss).
#include <math.h>

#include "pe_common.h"
#include "pe_render_device.h"

namespace pe {

constexpr int kMaxPairs = 32;

struct PairTable {
    int16_t k1[kMaxPairs];
    int16_t k2[kMaxPairs];
};

__device__ __forceinline__ float vec_value(const VecWindow& w, int axis, int radius, int y, int x,
                                           bool& inside) {
    inside = w.on && abs(y - w.sy) <= radius && abs(x - w.sx) <= radius;
    const int v = axis ? w.dy - y : w.dx - x;
    return inside ? (float)v : 0.0f;
}

__global__ __launch_bounds__(256) void vec_render_contig_kernel(const float* __restrict__ joints,
                                                                int K, PairTable pt, int C,
                                                                int nmaps, int H, int W,
                                                                float stride, int radius,
                                                                float* __restrict__ vec) {
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= nmaps) return;
    const long b = map / C;
    const int c = (int)(map - b * C);
    const int pair = c >> 2, half = (c >> 1) & 1, axis = c & 1;
    const VecWindow w = vec_window(joints + b * K * 3, pt.k1[pair], pt.k2[pair], half, stride);
    const int n4 = (H * W) >> 2;
    const int w4 = W >> 2;
    f32x4* dst = reinterpret_cast<f32x4*>(vec + map * (long)(H * W));
    for (int q = lane; q < n4; q += 64) {
        const int y = q / w4;
        const int x0 = (q - y * w4) * 4;
        bool in;
        f32x4 o;
        o.x = vec_value(w, axis, radius, y, x0 + 0, in);
        o.y = vec_value(w, axis, radius, y, x0 + 1, in);
        o.z = vec_value(w, axis, radius, y, x0 + 2, in);
        o.w = vec_value(w, axis, radius, y, x0 + 3, in);
        __builtin_nontemporal_store(o, dst + q);
    }
}

__global__ __launch_bounds__(256) void vec_render_strided_kernel(const float* __restrict__ joints,
                                                                 int B, int K, PairTable pt, int C,
                                                                 int H, int W, float stride,
                                                                 int radius, float* vec, long sB,
                                                                 long sC, long sH, long sW,
                                                                 int c_fastest) {
    const long total = (long)B * C * H * W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int b, c, y, x;
        long r = e;
        if (c_fastest) {  // enumerate as [B,H,W,C]
            c = (int)(r % C); r /= C;
            x = (int)(r % W); r /= W;
            y = (int)(r % H); r /= H;
            b = (int)r;
        } else {          // enumerate as [B,C,H,W]
            x = (int)(r % W); r /= W;
            y = (int)(r % H); r /= H;
            c = (int)(r % C); r /= C;
            b = (int)r;
        }
        const int pair = c >> 2, half = (c >> 1) & 1, axis = c & 1;
        const VecWindow w = vec_window(joints + (long)b * K * 3, pt.k1[pair], pt.k2[pair], half, stride);
        bool in;
        vec[b * sB + c * sC + y * sH + x * sW] = vec_value(w, axis, radius, y, x, in);
    }
}

struct VecLossArgs {
    const void* pred;
    const float* joints;
    const uint8_t* mask;  // [B,H,W], may be null
    void* grad;           // may be null
    float* partial;       // [B*C]
    int nmaps, K, C, H, W, radius;
    float stride, inv_n;
    PairTable pt;
};

template <bool BF16, bool HAS_MASK, bool HAS_GRAD>
__global__ __launch_bounds__(256) void vec_loss_kernel(VecLossArgs a) {
    constexpr int CH = BF16 ? 8 : 4;
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= a.nmaps) return;
    const long b = map / a.C;
    const int c = (int)(map - b * a.C);
    const int pair = c >> 2, half = (c >> 1) & 1, axis = c & 1;
    const VecWindow w = vec_window(a.joints + b * a.K * 3, a.pt.k1[pair], a.pt.k2[pair], half, a.stride);
    const int n = a.H * a.W;
    const int nch = n / CH;
    const int wch = a.W / CH;
    const u32x4* p16 = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.pred) +
                                                      map * (long)n * (BF16 ? 2 : 4));
    u32x4* g16 = HAS_GRAD ? reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.grad) +
                                                     map * (long)n * (BF16 ? 2 : 4))
                          : nullptr;
    const uint8_t* mrow = HAS_MASK ? a.mask + b * (long)n : nullptr;
    // Pass 1: the gradient is zero almost everywhere.  These stores depend on nothing, so they
    // stream out while the joints that place the window are still being loaded.
    if (HAS_GRAD) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll 4
        for (int q = lane; q < nch; q += 64) __builtin_nontemporal_store(zero, g16 + q);
    }
    // Pass 2: the chunks that touch the window, one per lane (a window of radius 2 has about 15),
    // so their pred loads are all in flight together.  A window chunk is written by another lane
    // than the one that zeroed it: the zero stores are waited for first, which orders the two
    // writes of that address.  Lane sums are folded in a fixed order (bit-stable loss).
    float acc = 0.0f;
    if (w.on) {
        const int ylo = max(w.sy - a.radius, 0), yhi = min(w.sy + a.radius, a.H - 1);
        const int xlo = max(w.sx - a.radius, 0), xhi = min(w.sx + a.radius, a.W - 1);
        const int clo = xlo / CH, chi = xhi / CH;
        const int ncx = chi - clo + 1;
        const int nwin = (yhi >= ylo && xhi >= xlo) ? (yhi - ylo + 1) * ncx : 0;
        if (HAS_GRAD && nwin > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int j = lane; j < nwin; j += 64) {
            const int y = ylo + j / ncx;
            const int cx = clo + j % ncx;
            const int q = y * wch + cx;
            const int x0 = cx * CH;
            float p[CH], gr[CH];
            const u32x4 pv = __builtin_nontemporal_load(p16 + q);
            if constexpr (BF16) {
                const uint32_t u[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[2 * i] = bf16_bits_to_float(u[i] & 0xffffu);
                    p[2 * i + 1] = bf16_bits_to_float(u[i] >> 16);
                }
            } else {
                p[0] = __uint_as_float(pv.x);
                p[1] = __uint_as_float(pv.y);
                p[2] = __uint_as_float(pv.z);
                p[3] = __uint_as_float(pv.w);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                bool in;
                const float t = vec_value(w, axis, a.radius, y, x0 + i, in);
                if (HAS_MASK) in = in && mrow[(long)q * CH + i] != 0;
                const float d = p[i] - t;
                const float sq = d * d;
                acc += in ? sq : 0.0f;
                gr[i] = in ? a.inv_n * d : 0.0f;
            }
            if (HAS_GRAD) {
                u32x4 o;
                if constexpr (BF16) {
                    o.x = pack_bf16x2_bits(gr[0], gr[1]);
                    o.y = pack_bf16x2_bits(gr[2], gr[3]);
                    o.z = pack_bf16x2_bits(gr[4], gr[5]);
                    o.w = pack_bf16x2_bits(gr[6], gr[7]);
                } else {
                    o.x = __float_as_uint(gr[0]);
                    o.y = __float_as_uint(gr[1]);
                    o.z = __float_as_uint(gr[2]);
                    o.w = __float_as_uint(gr[3]);
                }
                __builtin_nontemporal_store(o, g16 + q);
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) a.partial[map] = acc;
}

static int32_t fill_pairs(const int32_t* pairs_host, int32_t num_pairs, int32_t K, PairTable* pt,
                          const char* what) {
    PE_REQUIRE(num_pairs > 0 && num_pairs <= kMaxPairs, PE_E_UNSUPPORTED,
               "%s: num_pairs %d outside [1, %d]", what, num_pairs, kMaxPairs);
    PE_REQUIRE(pairs_host != nullptr, PE_E_INVALID, "%s: NULL pairs", what);
    for (int i = 0; i < kMaxPairs; ++i) pt->k1[i] = pt->k2[i] = 0;
    for (int i = 0; i < num_pairs; ++i) {
        const int32_t a = pairs_host[2 * i], b = pairs_host[2 * i + 1];
        PE_REQUIRE(a >= 0 && a < K && b >= 0 && b < K, PE_E_INVALID,
                   "%s: pair %d = (%d, %d) names a joint outside [0, %d)", what, i, a, b, K);
        pt->k1[i] = (int16_t)a;
        pt->k2[i] = (int16_t)b;
    }
    return PE_OK;
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_render_vecmaps_f32(const float* joints, int32_t B, int32_t K,
                                         const int32_t* pairs_host, int32_t num_pairs, int32_t H,
                                         int32_t W, float stride, int32_t radius, float* vec,
                                         int64_t sB, int64_t sC, int64_t sH, int64_t sW,
                                         pe_stream_t stream) {
    const char* what = "pe_render_vecmaps_f32";
    PE_REQUIRE(B >= 0 && K > 0 && K < 32768, PE_E_INVALID, "%s: bad batch or joint count", what);
    PE_REQUIRE(H > 0 && W > 0, PE_E_INVALID, "%s: H and W must be positive", what);
    PE_REQUIRE(radius >= 0 && radius <= PE_MAX_RADIUS, PE_E_UNSUPPORTED,
               "%s: radius %d outside [0, %d]", what, radius, PE_MAX_RADIUS);
    PE_REQUIRE(stride > 0.0f && isfinite(stride), PE_E_INVALID,
               "%s: stride must be positive and finite", what);
    PairTable pt;
    const int32_t rc = fill_pairs(pairs_host, num_pairs, K, &pt, what);
    if (rc != PE_OK) return rc;
    const int C = 4 * num_pairs;
    const int64_t nmaps = (int64_t)B * C;
    PE_REQUIRE(nmaps < 0x7fffffff && (int64_t)H * W < 0x7fffffff &&
                   nmaps * H * W < ((int64_t)1 << 40),
               PE_E_UNSUPPORTED, "%s: tensor too large", what);
    if (nmaps == 0) return PE_OK;
    PE_REQUIRE(joints && vec, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(sB >= 0 && sC >= 0 && sH >= 0 && sW >= 0, PE_E_INVALID,
               "%s: negative strides are not supported", what);
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)H * W;
    const bool contig = (sW == 1 && sH == W && sC == n && sB == (int64_t)C * n);
    if (contig && (W % 4) == 0 && aligned(vec, 16)) {
        pe::launch(vec_render_contig_kernel, (unsigned)((nmaps + 3) / 4), 256, 0, st, joints, K, pt, C, (int)nmaps, H, W, stride, radius, vec);
        return check_launch("pe_render_vecmaps_f32(contiguous)");
    }
    int64_t blocks = (nmaps * n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    pe::launch(vec_render_strided_kernel, (unsigned)blocks, 256, 0, st, joints, B, K, pt, C, H, W, stride,
                                                                radius, vec, sB, sC, sH, sW,
                                                                sC < sW ? 1 : 0);
    return check_launch("pe_render_vecmaps_f32(strided)");
}

extern "C" size_t pe_vecmap_workspace_bytes(int32_t B, int32_t num_pairs) {
    if (B <= 0 || num_pairs <= 0) return 16;
    return (((size_t)B * 4 * (size_t)num_pairs * sizeof(float)) + 15) & ~(size_t)15;
}

extern "C" int32_t pe_vecmap_loss_fwd_bwd(const void* pred, int32_t pred_dtype, const float* joints,
                                          const uint8_t* mask, int32_t B, int32_t K,
                                          const int32_t* pairs_host, int32_t num_pairs, int32_t H,
                                          int32_t W, float stride, int32_t radius, float* loss,
                                          float* per_channel, void* grad, void* workspace,
                                          pe_stream_t stream) {
    const char* what = "pe_vecmap_loss_fwd_bwd";
    PE_REQUIRE(B > 0 && K > 0 && K < 32768 && H > 0 && W > 0, PE_E_INVALID,
               "%s: B, K, H, W must be positive", what);
    PE_REQUIRE(pred_dtype == PE_DTYPE_F32 || pred_dtype == PE_DTYPE_BF16, PE_E_INVALID,
               "%s: unknown pred_dtype %d", what, pred_dtype);
    PE_REQUIRE(pred && joints && loss && workspace, PE_E_INVALID,
               "%s: pred, joints, loss and workspace must be non-NULL", what);
    PE_REQUIRE(radius >= 0 && radius <= PE_MAX_RADIUS, PE_E_UNSUPPORTED,
               "%s: radius %d outside [0, %d]", what, radius, PE_MAX_RADIUS);
    PE_REQUIRE(stride > 0.0f && isfinite(stride), PE_E_INVALID,
               "%s: stride must be positive and finite", what);
    VecLossArgs a;
    const int32_t rc = fill_pairs(pairs_host, num_pairs, K, &a.pt, what);
    if (rc != PE_OK) return rc;
    const bool bf16 = pred_dtype == PE_DTYPE_BF16;
    const int ch = bf16 ? 8 : 4;
    PE_REQUIRE(W % ch == 0, PE_E_UNSUPPORTED, "%s: W must be a multiple of %d (W = %d)", what, ch, W);
    PE_REQUIRE(aligned(pred, 16) && aligned(grad, 16), PE_E_INVALID,
               "%s: pred/grad not 16-byte aligned", what);
    const int C = 4 * num_pairs;
    const int64_t nmaps = (int64_t)B * C;
    PE_REQUIRE(nmaps < 0x7fffffff && (int64_t)H * W < 0x7fffffff, PE_E_UNSUPPORTED,
               "%s: B*C or H*W does not fit int32", what);
    a.pred = pred;
    a.joints = joints;
    a.mask = mask;
    a.grad = grad;
    a.partial = static_cast<float*>(workspace);
    a.nmaps = (int)nmaps;
    a.K = K;
    a.C = C;
    a.H = H;
    a.W = W;
    a.radius = radius;
    a.stride = stride;
    const double n = (double)B * C * H * W;
    a.inv_n = (float)(1.0 / n);
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)((nmaps + 3) / 4);
    const bool hm = mask != nullptr, hg = grad != nullptr;
#define PE_VL(BF_, M_, G_) pe::launch(vec_loss_kernel<BF_, M_, G_>, grid, 256, 0, st, a)
    if (bf16) {
        if (hm && hg) PE_VL(true, true, true);
        else if (hm) PE_VL(true, true, false);
        else if (hg) PE_VL(true, false, true);
        else PE_VL(true, false, false);
    } else {
        if (hm && hg) PE_VL(false, true, true);
        else if (hm) PE_VL(false, true, false);
        else if (hg) PE_VL(false, false, true);
        else PE_VL(false, false, false);
    }
#undef PE_VL
    const int32_t rc1 = check_launch("pe_vecmap_loss_fwd_bwd(stage 1)");
    if (rc1 != PE_OK) return rc1;
    return launch_map_sum_reduce(a.partial, B, C, (float)(0.5 / ((double)B * H * W)),
                                 (float)(0.5 / n), per_channel, loss, st,
                                 "pe_vecmap_loss_fwd_bwd(stage 2)");
}
