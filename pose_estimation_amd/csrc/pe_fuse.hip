// Multi-resolution fusion for high-resolution (HRNet-style) backbones:
//   out = relu?( base + sum_i nearest_upsample(low_i, 2^shift_i) )
// over NHWC bf16 maps, summed in float32 and rounded once.  The reference's config names no
// such backbone (its network is nets/mobilenet_v2.py, PoseEstimation.pyproj:32); this serves the
// 384x288 multi-branch configuration.
//
// Bound: HBM bandwidth.  Per output pixel and channel: 2 B read (base) + 2 B written; the
// low-resolution maps add 2 B / 4^shift and are re-read from cache by neighbouring pixels.
// One thread per 8 channels (16-byte loads and stores).
#include "pe_common.h"

namespace pe {

struct FuseArgs {
    const uint16_t* base;    // [B,H,W,C] or null
    const uint16_t* low[3];  // [B, H >> shift, W >> shift, C]
    int shift[3];
    int nlow;
};

__device__ __forceinline__ void add_bf16x8(float* acc, const uint4& v) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += bf16_bits_to_float(u[i] & 0xffffu);
        acc[2 * i + 1] += bf16_bits_to_float(u[i] >> 16);
    }
}

__global__ __launch_bounds__(256) void fuse_upsample_add_kernel(FuseArgs a, int B, int H, int W,
                                                                int C8, int relu,
                                                                uint4* __restrict__ out) {
    const long total = (long)B * H * W * C8;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    const long pix = t / C8;
    const int x = (int)(pix % W);
    const long r = pix / W;
    const int y = (int)(r % H);
    const long b = r / H;
    // all loads first, then the sum in a fixed order: base, low[0], low[1], low[2]
    uint4 vb = make_uint4(0u, 0u, 0u, 0u), vl[3];
    if (a.base) vb = reinterpret_cast<const uint4*>(a.base)[t];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vl[i] = make_uint4(0u, 0u, 0u, 0u);
        if (i < a.nlow) {
            const int s = a.shift[i];
            const long lp = (b * (H >> s) + (y >> s)) * (W >> s) + (x >> s);
            vl[i] = reinterpret_cast<const uint4*>(a.low[i])[lp * C8 + c8];
        }
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    add_bf16x8(acc, vb);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i < a.nlow) add_bf16x8(acc, vl[i]);
    if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaxf(acc[i], 0.0f);
    }
    uint4 o;
    o.x = pack_bf16x2_bits(acc[0], acc[1]);
    o.y = pack_bf16x2_bits(acc[2], acc[3]);
    o.z = pack_bf16x2_bits(acc[4], acc[5]);
    o.w = pack_bf16x2_bits(acc[6], acc[7]);
    out[t] = o;
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_fuse_upsample_add_bf16(const void* base, int32_t n_low,
                                             const void* const* low_host,
                                             const int32_t* log2_factor_host, int32_t B, int32_t H,
                                             int32_t W, int32_t C, int32_t relu, void* y,
                                             pe_stream_t stream) {
    const char* what = "pe_fuse_upsample_add_bf16";
    PE_REQUIRE(y != nullptr && (base != nullptr || n_low > 0), PE_E_INVALID, "%s: nothing to add", what);
    PE_REQUIRE(n_low >= 0 && n_low <= 3, PE_E_UNSUPPORTED, "%s: at most 3 low-resolution maps", what);
    PE_REQUIRE(n_low == 0 || (low_host && log2_factor_host), PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, PE_E_INVALID,
               "%s: sizes must be positive and C a multiple of 8", what);
    PE_REQUIRE((long)B * H * W * (C / 8) < (1L << 40), PE_E_UNSUPPORTED, "%s: too large", what);
    PE_REQUIRE(aligned(base, 16) && aligned(y, 16), PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    FuseArgs a{};
    a.base = static_cast<const uint16_t*>(base);
    a.nlow = n_low;
    for (int i = 0; i < n_low; ++i) {
        const int s = log2_factor_host[i];
        PE_REQUIRE(low_host[i] && aligned(low_host[i], 16), PE_E_INVALID,
                   "%s: low-resolution map %d is NULL or misaligned", what, i);
        PE_REQUIRE(s >= 1 && s <= 5 && H % (1 << s) == 0 && W % (1 << s) == 0, PE_E_INVALID,
                   "%s: map size %dx%d is not a multiple of the upsampling factor 2^%d", what, H, W, s);
        a.low[i] = static_cast<const uint16_t*>(low_host[i]);
        a.shift[i] = s;
    }
    const long total = (long)B * H * W * (C / 8);
    pe::launch(fuse_upsample_add_kernel, (unsigned)((total + 255) / 256), 256, 0, as_stream(stream), a, B, H, W, C / 8, relu, static_cast<uint4*>(y));
    return check_launch(what);
}

// ---------------------------------------------------------------------------------------------
// Backward of the fusion: g = dy masked by (y > 0) (no mask without ReLU) is the gradient of
// `base`; the gradient of low_i is the sum of g over each 2^shift x 2^shift block, added in
// float32 in row-major block order and rounded once.  Both kernels are HBM-bound: one thread
// per 8 channels of one output pixel (mask) or of one low-resolution pixel (block sum).
// ---------------------------------------------------------------------------------------------
namespace pe {

__device__ __forceinline__ void masked_bf16x8(float* g, const uint4& dv, const uint4& yv, bool mask) {
    const uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w};
    const uint32_t y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d0 = bf16_bits_to_float(d[i] & 0xffffu), d1 = bf16_bits_to_float(d[i] >> 16);
        const bool k0 = !mask || bf16_bits_to_float(y[i] & 0xffffu) > 0.0f;
        const bool k1 = !mask || bf16_bits_to_float(y[i] >> 16) > 0.0f;
        g[2 * i] = k0 ? d0 : 0.0f;
        g[2 * i + 1] = k1 ? d1 : 0.0f;
    }
}

__global__ __launch_bounds__(256) void relu_mask_kernel(const uint4* __restrict__ dy,
                                                        const uint4* __restrict__ y, long n,
                                                        uint4* __restrict__ g) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v[8];
        masked_bf16x8(v, dy[i], y[i], true);
        uint4 o;
        o.x = pack_bf16x2_bits(v[0], v[1]);
        o.y = pack_bf16x2_bits(v[2], v[3]);
        o.z = pack_bf16x2_bits(v[4], v[5]);
        o.w = pack_bf16x2_bits(v[6], v[7]);
        g[i] = o;
    }
}

__global__ __launch_bounds__(256) void block_sum_kernel(const uint4* __restrict__ dy,
                                                        const uint4* __restrict__ y, int B, int H,
                                                        int W, int C8, int shift,
                                                        uint4* __restrict__ dlow) {
    const int h = H >> shift, w = W >> shift, f = 1 << shift;
    const long total = (long)B * h * w * C8;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    const long pix = t / C8;
    const int lx = (int)(pix % w);
    const long r = pix / w;
    const int ly = (int)(r % h);
    const long b = r / h;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dyy = 0; dyy < f; ++dyy) {
        const long row = ((b * H + (ly << shift) + dyy) * W + (lx << shift)) * C8 + c8;
        for (int dxx = 0; dxx < f; ++dxx) {
            const long e = row + (long)dxx * C8;
            float v[8];
            masked_bf16x8(v, dy[e], y ? y[e] : dy[e], y != nullptr);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
    uint4 o;
    o.x = pack_bf16x2_bits(acc[0], acc[1]);
    o.y = pack_bf16x2_bits(acc[2], acc[3]);
    o.z = pack_bf16x2_bits(acc[4], acc[5]);
    o.w = pack_bf16x2_bits(acc[6], acc[7]);
    dlow[t] = o;
}

__global__ __launch_bounds__(256) void add_bf16_kernel(const uint4* __restrict__ a,
                                                       const uint4* __restrict__ b, long n,
                                                       uint4* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        add_bf16x8(v, a[i]);
        add_bf16x8(v, b[i]);
        uint4 o;
        o.x = pack_bf16x2_bits(v[0], v[1]);
        o.y = pack_bf16x2_bits(v[2], v[3]);
        o.z = pack_bf16x2_bits(v[4], v[5]);
        o.w = pack_bf16x2_bits(v[6], v[7]);
        out[i] = o;
    }
}

static unsigned stream_grid(long n) {
    long b = (n + 255) / 256;
    if (b > 256 * 128) b = 256 * 128;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace pe

extern "C" int32_t pe_fuse_upsample_add_bwd_bf16(const void* dy, const void* y, int32_t n_low,
                                                 void* const* dlow_host,
                                                 const int32_t* log2_factor_host, int32_t B,
                                                 int32_t H, int32_t W, int32_t C, void* dbase,
                                                 pe_stream_t stream) {
    const char* what = "pe_fuse_upsample_add_bwd_bf16";
    PE_REQUIRE(dy != nullptr, PE_E_INVALID, "%s: NULL dy", what);
    PE_REQUIRE(n_low >= 0 && n_low <= 3, PE_E_UNSUPPORTED, "%s: at most 3 low-resolution maps", what);
    PE_REQUIRE(n_low == 0 || (dlow_host && log2_factor_host), PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, PE_E_INVALID,
               "%s: sizes must be positive and C a multiple of 8", what);
    PE_REQUIRE((long)B * H * W * (C / 8) < (1L << 40), PE_E_UNSUPPORTED, "%s: too large", what);
    PE_REQUIRE(aligned(dy, 16) && aligned(y, 16) && aligned(dbase, 16), PE_E_INVALID,
               "%s: tensors must be 16-byte aligned", what);
    PE_REQUIRE(dbase == nullptr || y != nullptr, PE_E_INVALID,
               "%s: without a ReLU the gradient of base is dy itself; pass dbase = NULL", what);
    hipStream_t st = as_stream(stream);
    const int C8 = C / 8;
    for (int i = 0; i < n_low; ++i) {
        const int s = log2_factor_host[i];
        PE_REQUIRE(dlow_host[i] && aligned(dlow_host[i], 16), PE_E_INVALID,
                   "%s: gradient buffer %d is NULL or misaligned", what, i);
        PE_REQUIRE(s >= 1 && s <= 5 && H % (1 << s) == 0 && W % (1 << s) == 0, PE_E_INVALID,
                   "%s: map size %dx%d is not a multiple of the upsampling factor 2^%d", what, H, W, s);
        const long total = (long)B * (H >> s) * (W >> s) * C8;
        pe::launch(block_sum_kernel, (unsigned)((total + 255) / 256), 256, 0, st, static_cast<const uint4*>(dy), static_cast<const uint4*>(y), B, H, W, C8, s,
            static_cast<uint4*>(dlow_host[i]));
        const int32_t rc = check_launch(what);
        if (rc != PE_OK) return rc;
    }
    if (dbase) {
        const long n = (long)B * H * W * C8;
        pe::launch(relu_mask_kernel, stream_grid(n), 256, 0, st, static_cast<const uint4*>(dy),
                                                        static_cast<const uint4*>(y), n,
                                                        static_cast<uint4*>(dbase));
        return check_launch(what);
    }
    return PE_OK;
}

extern "C" int32_t pe_add_bf16(const void* a, const void* b, int64_t n, void* out,
                               pe_stream_t stream) {
    PE_REQUIRE(a && b && out && n > 0 && n % 8 == 0, PE_E_INVALID,
               "pe_add_bf16: NULL pointer or element count not a positive multiple of 8");
    PE_REQUIRE(aligned(a, 16) && aligned(b, 16) && aligned(out, 16), PE_E_INVALID,
               "pe_add_bf16: tensors must be 16-byte aligned");
    pe::launch(add_bf16_kernel, stream_grid(n / 8), 256, 0, as_stream(stream), static_cast<const uint4*>(a), static_cast<const uint4*>(b), n / 8, static_cast<uint4*>(out));
    return check_launch("pe_add_bf16");
}
