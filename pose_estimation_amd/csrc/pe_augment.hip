// GPU-side input pipeline in front of the stem: resize + scale jitter + crop + left-right flip +
// normalisation of uint8 frames, written straight into the bf16 space-to-depth image the stem
// kernels read, and the matching transform of the joint coordinates.
//
// The reference does this in its tf.data map function (dataset/pose_data.py,
// PoseEstimation.pyproj:27; absent); its configuration is printed at test.ipynb:995-1010:
// augmentation {flip_left_right, random_crop, scale_range [0.5, 2.0]}, preprocess
// {image_resize [320, 320]}.  Here one affine map per image carries all of it:
//     u = (xf + 0.5) * sx + tx - 0.5,   v = (y + 0.5) * sy + ty - 0.5,   xf = flip ? W-1-x : x
// maps output pixel (x, y) to source coordinates, sampled bilinearly (outside the frame = 0),
// then (value / 255 - mean) / std.  All arithmetic is float32 in a fixed order, so the result
// is bit-identical to the CPU restatement.
//
// Bound: HBM bandwidth (reads at most 4 x 3 bytes per output pixel from cache-friendly rows,
// writes 8 bytes per pixel).  One thread per 2x2 output block = one space-to-depth pixel.
#include <math.h>

#include "pe_common.h"

namespace pe {

struct AugNorm {
    float mean[3];
    float inv_std[3];
};

__device__ __forceinline__ float aug_tap(const uint8_t* __restrict__ img, int Hs, int Ws, int yy,
                                         int xx, int c) {
    const bool ok = (unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws;
    const long off = ok ? ((long)yy * Ws + xx) * 3 + c : 0;
    const float v = (float)img[off];
    return ok ? v : 0.0f;
}

// 8 bytes from any byte address (gfx950 global loads need no alignment).
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
    struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
    return reinterpret_cast<const U64*>(p)->v;
}

// value / 255 correctly rounded, in three instructions instead of the dozen of a general
// division: q = RN(v * y) with y = RN(1/255), the exact residual r = v - 255 q by one fused
// multiply-add, and q + r y rounded once more.  With y the correctly rounded reciprocal and
// 255's significand not all ones this is the correctly rounded quotient (Markstein), i.e. the
// same float as v / 255.0f; tests compare the kernel with NumPy's division bit for bit.
__device__ __forceinline__ float div255(float v) {
    const float y = 1.0f / 255.0f;
    const float q = v * y;
    const float r = __builtin_fmaf(-q, 255.0f, v);
    return __builtin_fmaf(r, y, q);
}

// A workgroup makes a tile of kTileX x kTileY space-to-depth pixels (2 kTileX x 2 kTileY output
// pixels) of one image.  The bilinear footprint of an output pixel separates into a column part
// (x0, fx), which depends on x alone, and a row part (y0, fy): the tile's 2 kTileX columns and
// 2 kTileY rows are worked out once, by the first threads, and shared through LDS, and the image
// index comes from the grid, so the per-image parameters are scalar loads and no thread divides.
constexpr int kTileX = 32, kTileY = 8;

__global__ __launch_bounds__(kTileX * kTileY) void augment_s2d_kernel(
    const uint8_t* __restrict__ img, int Hs, int Ws, const float* __restrict__ params, int H2,
    int W2, AugNorm nm, uint4* __restrict__ out) {
    __shared__ int cx0[2 * kTileX], ry0[2 * kTileY];
    __shared__ float cfx[2 * kTileX], rfy[2 * kTileY];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int W = 2 * W2;
    const float sx = params[b * 5 + 0], sy = params[b * 5 + 1];
    const float tx = params[b * 5 + 2], ty = params[b * 5 + 3];
    const bool flip = params[b * 5 + 4] != 0.0f;
    if (tid < 2 * kTileX) {
        const int x = 2 * kTileX * blockIdx.x + tid;
        const int xf = flip ? W - 1 - x : x;
        const float u = ((float)xf + 0.5f) * sx + tx - 0.5f;
        const float uf = floorf(u);
        cfx[tid] = u - uf;
        cx0[tid] = (int)fminf(fmaxf(uf, -1.0e9f), 1.0e9f);
    } else if (tid < 2 * kTileX + 2 * kTileY) {
        const int i = tid - 2 * kTileX;
        const int y = 2 * kTileY * blockIdx.y + i;
        const float v = ((float)y + 0.5f) * sy + ty - 0.5f;
        const float vf = floorf(v);
        rfy[i] = v - vf;
        ry0[i] = (int)fminf(fmaxf(vf, -1.0e9f), 1.0e9f);
    }
    __syncthreads();
    const int lx = tid % kTileX, ly = tid / kTileX;
    const int X = kTileX * blockIdx.x + lx, Y = kTileY * blockIdx.y + ly;
    if (X >= W2 || Y >= H2) return;
    const uint8_t* src = img + (long)b * Hs * Ws * 3;
    const long pitch = (long)Ws * 3;
    int x0[2], y0[2];
    float fx[2], fy[2];
    bool xin[2], yin[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        x0[r] = cx0[2 * lx + r];
        fx[r] = cfx[2 * lx + r];
        y0[r] = ry0[2 * ly + r];
        fy[r] = rfy[2 * ly + r];
        // Interior footprints (both taps in the frame, and the 8 bytes from the left tap on still
        // inside the row): the two taps of a row are 6 consecutive bytes, fetched by one
        // unaligned 8-byte load instead of six byte loads.  Same values, same arithmetic.
        xin[r] = x0[r] >= 0 && x0[r] + 3 <= Ws && (long)x0[r] * 3 + 8 <= pitch;
        yin[r] = y0[r] >= 0 && y0[r] + 1 < Hs;
    }
    const bool all_inner = xin[0] && xin[1] && yin[0] && yin[1];
    const long t = ((long)b * H2 + Y) * W2 + X;
    // A zoomed-out or shifted map leaves whole regions of the output outside the frame: all
    // sixteen taps are zero there and the block is a constant.  (0 + f*(0 - 0) = +0 whatever f
    // is, so this is the value the general path computes.)
    const bool rows_out = (y0[0] + 1 < 0 || y0[0] >= Hs) && (y0[1] + 1 < 0 || y0[1] >= Hs);
    const bool cols_out = (x0[0] + 1 < 0 || x0[0] >= Ws) && (x0[1] + 1 < 0 || x0[1] >= Ws);
    if (rows_out || cols_out) {
        float k[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) k[c] = (div255(0.0f) - nm.mean[c]) * nm.inv_std[c];
        u32x4 lo = {pack_bf16x2_bits(k[0], k[1]), pack_bf16x2_bits(k[2], k[0]),
                    pack_bf16x2_bits(k[1], k[2]), pack_bf16x2_bits(k[0], k[1])};
        u32x4 hi = {pack_bf16x2_bits(k[2], k[0]), pack_bf16x2_bits(k[1], k[2]), 0u, 0u};
        __builtin_nontemporal_store(lo, reinterpret_cast<u32x4*>(out) + 2 * t);
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x4*>(out) + 2 * t + 1);
        return;
    }
    if (!all_inner) {
        // Frame borders and whatever the map throws outside the frame: tap by tap, one value at a
        // time (rolled loops keep this rare path from costing the common one its registers).
        uint16_t* o = reinterpret_cast<uint16_t*>(out + 2 * t);
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int ry = i >> 1, rx = i & 1;
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                const float p00 = aug_tap(src, Hs, Ws, y0[ry], x0[rx], c);
                const float p01 = aug_tap(src, Hs, Ws, y0[ry], x0[rx] + 1, c);
                const float p10 = aug_tap(src, Hs, Ws, y0[ry] + 1, x0[rx], c);
                const float p11 = aug_tap(src, Hs, Ws, y0[ry] + 1, x0[rx] + 1, c);
                const float top = p00 + fx[rx] * (p01 - p00);
                const float bot = p10 + fx[rx] * (p11 - p10);
                const float val = top + fy[ry] * (bot - top);
                o[i * 3 + c] = (uint16_t)float_to_bf16_bits((div255(val) - nm.mean[c]) * nm.inv_std[c]);
            }
        }
        *reinterpret_cast<uint2*>(o + 12) = make_uint2(0u, 0u);
        return;
    }
    // the common case: eight independent loads in flight per lane
    uint32_t a[4][4];    // per pixel: low and high words of the upper and of the lower row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint8_t* r0 = src + (long)y0[i >> 1] * pitch + (long)x0[i & 1] * 3;
        const uint64_t u0 = load_u64_unaligned(r0);
        const uint64_t u1 = load_u64_unaligned(r0 + pitch);
        a[i][0] = (uint32_t)u0;
        a[i][1] = (uint32_t)(u0 >> 32);
        a[i][2] = (uint32_t)u1;
        a[i][3] = (uint32_t)(u1 >> 32);
    }
    float h[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ry = i >> 1, rx = i & 1;
        float p00[3], p01[3], p10[3], p11[3];
        // bytes 0..2 = left tap, 3..5 = right tap (byte 3 in the low word, 4 and 5 in the high)
        p00[0] = (float)(a[i][0] & 0xffu);
        p00[1] = (float)((a[i][0] >> 8) & 0xffu);
        p00[2] = (float)((a[i][0] >> 16) & 0xffu);
        p01[0] = (float)(a[i][0] >> 24);
        p01[1] = (float)(a[i][1] & 0xffu);
        p01[2] = (float)((a[i][1] >> 8) & 0xffu);
        p10[0] = (float)(a[i][2] & 0xffu);
        p10[1] = (float)((a[i][2] >> 8) & 0xffu);
        p10[2] = (float)((a[i][2] >> 16) & 0xffu);
        p11[0] = (float)(a[i][2] >> 24);
        p11[1] = (float)(a[i][3] & 0xffu);
        p11[2] = (float)((a[i][3] >> 8) & 0xffu);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = p00[c] + fx[rx] * (p01[c] - p00[c]);
            const float bot = p10[c] + fx[rx] * (p11[c] - p10[c]);
            const float val = top + fy[ry] * (bot - top);
            h[i * 3 + c] = (div255(val) - nm.mean[c]) * nm.inv_std[c];
        }
    }
    uint4 lo, hi;
    lo.x = pack_bf16x2_bits(h[0], h[1]);
    lo.y = pack_bf16x2_bits(h[2], h[3]);
    lo.z = pack_bf16x2_bits(h[4], h[5]);
    lo.w = pack_bf16x2_bits(h[6], h[7]);
    hi.x = pack_bf16x2_bits(h[8], h[9]);
    hi.y = pack_bf16x2_bits(h[10], h[11]);
    hi.z = 0u;
    hi.w = 0u;
    __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&lo), reinterpret_cast<u32x4*>(out) + 2 * t);
    __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&hi),
                                reinterpret_cast<u32x4*>(out) + 2 * t + 1);
}

struct JointPerm {
    int16_t to[64];
};

// Inverse of the image map for coordinates: xo = (x + 0.5 - tx) / sx - 0.5, mirrored when the
// image was flipped, and with left/right joints exchanged through `perm` in that case.
__global__ __launch_bounds__(256) void transform_joints_kernel(const float* __restrict__ jin, int B,
                                                               int K, const float* __restrict__ params,
                                                               int W, JointPerm perm,
                                                               float* __restrict__ jout) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * K) return;
    const int b = t / K, k = t - b * K;
    const float sx = params[b * 5 + 0], sy = params[b * 5 + 1];
    const float tx = params[b * 5 + 2], ty = params[b * 5 + 3];
    const bool flip = params[b * 5 + 4] != 0.0f;
    const int ks = flip ? perm.to[k] : k;   // output joint k takes source joint ks
    const float* s = jin + ((long)b * K + ks) * 3;
    float xo = (s[0] + 0.5f - tx) / sx - 0.5f;
    const float yo = (s[1] + 0.5f - ty) / sy - 0.5f;
    if (flip) xo = (float)(W - 1) - xo;
    float* d = jout + ((long)b * K + k) * 3;
    d[0] = xo;
    d[1] = yo;
    d[2] = s[2];
}

// The ignore mask follows the image map: output mask cell (x, y) of size `stride` pixels takes
// the value of the source mask cell (size src_stride pixels of the source frame) that its centre
// maps to, nearest sampling; centres that leave the source frame take `fill`.
__global__ __launch_bounds__(256) void augment_mask_kernel(const uint8_t* __restrict__ min, int B,
                                                           int Hms, int Wms, float src_stride,
                                                           const float* __restrict__ params,
                                                           int Hm, int Wm, float stride, int fill,
                                                           uint8_t* __restrict__ mout) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * Hm * Wm;
    if (t >= total) return;
    const int x = (int)(t % Wm);
    const long r = t / Wm;
    const int y = (int)(r % Hm);
    const long b = r / Hm;
    const float sx = params[b * 5 + 0], sy = params[b * 5 + 1];
    const float tx = params[b * 5 + 2], ty = params[b * 5 + 3];
    const bool flip = params[b * 5 + 4] != 0.0f;
    const float W = (float)Wm * stride;
    // centre of the cell in output pixel index coordinates, then the image map
    const float xc = ((float)x + 0.5f) * stride - 0.5f;
    const float yc = ((float)y + 0.5f) * stride - 0.5f;
    const float xf = flip ? (W - 1.0f) - xc : xc;
    const float u = (xf + 0.5f) * sx + tx - 0.5f;
    const float v = (yc + 0.5f) * sy + ty - 0.5f;
    const float cu = floorf((u + 0.5f) / src_stride);
    const float cv = floorf((v + 0.5f) / src_stride);
    int val = fill;
    if (cu >= 0.0f && cu < (float)Wms && cv >= 0.0f && cv < (float)Hms)
        val = min[(b * Hms + (long)cv) * Wms + (long)cu];
    mout[t] = (uint8_t)val;
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_augment_mask_u8(const uint8_t* mask_in, int32_t B, int32_t Hms, int32_t Wms,
                                      float src_stride, const float* params_device, int32_t Hm,
                                      int32_t Wm, float stride, int32_t fill, uint8_t* mask_out,
                                      pe_stream_t stream) {
    const char* what = "pe_augment_mask_u8";
    PE_REQUIRE(mask_in && params_device && mask_out, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && Hms > 0 && Wms > 0 && Hm > 0 && Wm > 0, PE_E_INVALID,
               "%s: sizes must be positive", what);
    PE_REQUIRE(src_stride > 0.0f && stride > 0.0f && isfinite(src_stride) && isfinite(stride),
               PE_E_INVALID, "%s: strides must be positive and finite", what);
    PE_REQUIRE(fill >= 0 && fill <= 255, PE_E_INVALID, "%s: fill must be a byte", what);
    PE_REQUIRE(mask_in != mask_out, PE_E_INVALID, "%s: in-place warp is not supported", what);
    const long total = (long)B * Hm * Wm;
    PE_REQUIRE(total < 0x7fffffffL && (long)B * Hms * Wms < 0x7fffffffL, PE_E_UNSUPPORTED,
               "%s: too many cells", what);
    pe::launch(augment_mask_kernel, (unsigned)((total + 255) / 256), 256, 0, as_stream(stream), mask_in, B, Hms, Wms, src_stride, params_device, Hm, Wm, stride, fill, mask_out);
    return check_launch(what);
}

extern "C" int32_t pe_augment_u8_s2d_bf16(const uint8_t* img, int32_t B, int32_t Hs, int32_t Ws,
                                          const float* params_device, int32_t H, int32_t W,
                                          const float* mean3, const float* std3, void* x_s2d,
                                          pe_stream_t stream) {
    const char* what = "pe_augment_u8_s2d_bf16";
    PE_REQUIRE(img && params_device && mean3 && std3 && x_s2d, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0,
               PE_E_INVALID, "%s: sizes must be positive and the output size even", what);
    PE_REQUIRE(aligned(x_s2d, 16), PE_E_INVALID, "%s: output misaligned", what);
    const long npix2 = (long)B * (H / 2) * (W / 2);
    PE_REQUIRE(npix2 < 0x7fffffffL && (long)Hs * Ws * 3 < 0x7fffffffL, PE_E_UNSUPPORTED,
               "%s: too many pixels", what);
    AugNorm nm;
    for (int c = 0; c < 3; ++c) {
        PE_REQUIRE(std3[c] > 0.0f, PE_E_INVALID, "%s: std must be positive", what);
        nm.mean[c] = mean3[c];
        nm.inv_std[c] = 1.0f / std3[c];
    }
    PE_REQUIRE(B <= 65535, PE_E_UNSUPPORTED, "%s: at most 65535 images per call", what);
    const dim3 grid((unsigned)((W / 2 + kTileX - 1) / kTileX), (unsigned)((H / 2 + kTileY - 1) / kTileY),
                    (unsigned)B);
    pe::launch(augment_s2d_kernel, grid, kTileX * kTileY, 0, as_stream(stream), img, Hs, Ws,
               params_device, H / 2, W / 2, nm, static_cast<uint4*>(x_s2d));
    return check_launch(what);
}

extern "C" int32_t pe_transform_joints_f32(const float* joints_in, int32_t B, int32_t K,
                                           const float* params_device, int32_t W,
                                           const int32_t* flip_perm_host, float* joints_out,
                                           pe_stream_t stream) {
    const char* what = "pe_transform_joints_f32";
    PE_REQUIRE(joints_in && params_device && joints_out, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && K > 0 && K <= 64 && W > 0, PE_E_UNSUPPORTED,
               "%s: need B > 0, 0 < K <= 64, W > 0", what);
    PE_REQUIRE(joints_in != joints_out, PE_E_INVALID, "%s: in-place transform is not supported", what);
    JointPerm perm;
    for (int k = 0; k < 64; ++k) perm.to[k] = (int16_t)k;
    if (flip_perm_host) {
        for (int k = 0; k < K; ++k) {
            PE_REQUIRE(flip_perm_host[k] >= 0 && flip_perm_host[k] < K, PE_E_INVALID,
                       "%s: flip_perm[%d] = %d out of range", what, k, flip_perm_host[k]);
            perm.to[k] = (int16_t)flip_perm_host[k];
        }
    }
    pe::launch(transform_joints_kernel, (unsigned)((B * K + 255) / 256), 256, 0, as_stream(stream), joints_in, B, K, params_device, W, perm, joints_out);
    return check_launch(what);
}
