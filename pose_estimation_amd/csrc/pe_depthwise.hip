// Depthwise 3x3 convolution (MobileNetV2's spatial filter), NHWC bf16, stride 1 or 2, any
// dilation, "same" padding (pad = dilation), with folded-BN scale/shift and ReLU / ReLU6.
//
// Bound: HBM bandwidth.  9 MACs per element against 4 bytes of traffic, so there is nothing
// for the matrix cores here: one thread owns 8 channels of one output pixel, every access is
// 16 bytes, the nine input rows of a pixel's neighbourhood come from L1/L2 after their first
// use, and the [9][C] weight block stays cache resident.  float32 accumulation in tap order.
#include "pe_common.h"

namespace pe {

__device__ __forceinline__ void dw_unpack(const uint4& v, float* f) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_bits_to_float(u[i] & 0xffffu);
        f[2 * i + 1] = bf16_bits_to_float(u[i] >> 16);
    }
}

__global__ __launch_bounds__(256) void depthwise3x3_kernel(const uint16_t* __restrict__ x, int B,
                                                           int H, int W, int C, int stride,
                                                           int dil, const uint16_t* __restrict__ w,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           int relu, uint16_t* __restrict__ y,
                                                           int Ho, int Wo) {
    const int C8 = C >> 3;
    const long total = (long)B * Ho * Wo * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const long b = r / Ho;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + (ky - 1) * dil;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + (kx - 1) * dil;
                if ((unsigned)ix >= (unsigned)W) continue;
                float xf[8], wf[8];
                dw_unpack(*reinterpret_cast<const uint4*>(x + (((b * H + iy) * W + ix) * (long)C8 + c8) * 8), xf);
                dw_unpack(*reinterpret_cast<const uint4*>(w + ((long)(ky * 3 + kx) * C8 + c8) * 8), wf);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += xf[i] * wf[i];
            }
        }
        if (scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c8 * 8);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(scale + c8 * 8 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] *= sc[i];
        }
        if (shift) {
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + c8 * 8);
            const f32x4 h1 = *reinterpret_cast<const f32x4*>(shift + c8 * 8 + 4);
            const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += sh[i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaxf(acc[i], 0.0f);
            if (relu == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fminf(acc[i], 6.0f);
            }
        }
        uint4 o;
        o.x = float_to_bf16_bits(acc[0]) | (float_to_bf16_bits(acc[1]) << 16);
        o.y = float_to_bf16_bits(acc[2]) | (float_to_bf16_bits(acc[3]) << 16);
        o.z = float_to_bf16_bits(acc[4]) | (float_to_bf16_bits(acc[5]) << 16);
        o.w = float_to_bf16_bits(acc[6]) | (float_to_bf16_bits(acc[7]) << 16);
        *reinterpret_cast<uint4*>(y + e * 8) = o;
    }
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_depthwise3x3_fwd_bf16(const void* x, int32_t B, int32_t H, int32_t W,
                                            int32_t C, int32_t stride, int32_t dilation,
                                            const void* w, const float* scale,
                                            const float* shift, int32_t relu, void* y,
                                            pe_stream_t stream) {
    const char* what = "pe_depthwise3x3_fwd_bf16";
    PE_REQUIRE(x && w && y, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, PE_E_UNSUPPORTED,
               "%s: sizes must be positive and C a multiple of 8", what);
    PE_REQUIRE((stride == 1 || stride == 2) && dilation >= 1 && dilation <= 64, PE_E_UNSUPPORTED,
               "%s: stride must be 1 or 2 and dilation in [1, 64]", what);
    PE_REQUIRE(relu >= 0 && relu <= 2, PE_E_INVALID, "%s: relu must be 0, 1 or 2", what);
    PE_REQUIRE(aligned(x, 16) && aligned(w, 16) && aligned(y, 16) && aligned(scale, 16) &&
                   aligned(shift, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    // "same" padding: pad = dilation, so Ho = floor((H - 1) / stride) + 1
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long total = (long)B * Ho * Wo * (C / 8);
    PE_REQUIRE((long)B * H * W * (C / 8) < (1L << 40), PE_E_UNSUPPORTED, "%s: tensor too large", what);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    depthwise3x3_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(
        static_cast<const uint16_t*>(x), B, H, W, C, stride, dilation,
        static_cast<const uint16_t*>(w), scale, shift, relu, static_cast<uint16_t*>(y), Ho, Wo);
    return check_launch(what);
}
