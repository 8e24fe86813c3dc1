// a4 / a5 forward entry points: they describe each layer to the implicit-GEMM engine
// (pe_igemm.hip) and hold the two bandwidth-bound companions of the stem (uint8 -> bf16
// space-to-depth preprocessing, 3x3/2 max pool).
#include "pe_common.h"
#include "pe_igemm.h"

namespace pe {

static void base_params(IgemmParams& p) { igemm_defaults(p); }

// ---------------------------------------------------------------------------------------------
// uint8 RGB -> normalised bf16, 2x2 space-to-depth, 16 channels per pixel: (ry, rx, c) in the
// first 12, zeros in the last 4.  One thread per output pixel: reads 2 x 6 bytes, writes 32 B.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_s2d_kernel(const uint8_t* __restrict__ img,
                                                             long npix2, int H2, int W2,
                                                             float m0, float m1, float m2,
                                                             float i0, float i1, float i2,
                                                             uint4* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= npix2) return;
    const int X = (int)(t % W2);
    const long r = t / W2;
    const int Y = (int)(r % H2);
    const long b = r / H2;
    const int W = W2 * 2;
    const float mean[3] = {m0, m1, m2};
    const float inv[3] = {i0, i1, i2};
    uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint16_t h[16];
#pragma unroll
    for (int i = 12; i < 16; ++i) h[i] = 0;
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
        const uint8_t* row = img + ((b * (2L * H2) + 2 * Y + ry) * W + 2 * X) * 3;
#pragma unroll
        for (int rx = 0; rx < 2; ++rx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ((float)row[rx * 3 + c] / 255.0f - mean[c]) * inv[c];
                h[(ry * 2 + rx) * 3 + c] = (uint16_t)float_to_bf16_bits(v);
            }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = h[2 * i] | ((uint32_t)h[2 * i + 1] << 16);
    out[2 * t] = make_uint4(o[0], o[1], o[2], o[3]);
    out[2 * t + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// 3x3 stride-2 pad-1 max pool over NHWC bf16; one thread per 8 output channels.  With `idx`
// it also records, per output element, the window position ky*3+kx of the FIRST maximum in
// scan order, which is all the backward pass needs.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const uint16_t* __restrict__ x, int B,
                                                           int H, int W, int C,
                                                           uint16_t* __restrict__ y,
                                                           uint8_t* __restrict__ idx) {
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long total = (long)B * Ho * Wo * C8;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    long r = t / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    float best[8];
    uint32_t pos[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        best[i] = -INFINITY;
        pos[i] = 255u;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(x + ((b * H + iy) * W + ix) * C + c8 * 8);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[2 * i] = bf16_bits_to_float(u[i] & 0xffffu);
                f[2 * i + 1] = bf16_bits_to_float(u[i] >> 16);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // strictly greater keeps the first maximum; the first valid cell always enters
                if (f[i] > best[i] || pos[i] == 255u) pos[i] = (uint32_t)(ky * 3 + kx);
                best[i] = fmaxf(best[i], f[i]);
            }
        }
    }
    if (idx) {
        uint2 pk;
        pk.x = pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24);
        pk.y = pos[4] | (pos[5] << 8) | (pos[6] << 16) | (pos[7] << 24);
        *reinterpret_cast<uint2*>(idx + ((b * Ho + oy) * Wo + ox) * (long)C + c8 * 8) = pk;
    }
    uint4 o;
    o.x = pack_bf16x2_bits(best[0], best[1]);
    o.y = pack_bf16x2_bits(best[2], best[3]);
    o.z = pack_bf16x2_bits(best[4], best[5]);
    o.w = pack_bf16x2_bits(best[6], best[7]);
    *reinterpret_cast<uint4*>(y + ((b * Ho + oy) * Wo + ox) * (long)C + c8 * 8) = o;
}

// Request for the fused batch-norm statistics of a forward conv (training).
struct StatReq {
    void* workspace;
    size_t bytes;
    int32_t* rows_out;
};

static int32_t attach_stats(IgemmParams& p, const StatReq* sr, const char* what) {
    if (sr == nullptr) return PE_OK;
    PE_REQUIRE(sr->workspace && sr->rows_out, PE_E_INVALID, "%s: NULL statistics workspace", what);
    PE_REQUIRE(aligned(sr->workspace, 16) && p.N % 8 == 0, PE_E_INVALID,
               "%s: statistics need an aligned workspace and N %% 8 == 0", what);
    p.stat_partial = static_cast<float*>(sr->workspace);
    p.stat_capacity_rows = (long)(sr->bytes / ((size_t)2 * p.N * sizeof(float)));
    p.stat_rows_out = sr->rows_out;
    return PE_OK;
}

}  // namespace pe

using namespace pe;

// The engine parameters of one forward convolution (no statistics attached).
static int32_t conv2d_fwd_params(const pe_conv_desc* d, const void* x, const void* w,
                                 const float* scale, const float* shift, const void* residual,
                                 void* y, IgemmParams& p) {
    PE_REQUIRE(d != nullptr, PE_E_INVALID, "pe_conv2d_fwd_bf16: NULL descriptor");
    PE_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= kMaxTaps, PE_E_UNSUPPORTED,
               "pe_conv2d_fwd_bf16: filter %dx%d not supported", d->KH, d->KW);
    PE_REQUIRE(d->stride >= 1 && d->pad >= 0, PE_E_INVALID, "pe_conv2d_fwd_bf16: bad stride/pad");
    PE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, PE_E_INVALID,
               "pe_conv2d_fwd_bf16: sizes must be positive");
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    PE_REQUIRE(Ho > 0 && Wo > 0, PE_E_INVALID, "pe_conv2d_fwd_bf16: empty output");
    PE_REQUIRE(!(d->out_f32_nchw && residual), PE_E_UNSUPPORTED,
               "pe_conv2d_fwd_bf16: residual with float32 NCHW output is not supported");
    base_params(p);
    p.x = static_cast<const uint16_t*>(x);
    p.w = static_cast<const uint16_t*>(w);
    p.y = y;
    p.scale = scale;
    p.shift = shift;
    p.res = static_cast<const uint16_t*>(residual);
    p.B = d->B;
    p.Hi = d->H;
    p.Wi = d->W;
    p.Cx = d->Cin;
    p.Cin = d->Cin;
    p.Ho = Ho;
    p.Wo = Wo;
    p.sy = p.sx = d->stride;
    p.N = d->Cout;
    if (d->Cin == 8 || d->Cin == 16 || d->Cin == 32) {
        // Narrow inputs: one 64-element K run covers P = 64/Cin horizontally adjacent pixels
        // (contiguous in NHWC), so a filter row of KW taps becomes ceil(KW/P) runs.  Weights
        // arrive packed to match, [Cout][KH][ceil(KW/P)][P*Cin], zero where a run reaches past
        // the filter; each 8-channel load checks its own pixel against the image border.
        const int P = 64 / d->Cin;
        const int KWc = (d->KW + P - 1) / P;
        PE_REQUIRE(d->KH * KWc <= kMaxTaps, PE_E_UNSUPPORTED,
                   "pe_conv2d_fwd_bf16: filter %dx%d not supported", d->KH, d->KW);
        p.Cin = 64;
        p.cshift = d->Cin == 8 ? 3 : (d->Cin == 16 ? 4 : 5);
        p.cmask = d->Cin - 1;
        p.T = d->KH * KWc;
        for (int kh = 0; kh < d->KH; ++kh)
            for (int c = 0; c < KWc; ++c) {
                p.tap_dy[kh * KWc + c] = (int16_t)(kh - d->pad);
                p.tap_dx[kh * KWc + c] = (int16_t)(c * P - d->pad);
            }
    } else {
        p.T = d->KH * d->KW;
        for (int kh = 0; kh < d->KH; ++kh)
            for (int kw = 0; kw < d->KW; ++kw) {
                p.tap_dy[kh * d->KW + kw] = (int16_t)(kh - d->pad);
                p.tap_dx[kh * d->KW + kw] = (int16_t)(kw - d->pad);
            }
    }
    p.OH = Ho;
    p.OW = Wo;
    p.ldc = d->Cout;
    p.relu = d->relu;
    p.out_mode = d->out_f32_nchw ? 1 : 0;
    return PE_OK;
}

static int32_t conv2d_fwd_impl(const pe_conv_desc* d, const void* x, const void* w,
                               const float* scale, const float* shift, const void* residual,
                               void* y, const StatReq* sr, pe_stream_t stream) {
    IgemmParams p;
    int32_t rc = conv2d_fwd_params(d, x, w, scale, shift, residual, y, p);
    if (rc != PE_OK) return rc;
    rc = attach_stats(p, sr, "pe_conv2d_fwd_bf16");
    if (rc != PE_OK) return rc;
    return igemm_launch(p, as_stream(stream), "pe_conv2d_fwd_bf16");
}

extern "C" int32_t pe_conv2d_fwd_chain_bf16(const void* x, const pe_conv_stage* stages,
                                            int32_t nstage, pe_stream_t stream) {
    PE_REQUIRE(x != nullptr && stages != nullptr && nstage >= 1 && nstage <= kMaxChain,
               PE_E_INVALID, "pe_conv2d_fwd_chain_bf16: 1 to %d stages", kMaxChain);
    IgemmParams p[kMaxChain];
    const void* in = x;
    for (int s = 0; s < nstage; ++s) {
        const pe_conv_stage& g = stages[s];
        PE_REQUIRE(g.w != nullptr && g.y != nullptr, PE_E_INVALID,
                   "pe_conv2d_fwd_chain_bf16: stage %d: NULL pointer", s);
        const int32_t rc = conv2d_fwd_params(&g.desc, in, g.w, g.scale, g.shift, g.residual, g.y,
                                             p[s]);
        if (rc != PE_OK) return rc;
        in = g.y;
    }
    return igemm_chain_launch(p, nstage, as_stream(stream), "pe_conv2d_fwd_chain_bf16");
}

extern "C" int32_t pe_conv2d_fwd_bf16(const pe_conv_desc* d, const void* x, const void* w,
                                      const float* scale, const float* shift,
                                      const void* residual, void* y, pe_stream_t stream) {
    return conv2d_fwd_impl(d, x, w, scale, shift, residual, y, nullptr, stream);
}

extern "C" int32_t pe_conv2d_fwd_bnstats_bf16(const pe_conv_desc* d, const void* x, const void* w,
                                              void* y, void* bn_workspace, size_t workspace_bytes,
                                              int32_t* rows_out, pe_stream_t stream) {
    PE_REQUIRE(d != nullptr && !d->relu && !d->out_f32_nchw, PE_E_INVALID,
               "pe_conv2d_fwd_bnstats_bf16: needs a descriptor without ReLU and bf16 output");
    const StatReq sr{bn_workspace, workspace_bytes, rows_out};
    return conv2d_fwd_impl(d, x, w, nullptr, nullptr, nullptr, y, &sr, stream);
}

// The engine parameters of one transposed conv: the four output-parity phases of one launch.
static int32_t deconv_fwd_params(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                 int32_t relu, const void* x, const void* w_phases,
                                 const float* scale, const float* shift, void* y,
                                 IgemmParams& p) {
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PE_E_INVALID,
               "pe_deconv4x4s2_fwd_bf16: sizes must be positive");
    // the four output-parity phases run in one launch (gridDim.y = 4)
    base_params(p);
    p.x = static_cast<const uint16_t*>(x);
    p.w = static_cast<const uint16_t*>(w_phases);
    p.y = y;
    p.scale = scale;
    p.shift = shift;
    p.B = B;
    p.Hi = H;
    p.Wi = W;
    p.Cx = Cin;
    p.Cin = Cin;
    p.N = Cout;
    p.wk_row = 4 * Cin;
    p.nphase = 4;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int ph = 2 * py + px;
            p.ph_t0[ph] = (int16_t)(4 * ph);
            p.ph_T[ph] = 4;
            p.ph_ooy[ph] = (int16_t)py;
            p.ph_oox[ph] = (int16_t)px;
            p.ph_Ho[ph] = H;
            p.ph_Wo[ph] = W;
            p.ph_woff[ph] = (long)ph * Cout * 4 * Cin;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int t = 4 * ph + 2 * a + b;
                    p.tap_dy[t] = (int16_t)(a == 0 ? 0 : (py == 0 ? -1 : 1));
                    p.tap_dx[t] = (int16_t)(b == 0 ? 0 : (px == 0 ? -1 : 1));
                    p.tap_slot[t] = (uint8_t)(2 * a + b);
                }
        }
    p.OH = 2 * H;
    p.OW = 2 * W;
    p.osy = p.osx = 2;
    p.ldc = Cout;
    p.relu = relu;
    return PE_OK;
}

static int32_t deconv_fwd_impl(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                               int32_t relu, const void* x, const void* w_phases,
                               const float* scale, const float* shift, void* y,
                               const StatReq* sr, pe_stream_t stream) {
    IgemmParams p;
    int32_t rc = deconv_fwd_params(B, H, W, Cin, Cout, relu, x, w_phases, scale, shift, y, p);
    if (rc != PE_OK) return rc;
    rc = attach_stats(p, sr, "pe_deconv4x4s2_fwd_bf16");
    if (rc != PE_OK) return rc;
    return igemm_launch(p, as_stream(stream), "pe_deconv4x4s2_fwd_bf16");
}

extern "C" int32_t pe_deconv4x4s2_head_fwd_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                                int32_t Cout, int32_t relu, const void* x,
                                                const void* w_phases, const float* scale,
                                                const float* shift, const void* head_w,
                                                const float* head_bias, int32_t head_N,
                                                float* out_f32_nchw, pe_stream_t stream) {
    const char* what = "pe_deconv4x4s2_head_fwd_bf16";
    PE_REQUIRE(x && w_phases && head_w && out_f32_nchw, PE_E_INVALID, "%s: NULL pointer", what);
    IgemmParams p;
    // (the layer's own output is never written; the engine's checks want a non-NULL y)
    const int32_t rc = deconv_fwd_params(B, H, W, Cin, Cout, relu, x, w_phases, scale, shift,
                                         out_f32_nchw, p);
    if (rc != PE_OK) return rc;
    p.head_w = static_cast<const uint16_t*>(head_w);
    p.head_bias = head_bias;
    p.head_out = out_f32_nchw;
    p.head_N = head_N;
    return igemm_head_launch(p, as_stream(stream), what);
}

extern "C" int32_t pe_deconv4x4s2_fwd_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                           int32_t Cout, int32_t relu, const void* x,
                                           const void* w_phases, const float* scale,
                                           const float* shift, void* y, pe_stream_t stream) {
    return deconv_fwd_impl(B, H, W, Cin, Cout, relu, x, w_phases, scale, shift, y, nullptr, stream);
}

extern "C" int32_t pe_deconv4x4s2_fwd_bnstats_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                                   int32_t Cout, const void* x,
                                                   const void* w_phases, void* y,
                                                   void* bn_workspace, size_t workspace_bytes,
                                                   int32_t* rows_out, pe_stream_t stream) {
    const StatReq sr{bn_workspace, workspace_bytes, rows_out};
    return deconv_fwd_impl(B, H, W, Cin, Cout, 0, x, w_phases, nullptr, nullptr, y, &sr, stream);
}

static int32_t stem7_fwd_impl(int32_t B, int32_t H2, int32_t W2, int32_t Cout, int32_t relu,
                              const void* x_s2d, const void* w, const float* scale,
                              const float* shift, void* y, const StatReq* sr,
                              pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H2 > 0 && W2 > 0 && Cout > 0, PE_E_INVALID,
               "pe_stem7x7s2_fwd_bf16: sizes must be positive");
    IgemmParams p;
    base_params(p);
    p.x = static_cast<const uint16_t*>(x_s2d);
    p.w = static_cast<const uint16_t*>(w);
    p.y = y;
    p.scale = scale;
    p.shift = shift;
    p.B = B;
    p.Hi = H2;
    p.Wi = W2;
    p.Cx = 16;
    p.Cin = 64;   // one tap = one kernel row = 4 pixels x 16 channels, contiguous in memory
    p.cshift = 4;
    p.cmask = 15;
    p.Ho = H2;
    p.Wo = W2;
    p.N = Cout;
    p.T = 4;
    for (int r = 0; r < 4; ++r) {
        p.tap_dy[r] = (int16_t)(r - 2);
        p.tap_dx[r] = -2;
    }
    p.OH = H2;
    p.OW = W2;
    p.ldc = Cout;
    p.relu = relu;
    const int32_t rc = attach_stats(p, sr, "pe_stem7x7s2_fwd_bf16");
    if (rc != PE_OK) return rc;
    return igemm_launch(p, as_stream(stream), "pe_stem7x7s2_fwd_bf16");
}

extern "C" int32_t pe_stem7x7s2_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                         int32_t relu, const void* x_s2d, const void* w,
                                         const float* scale, const float* shift, void* y,
                                         pe_stream_t stream) {
    return stem7_fwd_impl(B, H2, W2, Cout, relu, x_s2d, w, scale, shift, y, nullptr, stream);
}

extern "C" int32_t pe_stem7x7s2_fwd_bnstats_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                                 const void* x_s2d, const void* w, void* y,
                                                 void* bn_workspace, size_t workspace_bytes,
                                                 int32_t* rows_out, pe_stream_t stream) {
    const StatReq sr{bn_workspace, workspace_bytes, rows_out};
    return stem7_fwd_impl(B, H2, W2, Cout, 0, x_s2d, w, nullptr, nullptr, y, &sr, stream);
}

static int32_t stem3_fwd_impl(int32_t B, int32_t H2, int32_t W2, int32_t Cout, int32_t relu,
                              const void* x_s2d, const void* w, const float* scale,
                              const float* shift, void* y, const StatReq* sr,
                              pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H2 > 0 && W2 > 0 && Cout > 0, PE_E_INVALID,
               "pe_stem3x3s2_fwd_bf16: sizes must be positive");
    IgemmParams p;
    base_params(p);
    p.x = static_cast<const uint16_t*>(x_s2d);
    p.w = static_cast<const uint16_t*>(w);
    p.y = y;
    p.scale = scale;
    p.shift = shift;
    p.B = B;
    p.Hi = H2;
    p.Wi = W2;
    p.Cx = 16;
    p.Cin = 64;   // one tap = one row of the 2x2 footprint, padded to 4 pixels x 16 channels
    p.cshift = 4;
    p.cmask = 15;
    p.Ho = H2;
    p.Wo = W2;
    p.N = Cout;
    p.T = 2;
    for (int r = 0; r < 2; ++r) {
        p.tap_dy[r] = (int16_t)(r - 1);
        p.tap_dx[r] = -1;
    }
    p.OH = H2;
    p.OW = W2;
    p.ldc = Cout;
    p.relu = relu;
    const int32_t rc = attach_stats(p, sr, "pe_stem3x3s2_fwd_bf16");
    if (rc != PE_OK) return rc;
    return igemm_launch(p, as_stream(stream), "pe_stem3x3s2_fwd_bf16");
}

extern "C" int32_t pe_stem3x3s2_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                         int32_t relu, const void* x_s2d, const void* w,
                                         const float* scale, const float* shift, void* y,
                                         pe_stream_t stream) {
    return stem3_fwd_impl(B, H2, W2, Cout, relu, x_s2d, w, scale, shift, y, nullptr, stream);
}

extern "C" int32_t pe_stem3x3s2_fwd_bnstats_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                                 const void* x_s2d, const void* w, void* y,
                                                 void* bn_workspace, size_t workspace_bytes,
                                                 int32_t* rows_out, pe_stream_t stream) {
    const StatReq sr{bn_workspace, workspace_bytes, rows_out};
    return stem3_fwd_impl(B, H2, W2, Cout, 0, x_s2d, w, nullptr, nullptr, y, &sr, stream);
}

extern "C" int32_t pe_preprocess_u8_s2d_bf16(const uint8_t* img, int32_t B, int32_t H, int32_t W,
                                             const float* mean3, const float* std3, void* x_s2d,
                                             pe_stream_t stream) {
    PE_REQUIRE(img && mean3 && std3 && x_s2d, PE_E_INVALID, "pe_preprocess_u8_s2d_bf16: NULL pointer");
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, PE_E_INVALID,
               "pe_preprocess_u8_s2d_bf16: H and W must be positive and even");
    PE_REQUIRE(aligned(x_s2d, 16), PE_E_INVALID, "pe_preprocess_u8_s2d_bf16: output misaligned");
    const long npix2 = (long)B * (H / 2) * (W / 2);
    PE_REQUIRE(npix2 < 0x7fffffffL, PE_E_UNSUPPORTED, "pe_preprocess_u8_s2d_bf16: too many pixels");
    pe::launch(preprocess_s2d_kernel, (unsigned)((npix2 + 255) / 256), 256, 0, as_stream(stream), img, npix2, H / 2, W / 2, mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1],
        1.0f / std3[2], static_cast<uint4*>(x_s2d));
    return check_launch("pe_preprocess_u8_s2d_bf16");
}

extern "C" int32_t pe_maxpool3x3s2_fwd_bf16(const void* x, int32_t B, int32_t H, int32_t W,
                                            int32_t C, void* y, uint8_t* idx,
                                            pe_stream_t stream) {
    PE_REQUIRE(x && y, PE_E_INVALID, "pe_maxpool3x3s2_fwd_bf16: NULL pointer");
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
               PE_E_UNSUPPORTED, "pe_maxpool3x3s2_fwd_bf16: need even H, W and C %% 8 == 0");
    PE_REQUIRE(aligned(x, 16) && aligned(y, 16) && aligned(idx, 8), PE_E_INVALID,
               "pe_maxpool3x3s2_fwd_bf16: misaligned");
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    pe::launch(maxpool3x3s2_kernel, (unsigned)((total + 255) / 256), 256, 0, as_stream(stream), static_cast<const uint16_t*>(x), B, H, W, C, static_cast<uint16_t*>(y), idx);
    return check_launch("pe_maxpool3x3s2_fwd_bf16");
}
