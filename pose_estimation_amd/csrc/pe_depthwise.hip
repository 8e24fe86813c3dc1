// Depthwise 3x3 convolution (MobileNetV2's spatial filter), NHWC bf16, stride 1 or 2, any
// dilation, "same" padding (pad = dilation), with folded-BN scale/shift and ReLU / ReLU6.
//
// Bound: HBM bandwidth.  9 MACs per element against 4 bytes of traffic, so there is nothing
// for the matrix cores here; accumulation is float32 in tap order.  Stride 1 (15 of
// MobileNetV2's 17 layers): a thread owns 4 channels of a 2 x 4 block of outputs taken
// `dilation` apart, so the 4 x 6 inputs it loads serve 8 outputs (3 cache reads per output
// instead of 9) and the nine weight vectors are loaded once.  Stride 2: 8 channels of one
// output pixel per thread, 16-byte accesses.  No load sits behind a branch: out-of-image taps
// read a valid address and are zeroed, so all of a thread's loads are in flight together.

#include <stdlib.h>

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
    const bool small = total < 0x7fffffffL;   // uniform: 32-bit index split (a quarter of the cost)
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int c8, ox, oy;
        long b;
        if (small) {
            const unsigned eu = (unsigned)e;
            c8 = (int)(eu % (unsigned)C8);
            unsigned r = eu / (unsigned)C8;
            ox = (int)(r % (unsigned)Wo);
            r /= (unsigned)Wo;
            oy = (int)(r % (unsigned)Ho);
            b = r / (unsigned)Ho;
        } else {
            c8 = (int)(e % C8);
            long r = e / C8;
            ox = (int)(r % Wo);
            r /= Wo;
            oy = (int)(r % Ho);
            b = r / Ho;
        }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        // out-of-image taps load a valid address and contribute exact zeros, so the nine loads
        // carry no branches and are all in flight together
        uint4 xv[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + (ky - 1) * dil;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + (kx - 1) * dil;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const long off = ok ? (((b * H + iy) * W + ix) * (long)C8 + c8) * 8 : (long)c8 * 8;
                uint4 v = *reinterpret_cast<const uint4*>(x + off);
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                xv[ky * 3 + kx] = v;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float xf[8], wf[8];
            dw_unpack(xv[t], xf);
            dw_unpack(*reinterpret_cast<const uint4*>(w + ((long)t * C8 + c8) * 8), wf);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += xf[i] * wf[i];
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
        o.x = pack_bf16x2_bits(acc[0], acc[1]);
        o.y = pack_bf16x2_bits(acc[2], acc[3]);
        o.z = pack_bf16x2_bits(acc[4], acc[5]);
        o.w = pack_bf16x2_bits(acc[6], acc[7]);
        *reinterpret_cast<uint4*>(y + e * 8) = o;
    }
}

// Stride-1 kernel.  Outputs are grouped by residue (oy % d, ox % d); within a residue class the
// pixels d apart form a dense lattice, and a thread takes a 2 x 4 block of that lattice for FOUR
// channels (8-byte accesses: the accumulators, the nine widened weights and a row of inputs fit
// in about 100 registers, so five waves per SIMD stay resident).  Threads are ordered
// channel-group fastest, so a wave reads and writes 512 contiguous bytes of a pixel.
constexpr int DW_TY = 2, DW_TX = 4;

__device__ __forceinline__ void dw_unpack4(const uint2& v, float* f) {
    f[0] = bf16_bits_to_float(v.x & 0xffffu);
    f[1] = bf16_bits_to_float(v.x >> 16);
    f[2] = bf16_bits_to_float(v.y & 0xffffu);
    f[3] = bf16_bits_to_float(v.y >> 16);
}

__global__ __launch_bounds__(256) void depthwise3x3_s1_kernel(const uint16_t* __restrict__ x, int B,
                                                              int H, int W, int C, int dil,
                                                              const uint16_t* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              int relu, uint16_t* __restrict__ y,
                                                              int nty, int ntx, int flags) {
    const int flip = flags & 1;
    const bool nt = (flags & 2) != 0;
    const int C4 = C >> 2;
    const long total = (long)B * nty * ntx * C4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int c4, tx, ty;
    long b;
    if (total < 0x7fffffffL) {   // uniform: 32-bit index split
        const unsigned eu = (unsigned)e;
        c4 = (int)(eu % (unsigned)C4);
        unsigned r = eu / (unsigned)C4;
        tx = (int)(r % (unsigned)ntx);
        r /= (unsigned)ntx;
        ty = (int)(r % (unsigned)nty);
        b = r / (unsigned)nty;
    } else {
        c4 = (int)(e % C4);
        long r = e / C4;
        tx = (int)(r % ntx);
        r /= ntx;
        ty = (int)(r % nty);
        b = r / nty;
    }
    const int rx = tx % dil, X0 = (tx / dil) * DW_TX;
    const int ry = ty % dil, Y0 = (ty / dil) * DW_TY;

    float wf[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        dw_unpack4(*reinterpret_cast<const uint2*>(w + ((long)(flip ? 8 - t : t) * C4 + c4) * 4),
                   wf[t]);  // flip: the spatially mirrored filter = the stride-1 backward-data pass
    float acc[DW_TY][DW_TX][4];
#pragma unroll
    for (int a = 0; a < DW_TY; ++a)
#pragma unroll
        for (int c = 0; c < DW_TX; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][c][i] = 0.0f;

    // out-of-image inputs load a valid address and become zeros: no branches around the loads,
    // so a row of six is in flight at once
    const uint16_t* xb = x + (b * H * (long)W * C4 + c4) * 4;
#pragma unroll
    for (int j = 0; j < DW_TY + 2; ++j) {
        const int iy = (Y0 + j - 1) * dil + ry;
        uint2 xv[DW_TX + 2];
#pragma unroll
        for (int i = 0; i < DW_TX + 2; ++i) {
            const int ix = (X0 + i - 1) * dil + rx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            uint2 v = *reinterpret_cast<const uint2*>(xb + (ok ? ((long)iy * W + ix) * C4 * 4 : 0L));
            if (!ok) v = make_uint2(0u, 0u);
            xv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < DW_TX + 2; ++i) {
            float xf[4];
            dw_unpack4(xv[i], xf);
#pragma unroll
            for (int a = 0; a < DW_TY; ++a) {
                const int ky = j - a;
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int c = 0; c < DW_TX; ++c) {
                    const int kx = i - c;
                    if (kx < 0 || kx > 2) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[a][c][q] += xf[q] * wf[ky * 3 + kx][q];
                }
            }
        }
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
    const float scf[4] = {sc.x, sc.y, sc.z, sc.w}, shf[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int a = 0; a < DW_TY; ++a) {
        const int oy = (Y0 + a) * dil + ry;
#pragma unroll
        for (int c = 0; c < DW_TX; ++c) {
            const int ox = (X0 + c) * dil + rx;
            if (oy >= H || ox >= W) continue;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = acc[a][c][q];
                if (scale) v[q] *= scf[q];
                if (shift) v[q] += shf[q];
                if (relu) v[q] = fmaxf(v[q], 0.0f);
                if (relu == 2) v[q] = fminf(v[q], 6.0f);
            }
            uint2 o;
            o.x = pack_bf16x2_bits(v[0], v[1]);
            o.y = pack_bf16x2_bits(v[2], v[3]);
            uint16_t* dst = y + (((b * H + oy) * (long)W + ox) * C4 + c4) * 4;
            if (nt) {
                typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t ov = {o.x, o.y};
                __builtin_nontemporal_store(ov, reinterpret_cast<u32x2_t*>(dst));
            } else {
                *reinterpret_cast<uint2*>(dst) = o;
            }
        }
    }
}

// Backward-data in gather form: an input cell sums, over the taps that reach it, weight x
// output gradient.  dx[b,iy,ix,c] = sum_{ky,kx} w[ky*3+kx][c] * dy[b,oy,ox,c] with
// oy*stride + (ky-1)*dil = iy (and likewise for x).  One thread per 8 channels of an input
// cell, nine branch-free loads (taps that do not land on an output read a valid address and
// are zeroed), float32 sum in tap order, one rounding.
__global__ __launch_bounds__(256) void depthwise3x3_bwd_data_kernel(
    const uint16_t* __restrict__ dy, int B, int H, int W, int C, int stride, int dil,
    const uint16_t* __restrict__ w, uint16_t* __restrict__ dx, int Ho, int Wo) {
    const int C8 = C >> 3;
    const long total = (long)B * H * W * C8;
    const bool small = total < 0x7fffffffL;   // uniform: 32-bit index split (a quarter of the cost)
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int c8, ix, iy;
        long b;
        if (small) {
            const unsigned eu = (unsigned)e;
            c8 = (int)(eu % (unsigned)C8);
            unsigned r = eu / (unsigned)C8;
            ix = (int)(r % (unsigned)W);
            r /= (unsigned)W;
            iy = (int)(r % (unsigned)H);
            b = r / (unsigned)H;
        } else {
            c8 = (int)(e % C8);
            long r = e / C8;
            ix = (int)(r % W);
            r /= W;
            iy = (int)(r % H);
            b = r / H;
        }
        uint4 gv[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ny = iy - (ky - 1) * dil;
            const int oy = ny / stride;
            const bool oky = ny >= 0 && ny == oy * stride && oy < Ho;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nx = ix - (kx - 1) * dil;
                const int ox = nx / stride;
                const bool ok = oky && nx >= 0 && nx == ox * stride && ox < Wo;
                const long off = ok ? (((b * Ho + oy) * Wo + ox) * (long)C8 + c8) * 8 : (long)c8 * 8;
                uint4 v = *reinterpret_cast<const uint4*>(dy + off);
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                gv[ky * 3 + kx] = v;
            }
        }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float gf[8], wf[8];
            dw_unpack(gv[t], gf);
            dw_unpack(*reinterpret_cast<const uint4*>(w + ((long)t * C8 + c8) * 8), wf);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += gf[i] * wf[i];
        }
        uint4 o;
        o.x = pack_bf16x2_bits(acc[0], acc[1]);
        o.y = pack_bf16x2_bits(acc[2], acc[3]);
        o.z = pack_bf16x2_bits(acc[4], acc[5]);
        o.w = pack_bf16x2_bits(acc[6], acc[7]);
        *reinterpret_cast<uint4*>(dx + e * 8) = o;
    }
}

// Backward-weight: dw[t][c] = sum over output cells of dy[cell][c] * x[cell*stride + tap t][c].
// A reduction over every cell per channel, in two ordered stages (no float atomics): a workgroup
// walks a contiguous range of output cells, thread (row lane, channel group) keeping 9 x 8
// float32 sums, folds its row lanes through LDS in lane order and writes one partial
// [9][C] block; the second kernel adds the blocks in order and accumulates into dw.
// 512 partial blocks (two workgroups per CU) measured best beside the main chain of the
// MobileNetV2 step: 256 / 512 / 1024 / 2048 -> 15.97 / 15.83 / 16.07 / 16.35 ms.
constexpr int kDwMaxBlocks = 512;    // the workspace is sized for this many partial blocks
constexpr int kDwBlocks = 512;

__global__ __launch_bounds__(256) void depthwise3x3_bwd_weight_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int B, int H, int W, int C,
    int stride, int dil, int Ho, int Wo, long cells_blk, int rl, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [rl][9][C]
    const int C8 = C >> 3;
    const int tid = threadIdx.x;
    const int cgi = tid % C8;
    const int rli = tid / C8;
    const long M = (long)B * Ho * Wo;
    const long m0 = (long)blockIdx.x * cells_blk;
    long m1 = m0 + cells_blk;
    if (m1 > M) m1 = M;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i] = 0.0f;
    if (rli < rl) {
        // (image, row, column) of this thread's first cell; the next one lies rl cells on, which
        // is a constant (image, row, column) step with one carry per digit: the loop divides
        // nothing (three 64-bit divisions per cell cost more than the cell's 72 multiply-adds)
        const int HoWo = Ho * Wo;
        long mm = m0 + rli;
        int b = (int)(mm / HoWo);
        int rem = (int)(mm - (long)b * HoWo);
        int oy = rem / Wo, ox = rem - (rem / Wo) * Wo;
        const int sb = rl / HoWo, sy = (rl % HoWo) / Wo, sx = (rl % HoWo) % Wo;
        for (long m = m0 + rli; m < m1; m += rl) {
            float gf[8];
            dw_unpack(*reinterpret_cast<const uint4*>(dy + (m * C8 + cgi) * 8), gf);
            uint4 xv[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * stride + (ky - 1) * dil;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * stride + (kx - 1) * dil;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    const long off = ok ? ((long)((b * H + iy) * W + ix) * C8 + cgi) * 8 : (long)cgi * 8;
                    uint4 v = *reinterpret_cast<const uint4*>(x + off);
                    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                    xv[ky * 3 + kx] = v;
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float xf[8];
                dw_unpack(xv[t], xf);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[t][i] += gf[i] * xf[i];
            }
            ox += sx;
            const int cx = ox >= Wo ? 1 : 0;
            ox -= cx ? Wo : 0;
            oy += sy + cx;
            const int cy = oy >= Ho ? 1 : 0;
            oy -= cy ? Ho : 0;
            b += sb + cy;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[((long)rli * 9 + t) * C + cgi * 8 + i] = acc[t][i];
    }
    __syncthreads();
    for (int e = tid; e < 9 * C; e += 256) {
        float s = 0.0f;
        for (int r = 0; r < rl; ++r) s += red[(long)r * 9 * C + e];
        partial[(long)blockIdx.x * 9 * C + e] = s;
    }
}

// Second stage: a workgroup folds the partial blocks of 4 elements; thread (slice, element) adds
// blocks slice, slice + 64, ... eight loads at a time, then slice 0 adds the 64 slice sums in
// order.  The order of additions depends on (nblk, n) alone.
__global__ __launch_bounds__(256) void depthwise3x3_bwd_weight_fold_kernel(
    const float* __restrict__ partial, int nblk, int n, float* __restrict__ dw) {
    __shared__ float red[64][5];
    const int el = threadIdx.x & 3;
    const int slice = threadIdx.x >> 2;
    const int e = blockIdx.x * 4 + el;
    float a = 0.0f;
    if (e < n) {
        for (int b0 = slice; b0 < nblk; b0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * 64;
                const float x = partial[(long)(b < nblk ? b : b0) * n + e];
                v[u] = b < nblk ? x : 0.0f;
            }
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    red[slice][el] = a;
    __syncthreads();
    if (slice != 0 || e >= n) return;
    float s = 0.0f;
    for (int i = 0; i < 64; ++i) s += red[i][el];
    dw[e] += s;
}

}  // namespace pe

using namespace pe;

static int32_t dw_check(const char* what, int32_t B, int32_t H, int32_t W, int32_t C,
                        int32_t stride, int32_t dilation) {
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, PE_E_UNSUPPORTED,
               "%s: sizes must be positive and C a multiple of 8", what);
    PE_REQUIRE((stride == 1 || stride == 2) && dilation >= 1 && dilation <= 64, PE_E_UNSUPPORTED,
               "%s: stride must be 1 or 2 and dilation in [1, 64]", what);
    PE_REQUIRE((long)B * H * W * (C / 8) < (1L << 40), PE_E_UNSUPPORTED, "%s: tensor too large", what);
    return PE_OK;
}

extern "C" int32_t pe_depthwise3x3_bwd_data_bf16(const void* dy, int32_t B, int32_t H, int32_t W,
                                                 int32_t C, int32_t stride, int32_t dilation,
                                                 const void* w, void* dx, pe_stream_t stream) {
    const char* what = "pe_depthwise3x3_bwd_data_bf16";
    PE_REQUIRE(dy && w && dx, PE_E_INVALID, "%s: NULL pointer", what);
    const int32_t rc = dw_check(what, B, H, W, C, stride, dilation);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(aligned(dy, 16) && aligned(w, 16) && aligned(dx, 16), PE_E_INVALID,
               "%s: tensors must be 16-byte aligned", what);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long total = (long)B * H * W * (C / 8);
    if (stride == 1) {
        // stride 1: backward-data is the forward conv of dy with the mirrored filter, so it
        // runs on the register-tiled forward kernel (float32 sum in mirrored tap order)
        const int nty = dilation * (((H + dilation - 1) / dilation + DW_TY - 1) / DW_TY);
        const int ntx = dilation * (((W + dilation - 1) / dilation + DW_TX - 1) / DW_TX);
        const long threads = (long)B * nty * ntx * (C / 4);
        PE_REQUIRE((threads + 255) / 256 < 0x7fffffffL, PE_E_UNSUPPORTED, "%s: grid too large", what);
        pe::launch(depthwise3x3_s1_kernel, (unsigned)((threads + 255) / 256), 256, 0, as_stream(stream), static_cast<const uint16_t*>(dy), B, H, W, C, dilation, static_cast<const uint16_t*>(w),
            nullptr, nullptr, 0, static_cast<uint16_t*>(dx), nty, ntx, 1);
        return check_launch(what);
    }
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    pe::launch(depthwise3x3_bwd_data_kernel, (unsigned)blocks, 256, 0, as_stream(stream), static_cast<const uint16_t*>(dy), B, H, W, C, stride, dilation,
        static_cast<const uint16_t*>(w), static_cast<uint16_t*>(dx), Ho, Wo);
    return check_launch(what);
}

extern "C" size_t pe_depthwise3x3_bwd_weight_workspace_bytes(int32_t C) {
    return C > 0 ? (size_t)kDwMaxBlocks * 9 * (size_t)C * sizeof(float) : 16;
}

extern "C" int32_t pe_depthwise3x3_bwd_weight_bf16(const void* x, const void* dy, int32_t B,
                                                   int32_t H, int32_t W, int32_t C, int32_t stride,
                                                   int32_t dilation, float* dw, void* workspace,
                                                   size_t workspace_bytes, pe_stream_t stream) {
    const char* what = "pe_depthwise3x3_bwd_weight_bf16";
    PE_REQUIRE(x && dy && dw && workspace, PE_E_INVALID, "%s: NULL pointer", what);
    const int32_t rc = dw_check(what, B, H, W, C, stride, dilation);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(C <= 2048, PE_E_UNSUPPORTED, "%s: at most 2048 channels", what);
    PE_REQUIRE((long)B * H * W < 0x7fffffffL, PE_E_UNSUPPORTED, "%s: pixel count does not fit int32",
               what);
    PE_REQUIRE(aligned(x, 16) && aligned(dy, 16) && aligned(workspace, 16), PE_E_INVALID,
               "%s: tensors must be 16-byte aligned", what);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long M = (long)B * Ho * Wo;
    const int C8 = C / 8;
    // row lanes: as many as fit 256 threads and 64 KB of LDS for the fold
    int rl = 256 / C8;
    const int rl_lds = (int)((64 * 1024) / ((size_t)9 * C * sizeof(float)));
    if (rl > rl_lds) rl = rl_lds;
    PE_REQUIRE(rl >= 1, PE_E_UNSUPPORTED, "%s: too many channels for the fold", what);
    static const char* tune = getenv("PE_DW_WGRAD_BLOCKS");   // tuning aid, at most kDwMaxBlocks
    long nb = tune ? atol(tune) : kDwBlocks;
    if (nb < 1 || nb > kDwMaxBlocks) nb = kDwBlocks;
    long cells = (M + nb - 1) / nb;
    if (cells < (long)rl * 8) cells = (long)rl * 8;
    const int nblk = (int)((M + cells - 1) / cells);
    PE_REQUIRE((size_t)nblk * 9 * C * sizeof(float) <= workspace_bytes, PE_E_INVALID,
               "%s: workspace too small", what);
    hipStream_t st = as_stream(stream);
    pe::launch(depthwise3x3_bwd_weight_kernel, nblk, 256, (size_t)rl * 9 * C * sizeof(float), st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(dy), B, H, W, C, stride,
        dilation, Ho, Wo, cells, rl, static_cast<float*>(workspace));
    int32_t rc2 = check_launch(what);
    if (rc2 != PE_OK) return rc2;
    pe::launch(depthwise3x3_bwd_weight_fold_kernel, (9 * C + 3) / 4, 256, 0, st, static_cast<const float*>(workspace), nblk, 9 * C, dw);
    return check_launch(what);
}

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
    // outputs of 64 MB and more (several times the L2, a quarter of the last-level cache) are
    // written with non-temporal stores: 1.5 % on the batch-256 MobileNetV2 forward, neutral on
    // the training step.  PE_DW_NT_MB: tuning aid, threshold in MB, 0 = off.
    static const char* nts = getenv("PE_DW_NT_MB");
    const long nt_thr = nts ? atol(nts) : 64;
    const bool nt = nt_thr > 0 && (long)B * Ho * Wo * C * 2 >= nt_thr * 1000000L;
    if (stride == 1) {
        // lattice tiles per residue class, residues interleaved: nty = d * ceil(ceil(H/d) / 2)
        const int nty = dilation * (((H + dilation - 1) / dilation + DW_TY - 1) / DW_TY);
        const int ntx = dilation * (((W + dilation - 1) / dilation + DW_TX - 1) / DW_TX);
        const long threads = (long)B * nty * ntx * (C / 4);
        PE_REQUIRE((threads + 255) / 256 < 0x7fffffffL, PE_E_UNSUPPORTED, "%s: grid too large", what);
        pe::launch(depthwise3x3_s1_kernel, (unsigned)((threads + 255) / 256), 256, 0, as_stream(stream), static_cast<const uint16_t*>(x), B, H, W, C, dilation, static_cast<const uint16_t*>(w),
            scale, shift, relu, static_cast<uint16_t*>(y), nty, ntx, nt ? 2 : 0);
        return check_launch(what);
    }
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    pe::launch(depthwise3x3_kernel, (unsigned)blocks, 256, 0, as_stream(stream), static_cast<const uint16_t*>(x), B, H, W, C, stride, dilation,
        static_cast<const uint16_t*>(w), scale, shift, relu, static_cast<uint16_t*>(y), Ho, Wo);
    return check_launch(what);
}
