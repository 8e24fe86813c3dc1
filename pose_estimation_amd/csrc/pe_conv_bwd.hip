// a4 / a5 backward entry points.
//
// Backward-data passes are gather-convs over the output gradient with transposed weights
// ([Cin][taps][Cout]) and run on the forward engine (pe_igemm.hip): stride-1 convs as one
// launch, stride-2 convs as up to four input-parity phases with a scattered store, the 4x4/2
// transposed conv as one 16-tap stride-2 conv.  Backward-weight passes run on pe_wgrad.hip.
// The rest of the file is the small bandwidth-bound glue of the backward graph: weight
// transposes, max-pool backward, the float32-NCHW -> bf16-NHWC hand-over of the loss gradient.
#include "pe_common.h"
#include <math.h>

#include "pe_igemm.h"
#include "pe_wgrad.h"

namespace pe {

// in [A][T][C] bf16 -> out[c][t0 + t][a] with rows of `slots * apitch` elements.
// 64 x 64 tiles through LDS so both sides move whole 128-byte runs.
__global__ __launch_bounds__(256) void transpose_w_kernel(const uint16_t* __restrict__ in, int A,
                                                          int T, int C,
                                                          uint16_t* __restrict__ out, int slots,
                                                          int apitch, int t0) {
    __shared__ uint16_t tile[64][66];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int a = a0 + r, c = c0 + tx;
        tile[r][tx] = (a < A && c < C) ? in[((long)a * T + t) * C + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, a = a0 + tx;
        if (c < C && a < A) out[(long)c * slots * apitch + (long)(t0 + t) * apitch + a] = tile[tx][r];
    }
}

// The same transpose for a whole table of weights in one launch (a training step re-packs
// about 70 small tensors; one launch each would be launch-latency bound).  Workgroup b works
// on tile b - tile_begin of the job whose tile range holds b.
__global__ __launch_bounds__(256) void transpose_w_batched_kernel(const pe_transpose_job* __restrict__ jobs,
                                                                  int njobs) {
    __shared__ uint16_t tile[64][66];
    __shared__ int job_s;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;  // last job with tile_begin <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].tile_begin <= (int)blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        job_s = lo;
    }
    __syncthreads();
    const pe_transpose_job jb = jobs[job_s];
    const int A = jb.A, T = jb.T, C = jb.C;
    // the job's taps may be a window [src_t0, src_t0 + T) of a wider source [A][src_T][C]
    const int srcT = jb.src_T > 0 ? jb.src_T : T;
    const int ntc = (C + 63) / 64, nta = (A + 63) / 64;
    int local = blockIdx.x - jb.tile_begin;
    const int t = local / (ntc * nta);
    if (t >= T) return;  // uniform
    local -= t * ntc * nta;
    const int a0 = (local / ntc) * 64;
    const int c0 = (local % ntc) * 64;
    const uint16_t* in = static_cast<const uint16_t*>(jb.in);
    uint16_t* out = static_cast<uint16_t*>(jb.out);
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int a = a0 + r, c = c0 + tx;
        tile[r][tx] =
            (a < A && c < C) ? in[((long)a * srcT + jb.src_t0 + t) * C + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, a = a0 + tx;
        if (c < C && a < A)
            out[(long)c * jb.slots * jb.apitch + (long)(jb.t0 + t) * jb.apitch + a] = tile[tx][r];
    }
}

// Max-pool 3x3/2 pad 1 backward in gather form: an input cell receives the gradient of every
// window whose recorded first-maximum position (idx = ky*3+kx, written by the forward pass) is
// that cell.  Reads idx and dy of at most four windows per cell; never touches x.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const uint8_t* __restrict__ idx,
                                                               const uint16_t* __restrict__ dy,
                                                               int B, int H, int W, int C,
                                                               uint16_t* __restrict__ dx) {
    // one thread per 8 channels of one input cell, 16-byte stores
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long total = (long)B * H * W * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H);
        const long b = r / H;
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = 0.0f;
        // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const uint32_t me = (uint32_t)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));
                const long o = (((b * Ho + oy) * Wo + ox) * (long)C8 + c8) * 8;
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
                const uint4 dv = *reinterpret_cast<const uint4*>(dy + o);
                const uint32_t u[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t word = i < 2 ? pk.x : pk.y;
                    const uint32_t p0 = (word >> (16 * (i & 1))) & 0xffu;
                    const uint32_t p1 = (word >> (16 * (i & 1) + 8)) & 0xffu;
                    if (p0 == me) g[2 * i] += bf16_bits_to_float(u[i] & 0xffffu);
                    if (p1 == me) g[2 * i + 1] += bf16_bits_to_float(u[i] >> 16);
                }
            }
        }
        uint4 o;
        o.x = pack_bf16x2_bits(g[0], g[1]);
        o.y = pack_bf16x2_bits(g[2], g[3]);
        o.z = pack_bf16x2_bits(g[4], g[5]);
        o.w = pack_bf16x2_bits(g[6], g[7]);
        *reinterpret_cast<uint4*>(dx + e * 8) = o;
    }
}

// Loss gradient hand-over: float32 [B][K][HW] -> bf16 [B*HW][Kpad] (zero padded).
__global__ __launch_bounds__(256) void final_grad_prepare_kernel(const float* __restrict__ g,
                                                                 long npix, int K, int HW,
                                                                 int Kpad,
                                                                 uint16_t* __restrict__ out) {
    // reads are coalesced along pixels (one plane per k); the [256 pixels][Kpad] tile is
    // transposed through LDS so the bf16 rows leave as whole 16-byte chunks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pitch = Kpad + 8;
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
    const long pix0 = (long)blockIdx.x * 256;
    const long pix = pix0 + threadIdx.x;
    const bool ok = pix < npix;
    const long b = ok ? pix / HW : 0;
    const int pl = ok ? (int)(pix - b * HW) : 0;
    for (int k = 0; k < Kpad; ++k) {
        float v = 0.0f;
        if (ok && k < K) v = g[(b * K + k) * HW + pl];
        tile[threadIdx.x * pitch + k] = (uint16_t)float_to_bf16_bits(v);
    }
    __syncthreads();
    const int cpr = Kpad / 8;  // 16-byte chunks per pixel row
    for (int ch = threadIdx.x; ch < 256 * cpr; ch += 256) {
        const int row = ch / cpr;
        const int cc = ch - row * cpr;
        if (pix0 + row < npix)
            *reinterpret_cast<uint4*>(out + (pix0 + row) * Kpad + cc * 8) =
                *reinterpret_cast<const uint4*>(tile + row * pitch + cc * 8);
    }
}

// dbias[k] += sum over images and pixels of g[b][k][:], one 1024-thread workgroup per k, summed
// in a fixed order (thread-strided 16-byte partials, wave shuffle tree, waves in order).
__global__ __launch_bounds__(1024) void bias_grad_kernel(const float* __restrict__ g, int B, int K,
                                                         int HW, float* __restrict__ dbias) {
    __shared__ float red[16];
    const int k = blockIdx.x;
    float s = 0.0f;
    if ((HW & 3) == 0) {
        const int hw4 = HW >> 2;
        const long n4 = (long)B * hw4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (long j = threadIdx.x; j < n4; j += 1024) {
            const long b = j / hw4;
            const int off = (int)(j - b * hw4);
            acc += __builtin_nontemporal_load(
                reinterpret_cast<const f32x4*>(g + (b * K + k) * (long)HW) + off);
        }
        s = (acc.x + acc.y) + (acc.z + acc.w);
    } else {
        for (int b = 0; b < B; ++b) {
            const float* plane = g + ((long)b * K + k) * HW;
            for (int i = threadIdx.x; i < HW; i += 1024) s += plane[i];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        dbias[k] += t;
    }
}

__global__ __launch_bounds__(256) void mul_f32_kernel(float* __restrict__ a,
                                                      const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        a[i] *= b[i];
}

}  // namespace pe

using namespace pe;

// Backward-data of a conv whose OUTPUT has 8, 16 or 32 channels: the gradient dy is the GEMM's
// "activation" operand and one 64-element K run then spans P = 64/Cout horizontally adjacent dy
// pixels.  Inside one input-parity class (py, px) and one filter row kh, the taps kw that reach
// that class read dy columns (px + pad - kw) / s; sorted by that offset they are grouped into
// runs of P adjacent columns.  A run is one engine tap and owns one slot of the re-packed
// weights w_t [Cin][slots][P * Cout]; (kh, kw) sits in sub-block `sub` of its run's slot, and
// sub-blocks no tap reaches hold zeros.  Slots are numbered py, px, kh, run, in that order.
// Returns the slot count (0 when it does not fit the tap table).
static int narrow_bwd_layout(int KH, int KW, int s, int pad, int P, int32_t* slot_of_tap,
                             int32_t* sub_of_tap) {
    int nslot = 0;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px)
            for (int kh = 0; kh < KH; ++kh) {
                if ((py + pad - kh) % s != 0) continue;
                // kw descending = dy column offset ascending
                int run_start = 0;
                bool open = false;
                for (int kw = KW - 1; kw >= 0; --kw) {
                    if ((px + pad - kw) % s != 0) continue;
                    const int off = (px + pad - kw) / s;
                    if (!open || off >= run_start + P) {
                        run_start = off;
                        open = true;
                        ++nslot;
                    }
                    if (slot_of_tap) {
                        slot_of_tap[kh * KW + kw] = nslot - 1;
                        sub_of_tap[kh * KW + kw] = off - run_start;
                    }
                }
            }
    return nslot;
}

extern "C" int32_t pe_conv2d_bwd_data_narrow_layout(int32_t KH, int32_t KW, int32_t stride,
                                                    int32_t pad, int32_t Cout,
                                                    int32_t* slot_of_tap, int32_t* sub_of_tap,
                                                    int32_t* nslots) {
    PE_REQUIRE(Cout == 8 || Cout == 16 || Cout == 32, PE_E_INVALID,
               "pe_conv2d_bwd_data_narrow_layout: Cout must be 8, 16 or 32");
    PE_REQUIRE(KH > 0 && KW > 0 && KH * KW <= kMaxTaps && (stride == 1 || stride == 2) && pad >= 0 &&
                   slot_of_tap && sub_of_tap && nslots,
               PE_E_INVALID, "pe_conv2d_bwd_data_narrow_layout: bad arguments");
    *nslots = narrow_bwd_layout(KH, KW, stride, pad, 64 / Cout, slot_of_tap, sub_of_tap);
    return PE_OK;
}

// The consumer side of a fused batch-norm backward: dx is the gradient of an activation
// a = act(z * scale + shift); the launch stores it masked by the activation and leaves the two
// reductions batch-norm backward needs (sum g, sum g*z per channel) in the workspace.
struct MaskReq {
    const void* z;
    const float* act_scale;
    const float* act_shift;
    int32_t relu;            // 1: ReLU, 2: ReLU6
    const void* act;         // residual layers: the stored activation gives the mask (ReLU)
    void* workspace;
    size_t bytes;
    int32_t* rows_out;
};

static int32_t attach_mask(IgemmParams& p, const MaskReq* mr, const char* what) {
    if (mr == nullptr) return PE_OK;
    PE_REQUIRE(mr->z && mr->workspace && mr->rows_out && (mr->act || (mr->act_scale && mr->act_shift)),
               PE_E_INVALID, "%s: NULL mask operand", what);
    PE_REQUIRE(mr->relu == 1 || (mr->relu == 2 && !mr->act), PE_E_INVALID,
               "%s: relu must be 1 or 2 (1 with a stored activation)", what);
    PE_REQUIRE(aligned(mr->workspace, 16) && p.N % 8 == 0, PE_E_INVALID,
               "%s: the fused mask needs an aligned workspace and Cin %% 8 == 0", what);
    PE_REQUIRE(p.res == nullptr || mr->act != nullptr, PE_E_UNSUPPORTED,
               "%s: a residual needs the mask from the stored activation", what);
    PE_REQUIRE(aligned(mr->z, 16) && aligned(mr->act, 16), PE_E_INVALID,
               "%s: z and the activation must be 16-byte aligned", what);
    p.mask_a = static_cast<const uint16_t*>(mr->act);
    p.mask_z = static_cast<const uint16_t*>(mr->z);
    p.mask_sc = mr->act_scale;
    p.mask_sh = mr->act_shift;
    p.mask_hi = mr->relu == 2 ? 6.0f : INFINITY;
    p.stat_partial = static_cast<float*>(mr->workspace);
    p.stat_capacity_rows = (long)(mr->bytes / ((size_t)2 * p.N * sizeof(float)));
    p.stat_rows_out = mr->rows_out;
    return PE_OK;
}

static int32_t conv2d_bwd_data_impl(const pe_conv_desc* d, const void* dy, const void* w_t,
                                    const void* residual, void* dx, const MaskReq* mr,
                                    pe_stream_t stream) {
    PE_REQUIRE(d != nullptr, PE_E_INVALID, "pe_conv2d_bwd_data_bf16: NULL descriptor");
    PE_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= kMaxTaps, PE_E_UNSUPPORTED,
               "pe_conv2d_bwd_data_bf16: filter %dx%d not supported", d->KH, d->KW);
    PE_REQUIRE(d->stride == 1 || d->stride == 2, PE_E_UNSUPPORTED,
               "pe_conv2d_bwd_data_bf16: stride must be 1 or 2");
    PE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->pad >= 0,
               PE_E_INVALID, "pe_conv2d_bwd_data_bf16: bad sizes");
    const int s = d->stride;
    const int Ho = (d->H + 2 * d->pad - d->KH) / s + 1;
    const int Wo = (d->W + 2 * d->pad - d->KW) / s + 1;
    PE_REQUIRE(Ho > 0 && Wo > 0, PE_E_INVALID, "pe_conv2d_bwd_data_bf16: empty output");
    hipStream_t st = as_stream(stream);
    const int T = d->KH * d->KW;
    const bool narrow = d->Cout == 8 || d->Cout == 16 || d->Cout == 32;
    const int P = narrow ? 64 / d->Cout : 1;
    int32_t slot_of[kMaxTaps], sub_of[kMaxTaps];
    int nslots = T;
    if (narrow) {
        nslots = narrow_bwd_layout(d->KH, d->KW, s, d->pad, P, slot_of, sub_of);
        PE_REQUIRE(nslots <= kMaxTaps, PE_E_UNSUPPORTED,
                   "pe_conv2d_bwd_data_bf16: filter %dx%d not supported", d->KH, d->KW);
    }

    // which phases have taps?
    bool need_zero = false;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            int cnt = 0;
            for (int kh = 0; kh < d->KH; ++kh)
                for (int kw = 0; kw < d->KW; ++kw)
                    if (((py + d->pad - kh) % s) == 0 && ((px + d->pad - kw) % s) == 0) ++cnt;
            if (cnt == 0) need_zero = true;
        }
    if (need_zero) {
        PE_REQUIRE(residual == nullptr, PE_E_UNSUPPORTED,
                   "pe_conv2d_bwd_data_bf16: residual needs every input parity to have a tap");
        PE_REQUIRE(dx != nullptr, PE_E_INVALID, "pe_conv2d_bwd_data_bf16: NULL dx");
        const hipError_t e = pe::memset_async(dx, 0, (size_t)d->B * d->H * d->W * d->Cin * 2, st);
        PE_REQUIRE(e == hipSuccess, PE_E_HIP, "pe_conv2d_bwd_data_bf16: memset: %s",
                   hipGetErrorString(e));
    }
    // every input-parity class that has taps becomes one phase of a single launch
    IgemmParams p;
    igemm_defaults(p);
    int t = 0, nph = 0;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int Hp = (d->H - py + s - 1) / s, Wp = (d->W - px + s - 1) / s;
            if (Hp <= 0 || Wp <= 0) continue;
            const int tstart = t;
            for (int kh = 0; kh < d->KH; ++kh) {
                int last_slot = -1;
                for (int i = 0; i < d->KW; ++i) {
                    const int kw = narrow ? d->KW - 1 - i : i;  // narrow: dy column ascending
                    const int ny = py + d->pad - kh, nx = px + d->pad - kw;
                    if ((ny % s) != 0 || (nx % s) != 0) continue;
                    if (narrow) {
                        // one engine tap per run: the tap that opens the run (sub-block 0)
                        if (slot_of[kh * d->KW + kw] == last_slot) continue;
                        last_slot = slot_of[kh * d->KW + kw];
                        p.tap_slot[t] = (uint8_t)last_slot;
                    } else {
                        p.tap_slot[t] = (uint8_t)(kh * d->KW + kw);
                    }
                    p.tap_dy[t] = (int16_t)(ny / s);
                    p.tap_dx[t] = (int16_t)(nx / s);
                    ++t;
                }
            }
            if (t == tstart) continue;
            p.ph_t0[nph] = (int16_t)tstart;
            p.ph_T[nph] = (int16_t)(t - tstart);
            p.ph_ooy[nph] = (int16_t)py;
            p.ph_oox[nph] = (int16_t)px;
            p.ph_Ho[nph] = Hp;
            p.ph_Wo[nph] = Wp;
            p.ph_woff[nph] = 0;
            ++nph;
        }
    PE_REQUIRE(nph > 0, PE_E_INVALID, "pe_conv2d_bwd_data_bf16: no taps reach the input");
    p.nphase = nph;
    p.x = static_cast<const uint16_t*>(dy);
    p.w = static_cast<const uint16_t*>(w_t);
    p.y = dx;
    p.res = static_cast<const uint16_t*>(residual);
    p.B = d->B;
    p.Hi = Ho;
    p.Wi = Wo;
    p.Cx = d->Cout;
    p.Cin = narrow ? 64 : d->Cout;
    if (narrow) {
        p.cshift = d->Cout == 8 ? 3 : (d->Cout == 16 ? 4 : 5);
        p.cmask = d->Cout - 1;
    }
    p.N = d->Cin;
    p.wk_row = nslots * p.Cin;
    p.OH = d->H;
    p.OW = d->W;
    p.osy = p.osx = s;
    p.ldc = d->Cin;
    if (mr) {
        PE_REQUIRE(!need_zero, PE_E_UNSUPPORTED,
                   "pe_conv2d_bwd_data_bnmask_bf16: every input parity must have a tap");
        const int32_t rc = attach_mask(p, mr, "pe_conv2d_bwd_data_bnmask_bf16");
        if (rc != PE_OK) return rc;
    }
    return igemm_launch(p, st, mr ? "pe_conv2d_bwd_data_bnmask_bf16" : "pe_conv2d_bwd_data_bf16");
}

extern "C" int32_t pe_conv2d_bwd_data_bf16(const pe_conv_desc* d, const void* dy, const void* w_t,
                                           const void* residual, void* dx, pe_stream_t stream) {
    return conv2d_bwd_data_impl(d, dy, w_t, residual, dx, nullptr, stream);
}

extern "C" int32_t pe_conv2d_bwd_data_bnmask_bf16(const pe_conv_desc* d, const void* dy,
                                                  const void* w_t, const void* z,
                                                  const float* act_scale, const float* act_shift,
                                                  int32_t relu, void* g, void* bn_workspace,
                                                  size_t workspace_bytes, int32_t* rows_out,
                                                  pe_stream_t stream) {
    const MaskReq mr{z, act_scale, act_shift, relu, nullptr, bn_workspace, workspace_bytes, rows_out};
    return conv2d_bwd_data_impl(d, dy, w_t, nullptr, g, &mr, stream);
}

extern "C" int32_t pe_conv2d_bwd_data_resmask_bf16(const pe_conv_desc* d, const void* dy,
                                                   const void* w_t, const void* residual,
                                                   const void* act, const void* z, void* g,
                                                   void* bn_workspace, size_t workspace_bytes,
                                                   int32_t* rows_out, pe_stream_t stream) {
    PE_REQUIRE(act != nullptr, PE_E_INVALID, "pe_conv2d_bwd_data_resmask_bf16: NULL activation");
    const MaskReq mr{z, nullptr, nullptr, 1, act, bn_workspace, workspace_bytes, rows_out};
    return conv2d_bwd_data_impl(d, dy, w_t, residual, g, &mr, stream);
}

static int32_t deconv_bwd_data_impl(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                    const void* dy, const void* w_t, void* dx, const MaskReq* mr,
                                    pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PE_E_INVALID,
               "pe_deconv4x4s2_bwd_data_bf16: sizes must be positive");
    IgemmParams p;
    igemm_defaults(p);
    int t = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int dyy = a == 0 ? 0 : (py == 0 ? -1 : 1);
                    const int dxx = b == 0 ? 0 : (px == 0 ? -1 : 1);
                    p.tap_dy[t] = (int16_t)(py - 2 * dyy);
                    p.tap_dx[t] = (int16_t)(px - 2 * dxx);
                    p.tap_slot[t] = (uint8_t)((2 * py + px) * 4 + 2 * a + b);
                    ++t;
                }
    p.x = static_cast<const uint16_t*>(dy);
    p.w = static_cast<const uint16_t*>(w_t);
    p.y = dx;
    p.B = B;
    p.Hi = 2 * H;
    p.Wi = 2 * W;
    p.Cx = Cout;
    p.Cin = Cout;
    p.Ho = H;
    p.Wo = W;
    p.sy = p.sx = 2;
    p.N = Cin;
    p.T = 16;
    p.OH = H;
    p.OW = W;
    p.ldc = Cin;
    const int32_t rc = attach_mask(p, mr, "pe_deconv4x4s2_bwd_data_bnmask_bf16");
    if (rc != PE_OK) return rc;
    return igemm_launch(p, as_stream(stream),
                        mr ? "pe_deconv4x4s2_bwd_data_bnmask_bf16" : "pe_deconv4x4s2_bwd_data_bf16");
}

extern "C" int32_t pe_deconv4x4s2_bwd_data_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                                int32_t Cout, const void* dy, const void* w_t,
                                                void* dx, pe_stream_t stream) {
    return deconv_bwd_data_impl(B, H, W, Cin, Cout, dy, w_t, dx, nullptr, stream);
}

extern "C" int32_t pe_deconv4x4s2_bwd_data_bnmask_bf16(int32_t B, int32_t H, int32_t W,
                                                       int32_t Cin, int32_t Cout, const void* dy,
                                                       const void* w_t, const void* z,
                                                       const float* act_scale,
                                                       const float* act_shift, int32_t relu,
                                                       void* g, void* bn_workspace,
                                                       size_t workspace_bytes, int32_t* rows_out,
                                                       pe_stream_t stream) {
    const MaskReq mr{z, act_scale, act_shift, relu, nullptr, bn_workspace, workspace_bytes, rows_out};
    return deconv_bwd_data_impl(B, H, W, Cin, Cout, dy, w_t, g, &mr, stream);
}

// pe_conv_desc + tensors -> the weight-gradient GEMM's parameters
static int32_t conv_wgrad_params(const pe_conv_desc* d, const void* x, const void* dy, int32_t ldy,
                                 float* dw, WgradParams& p, const char* what) {
    PE_REQUIRE(d != nullptr, PE_E_INVALID, "%s: NULL descriptor", what);
    PE_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= kMaxTaps, PE_E_UNSUPPORTED,
               "%s: filter %dx%d not supported", what, d->KH, d->KW);
    PE_REQUIRE(d->stride >= 1 && d->pad >= 0 && d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 &&
                   d->Cout > 0,
               PE_E_INVALID, "%s: bad sizes", what);
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    const int Npad = (d->Cout + 7) / 8 * 8;
    PE_REQUIRE(ldy >= Npad, PE_E_INVALID, "%s: dy pitch %d smaller than Cout rounded up to 8", what,
               ldy);
    wgrad_defaults(p);
    p.x = static_cast<const uint16_t*>(x);
    p.dy = static_cast<const uint16_t*>(dy);
    p.dw = dw;
    p.B = d->B;
    p.Hi = d->H;
    p.Wi = d->W;
    p.Cx = d->Cin;
    p.Ho = Ho;
    p.Wo = Wo;
    p.sy = p.sx = d->stride;
    p.N = Npad;
    p.Nw = d->Cout;
    if (d->Cin == 8 || d->Cin == 16 || d->Cin == 32) {
        // narrow inputs: dw comes out in the forward conv's narrow packing
        // [Cout][KH][ceil(KW/P)][P*Cin]; elements of runs that reach past the filter receive the
        // correlation with the pixel beyond it and must be masked off by the caller
        const int P = 64 / d->Cin;
        const int KWc = (d->KW + P - 1) / P;
        PE_REQUIRE(d->KH * KWc <= kMaxTaps, PE_E_UNSUPPORTED, "%s: filter %dx%d not supported",
                   what, d->KH, d->KW);
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
        p.Cin = d->Cin;
        p.T = d->KH * d->KW;
        for (int kh = 0; kh < d->KH; ++kh)
            for (int kw = 0; kw < d->KW; ++kw) {
                p.tap_dy[kh * d->KW + kw] = (int16_t)(kh - d->pad);
                p.tap_dx[kh * d->KW + kw] = (int16_t)(kw - d->pad);
            }
    }
    p.OH = Ho;
    p.OW = Wo;
    p.ldy = ldy;
    return PE_OK;
}

extern "C" int32_t pe_conv2d_bwd_weight_bf16(const pe_conv_desc* d, const void* x, const void* dy,
                                             int32_t ldy, float* dw, void* workspace,
                                             size_t workspace_bytes, pe_stream_t stream) {
    WgradParams p;
    const int32_t rc = conv_wgrad_params(d, x, dy, ldy, dw, p, "pe_conv2d_bwd_weight_bf16");
    if (rc != PE_OK) return rc;
    return wgrad_launch(p, workspace, workspace_bytes, as_stream(stream),
                        "pe_conv2d_bwd_weight_bf16");
}

extern "C" size_t pe_wgrad_batched_workspace_bytes(int64_t total_weight_elems) {
    return wgrad_batch_workspace_bytes(total_weight_elems > 0 ? (long)total_weight_elems : 0);
}

extern "C" int32_t pe_conv2d_bwd_weight_batched_bf16(const pe_wgrad_job* jobs, int32_t njobs,
                                                     void* workspace, size_t workspace_bytes,
                                                     pe_stream_t stream) {
    const char* what = "pe_conv2d_bwd_weight_batched_bf16";
    PE_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= PE_WGRAD_MAX_BATCH, PE_E_INVALID,
               "%s: 1 to %d jobs per call", what, PE_WGRAD_MAX_BATCH);
    static_assert(PE_WGRAD_MAX_BATCH == kWgradMaxBatch, "header and kernel disagree");
    WgradParams ps[kWgradMaxBatch];
    for (int j = 0; j < njobs; ++j) {
        const int32_t rc =
            conv_wgrad_params(&jobs[j].desc, jobs[j].x, jobs[j].dy, jobs[j].ldy, jobs[j].dw, ps[j], what);
        if (rc != PE_OK) return rc;
    }
    return wgrad_launch_batch(ps, njobs, workspace, workspace_bytes, as_stream(stream), what);
}

extern "C" int32_t pe_deconv4x4s2_bwd_weight_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                                  int32_t Cout, const void* x, const void* dy,
                                                  float* dw_phases, void* workspace,
                                                  size_t workspace_bytes, pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PE_E_INVALID,
               "pe_deconv4x4s2_bwd_weight_bf16: sizes must be positive");
    PE_REQUIRE(dw_phases != nullptr, PE_E_INVALID, "pe_deconv4x4s2_bwd_weight_bf16: NULL dw");
    // the four output parities are four weight gradients of the same size: one batched launch
    WgradParams ps[4];
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            WgradParams& p = ps[2 * py + px];
            wgrad_defaults(p);
            p.x = static_cast<const uint16_t*>(x);
            p.dy = static_cast<const uint16_t*>(dy);
            p.dw = dw_phases + (size_t)(2 * py + px) * Cout * 4 * Cin;
            p.B = B;
            p.Hi = H;
            p.Wi = W;
            p.Cx = Cin;
            p.Cin = Cin;
            p.Ho = H;
            p.Wo = W;
            p.N = Cout;
            p.Nw = Cout;
            p.T = 4;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    p.tap_dy[2 * a + b] = (int16_t)(a == 0 ? 0 : (py == 0 ? -1 : 1));
                    p.tap_dx[2 * a + b] = (int16_t)(b == 0 ? 0 : (px == 0 ? -1 : 1));
                }
            p.OH = 2 * H;
            p.OW = 2 * W;
            p.osy = p.osx = 2;
            p.ooy = py;
            p.oox = px;
            p.ldy = Cout;
        }
    return wgrad_launch_batch(ps, 4, workspace, workspace_bytes, as_stream(stream),
                              "pe_deconv4x4s2_bwd_weight_bf16");
}

extern "C" int32_t pe_stem7x7s2_bwd_weight_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                                const void* x_s2d, const void* dy, float* dw,
                                                void* workspace, size_t workspace_bytes,
                                                pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H2 > 0 && W2 > 0 && Cout > 0, PE_E_INVALID,
               "pe_stem7x7s2_bwd_weight_bf16: sizes must be positive");
    WgradParams p;
    wgrad_defaults(p);
    p.x = static_cast<const uint16_t*>(x_s2d);
    p.dy = static_cast<const uint16_t*>(dy);
    p.dw = dw;
    p.B = B;
    p.Hi = H2;
    p.Wi = W2;
    p.Cx = 16;
    p.Cin = 64;
    p.cshift = 4;
    p.cmask = 15;
    p.Ho = H2;
    p.Wo = W2;
    p.N = Cout;
    p.Nw = Cout;
    p.T = 4;
    for (int r = 0; r < 4; ++r) {
        p.tap_dy[r] = (int16_t)(r - 2);
        p.tap_dx[r] = -2;
    }
    p.OH = H2;
    p.OW = W2;
    p.ldy = Cout;
    return wgrad_launch(p, workspace, workspace_bytes, as_stream(stream),
                        "pe_stem7x7s2_bwd_weight_bf16");
}

extern "C" int32_t pe_stem3x3s2_bwd_weight_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                                const void* x_s2d, const void* dy, float* dw,
                                                void* workspace, size_t workspace_bytes,
                                                pe_stream_t stream) {
    PE_REQUIRE(B > 0 && H2 > 0 && W2 > 0 && Cout > 0, PE_E_INVALID,
               "pe_stem3x3s2_bwd_weight_bf16: sizes must be positive");
    WgradParams p;
    wgrad_defaults(p);
    p.x = static_cast<const uint16_t*>(x_s2d);
    p.dy = static_cast<const uint16_t*>(dy);
    p.dw = dw;
    p.B = B;
    p.Hi = H2;
    p.Wi = W2;
    p.Cx = 16;
    p.Cin = 64;
    p.cshift = 4;
    p.cmask = 15;
    p.Ho = H2;
    p.Wo = W2;
    p.N = (Cout + 7) / 8 * 8;
    p.Nw = Cout;
    p.T = 2;
    for (int r = 0; r < 2; ++r) {
        p.tap_dy[r] = (int16_t)(r - 1);
        p.tap_dx[r] = -1;
    }
    p.OH = H2;
    p.OW = W2;
    p.ldy = Cout;
    return wgrad_launch(p, workspace, workspace_bytes, as_stream(stream),
                        "pe_stem3x3s2_bwd_weight_bf16");
}

extern "C" int32_t pe_transpose_weights_bf16(const void* in, int32_t A, int32_t T, int32_t C,
                                             void* out, int32_t slots, int32_t apitch, int32_t t0,
                                             pe_stream_t stream) {
    PE_REQUIRE(in && out, PE_E_INVALID, "pe_transpose_weights_bf16: NULL pointer");
    PE_REQUIRE(A > 0 && T > 0 && C > 0 && slots >= t0 + T && t0 >= 0 && apitch >= A, PE_E_INVALID,
               "pe_transpose_weights_bf16: bad sizes");
    PE_REQUIRE(T <= 65535, PE_E_UNSUPPORTED, "pe_transpose_weights_bf16: too many taps");
    dim3 grid((C + 63) / 64, (A + 63) / 64, T);
    pe::launch(transpose_w_kernel, grid, 256, 0, as_stream(stream), static_cast<const uint16_t*>(in), A, T, C, static_cast<uint16_t*>(out), slots, apitch, t0);
    return check_launch("pe_transpose_weights_bf16");
}

extern "C" int32_t pe_transpose_weights_batched_bf16(const pe_transpose_job* jobs_device,
                                                     int32_t njobs, int32_t total_tiles,
                                                     pe_stream_t stream) {
    PE_REQUIRE(jobs_device != nullptr && njobs > 0 && total_tiles > 0, PE_E_INVALID,
               "pe_transpose_weights_batched_bf16: empty job table");
    pe::launch(transpose_w_batched_kernel, (unsigned)total_tiles, 256, 0, as_stream(stream), jobs_device,
                                                                                    njobs);
    return check_launch("pe_transpose_weights_batched_bf16");
}

extern "C" int32_t pe_maxpool3x3s2_bwd_bf16(const uint8_t* idx, const void* dy, int32_t B,
                                            int32_t H, int32_t W, int32_t C, void* dx,
                                            pe_stream_t stream) {
    PE_REQUIRE(idx && dy && dx, PE_E_INVALID, "pe_maxpool3x3s2_bwd_bf16: NULL pointer");
    PE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
               PE_E_UNSUPPORTED, "pe_maxpool3x3s2_bwd_bf16: need even H, W and C %% 8 == 0");
    PE_REQUIRE(aligned(idx, 8) && aligned(dy, 16) && aligned(dx, 16), PE_E_INVALID,
               "pe_maxpool3x3s2_bwd_bf16: tensors must be 16-byte aligned (idx: 8)");
    const long total = (long)B * H * W * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    pe::launch(maxpool3x3s2_bwd_kernel, (unsigned)blocks, 256, 0, as_stream(stream), idx, static_cast<const uint16_t*>(dy), B, H, W, C, static_cast<uint16_t*>(dx));
    return check_launch("pe_maxpool3x3s2_bwd_bf16");
}

extern "C" int32_t pe_final_grad_prepare_bf16(const float* g_nchw, int32_t B, int32_t K,
                                              int32_t HW, int32_t Kpad, void* out_nhwc,
                                              float* dbias, pe_stream_t stream) {
    PE_REQUIRE(g_nchw && out_nhwc, PE_E_INVALID, "pe_final_grad_prepare_bf16: NULL pointer");
    PE_REQUIRE(B > 0 && K > 0 && HW > 0 && Kpad >= K && Kpad % 8 == 0 && Kpad <= 256,
               PE_E_UNSUPPORTED,
               "pe_final_grad_prepare_bf16: need K > 0, Kpad >= K, Kpad %% 8 == 0, Kpad <= 256");
    PE_REQUIRE(aligned(out_nhwc, 16), PE_E_INVALID, "pe_final_grad_prepare_bf16: output misaligned");
    const long npix = (long)B * HW;
    const size_t lds = (size_t)256 * (Kpad + 8) * sizeof(uint16_t);
    if (lds > 64 * 1024) {   // 160 KB of LDS per CU; anything above 64 KB is opt-in
        const hipError_t e = pe::ensure_dynamic_lds<&final_grad_prepare_kernel>(
            256 * (256 + 8) * (int)sizeof(uint16_t));
        PE_REQUIRE(e == hipSuccess, PE_E_HIP, "pe_final_grad_prepare_bf16: LDS opt-in: %s",
                   hipGetErrorString(e));
    }
    pe::launch(final_grad_prepare_kernel, (unsigned)((npix + 255) / 256), 256, lds, as_stream(stream), g_nchw, npix, K, HW, Kpad, static_cast<uint16_t*>(out_nhwc));
    const int32_t rc = check_launch("pe_final_grad_prepare_bf16");
    if (rc != PE_OK || dbias == nullptr) return rc;
    pe::launch(bias_grad_kernel, (unsigned)K, 1024, 0, as_stream(stream), g_nchw, B, K, HW, dbias);
    return check_launch("pe_final_grad_prepare_bf16 (bias)");
}

extern "C" size_t pe_wgrad_workspace_bytes(int64_t weight_elems) {
    return wgrad_workspace_bytes(weight_elems > 0 ? (long)weight_elems : 0);
}

extern "C" int32_t pe_mul_f32(float* a, const float* b, int64_t n, pe_stream_t stream) {
    PE_REQUIRE(n >= 0, PE_E_INVALID, "pe_mul_f32: negative n");
    if (n == 0) return PE_OK;
    PE_REQUIRE(a && b, PE_E_INVALID, "pe_mul_f32: NULL pointer");
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    pe::launch(mul_f32_kernel, (unsigned)blocks, 256, 0, as_stream(stream), a, b, n);
    return check_launch("pe_mul_f32");
}
