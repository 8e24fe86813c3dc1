// Batch normalisation for training, over NHWC bf16 tensors viewed as [M][C] (M = B*H*W).
//
// Bound: HBM bandwidth.  Forward: one reduction pass over z (sum, sum of squares per channel),
// a finalize launch, one element-wise pass (normalise + residual + ReLU).  Backward: one
// reduction pass (sum g, sum g*z with g = da masked by the ReLU), a finalize launch, one
// element-wise pass producing dz (and g itself when the residual branch needs it).
// Reductions are two-stage and ordered (per-block partials in the workspace, folded by one
// workgroup), so results are bit-stable from run to run.  Every access is 16 bytes.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "pe_common.h"

namespace pe {

constexpr int kBnMaxBlocks = 1024;

struct BnGeom {
    int cg;         // 16-byte channel groups per row (C / 8)
    int rl;         // rows a workgroup walks in parallel (256 / cg, rounded down: threads past
                    // rl * cg only help with the final fold)
    int rows_blk;   // rows per workgroup
    int nblk;       // workgroups
};

static BnGeom bn_geom(long M, int C) {
    BnGeom g;
    g.cg = C / 8;
    g.rl = 256 / g.cg;
    // at least 16 rows per walking thread: fewer partials for the second stage to fold
    long rows = (M + kBnMaxBlocks - 1) / kBnMaxBlocks;
    rows = (rows + g.rl - 1) / g.rl * g.rl;
    if (rows < (long)g.rl * 16) rows = (long)g.rl * 16;
    g.rows_blk = (int)rows;
    g.nblk = (int)((M + rows - 1) / rows);
    return g;
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_bits_to_float(u[i] & 0xffffu);
        f[2 * i + 1] = bf16_bits_to_float(u[i] >> 16);
    }
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 o;
    o.x = pack_bf16x2_bits(f[0], f[1]);
    o.y = pack_bf16x2_bits(f[2], f[3]);
    o.z = pack_bf16x2_bits(f[4], f[5]);
    o.w = pack_bf16x2_bits(f[6], f[7]);
    return o;
}

// The bf16 activation bn_apply stored for a layer WITHOUT a residual, recomputed from z: the same
// float operations (multiply, add, clamp) and the same rounding, so masks derived from it equal
// masks derived from the stored tensor bit for bit and backward need not read that tensor.
__device__ __forceinline__ void recompute_act8(const float* zf, const float* __restrict__ msc,
                                               const float* __restrict__ msh, int c0, float act_hi,
                                               float* af) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = zf[i] * msc[c0 + i] + msh[c0 + i];
        v = fmaxf(v, 0.0f);
        if (act_hi == 6.0f) v = fminf(v, 6.0f);
        af[i] = bf16_bits_to_float(float_to_bf16_bits(v));
    }
}

// MODE 0: s1 = sum z, s2 = sum z*z.   MODE 1: g = da * (0 < a < act_hi if a given); s1 = sum g,
// s2 = sum g*z.  act_hi = +inf for ReLU, 6 for ReLU6.  partial layout: [nblk][2][C].
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const uint16_t* __restrict__ z,
                                                        const uint16_t* __restrict__ da,
                                                        const uint16_t* __restrict__ a,
                                                        const float* __restrict__ msc,
                                                        const float* __restrict__ msh,
                                                        float act_hi, long M, int C, BnGeom g,
                                                        float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [rl][2][C]
    const int tid = threadIdx.x;
    const int cgi = tid % g.cg;
    const int rli = tid / g.cg;
    const bool walker = rli < g.rl;
    const long row0 = (long)blockIdx.x * g.rows_blk;
    long row1 = row0 + g.rows_blk;
    if (row1 > M) row1 = M;
    if (!walker) row1 = row0;  // no rows for the spare threads
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.0f;
    // a workgroup walks its rows U at a time: with at most 1024 workgroups on 256 CUs, one
    // 16-byte load per thread in flight is half of what HBM needs; U loads are issued together
    // (rows past the end re-read the last row and are masked out of the sums)
    constexpr int U = MODE == 0 ? 4 : 2;
    for (long r = row0 + rli; r < row1; r += (long)U * g.rl) {
        uint4 zv[U], gv[U], av[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + (long)u * g.rl;
            live[u] = rr < row1;
            const long off = (live[u] ? rr : r) * C + cgi * 8;
            zv[u] = *reinterpret_cast<const uint4*>(z + off);
            if (MODE == 1) {
                gv[u] = *reinterpret_cast<const uint4*>(da + off);
                if (a) av[u] = *reinterpret_cast<const uint4*>(a + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            float zf[8];
            unpack8(zv[u], zf);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s1[i] += zf[i];
                    s2[i] += zf[i] * zf[i];
                }
            } else {
                float gf[8];
                unpack8(gv[u], gf);
                if (a || msc) {
                    float af[8];
                    if (a) unpack8(av[u], af);
                    else recompute_act8(zf, msc, msh, cgi * 8, act_hi, af);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        gf[i] = (af[i] > 0.0f && af[i] < act_hi) ? gf[i] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s1[i] += gf[i];
                    s2[i] += gf[i] * zf[i];
                }
            }
        }
    }
    if (walker) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[(rli * 2 + 0) * C + cgi * 8 + i] = s1[i];
            red[(rli * 2 + 1) * C + cgi * 8 + i] = s2[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * C; e += 256) {
        float t = 0.0f;
        for (int r = 0; r < g.rl; ++r) t += red[r * 2 * C + e];
        partial[(long)blockIdx.x * 2 * C + e] = t;
    }
}

// Second reduction stage.  A workgroup folds the per-block partials of 4 channels with all of
// its 256 threads: thread t adds blocks t, t+256, ... eight at a time (the stage is bound by
// latency, so every load a thread will ever need is in flight after one or two rounds: a 16-byte
// load per sum covers the 4 channels), the 64 lane sums of a wave are added by a butterfly of
// lane exchanges and the four wave sums through LDS by threads 0..3, which return their
// channel (every other thread returns -1).  The order of additions is fixed by (nblk, C) alone.
// Launch with C/4 workgroups of 256 threads; C is a multiple of 8 and the rows 16-byte aligned.
constexpr int kFoldCh = 4;
constexpr int kFoldUnroll = 8;

__device__ __forceinline__ float wave_butterfly_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // same value in every lane
    return v;
}

__device__ __forceinline__ int fold_partials(const float* __restrict__ partial, int nblk, int C,
                                             float& s1, float& s2) {
    __shared__ float red[4][2 * kFoldCh];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * kFoldCh;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
    for (int b0 = tid; b0 < nblk; b0 += 256 * kFoldUnroll) {
        f32x4 v1[kFoldUnroll], v2[kFoldUnroll];
#pragma unroll
        for (int u = 0; u < kFoldUnroll; ++u) {
            const int b = b0 + u * 256;
            const bool ok = b < nblk;
            const float* row = partial + (long)(ok ? b : b0) * 2 * C + c0;
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(row);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(row + C);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            v1[u] = ok ? x1 : zero;
            v2[u] = ok ? x2 : zero;
        }
        a1 += ((v1[0] + v1[1]) + (v1[2] + v1[3])) + ((v1[4] + v1[5]) + (v1[6] + v1[7]));
        a2 += ((v2[0] + v2[1]) + (v2[2] + v2[3])) + ((v2[4] + v2[5]) + (v2[6] + v2[7]));
    }
    const float w[2 * kFoldCh] = {wave_butterfly_sum(a1.x), wave_butterfly_sum(a1.y),
                                  wave_butterfly_sum(a1.z), wave_butterfly_sum(a1.w),
                                  wave_butterfly_sum(a2.x), wave_butterfly_sum(a2.y),
                                  wave_butterfly_sum(a2.z), wave_butterfly_sum(a2.w)};
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 2 * kFoldCh; ++i) red[tid >> 6][i] = w[i];
    }
    __syncthreads();
    if (tid >= kFoldCh) return -1;
    s1 = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    s2 = (red[0][kFoldCh + tid] + red[1][kFoldCh + tid]) +
         (red[2][kFoldCh + tid] + red[3][kFoldCh + tid]);
    return c0 + tid;
}

struct BnFwdOut {
    float *mean, *rstd, *scale, *shift, *running_mean, *running_var;
};

__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* __restrict__ partial,
                                                              int nblk, int C, float inv_m,
                                                              float unbias, const float* gamma,
                                                              const float* beta, float eps,
                                                              float momentum, BnFwdOut o) {
    // the per-channel parameters are requested before the fold, so they arrive with the rows
    const int cp = blockIdx.x * kFoldCh + (threadIdx.x & (kFoldCh - 1));
    const float ga = gamma[cp], be = beta[cp];
    const float rm = o.running_mean ? o.running_mean[cp] : 0.0f;
    const float rv = o.running_mean ? o.running_var[cp] : 0.0f;
    float s1, s2;
    const int c = fold_partials(partial, nblk, C, s1, s2);
    if (c >= 0) {   // c == cp for the threads that get here
        const float mean = s1 * inv_m;
        float var = s2 * inv_m - mean * mean;
        var = fmaxf(var, 0.0f);
        const float rstd = 1.0f / sqrtf(var + eps);
        const float sc = ga * rstd;
        o.mean[c] = mean;
        o.rstd[c] = rstd;
        o.scale[c] = sc;
        o.shift[c] = be - mean * sc;
        if (o.running_mean) {
            o.running_mean[c] = (1.0f - momentum) * rm + momentum * mean;
            o.running_var[c] = (1.0f - momentum) * rv + momentum * (var * unbias);
        }
    }
}

// dz = A*(g - Bc - (z - mean)*Cc) with A = gamma*rstd, Bc = sum(g)/M, Cc = dgamma*rstd/M.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                              int nblk, int C, float inv_m,
                                                              const float* gamma,
                                                              const float* mean,
                                                              const float* rstd, float* dgamma,
                                                              float* dbeta, float* coefA,
                                                              float* coefB, float* coefC,
                                                              int frozen) {
    // the per-channel parameters are requested before the fold, so they arrive with the rows
    const int cp = blockIdx.x * kFoldCh + (threadIdx.x & (kFoldCh - 1));
    const float ga = gamma[cp], mu = mean[cp], rs = rstd[cp];
    float s1, s2;
    const int c = fold_partials(partial, nblk, C, s1, s2);
    if (c >= 0) {   // c == cp for the threads that get here
        const float dg = (s2 - mu * s1) * rs;
        dgamma[c] = dg;
        dbeta[c] = s1;
        coefA[c] = ga * rs;
        // frozen statistics: mean and rstd are constants of the layer, so the two terms that
        // carry the gradient through them vanish and dz = gamma * rstd * g
        coefB[c] = frozen ? 0.0f : s1 * inv_m;
        coefC[c] = frozen ? 0.0f : dg * rs * inv_m;
    }
}

// Folded constants of every batch-norm layer of a network in one launch: workgroup j walks the
// channels of job j.
__global__ __launch_bounds__(256) void bn_fold_kernel(const pe_bn_fold_job* __restrict__ jobs,
                                                      float eps) {
    const pe_bn_fold_job j = jobs[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += 256) {
        const float mean = j.running_mean[c];
        const float rstd = 1.0f / sqrtf(j.running_var[c] + eps);
        const float sc = j.gamma[c] * rstd;
        j.scale[c] = sc;
        j.shift[c] = j.beta[c] - mean * sc;
        if (j.mean) j.mean[c] = mean;
        if (j.rstd) j.rstd[c] = rstd;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const uint16_t* __restrict__ z, long nchunk,
                                                       int cg, const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const uint16_t* __restrict__ res, int relu,
                                                       uint16_t* __restrict__ y) {
    // the chunk's channel group is carried along the grid-stride loop (one 64-bit modulo per
    // thread instead of one per chunk)
    const int cstep = (int)(((long)gridDim.x * 256) % cg);
    int cgi = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (long)gridDim.x * 256) {
        const int c0 = cgi * 8;
        cgi += cstep;
        cgi -= cgi >= cg ? cg : 0;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(z + i * 8), v);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(scale + c0 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + c0);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(shift + c0 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
        if (res) {
            float r[8];
            unpack8(*reinterpret_cast<const uint4*>(res + i * 8), r);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += r[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.0f);
            if (relu == 2) {  // ReLU6
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fminf(v[k], 6.0f);
            }
        }
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(v);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const uint16_t* __restrict__ da, const uint16_t* __restrict__ a, const uint16_t* __restrict__ z,
    long nchunk, int cg, const float* __restrict__ mean, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, float act_hi,
    const float* __restrict__ msc, const float* __restrict__ msh,
    uint16_t* __restrict__ dz, uint16_t* __restrict__ dmasked) {
    const int cstep = (int)(((long)gridDim.x * 256) % cg);
    int cgi = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (long)gridDim.x * 256) {
        const int c0 = cgi * 8;
        cgi += cstep;
        cgi -= cgi >= cg ? cg : 0;
        float g[8], zf[8];
        unpack8(*reinterpret_cast<const uint4*>(da + i * 8), g);
        unpack8(*reinterpret_cast<const uint4*>(z + i * 8), zf);
        if (a || msc) {
            float af[8];
            if (a) unpack8(*reinterpret_cast<const uint4*>(a + i * 8), af);
            else recompute_act8(zf, msc, msh, c0, act_hi, af);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = (af[k] > 0.0f && af[k] < act_hi) ? g[k] : 0.0f;
        }
        if (dmasked) *reinterpret_cast<uint4*>(dmasked + i * 8) = pack8(g);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            o[k] = coefA[c] * (g[k] - coefB[c] - (zf[k] - mean[c]) * coefC[c]);
        }
        *reinterpret_cast<uint4*>(dz + i * 8) = pack8(o);
    }
}

// ---- finalize folded into the element-wise pass ---------------------------------------------
// One launch: workgroups [0, nfold) fold the partial rows of 4 channels each and publish the
// per-channel coefficients; the other workgroups prefetch their first chunk, wait for the
// publication and run the element-wise pass.  The hand-off inside the launch follows the
// agent-scope rules: every coefficient is written with an agent-scope atomic store (write-
// through), each storing wave drains its stores, the workgroup's barrier, then the workgroup
// adds to the arrival counter; a consumer polls ONE word relaxed, then one agent-scope acquire
// (drops the CU's stale lines), then plain loads.
// The counter is kept in kBnReplicas copies on cache lines of their own (a fold workgroup adds
// to every copy with one wave instruction, a consumer polls copy blockIdx % kBnReplicas), so
// a couple of thousand polling workgroups do not queue on one address.  It is never reset: the
// library remembers on the host how far each sync buffer has counted and hands the launch its
// starting value (`base`).  sync[kBnTimeoutWord] counts waits that timed out (never expected;
// such a workgroup folds for itself).
constexpr int kBnReplicas = 16;
constexpr int kBnReplicaStride = 32;                           // ints: one 128-byte line each
constexpr int kBnTimeoutWord = kBnReplicas * kBnReplicaStride;
constexpr int kBnSyncInts = kBnTimeoutWord + kBnReplicaStride; // what pe_bn_sync_bytes reports
constexpr unsigned kBnSpinLimit = 1u << 15;

__device__ __forceinline__ void publish_f32(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by every thread of a fold workgroup after its publish_f32 stores.
__device__ __forceinline__ void signal_fold_done(int* sync) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    // relaxed: the coefficients went out write-through and are drained, so no release (which
    // would write back the whole L2, full of the previous launch's output) is needed
    if (threadIdx.x < kBnReplicas)
        __hip_atomic_fetch_add(&sync[threadIdx.x * kBnReplicaStride], 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// Called by every thread of a consumer workgroup.  Returns false if the wait timed out.
__device__ __forceinline__ bool wait_fold_done(int* sync, unsigned base, int nfold) {
    __shared__ int ok_flag;
    if (threadIdx.x == 0) {
        const int* word = sync + (blockIdx.x % kBnReplicas) * kBnReplicaStride;
        unsigned spins = 0;
        int ok = 1;
        while ((unsigned)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                   base < (unsigned)nfold) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins >= kBnSpinLimit) {
                ok = 0;
                __hip_atomic_fetch_add(&sync[kBnTimeoutWord], 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        ok_flag = ok;
    }
    __syncthreads();   // every wave loads the coefficients behind this barrier
    return ok_flag != 0;
}

// 8 published floats (32-byte aligned) by two 16-byte loads that bypass the CU's L1: the form a
// consumer may use in place of an agent-scope acquire, whose L1 invalidate costs tens of
// microseconds with a few thousand workgroups in the launch.
__device__ __forceinline__ void load_published8(const float* p, float* out) {
    f32x4 v0, v1;
    asm volatile(
        "global_load_dwordx4 %0, %2, off sc1\n\t"
        "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1)
        : "v"(p)
        : "memory");
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w;
    out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
}

// Serial fold of one channel's partial rows (the time-out path only).
__device__ __forceinline__ void fold_one_channel(const float* __restrict__ partial, int nblk,
                                                 int C, int c, float& s1, float& s2) {
    s1 = 0.0f;
    s2 = 0.0f;
    for (int b = 0; b < nblk; ++b) {
        s1 += partial[(long)b * 2 * C + c];
        s2 += partial[(long)b * 2 * C + C + c];
    }
}

__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(
    const float* __restrict__ partial, int nblk, int C, float inv_m, float unbias,
    const float* gamma, const float* beta, float eps, float momentum, BnFwdOut o, int nfold,
    int* sync, unsigned base, const uint16_t* __restrict__ z, long nchunk, int cg,
    const uint16_t* __restrict__ res, int relu, uint16_t* __restrict__ y) {
    const int total = gridDim.x;
    if ((int)blockIdx.x < nfold) {
        float s1, s2;
        const int c = fold_partials(partial, nblk, C, s1, s2);
        if (c >= 0) {
            const float mean = s1 * inv_m;
            float var = s2 * inv_m - mean * mean;
            var = fmaxf(var, 0.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            const float sc = gamma[c] * rstd;
            // the four rows may share cache lines (small C): every store to a line that holds
            // published words is a write-through store, so no XCD's L2 keeps a partial copy
            publish_f32(o.mean + c, mean);
            publish_f32(o.rstd + c, rstd);
            publish_f32(o.scale + c, sc);
            publish_f32(o.shift + c, beta[c] - mean * sc);
            if (o.running_mean) {
                o.running_mean[c] = (1.0f - momentum) * o.running_mean[c] + momentum * mean;
                o.running_var[c] = (1.0f - momentum) * o.running_var[c] + momentum * (var * unbias);
            }
        }
        signal_fold_done(sync);
        return;
    }
    // the launch makes (total - nfold) * 256 a multiple of cg, so a thread keeps its channels
    const long stride = (long)(total - nfold) * 256;
    long i = (long)((int)blockIdx.x - nfold) * 256 + threadIdx.x;
    const int c0 = (int)(i % cg) * 8;
    // the first chunk is on its way while the fold workgroups work
    uint4 zv = make_uint4(0u, 0u, 0u, 0u), rv = make_uint4(0u, 0u, 0u, 0u);
    if (i < nchunk) {
        zv = *reinterpret_cast<const uint4*>(z + i * 8);
        if (res) rv = *reinterpret_cast<const uint4*>(res + i * 8);
    }
    // a block's 256 consecutive chunks cover at most min(cg, 256) channel groups: its first
    // threads fetch each group's coefficients once and hand them on through LDS
    __shared__ float lco[2][256][8];
    const int ngrp = cg < 256 ? cg : 256;
    const int slot = threadIdx.x % ngrp;
    float sc[8], sh[8];
    if (wait_fold_done(sync, base, nfold)) {
        if ((int)threadIdx.x < ngrp) {
            load_published8(o.scale + c0, sc);
            load_published8(o.shift + c0, sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lco[0][slot][k] = sc[k];
                lco[1][slot][k] = sh[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sc[k] = lco[0][slot][k];
            sh[k] = lco[1][slot][k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s1, s2;
            fold_one_channel(partial, nblk, C, c0 + k, s1, s2);
            const float mean = s1 * inv_m;
            const float var = fmaxf(s2 * inv_m - mean * mean, 0.0f);
            sc[k] = gamma[c0 + k] * (1.0f / sqrtf(var + eps));
            sh[k] = beta[c0 + k] - mean * sc[k];
        }
    }
    for (; i < nchunk; i += stride) {
        float v[8];
        unpack8(zv, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
        if (res) {
            float r[8];
            unpack8(rv, r);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += r[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.0f);
            if (relu == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fminf(v[k], 6.0f);
            }
        }
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(v);
        const long nx = i + stride;
        if (nx < nchunk) {
            zv = *reinterpret_cast<const uint4*>(z + nx * 8);
            if (res) rv = *reinterpret_cast<const uint4*>(res + nx * 8);
        }
    }
}

// Backward twin: the fold workgroups produce dgamma / dbeta and the three coefficient rows, the
// others the element-wise pass dz = A*(g - B - (z - mean)*C) (see bn_bwd_apply_kernel).
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(
    const float* __restrict__ partial, int nblk, int C, float inv_m, const float* gamma,
    const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coefA, float* coefB,
    float* coefC, int frozen, int nfold, int* sync, unsigned base,
    const uint16_t* __restrict__ da, const uint16_t* __restrict__ a,
    const uint16_t* __restrict__ z, long nchunk, int cg, float act_hi,
    const float* __restrict__ msc, const float* __restrict__ msh, uint16_t* __restrict__ dz,
    uint16_t* __restrict__ dmasked) {
    const int total = gridDim.x;
    if ((int)blockIdx.x < nfold) {
        float s1, s2;
        const int c = fold_partials(partial, nblk, C, s1, s2);
        if (c >= 0) {
            const float dg = (s2 - mean[c] * s1) * rstd[c];
            dgamma[c] = dg;
            dbeta[c] = s1;
            publish_f32(coefA + c, gamma[c] * rstd[c]);
            publish_f32(coefB + c, frozen ? 0.0f : s1 * inv_m);
            publish_f32(coefC + c, frozen ? 0.0f : dg * rstd[c] * inv_m);
        }
        signal_fold_done(sync);
        return;
    }
    const long stride = (long)(total - nfold) * 256;   // a multiple of cg: channels stay put
    long i = (long)((int)blockIdx.x - nfold) * 256 + threadIdx.x;
    const int c0 = (int)(i % cg) * 8;
    uint4 gv = make_uint4(0u, 0u, 0u, 0u), zv = gv, av = gv;
    if (i < nchunk) {
        gv = *reinterpret_cast<const uint4*>(da + i * 8);
        zv = *reinterpret_cast<const uint4*>(z + i * 8);
        if (a) av = *reinterpret_cast<const uint4*>(a + i * 8);
    }
    float cA[8], cB[8], cC[8], cM[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cM[k] = mean[c0 + k];   // written by an earlier launch
    __shared__ float lco[3][256][8];
    const int ngrp = cg < 256 ? cg : 256;
    const int slot = threadIdx.x % ngrp;
    if (wait_fold_done(sync, base, nfold)) {
        if ((int)threadIdx.x < ngrp) {
            load_published8(coefA + c0, cA);
            load_published8(coefB + c0, cB);
            load_published8(coefC + c0, cC);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lco[0][slot][k] = cA[k];
                lco[1][slot][k] = cB[k];
                lco[2][slot][k] = cC[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cA[k] = lco[0][slot][k];
            cB[k] = lco[1][slot][k];
            cC[k] = lco[2][slot][k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            float s1, s2;
            fold_one_channel(partial, nblk, C, c, s1, s2);
            const float dg = (s2 - mean[c] * s1) * rstd[c];
            cA[k] = gamma[c] * rstd[c];
            cB[k] = frozen ? 0.0f : s1 * inv_m;
            cC[k] = frozen ? 0.0f : dg * rstd[c] * inv_m;
        }
    }
    for (; i < nchunk; i += stride) {
        float g[8], zf[8];
        unpack8(gv, g);
        unpack8(zv, zf);
        if (a || msc) {
            float af[8];
            if (a) unpack8(av, af);
            else recompute_act8(zf, msc, msh, c0, act_hi, af);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = (af[k] > 0.0f && af[k] < act_hi) ? g[k] : 0.0f;
        }
        if (dmasked) *reinterpret_cast<uint4*>(dmasked + i * 8) = pack8(g);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = cA[k] * (g[k] - cB[k] - (zf[k] - cM[k]) * cC[k]);
        *reinterpret_cast<uint4*>(dz + i * 8) = pack8(o);
        const long nx = i + stride;
        if (nx < nchunk) {
            gv = *reinterpret_cast<const uint4*>(da + nx * 8);
            zv = *reinterpret_cast<const uint4*>(z + nx * 8);
            if (a) av = *reinterpret_cast<const uint4*>(a + nx * 8);
        }
    }
}

static unsigned ew_grid(long nchunk) {
    long b = (nchunk + 255) / 256;
    // one 16-byte chunk per thread up to 128 workgroups per CU, a grid-stride loop beyond: with
    // the cap at 16 per CU the large MobileNetV2 tensors lost 2 % of a training step
    if (b > 256 * 128) b = 256 * 128;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// first 128-byte boundary at or after p
static float* coef_rows(float* p) {
    return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(p) + 127) & ~(uintptr_t)127);
}

static int32_t bn_check(const char* what, long M, int C) {
    PE_REQUIRE(M > 0 && C > 0, PE_E_INVALID, "%s: M and C must be positive", what);
    PE_REQUIRE(C % 8 == 0 && C <= 2048, PE_E_UNSUPPORTED,
               "%s: C must be a multiple of 8, at most 2048 (C = %d)", what, C);
    return PE_OK;
}

}  // namespace pe

using namespace pe;

extern "C" size_t pe_bn_workspace_bytes(int64_t M, int32_t C) {
    if (M <= 0 || C <= 0 || C % 8 != 0 || C > 2048) return 16;
    // per-block partials (or one row per 64-pixel conv tile when the statistics come from the
    // conv epilogue) + three per-channel coefficient rows for the backward pass
    size_t rows = (size_t)(M / 64) + 8;
    if (rows < (size_t)kBnMaxBlocks) rows = kBnMaxBlocks;
    return (rows * 2 + 3) * (size_t)C * sizeof(float) + 256;   // + alignment of the coefficients
}

extern "C" int32_t pe_bn_finalize_bf16(const void* workspace, int32_t rows, int64_t M, int32_t C,
                                       const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var,
                                       float* mean, float* rstd, float* scale, float* shift,
                                       pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_finalize_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(workspace && gamma && beta && mean && rstd && scale && shift && rows > 0,
               PE_E_INVALID, "pe_bn_finalize_bf16: NULL pointer or no rows");
    PE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), PE_E_INVALID,
               "pe_bn_finalize_bf16: give both running buffers or neither");
    PE_REQUIRE(aligned(workspace, 16), PE_E_INVALID,
               "pe_bn_finalize_bf16: workspace must be 16-byte aligned");
    const BnFwdOut o{mean, rstd, scale, shift, running_mean, running_var};
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    pe::launch(bn_fwd_finalize_kernel, (C + kFoldCh - 1) / kFoldCh, 256, 0, as_stream(stream), static_cast<const float*>(workspace), rows, C, (float)(1.0 / (double)M), unbias, gamma, beta,
        eps, momentum, o);
    return check_launch("pe_bn_finalize_bf16");
}

// How far each sync buffer has counted (see the kernels' comment): the next launch on it starts
// at this value.  Launches that share a buffer are issued from one stream, in call order.
static std::mutex g_sync_mutex;
static std::unordered_map<const void*, uint32_t> g_sync_count;

static uint32_t sync_take(const void* sync, int nfold) {
    std::lock_guard<std::mutex> lock(g_sync_mutex);
    uint32_t& c = g_sync_count[sync];
    const uint32_t base = c;
    c += (uint32_t)nfold;
    return base;
}

// grid of a fused launch: the fold workgroups first, then the element-wise ones
static unsigned fused_apply_blocks(long nchunk, int cg) {
    long b = (nchunk + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    // blocks * 256 must be a multiple of cg so that a thread's channel group never changes
    int g = cg, t = 256;
    while (t) { const int r = g % t; g = t; t = r; }   // g = gcd(cg, 256)
    const long unit = cg / g;
    b = b / unit * unit;
    if (b < unit) b = unit;
    return (unsigned)b;
}

extern "C" int32_t pe_bn_finalize_apply_bf16(const void* workspace, int32_t rows, const void* z,
                                             int64_t M, int32_t C, const float* gamma,
                                             const float* beta, float eps, float momentum,
                                             float* running_mean, float* running_var,
                                             float* mean, float* rstd, float* scale,
                                             float* shift, const void* residual, int32_t relu,
                                             void* y, int32_t* sync, pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_finalize_apply_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(workspace && z && y && gamma && beta && mean && rstd && scale && shift && sync &&
                   rows > 0,
               PE_E_INVALID, "pe_bn_finalize_apply_bf16: NULL pointer or no rows");
    PE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), PE_E_INVALID,
               "pe_bn_finalize_apply_bf16: give both running buffers or neither");
    PE_REQUIRE(aligned(z, 16) && aligned(y, 16) && aligned(residual, 16) && aligned(sync, 128) &&
                   aligned(scale, 32) && aligned(shift, 32) && aligned(workspace, 16),
               PE_E_INVALID,
               "pe_bn_finalize_apply_bf16: tensors must be 16-byte aligned, scale / shift 32-byte, "
               "sync 128-byte");
    // The arrival-counter base handed to the kernel comes from a host-side counter that
    // advances per call; a launch replayed from a tape would carry a stale base and its
    // consumer workgroups would not wait for the fold.
    PE_REQUIRE(!pe::tape_recording(), PE_E_UNSUPPORTED,
               "%s: the one-launch (sync) form cannot be recorded on a launch tape", "pe_bn_finalize_apply_bf16");
    const BnFwdOut o{mean, rstd, scale, shift, running_mean, running_var};
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    const int nfold = (C + kFoldCh - 1) / kFoldCh;
    const long nchunk = M * (C / 8);
    const unsigned grid = (unsigned)nfold + fused_apply_blocks(nchunk, C / 8);
    pe::launch(bn_fwd_fused_kernel, grid, 256, 0, as_stream(stream), static_cast<const float*>(workspace), rows, C, (float)(1.0 / (double)M), unbias, gamma, beta,
        eps, momentum, o, nfold, sync, sync_take(sync, nfold), static_cast<const uint16_t*>(z),
        nchunk, C / 8,
        static_cast<const uint16_t*>(residual), relu, static_cast<uint16_t*>(y));
    return check_launch("pe_bn_finalize_apply_bf16");
}

extern "C" int32_t pe_bn_stats_bf16(const void* z, int64_t M, int32_t C, const float* gamma,
                                    const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* mean,
                                    float* rstd, float* scale, float* shift, void* workspace,
                                    pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_stats_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(z && gamma && beta && mean && rstd && scale && shift && workspace, PE_E_INVALID,
               "pe_bn_stats_bf16: NULL pointer");
    PE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), PE_E_INVALID,
               "pe_bn_stats_bf16: give both running buffers or neither");
    PE_REQUIRE(aligned(z, 16) && aligned(workspace, 16), PE_E_INVALID,
               "pe_bn_stats_bf16: z and workspace must be 16-byte aligned");
    const BnGeom g = bn_geom(M, C);
    float* partial = static_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    pe::launch(bn_reduce_kernel<0>, g.nblk, 256, (size_t)g.rl * 2 * C * sizeof(float), st, static_cast<const uint16_t*>(z), nullptr, nullptr, nullptr, nullptr, 0.0f, M, C, g, partial);
    rc = check_launch("pe_bn_stats_bf16(reduce)");
    if (rc != PE_OK) return rc;
    const BnFwdOut o{mean, rstd, scale, shift, running_mean, running_var};
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    pe::launch(bn_fwd_finalize_kernel, (C + kFoldCh - 1) / kFoldCh, 256, 0, st, partial, g.nblk, C,
                                                           (float)(1.0 / (double)M), unbias, gamma,
                                                           beta, eps, momentum, o);
    return check_launch("pe_bn_stats_bf16(finalize)");
}

extern "C" int32_t pe_bn_stats_apply_bf16(const void* z, int64_t M, int32_t C,
                                          const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean,
                                          float* running_var, float* mean, float* rstd,
                                          float* scale, float* shift, const void* residual,
                                          int32_t relu, void* y, void* workspace, int32_t* sync,
                                          pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_stats_apply_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(z && workspace, PE_E_INVALID, "pe_bn_stats_apply_bf16: NULL pointer");
    PE_REQUIRE(aligned(z, 16) && aligned(workspace, 16), PE_E_INVALID,
               "pe_bn_stats_apply_bf16: z and workspace must be 16-byte aligned");
    PE_REQUIRE(!pe::tape_recording(), PE_E_UNSUPPORTED,
               "pe_bn_stats_apply_bf16: the one-launch (sync) form cannot be recorded on a "
               "launch tape");
    const BnGeom g = bn_geom(M, C);
    pe::launch(bn_reduce_kernel<0>, g.nblk, 256, (size_t)g.rl * 2 * C * sizeof(float), as_stream(stream), static_cast<const uint16_t*>(z), nullptr, nullptr, nullptr, nullptr, 0.0f, M, C, g,
        static_cast<float*>(workspace));
    rc = check_launch("pe_bn_stats_apply_bf16(reduce)");
    if (rc != PE_OK) return rc;
    return pe_bn_finalize_apply_bf16(workspace, g.nblk, z, M, C, gamma, beta, eps, momentum,
                                     running_mean, running_var, mean, rstd, scale, shift, residual,
                                     relu, y, sync, stream);
}

extern "C" int32_t pe_bn_apply_bf16(const void* z, int64_t M, int32_t C, const float* scale,
                                    const float* shift, const void* residual, int32_t relu,
                                    void* y, pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_apply_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(z && scale && shift && y, PE_E_INVALID, "pe_bn_apply_bf16: NULL pointer");
    PE_REQUIRE(aligned(z, 16) && aligned(y, 16) && aligned(residual, 16) && aligned(scale, 16) &&
                   aligned(shift, 16),
               PE_E_INVALID, "pe_bn_apply_bf16: tensors must be 16-byte aligned");
    const long nchunk = M * (C / 8);
    pe::launch(bn_apply_kernel, ew_grid(nchunk), 256, 0, as_stream(stream), static_cast<const uint16_t*>(z), nchunk, C / 8, scale, shift,
        static_cast<const uint16_t*>(residual), relu, static_cast<uint16_t*>(y));
    return check_launch("pe_bn_apply_bf16");
}

extern "C" int32_t pe_bn_bwd_bf16(const void* da, const void* a, int32_t relu, const void* z,
                                  int64_t M, int32_t C, const float* gamma, const float* mean,
                                  const float* rstd, const float* act_scale,
                                  const float* act_shift, float* dgamma, float* dbeta, void* dz,
                                  void* dmasked, void* workspace, int32_t frozen,
                                  int32_t* sync, pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_bwd_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(da && z && gamma && mean && rstd && dgamma && dbeta && dz && workspace,
               PE_E_INVALID, "pe_bn_bwd_bf16: NULL pointer");
    PE_REQUIRE(aligned(da, 16) && aligned(a, 16) && aligned(z, 16) && aligned(dz, 16) &&
                   aligned(dmasked, 16) && aligned(workspace, 16),
               PE_E_INVALID, "pe_bn_bwd_bf16: tensors must be 16-byte aligned");
    PE_REQUIRE(relu == 1 || relu == 2, PE_E_INVALID,
               "pe_bn_bwd_bf16: relu must be 1 (ReLU) or 2 (ReLU6); it only matters when a is given");
    PE_REQUIRE((act_scale == nullptr) == (act_shift == nullptr) && !(a && act_scale), PE_E_INVALID,
               "pe_bn_bwd_bf16: give the stored activation a, or act_scale and act_shift, not both");
    PE_REQUIRE(sync == nullptr || !pe::tape_recording(), PE_E_UNSUPPORTED,
               "pe_bn_bwd_bf16: the one-launch (sync) form cannot be recorded on a launch tape");
    const float act_hi = relu == 2 ? 6.0f : INFINITY;
    const BnGeom g = bn_geom(M, C);
    float* partial = static_cast<float*>(workspace);
    // on cache lines of their own: no plain load of the launch may touch a published line
    float* coefA = coef_rows(partial + (size_t)kBnMaxBlocks * 2 * C);
    float* coefB = coefA + C;
    float* coefC = coefB + C;
    hipStream_t st = as_stream(stream);
    pe::launch(bn_reduce_kernel<1>, g.nblk, 256, (size_t)g.rl * 2 * C * sizeof(float), st, static_cast<const uint16_t*>(z), static_cast<const uint16_t*>(da),
        static_cast<const uint16_t*>(a), act_scale, act_shift, act_hi, M, C, g, partial);
    rc = check_launch("pe_bn_bwd_bf16(reduce)");
    if (rc != PE_OK) return rc;
    if (sync != nullptr) {   // finalize and element-wise pass in one launch
        const int nfold = (C + kFoldCh - 1) / kFoldCh;
        const long nch = M * (C / 8);
        pe::launch(bn_bwd_fused_kernel, (unsigned)nfold + fused_apply_blocks(nch, C / 8), 256, 0, st, partial, g.nblk, C, (float)(1.0 / (double)M), gamma, mean, rstd, dgamma, dbeta, coefA,
            coefB, coefC, frozen != 0, nfold, sync, sync_take(sync, nfold),
            static_cast<const uint16_t*>(da),
            static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(z), nch, C / 8, act_hi,
            act_scale, act_shift, static_cast<uint16_t*>(dz), static_cast<uint16_t*>(dmasked));
        return check_launch("pe_bn_bwd_bf16(fused)");
    }
    pe::launch(bn_bwd_finalize_kernel, (C + kFoldCh - 1) / kFoldCh, 256, 0, st, partial, g.nblk, C,
                                                           (float)(1.0 / (double)M), gamma, mean,
                                                           rstd, dgamma, dbeta, coefA, coefB, coefC,
                                                           frozen != 0);
    rc = check_launch("pe_bn_bwd_bf16(finalize)");
    if (rc != PE_OK) return rc;
    const long nchunk = M * (C / 8);
    pe::launch(bn_bwd_apply_kernel, ew_grid(nchunk), 256, 0, st, static_cast<const uint16_t*>(da), static_cast<const uint16_t*>(a),
        static_cast<const uint16_t*>(z), nchunk, C / 8, mean, coefA, coefB, coefC, act_hi,
        act_scale, act_shift, static_cast<uint16_t*>(dz), static_cast<uint16_t*>(dmasked));
    return check_launch("pe_bn_bwd_bf16(apply)");
}

extern "C" int32_t pe_bn_bwd_from_partials_bf16(const void* g, const void* z, int32_t rows,
                                                int64_t M, int32_t C, const float* gamma,
                                                const float* mean, const float* rstd,
                                                float* dgamma, float* dbeta, void* dz,
                                                void* workspace, int32_t frozen,
                                                int32_t* sync, pe_stream_t stream) {
    int32_t rc = bn_check("pe_bn_bwd_from_partials_bf16", M, C);
    if (rc != PE_OK) return rc;
    PE_REQUIRE(g && z && gamma && mean && rstd && dgamma && dbeta && dz && workspace && rows > 0,
               PE_E_INVALID, "pe_bn_bwd_from_partials_bf16: NULL pointer or no rows");
    PE_REQUIRE(aligned(g, 16) && aligned(z, 16) && aligned(dz, 16) && aligned(workspace, 16),
               PE_E_INVALID, "pe_bn_bwd_from_partials_bf16: tensors must be 16-byte aligned");
    const size_t cap = (pe_bn_workspace_bytes(M, C) - 256) / ((size_t)C * sizeof(float));
    PE_REQUIRE((size_t)rows * 2 + 3 <= cap, PE_E_INVALID,
               "pe_bn_bwd_from_partials_bf16: %d rows do not fit the workspace of (M, C)", rows);
    PE_REQUIRE(sync == nullptr || !pe::tape_recording(), PE_E_UNSUPPORTED,
               "pe_bn_bwd_from_partials_bf16: the one-launch (sync) form cannot be recorded on a "
               "launch tape");
    float* partial = static_cast<float*>(workspace);
    float* coefA = coef_rows(partial + (size_t)rows * 2 * C);   // behind the rows the conv wrote
    float* coefB = coefA + C;
    float* coefC = coefB + C;
    hipStream_t st = as_stream(stream);
    if (sync != nullptr) {   // one launch; g is already masked: no activation operand
        const int nfold = (C + kFoldCh - 1) / kFoldCh;
        const long nch = M * (C / 8);
        pe::launch(bn_bwd_fused_kernel, (unsigned)nfold + fused_apply_blocks(nch, C / 8), 256, 0, st, partial, rows, C, (float)(1.0 / (double)M), gamma, mean, rstd, dgamma, dbeta, coefA,
            coefB, coefC, frozen != 0, nfold, sync, sync_take(sync, nfold),
            static_cast<const uint16_t*>(g), nullptr,
            static_cast<const uint16_t*>(z), nch, C / 8, INFINITY, nullptr, nullptr,
            static_cast<uint16_t*>(dz), nullptr);
        return check_launch("pe_bn_bwd_from_partials_bf16(fused)");
    }
    pe::launch(bn_bwd_finalize_kernel, (C + kFoldCh - 1) / kFoldCh, 256, 0, st, partial, rows, C, (float)(1.0 / (double)M), gamma, mean, rstd, dgamma, dbeta, coefA, coefB,
        coefC, frozen != 0);
    rc = check_launch("pe_bn_bwd_from_partials_bf16(finalize)");
    if (rc != PE_OK) return rc;
    const long nchunk = M * (C / 8);
    // g is already masked: no activation operand
    pe::launch(bn_bwd_apply_kernel, ew_grid(nchunk), 256, 0, st, static_cast<const uint16_t*>(g), nullptr, static_cast<const uint16_t*>(z), nchunk, C / 8,
        mean, coefA, coefB, coefC, INFINITY, nullptr, nullptr, static_cast<uint16_t*>(dz), nullptr);
    return check_launch("pe_bn_bwd_from_partials_bf16(apply)");
}

extern "C" int32_t pe_bn_fold_f32(const pe_bn_fold_job* jobs_device, int32_t njobs, float eps,
                                  pe_stream_t stream) {
    if (njobs == 0) return PE_OK;
    PE_REQUIRE(jobs_device != nullptr && njobs > 0, PE_E_INVALID,
               "pe_bn_fold_f32: NULL job table or negative count");
    PE_REQUIRE(eps >= 0.0f && isfinite(eps), PE_E_INVALID, "pe_bn_fold_f32: eps must be >= 0");
    pe::launch(bn_fold_kernel, (unsigned)njobs, 256, 0, as_stream(stream), jobs_device, eps);
    return check_launch("pe_bn_fold_f32");
}

extern "C" size_t pe_bn_sync_bytes(void) { return (size_t)kBnSyncInts * sizeof(int32_t); }

extern "C" int32_t pe_bn_sync_init(int32_t* sync, pe_stream_t stream) {
    PE_REQUIRE(sync != nullptr && aligned(sync, 128), PE_E_INVALID,
               "pe_bn_sync_init: sync must be a 128-byte aligned device pointer");
    const hipError_t e = pe::memset_async(sync, 0, pe_bn_sync_bytes(), as_stream(stream));
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "pe_bn_sync_init: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_sync_mutex);
    g_sync_count[sync] = 0;
    return PE_OK;
}
