// Backward-weight (weight-gradient) GEMM for every conv on the path:
//
//   dW[n][tap*Cin + c] += sum_m dY[opix(m)][n] * X[ipix(m, tap)][c]
//
// The reduction runs over pixels (m), the strided axis of both NHWC operands, so tiles are
// staged row-major [m][channels] and the MFMA fragments are fetched with the transposing LDS
// read ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and each lane
// receives one column, i.e. four consecutive m of one channel.  Two such reads make the 8-deep
// K fragment of mfma_f32_16x16x32_bf16; the m order inside a fragment is (4g+j, 16+4g+j) for
// lane group g, identical for both operands.
//
// Workgroup tile: 128 (n) x 128 (tap-channel columns), K step 64 pixels.  Both tiles go
// global -> LDS with global_load_lds_dwordx4 into one buffer of unpadded 256-byte rows (a wave
// instruction writes 4 rows); 16-byte slot s of row r holds channel chunk s ^ (2*(r&7)), which
// puts the eight rows a 32-lane half of a transposing read touches on disjoint banks.  The
// swizzle is applied to the per-lane source address and again on the read.  Pixels outside
// the image or past the end of the split read a 16-byte block of zeros.  128 VGPRs and
// <= 34 KB of LDS: four workgroups per CU cover each other's load latency.
//
// The pixel range is split across gridDim.y.  Each split stores its partial tile to a slab of
// the workspace (LDS-staged so a wave writes 256 contiguous bytes) and wgrad_reduce_kernel adds
// the slabs to dW in a fixed order: no float atomics, so a step repeats bit for bit.
#include <stdlib.h>

#include <type_traits>

#include "pe_common.h"
#include "pe_wgrad.h"

namespace pe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int WG_BK = 64;      // pixels per K step
constexpr int WG_ROWB = 256;   // bytes per LDS row (128 bf16, unpadded)
constexpr int WG_TILE = WG_BK * WG_ROWB;

__device__ __attribute__((aligned(16))) uint32_t g_wzero16[4] = {0u, 0u, 0u, 0u};

// q = m / d, r = m % d for 0 <= m < 2^31 and quotients far below 2^20 (float reciprocal + fix-up)
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
    q = (int)((float)m * inv_d);
    r = m - q * d;
    const int fix = (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);  // branch-free: this runs every K step
    q += fix;
    r -= fix * d;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off_lo) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + off_lo));
    const s16x4 hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + off_lo + 16 * WG_ROWB));
    s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

// FLAT: a 1x1 stride-1 conv with an identity output grid, where input pixel = output pixel =
// GEMM row and the per-step index split disappears (two thirds of the network's layers).
template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS = 1: one LDS stage, load - barrier - multiply - barrier; four workgroups per CU cover each
// other's load latency.  NS >= 2: a ring of NS stages with the loads of the next NS-1 K steps in
// flight (counted vmcnt, one raw barrier per step), for launches that put only one or two
// workgroups on a CU: every pixel split costs a slab of the weight's size, so large weights run
// with few splits and must hide their own latency.
// `bid` of `nwg`: this workgroup's index among the workgroups of ONE weight gradient (the whole
// grid, or one job's share of a batched launch).
template <bool FLAT, int NS>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, int bid, const int nwg,
                                           char* smem) {
    constexpr int STAGE = 2 * WG_TILE;   // dy tile, then x tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wave_c = wid >> 1;  // column half of the output tile
    const int wave_n = wid & 1;   // n half

    const int ktot = p.T * p.Cin;
    const int ntn = (p.N + 127) / 128;
    // XCD-aware order: workgroups that share an XCD (ids congruent mod 8) take a contiguous run
    // of (split, tile) pairs, so the tiles of one pixel split, which all read the same dy and x
    // rows, find them in ONE L2 instead of pulling a copy into each of the eight
    {
        const int qq = nwg >> 3, r8 = nwg & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
    }
    const int split = bid / p.tiles;
    const int tile = bid - split * p.tiles;
    const int tile_c = tile / ntn;
    const int tile_n = tile - tile_c * ntn;
    const int n0 = tile_n * 128;
    const int c0 = tile_c * 128;

    const long M = (long)p.B * p.Ho * p.Wo;
    const long ms = (long)split * p.m_per_split;
    long me = ms + p.m_per_split;
    if (me > M) me = M;

    // staging: wave w, instruction i covers rows 16*i + 4*w .. +3; lane l -> row +(l>>4), slot
    // l&15, which must receive channel chunk (l&15) ^ (2*(row&7)); row&7 == (tid>>4)&7
    const int rr = tid >> 4;
    const int ls = (tid & 15) ^ (2 * (rr & 7));
    const int nn = n0 + ls * 8;
    const bool n_ok = nn < p.N;
    const int col = c0 + ls * 8;
    const bool c_ok = col < ktot;
    int tdy = 0, tdx = 0, xch = 0;
    if (c_ok) {
        const int tap = col / p.Cin;
        const int ci = col - tap * p.Cin;
        tdy = p.tap_dy[tap];
        tdx = p.tap_dx[tap] + (ci >> p.cshift);
        xch = ci & p.cmask;
    }
    const int HoWo = p.Ho * p.Wo;
    const int OHW = p.OH * p.OW;
    const float inv_howo = 1.0f / (float)HoWo;
    const float inv_wo = 1.0f / (float)p.Wo;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_wzero16);

    f32x4 acc[4][4];  // [column tile][n tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing-read geometry: lane 4q+pp of a 16-lane group g addresses row 4g+q (second
    // read: +16), 8 bytes at column 4pp of a 16-column tile (slot 2t + (pp>>1))
    const int g = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pp = lane & 3;
    const int frow = 4 * g + q;
    const int fsw = 2 * (frow & 7);
    const int fsub = (pp >> 1);
    const int fbyte = 8 * (pp & 1);
    int yoff[4], xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        yoff[t] = frow * WG_ROWB + (((2 * (wave_n * 4 + t) + fsub) ^ fsw) * 16) + fbyte;
        xoff[t] = frow * WG_ROWB + (((2 * (wave_c * 4 + t) + fsub) ^ fsw) * 16) + fbyte;
    }

    // Output-grid coordinates (image, row, column) of this thread's first staging row at the
    // first K step.  Its other rows lie 16 pixels apart and the next K step 64 pixels on: both
    // are reached by adding a constant (image, row, column) step with one carry per digit, so
    // the K loop holds no division (the two divisions per row and step it replaces cost more
    // vector-ALU time than the step's 32 MFMAs).  The host keeps B*Ho*Wo and B*Hi*Wi below 2^31.
    int cb = 0, coy = 0, cox = 0;
    int s16b = 0, s16y = 0, s16x = 0, s64b = 0, s64y = 0, s64x = 0;
    if constexpr (!FLAT) {
        int rem;
        fast_divmod((int)(ms + rr), HoWo, inv_howo, cb, rem);
        fast_divmod(rem, p.Wo, inv_wo, coy, cox);
        s16b = 16 / HoWo;
        s16y = (16 % HoWo) / p.Wo;
        s16x = (16 % HoWo) % p.Wo;
        s64b = 64 / HoWo;
        s64y = (64 % HoWo) / p.Wo;
        s64x = (64 % HoWo) % p.Wo;
    }
    auto advance = [&](int& b, int& oy, int& ox, int db, int dyy, int dxx) {
        ox += dxx;
        const int cx = ox >= p.Wo ? 1 : 0;
        ox -= cx ? p.Wo : 0;
        oy += dyy + cx;
        const int cy = oy >= p.Ho ? 1 : 0;
        oy -= cy ? p.Ho : 0;
        b += db + cy;
    };

    // issues the loads of the K step that starts at pixel mb into `stage`; K steps are issued
    // in order, one call per step
    auto issue = [&](char* stage, long mb) {
        char* ytile = stage;
        char* xtile = stage + WG_TILE;
        int b = cb, oy = coy, ox = cox;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = mb + rr + 16 * i;
            const bool mv = m < me;
            const uint16_t* ysrc;
            const uint16_t* xsrc;
            if constexpr (FLAT) {
                ysrc = (mv && n_ok) ? p.dy + m * p.ldy + nn : zero;
                xsrc = (mv && c_ok) ? p.x + m * p.Cx + xch : zero;
            } else {
                const int opix = b * OHW + (oy * p.osy + p.ooy) * p.OW + ox * p.osx + p.oox;
                ysrc = (mv && n_ok) ? p.dy + (long)opix * p.ldy + nn : zero;
                const int iy = oy * p.sy + tdy;
                const int ix = ox * p.sx + tdx;
                const bool inb = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                const int ipix = (b * p.Hi + iy) * p.Wi + ix;
                xsrc = (mv && c_ok && inb) ? p.x + ((long)ipix * p.Cx + xch) : zero;
                if (i < 3) advance(b, oy, ox, s16b, s16y, s16x);
            }
            __builtin_amdgcn_global_load_lds((gptr_t)ysrc,
                                             (lptr_t)(ytile + (16 * i + 4 * wid) * WG_ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc,
                                             (lptr_t)(xtile + (16 * i + 4 * wid) * WG_ROWB), 16, 0, 0);
        }
        if constexpr (!FLAT) advance(cb, coy, cox, s64b, s64y, s64x);
    };
    auto multiply = [&](const char* stage) {
        const char* ytile = stage;
        const char* xtile = stage + WG_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fx[4], fy[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fx[a] = tr_frag(xtile, ks * 32 * WG_ROWB + xoff[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) fy[b] = tr_frag(ytile, ks * 32 * WG_ROWB + yoff[b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[a], fy[b], acc[a][b], 0, 0, 0);
        }
    };

    // The ring's fragment reads are written in assembly: for the transposing-read intrinsic the
    // compiler waits for EVERY outstanding load (vmcnt(0)) before the first read of a K step,
    // which would drain the prefetched stages and serialise the ring.  An asm read is invisible
    // to that pass; its result is awaited by the explicit lgkmcnt wait below, and the scheduling
    // barrier keeps the multiplies behind that wait.
    auto tr_read = [](uint32_t addr, auto off) -> uint64_t {
        uint64_t v;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(decltype(off)::value));
        return v;
    };
    auto frag_of = [](uint64_t lo, uint64_t hi) -> bf16x8 {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 v = {lo, hi};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto multiply_ring = [&](uint32_t stage_addr) {
        // stage_addr: LDS byte address of the stage; dy tile first, x tile at +WG_TILE
#define PE_TR(BASE, OFF) tr_read((BASE), std::integral_constant<int, (OFF)>{})
#define PE_KS(KS)                                                                              \
        {                                                                                      \
            uint64_t xl[4], xh[4], yl[4], yh[4];                                               \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                    \
                xl[a] = PE_TR(stage_addr + xoff[a], WG_TILE + (KS) * 32 * WG_ROWB);            \
                xh[a] = PE_TR(stage_addr + xoff[a], WG_TILE + (KS) * 32 * WG_ROWB + 16 * WG_ROWB); \
            }                                                                                  \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                    \
                yl[b] = PE_TR(stage_addr + yoff[b], (KS) * 32 * WG_ROWB);                      \
                yh[b] = PE_TR(stage_addr + yoff[b], (KS) * 32 * WG_ROWB + 16 * WG_ROWB);       \
            }                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                      \
                _Pragma("unroll") for (int b = 0; b < 4; ++b)                                  \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                       \
                        frag_of(xl[a], xh[a]), frag_of(yl[b], yh[b]), acc[a][b], 0, 0, 0);     \
        }
        PE_KS(0)
        PE_KS(1)
#undef PE_KS
#undef PE_TR
    };

    if constexpr (NS >= 2) {
        // see the ring in pe_igemm.hip: a stage is read only after the counted wait that retires
        // its loads AND a barrier, and restaged only after a barrier that follows its last read
        constexpr int LPT = 8;   // staging instructions per thread and K step
        static_assert((NS - 2) * LPT <= 63, "too many loads in flight for the vmcnt field");
        const int KT = (int)((me - ms + WG_BK - 1) / WG_BK);
        const uint32_t smem_addr = (uint32_t)(uintptr_t)(lptr_t)smem;
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < KT) issue(smem + st * STAGE, ms + (long)st * WG_BK);
        int cur = 0, nxt = NS - 1;
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + NS - 1 <= KT) wg_wait_vmcnt<(NS - 2) * LPT>();
            else wg_wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < KT) issue(smem + nxt * STAGE, ms + (long)(kt + NS - 1) * WG_BK);
            multiply_ring(smem_addr + (uint32_t)(cur * STAGE));
            if (++cur == NS) cur = 0;
            if (++nxt == NS) nxt = 0;
        }
        __syncthreads();   // the epilogue reuses the stages
    } else {
        for (long mb = ms; mb < me; mb += WG_BK) {
            issue(smem, mb);
            __syncthreads();  // vmcnt drained by the compiler: tiles are in LDS
            multiply(smem);
            __syncthreads();  // tiles consumed before the next K step overwrites them
        }
    }

    // ---- epilogue: n halves through LDS [64 n][128 cols] (pitch 132 floats), then stores in
    // which a wave writes 64 consecutive floats of one row
    constexpr int CP = 132;
    float* cs = reinterpret_cast<float*>(smem);
    const int lr = lane & 15;
    const int cq = (lane >> 4) * 4;
    const int cl = tid & 127;
    const int colg = c0 + cl;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wave_n == half) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    *reinterpret_cast<f32x4*>(cs + (b * 16 + lr) * CP + wave_c * 64 + a * 16 + cq) =
                        acc[a][b];
        }
        __syncthreads();
        if (colg < ktot) {
            // one split: this workgroup owns the elements, add in place; several splits: each
            // writes its own slab with plain stores and wgrad_reduce_kernel folds them in order
            float* dst = p.splits == 1 ? p.dw : p.slab + (long)split * p.slab_stride;
            for (int nl = tid >> 7; nl < 64; nl += 2) {
                const int n = n0 + half * 64 + nl;
                if (n >= p.Nw) break;
                const long e = (long)n * ktot + colg;
                const float v = cs[nl * CP + cl];
                dst[e] = p.splits == 1 ? dst[e] + v : v;
            }
        }
        __syncthreads();
    }
}

template <bool FLAT, int NS>
__global__ __launch_bounds__(256, NS == 1 ? 4 : (NS == 2 ? 2 : 1)) void wgrad_kernel(
    const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_body<FLAT, NS>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

// Several weight gradients in ONE launch.  At the training batch a single layer's gradient is a
// few microseconds of work wrapped in some 19 us of launch, fill, drain and fold latency, and
// it needs many pixel splits (each a slab of the weight's size) just to occupy the chip.  The
// layers whose gradients are due together share a launch instead: job j owns workgroups
// [start[j], start[j+1]), every job runs the general (not FLAT) single-stage body, four
// workgroups per CU, and one more launch folds all the slabs.
struct WgradBatch {
    int n;
    int start[kWgradMaxBatch + 1];
    WgradParams job[kWgradMaxBatch];
};
static_assert(sizeof(WgradBatch) <= 4096, "the batch travels in the kernel arguments");

__global__ __launch_bounds__(256, 4) void wgrad_batched_kernel(const WgradBatch bt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int j = 0;
    for (int i = 1; i < bt.n; ++i) j += (int)blockIdx.x >= bt.start[i] ? 1 : 0;
    wgrad_body<false, 1>(bt.job[j], (int)blockIdx.x - bt.start[j], bt.start[j + 1] - bt.start[j],
                         smem);
}

// dw[e] += sum over slabs of slab[s][e], in an order fixed by (S, SL) alone, so results repeat
// bit for bit.  A workgroup owns 256/SL groups of four elements; split lane j adds slabs
// j, j+SL, ... and the SL partial sums are folded in lane order through LDS.  Small weights
// with hundreds of slabs get SL = 16 so the slab loop stays short; P must be a multiple of 4.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int S,
                                                           long P, float* __restrict__ dw) {
    constexpr int COLS = 256 / SL;
    __shared__ f32x4 part[SL > 1 ? 256 : 1];
    const int colt = threadIdx.x % COLS;
    const int sl = threadIdx.x / COLS;
    const long i = (long)blockIdx.x * COLS + colt;  // group of four elements
    const bool ok = 4 * i < P;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* src = slab + 4 * i;
        int s = sl;
        for (; s + 3 * SL < S; s += 4 * SL) {
            const f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * P));
            const f32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + SL) * P));
            const f32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + 2 * SL) * P));
            const f32x4 v3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + 3 * SL) * P));
            a += v0;
            a += v1;
            a += v2;
            a += v3;
        }
        for (; s < S; s += SL)
            a += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * P));
    }
    if constexpr (SL > 1) {
        part[threadIdx.x] = a;
        __syncthreads();
        if (sl != 0) return;
        for (int j = 1; j < SL; ++j) a += part[j * COLS + colt];
    }
    if (ok) {
        f32x4* d = reinterpret_cast<f32x4*>(dw + 4 * i);
        *d = *d + a;
    }
}

static int32_t wgrad_check(const WgradParams& p, const char* what);

int32_t wgrad_launch(WgradParams& p, void* workspace, size_t workspace_bytes, hipStream_t st,
                     const char* what) {
    const int32_t ok = wgrad_check(p, what);
    if (ok != PE_OK) return ok;
    const long M = (long)p.B * p.Ho * p.Wo;
    const long ktot = (long)p.T * p.Cin;
    const long P = (long)p.Nw * ktot;
    const long tiles = ((p.N + 127) / 128) * ((ktot + 127) / 128);
    static const char* tune = getenv("PE_WGRAD_BLOCKS");  // tuning aid, at most kWgradMaxBlocks
    static const char* tune_t = getenv("PE_WGRAD_BIG_TILES");  // tuning aid
    const long big_tiles = tune_t ? atol(tune_t) : kWgradBigTiles;
    // a weight with many output tiles fills the chip with fewer pixel splits, and every split
    // costs a slab of the weight's size to write and fold
    const bool flat = p.T == 1 && p.tap_dy[0] == 0 && p.tap_dx[0] == 0 && p.sy == 1 && p.sx == 1 &&
                      p.Hi == p.Ho && p.Wi == p.Wo && p.OH == p.Ho && p.OW == p.Wo && p.osy == 1 &&
                      p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.cshift == 31;
    // 1x1 layers run the two-stage ring with one workgroup per CU (fewer slabs to write and
    // fold: 237 against 218 TFLOP/s over the ResNet-50 1x1 shapes at batch 64); the others keep
    // the single stage, whose per-step index arithmetic needs several workgroups per CU to hide
    long target = tune ? atol(tune)
                       : ((tiles >= big_tiles || flat) ? kWgradTargetBlocks / 2 : kWgradTargetBlocks);
    if (target < 1 || target > kWgradMaxBlocks) target = kWgradTargetBlocks;
    long splits = (target + tiles - 1) / tiles;
    const long max_splits = (M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long mps = (M + splits - 1) / splits;
    mps = (mps + WG_BK - 1) / WG_BK * WG_BK;
    splits = (M + mps - 1) / mps;
    p.m_per_split = mps;
    p.slab = static_cast<float*>(workspace);
    p.slab_stride = P;
    if (splits > 1) {
        PE_REQUIRE(workspace != nullptr && aligned(workspace, 16) && aligned(p.dw, 16), PE_E_INVALID,
                   "%s: workspace and dw must be 16-byte aligned and non-NULL", what);
        PE_REQUIRE((size_t)splits * (size_t)P * sizeof(float) <= workspace_bytes, PE_E_INVALID,
                   "%s: workspace too small (%zu bytes needed)", what,
                   (size_t)splits * (size_t)P * sizeof(float));
    }
    const size_t lds1 = 64 * 132 * sizeof(float);  // epilogue half tile (33.8 KB) > K tiles (32 KB)
    // PE_WGRAD_NS forces the stage count (1, 2, 3 or 4)
    static const char* tune_ns = getenv("PE_WGRAD_NS");
    const long nwg = tiles * splits;
    int ns = (flat && nwg <= 640) ? 2 : 1;
    if (tune_ns) ns = atoi(tune_ns);
    p.tiles = (int)tiles;
    p.splits = (int)splits;
    const dim3 grid((unsigned)(tiles * splits));
#define PE_WG(FLAT_, NS_)                                                                 \
    {                                                                                     \
        constexpr size_t lds = NS_ == 1 ? lds1 : (size_t)NS_ * 2 * WG_TILE;              \
        if constexpr (lds > 64 * 1024) {                                                  \
            const hipError_t e = ensure_dynamic_lds<&wgrad_kernel<FLAT_, NS_>>((int)lds); \
            PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in (%d bytes): %s", what,  \
                       (int)lds, hipGetErrorString(e));                                   \
        }                                                                                 \
        pe::launch(wgrad_kernel<FLAT_, NS_>, grid, 256, lds, st, p);                              \
    }
    if (flat) {
        if (ns == 4) PE_WG(true, 4) else if (ns == 3) PE_WG(true, 3)
        else if (ns == 2) PE_WG(true, 2) else PE_WG(true, 1)
    } else {
        if (ns == 4) PE_WG(false, 4) else if (ns == 3) PE_WG(false, 3)
        else if (ns == 2) PE_WG(false, 2) else PE_WG(false, 1)
    }
#undef PE_WG
    int32_t rc = check_launch(what);
    if (rc != PE_OK || splits == 1) return rc;
    const long groups = P >> 2;
    if (splits >= 64) {
        pe::launch(wgrad_reduce_kernel<16>, (unsigned)((groups + 15) / 16), 256, 0, st, p.slab, (int)splits,
                                                                              P, p.dw);
    } else if (splits >= 8) {
        pe::launch(wgrad_reduce_kernel<4>, (unsigned)((groups + 63) / 64), 256, 0, st, p.slab, (int)splits,
                                                                             P, p.dw);
    } else {
        pe::launch(wgrad_reduce_kernel<1>, (unsigned)((groups + 255) / 256), 256, 0, st, p.slab, (int)splits,
                                                                              P, p.dw);
    }
    return check_launch(what);
}

struct WgradFoldJob {
    const float* slab;
    float* dw;
    long P;
    int S;
};
struct WgradFoldBatch {
    int n;
    int start[kWgradMaxBatch + 1];
    WgradFoldJob job[kWgradMaxBatch];
};

// wgrad_reduce_kernel<4> for every job of a batched launch: 64 groups of four elements per
// workgroup, four split lanes, slabs added in the same fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgradFoldBatch fb) {
    constexpr int SL = 4, COLS = 64;
    __shared__ f32x4 part[256];
    int j = 0;
    for (int i = 1; i < fb.n; ++i) j += (int)blockIdx.x >= fb.start[i] ? 1 : 0;
    const float* slab = fb.job[j].slab;
    float* dw = fb.job[j].dw;
    const long P = fb.job[j].P;
    const int S = fb.job[j].S;
    const int colt = threadIdx.x % COLS;
    const int sl = threadIdx.x / COLS;
    const long i = (long)((int)blockIdx.x - fb.start[j]) * COLS + colt;
    const bool ok = 4 * i < P;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* src = slab + 4 * i;
        int s = sl;
        for (; s + 3 * SL < S; s += 4 * SL) {
            const f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * P));
            const f32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + SL) * P));
            const f32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + 2 * SL) * P));
            const f32x4 v3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + 3 * SL) * P));
            a += v0;
            a += v1;
            a += v2;
            a += v3;
        }
        for (; s < S; s += SL)
            a += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * P));
    }
    part[threadIdx.x] = a;
    __syncthreads();
    if (sl != 0) return;
    for (int q = 1; q < SL; ++q) a += part[q * COLS + colt];
    if (ok) {
        f32x4* d = reinterpret_cast<f32x4*>(dw + 4 * i);
        *d = *d + a;
    }
}

static int32_t wgrad_check(const WgradParams& p, const char* what) {
    PE_REQUIRE(p.x && p.dy && p.dw, PE_E_INVALID, "%s: NULL tensor pointer", what);
    PE_REQUIRE(p.B > 0 && p.Hi > 0 && p.Wi > 0 && p.Ho > 0 && p.Wo > 0 && p.N > 0 && p.Nw > 0 &&
                   p.Nw <= p.N,
               PE_E_INVALID, "%s: sizes must be positive", what);
    PE_REQUIRE(p.N % 8 == 0 && p.Cin % 8 == 0 && p.Cx % 8 == 0 && p.ldy % 8 == 0, PE_E_UNSUPPORTED,
               "%s: channel counts must be multiples of 8", what);
    PE_REQUIRE(p.T >= 1 && p.T <= kMaxTaps, PE_E_UNSUPPORTED, "%s: %d taps (max %d)", what, p.T,
               kMaxTaps);
    PE_REQUIRE(aligned(p.x, 16) && aligned(p.dy, 16) && aligned(p.dw, 4), PE_E_INVALID,
               "%s: tensors must be 16-byte aligned", what);
    const long M = (long)p.B * p.Ho * p.Wo;
    PE_REQUIRE(M < 0x7fffffffL && (long)p.B * p.OH * p.OW < 0x7fffffffL &&
                   (long)p.B * p.Hi * p.Wi < 0x7fffffffL &&
                   (long)p.Ho * p.Wo < (1L << 22) && p.B < (1 << 20),
               PE_E_UNSUPPORTED, "%s: pixel counts out of range", what);
    return PE_OK;
}

int32_t wgrad_launch_batch(WgradParams* ps, int n, void* workspace, size_t workspace_bytes,
                           hipStream_t st, const char* what) {
    PE_REQUIRE(ps != nullptr && n >= 1 && n <= kWgradMaxBatch, PE_E_INVALID,
               "%s: 1 to %d jobs per call", what, kWgradMaxBatch);
    if (n == 1) return wgrad_launch(ps[0], workspace, workspace_bytes, st, what);
    WgradBatch bt;
    WgradFoldBatch fb;
    bt.n = n;
    fb.n = 0;
    // work of a job = output tiles x pixels; the chip's 4 x 256 workgroup slots are shared out
    // in proportion, a job's share divided over its tiles gives its pixel splits
    double work[kWgradMaxBatch], total = 0.0;
    long tiles[kWgradMaxBatch], Ms[kWgradMaxBatch];
    for (int j = 0; j < n; ++j) {
        const int32_t rc = wgrad_check(ps[j], what);
        if (rc != PE_OK) return rc;
        const long ktot = (long)ps[j].T * ps[j].Cin;
        tiles[j] = ((ps[j].N + 127) / 128) * ((ktot + 127) / 128);
        Ms[j] = (long)ps[j].B * ps[j].Ho * ps[j].Wo;
        work[j] = (double)tiles[j] * (double)Ms[j];
        total += work[j];
    }
    static const char* tune = getenv("PE_WGRAD_BATCH_BLOCKS");  // tuning aid
    long budget = tune ? atol(tune) : kWgradBatchBlocks;
    if (budget < n || budget > kWgradMaxBlocks) budget = kWgradBatchBlocks;
    size_t used = 0;
    int blocks = 0, fold_blocks = 0;
    for (int j = 0; j < n; ++j) {
        WgradParams& p = ps[j];
        const long P = (long)p.Nw * p.T * p.Cin;
        long share = (long)((double)budget * work[j] / total + 0.5);
        long splits = (share + tiles[j] - 1) / tiles[j];
        const long max_splits = (Ms[j] + 255) / 256;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        long mps = (Ms[j] + splits - 1) / splits;
        mps = (mps + WG_BK - 1) / WG_BK * WG_BK;
        splits = (Ms[j] + mps - 1) / mps;
        p.m_per_split = mps;
        p.tiles = (int)tiles[j];
        p.splits = (int)splits;
        p.slab_stride = P;
        p.slab = nullptr;
        if (splits > 1) {
            PE_REQUIRE(P % 4 == 0, PE_E_UNSUPPORTED, "%s: weight size not a multiple of 4", what);
            const size_t need = (size_t)splits * (size_t)P * sizeof(float);
            PE_REQUIRE(aligned(p.dw, 16), PE_E_INVALID, "%s: dw must be 16-byte aligned", what);
            PE_REQUIRE(workspace != nullptr && aligned(workspace, 16) &&
                           used + need <= workspace_bytes,
                       PE_E_INVALID, "%s: workspace too small (%zu bytes needed so far)", what,
                       used + need);
            p.slab = reinterpret_cast<float*>(static_cast<char*>(workspace) + used);
            used += (need + 255) & ~(size_t)255;
            WgradFoldJob& f = fb.job[fb.n];
            f.slab = p.slab;
            f.dw = p.dw;
            f.P = P;
            f.S = (int)splits;
            fb.start[fb.n] = fold_blocks;
            fold_blocks += (int)(((P >> 2) + 63) / 64);
            ++fb.n;
        }
        bt.start[j] = blocks;
        bt.job[j] = p;
        blocks += (int)(tiles[j] * splits);
    }
    bt.start[n] = blocks;
    fb.start[fb.n] = fold_blocks;
    const size_t lds1 = 64 * 132 * sizeof(float);
    pe::launch(wgrad_batched_kernel, (unsigned)blocks, 256, lds1, st, bt);
    int32_t rc = check_launch(what);
    if (rc != PE_OK || fb.n == 0) return rc;
    pe::launch(wgrad_reduce_batched_kernel, (unsigned)fold_blocks, 256, 0, st, fb);
    return check_launch(what);
}

size_t wgrad_batch_workspace_bytes(long P_total) {
    // sum of splits_j * P_j <= 16384 * (budget + tiles) + P_total, each slab rounded up to 256 B
    return (size_t)(kWgradMaxBlocks + 64) * 16384 * sizeof(float) +
           2 * (size_t)P_total * sizeof(float) + 256 * (size_t)kWgradMaxBatch;
}

size_t wgrad_workspace_bytes(long P) {
    // splits * P <= (target/tiles + 1) * P with tiles >= P / 16384
    const size_t by_target = (size_t)(kWgradMaxBlocks + 1) * 16384 * sizeof(float);
    const size_t two = 2 * (size_t)P * sizeof(float);
    return (by_target > two ? by_target : two) + (size_t)P * sizeof(float);
}

}  // namespace pe
