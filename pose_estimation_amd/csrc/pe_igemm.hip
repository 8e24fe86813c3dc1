// Implicit-GEMM convolution kernel: bf16 MFMA 16x16x32, fp32 accumulate, direct-to-LDS tiles.
//
// Workgroup = WM x WN waves (WM = 2 or 4 along pixels, WN = 2 or 4 along channels), output tile
// BM pixels x BN channels, K step 64; every wave owns a (BM/WM) x (BN/WN) sub-tile.  Tiles go global -> LDS
// with global_load_lds_dwordx4 (16 B per lane, no VGPR staging, no ds_write pass) into 128-byte
// rows.  Three pipelines exist (template parameter PIPE):
//   0, single buffer : load, barrier, MFMA, barrier; <= 35 KB of LDS and 128 VGPRs, so four
//                   256-thread workgroups share a CU and cover each other's load latency;
//   1, double buffer : the loads of K step k+1 are in flight while step k is multiplied, one
//                   __syncthreads per step (which drains every outstanding load);
//   N >= 2, ring   : N LDS stages, the loads of K steps k+1 .. k+N-1 in flight while step k is
//                   multiplied.  One raw s_barrier per step and a COUNTED s_waitcnt vmcnt that
//                   retires only step k's loads, so younger tiles stay in flight across the
//                   barrier and the workgroup hides its own load latency: what large tiles
//                   (256 pixels, eight waves, one workgroup per CU) need.
//
// A wave instruction writes 1 KiB = 8 consecutive LDS rows, lane l to (row l>>3, 16-byte slot
// l&7).  Rows are unpadded, so bank conflicts are avoided by an XOR swizzle: slot s of row r holds
// K-chunk s ^ (r & 7); with it the 16-lane groups of ds_read_b128 touch 16 distinct slots.  The
// swizzle is applied to the per-lane SOURCE address (the LDS destination of a global_load_lds is
// not a per-lane choice) and again when fragments are read.  Rows or taps that fall outside the
// image read a 16-byte block of zeros instead of being predicated off.
//
// The MFMA "A" operand is the weight tile and the "B" operand the activation tile, so a lane's
// four accumulator registers are four consecutive output channels of one pixel.  The epilogue
// (scale, shift, residual, ReLU, bf16 convert) stages the fp32 tile through LDS in two row
// halves so that residual loads and output stores are whole 16-byte chunks, 2 * BN bytes
// contiguous per pixel.
//
// Several "phases" (output-parity classes of a stride-2 transposed conv or of a stride-2
// backward-data pass) run in one launch along gridDim.y, each with its own taps, weight block,
// row grid and output offset, so small layers still fill the chip.
#include <stdio.h>
#include <stdlib.h>

#include "pe_igemm.h"

namespace pe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BK = 64;
constexpr int ROWB = 128;  // bytes per LDS row (64 bf16, unpadded)

__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    return pack_bf16x2_bits(a, b);
}

// q = m / d, r = m % d for 0 <= m < 2^31, d >= 1 and quotients below 2^20: float reciprocal
// estimate, then one fix-up step either way (an integer division costs about 40 instructions).
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
    q = (int)((float)m * inv_d);
    r = m - q * d;
    const int fix = (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    q += fix;
    r -= fix * d;
}

// s_waitcnt vmcnt(N): all but this wave's N youngest vector-memory operations are done.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// FAST: addresses of the staging loads cost a 64-bit add and a select instead of a multiply
// chain: every tile row keeps the pointer of its centre pixel and one bit per tap ("this tap
// lies inside the image"), and a K step adds a wave-uniform byte offset.  Needs single-pixel K
// runs, Cin a multiple of 64 and at most 32 taps per phase (igemm_launch checks).
// XCD-aware tile order: workgroups that share an XCD (ids congruent mod 8) take a contiguous
// run of tiles, so the n-tiles of one pixel panel reuse that XCD's L2.
__device__ __forceinline__ int xcd_tile_order(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// One output tile: tile `bid` (in pixel-panel-major order, n-tiles of a panel adjacent) of
// phase `ph`, computed by the whole workgroup in the LDS at `smem`.
// HEAD: the epilogue keeps the bf16 output tile in LDS and multiplies it by the pointwise head
// p.head_w (see igemm_head_kernel); nothing is stored to p.y.
template <int BM, int BN, int WM, int PIPE, bool PF, bool MASK, int WN, bool FAST, bool RES,
          bool HEAD = false>
__device__ __forceinline__ void igemm_tile(const IgemmParams& p, char* smem, int bid, const int ph) {
    constexpr bool DB = PIPE == 1;    // the double buffer with draining barriers
    constexpr int NS = PIPE >= 2 ? PIPE : 0;   // ring stages
    constexpr int NT = WM * WN * 64;  // threads
    constexpr int RPI = NT / 8;       // tile rows one staging instruction of the workgroup covers
    constexpr int WROWS = BM / WM;    // pixels per wave
    constexpr int WCOLS = BN / WN;    // channels per wave
    constexpr int TM = WROWS / 16;    // 16-pixel tiles per wave
    constexpr int TN = WCOLS / 16;    // 16-channel tiles per wave
    constexpr int XI = BM / RPI;      // activation rows staged per thread
    constexpr int WI = BN / RPI;      // weight rows staged per thread
    constexpr int XBYTES = BM * ROWB;
    constexpr int TILEB = (BM + BN) * ROWB;
    static_assert(XI >= 1 && WI >= 1 && TM >= 1 && TN >= 1, "tile too small for the workgroup");
    static_assert(WM % 2 == 0, "the epilogue stages the tile in two halves of the wave rows");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // the wave index as a scalar: LDS destinations of the staging loads (M0) stay in SGPRs
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN;
    const int wave_n = wid % WN;

    // ---- phase (uniform)
    const int Ho = p.ph_Ho[ph], Wo = p.ph_Wo[ph];
    const int T = p.ph_T[ph];
    const int t0 = p.ph_t0[ph];
    const int ooy = p.ph_ooy[ph], oox = p.ph_oox[ph];
    const uint16_t* wbase = p.w + p.ph_woff[ph];

    const int HoWo = Ho * Wo;
    const long M = (long)p.B * HoWo;
    const int ntn = (p.N + BN - 1) / BN;
    const long ntiles = ((M + BM - 1) / BM) * ntn;

    if (bid >= ntiles) return;  // uniform: phases may have fewer tiles than the grid
    const int tile_m = bid / ntn;
    const int tile_n = bid - tile_m * ntn;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging: wave w, instruction i covers rows RPI*i + 8*w .. +7; lane l -> row +(l>>3),
    // slot l&7, which must receive K-chunk (l&7) ^ (row&7) = (l&7) ^ ((l>>3)&7)
    const int r0 = tid >> 3;
    const int cc = (lane & 7) ^ ((lane >> 3) & 7);
    const float inv_howo = 1.0f / (float)HoWo;
    const float inv_wo = 1.0f / (float)Wo;
    int iy0[XI], ix0[XI], pixbase[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const long m = m0 + r0 + RPI * i;
        if (m < M) {
            int b, rem, oy, ox;
            fast_divmod((int)m, HoWo, inv_howo, b, rem);
            fast_divmod(rem, Wo, inv_wo, oy, ox);
            iy0[i] = oy * p.sy;
            ix0[i] = ox * p.sx;
            pixbase[i] = b * p.Hi * p.Wi;
        } else {
            iy0[i] = -(1 << 20);  // never in range
            ix0[i] = 0;
            pixbase[i] = 0;
        }
    }
    const int kc = (p.Cin + BK - 1) / BK;   // K steps per tap; the last may be part-filled
    const int KT = T * kc;       // K steps in total
    const long ktot = p.wk_row;  // weight row length
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero16);

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lr = lane & 15;
    const int sw = lane & 7;  // (row & 7) of every fragment row this lane reads
    const int xs_off = (wave_m * WROWS + lr) * ROWB;
    const int ws_off = XBYTES + (wave_n * WCOLS + lr) * ROWB;

    // FAST addressing state (see the template comment)
    const char* xrow[FAST ? XI : 1];
    uint32_t xmask[FAST ? XI : 1];
    const char* wrow[FAST ? WI : 1];
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            xrow[i] = reinterpret_cast<const char*>(
                p.x + ((long)(pixbase[i] + iy0[i] * p.Wi + ix0[i]) * p.Cx + cc * 8));
            xmask[i] = 0u;
        }
        for (int t = 0; t < T; ++t) {
            const int inf = p.tap_info[(t0 + t) & (kMaxTaps - 1)];
            const int dy = (int)(int8_t)(inf & 0xff);
            const int dx = (int)(int8_t)((inf >> 8) & 0xff);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bool ok = (unsigned)(iy0[i] + dy) < (unsigned)p.Hi &&
                                (unsigned)(ix0[i] + dx) < (unsigned)p.Wi;
                xmask[i] |= ok ? (1u << t) : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            // channels past N read the last real row: their products are never stored
            int n = n0 + r0 + RPI * i;
            n = n < p.N ? n : p.N - 1;
            wrow[i] = reinterpret_cast<const char*>(wbase + ((long)n * ktot + cc * 8));
        }
    }

    int tap = t0, cstep = 0;
    // (dy, dx, weight slot) of the current tap, one dword from the kernel arguments (a scalar
    // load); the next tap's dword is fetched a whole K step before it is needed
    int info = p.tap_info[t0];
    // issues the loads of the next K step into the tile at `dst` and advances (tap, cstep)
    auto issue = [&](char* dst) {
        const int dy = (int)(int8_t)(info & 0xff);
        const int dx = (int)(int8_t)((info >> 8) & 0xff);
        if constexpr (FAST) {
            // wave-uniform byte offsets of this K step
            const long xoff = ((long)(dy * p.Wi + dx) * p.Cx + cstep * BK) * 2;
            const long woff = ((long)(info >> 16) * p.Cin + cstep * BK) * 2;
            const uint32_t bit = 1u << (tap - t0);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const char* src = (xmask[i] & bit) ? xrow[i] + xoff
                                                   : reinterpret_cast<const char*>(zero);
                __builtin_amdgcn_global_load_lds(
                    (gptr_t)src, (lptr_t)(dst + (RPI * i + 8 * wid) * ROWB), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WI; ++i)
                __builtin_amdgcn_global_load_lds(
                    (gptr_t)(wrow[i] + woff),
                    (lptr_t)(dst + XBYTES + (RPI * i + 8 * wid) * ROWB), 16, 0, 0);
            if (++cstep == kc) {
                cstep = 0;
                ++tap;
                info = p.tap_info[tap & (kMaxTaps - 1)];
            }
            return;
        }
        const int coff = cstep * BK + cc * 8;
        const bool kin = coff < p.Cin;  // Cin % 8 == 0: a 16-byte chunk is whole or absent
        const long wk = (long)(info >> 16) * p.Cin + coff;  // K offset in a weight row
        const int xextra = coff >> p.cshift;
        const int ch = coff & p.cmask;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int iy = iy0[i] + dy;
            const int ix = ix0[i] + dx + xextra;
            const bool ok = kin && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const long off = (long)(pixbase[i] + iy * p.Wi + ix) * p.Cx + ch;
            const uint16_t* src = ok ? p.x + off : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (RPI * i + 8 * wid) * ROWB),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int n = n0 + r0 + RPI * i;
            const uint16_t* src = (kin && n < p.N) ? wbase + ((long)n * ktot + wk) : zero;
            __builtin_amdgcn_global_load_lds(
                (gptr_t)src, (lptr_t)(dst + XBYTES + (RPI * i + 8 * wid) * ROWB), 16, 0, 0);
        }
        if (++cstep == kc) {
            cstep = 0;
            ++tap;
            info = p.tap_info[tap & (kMaxTaps - 1)];
        }
    };
    auto multiply = [&](const char* tile) {
        const char* xs = tile + xs_off;
        const char* ws = tile + ws_off;
        if constexpr (PF) {
            // every fragment of the K step is requested before the first MFMA, so LDS reads
            // stream underneath the multiplies (costs 2 x (TN + TM) x 4 VGPRs)
            bf16x8 af[2][TN], bfr[2][TM];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot0 = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
#pragma unroll
                for (int a = 0; a < TN; ++a)
                    af[kk][a] = *reinterpret_cast<const bf16x8*>(ws + a * 16 * ROWB + slot0);
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    bfr[kk][b] = *reinterpret_cast<const bf16x8*>(xs + b * 16 * ROWB + slot0);
            }
            asm volatile("" ::: "memory");  // keeps the reads above the multiplies
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][a], bfr[kk][b],
                                                                            acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[TN], bfr[TM];
                const int slot0 = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
#pragma unroll
                for (int a = 0; a < TN; ++a)
                    af[a] = *reinterpret_cast<const bf16x8*>(ws + a * 16 * ROWB + slot0);
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    bfr[b] = *reinterpret_cast<const bf16x8*>(xs + b * 16 * ROWB + slot0);
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    if constexpr (NS >= 2) {
        // Ring of NS stages.  Iteration kt: (1) wait until this wave's loads of tile kt have
        // landed, leaving the younger tiles in flight; (2) barrier: every wave's share of tile
        // kt is in LDS and every wave has finished reading tile kt-1; (3) restage the slot tile
        // kt-1 occupied with tile kt+NS-1; (4) multiply tile kt.  A tile is therefore read only
        // after the wait that retires it AND a barrier, and a slot is overwritten only after a
        // barrier that follows its last read.
        constexpr int LPT = XI + WI;    // staging instructions per thread and tile
        static_assert((NS - 2) * LPT <= 63, "too many loads in flight for the vmcnt field");
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < KT) issue(smem + s * TILEB);
        int cur = 0, nxt = NS - 1;
        for (int kt = 0; kt < KT; ++kt) {
            // tiles issued so far: min(KT, kt + NS - 1); all but tile kt's may stay in flight
            if (kt + NS - 1 <= KT) wait_vmcnt<(NS - 2) * LPT>();
            else wait_vmcnt<0>();       // the last NS-2 steps: nothing younger worth keeping
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of tile kt-1 are back
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < KT) issue(smem + nxt * TILEB);
            __builtin_amdgcn_s_setprio(1);
            multiply(smem + cur * TILEB);
            __builtin_amdgcn_s_setprio(0);
            if (++cur == NS) cur = 0;
            if (++nxt == NS) nxt = 0;
        }
        __syncthreads();   // the epilogue reuses the stages
    } else if constexpr (DB) {
        issue(smem);
        __syncthreads();  // the compiler drains vmcnt before a barrier: tile 0 is in LDS
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * TILEB;
            if (kt + 1 < KT) issue(smem + ((kt + 1) & 1) * TILEB);  // lands while we multiply
            multiply(cur);
            __syncthreads();  // next tile landed (vmcnt drained) and everyone is done with `cur`
        }
    } else {
        for (int kt = 0; kt < KT; ++kt) {
            issue(smem);
            __syncthreads();  // the compiler drains vmcnt before this barrier: the tile is in LDS
            multiply(smem);
            __syncthreads();  // everyone has read the tile before the next K step overwrites it
        }
    }

    if constexpr (HEAD) {
        // Fused pointwise head.  Every wave has passed the K loop's last barrier, so the K tile
        // is free: the bf16 output tile goes there in the activation-tile layout of the K loop
        // (two planes of 64 channels, 128-byte rows, 16-byte slots swizzled by row & 7),
        // written straight from the accumulators (a lane holds 4 consecutive channels of one
        // pixel: one 8-byte store), and the head's weights for this tile's BN input channels sit
        // behind it.  The values are what the unfused layer stores (same fmaf, ReLU and
        // rounding), and the head's MFMAs run over the channels in the order a separate 1x1
        // launch walks them, so the result is bit-identical to two launches.
        static_assert(BM == 128 && BN == 128 && WM == 2 && WN == 2 && PIPE == 0 && !MASK,
                      "the fused head is written for the single-stage 128 x 128 tile");
        // The lane index as the epilogue sees it is opaque to the optimiser, so none of the
        // epilogue's index arithmetic is computed (and kept in registers) ahead of the K loop.
        int le = lane;
        asm volatile("" : "+v"(le));
        const int elr = le & 15, eq = le >> 4;
        const int R = p.head_N;                       // weight rows (<= 32)
        char* a3 = smem;
        char* wfs = smem + 2 * BM * ROWB;             // [2][R][128 B]
        // head weights of channels n0 .. n0+127: 2 planes x R rows x 8 chunks of 16 bytes
        const int et = wid * 64 + le;
        uint4 wreg[2];
        int wdst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = et + i * NT;
            wdst[i] = -1;
            if (id < 2 * R * 8) {
                const int s = id / (R * 8), rem = id - s * (R * 8);
                const int row = rem >> 3, ck = rem & 7;
                wreg[i] = *reinterpret_cast<const uint4*>(p.head_w + (long)row * p.N + n0 + s * 64 +
                                                          ck * 8);
                wdst[i] = (s * R + row) * ROWB + ((ck ^ (row & 7)) << 4);
            }
        }
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int c = a * 16 + eq * 4;            // channel inside this wave's plane
            const int nn = n0 + wave_n * WCOLS + c;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + nn);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + nn);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int row = wave_m * WROWS + b * 16 + elr;
                float v0 = acc[a][b].x, v1 = acc[a][b].y, v2 = acc[a][b].z, v3 = acc[a][b].w;
                if (p.scale || p.shift) {
                    v0 = fmaf(v0, sc.x, sh.x);
                    v1 = fmaf(v1, sc.y, sh.y);
                    v2 = fmaf(v2, sc.z, sh.z);
                    v3 = fmaf(v3, sc.w, sh.w);
                }
                if (p.relu) {
                    v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f);
                    v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                    if (p.relu == 2) {
                        v0 = fminf(v0, 6.0f); v1 = fminf(v1, 6.0f);
                        v2 = fminf(v2, 6.0f); v3 = fminf(v3, 6.0f);
                    }
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(a3 + wave_n * (BM * ROWB) + row * ROWB +
                                          ((((c >> 3) ^ (row & 7)) << 4) | ((c & 7) << 1))) = o;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (wdst[i] >= 0) *reinterpret_cast<uint4*>(wfs + wdst[i]) = wreg[i];
        // The accumulators are dead from here on.  The head's sums over the channel tiles
        // before this one are parked in the head's own output (float32, written by this same
        // thread a tile ago, so they come back from the L2) instead of in registers across the
        // K loop: 16 registers fewer, which is what lets four workgroups share a CU.  The
        // MFMA chain continues from the parked value, so the order of additions is still that
        // of one launch over all channels.  Offsets are 32-bit (igemm_head_launch checks).
        const int OHWh = p.OH * p.OW;
        int hoff[2];                                  // element offset of (image, pixel), -1 = none
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long m = m0 + (2 * wid + pt) * 16 + elr;
            hoff[pt] = -1;
            if (m < M) {
                int bb, rem, oy, ox;
                fast_divmod((int)m, HoWo, inv_howo, bb, rem);
                fast_divmod(rem, Wo, inv_wo, oy, ox);
                hoff[pt] = bb * R * OHWh + (oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
            }
        }
        f32x4 hacc[4];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (tile_n > 0 && hoff[pt] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = tn * 16 + eq * 4 + j;
                        // agent-scope load: served by the L2 the store went to, never by a
                        // line the CU's L1 may have kept from before it
                        if (n < R)
                            v[j] = __hip_atomic_load(p.head_out + (hoff[pt] + n * OHWh),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                hacc[pt * 2 + tn] = f32x4{v[0], v[1], v[2], v[3]};
            }
        __syncthreads();
        // wave `wid` multiplies pixel tiles 2*wid, 2*wid+1 (16 pixels each) by both 16-row
        // weight tiles; rows past R read row R-1, their products are never stored
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 wf[2];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    int wrow = tn * 16 + elr;
                    wrow = wrow < R ? wrow : R - 1;
                    wf[tn] = *reinterpret_cast<const bf16x8*>(
                        wfs + (s * R + wrow) * ROWB + (((kk * 4 + eq) ^ (wrow & 7)) << 4));
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int row = (2 * wid + pt) * 16 + elr;
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(
                        a3 + s * (BM * ROWB) + row * ROWB + (((kk * 4 + eq) ^ (elr & 7)) << 4));
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        hacc[pt * 2 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            wf[tn], xf, hacc[pt * 2 + tn], 0, 0, 0);
                }
            }
        const bool last = tile_n == ntn - 1;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            if (hoff[pt] < 0) continue;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const f32x4 hv = hacc[pt * 2 + tn];
                const float v[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = tn * 16 + eq * 4 + j;
                    if (n < R)
                        p.head_out[hoff[pt] + n * OHWh] =
                            (last && p.head_bias) ? v[j] + p.head_bias[n] : v[j];
                }
            }
        }
        // The next tile's K loop overwrites a3: every wave's LDS reads must be back, nothing
        // more.  A raw barrier, because __syncthreads() would also wait for the stores above
        // to complete (a microsecond or two per tile with nothing to overlap); they drain
        // under the next tile's first loads instead.  This thread reads its parked sums back a
        // whole K loop later, behind that loop's vmcnt(0) waits, which retire these older
        // stores first.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return;
    }

    // ---- epilogue
    const bool identity = (p.osy == 1 && p.osx == 1 && ooy == 0 && oox == 0 && p.OH == Ho &&
                           p.OW == Wo);
    const int OHW = p.OH * p.OW;
    const int nq = (lane >> 4) * 4;

    if (p.out_mode == 0 && (p.N & 7) == 0 && (p.ldc & 7) == 0) {
        // The fp32 tile goes through LDS in two ROW halves (the wave_m halves), so that every
        // store and residual load of a pixel covers the tile's full BN-channel run (256
        // contiguous bytes at BN = 128) instead of half of it.
        constexpr int RH = BM / 2;      // rows staged per pass
        constexpr int CP = BN + 4;      // float pitch: 8-lane ds_write_b128 groups hit distinct banks
        constexpr int CPR = BN / 8;     // 16-byte output chunks per pixel row
        constexpr int RPP = NT / CPR;   // pixel rows per iteration of the store loop
        constexpr int NPASS = (RH + RPP - 1) / RPP;
        constexpr int SG = NT / BN;     // statistics: thread groups over rows
        constexpr int RPG = RH / SG;    // rows per group and pass
        static_assert(SG >= 1 && RPG >= 1, "tile too small for the statistics fold");
        float* cs = reinterpret_cast<float*>(smem);
        float* sred = cs + RH * CP;     // [SG][2][BN]
        const int chunk = tid % CPR;
        const int prow = tid / CPR;
        const int n = n0 + chunk * 8;
        const int scol = tid % BN, sgrp = tid / BN;
        float st1 = 0.0f, st2 = 0.0f;   // running column sums of the fused statistics
        // MASK: this thread's chunk of 8 channels keeps its own sums over the rows it stores
        float ms1[MASK ? 8 : 1], ms2[MASK ? 8 : 1];
        if constexpr (MASK) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ms1[i] = ms2[i] = 0.0f;
        }
        // outputs far larger than the caches are streamed past them (uniform)
        const bool nt_store = p.nt_store != 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // output pixel of each store iteration; -1 = no store
            long opixs[NPASS];
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int pr = prow + pass * RPP;
                const long m = m0 + half * RH + pr;
                long opix = m;
                if (!identity) {
                    int bb, rem, oy, ox;
                    fast_divmod((int)(m < M ? m : 0), HoWo, inv_howo, bb, rem);
                    fast_divmod(rem, Wo, inv_wo, oy, ox);
                    opix = (long)bb * OHW + (oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
                }
                opixs[pass] = (pr < RH && m < M) ? opix : -1;
            }
            // the residual rows of every iteration are requested before the tile is staged, so
            // the loads are in flight across the LDS round trip instead of one per iteration
            uint4 rres[NPASS];
            if constexpr (MASK) {   // the rows of z this thread will mask with
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    rres[pass] = make_uint4(0u, 0u, 0u, 0u);
                    if (opixs[pass] >= 0 && n < p.N)
                        rres[pass] = *reinterpret_cast<const uint4*>(p.mask_z + opixs[pass] * p.ldc + n);
                }
            } else if (p.res) {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    rres[pass] = make_uint4(0u, 0u, 0u, 0u);
                    if (opixs[pass] >= 0 && n < p.N) {
                        const uint16_t* src = p.res + opixs[pass] * p.ldc + n;
                        if (nt_store) {   // as large as the output and read exactly once
                            const u32x4 rv =
                                __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
                            rres[pass] = make_uint4(rv.x, rv.y, rv.z, rv.w);
                        } else {
                            rres[pass] = *reinterpret_cast<const uint4*>(src);
                        }
                    }
                }
            }
            if (wave_m / (WM / 2) == half) {
                const int rbase = (wave_m % (WM / 2)) * WROWS;
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int col = wave_n * WCOLS + a * 16 + nq;
                    const int nn = n0 + col;
                    if (p.scale || p.shift) {  // uniform: training convs carry neither
                        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                        if (nn < p.N) {
                            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + nn);
                            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + nn);
                        }
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            const int pr = rbase + b * 16 + lr;
                            f32x4 v;  // one fused multiply-add per element
                            v.x = fmaf(acc[a][b].x, sc.x, sh.x);
                            v.y = fmaf(acc[a][b].y, sc.y, sh.y);
                            v.z = fmaf(acc[a][b].z, sc.z, sh.z);
                            v.w = fmaf(acc[a][b].w, sc.w, sh.w);
                            *reinterpret_cast<f32x4*>(cs + pr * CP + col) = v;
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            const int pr = rbase + b * 16 + lr;
                            *reinterpret_cast<f32x4*>(cs + pr * CP + col) = acc[a][b];
                        }
                    }
                }
            }
            __syncthreads();
            if (n < p.N) {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int pr = prow + pass * RPP;
                    const long opix = opixs[pass];
                    if (opix < 0) continue;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + pr * CP + chunk * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + pr * CP + chunk * 8 + 4);
                    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    if constexpr (MASK) {
                        const uint4 zv = rres[pass];
                        const uint32_t zu[4] = {zv.x, zv.y, zv.z, zv.w};
                        if constexpr (RES) {   // mask from the stored activation (+ residual)
                            const uint4 av = *reinterpret_cast<const uint4*>(p.mask_a + opix * p.ldc + n);
                            const uint32_t au[4] = {av.x, av.y, av.z, av.w};
                            if (p.res) {
                                const uint4 rv =
                                    *reinterpret_cast<const uint4*>(p.res + opix * p.ldc + n);
                                const uint32_t ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    v[2 * i] += bf16_bits_to_float(ru[i] & 0xffffu);
                                    v[2 * i + 1] += bf16_bits_to_float(ru[i] >> 16);
                                }
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const uint32_t zb = (i & 1) ? (zu[i >> 1] >> 16) : (zu[i >> 1] & 0xffffu);
                                const uint32_t ab = (i & 1) ? (au[i >> 1] >> 16) : (au[i >> 1] & 0xffffu);
                                const float zf = bf16_bits_to_float(zb);
                                const float g = bf16_bits_to_float(ab) > 0.0f
                                                    ? bf16_bits_to_float(float_to_bf16_bits(v[i]))
                                                    : 0.0f;
                                v[i] = g;
                                ms1[i] += g;
                                ms2[i] += g * zf;
                            }
                        } else {
                        // scale and shift of this thread's 8 channels (n < N and N % 8 == 0 here);
                        // re-read per pass from the cache rather than held across the epilogue
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.mask_sc + n);
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.mask_sc + n + 4);
                        const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.mask_sh + n);
                        const f32x4 h1 = *reinterpret_cast<const f32x4*>(p.mask_sh + n + 4);
                        const float msc8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const float msh8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float zf = bf16_bits_to_float((i & 1) ? (zu[i >> 1] >> 16)
                                                                        : (zu[i >> 1] & 0xffffu));
                            float a = zf * msc8[i] + msh8[i];
                            a = fmaxf(a, 0.0f);
                            if (p.mask_hi == 6.0f) a = fminf(a, 6.0f);
                            a = bf16_bits_to_float(float_to_bf16_bits(a));
                            const float g = (a > 0.0f && a < p.mask_hi)
                                                ? bf16_bits_to_float(float_to_bf16_bits(v[i]))
                                                : 0.0f;
                            v[i] = g;
                            ms1[i] += g;
                            ms2[i] += g * zf;
                        }
                        }
                    } else if (p.res) {
                        const uint4 rv = rres[pass];
                        const uint32_t u[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[2 * i] += bf16_bits_to_float(u[i] & 0xffffu);
                            v[2 * i + 1] += bf16_bits_to_float(u[i] >> 16);
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.0f);
                        if (p.relu == 2) {  // ReLU6
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = fminf(v[i], 6.0f);
                        }
                    }
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]);
                    o.w = pack_bf16x2(v[6], v[7]);
                    uint16_t* dst = reinterpret_cast<uint16_t*>(p.y) + opix * p.ldc + n;
                    if (nt_store) {
                        const u32x4 ov = {o.x, o.y, o.z, o.w};
                        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(dst));
                    } else {
                        *reinterpret_cast<uint4*>(dst) = o;
                    }
                }
            }
            // Batch-norm statistics of what was just stored (the bf16-rounded tile): per column,
            // sum and sum of squares over the tile's rows, in a fixed order.  Rows past M hold
            // exact zeros (their inputs were the zero block and a BN conv has no bias).
            if (!MASK && p.stat_partial) {
#pragma unroll 8
                for (int r = sgrp * RPG; r < (sgrp + 1) * RPG; ++r) {
                    const float v =
                        __uint_as_float(pack_bf16x2_bits(cs[r * CP + scol], 0.0f) << 16);
                    st1 += v;
                    st2 += v * v;
                }
            }
            __syncthreads();  // the staged half is consumed before the next one overwrites it
        }
        if constexpr (MASK) {
            // per-thread sums -> [row group][2][BN] in LDS (the staged tile is consumed), folded
            // over the row groups in order
            float* mred = cs;   // RPP * 2 * BN = 16 * NT floats: lds_bytes() reserves them
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mred[(prow * 2 + 0) * BN + chunk * 8 + i] = ms1[i];
                mred[(prow * 2 + 1) * BN + chunk * 8 + i] = ms2[i];
            }
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += NT) {
                const int sidx = e / BN, col = e % BN;
                float t = 0.0f;
                for (int g2 = 0; g2 < RPP; ++g2) t += mred[(g2 * 2 + sidx) * BN + col];
                const int nn = n0 + col;
                if (nn < p.N)
                    p.stat_partial[((long)(ph * p.stat_ntm + tile_m) * 2 + sidx) * p.N + nn] = t;
            }
            return;
        }
        if (p.stat_partial) {
            sred[(sgrp * 2 + 0) * BN + scol] = st1;
            sred[(sgrp * 2 + 1) * BN + scol] = st2;
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += NT) {
                const int sidx = e / BN, col = e % BN;
                float t = 0.0f;
#pragma unroll
                for (int g2 = 0; g2 < SG; ++g2) t += sred[(g2 * 2 + sidx) * BN + col];
                const int nn = n0 + col;
                if (nn < p.N)
                    p.stat_partial[((long)(ph * p.stat_ntm + tile_m) * 2 + sidx) * p.N + nn] = t;
            }
        }
        return;
    }

    // general path: channel counts that are not multiples of 8, float32 NCHW output
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const long m = m0 + wave_m * WROWS + b * 16 + lr;
        if (m >= M) continue;
        const int bb = (int)(m / HoWo);
        const int rem = (int)(m - (long)bb * HoWo);
        const int oy = rem / Wo;
        const int ox = rem - oy * Wo;
        const int opl = (oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
        const long opix = (long)bb * OHW + opl;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + wave_n * WCOLS + a * 16 + nq;
            const float v[4] = {acc[a][b].x, acc[a][b].y, acc[a][b].z, acc[a][b].w};
            for (int j = 0; j < 4 && n + j < p.N; ++j) {
                float t = v[j] * (p.scale ? p.scale[n + j] : 1.0f) + (p.shift ? p.shift[n + j] : 0.0f);
                if (p.out_mode == 0) {
                    if (p.res) t += bf16_bits_to_float(p.res[opix * p.ldc + n + j]);
                    if (p.relu) t = fmaxf(t, 0.0f);
                    if (p.relu == 2) t = fminf(t, 6.0f);
                    reinterpret_cast<uint16_t*>(p.y)[opix * p.ldc + n + j] = (uint16_t)float_to_bf16_bits(t);
                } else {
                    if (p.relu) t = fmaxf(t, 0.0f);
                    if (p.relu == 2) t = fminf(t, 6.0f);
                    reinterpret_cast<float*>(p.y)[((long)bb * p.N + n + j) * OHW + opl] = t;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int PIPE, bool PF, int MINB, bool MASK = false, int WN = 2,
          bool FAST = false, bool RES = false>
__global__ __launch_bounds__(WM * WN * 64, MINB)
__attribute__((amdgpu_waves_per_eu(MINB * WM * WN / 4, MINB * WM * WN / 4))) void igemm_kernel(const IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    igemm_tile<BM, BN, WM, PIPE, PF, MASK, WN, FAST, RES>(
        p, smem, xcd_tile_order(blockIdx.x, gridDim.x), blockIdx.y);
}

template <int BM, int BN, int PIPE, int NT = 256, bool MASK = false>
static constexpr size_t lds_bytes() {
    const size_t ktile = (size_t)(BM + BN) * ROWB * (PIPE == 0 ? 1 : (PIPE == 1 ? 2 : PIPE));
    // epilogue half tile + the per-group column sums of the fused batch-norm statistics
    const size_t etile = (size_t)(BM / 2) * (BN + 4) * sizeof(float) + 2 * 512 * sizeof(float);
    const size_t mtile = MASK ? (size_t)NT * 16 * sizeof(float) : 0;  // masked sums per thread
    const size_t m = ktile > etile ? ktile : etile;
    return m > mtile ? m : mtile;
}

template <int BM, int BN, int WM, int PIPE, bool PF, int MINB, bool MASK = false, int WN = 2,
          bool FAST = false, bool RES = false>
static int32_t launch_cfg(IgemmParams& p, long maxM, hipStream_t st, const char* what) {
    const long ntm = (maxM + BM - 1) / BM;
    p.stat_ntm = (int)ntm;
    if (p.stat_partial) {
        PE_REQUIRE(ntm * p.nphase <= p.stat_capacity_rows, PE_E_INVALID,
                   "%s: statistics workspace holds %ld rows, %ld needed", what,
                   (long)p.stat_capacity_rows, ntm * p.nphase);
        if (p.stat_rows_out) *p.stat_rows_out = (int32_t)(ntm * p.nphase);
    }
    const long ntn = (p.N + BN - 1) / BN;
    const dim3 grid((unsigned)(ntm * ntn), (unsigned)p.nphase);
    constexpr size_t lds = lds_bytes<BM, BN, PIPE, WM * WN * 64, MASK>();
    static_assert(lds <= 160 * 1024, "tile does not fit the CU's LDS");
    if constexpr (lds > 64 * 1024) {
        const hipError_t e = ensure_dynamic_lds<
            &igemm_kernel<BM, BN, WM, PIPE, PF, MINB, MASK, WN, FAST, RES>>((int)lds);
        PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in (%d bytes): %s", what, (int)lds,
                   hipGetErrorString(e));
    }
    pe::launch(igemm_kernel<BM, BN, WM, PIPE, PF, MINB, MASK, WN, FAST, RES>, grid, WM * WN * 64, lds, st, p);
    return check_launch(what);
}

// Fills in the derived fields of a launch (single phase, packed tap words, non-temporal choice)
// and checks what every kernel of the family assumes.  maxM: GEMM rows of the largest phase.
static int32_t igemm_prepare(IgemmParams& p, long& maxM, const char* what) {
    if (p.wk_row == 0) p.wk_row = p.T * p.Cin;
    if (p.nphase <= 0) {  // single phase described by the scalar fields
        p.nphase = 1;
        p.ph_t0[0] = 0;
        p.ph_T[0] = (int16_t)p.T;
        p.ph_ooy[0] = (int16_t)p.ooy;
        p.ph_oox[0] = (int16_t)p.oox;
        p.ph_Ho[0] = p.Ho;
        p.ph_Wo[0] = p.Wo;
        p.ph_woff[0] = 0;
    }
    for (int i = 0; i < kMaxTaps; ++i)
        p.tap_info[i] = (int32_t)((uint32_t)(uint8_t)(int8_t)p.tap_dy[i] |
                                  ((uint32_t)(uint8_t)(int8_t)p.tap_dx[i] << 8) |
                                  ((uint32_t)p.tap_slot[i] << 16));
    PE_REQUIRE(p.x && p.w && p.y, PE_E_INVALID, "%s: NULL tensor pointer", what);
    PE_REQUIRE(p.B > 0 && p.Hi > 0 && p.Wi > 0 && p.N > 0 && p.nphase <= kMaxPhases, PE_E_INVALID,
               "%s: sizes must be positive", what);
    // a tap's K run is walked in steps of 64; a last step that is only part-filled loads the
    // zero block for the chunks past the run (weights and activations alike)
    PE_REQUIRE(p.Cin > 0 && p.Cin % 8 == 0, PE_E_UNSUPPORTED,
               "%s: K per tap (%d) must be a multiple of 8", what, p.Cin);
    PE_REQUIRE(p.cshift == 31 || p.Cin == BK, PE_E_INVALID,
               "%s: K runs that span pixels are %d long", what, BK);
    PE_REQUIRE(p.Cx % 8 == 0, PE_E_UNSUPPORTED, "%s: channels per pixel (%d) must be a multiple of 8",
               what, p.Cx);
    PE_REQUIRE(aligned(p.x, 16) && aligned(p.w, 16) && aligned(p.y, 16) && aligned(p.res, 16) &&
                   aligned(p.scale, 16) && aligned(p.shift, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    PE_REQUIRE(p.out_mode == 1 || p.ldc % 4 == 0, PE_E_UNSUPPORTED,
               "%s: output channel pitch must be a multiple of 4", what);
    maxM = 0;
    for (int i = 0; i < p.nphase; ++i) {
        PE_REQUIRE(p.ph_Ho[i] > 0 && p.ph_Wo[i] > 0 && p.ph_T[i] >= 1 &&
                       p.ph_t0[i] + p.ph_T[i] <= kMaxTaps,
                   PE_E_INVALID, "%s: bad phase %d", what, i);
        const long M = (long)p.B * p.ph_Ho[i] * p.ph_Wo[i];
        if (M > maxM) maxM = M;
    }
    if (p.stat_partial) {
        // (a residual only together with the mask from a stored activation, which adds it
        // before masking)
        PE_REQUIRE(!p.scale && !p.shift && (!p.res || p.mask_a) && !p.relu && p.out_mode == 0 &&
                       (p.N & 7) == 0 && (p.ldc & 7) == 0,
                   PE_E_UNSUPPORTED,
                   "%s: fused statistics need a plain bf16 NHWC output (no scale, shift, residual "
                   "or ReLU) and channel counts that are multiples of 8", what);
        for (int i = 1; i < p.nphase; ++i)
            PE_REQUIRE(p.ph_Ho[i] == p.ph_Ho[0] && p.ph_Wo[i] == p.ph_Wo[0], PE_E_UNSUPPORTED,
                       "%s: fused statistics need equally sized phases", what);
    }
    if (p.mask_z) {
        PE_REQUIRE(p.stat_partial && (p.mask_a || (p.mask_sc && p.mask_sh)), PE_E_INVALID,
                   "%s: the fused activation mask needs scale and shift (or the stored "
                   "activation) and a statistics workspace", what);
        PE_REQUIRE(aligned(p.mask_z, 16) && aligned(p.mask_sc, 16) && aligned(p.mask_sh, 16),
                   PE_E_INVALID, "%s: z, scale and shift must be 16-byte aligned", what);
    }
    // the kernel's reciprocal-based index split needs quotients below 2^20
    PE_REQUIRE(p.B < (1 << 20) && maxM / p.B < (1L << 22), PE_E_UNSUPPORTED,
               "%s: batch or map size out of range", what);
    for (int i = 0; i < kMaxTaps; ++i)
        PE_REQUIRE(p.tap_dy[i] >= -128 && p.tap_dy[i] <= 127 && p.tap_dx[i] >= -128 &&
                       p.tap_dx[i] <= 127,
                   PE_E_UNSUPPORTED, "%s: tap offset out of range", what);
    PE_REQUIRE((long)p.B * p.Hi * p.Wi < 0x7fffffffL && maxM < 0x7fffffffL &&
                   (long)p.B * p.OH * p.OW < 0x7fffffffL,
               PE_E_UNSUPPORTED, "%s: pixel count does not fit int32", what);

    {
        static const char* nts = getenv("PE_IGEMM_NT_MB");  // tuning aid: threshold in MB, 0 = off
        const long thr = nts ? atol(nts) : 64;
        p.nt_store = thr > 0 && (long)p.B * p.OH * p.OW * p.ldc * 2 >= thr * 1000000L;
    }
    return PE_OK;
}

// cheap staging addresses are possible (see the kernel's FAST parameter)
static bool igemm_fast_ok(const IgemmParams& p) {
    bool fast = p.cshift == 31 && p.Cin % BK == 0;
    for (int i = 0; i < p.nphase; ++i) fast = fast && p.ph_T[i] <= 32;
    return fast;
}

int32_t igemm_launch(const IgemmParams& pin, hipStream_t st, const char* what) {
    IgemmParams p = pin;
    long maxM = 0;
    {
        const int32_t rc = igemm_prepare(p, maxM, what);
        if (rc != PE_OK) return rc;
    }
    // tile choice: 128-pixel tiles unless they leave the chip under-filled; when even 64-pixel
    // tiles give fewer than two workgroups per CU, 64 x 64 tiles with the double-buffered
    // pipeline (each workgroup hides its own load latency) measured 15-25 % faster
    int bn = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);  // 32: the narrow branches of multi-branch nets
    const long ntn = (p.N + bn - 1) / bn;
    const long wg128 = ((maxM + 127) / 128) * ntn * p.nphase;
    int bm = (wg128 < 768 && maxM > 64) ? 64 : 128;
    int wm = 2, db = 0, pf = 0, minb = 4;
    if (bm == 64 && bn == 128 && ((maxM + 63) / 64) * ntn * p.nphase < 512) {
        bn = 64;
        db = 1;
        minb = 2;
    }
    // cheap addressing where the layer allows it (PE_IGEMM_FAST=0: off, for comparison)
    bool fast = igemm_fast_ok(p);
    {
        static const char* fs = getenv("PE_IGEMM_FAST");
        if (fs && atoi(fs) == 0) fast = false;
    }
    // Few tiles AND a long reduction (layer4 of the ResNet at batch 64: 384 workgroups, 32 to 72
    // K steps each): with at most two workgroups on a CU nothing else hides a K step's load
    // latency, so these run the three-stage ring (PE_IGEMM_RING=0: the double buffer instead).
    // 3x3 512->512 at 8x6: 44.5 -> 35.9 us, 1x1 2048->512: 21.7 -> 18.4 us; shorter reductions
    // measured the same either way.
    bool ring3 = false;
    if (db == 1 && fast) {
        long kmin = 1L << 40;
        for (int i = 0; i < p.nphase; ++i) kmin = std::min(kmin, (long)p.ph_T[i] * p.Cin);
        static const char* rs = getenv("PE_IGEMM_RING");
        ring3 = kmin >= 2048 && !(rs && atoi(rs) == 0);
    }
    int wn = 2;
    static const char* force = getenv("PE_IGEMM_CFG");  // "BM,BN,WM,PIPE,PF,MINB[,WN]": tuning aid
    if (force) sscanf(force, "%d,%d,%d,%d,%d,%d,%d", &bm, &bn, &wm, &db, &pf, &minb, &wn);
    // forced configurations apply only where the tile fits the layer (a 256-channel tile on a
    // 64-channel layer would multiply padding): otherwise the automatic choice stands
#define PE_CFG_RING(BM_, BN_, WM_, WN_, NS_, MINB_)                                        \
    if (force && bm == BM_ && bn == BN_ && wm == WM_ && wn == WN_ && db == NS_ &&           \
        minb == MINB_ && !p.mask_z) {                                                       \
        if (fast)                                                                           \
            return launch_cfg<BM_, BN_, WM_, NS_, false, MINB_, false, WN_, true>(p, maxM, st, what); \
        return launch_cfg<BM_, BN_, WM_, NS_, false, MINB_, false, WN_>(p, maxM, st, what); \
    }
    PE_CFG_RING(256, 128, 4, 2, 3, 1)
    PE_CFG_RING(256, 256, 4, 2, 2, 1)
    PE_CFG_RING(128, 128, 2, 2, 2, 2)
    PE_CFG_RING(64, 64, 2, 2, 3, 2)
#undef PE_CFG_RING
    if (!force && ring3 && !p.mask_a) {
        if (p.mask_z) return launch_cfg<64, 64, 2, 3, false, 2, true, 2, true>(p, maxM, st, what);
        return launch_cfg<64, 64, 2, 3, false, 2, false, 2, true>(p, maxM, st, what);
    }
#define PE_CFG(BM_, BN_, WM_, DB_, PF_, MINB_)                                        \
    if (bm == BM_ && bn == BN_ && wm == WM_ && db == (DB_ ? 1 : 0) && pf == (PF_ ? 1 : 0) && \
        minb == MINB_) {                                                               \
        return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_>(p, maxM, st, what);          \
    }
#define PE_CFG_FAST(BM_, BN_, WM_, DB_, PF_, MINB_)                                   \
    if (bm == BM_ && bn == BN_ && wm == WM_ && db == (DB_ ? 1 : 0) && pf == (PF_ ? 1 : 0) && \
        minb == MINB_) {                                                               \
        if (fast)                                                                      \
            return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_, false, 2, true>(p, maxM, st, what); \
        return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_>(p, maxM, st, what);          \
    }
#define PE_CFG_MASK(BM_, BN_, WM_, DB_, PF_, MINB_)                                   \
    if (p.mask_z && bm == BM_ && bn == BN_ && wm == WM_ && db == (DB_ ? 1 : 0) &&     \
        pf == (PF_ ? 1 : 0) && minb == MINB_) {                                        \
        if (p.mask_a && fast)                                                          \
            return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_, true, 2, true, true>(p, maxM, st, what); \
        if (p.mask_a)                                                                  \
            return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_, true, 2, false, true>(p, maxM, st, what); \
        if (fast)                                                                      \
            return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_, true, 2, true>(p, maxM, st, what); \
        return launch_cfg<BM_, BN_, WM_, DB_, PF_, MINB_, true>(p, maxM, st, what);    \
    }
    // The masked epilogue of the 128 x 128 tile needs some 156 registers; at four workgroups per
    // CU (a cap of 128) it spilled 17 to 28 of them to scratch memory, which cost the ResNet-50
    // training step 4 % (8.08 against 7.74 ms).  Three per CU fit without spilling; two measured
    // the same (PE_IGEMM_MASK_MINB: 2, 3 or 4, for comparison).
    if (p.mask_z && bm == 128 && bn == 128 && !force) {
        static const char* mm = getenv("PE_IGEMM_MASK_MINB");
        minb = mm ? atoi(mm) : 3;
    }
    PE_CFG_MASK(128, 128, 2, false, false, 4)
    PE_CFG_MASK(128, 128, 2, false, false, 3)
    PE_CFG_MASK(128, 128, 2, false, false, 2)
    PE_CFG_MASK(128, 64, 2, false, false, 4)
    PE_CFG_MASK(64, 128, 2, false, false, 4)
    PE_CFG_MASK(64, 64, 2, false, false, 4)
    PE_CFG_MASK(128, 32, 2, false, false, 4)
    PE_CFG_MASK(64, 32, 2, false, false, 4)
    PE_CFG_MASK(64, 64, 2, true, false, 2)
#undef PE_CFG_MASK
    PE_REQUIRE(!p.mask_z, PE_E_UNSUPPORTED, "%s: no masked kernel for tile %dx%d", what, bm, bn);
    // 64 output channels and 1024 or more 128-pixel tiles: 256 pixels x 64 channels, four waves along the
    // pixels, so a wave multiplies 64 x 64 and reads a third less LDS per MFMA than the 64 x 32
    // wave tile of the 128 x 64 workgroup: the 3x3 64 -> 64 layer at 64x48, batch 256, 103.5 ->
    // 91.9 us (PE_IGEMM_T256: 0 = off, 3 / 4 = workgroups per CU; 4 is the default).
    static const char* t256min = getenv("PE_IGEMM_T256_MIN");   // tuning aid: least 128-pixel tiles
    if (!force && bm == 128 && bn == 64 && p.N > 32 && wg128 >= (t256min ? atol(t256min) : 1024)) {
        static const char* ts = getenv("PE_IGEMM_T256");
        const int t256 = ts ? atoi(ts) : 4;
        if (t256 == 4) {
            if (fast) return launch_cfg<256, 64, 4, 0, false, 4, false, 1, true>(p, maxM, st, what);
            return launch_cfg<256, 64, 4, 0, false, 4, false, 1>(p, maxM, st, what);
        }
        if (t256 == 3) {
            if (fast) return launch_cfg<256, 64, 4, 0, false, 3, false, 1, true>(p, maxM, st, what);
            return launch_cfg<256, 64, 4, 0, false, 3, false, 1>(p, maxM, st, what);
        }
    }
    PE_CFG_FAST(128, 128, 2, false, false, 4)
    PE_CFG_FAST(128, 64, 2, false, false, 4)
    PE_CFG_FAST(64, 128, 2, false, false, 4)
    PE_CFG_FAST(64, 64, 2, false, false, 4)
    PE_CFG_FAST(128, 32, 2, false, false, 4)
    PE_CFG_FAST(64, 32, 2, false, false, 4)
    PE_CFG(128, 128, 2, false, true, 3)
    PE_CFG(128, 128, 2, false, true, 4)
    PE_CFG(128, 128, 2, true, true, 2)
    PE_CFG(128, 128, 2, true, false, 2)
    PE_CFG(256, 128, 4, false, true, 1)
    PE_CFG(256, 128, 4, true, true, 1)
    PE_CFG(128, 64, 2, false, true, 4)
    PE_CFG(64, 128, 2, true, false, 2)
    PE_CFG_FAST(64, 64, 2, true, false, 2)
#undef PE_CFG_FAST
#undef PE_CFG
    PE_REQUIRE(false, PE_E_UNSUPPORTED, "%s: no kernel for tile %dx%d wm %d db %d pf %d minb %d", what, bm, bn, wm, db, pf, minb);
    return check_launch(what);
}

// ---- layer + pointwise head in one launch ----------------------------------------------------
// A workgroup owns 128 GEMM rows (pixels) of one phase and walks BOTH 128-channel tiles of the
// layer's 256 outputs, one after the other; after each it multiplies the bf16 tile it would have
// stored by the head's weights for those channels (igemm_tile, HEAD), carrying the head's
// accumulators across the two tiles.  The layer's output never reaches memory: for the last
// transposed conv of the pose head that is a 2 x 402 MB round trip and a launch.
template <int MINB>
__global__ __launch_bounds__(256, MINB)
__attribute__((amdgpu_waves_per_eu(MINB, MINB))) void igemm_head_kernel(const IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile_m = xcd_tile_order(blockIdx.x, gridDim.x);
    const int ntn = p.N / 128;
#pragma unroll 1
    for (int nt = 0; nt < ntn; ++nt) {
        // opaque to the optimiser: otherwise everything that depends on the pixel tile alone
        // is hoisted out of this loop and stays in registers across the epilogue, which then
        // spills at four workgroups per CU
        int bid = tile_m * ntn + nt;
        asm volatile("" : "+s"(bid));
        igemm_tile<128, 128, 2, 0, false, false, 2, true, false, true>(p, smem, bid, blockIdx.y);
    }
}

int32_t igemm_head_launch(const IgemmParams& pin, hipStream_t st, const char* what) {
    IgemmParams p = pin;
    long maxM = 0;
    {
        const int32_t rc = igemm_prepare(p, maxM, what);
        if (rc != PE_OK) return rc;
    }
    PE_REQUIRE(p.head_w && p.head_out && p.head_N >= 1 && p.head_N <= 32, PE_E_INVALID,
               "%s: the head needs weights, an output and 1 to 32 channels", what);
    PE_REQUIRE(aligned(p.head_w, 16) && aligned(p.head_out, 4), PE_E_INVALID,
               "%s: head weights must be 16-byte aligned", what);
    PE_REQUIRE(p.N % 128 == 0 && p.N <= 512 && igemm_fast_ok(p) && !p.res && !p.stat_partial && !p.mask_z &&
                   p.out_mode == 0,
               PE_E_UNSUPPORTED,
               "%s: the fused head covers layers with 128, 256, 384 or 512 output channels, input "
               "channels a multiple of 64, no residual and no statistics", what);
    PE_REQUIRE((long)p.B * p.head_N * p.OH * p.OW < 0x7fffffffL, PE_E_UNSUPPORTED,
               "%s: head output does not fit 32-bit element offsets", what);
    const long ntm = (maxM + 127) / 128;
    const dim3 grid((unsigned)ntm, (unsigned)p.nphase);
    const size_t lds = (size_t)2 * 128 * ROWB + (size_t)2 * p.head_N * ROWB;
    static const char* mb = getenv("PE_IGEMM_HEAD_MINB");   // tuning aid: 3 or 4 workgroups per CU
    if (mb && atoi(mb) == 3)
        pe::launch(igemm_head_kernel<3>, grid, 256, lds, st, p);
    else
        pe::launch(igemm_head_kernel<4>, grid, 256, lds, st, p);
    return check_launch(what);
}

// ---- chained launches ------------------------------------------------------------------------
// Several convolutions over the SAME pixel tiles in one launch: a workgroup computes its 256
// pixels of stage 0 (all n-tiles, one after the other), then the same pixels of stage 1, whose
// input is what it has just stored, and so on.  Stages after the first are therefore 1x1
// stride-1 layers reading the previous stage's output: a pixel's inputs were all written by the
// same workgroup, whose stores have reached the L2 (vmcnt drained, workgroup barrier) before its
// next stage loads them.  What it buys: one launch instead of up to three, the matrix-core-
// bound stage of one workgroup runs beside the bandwidth-bound stage of its neighbours on the
// CU, and a stage's input comes from the cache that its producer just filled.
__global__ __launch_bounds__(256, 3)
__attribute__((amdgpu_waves_per_eu(3, 3))) void igemm_chain_kernel(const IgemmChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile_m = xcd_tile_order(blockIdx.x, gridDim.x);
    for (int s = 0; s < a.nstage; ++s) {
        const IgemmParams& p = a.st[s];
        const int ntn = (p.N + 63) / 64;
        for (int j = 0; j < ntn; ++j) {
            igemm_tile<256, 64, 4, 0, false, false, 1, true, false>(p, smem, tile_m * ntn + j, 0);
            // the tile's stores are in the L2 and every wave is done with the LDS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

int32_t igemm_chain_launch(const IgemmParams* stages, int nstage, hipStream_t st, const char* what) {
    PE_REQUIRE(stages != nullptr && nstage >= 1 && nstage <= kMaxChain, PE_E_INVALID,
               "%s: 1 to %d stages", what, kMaxChain);
    IgemmChainArgs a;
    a.nstage = nstage;
    long M = 0;
    for (int s = 0; s < nstage; ++s) {
        IgemmParams& p = a.st[s];
        p = stages[s];
        long maxM = 0;
        const int32_t rc = igemm_prepare(p, maxM, what);
        if (rc != PE_OK) return rc;
        PE_REQUIRE(p.nphase == 1 && p.out_mode == 0 && (p.N & 7) == 0 && (p.ldc & 7) == 0 &&
                       !p.stat_partial && !p.mask_z && igemm_fast_ok(p),
                   PE_E_UNSUPPORTED,
                   "%s: stage %d must be a single-phase bf16 NHWC layer with channel counts that "
                   "are multiples of 64 in and 8 out", what, s);
        if (s == 0) {
            M = maxM;
        } else {
            const IgemmParams& q = a.st[s - 1];
            PE_REQUIRE(maxM == M && p.T == 1 && p.tap_dy[0] == 0 && p.tap_dx[0] == 0 &&
                           p.sy == 1 && p.sx == 1 && p.Hi == q.OH && p.Wi == q.OW &&
                           p.Cx == q.ldc && p.Cin == q.N &&
                           p.x == static_cast<const uint16_t*>(q.y),
                       PE_E_INVALID,
                       "%s: stage %d must be a 1x1 stride-1 layer over the output of stage %d",
                       what, s, s - 1);
            a.st[s - 1].nt_store = 0;   // read again by this stage: keep it in the cache
        }
        PE_REQUIRE(p.osy == 1 && p.osx == 1 && p.ph_ooy[0] == 0 && p.ph_oox[0] == 0 &&
                       p.OH == p.ph_Ho[0] && p.OW == p.ph_Wo[0],
                   PE_E_UNSUPPORTED, "%s: stage %d scatters its output", what, s);
    }
    constexpr size_t lds = lds_bytes<256, 64, 0, 256, false>();
    pe::launch(igemm_chain_kernel, (unsigned)((M + 255) / 256), 256, lds, st, a);
    return check_launch(what);
}

}  // namespace pe
