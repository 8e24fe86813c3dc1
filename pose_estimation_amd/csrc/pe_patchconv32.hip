// 3x3 stride-1 pad-1 convolution with 32 input and 32 output channels (+ scale, shift, residual,
// ReLU), NHWC bf16: the 3x3 layers of the highest-resolution branch of the multi-branch
// (HRNet-style) backbone, for inference.
//
// Through the implicit-GEMM engine such a layer is a launch of thousands of workgroups with a
// K loop of six steps each: prologue and epilogue dominate (30 us for 57-85 MB of tensors at
// batch 64, 96x72).  Here a persistent workgroup keeps the whole filter in LDS (9 taps x 32 rows
// x 64 bytes = 18 KB), stages the 18 x 18-pixel input patch of a 16 x 16 output tile once
// (direct-to-LDS loads; a tap is an address offset) and works like pe_patchconv.hip: two groups
// of four waves in opposite phases, one multiplying tile t (a wave: 64 pixels x 32 channels,
// one MFMA K step per tap) while the other requests the patch of tile t + 1, converts tile
// t - 1 and stores it straight from registers.  With 32 channels a pixel is 64 bytes, so after
// the row swap (v_permlane16_swap_b32) one store instruction writes 16 WHOLE pixels, 1 KB
// contiguous.  The residual of a tile is requested when its multiply starts.  The workgroup
// needs 60 KB of LDS, so it shares a CU with the engine's workgroups of other streams.
//
// Weights are read in the engine's narrow packing [32][3][2][64] (two pixels per 64-element K
// run, zero past the filter's third column): tap (ky, kx) of output channel n starts at element
// n * 384 + ky * 128 + kx * 32.  The engine multiplies the same taps in the same order (its
// fourth, all-zero half run adds exact zeros), so the result equals pe_conv2d_fwd_bf16 bit for
// bit.
//
// LDS layout: 64-byte rows (a patch pixel, or a filter row) of four 16-byte slots; slot s of
// pixel column px (filter row r) holds channel chunk s ^ ((px >> 1) & 3): the 16-lane groups of
// a ds_read_b128 then touch 16 distinct 16-byte bank sets for every tap offset.
#include "pe_common.h"

namespace pe {

typedef __bf16 p32_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned p32_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned p32_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* p32_gptr_t;
typedef __attribute__((address_space(3))) void* p32_lptr_t;

constexpr int P32_T = 16;                             // output tile edge
constexpr int P32_P = P32_T + 2;                      // patch edge
constexpr int P32_NPIX = P32_P * P32_P;               // 324 patch pixels
constexpr int P32_NBLK = (P32_NPIX + 15) / 16;        // 21 wave instructions of 16 pixels
constexpr int P32_XBYTES = P32_NBLK * 1024;           // 21 504 per patch buffer
constexpr int P32_WBYTES = 9 * 32 * 64;               // 18 432
constexpr int P32_LDS = P32_WBYTES + 2 * P32_XBYTES + 2 * 32 * 4;

__device__ __attribute__((aligned(16))) uint32_t g_p32_zero16[4] = {0u, 0u, 0u, 0u};

struct Patch32Params {
    const uint16_t* x;     // [B][H][W][32]
    const uint16_t* w;     // [32][3][2][64] (narrow packing)
    const float* scale;
    const float* shift;
    const uint16_t* res;   // [B][H][W][32] or null
    uint16_t* y;           // [B][H][W][32]
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
};

struct P32Tile {
    int b, y0, x0;
};

__device__ __forceinline__ P32Tile p32_tile(const Patch32Params& p, int t) {
    const int tx0 = t % p.tiles_x;
    const int r = t / p.tiles_x;
    return P32Tile{r / p.tiles_y, (r % p.tiles_y) * P32_T, tx0 * P32_T};
}

__device__ __forceinline__ uint32_t p32_act_pack(float v0, float v1, const int relu) {
    if (relu) {
        v0 = fmaxf(v0, 0.0f);
        v1 = fmaxf(v1, 0.0f);
        if (relu == 2) {
            v0 = fminf(v0, 6.0f);
            v1 = fminf(v1, 6.0f);
        }
    }
    return pack_bf16x2_bits(v0, v1);
}

template <int RELU, bool AFFINE, bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_c32_kernel(const Patch32Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    char* const xb0 = smem + P32_WBYTES;
    float* ssl = reinterpret_cast<float*>(xb0 + 2 * P32_XBYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int g = wid >> 2;                               // group: 0 or 1
    const int cw = wid & 3;                               // wave inside the group
    const int lr = lane & 15, kc = lane >> 4;
    char* const xl = xb0 + g * P32_XBYTES;                // this group's patch buffer

    // ---- filter: 288 rows (tap * 32 + output channel) of 64 bytes, 16 rows per wave
    // instruction; lane l -> row l >> 2, slot l & 3
    for (int blk = wid; blk < 18; blk += 8) {             // uniform
        const int row = blk * 16 + (lane >> 2);
        const int tap = row >> 5, n = row & 31;
        const int c = (lane & 3) ^ ((row >> 1) & 3);
        const uint16_t* src = p.w + n * 384 + (tap / 3) * 128 + (tap % 3) * 32 + c * 8;
        __builtin_amdgcn_global_load_lds((p32_gptr_t)src, (p32_lptr_t)(wl + blk * 1024), 16, 0, 0);
    }
    if (tid < 32) {
        ssl[tid] = p.scale ? p.scale[tid] : 1.0f;
        ssl[32 + tid] = p.shift ? p.shift[tid] : 0.0f;
    }
    // tiles of this workgroup: blockIdx.x + k * gridDim.x, k = 0 .. n - 1; group g takes k = g mod 2
    const int n = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // Patch requests of this wave: blocks j = cw, cw + 4, ... of 16 pixels; the lane's part of
    // the address is worked out once.
    constexpr int NB = (P32_NBLK + 3) / 4;                // 6
    int prel[NB], pyx[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pp = (i * 4 + cw) * 16 + (lane >> 2);
        const int py = pp / P32_P, px = pp - py * P32_P;
        const int c = (lane & 3) ^ ((px >> 1) & 3);
        prel[i] = (py * p.W + px) * 64 + c * 16;
        pyx[i] = pp < P32_NPIX ? (py | (px << 16)) : 0x4000;   // a row no image has
    }
    const char* const zero = reinterpret_cast<const char*>(g_p32_zero16);
    auto issue_patch = [&](int t) {
        if (t >= p.ntiles) return;                        // uniform
        const P32Tile tl = p32_tile(p, t);
        const int sy0 = tl.y0 - 1, sx0 = tl.x0 - 1;
        const char* origin = reinterpret_cast<const char*>(p.x) +
                             (((long)tl.b * p.H + sy0) * p.W + sx0) * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i * 4 + cw >= P32_NBLK) break;            // uniform
            const bool ok = (unsigned)(sy0 + (pyx[i] & 0xffff)) < (unsigned)p.H &&
                            (unsigned)(sx0 + (pyx[i] >> 16)) < (unsigned)p.W;
            const char* src = ok ? origin + prel[i] : zero;
            __builtin_amdgcn_global_load_lds((p32_gptr_t)src, (p32_lptr_t)(xl + (i * 4 + cw) * 1024),
                                             16, 0, 0);
        }
    };
    issue_patch(blockIdx.x + g * gridDim.x);              // tile k = g (or none)

    // Fragment addresses: the per-lane part once, taps and rows as immediate offsets.
    const uint32_t lds_w = (uint32_t)(uintptr_t)(p32_lptr_t)wl;
    const uint32_t lds_x = (uint32_t)(uintptr_t)(p32_lptr_t)xl;
    const uint32_t abase = lds_w + lr * 64 + ((kc ^ ((lr >> 1) & 3)) << 4);
    uint32_t bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
        bbase[dx] = lds_x + (cw * 4 * P32_P + lr + dx) * 64 + ((kc ^ (((lr + dx) >> 1) & 3)) << 4);
    const int chb = ((kc & 1) * 32) + ((kc >> 1) * 16);   // lane's 16 bytes of a 64-byte pixel

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[2][4];
    p32_u32x4 R[RES ? 4 : 1];
    for (int f = 0; f <= n; ++f) {
        if (((f ^ g) & 1) == 0) {
            // ---- multiply tile f (if there is one)
            if (f < n) {
                if constexpr (RES) {
                    // the tile's residual, consumed a phase later; a pixel outside the image
                    // reads the nearest one inside (unconditional loads keep the queue countable)
                    const P32Tile tm = p32_tile(p, blockIdx.x + f * gridDim.x);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int oy = tm.y0 + cw * 4 + mt, ox = tm.x0 + lr;
                        const int cy = oy < p.H ? oy : p.H - 1, cx = ox < p.W ? ox : p.W - 1;
                        R[mt] = *reinterpret_cast<const p32_u32x4*>(
                            reinterpret_cast<const char*>(p.res) +
                            (((long)tm.b * p.H + cy) * p.W + cx) * 64 + chb);
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // The 6 fragment reads of tap s + 1 are issued before tap s is multiplied and
                // tap s waits with a counted lgkmcnt(6) (LDS reads return in order); reads and
                // waits are inline assembly so that the compiler cannot sink the reads to their
                // first use, and a wait names the registers it guards so that no MFMA moves
                // above it.
                p32_u32x4 fa[2][2], fb[2][4];
                auto load_step = [&](int s, p32_u32x4(&qa)[2], p32_u32x4(&qb)[4]) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(qa[a]) : "v"(abase), "n"(s * 2048 + a * 1024));
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(qb[mt])
                                     : "v"(bbase[s % 3]), "n"((mt + s / 3) * (P32_P * 64)));
                };
                load_step(0, fa[0], fb[0]);
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    p32_u32x4(&qa)[2] = fa[s & 1];
                    p32_u32x4(&qb)[4] = fb[s & 1];
                    if (s + 1 < 9) {
                        load_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                        asm volatile("s_waitcnt lgkmcnt(6)"
                                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qb[0]), "+v"(qb[1]),
                                       "+v"(qb[2]), "+v"(qb[3]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qb[0]), "+v"(qb[1]),
                                       "+v"(qb[2]), "+v"(qb[3]));
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(p32_bf16x8, qa[a]),
                                __builtin_bit_cast(p32_bf16x8, qb[mt]), acc[a][mt], 0, 0, 0);
                }
            }
            // every read of the patch has returned before the barrier frees the buffer
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (f >= 1) {
            // ---- finish tile f - 1.  The residual, requested a phase ago, is waited for HERE,
            // before the next patch is requested: waited for at its first use further down, the
            // compiler's vmcnt(0) would also wait for that patch.
            if constexpr (RES) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(R[mt]));
            }
            // next patch first (the loads fly under the conversion)
            issue_patch(blockIdx.x + (f + 1) * gridDim.x);
            const P32Tile tl = p32_tile(p, blockIdx.x + (f - 1) * gridDim.x);
            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ssl + kc * 4);
            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(ssl + 32 + kc * 4);
            const f32x4 sc1 = *reinterpret_cast<const f32x4*>(ssl + 16 + kc * 4);
            const f32x4 sh1 = *reinterpret_cast<const f32x4*>(ssl + 48 + kc * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 a = acc[0][mt], b = acc[1][mt];
                float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                if (AFFINE) {
                    v[0] = fmaf(a.x, sc0.x, sh0.x);
                    v[1] = fmaf(a.y, sc0.y, sh0.y);
                    v[2] = fmaf(a.z, sc0.z, sh0.z);
                    v[3] = fmaf(a.w, sc0.w, sh0.w);
                    v[4] = fmaf(b.x, sc1.x, sh1.x);
                    v[5] = fmaf(b.y, sc1.y, sh1.y);
                    v[6] = fmaf(b.z, sc1.z, sh1.z);
                    v[7] = fmaf(b.w, sc1.w, sh1.w);
                }
                if constexpr (RES) {
                    // the residual arrives as 8 contiguous channels per lane; the row swap that
                    // builds that layout for the store takes it apart again
                    p32_u32x4 r = R[mt];
                    asm volatile("" : "+v"(r));
                    const p32_u32x2 ra = __builtin_amdgcn_permlane16_swap(r.x, r.z, false, false);
                    const p32_u32x2 rb = __builtin_amdgcn_permlane16_swap(r.y, r.w, false, false);
                    const uint32_t ru[4] = {ra.x, rb.x, ra.y, rb.y};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[2 * i] += bf16_bits_to_float(ru[i] & 0xffffu);
                        v[2 * i + 1] += bf16_bits_to_float(ru[i] >> 16);
                    }
                }
                const uint32_t x0 = p32_act_pack(v[0], v[1], RELU), x1 = p32_act_pack(v[2], v[3], RELU);
                const uint32_t y0 = p32_act_pack(v[4], v[5], RELU), y1 = p32_act_pack(v[6], v[7], RELU);
                // odd rows of x <-> even rows of y: every lane gets 8 contiguous channels
                const p32_u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const p32_u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const uint4 o = make_uint4(s0.x, s1.x, s0.y, s1.y);
                const int oy = tl.y0 + cw * 4 + mt, ox = tl.x0 + lr;
                if (oy < p.H && ox < p.W)
                    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.y) +
                                              ((((long)tl.b * p.H + oy) * p.W + ox) * 64 + chb)) = o;
            }
            // The patch requests are older than the 4 stores, and vector memory operations
            // retire in order: a tile that lies wholly inside the image issued all of them, so
            // "all but 4" means the patch has landed; an edge tile may have skipped some.
            const bool full = tl.y0 + P32_T <= p.H && tl.x0 + P32_T <= p.W;
            if (full) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

}  // namespace pe

using namespace pe;

template <int R, bool A, bool S>
static hipError_t p32_launch_variant(const Patch32Params& p, unsigned nwg, hipStream_t stream) {
    const hipError_t e = pe::ensure_dynamic_lds<&conv3x3_c32_kernel<R, A, S>>(P32_LDS);
    if (e == hipSuccess)
        pe::launch(conv3x3_c32_kernel<R, A, S>, nwg, 512, (size_t)P32_LDS, stream, p);
    return e;
}

extern "C" int32_t pe_conv3x3_c32_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                           const void* w_narrow, const float* scale,
                                           const float* shift, const void* residual, int32_t relu,
                                           void* y, pe_stream_t stream) {
    const char* what = "pe_conv3x3_c32_fwd_bf16";
    PE_REQUIRE(x && w_narrow && y, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0, PE_E_INVALID, "%s: sizes must be positive", what);
    PE_REQUIRE(H <= 8192 && W <= 8192, PE_E_UNSUPPORTED, "%s: images above 8192 x 8192", what);
    PE_REQUIRE(relu >= 0 && relu <= 2, PE_E_INVALID, "%s: relu must be 0, 1 or 2", what);
    PE_REQUIRE(aligned(x, 16) && aligned(w_narrow, 16) && aligned(y, 16) && aligned(scale, 16) &&
                   aligned(shift, 16) && aligned(residual, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    Patch32Params p;
    p.x = static_cast<const uint16_t*>(x);
    p.w = static_cast<const uint16_t*>(w_narrow);
    p.scale = scale;
    p.shift = shift;
    p.res = static_cast<const uint16_t*>(residual);
    p.y = static_cast<uint16_t*>(y);
    p.B = B;
    p.H = H;
    p.W = W;
    p.tiles_x = (W + P32_T - 1) / P32_T;
    p.tiles_y = (H + P32_T - 1) / P32_T;
    const long ntiles = (long)B * p.tiles_x * p.tiles_y;
    PE_REQUIRE(ntiles < 0x3fffffffL, PE_E_UNSUPPORTED, "%s: too many tiles", what);
    p.ntiles = (int)ntiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned nwg = (unsigned)(ntiles < cus ? ntiles : cus);   // one persistent workgroup per CU
    const bool affine = scale != nullptr || shift != nullptr;
    const bool res = residual != nullptr;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipSuccess;
#define P32_CASE(R, A, S) \
    if (relu == R && affine == A && res == S) e = p32_launch_variant<R, A, S>(p, nwg, st);
    P32_CASE(0, false, false) P32_CASE(1, false, false) P32_CASE(2, false, false)
    P32_CASE(0, true, false) P32_CASE(1, true, false) P32_CASE(2, true, false)
    P32_CASE(0, false, true) P32_CASE(1, false, true) P32_CASE(2, false, true)
    P32_CASE(0, true, true) P32_CASE(1, true, true) P32_CASE(2, true, true)
#undef P32_CASE
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in: %s", what, hipGetErrorString(e));
    return check_launch(what);
}
