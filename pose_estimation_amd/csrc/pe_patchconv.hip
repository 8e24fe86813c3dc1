// 3x3 stride-1 pad-1 convolution with 64 input and 64 output channels (+ scale, shift, ReLU),
// NHWC bf16: the 3x3 layers of the first ResNet stage, for inference.
//
// Through the implicit-GEMM engine such a layer re-stages every activation tile once per tap (9
// times) for only 64 output channels and is bound by L2 -> LDS traffic, not by the matrix cores
// (630 TFLOP/s at batch 256).  Here one persistent workgroup per CU keeps the WHOLE filter in
// LDS (9 taps x 64 rows x 128 bytes = 72 KB, loaded once) and stages, per 16 x 16-pixel output
// tile, the 18 x 18-pixel input patch ONCE (128 bytes per pixel, direct-to-LDS loads); a tap is
// an address offset into the patch.  Staged bytes per output tile fall from 360 KB to 41 KB.
//
// Two groups of waves in opposite phases.  All eight waves multiply: group g (waves 4g .. 4g+3,
// one per SIMD) owns patch buffer g and every second tile of the workgroup.  Time is cut into
// phases by workgroup barriers; in phase f group (f & 1) multiplies tile f from its buffer while
// the other group finishes tile f - 1: it requests the patch of tile f + 1 into its own buffer
// (free since the barrier: its reads ended with its multiply), then converts its accumulators
// and stores them straight from registers.  A SIMD therefore always holds one wave that
// multiplies and one that converts, stores and requests, and the matrix cores do not wait for
// an epilogue.  (With four multiplying waves that all converted at the same barrier, and four
// helper waves moving data, the same layer took 82 us; this form 56-60 us; the multiplies alone
// 38 us.)
//
// No LDS staging of the output: a lane's four accumulators of (channel tile a, pixel tile mt)
// are channels 16 a + 4 kc .. + 3 of pixel 16 mt + lr (kc = lane >> 4).  Packed to bf16 they
// are 8 bytes; v_permlane16_swap_b32 exchanges the odd 16-lane rows of one register with the
// even rows of another, which, applied to the packed values of channel tiles a and a + 1, hands
// every lane 16 contiguous bytes: rows 0 / 2 channels 16 a + {0..7, 8..15}, rows 1 / 3 the same
// of tile a + 1.  One store instruction then writes 64 contiguous bytes of each of 16 pixels.
//
// Layout: a patch pixel is one 128-byte LDS row of 8 x 16-byte slots, slot s holding channel
// chunk s ^ (patch column & 7).  The 16-lane groups of a ds_read_b128 read consecutive pixels
// of one patch row (rows are 18 pixels, an even count, so pixel parity = column parity), which
// makes (pixel parity, slot) distinct inside a group: conflict-free, as in the engine's tiles,
// and a tap's row offset becomes a compile-time constant.  The filter planes use the engine's
// weight-tile layout.
//
// MFMA operand roles, K order (tap by tap, two 32-channel halves each) and the epilogue's
// operations are those of the engine, so the result equals pe_conv2d_fwd_bf16 bit for bit.
#include "pe_common.h"

namespace pe {

typedef __bf16 pc_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pc_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* pc_gptr_t;
typedef __attribute__((address_space(3))) void* pc_lptr_t;

constexpr int PC_TX = 16;                         // output tile width
constexpr int PC_P = PC_TX + 2;                   // patch width
constexpr int PC_WBYTES = 9 * 64 * 128;           // 73 728: the 3x3 filter
constexpr int PC_N3 = 256;                        // output channels of the fused pointwise layer
constexpr int PC_W3BYTES = PC_N3 * 128;           // 32 768: its weights
constexpr int PC_W1BYTES = 64 * PC_N3 * 2;        // 32 768: the next block's 256 -> 64 layer

// Geometry of a variant.  Alone, the 3x3 layer works on 16 x 16 tiles (4 rows per wave).  With
// the pointwise layer the tile is 8 rows high (2 rows per wave): its smaller patch buffers make
// room for the pointwise weights in LDS, and a wave's share of the 256-channel output (32 pixels)
// leaves registers for residual loads several chunks ahead.
//
// With the NEXT block's first pointwise layer (256 -> 64) in the same launch as well (PW 4) its
// weights take another 32 KB of LDS.  That leaves room for ONE patch buffer of a 4-row tile (one
// row per wave), shared by the two groups: a phase then has a second barrier, behind the
// multiplying group's last read of the buffer, after which the other group requests the patch
// it multiplies in the next phase (the multiply is a small part of such a phase; the rest of
// it belongs to the finishing group's loads and stores).
template <int PW>
struct PcGeo {
    static constexpr bool NEXT = PW >= 4;
    static constexpr bool PROJ = PW == 3;
    static constexpr int TY = NEXT ? 4 : PW ? 8 : 16; // output tile height
    static constexpr int MT = TY / 4;                 // 16-pixel rows per wave
    static constexpr int NBUF = NEXT ? 1 : 2;         // patch buffers
    static constexpr int NPIX = (TY + 2) * PC_P;      // patch pixels: 324 / 180 / 108
    static constexpr int NINSTR = (NPIX + 7) / 8;     // wave instructions of 8 pixels: 41 / 23 / 14
    static constexpr int XBYTES = NINSTR * 1024;      // per patch buffer: 41 984 / 23 552 / 14 336
    static constexpr int W3B = PW ? PC_W3BYTES : 0;
    static constexpr int W1B = NEXT ? PC_W1BYTES : 0;
    static constexpr int LDS = PC_WBYTES + W3B + W1B + NBUF * XBYTES + 2 * 64 * 4 +
                               (PW ? 2 * PC_N3 * 4 : 0) + (PROJ ? 2 * PC_N3 * 4 : 0) +
                               (NEXT ? 2 * 64 * 4 : 0);
};

__device__ __attribute__((aligned(16))) uint32_t g_pc_zero16[4] = {0u, 0u, 0u, 0u};
// where the lanes of pixels outside the image store (never read): lets every store of the fused
// kernel be unconditional, so that the compiler can count the operations in flight
__device__ __attribute__((aligned(16))) uint32_t g_pc_sink[512];

struct PatchConvParams {
    const uint16_t* x;     // [B][H][W][64]
    const uint16_t* w;     // [64][3][3][64]
    const float* scale;
    const float* shift;
    uint16_t* y;           // [B][H][W][64], or [B][H][W][256] with the pointwise layer
    const uint16_t* w3;    // pointwise layer: [256][64], its scale / shift [256] and the
    const float* scale3;   // residual [B][H][W][256] added before its ReLU
    const float* shift3;
    const uint16_t* res;
    const uint16_t* x0;    // projected residual (PW 3): the block's input [B][H][W][64], the
    const uint16_t* wd;    // projection's weights [256][64] and its scale / shift [256]
    const float* scaled;
    const float* shiftd;
    const uint16_t* w1;    // the next block's first pointwise layer (PW 4): weights [64][256],
    const float* scale1;   // scale / shift [64] (ReLU behind them) and its output
    const float* shift1;   // [B][H][W][64]
    uint16_t* z;
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
};

struct PcTile {
    int b, y0, x0;
};

template <int TY>
__device__ __forceinline__ PcTile pc_tile(const PatchConvParams& p, int t) {
    const int tx0 = t % p.tiles_x;
    const int r = t / p.tiles_x;
    return PcTile{r / p.tiles_y, (r % p.tiles_y) * TY, tx0 * PC_TX};
}

__device__ __forceinline__ uint32_t pc_act_pack(float v0, float v1, float s0, float s1, float h0,
                                                float h1, const bool affine, const int relu) {
    if (affine) {
        v0 = fmaf(v0, s0, h0);
        v1 = fmaf(v1, s1, h1);
    }
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

template <int RELU, bool AFFINE, int PW>
__global__ __launch_bounds__(512, 1) void conv3x3_c64_kernel(const PatchConvParams p) {
    using G = PcGeo<PW>;
    constexpr int MT = G::MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                      // 3x3 filter
    char* w3l = smem + PC_WBYTES;                         // pointwise weights (PW only)
    char* w1l = smem + PC_WBYTES + G::W3B;                // the next block's pointwise weights (NEXT)
    char* const xb0 = smem + PC_WBYTES + G::W3B + G::W1B; // the patch buffer(s)
    float* ssl = reinterpret_cast<float*>(xb0 + G::NBUF * G::XBYTES);
    float* ssl3 = ssl + 128;
    float* ssld = ssl3 + 2 * PC_N3;                       // PW 3 only
    constexpr bool PROJ = G::PROJ;
    constexpr bool NEXT = G::NEXT;
    float* ssl1 = ssld + (PROJ ? 2 * PC_N3 : 0);          // NEXT only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int g = wid >> 2;                               // group: 0 or 1
    const int cw = wid & 3;                               // wave inside the group
    const int lr = lane & 15, kc = lane >> 4;
    char* const xl = xb0 + (G::NBUF == 2 ? g : 0) * G::XBYTES;   // this group's patch buffer

    // ---- weights: rows of 128 bytes; lane l of a wave instruction -> row l>>3, slot l&7, which
    // receives channel chunk (l & 7) ^ (row & 7)
    {
        const int cc = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int R = i * 64 + (tid >> 3);            // tap * 64 + output channel
            const uint16_t* src = p.w + (long)(R & 63) * 576 + (R >> 6) * 64 + cc * 8;
            __builtin_amdgcn_global_load_lds((pc_gptr_t)src,
                                             (pc_lptr_t)(wl + (i * 64 + 8 * wid) * 128), 16, 0, 0);
        }
        if constexpr (PW != 0) {
#pragma unroll
            for (int i = 0; i < PC_N3 / 64; ++i) {
                const int R = i * 64 + (tid >> 3);        // output channel of the pointwise layer
                const uint16_t* src = p.w3 + (long)R * 64 + cc * 8;
                __builtin_amdgcn_global_load_lds(
                    (pc_gptr_t)src, (pc_lptr_t)(w3l + (i * 64 + 8 * wid) * 128), 16, 0, 0);
            }
        }
        if constexpr (NEXT) {
            // Eight planes, one per 32-channel K step, of 64 rows (output channels) x 64 bytes;
            // a wave instruction fills 16 rows: lane l -> row l >> 2, slot l & 3, which receives
            // channel chunk (l & 3) ^ ((row >> 1) & 3) (conflict-free fragment reads, as the
            // filter rows of pe_patchconv32.hip).
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int blk = i * 8 + wid;              // 0..31: plane blk >> 2, rows 16 (blk & 3) ..
                const int row = (blk & 3) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 1) & 3);
                const uint16_t* src = p.w1 + row * PC_N3 + (blk >> 2) * 32 + c * 8;
                __builtin_amdgcn_global_load_lds((pc_gptr_t)src, (pc_lptr_t)(w1l + blk * 1024), 16,
                                                 0, 0);
            }
        }
    }
    if (tid < 64) {
        ssl[tid] = p.scale ? p.scale[tid] : 1.0f;
        ssl[64 + tid] = p.shift ? p.shift[tid] : 0.0f;
    }
    if (PW && tid < PC_N3) {
        ssl3[tid] = p.scale3 ? p.scale3[tid] : 1.0f;
        ssl3[PC_N3 + tid] = p.shift3 ? p.shift3[tid] : 0.0f;
        if constexpr (PROJ) {
            ssld[tid] = p.scaled ? p.scaled[tid] : 1.0f;
            ssld[PC_N3 + tid] = p.shiftd ? p.shiftd[tid] : 0.0f;
        }
    }
    if (NEXT && tid < 64) {
        ssl1[tid] = p.scale1 ? p.scale1[tid] : 1.0f;
        ssl1[64 + tid] = p.shift1 ? p.shift1[tid] : 0.0f;
    }
    // tiles of this workgroup: blockIdx.x + k * gridDim.x, k = 0 .. n - 1; group g takes k = g mod 2
    const int n = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // Patch requests of this wave: blocks j = cw, cw + 4, ... of 8 pixels.  What depends on the
    // lane alone (its pixel's row and column in the patch, its byte offset from the patch
    // origin) is worked out once; per tile a request costs two compares and a select.  The
    // variant with the projected residual is short of registers and works it out per request
    // instead (a spilled value would be worse than both: its reload is a vector memory
    // operation, and waiting for it drains every load and store in flight).
    constexpr int NB = (G::NINSTR + 3) / 4;               // 11 / 6
    constexpr bool KEEP = !PROJ;
    int prel[KEEP ? NB : 1], pyx[KEEP ? NB : 1];
    auto patch_lane = [&](int i, int& rel, int& yx) {
        int ln = lane;
        if constexpr (!KEEP) asm volatile("" : "+v"(ln));   // per request, not hoisted
        const int pp = (i * 4 + cw) * 8 + (ln >> 3);
        const int py = pp / PC_P, px = pp - py * PC_P;
        const int cc = (ln & 7) ^ (px & 7);
        rel = (py * p.W + px) * 128 + cc * 16;
        yx = pp < G::NPIX ? (py | (px << 16)) : 0x4000;   // a row no image has
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < NB; ++i) patch_lane(i, prel[i], pyx[i]);
    }
    const char* const zero = reinterpret_cast<const char*>(g_pc_zero16);
    auto issue_patch = [&](int t) {
        if (t >= p.ntiles) return;                        // uniform
        const PcTile tl = pc_tile<G::TY>(p, t);
        const int sy0 = tl.y0 - 1, sx0 = tl.x0 - 1;
        const char* origin = reinterpret_cast<const char*>(p.x) +
                             (((long)tl.b * p.H + sy0) * p.W + sx0) * 128;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i * 4 + cw >= G::NINSTR) break;           // uniform
            int rel, yx;
            if constexpr (KEEP) {
                rel = prel[i];
                yx = pyx[i];
            } else {
                patch_lane(i, rel, yx);
            }
            const bool ok = (unsigned)(sy0 + (yx & 0xffff)) < (unsigned)p.H &&
                            (unsigned)(sx0 + (yx >> 16)) < (unsigned)p.W;
            const char* src = ok ? origin + rel : zero;
            __builtin_amdgcn_global_load_lds((pc_gptr_t)src, (pc_lptr_t)(xl + (i * 4 + cw) * 1024),
                                             16, 0, 0);
        }
    };
    // tile k = g (or none); with one shared buffer group 1 asks for its first patch in phase 0
    if (!NEXT || g == 0) issue_patch(blockIdx.x + g * gridDim.x);

    // Fragment addresses, all of the per-lane part worked out once: a K step then issues its
    // LDS reads with immediate offsets and no vector arithmetic (one wave per SIMD multiplies,
    // and a wave issues at most one instruction every four cycles, so every address
    // instruction in the loop displaces an MFMA).  The patch's slot swizzle goes by the pixel's
    // COLUMN (rows are 18 pixels = an even count, which keeps reads conflict-free), so that a
    // tap's row offset is a constant.
    const int aoff = lr * 128 + ((kc ^ (lr & 7)) << 4);
    const uint32_t lds_w = (uint32_t)(uintptr_t)(pc_lptr_t)wl;     // LDS byte addresses
    const uint32_t lds_x = (uint32_t)(uintptr_t)(pc_lptr_t)xl;
    uint32_t abase[2], bbase[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        abase[kk] = lds_w + (aoff ^ (kk << 6));
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            bbase[dx][kk] = lds_x + (((cw * MT * PC_P + lr + dx) * 128 +
                                      ((kc ^ ((lr + dx) & 7)) << 4)) ^ (kk << 6));
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[4][MT];
    // residual of the tile being multiplied (PW): requested when its multiply starts, consumed
    // a phase later by the same waves, so the loads of BOTH groups are in flight at any time
    constexpr int NQ = PC_N3 / 32;                        // chunks of 32 output channels
    constexpr int QMID = 2;                               // NEXT: chunks finished before the shared
                                                          // buffer is asked for again
    // (PROJ: the projection's output for the tile, worked out by the multiplying group behind
    // its multiply, see there)
    pc_u32x4 R[PW ? NQ : 1][MT];
    // projected residual (PW 3): the block input's pixels of the tile as an MFMA operand, and
    // three 32-channel chunks of the projection's weights at a time (chunk q in slot q % 3; a
    // slot is refilled with chunk q + 3 as soon as chunk q has been multiplied)
    constexpr int ADS = 2;
    pc_u32x4 Bx[PROJ ? MT : 1][2], AD[PROJ ? ADS : 1][2][2];
    const int wdo = (lr * 64 + kc * 8) * 2;               // this lane's byte offset in a 16-row block of wd
    auto load_ad = [&](int q) {
        // (the chunk's offset made opaque: one address per chunk, worked out where it is used,
        // instead of 32 addresses kept in registers)
        unsigned o = wdo + q * 32 * 64 * 2;
        asm volatile("" : "+v"(o));
        const char* base = reinterpret_cast<const char*>(p.wd) + o;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                AD[PROJ ? q % ADS : 0][nt][kk] =
                    *reinterpret_cast<const pc_u32x4*>(base + (nt * 16 * 64 + kk * 32) * 2);
    };
    const int chb = ((kc & 1) * 32) + ((kc >> 1) * 16);   // lane's 16 bytes inside 64 of a pixel
    for (int f = 0; f <= n; ++f) {
        if (((f ^ g) & 1) == 0) {
            // ---- multiply tile f (if there is one)
            if (f < n) {
                if constexpr (PROJ) {
                    const PcTile tm = pc_tile<G::TY>(p, blockIdx.x + f * gridDim.x);
                    int k16 = kc * 16;                    // (opaque: worked out here, not kept)
                    asm volatile("" : "+v"(k16));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int oy = tm.y0 + cw * MT + mt, ox = tm.x0 + lr;
                        const int cy = oy < p.H ? oy : p.H - 1, cx = ox < p.W ? ox : p.W - 1;
                        const char* xp = reinterpret_cast<const char*>(p.x0) +
                                         (((long)tm.b * p.H + cy) * p.W + cx) * 128 + k16;
                        Bx[mt][0] = *reinterpret_cast<const pc_u32x4*>(xp);
                        Bx[mt][1] = *reinterpret_cast<const pc_u32x4*>(xp + 64);
                    }
#pragma unroll
                    for (int q = 0; q < ADS; ++q) load_ad(q);
                } else if constexpr (PW != 0) {
                    // A pixel outside the image reads the residual of the nearest one inside
                    // (an unconditional load keeps the queue countable).  Ordinary loads: the
                    // two 64-byte halves of a line are asked for by different instructions, and
                    // streaming (non-temporal) loads measured 10 % slower.
                    const PcTile tm = pc_tile<G::TY>(p, blockIdx.x + f * gridDim.x);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int oy = tm.y0 + cw * MT + mt, ox = tm.x0 + lr;
                        const int cy = oy < p.H ? oy : p.H - 1, cx = ox < p.W ? ox : p.W - 1;
                        const char* rp = reinterpret_cast<const char*>(p.res) +
                                         (((long)tm.b * p.H + cy) * p.W + cx) * (PC_N3 * 2) + chb;
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            R[q][mt] = *reinterpret_cast<const pc_u32x4*>(rp + q * 64);
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // The fragment reads of K step s + 1 are issued before step s is multiplied
                // and step s waits with a COUNTED lgkmcnt: LDS reads return in order, so "all
                // but the youngest 4 + MT" is exactly step s.  The reads and the waits are
                // inline assembly because the compiler, left to schedule them, sinks every read
                // down to its first use and waits for an empty queue (one exposed LDS round
                // trip per 8 MFMAs: measured 55 % of the register-only MFMA rate).  A wait
                // names the registers it guards as operands, so no MFMA can move above it.
                pc_u32x4 fa[2][4], fb[2][MT];
                auto load_step = [&](int s, pc_u32x4(&qa)[4], pc_u32x4(&qb)[MT]) {
                    const int tap = s >> 1, kk = s & 1;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int off = tap * 8192 + a * 2048;
                        if (off < 65536)
                            asm volatile("ds_read_b128 %0, %1 offset:%2"
                                         : "=v"(qa[a]) : "v"(abase[kk]), "n"(off < 65536 ? off : 0));
                        else {
                            // the last tap lies past the 16-bit offset field: its base is
                            // made on the spot (two registers less across the whole kernel)
                            uint32_t hi = abase[kk];
                            asm volatile("" : "+v"(hi));
                            hi += 65536;
                            asm volatile("ds_read_b128 %0, %1 offset:%2"
                                         : "=v"(qa[a]) : "v"(hi), "n"(off >= 65536 ? off - 65536 : 0));
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(qb[mt])
                                     : "v"(bbase[tap % 3][kk]), "n"((mt + tap / 3) * (PC_P * 128)));
                };
                load_step(0, fa[0], fb[0]);
#pragma unroll
                for (int s = 0; s < 18; ++s) {
                    pc_u32x4(&qa)[4] = fa[s & 1];
                    pc_u32x4(&qb)[MT] = fb[s & 1];
                    if (s + 1 < 18) load_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                    if constexpr (MT == 4) {
                        if (s + 1 < 18)
                            asm volatile("s_waitcnt lgkmcnt(8)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]));
                    } else if constexpr (MT == 2) {
                        if (s + 1 < 18)
                            asm volatile("s_waitcnt lgkmcnt(6)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]), "+v"(qb[1]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]), "+v"(qb[1]));
                    } else {
                        if (s + 1 < 18)
                            asm volatile("s_waitcnt lgkmcnt(5)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]),
                                           "+v"(qb[0]));
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(pc_bf16x8, qa[a]),
                                __builtin_bit_cast(pc_bf16x8, qb[mt]), acc[a][mt], 0, 0, 0);
                }
            }
            // every read of the patch has returned before the barrier frees the buffer
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) {
                __builtin_amdgcn_s_barrier();             // the shared buffer is free from here
                asm volatile("" ::: "memory");
            }
            if constexpr (PROJ) {
                // The projection of the skip path for the tile just multiplied, while this group
                // would otherwise wait for the other one: rounded to bf16 as its own launch
                // would store it and kept in registers until this group finishes the tile, a
                // phase later.  (Inside the chunk loop of the finishing phase the weight chunks
                // would queue behind that phase's stores: vector memory operations retire in
                // order.)
                if (f < n) {
                    int k4 = kc * 4;
                    asm volatile("" : "+v"(k4));
                    const float* sdp = ssld + k4;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        f32x4 cd[2][MT];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) cd[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
                                    cd[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(pc_bf16x8, AD[q % ADS][nt][kk]),
                                        __builtin_bit_cast(pc_bf16x8, Bx[mt][kk]), cd[nt][mt], 0, 0, 0);
                        // the slot's next chunk, behind fences so that it stays here
                        asm volatile("" ::: "memory");
                        if (q + ADS < NQ) load_ad(q + ADS);
                        asm volatile("" ::: "memory");
                        // scale, shift, no ReLU
                        const f32x4 d0 = *reinterpret_cast<const f32x4*>(sdp + q * 32);
                        const f32x4 e0 = *reinterpret_cast<const f32x4*>(sdp + PC_N3 + q * 32);
                        const f32x4 d1 = *reinterpret_cast<const f32x4*>(sdp + q * 32 + 16);
                        const f32x4 e1 = *reinterpret_cast<const f32x4*>(sdp + PC_N3 + q * 32 + 16);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const f32x4 a = cd[0][mt], b = cd[1][mt];
                            R[q][mt] = pc_u32x4{
                                pack_bf16x2_bits(fmaf(a.x, d0.x, e0.x), fmaf(a.y, d0.y, e0.y)),
                                pack_bf16x2_bits(fmaf(a.z, d0.z, e0.z), fmaf(a.w, d0.w, e0.w)),
                                pack_bf16x2_bits(fmaf(b.x, d1.x, e1.x), fmaf(b.y, d1.y, e1.y)),
                                pack_bf16x2_bits(fmaf(b.z, d1.z, e1.z), fmaf(b.w, d1.w, e1.w))};
                        }
                    }
                }
            }
        } else if (NEXT && f == 0) {
            // shared buffer: this group's first patch, behind the other group's first multiply
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_patch(blockIdx.x + (f + 1) * gridDim.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (f >= 1) {
            // ---- finish tile f - 1: next patch first (the loads fly under the conversion); with
            // the shared buffer, behind the barrier in the chunk loop below
            if constexpr (!NEXT) issue_patch(blockIdx.x + (f + 1) * gridDim.x);
            const PcTile tl = pc_tile<G::TY>(p, blockIdx.x + (f - 1) * gridDim.x);
            const bool full = tl.y0 + G::TY <= p.H && tl.x0 + PC_TX <= p.W;
            if constexpr (PW != 0) {
                // ---- the 64-channel tile, rounded to bf16 as the 3x3 layer's output would be,
                // is the activation operand of the pointwise layer; it never leaves registers.
                // (The scale / shift tables and the pointwise weights lie past the 16-bit
                // offset field of an LDS read: one opaque per-lane OFFSET each, or every row's
                // address is kept in a register.  An offset, not a pointer: the pointer must
                // stay visibly an LDS one, or the reads become flat loads that cannot be
                // waited for by count.)
                int ln = lane;                            // worked out again here, from the lane
                asm volatile("" : "+v"(ln));              // number alone, instead of being kept
                const int lr_ = ln & 15, kc_ = ln >> 4;
                const int k4 = kc_ * 4;
                const int a3o0 = lr_ * 128 + ((kc_ ^ (lr_ & 7)) << 4), a3o1 = a3o0 ^ 64;
                const float* s2p = ssl + k4;
                uint32_t P[4][MT][2];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(s2p + a * 16);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(s2p + 64 + a * 16);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 c = acc[a][mt];
                        P[a][mt][0] = pc_act_pack(c.x, c.y, sc.x, sc.y, sh.x, sh.y, AFFINE, RELU);
                        P[a][mt][1] = pc_act_pack(c.z, c.w, sc.z, sc.w, sh.z, sh.w, AFFINE, RELU);
                    }
                }
                // B fragment of (pixel row mt, 32-channel half kk): lane (lr, kc) needs channels
                // 32 kk + 8 kc .. + 7 of pixel lr, which sit in lanes (lr, 2 (kc & 1)) and the row
                // after it, in channel tile 2 kk + (kc >> 1).  Two row swaps bring them home: the
                // upper 32 lanes of tile 2 kk against the lower 32 of tile 2 kk + 1, then odd
                // against even rows.
                pc_u32x4 Bc[MT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const pc_u32x2 t = __builtin_amdgcn_permlane32_swap(
                                P[2 * kk][mt][e], P[2 * kk + 1][mt][e], false, false);
                            const pc_u32x2 u = __builtin_amdgcn_permlane16_swap(t.x, t.y, false, false);
                            Bc[mt][kk][e] = u.x;
                            Bc[mt][kk][2 + e] = u.y;
                        }
                // One pointer per pixel row into the output; a chunk of 32 channels is an immediate
                // offset from it (q * 64 bytes).  A pixel outside the image stores to a sink: an
                // unconditional store keeps the queue countable.
                char* yp[MT];
                int chbo = chb;                           // (opaque: not folded into a kept address)
                asm volatile("" : "+v"(chbo));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int oy = tl.y0 + cw * MT + mt, ox = tl.x0 + lr;
                    const bool ok = oy < p.H && ox < p.W;
                    yp[mt] = ok ? reinterpret_cast<char*>(p.y) +
                                      ((((long)tl.b * p.H + oy) * p.W + ox) * (PC_N3 * 2) + chbo)
                                : reinterpret_cast<char*>(g_pc_sink) + lane * 16;
                }
                f32x4 acc1[4][MT];                        // the next block's 64-channel tile (NEXT)
                if constexpr (NEXT) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // One wave per SIMD works through the chunks, so an LDS round trip it waits
                // for is lost time.  What chunk q + 1 reads from LDS (pointwise weights, scale /
                // shift rows, the next layer's weights: 8 or 12 reads) is requested at the top
                // of chunk q into a second set of registers and waited for at the top of chunk
                // q + 1, when nothing younger is in the queue.  Inline assembly, for the reason
                // given at the K loop.
                pc_u32x4 LA3[2][2][2], LT[2][4], LA1[NEXT ? 2 : 1][4];
                const uint32_t a3b[2] = {lds_w + PC_WBYTES + a3o0, lds_w + PC_WBYTES + a3o1};
                const uint32_t tb = (uint32_t)(uintptr_t)(pc_lptr_t)ssl3 + k4 * 4;
                uint32_t a1b = 0;
                if constexpr (NEXT)
                    a1b = (uint32_t)(uintptr_t)(pc_lptr_t)w1l + (lr_ * 64) + ((kc_ ^ ((lr_ >> 1) & 3)) << 4);
                auto load_chunk = [&](int q, pc_u32x4(&A3)[2][2], pc_u32x4(&T)[4], pc_u32x4(&A1)[4]) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            asm volatile("ds_read_b128 %0, %1 offset:%2"
                                         : "=v"(A3[nt][kk]) : "v"(a3b[kk]), "n"((q * 32 + nt * 16) * 128));
                    // scale, shift of channel tile 2 q, then of tile 2 q + 1
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(T[0]) : "v"(tb), "n"(q * 128));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(T[1]) : "v"(tb), "n"(PC_N3 * 4 + q * 128));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(T[2]) : "v"(tb), "n"(q * 128 + 64));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(T[3]) : "v"(tb), "n"(PC_N3 * 4 + q * 128 + 64));
                    if constexpr (NEXT) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            asm volatile("ds_read_b128 %0, %1 offset:%2"
                                         : "=v"(A1[nt]) : "v"(a1b), "n"(q * 4096 + nt * 1024));
                    }
                };
                load_chunk(0, LA3[0], LT[0], LA1[0]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int cur = q & 1, c1s = NEXT ? cur : 0;
                    if constexpr (NEXT)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(LA3[cur][0][0]), "+v"(LA3[cur][0][1]), "+v"(LA3[cur][1][0]),
                                       "+v"(LA3[cur][1][1]), "+v"(LT[cur][0]), "+v"(LT[cur][1]),
                                       "+v"(LT[cur][2]), "+v"(LT[cur][3]), "+v"(LA1[cur][0]),
                                       "+v"(LA1[cur][1]), "+v"(LA1[cur][2]), "+v"(LA1[cur][3]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(LA3[cur][0][0]), "+v"(LA3[cur][0][1]), "+v"(LA3[cur][1][0]),
                                       "+v"(LA3[cur][1][1]), "+v"(LT[cur][0]), "+v"(LT[cur][1]),
                                       "+v"(LT[cur][2]), "+v"(LT[cur][3]));
                    if constexpr (NEXT) {
                        if (q == QMID) {
                            // the other group has finished reading the shared patch buffer
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            issue_patch(blockIdx.x + (f + 1) * gridDim.x);
                        }
                    }
                    if (q + 1 < NQ)
                        load_chunk(q + 1, LA3[(q + 1) & 1], LT[(q + 1) & 1], LA1[NEXT ? (q + 1) & 1 : 0]);
                    // the chunk's weights: rows 32 q .. + 31 of the pointwise filter
                    pc_u32x4(&A3)[2][2] = LA3[cur];
                    f32x4 c3[2][MT];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) c3[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                c3[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(pc_bf16x8, A3[nt][kk]),
                                    __builtin_bit_cast(pc_bf16x8, Bc[mt][kk]), c3[nt][mt], 0, 0, 0);
                    const f32x4 s0 = __builtin_bit_cast(f32x4, LT[cur][0]);
                    const f32x4 h0 = __builtin_bit_cast(f32x4, LT[cur][1]);
                    const f32x4 s1 = __builtin_bit_cast(f32x4, LT[cur][2]);
                    const f32x4 h1 = __builtin_bit_cast(f32x4, LT[cur][3]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        // The residual arrives as 8 contiguous channels per lane; the same row
                        // swap that builds that layout for the store takes it apart again.
                        // (Pinned here by a volatile statement: the optimiser would otherwise run
                        // the swaps right behind the load, chunks early, and wait for it there.)
                        uint32_t ru[4];                 // residual: channel tile 2q (2 words), 2q+1 (2)
                        if constexpr (PROJ) {
                            pc_u32x4 r = R[q][mt];
                            asm volatile("" : "+v"(r));
                            ru[0] = r.x;
                            ru[1] = r.y;
                            ru[2] = r.z;
                            ru[3] = r.w;
                        } else {
                            pc_u32x4 r = R[q][mt];
                            asm volatile("" : "+v"(r));
                            const pc_u32x2 ra = __builtin_amdgcn_permlane16_swap(r.x, r.z, false, false);
                            const pc_u32x2 rb = __builtin_amdgcn_permlane16_swap(r.y, r.w, false, false);
                            ru[0] = ra.x;
                            ru[1] = rb.x;
                            ru[2] = ra.y;
                            ru[3] = rb.y;
                        }
                        const f32x4 a = c3[0][mt], b = c3[1][mt];
                        float v[8] = {fmaf(a.x, s0.x, h0.x), fmaf(a.y, s0.y, h0.y), fmaf(a.z, s0.z, h0.z),
                                      fmaf(a.w, s0.w, h0.w), fmaf(b.x, s1.x, h1.x), fmaf(b.y, s1.y, h1.y),
                                      fmaf(b.z, s1.z, h1.z), fmaf(b.w, s1.w, h1.w)};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[2 * i] += bf16_bits_to_float(ru[i] & 0xffffu);
                            v[2 * i + 1] += bf16_bits_to_float(ru[i] >> 16);
                        }
                        if (PW != 2) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.0f);
                        }
                        const uint32_t x0 = pack_bf16x2_bits(v[0], v[1]), x1 = pack_bf16x2_bits(v[2], v[3]);
                        const uint32_t y0 = pack_bf16x2_bits(v[4], v[5]), y1 = pack_bf16x2_bits(v[6], v[7]);
                        const pc_u32x2 o0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                        const pc_u32x2 o1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                        const pc_u32x4 o = {o0.x, o1.x, o0.y, o1.y};
                        *reinterpret_cast<pc_u32x4*>(yp[mt] + q * 64) = o;
                        if constexpr (NEXT) {
                            // The chunk just made (32 of the 256 channels, rounded to bf16 as
                            // stored) is K step q of the next block's pointwise layer: the same
                            // two row swaps as above turn it into that layer's activation operand.
                            pc_u32x4 Bn;
                            {
                                const pc_u32x2 t0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                                const pc_u32x2 u0 = __builtin_amdgcn_permlane16_swap(t0.x, t0.y, false, false);
                                const pc_u32x2 t1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                                const pc_u32x2 u1 = __builtin_amdgcn_permlane16_swap(t1.x, t1.y, false, false);
                                Bn = pc_u32x4{u0.x, u1.x, u0.y, u1.y};
                            }
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(pc_bf16x8, LA1[c1s][nt]),
                                    __builtin_bit_cast(pc_bf16x8, Bn), acc1[nt][mt], 0, 0, 0);
                        }
                    }
                }
                if constexpr (NEXT) {
                    // ---- the next block's tile: scale, shift, ReLU, 16-byte stores as the 3x3
                    // layer alone would make them
                    const float* s1p = ssl1 + k4;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int oy = tl.y0 + cw * MT + mt, ox = tl.x0 + lr;
                        const bool ok = oy < p.H && ox < p.W;
                        char* zp = ok ? reinterpret_cast<char*>(p.z) +
                                            ((((long)tl.b * p.H + oy) * p.W + ox) * 128 + chb)
                                      : reinterpret_cast<char*>(g_pc_sink) + lane * 16;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s1p + 32 * j);
                            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s1p + 64 + 32 * j);
                            const f32x4 sc1 = *reinterpret_cast<const f32x4*>(s1p + 32 * j + 16);
                            const f32x4 sh1 = *reinterpret_cast<const f32x4*>(s1p + 64 + 32 * j + 16);
                            const f32x4 c0 = acc1[2 * j][mt], c1 = acc1[2 * j + 1][mt];
                            const uint32_t x0 = pc_act_pack(c0.x, c0.y, sc0.x, sc0.y, sh0.x, sh0.y, true, 1);
                            const uint32_t x1 = pc_act_pack(c0.z, c0.w, sc0.z, sc0.w, sh0.z, sh0.w, true, 1);
                            const uint32_t y0 = pc_act_pack(c1.x, c1.y, sc1.x, sc1.y, sh1.x, sh1.y, true, 1);
                            const uint32_t y1 = pc_act_pack(c1.z, c1.w, sc1.z, sc1.w, sh1.z, sh1.w, true, 1);
                            const pc_u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                            const pc_u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                            *reinterpret_cast<pc_u32x4*>(zp + j * 64) = pc_u32x4{s0.x, s1.x, s0.y, s1.y};
                        }
                    }
                    // the patch was requested behind chunk QMID - 1: every store after it is counted
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NQ - QMID) * MT + 2 * MT) : "memory");
                } else {
                    // The patch requests are older than this tile's NQ * MT stores, every one of
                    // which was issued (to the image or to the sink), and vector memory operations
                    // retire in order: "all but NQ * MT" means the patch has landed.
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ * MT) : "memory");
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {             // channel-tile pairs (0, 1), (2, 3)
                    const int n0 = (2 * j) * 16 + kc * 4, n1 = n0 + 16;
                    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ssl + n0);
                    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(ssl + 64 + n0);
                    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(ssl + n1);
                    const f32x4 sh1 = *reinterpret_cast<const f32x4*>(ssl + 64 + n1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 c0 = acc[2 * j][mt], c1 = acc[2 * j + 1][mt];
                        const uint32_t x0 = pc_act_pack(c0.x, c0.y, sc0.x, sc0.y, sh0.x, sh0.y, AFFINE, RELU);
                        const uint32_t x1 = pc_act_pack(c0.z, c0.w, sc0.z, sc0.w, sh0.z, sh0.w, AFFINE, RELU);
                        const uint32_t y0 = pc_act_pack(c1.x, c1.y, sc1.x, sc1.y, sh1.x, sh1.y, AFFINE, RELU);
                        const uint32_t y1 = pc_act_pack(c1.z, c1.w, sc1.z, sc1.w, sh1.z, sh1.w, AFFINE, RELU);
                        // odd rows of x <-> even rows of y
                        const pc_u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                        const pc_u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                        const uint4 o = make_uint4(s0.x, s1.x, s0.y, s1.y);
                        const int oy = tl.y0 + cw * MT + mt, ox = tl.x0 + lr;
                        if (oy < p.H && ox < p.W)
                            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.y) +
                                                      ((((long)tl.b * p.H + oy) * p.W + ox) * 128 +
                                                       j * 64 + chb)) = o;
                    }
                }
                // The patch requests are older than the 2 MT stores, and vector memory
                // operations retire in order: a tile that lies wholly inside the image issued
                // all of them, so "all but 2 MT" means the patch has landed; an edge tile may
                // have skipped some.
                if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

}  // namespace pe

using namespace pe;

template <int R, bool A, int P>
static hipError_t pc_launch_variant(const PatchConvParams& p, unsigned nwg, hipStream_t stream) {
    const hipError_t e = pe::ensure_dynamic_lds<&conv3x3_c64_kernel<R, A, P>>(PcGeo<P>::LDS);
    if (e == hipSuccess)
        pe::launch(conv3x3_c64_kernel<R, A, P>, nwg, 512, (size_t)PcGeo<P>::LDS, stream, p);
    return e;
}

// Shared launch: pw 0 = the 3x3 layer alone, 1 / 2 = with the pointwise layer (ReLU / none).
static int32_t pc_launch(const char* what, int32_t B, int32_t H, int32_t W, const void* x,
                         const void* w, const float* scale, const float* shift, int32_t relu,
                         const void* w3, const float* scale3, const float* shift3,
                         const void* residual, const void* x0, const void* wd, const float* scaled,
                         const float* shiftd, int pw, void* y, pe_stream_t stream,
                         const void* w1 = nullptr, const float* scale1 = nullptr,
                         const float* shift1 = nullptr, void* z = nullptr) {
    PE_REQUIRE(x && w && y, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0, PE_E_INVALID, "%s: sizes must be positive", what);
    PE_REQUIRE(H <= 8192 && W <= 8192, PE_E_UNSUPPORTED, "%s: images above 8192 x 8192", what);
    PE_REQUIRE(relu >= 0 && relu <= 2, PE_E_INVALID, "%s: relu must be 0, 1 or 2", what);
    PE_REQUIRE(aligned(x, 16) && aligned(w, 16) && aligned(y, 16) && aligned(scale, 16) &&
                   aligned(shift, 16) && aligned(w3, 16) && aligned(scale3, 16) &&
                   aligned(shift3, 16) && aligned(residual, 16) && aligned(x0, 16) &&
                   aligned(wd, 16) && aligned(scaled, 16) && aligned(shiftd, 16) &&
                   aligned(w1, 16) && aligned(scale1, 16) && aligned(shift1, 16) && aligned(z, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    PatchConvParams p;
    p.x = static_cast<const uint16_t*>(x);
    p.w = static_cast<const uint16_t*>(w);
    p.scale = scale;
    p.shift = shift;
    p.y = static_cast<uint16_t*>(y);
    p.w3 = static_cast<const uint16_t*>(w3);
    p.scale3 = scale3;
    p.shift3 = shift3;
    p.res = static_cast<const uint16_t*>(residual);
    p.x0 = static_cast<const uint16_t*>(x0);
    p.wd = static_cast<const uint16_t*>(wd);
    p.scaled = scaled;
    p.shiftd = shiftd;
    p.w1 = static_cast<const uint16_t*>(w1);
    p.scale1 = scale1;
    p.shift1 = shift1;
    p.z = static_cast<uint16_t*>(z);
    p.B = B;
    p.H = H;
    p.W = W;
    const int ty = pw >= 4 ? PcGeo<4>::TY : pw ? PcGeo<1>::TY : PcGeo<0>::TY;
    static_assert(PcGeo<3>::TY == PcGeo<1>::TY, "tile heights");
    p.tiles_x = (W + PC_TX - 1) / PC_TX;
    p.tiles_y = (H + ty - 1) / ty;
    const long ntiles = (long)B * p.tiles_x * p.tiles_y;
    PE_REQUIRE(ntiles < 0x3fffffffL, PE_E_UNSUPPORTED, "%s: too many tiles", what);
    p.ntiles = (int)ntiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned nwg = (unsigned)(ntiles < cus ? ntiles : cus);   // one persistent workgroup per CU
    const bool affine = scale != nullptr || shift != nullptr;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipSuccess;
    bool launched = true;
    if (pw == 0 && !affine && relu == 0) e = pc_launch_variant<0, false, 0>(p, nwg, st);
    else if (pw == 0 && !affine && relu == 1) e = pc_launch_variant<1, false, 0>(p, nwg, st);
    else if (pw == 0 && !affine && relu == 2) e = pc_launch_variant<2, false, 0>(p, nwg, st);
    else if (pw == 0 && affine && relu == 0) e = pc_launch_variant<0, true, 0>(p, nwg, st);
    else if (pw == 0 && affine && relu == 1) e = pc_launch_variant<1, true, 0>(p, nwg, st);
    else if (pw == 0 && affine && relu == 2) e = pc_launch_variant<2, true, 0>(p, nwg, st);
    else if (pw == 1 && affine && relu == 1) e = pc_launch_variant<1, true, 1>(p, nwg, st);
    else if (pw == 2 && affine && relu == 1) e = pc_launch_variant<1, true, 2>(p, nwg, st);
    else if (pw == 3 && affine && relu == 1) e = pc_launch_variant<1, true, 3>(p, nwg, st);
    else if (pw == 4 && affine && relu == 1) e = pc_launch_variant<1, true, 4>(p, nwg, st);
    else launched = false;
    PE_REQUIRE(launched, PE_E_UNSUPPORTED,
               "%s: the fused pointwise layer needs scale / shift and ReLU on the 3x3 layer", what);
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in: %s", what, hipGetErrorString(e));
    return check_launch(what);
}

extern "C" int32_t pe_conv3x3_c64_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                           const void* w, const float* scale, const float* shift,
                                           int32_t relu, void* y, pe_stream_t stream) {
    return pc_launch("pe_conv3x3_c64_fwd_bf16", B, H, W, x, w, scale, shift, relu, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, y, stream);
}

extern "C" int32_t pe_conv3x3_c64_pw_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                              const void* w, const float* scale, const float* shift,
                                              int32_t relu, const void* w3, int32_t Cout3,
                                              const float* scale3, const float* shift3,
                                              const void* residual, int32_t relu3, void* y,
                                              pe_stream_t stream) {
    const char* what = "pe_conv3x3_c64_pw_fwd_bf16";
    PE_REQUIRE(w3 && residual, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(Cout3 == PC_N3, PE_E_UNSUPPORTED, "%s: the pointwise layer must have %d outputs", what,
               PC_N3);
    PE_REQUIRE(relu3 == 0 || relu3 == 1, PE_E_INVALID, "%s: relu3 must be 0 or 1", what);
    return pc_launch(what, B, H, W, x, w, scale, shift, relu, w3, scale3, shift3, residual, nullptr,
                     nullptr, nullptr, nullptr, relu3 ? 1 : 2, y, stream);
}

extern "C" int32_t pe_conv3x3_c64_pw_proj_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                                   const void* w, const float* scale,
                                                   const float* shift, int32_t relu, const void* w3,
                                                   int32_t Cout3, const float* scale3,
                                                   const float* shift3, const void* x0,
                                                   const void* wd, const float* scaled,
                                                   const float* shiftd, void* y, pe_stream_t stream) {
    const char* what = "pe_conv3x3_c64_pw_proj_fwd_bf16";
    PE_REQUIRE(w3 && x0 && wd, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(Cout3 == PC_N3, PE_E_UNSUPPORTED, "%s: the pointwise layers must have %d outputs", what,
               PC_N3);
    return pc_launch(what, B, H, W, x, w, scale, shift, relu, w3, scale3, shift3, nullptr, x0, wd,
                     scaled, shiftd, 3, y, stream);
}

extern "C" int32_t pe_conv3x3_c64_pw_next_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                                   const void* w, const float* scale,
                                                   const float* shift, int32_t relu, const void* w3,
                                                   int32_t Cout3, const float* scale3,
                                                   const float* shift3, const void* residual,
                                                   void* y, const void* w_next, int32_t Cout_next,
                                                   const float* scale_next, const float* shift_next,
                                                   void* y_next, pe_stream_t stream) {
    const char* what = "pe_conv3x3_c64_pw_next_fwd_bf16";
    PE_REQUIRE(w3 && residual && w_next && y_next, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(Cout3 == PC_N3, PE_E_UNSUPPORTED, "%s: the pointwise layer must have %d outputs", what,
               PC_N3);
    PE_REQUIRE(Cout_next == 64, PE_E_UNSUPPORTED, "%s: the next pointwise layer must have 64 outputs",
               what);
    return pc_launch(what, B, H, W, x, w, scale, shift, relu, w3, scale3, shift3, residual, nullptr,
                     nullptr, nullptr, nullptr, 4, y, stream, w_next, scale_next, shift_next, y_next);
}
