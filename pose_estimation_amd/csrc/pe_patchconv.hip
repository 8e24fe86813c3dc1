// 3x3 stride-1 pad-1 convolution with 64 input and 64 output channels (+ scale, shift, ReLU),
// NHWC bf16: the 3x3 layers of the first ResNet stage, for inference.
//
// Through the implicit-GEMM engine such a layer re-stages every activation tile once per tap (9
// times) for only 64 output channels and is bound by L2 -> LDS traffic, not by the matrix cores
// (630 TFLOP/s at batch 256).  Here one persistent workgroup per CU keeps the WHOLE filter in
// LDS (9 taps x 64 rows x 128 bytes = 72 KB, loaded once) and stages, per 16 x 16-pixel output
// tile, the 18 x 18-pixel input patch ONCE (128 bytes per pixel, direct-to-LDS loads, the next
// tile's patch in flight in a second buffer while the current one is multiplied); a tap is an
// address offset into the patch.  Staged bytes per output tile fall from 360 KB to 41 KB.
//
// Layout: a patch pixel is one 128-byte LDS row of 8 x 16-byte slots, slot s holding channel
// chunk s ^ (pixel index & 7); the 16 lanes of an MFMA B-fragment read 16 consecutive pixels of
// a patch row, hence 16 distinct (pixel & 7, chunk) combinations per 8: conflict-free, as in the
// engine's tiles.  The filter planes use the engine's weight-tile layout.
//
// MFMA operand roles, K order (tap by tap, two 32-channel halves each) and the epilogue's
// operations are those of the engine, so the result equals pe_conv2d_fwd_bf16 bit for bit.
#include "pe_common.h"

namespace pe {

typedef __bf16 pc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* pc_gptr_t;
typedef __attribute__((address_space(3))) void* pc_lptr_t;

constexpr int PC_T = 16;                          // output tile edge
constexpr int PC_P = PC_T + 2;                    // patch edge
constexpr int PC_NPIX = PC_P * PC_P;              // 324 patch pixels
constexpr int PC_NINSTR = (PC_NPIX + 7) / 8;      // 41 wave instructions of 8 pixels
constexpr int PC_WBYTES = 9 * 64 * 128;           // 73 728
constexpr int PC_XBYTES = PC_NINSTR * 1024;       // 41 984 per patch buffer
constexpr int PC_LDS = PC_WBYTES + 2 * PC_XBYTES + 2 * 64 * 4;

__device__ __attribute__((aligned(16))) uint32_t g_pc_zero16[4] = {0u, 0u, 0u, 0u};

struct PatchConvParams {
    const uint16_t* x;     // [B][H][W][64]
    const uint16_t* w;     // [64][3][3][64]
    const float* scale;
    const float* shift;
    uint16_t* y;           // [B][H][W][64]
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
    int relu;
};

struct PcTile {
    int b, y0, x0;
};

__device__ __forceinline__ PcTile pc_tile(const PatchConvParams& p, int t) {
    const int tx0 = t % p.tiles_x;
    const int r = t / p.tiles_x;
    return PcTile{r / p.tiles_y, (r % p.tiles_y) * PC_T, tx0 * PC_T};
}

// Helper wave `hw` (0..3) requests its share of the 18 x 18 patch of tile `t` into `buf`
// (direct-to-LDS, asynchronous): the 1 KB blocks j = hw, hw + 4, ... of the buffer.
__device__ __forceinline__ void pc_issue_patch(const PatchConvParams& p, int t, char* buf, int hw,
                                               int lane) {
    if (t >= p.ntiles) return;                    // uniform
    const PcTile tl = pc_tile(p, t);
    const int sy0 = tl.y0 - 1, sx0 = tl.x0 - 1;
    const uint16_t* img = p.x + (long)tl.b * p.H * p.W * 64;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_pc_zero16);
#pragma unroll
    for (int i = 0; i < (PC_NINSTR + 3) / 4; ++i) {
        const int j = i * 4 + hw;                 // this wave's block: 8 patch pixels
        if (j >= PC_NINSTR) break;                // uniform
        const int pp = j * 8 + (lane >> 3);
        const int py = pp / PC_P, px = pp - py * PC_P;
        const int sy = sy0 + py, sx = sx0 + px;
        const int cc = (lane & 7) ^ (pp & 7);
        const bool ok = pp < PC_NPIX && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
        const uint16_t* src = ok ? img + ((long)sy * p.W + sx) * 64 + cc * 8 : zero;
        __builtin_amdgcn_global_load_lds((pc_gptr_t)src, (pc_lptr_t)(buf + j * 1024), 16, 0, 0);
    }
}

// Helper wave `hw` writes its share of the finished output tile in `buf` to memory: the rows
// of blocks j = hw, hw + 4, ..., hw + 28, i.e. exactly the LDS bytes its own next patch requests
// overwrite, so the wave's own lgkmcnt orders the two and no barrier between helpers is needed.
__device__ __forceinline__ void pc_store_tile(const PatchConvParams& p, const PcTile& tl,
                                              const char* buf, int hw, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = (i * 4 + hw) * 8 + (lane >> 3), ch = lane & 7;
        const int oy = tl.y0 + (m >> 4), ox = tl.x0 + (m & 15);
        if (oy < p.H && ox < p.W) {
            const uint4 v = *reinterpret_cast<const uint4*>(buf + m * 128 + ((ch ^ (m & 7)) << 4));
            *reinterpret_cast<uint4*>(p.y + (((long)tl.b * p.H + oy) * p.W + ox) * 64 + ch * 8) = v;
        }
    }
}

// 512 threads: waves 0-3 multiply and convert, waves 4-7 ("helpers", one per SIMD beside a
// multiplying wave) move data: they write the previous tile's output rows to memory and request
// the next tile's patch while the current tile is multiplied.  Alone on its SIMD a multiplying
// wave would do all of that itself, in sequence (measured: multiply 45 us, convert 16, store 13,
// requests and waits 27 of a 100 us launch); the helpers take the last two off its path.
__global__ __launch_bounds__(512, 1) void conv3x3_c64_kernel(const PatchConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    char* const xb0 = smem + PC_WBYTES;                   // two patch buffers
    float* ssl = reinterpret_cast<float*>(smem + PC_WBYTES + 2 * PC_XBYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const bool helper = wid >= 4;
    const int cw = wid & 3;                               // index among the multiplying / helper waves
    const int lr = lane & 15, kc = lane >> 4;

    // ---- filter: 9 x 64 rows of 128 bytes; lane l of a wave instruction -> row l>>3, slot l&7
    {
        const int cc = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int R = i * 64 + (tid >> 3);            // tap * 64 + output channel
            const uint16_t* src = p.w + (long)(R & 63) * 576 + (R >> 6) * 64 + cc * 8;
            __builtin_amdgcn_global_load_lds((pc_gptr_t)src,
                                             (pc_lptr_t)(wl + (i * 64 + 8 * wid) * 128), 16, 0, 0);
        }
    }
    if (tid < 64) {
        ssl[tid] = p.scale ? p.scale[tid] : 1.0f;
        ssl[64 + tid] = p.shift ? p.shift[tid] : 0.0f;
    }
    const bool affine = p.scale != nullptr || p.shift != nullptr;
    if (helper) pc_issue_patch(p, blockIdx.x, xb0, cw, lane);

    const int aoff = lr * 128 + ((kc ^ (lr & 7)) << 4);
    const int nq = kc * 4;

    // the filter and the first patch have landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur = 0;
    PcTile prev{0, 0, 0};
    bool have_prev = false;
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x, cur ^= 1) {
        char* xl = xb0 + cur * PC_XBYTES;
        char* xo = xb0 + (cur ^ 1) * PC_XBYTES;
        f32x4 acc[4][4];
        if (helper) {
            // the other buffer holds the previous tile's output: out to memory, then the next
            // tile's patch over it
            if (have_prev) pc_store_tile(p, prev, xo, cw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pc_issue_patch(p, t + gridDim.x, xo, cw, lane);
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // 18 K steps (tap, 32-channel half).  The fragments of step s + 1 are requested
            // before step s is multiplied, into the other of two register sets, so that with
            // one multiplying wave per SIMD an LDS round trip hides behind 16 MFMAs instead
            // of stalling them.
            pc_bf16x8 af[2][4], bf[2][4];
            auto load_step = [&](int s, pc_bf16x8(&fa)[4], pc_bf16x8(&fb)[4]) {
                const int tap = s >> 1, kk = s & 1;
                // byte offset of (pixel tile mt, tap) in the patch for the first 32-channel
                // half (the second half is that offset ^ 64).  Recomputed per step from a value
                // the optimiser cannot see through: precomputed for all 36 (mt, tap) pairs
                // they would sit in registers for the whole kernel and spill.
                int lt = lr;
                asm volatile("" : "+v"(lt));
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    fa[a] = *reinterpret_cast<const pc_bf16x8*>(wl + tap * 8192 + a * 2048 +
                                                                (aoff ^ (kk << 6)));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int pp = (cw * 4 + mt + tap / 3) * PC_P + lt + tap % 3;
                    const int boff = pp * 128 + ((kc ^ (pp & 7)) << 4);
                    fb[mt] = *reinterpret_cast<const pc_bf16x8*>(xl + (boff ^ (kk << 6)));
                }
            };
            load_step(0, af[0], bf[0]);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18) load_step(s + 1, af[(s + 1) & 1], bf[(s + 1) & 1]);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            af[s & 1][a], bf[s & 1][mt], acc[a][mt], 0, 0, 0);
            }
        }
        // every multiplying wave is done with this tile's patch: its output takes its place
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (helper) {
            // the next patch has landed (and the stores are out) before the tile is handed over
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n = a * 16 + nq;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + n);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + 64 + n);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = (cw * 4 + mt) * 16 + lr;
                    float v0 = acc[a][mt].x, v1 = acc[a][mt].y, v2 = acc[a][mt].z, v3 = acc[a][mt].w;
                    if (affine) {
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
                    o.x = pack_bf16x2_bits(v0, v1);
                    o.y = pack_bf16x2_bits(v2, v3);
                    *reinterpret_cast<uint2*>(xl + m * 128 + ((((n >> 3) ^ (m & 7)) << 4) | ((n & 7) << 1))) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        prev = pc_tile(p, t);
        have_prev = true;
    }
    // the last tile's output sits in the buffer the loop used last
    if (helper && have_prev) pc_store_tile(p, prev, xb0 + (cur ^ 1) * PC_XBYTES, cw, lane);
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_conv3x3_c64_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                           const void* w, const float* scale, const float* shift,
                                           int32_t relu, void* y, pe_stream_t stream) {
    const char* what = "pe_conv3x3_c64_fwd_bf16";
    PE_REQUIRE(x && w && y, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H > 0 && W > 0, PE_E_INVALID, "%s: sizes must be positive", what);
    PE_REQUIRE(relu >= 0 && relu <= 2, PE_E_INVALID, "%s: relu must be 0, 1 or 2", what);
    PE_REQUIRE(aligned(x, 16) && aligned(w, 16) && aligned(y, 16) && aligned(scale, 16) &&
                   aligned(shift, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    PatchConvParams p;
    p.x = static_cast<const uint16_t*>(x);
    p.w = static_cast<const uint16_t*>(w);
    p.scale = scale;
    p.shift = shift;
    p.y = static_cast<uint16_t*>(y);
    p.B = B;
    p.H = H;
    p.W = W;
    p.relu = relu;
    p.tiles_x = (W + PC_T - 1) / PC_T;
    p.tiles_y = (H + PC_T - 1) / PC_T;
    const long ntiles = (long)B * p.tiles_x * p.tiles_y;
    PE_REQUIRE(ntiles < 0x3fffffffL, PE_E_UNSUPPORTED, "%s: too many tiles", what);
    p.ntiles = (int)ntiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long nwg = ntiles < cus ? ntiles : cus;   // one persistent workgroup per CU (154 KB of LDS)
    const hipError_t e = pe::ensure_dynamic_lds<&conv3x3_c64_kernel>(PC_LDS);
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in: %s", what, hipGetErrorString(e));
    pe::launch(conv3x3_c64_kernel, (unsigned)nwg, 512, (size_t)PC_LDS, as_stream(stream), p);
    return check_launch(what);
}
