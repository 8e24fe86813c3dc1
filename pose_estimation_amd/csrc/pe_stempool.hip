// 7x7 stride-2 stem + folded batch norm + ReLU + 3x3 stride-2 max pool in ONE launch (inference).
//
// As separate launches the stem writes its 64-channel map at half resolution (402 MB at batch
// 256) and the pool reads it back; both are bound by that traffic and by launch-to-launch
// latency chains (the stem's K loop is four steps long).  Here a workgroup works on 4 x 16
// tiles of POOLED pixels: per tile it computes the 9 x 33 stem outputs that tile's windows cover
// (the windows of neighbouring tiles overlap by one row / column, so 16 % of the stem is
// computed twice), keeps them in LDS and writes only the pooled map.  Workgroups are
// persistent (two per CU, the filter loaded once each, the next tile's input patch fetched
// into registers while the current one is multiplied).
// Measured at batch 256 (256x192 images): 147 us against 188 + 122 us for the two launches;
// by phase (each switched off in turn): multiply 79 us, normalise + round 45, pool 21, patch
// staging 28 -- the phases of the two workgroups of a CU do not hide behind one another
// (a 16x16x32 MFMA holds the SIMD's vector issue for half of its cycles), and delaying one of
// the two by up to a tile changed nothing.
//
// The stem is a 4x4 stride-1 conv over the 2x2 space-to-depth image (16 channels per pixel):
// output (oy, ox) reads pixels (oy - 2 + r, ox - 2 + q), r, q = 0..3, and its K axis is
// [r][q][16 channels] = 256 elements, the layout of the packed weights [64][4][64].  Instead of
// an im2col tile per K step (each input pixel staged 4 x for its 4 column offsets), the
// workgroup stages the 12 x 36-pixel input patch ONCE and every lane builds its MFMA B
// fragment by address: fragment chunk kc of K step (r, kk) is 8 channels (half a pixel) of
// patch pixel (ty + r, tx + 2 kk + (lane >> 5)).  Consecutive lanes read consecutive pixels, 32
// bytes apart, and the two halves of a pixel go to the two 16-lane halves of a ds_read_b128
// group, which makes those reads bank-conflict free without a swizzle.  The whole 64 x 256
// filter (32 KB) sits in LDS in the swizzled 128-byte-row layout of the implicit-GEMM engine.
//
// MFMA operand roles, K order (r, then kk), fmaf / ReLU / rounding are those of
// pe_stem7x7s2_fwd_bf16, and a maximum is exact, so the result equals
// pe_maxpool3x3s2_fwd_bf16(pe_stem7x7s2_fwd_bf16(x)) bit for bit (ReLU makes every value >= 0,
// so cells outside the image count as 0).
#include "pe_common.h"

namespace pe {

typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef short sp_s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* sp_gptr_t;
typedef __attribute__((address_space(3))) void* sp_lptr_t;

constexpr int SP_PH = 4, SP_PW = 16;              // pooled tile
constexpr int SP_CH = 2 * SP_PH + 1;              // 9 stem rows
constexpr int SP_CW = 2 * SP_PW + 1;              // 33 stem columns
constexpr int SP_M = SP_CH * SP_CW;               // 297 stem pixels per tile
constexpr int SP_XH = SP_CH + 3, SP_XW = SP_CW + 3;   // 12 x 36 input patch
constexpr int SP_TM = 5;                          // 16-pixel tiles per wave (4 waves x 5 x 16 = 320)
constexpr int SP_WBYTES = 4 * 64 * 128;           // filter: 4 planes of 64 rows x 128 bytes
constexpr int SP_XBYTES = SP_XH * SP_XW * 32;     // patch: 32 bytes per pixel
constexpr int SP_OBYTES = ((SP_M + 7) / 8 * 8) * 128;   // stem tile: 128 bytes per pixel
// filter resident for the whole launch; the patch and the stem tile take turns in one region
constexpr int SP_RBYTES = SP_XBYTES > SP_OBYTES ? SP_XBYTES : SP_OBYTES;
constexpr int SP_LDS = SP_WBYTES + SP_RBYTES + 2 * 64 * 4;   // + scale and shift

struct StemPoolParams {
    const uint16_t* x;     // [B][H2][W2][16] bf16
    const uint16_t* w;     // [64][4][64] bf16
    const float* scale;    // [64] or null
    const float* shift;    // [64] or null
    uint16_t* y;           // [B][Hp][Wp][64] bf16
    int B, H2, W2, Hp, Wp;
    int tiles_x, tiles_y;
    int ntiles;
};

// The four 16-byte pieces of the input patch this thread stages for tile `t` (zeros outside the
// image or past the last tile): issued a whole tile ahead, so the loads fly under the multiply.
__device__ __forceinline__ void sp_fetch_patch(const StemPoolParams& p, int t, int tid, uint4 (&v)[4]) {
    const int tx0 = t % p.tiles_x;
    const int r = t / p.tiles_x;
    const int ty0 = r % p.tiles_y;
    const int b = r / p.tiles_y;
    const int sy0 = 2 * ty0 * SP_PH - 3, sx0 = 2 * tx0 * SP_PW - 3;   // patch origin
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + i * 256;
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (t < p.ntiles && id < SP_XH * SP_XW * 2) {
            const int pix = id >> 1, h = id & 1;
            const int py = pix / SP_XW, px = pix - py * SP_XW;
            const int sy = sy0 + py, sx = sx0 + px;
            if ((unsigned)sy < (unsigned)p.H2 && (unsigned)sx < (unsigned)p.W2)
                v[i] = *reinterpret_cast<const uint4*>(
                    p.x + (((long)b * p.H2 + sy) * p.W2 + sx) * 16 + h * 8);
        }
    }
}

// Persistent: two workgroups per CU, each walking tiles blockIdx.x, + gridDim.x, ...  The filter
// is loaded once per workgroup; while one workgroup of a CU multiplies, the other normalises,
// pools and stores.
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                      // filter planes
    char* xl = smem + SP_WBYTES;          // input patch ...
    char* ol = smem + SP_WBYTES;          // ... and, after the multiply, the stem tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15;

    // ---- filter: 256 rows of 128 bytes, 8 rows per wave instruction, lane l -> row l>>3,
    // slot l&7 receives K chunk (l&7) ^ (row&7)
    {
        const int cc = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int R = i * 32 + (tid >> 3);            // plane * 64 + output channel
            const uint16_t* src = p.w + (long)(R & 63) * 256 + (R >> 6) * 64 + cc * 8;
            __builtin_amdgcn_global_load_lds((sp_gptr_t)src,
                                             (sp_lptr_t)(wl + (i * 32 + 8 * wid) * 128), 16, 0, 0);
        }
    }
    // per-lane constants of the multiply
    int xoff[SP_TM];
#pragma unroll
    for (int bt = 0; bt < SP_TM; ++bt) {
        int m = (wid * SP_TM + bt) * 16 + lr;
        m = m < SP_M ? m : SP_M - 1;                      // padding pixels read a real one
        const int ty = m / SP_CW, tx = m - ty * SP_CW;
        xoff[bt] = (ty * SP_XW + tx) * 32 + (lane >> 5) * 32 + ((lane >> 4) & 1) * 16;
    }
    const int woff = lr * 128;
    const int nq = (lane >> 4) * 4;
    const bool affine = p.scale != nullptr || p.shift != nullptr;
    // scale and shift in LDS (read per tile by every lane); identity values when absent
    float* ssl = reinterpret_cast<float*>(smem + SP_WBYTES + SP_RBYTES);
    if (tid < 64) {
        ssl[tid] = p.scale ? p.scale[tid] : 1.0f;
        ssl[64 + tid] = p.shift ? p.shift[tid] : 0.0f;
    }
    // per-lane constants of the epilogue: LDS row and tile position of each stem pixel
    int orow[SP_TM], opos[SP_TM];
#pragma unroll
    for (int bt = 0; bt < SP_TM; ++bt) {
        const int m = (wid * SP_TM + bt) * 16 + lr;
        const int ty = m / SP_CW, tx = m - ty * SP_CW;
        orow[bt] = m < SP_M ? m * 128 + ((m & 7) << 4) : -1;     // row base with the swizzle term
        opos[bt] = (ty << 8) | tx;
    }

    uint4 pre[4];
    sp_fetch_patch(p, blockIdx.x, tid, pre);

    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int tx0 = t % p.tiles_x;
        const int rr = t / p.tiles_x;
        const int ty0 = rr % p.tiles_y;
        const int b = rr / p.tiles_y;
        const int pi0 = ty0 * SP_PH, pj0 = tx0 * SP_PW;       // pooled origin
        const int cy0 = 2 * pi0 - 1, cx0 = 2 * pj0 - 1;       // stem origin (may be -1)

        // ---- this tile's patch (fetched a tile ago) into LDS; then fetch the next one
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256;
            if (id < SP_XH * SP_XW * 2) *reinterpret_cast<uint4*>(xl + id * 16) = pre[i];
        }
        __syncthreads();   // (on the first tile this also drains the filter's direct-to-LDS loads)
        sp_fetch_patch(p, t + gridDim.x, tid, pre);

        // ---- multiply: wave `wid` owns stem pixels (wid * 5 + bt) * 16 + lr, all 64 channels
        f32x4 acc[4][SP_TM];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bt = 0; bt < SP_TM; ++bt) acc[a][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                sp_bf16x8 af[4], bf[SP_TM];
                const int slot = ((kk * 4 + (lane >> 4)) ^ (lr & 7)) * 16;
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    af[a] = *reinterpret_cast<const sp_bf16x8*>(wl + r * 8192 + a * 2048 + woff + slot);
#pragma unroll
                for (int bt = 0; bt < SP_TM; ++bt)
                    bf[bt] = *reinterpret_cast<const sp_bf16x8*>(xl + xoff[bt] +
                                                                 (r * SP_XW + kk * 2) * 32);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int bt = 0; bt < SP_TM; ++bt)
                        acc[a][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[bt],
                                                                             acc[a][bt], 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave is done with the patch (raw: the prefetch stays in flight)
        asm volatile("" ::: "memory");

        // ---- normalise, ReLU, round; stem pixels outside the image become 0; 128-byte rows
        // with the 16-byte slot swizzled by pixel & 7
        bool inimg[SP_TM];
#pragma unroll
        for (int bt = 0; bt < SP_TM; ++bt)
            inimg[bt] = (unsigned)(cy0 + (opos[bt] >> 8)) < (unsigned)p.H2 &&
                        (unsigned)(cx0 + (opos[bt] & 255)) < (unsigned)p.W2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = a * 16 + nq;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + n);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + 64 + n);
            // (n >> 3) ^ (m & 7) in bytes: the row base carries (m & 7) << 4
            const int cslot = ((n >> 3) << 4), cbyte = (n & 7) << 1;
#pragma unroll
            for (int bt = 0; bt < SP_TM; ++bt) {
                if (orow[bt] < 0) continue;
                float v0 = acc[a][bt].x, v1 = acc[a][bt].y, v2 = acc[a][bt].z, v3 = acc[a][bt].w;
                if (affine) {
                    v0 = fmaf(v0, sc.x, sh.x);
                    v1 = fmaf(v1, sc.y, sh.y);
                    v2 = fmaf(v2, sc.z, sh.z);
                    v3 = fmaf(v3, sc.w, sh.w);
                }
                const bool in = inimg[bt];
                v0 = in ? fmaxf(v0, 0.0f) : 0.0f;
                v1 = in ? fmaxf(v1, 0.0f) : 0.0f;
                v2 = in ? fmaxf(v2, 0.0f) : 0.0f;
                v3 = in ? fmaxf(v3, 0.0f) : 0.0f;
                uint2 o;
                o.x = pack_bf16x2_bits(v0, v1);
                o.y = pack_bf16x2_bits(v2, v3);
                // row base = m * 128 + ((m & 7) << 4): XOR-ing the slot bits of the channel in
                const int rowb = orow[bt] & ~0x70, sw = orow[bt] & 0x70;
                *reinterpret_cast<uint2*>(ol + rowb + ((cslot ^ sw) | cbyte)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        // ---- pool: 4 x 16 pooled pixels x 8 chunks of 8 channels = 512 items, two per thread
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + k * 256;
            const int c8 = item & 7, pj = (item >> 3) & (SP_PW - 1), pi = item >> 7;
            const int oy = pi0 + pi, ox = pj0 + pj;
            if (oy >= p.Hp || ox >= p.Wp) continue;
            // every value is >= 0 after the ReLU, so bf16 bit patterns order like signed 16-bit
            // integers (a -0.0, the one negative pattern that could occur, loses to +0)
            sp_s16x2 best[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) best[i] = sp_s16x2{0, 0};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int m = (2 * pi + ky) * SP_CW + 2 * pj + kx;
                    const uint4 v = *reinterpret_cast<const uint4*>(ol + m * 128 + ((c8 ^ (m & 7)) << 4));
                    best[0] = __builtin_elementwise_max(best[0], __builtin_bit_cast(sp_s16x2, v.x));
                    best[1] = __builtin_elementwise_max(best[1], __builtin_bit_cast(sp_s16x2, v.y));
                    best[2] = __builtin_elementwise_max(best[2], __builtin_bit_cast(sp_s16x2, v.z));
                    best[3] = __builtin_elementwise_max(best[3], __builtin_bit_cast(sp_s16x2, v.w));
                }
            uint4 o;
            o.x = __builtin_bit_cast(uint32_t, best[0]);
            o.y = __builtin_bit_cast(uint32_t, best[1]);
            o.z = __builtin_bit_cast(uint32_t, best[2]);
            o.w = __builtin_bit_cast(uint32_t, best[3]);
            *reinterpret_cast<uint4*>(p.y + (((long)b * p.Hp + oy) * p.Wp + ox) * 64 + c8 * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the pool's reads are back before the next patch lands there
        asm volatile("" ::: "memory");
    }
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_stem7x7s2_pool_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                              const void* x_s2d, const void* w, const float* scale,
                                              const float* shift, void* y_pooled,
                                              pe_stream_t stream) {
    const char* what = "pe_stem7x7s2_pool_fwd_bf16";
    PE_REQUIRE(x_s2d && w && y_pooled, PE_E_INVALID, "%s: NULL pointer", what);
    PE_REQUIRE(B > 0 && H2 > 0 && W2 > 0 && H2 % 2 == 0 && W2 % 2 == 0, PE_E_INVALID,
               "%s: need a positive batch and even, positive H2 and W2", what);
    PE_REQUIRE(Cout == 64, PE_E_UNSUPPORTED, "%s: written for 64 output channels (got %d)", what,
               Cout);
    PE_REQUIRE(aligned(x_s2d, 16) && aligned(w, 16) && aligned(y_pooled, 16) && aligned(scale, 16) &&
                   aligned(shift, 16),
               PE_E_INVALID, "%s: tensors must be 16-byte aligned", what);
    StemPoolParams p;
    p.x = static_cast<const uint16_t*>(x_s2d);
    p.w = static_cast<const uint16_t*>(w);
    p.scale = scale;
    p.shift = shift;
    p.y = static_cast<uint16_t*>(y_pooled);
    p.B = B;
    p.H2 = H2;
    p.W2 = W2;
    p.Hp = H2 / 2;
    p.Wp = W2 / 2;
    p.tiles_x = (p.Wp + SP_PW - 1) / SP_PW;
    p.tiles_y = (p.Hp + SP_PH - 1) / SP_PH;
    const long ntiles = (long)B * p.tiles_x * p.tiles_y;
    PE_REQUIRE(ntiles < 0x3fffffffL, PE_E_UNSUPPORTED, "%s: too many tiles", what);
    p.ntiles = (int)ntiles;
    // two persistent workgroups per CU (71 KB of LDS each)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long nwg = ntiles < 2L * cus ? ntiles : 2L * cus;
    const hipError_t e = pe::ensure_dynamic_lds<&stem_pool_kernel>(SP_LDS);
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "%s: LDS opt-in: %s", what, hipGetErrorString(e));
    pe::launch(stem_pool_kernel, (unsigned)nwg, 256, (size_t)SP_LDS, as_stream(stream), p);
    return check_launch(what);
}
