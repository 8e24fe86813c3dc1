// Backward-weight (weight-gradient) GEMM for every conv on the path:
//
//   dW[n][tap*Cin + c] += sum_m dY[opix(m)][n] * X[ipix(m, tap)][c]
//
// The reduction runs over pixels (m), which is the strided axis of both NHWC operands, so the
// tiles are staged row-major [m][channels] (16-byte coalesced loads) and the MFMA fragments are
// fetched with the transposing LDS read ds_read_b64_tr_b16: a 16-lane group reads a 4-row x
// 16-column block and each lane receives one column, i.e. four consecutive m of one channel.
// Two such reads make the 8-deep K fragment of mfma_f32_16x16x32_bf16; the m order inside a
// fragment is (4g+j, 16+4g+j) for lane group g, identical for both operands, which also puts
// the eight rows a 32-lane half touches on disjoint banks (row pitch = 8 dwords mod 64).
//
// Workgroup tile: 128 (n) x 128 (tap-channel columns), K step 64 pixels, split over pixels
// across gridDim.y; partial tiles are added to dW with fp32 atomics from an LDS-staged,
// 256-byte-contiguous layout (the caller zeroes dW).
#include "pe_common.h"
#include "pe_wgrad.h"

namespace pe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_BK = 64;       // pixels per K step
constexpr int WG_PITCH = 144;   // bf16 elements per LDS row (128 + 16): 288 B = 72 dwords = 8*9

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* lds_tile, int row_base, int col) {
    // lds_tile: [64][WG_PITCH]; returns the 8 K values of column `col` for this lane's group.
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr)(lds_tile + row_base * WG_PITCH + col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr)(lds_tile + (row_base + 16) * WG_PITCH + col));
    s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* lds = reinterpret_cast<uint16_t*>(smem);
    constexpr int TILE = WG_BK * WG_PITCH;   // one operand tile
    constexpr int BUF = 2 * TILE;            // dY tile + X tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wave_c = wid >> 1;   // column half of the output tile
    const int wave_n = wid & 1;    // n half

    const int ktot = p.T * p.Cin;
    const int ntn = (p.N + 127) / 128;
    const int tile_c = blockIdx.x / ntn;
    const int tile_n = blockIdx.x - tile_c * ntn;
    const int n0 = tile_n * 128;
    const int c0 = tile_c * 128;

    const long M = (long)p.B * p.Ho * p.Wo;
    const long ms = (long)blockIdx.y * p.m_per_split;
    long me = ms + p.m_per_split;
    if (me > M) me = M;

    // staging: thread owns 16-byte chunk `ch16` (8 channels) of rows (tid>>4) + 16*i
    const int ch16 = tid & 15;
    const int rr = tid >> 4;
    const int nn = n0 + ch16 * 8;
    const bool n_ok = nn < p.N;
    const int col = c0 + ch16 * 8;
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

    uint4 gy[4], gx[4];
    auto load_tile = [&](long mbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = mbase + rr + 16 * i;
            gy[i] = make_uint4(0, 0, 0, 0);
            gx[i] = make_uint4(0, 0, 0, 0);
            if (m < me) {
                const int b = (int)(m / HoWo);
                const int rem = (int)(m - (long)b * HoWo);
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                if (n_ok) {
                    const long opix = (long)b * OHW + (oy * p.osy + p.ooy) * p.OW + ox * p.osx + p.oox;
                    gy[i] = *reinterpret_cast<const uint4*>(p.dy + opix * p.ldy + nn);
                }
                const int iy = oy * p.sy + tdy;
                const int ix = ox * p.sx + tdx;
                if (c_ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
                    const long off = ((long)b * p.Hi * p.Wi + (long)iy * p.Wi + ix) * p.Cx + xch;
                    gx[i] = *reinterpret_cast<const uint4*>(p.x + off);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        uint16_t* ys = lds + buf * BUF;
        uint16_t* xs = ys + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(ys + (rr + 16 * i) * WG_PITCH + ch16 * 8) = gy[i];
            *reinterpret_cast<uint4*>(xs + (rr + 16 * i) * WG_PITCH + ch16 * 8) = gx[i];
        }
    };

    f32x4 acc[4][4];   // [column tile][n tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing-read lane geometry: lane 4q+p of a 16-lane group addresses row q, columns 4p..
    const int g = lane >> 4;
    const int q = (lane & 15) >> 2;
    const int pp = lane & 3;
    const int row_in = 4 * g + q;     // row inside a 32-pixel K sub-step (second read: +16)
    const int col_in = 4 * pp;        // column offset inside a 16-column tile

    if (ms < me) {
        load_tile(ms);
        store_tile(0);
    }
    __syncthreads();
    int it = 0;
    for (long mb = ms; mb < me; mb += WG_BK, ++it) {
        const int buf = it & 1;
        const bool more = mb + WG_BK < me;
        if (more) load_tile(mb + WG_BK);
        const uint16_t* ys = lds + buf * BUF + wave_n * 64;
        const uint16_t* xs = lds + buf * BUF + TILE + wave_c * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fx[4], fy[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fx[a] = tr_frag(xs, ks * 32 + row_in, a * 16 + col_in);
#pragma unroll
            for (int b = 0; b < 4; ++b) fy[b] = tr_frag(ys, ks * 32 + row_in, b * 16 + col_in);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[a], fy[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: tile -> LDS [n][col] (pitch 132 floats) -> 256-byte-contiguous atomics
    constexpr int CP = 132;
    float* cs = reinterpret_cast<float*>(smem);
    const int lr = lane & 15;
    const int cq = (lane >> 4) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int nl = wave_n * 64 + b * 16 + lr;
            const int cl = wave_c * 64 + a * 16 + cq;
            *reinterpret_cast<f32x4*>(cs + nl * CP + cl) = acc[a][b];
        }
    __syncthreads();
    const int cl = tid & 127;
    const int colg = c0 + cl;
    if (colg < ktot) {
        for (int nl = tid >> 7; nl < 128; nl += 2) {
            const int n = n0 + nl;
            if (n >= p.Nw) break;
            atomicAdd(p.dw + (long)n * ktot + colg, cs[nl * CP + cl]);
        }
    }
}

int32_t wgrad_launch(WgradParams& p, hipStream_t st, const char* what) {
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
    PE_REQUIRE(M < 0x7fffffffL && (long)p.B * p.OH * p.OW < 0x7fffffffL, PE_E_UNSUPPORTED,
               "%s: pixel count does not fit int32", what);
    const long ktot = (long)p.T * p.Cin;
    const long tiles = ((p.N + 127) / 128) * ((ktot + 127) / 128);
    long splits = (768 + tiles - 1) / tiles;
    const long max_splits = (M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long mps = (M + splits - 1) / splits;
    mps = (mps + WG_BK - 1) / WG_BK * WG_BK;
    splits = (M + mps - 1) / mps;
    p.m_per_split = mps;
    static bool attr_set = false;
    const size_t lds = 2 * 2 * WG_BK * WG_PITCH * sizeof(uint16_t);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    wgrad_kernel<<<dim3((unsigned)tiles, (unsigned)splits), 256, lds, st>>>(p);
    return check_launch(what);
}

}  // namespace pe
