// a3. Masked per-joint MSE loss and gradient, with an optional fused target render.
//
// Bound: HBM bandwidth.  Unfused float32: read pred + read target + write grad = 3 x 4 B per
// element; fused (target rendered from the joints in registers): 2 x 4 B, or 2 x 2 B with bf16
// pred/grad.  One wave per (b,k) map, 16-byte accesses.  Stage 1 leaves one partial sum per map
// in the workspace; stage 2 (one workgroup) folds them in a fixed order, so the loss is
// bit-stable from run to run (no float atomics).
#include <math.h>

#include <mutex>

#include "pe_common.h"
#include "pe_render_device.h"

namespace pe {

struct MseArgs {
    const void* pred;
    const float* target;      // unfused
    const float* joints;      // fused
    const float* weight_in;   // unfused, may be null (= 1)
    float* weight_out;        // fused, may be null
    const uint8_t* mask;      // may be null
    void* grad;               // may be null
    float* partial;           // [nmaps]
    int nmaps, K, H, W;
    float stride;
    int radius;
    float inv_n;
};

template <bool BF16, bool FUSED, bool HAS_MASK, bool HAS_GRAD>
__global__ __launch_bounds__(256) void mse_map_kernel(MseArgs a, GaussTable tab) {
    constexpr int CH = BF16 ? 8 : 4;  // elements per 16-byte pred access
    __shared__ float g[PE_MAX_RADIUS + 1];
    if (FUSED) {
        if (threadIdx.x <= PE_MAX_RADIUS) g[threadIdx.x] = tab.g[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long map = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= a.nmaps) return;  // wave-uniform
    const int n = a.H * a.W;
    const int nch = n / CH;
    const long b = map / a.K;

    float w;
    JointWindow jw{0, 0, false};
    if (FUSED) {
        jw = joint_window(a.joints + map * 3, a.H, a.W, a.stride, a.radius);
        w = jw.on ? 1.0f : 0.0f;
        if (lane == 0 && a.weight_out) a.weight_out[map] = w;
    } else {
        w = a.weight_in ? a.weight_in[map] : 1.0f;
    }
    const float cg = (w * w) * a.inv_n;
    const int wch = a.W / CH;  // chunks per row (fused path only)

    const u32x4* p16 = reinterpret_cast<const u32x4*>(
        reinterpret_cast<const char*>(a.pred) + map * (long)n * (BF16 ? 2 : 4));
    const f32x4* t4 = FUSED ? nullptr : reinterpret_cast<const f32x4*>(a.target + map * (long)n);
    u32x4* g16 = HAS_GRAD ? reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.grad) +
                                                     map * (long)n * (BF16 ? 2 : 4))
                          : nullptr;
    const uint8_t* mrow = HAS_MASK ? a.mask + b * (long)n : nullptr;

    float acc = 0.0f;
#pragma unroll 2
    for (int c = lane; c < nch; c += 64) {
        float p[CH], t[CH];
        const u32x4 pv = __builtin_nontemporal_load(p16 + c);
        if constexpr (BF16) {
            const uint32_t u[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[2 * i] = bf16_bits_to_float(u[i] & 0xffffu);
                p[2 * i + 1] = bf16_bits_to_float(u[i] >> 16);
            }
        } else {
            p[0] = __uint_as_float(pv.x);
            p[1] = __uint_as_float(pv.y);
            p[2] = __uint_as_float(pv.z);
            p[3] = __uint_as_float(pv.w);
        }
        if (FUSED) {
            const int y = c / wch;
            const int x0 = (c - y * wch) * CH;
            const int ady = abs(y - jw.cy);
            const bool row_on = jw.on && ady <= a.radius && x0 + (CH - 1) >= jw.cx - a.radius &&
                                x0 <= jw.cx + a.radius;
            const float gy = row_on ? g[ady] : 0.0f;
#pragma unroll
            for (int i = 0; i < CH; ++i)
                t[i] = row_on ? gauss_x(g, x0 + i, jw.cx, a.radius) * gy : 0.0f;
        } else {
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const f32x4 tv = __builtin_nontemporal_load(t4 + c * (CH / 4) + q);
                t[4 * q + 0] = tv.x;
                t[4 * q + 1] = tv.y;
                t[4 * q + 2] = tv.z;
                t[4 * q + 3] = tv.w;
            }
        }
        bool m[CH];
        if (HAS_MASK) {
            if constexpr (BF16) {
                const uint2 mv = *reinterpret_cast<const uint2*>(mrow + (long)c * CH);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m[i] = ((mv.x >> (8 * i)) & 0xffu) != 0;
                    m[4 + i] = ((mv.y >> (8 * i)) & 0xffu) != 0;
                }
            } else {
                const uint32_t mv = *reinterpret_cast<const uint32_t*>(mrow + (long)c * CH);
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = ((mv >> (8 * i)) & 0xffu) != 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) m[i] = true;
        }
        float gr[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float d = p[i] - t[i];
            const float e = w * d;
            const float q = e * e;
            acc += m[i] ? q : 0.0f;
            gr[i] = m[i] ? cg * d : 0.0f;
        }
        if (HAS_GRAD) {
            u32x4 o;
            if constexpr (BF16) {
                o.x = pack_bf16x2_bits(gr[0], gr[1]);
                o.y = pack_bf16x2_bits(gr[2], gr[3]);
                o.z = pack_bf16x2_bits(gr[4], gr[5]);
                o.w = pack_bf16x2_bits(gr[6], gr[7]);
            } else {
                o.x = __float_as_uint(gr[0]);
                o.y = __float_as_uint(gr[1]);
                o.z = __float_as_uint(gr[2]);
                o.w = __float_as_uint(gr[3]);
            }
            __builtin_nontemporal_store(o, g16 + c);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) a.partial[map] = acc;
}

// Stage 2: per-joint sums over the batch, then the K joint sums added in order.  One workgroup
// of 1024 threads.  partial is [B][K], so consecutive threads take consecutive joints (coalesced
// reads) and the S = 1024 / K thread slices share the batch: slice s adds rows s, s + S, ... in
// order, eight independent loads at a time (the stage is bound by load latency, not by bytes),
// then the slices are added in order.  The order of additions depends on (B, K) alone.
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void mse_reduce_kernel(
    const float* __restrict__ partial, int B, int K, float joint_scale, float loss_scale,
    float* per_joint, float* loss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sums = reinterpret_cast<float*>(smem);   // [K] joint sums
    float* red = sums + K;                          // [S][Kc] slice sums of the current chunk
    const int tid = threadIdx.x;
    const int Kc = K < kReduceThreads ? K : kReduceThreads;   // joints per chunk
    const int S = kReduceThreads / Kc;
    const int kl = tid % Kc, sl = tid / Kc;
    for (int k0 = 0; k0 < K; k0 += Kc) {
        const int k = k0 + kl;
        float acc = 0.0f;
        if (sl < S && k < K) {
            const float* col = partial + k;
            int b = sl;
            for (; b + 7 * S < B; b += 8 * S) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(long)(b + u * S) * K];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; b < B; b += S) acc += col[(long)b * K];
            red[sl * Kc + kl] = acc;
        }
        __syncthreads();
        if (sl == 0 && k < K) {
            float t = red[kl];
            for (int q = 1; q < S; ++q) t += red[q * Kc + kl];
            sums[k] = t;
            if (per_joint) per_joint[k] = t * joint_scale;
        }
        __syncthreads();
    }
    if (tid < 64) {   // lane l adds joints l, l + 64, ... in order, then the butterfly
        float tot = 0.0f;
        for (int k = tid; k < K; k += 64) tot += sums[k];
        tot = wave_sum(tot);
        if (tid == 0) *loss = tot * loss_scale;
    }
}

int32_t launch_map_sum_reduce(const float* partial, int B, int K, float per_map_scale,
                              float loss_scale, float* per_map, float* loss, hipStream_t st,
                              const char* what) {
    pe::launch(mse_reduce_kernel, 1, kReduceThreads, ((size_t)K + kReduceThreads) * sizeof(float), st, partial, B, K, per_map_scale, loss_scale, per_map, loss);
    return check_launch(what);
}

template <bool BF16, bool FUSED>
static void launch_mse(const MseArgs& a, const GaussTable& tab, unsigned grid, hipStream_t st) {
    const bool hm = a.mask != nullptr, hg = a.grad != nullptr;
    if (hm && hg) pe::launch(mse_map_kernel<BF16, FUSED, true, true>, grid, 256, 0, st, a, tab);
    else if (hm) pe::launch(mse_map_kernel<BF16, FUSED, true, false>, grid, 256, 0, st, a, tab);
    else if (hg) pe::launch(mse_map_kernel<BF16, FUSED, false, true>, grid, 256, 0, st, a, tab);
    else pe::launch(mse_map_kernel<BF16, FUSED, false, false>, grid, 256, 0, st, a, tab);
}

}  // namespace pe

extern "C" size_t pe_mse_workspace_bytes(int32_t B, int32_t K) {
    if (B <= 0 || K <= 0) return 16;
    return (((size_t)B * (size_t)K * sizeof(float)) + 15) & ~(size_t)15;
}

namespace pe {

// Validation and the two launches shared by pe_mse_fwd_bwd and pe_mse_fwd_bwd_static; `partial`
// is the per-map scratch (B*K floats of device memory), whoever owns it.
static int32_t mse_run(const void* pred, int32_t pred_dtype, const float* target,
                       const float* joints, float* weight, const uint8_t* mask, int32_t B,
                       int32_t K, int32_t H, int32_t W, float stride, float sigma, int32_t radius,
                       float* loss, float* per_joint, void* grad, float* partial,
                       pe_stream_t stream) {
    PE_REQUIRE(B > 0 && K > 0 && H > 0 && W > 0, PE_E_INVALID,
               "pe_mse_fwd_bwd: B, K, H, W must be positive");
    PE_REQUIRE(pred_dtype == PE_DTYPE_F32 || pred_dtype == PE_DTYPE_BF16, PE_E_INVALID,
               "pe_mse_fwd_bwd: unknown pred_dtype %d", pred_dtype);
    PE_REQUIRE(pred && loss && partial, PE_E_INVALID,
               "pe_mse_fwd_bwd: pred, loss and workspace must be non-NULL");
    PE_REQUIRE((target != nullptr) != (joints != nullptr), PE_E_INVALID,
               "pe_mse_fwd_bwd: exactly one of target and joints must be given");
    PE_REQUIRE(K <= 4096, PE_E_UNSUPPORTED, "pe_mse_fwd_bwd: K > 4096");
    const int64_t nmaps = (int64_t)B * K;
    PE_REQUIRE(nmaps < 0x7fffffff && (int64_t)H * W < 0x7fffffff, PE_E_UNSUPPORTED,
               "pe_mse_fwd_bwd: B*K or H*W does not fit int32");
    const bool bf16 = pred_dtype == PE_DTYPE_BF16;
    const int ch = bf16 ? 8 : 4;
    const bool fused = joints != nullptr;
    if (fused) {
        PE_REQUIRE(radius >= 0 && radius <= PE_MAX_RADIUS, PE_E_UNSUPPORTED,
                   "pe_mse_fwd_bwd: radius %d outside [0, %d]", radius, PE_MAX_RADIUS);
        PE_REQUIRE(sigma > 0.0f && isfinite(sigma) && stride > 0.0f && isfinite(stride),
                   PE_E_INVALID, "pe_mse_fwd_bwd: sigma and stride must be positive and finite");
        PE_REQUIRE(W % ch == 0, PE_E_UNSUPPORTED,
                   "pe_mse_fwd_bwd: fused rendering needs W %% %d == 0 (W = %d)", ch, W);
    } else {
        PE_REQUIRE(((int64_t)H * W) % ch == 0, PE_E_UNSUPPORTED,
                   "pe_mse_fwd_bwd: H*W must be a multiple of %d", ch);
        PE_REQUIRE(aligned(target, 16), PE_E_INVALID, "pe_mse_fwd_bwd: target not 16-byte aligned");
    }
    PE_REQUIRE(aligned(pred, 16) && aligned(grad, 16), PE_E_INVALID,
               "pe_mse_fwd_bwd: pred/grad not 16-byte aligned");
    PE_REQUIRE(mask == nullptr || aligned(mask, 8), PE_E_INVALID,
               "pe_mse_fwd_bwd: mask not 8-byte aligned");

    MseArgs a;
    a.pred = pred;
    a.target = target;
    a.joints = joints;
    a.weight_in = fused ? nullptr : weight;
    a.weight_out = fused ? weight : nullptr;
    a.mask = mask;
    a.grad = grad;
    a.partial = partial;
    a.nmaps = (int)nmaps;
    a.K = K;
    a.H = H;
    a.W = W;
    a.stride = stride;
    a.radius = radius;
    const double n = (double)B * K * H * W;
    a.inv_n = (float)(1.0 / n);
    GaussTable tab;
    if (fused) fill_gauss_table(sigma, radius, &tab);
    else fill_gauss_table(1.0f, 0, &tab);

    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)((nmaps + 3) / 4);
    if (bf16 && fused) launch_mse<true, true>(a, tab, grid, st);
    else if (bf16) launch_mse<true, false>(a, tab, grid, st);
    else if (fused) launch_mse<false, true>(a, tab, grid, st);
    else launch_mse<false, false>(a, tab, grid, st);
    int32_t rc = check_launch("pe_mse_fwd_bwd(stage 1)");
    if (rc != PE_OK) return rc;
    const float joint_scale = (float)(0.5 / ((double)B * H * W));
    const float loss_scale = (float)(0.5 / n);
    return launch_map_sum_reduce(a.partial, B, K, joint_scale, loss_scale, per_joint, loss, st,
                                 "pe_mse_fwd_bwd(stage 2)");
}

}  // namespace pe

extern "C" int32_t pe_mse_fwd_bwd(const void* pred, int32_t pred_dtype, const float* target,
                                  const float* joints, float* weight, const uint8_t* mask,
                                  int32_t B, int32_t K, int32_t H, int32_t W, float stride,
                                  float sigma, int32_t radius, float* loss, float* per_joint,
                                  void* grad, void* workspace, pe_stream_t stream) {
    return pe::mse_run(pred, pred_dtype, target, joints, weight, mask, B, K, H, W, stride, sigma,
                       radius, loss, per_joint, grad, static_cast<float*>(workspace), stream);
}

// ---- The allocation-free variant ---------------------------------------------------------
// The per-map partial sums live in a file-scope device array instead of a caller workspace.
// Every GPU the library runs on holds its own copy (256 KB) at its own address until the
// process ends.  In host code the array's NAME is only a stand-in for that memory: the device
// address comes from hipGetSymbolAddress, once per device with that device current, and only
// that pointer is ever handed to a kernel.
static __device__ __align__(256) float scratch[PE_MSE_STATIC_MAX_MAPS];

namespace pe {

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static float* g_scratch_ptr[kMaxDevices];  // device address of `scratch`, per device ordinal

static int32_t static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    PE_REQUIRE(e == hipSuccess, PE_E_HIP, "pe_mse_fwd_bwd_static: hipGetDevice: %s",
               hipGetErrorString(e));
    PE_REQUIRE(dev >= 0 && dev < kMaxDevices, PE_E_UNSUPPORTED,
               "pe_mse_fwd_bwd_static: device ordinal %d outside [0, %d)", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        PE_REQUIRE(e == hipSuccess && ptr != nullptr, PE_E_HIP,
                   "pe_mse_fwd_bwd_static: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                   hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return PE_OK;
}

}  // namespace pe

extern "C" int32_t pe_mse_fwd_bwd_static(const void* pred, int32_t pred_dtype,
                                         const float* target, const float* joints,
                                         float* weight, const uint8_t* mask, int32_t B, int32_t K,
                                         int32_t H, int32_t W, float stride, float sigma,
                                         int32_t radius, float* loss, float* per_joint,
                                         void* grad, pe_stream_t stream) {
    PE_REQUIRE(B > 0 && K > 0, PE_E_INVALID, "pe_mse_fwd_bwd_static: B and K must be positive");
    PE_REQUIRE((int64_t)B * K <= PE_MSE_STATIC_MAX_MAPS, PE_E_UNSUPPORTED,
               "pe_mse_fwd_bwd_static: B*K = %lld maps exceeds the static scratch of %d maps; "
               "use pe_mse_fwd_bwd with a caller workspace",
               (long long)B * K, PE_MSE_STATIC_MAX_MAPS);
    float* ptr = nullptr;
    const int32_t rc = pe::static_scratch(&ptr);
    if (rc != PE_OK) return rc;
    // Stage 1 writes partial[0 .. B*K) in full and stage 2 reads nothing beyond it, so what an
    // earlier (larger) call left behind is never seen and nothing has to be cleared.
    return pe::mse_run(pred, pred_dtype, target, joints, weight, mask, B, K, H, W, stride, sigma,
                       radius, loss, per_joint, grad, ptr, stream);
}
