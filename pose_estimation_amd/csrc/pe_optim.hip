// a6. Fused optimizer steps over a flat float32 arena (Adam, momentum SGD).
//
// Bound: HBM bandwidth.  Adam moves 16 B in (p, m, v, g) and 12 B + 2 B out (p, m, v, bf16
// copy) per parameter: 30 B.  One grid-stride kernel, 16-byte accesses, a scalar tail.
// Arithmetic is plain float32 in a fixed order with FMA contraction off (the file is built
// with -ffp-contract=off), so the result is bit-identical to the float32 NumPy restatement.
#include <math.h>

#include "pe_common.h"

namespace pe {

struct AdamHyper {
    float beta1, beta2, omb1, omb2, eps, step_size, denom_scale, grad_scale;
};

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g,
                                         const AdamHyper& h) {
    g = g * h.grad_scale;
    m = h.beta1 * m + h.omb1 * g;
    v = h.beta2 * v + (h.omb2 * g) * g;
    const float den = sqrtf(v) * h.denom_scale + h.eps;
    p = p - (h.step_size * m) / den;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v,
                                                   const float* __restrict__ grad,
                                                   uint16_t* __restrict__ pb, long n,
                                                   AdamHyper h) {
    const long n4 = n >> 2;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const long nth = (long)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(grad);
    uint2* pb4 = reinterpret_cast<uint2*>(pb);
    for (long i = tid; i < n4; i += nth) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const f32x4 gg = __builtin_nontemporal_load(g4 + i);
        adam_one(pp.x, mm.x, vv.x, gg.x, h);
        adam_one(pp.y, mm.y, vv.y, gg.y, h);
        adam_one(pp.z, mm.z, vv.z, gg.z, h);
        adam_one(pp.w, mm.w, vv.w, gg.w, h);
        p4[i] = pp;
        m4[i] = mm;
        v4[i] = vv;
        if (pb) {
            uint2 o;
            o.x = pack_bf16x2_bits(pp.x, pp.y);
            o.y = pack_bf16x2_bits(pp.z, pp.w);
            pb4[i] = o;
        }
    }
    for (long i = n4 * 4 + tid; i < n; i += nth) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, mm, vv, grad[i], h);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (pb) pb[i] = (uint16_t)float_to_bf16_bits(pp);
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ mom,
                                                  const float* __restrict__ grad,
                                                  uint16_t* __restrict__ pb, long n, float lr,
                                                  float momentum, float grad_scale) {
    const long n4 = n >> 2;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const long nth = (long)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(mom);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(grad);
    uint2* pb4 = reinterpret_cast<uint2*>(pb);
    auto one = [&](float& pp, float& mm, float g) {
        g = g * grad_scale;
        mm = momentum * mm + g;
        pp = pp - lr * mm;
    };
    for (long i = tid; i < n4; i += nth) {
        float4 pp = p4[i], mm = m4[i];
        const f32x4 gg = __builtin_nontemporal_load(g4 + i);
        one(pp.x, mm.x, gg.x);
        one(pp.y, mm.y, gg.y);
        one(pp.z, mm.z, gg.z);
        one(pp.w, mm.w, gg.w);
        p4[i] = pp;
        m4[i] = mm;
        if (pb) {
            uint2 o;
            o.x = pack_bf16x2_bits(pp.x, pp.y);
            o.y = pack_bf16x2_bits(pp.z, pp.w);
            pb4[i] = o;
        }
    }
    for (long i = n4 * 4 + tid; i < n; i += nth) {
        float pp = p[i], mm = mom[i];
        one(pp, mm, grad[i]);
        p[i] = pp;
        mom[i] = mm;
        if (pb) pb[i] = (uint16_t)float_to_bf16_bits(pp);
    }
}

static unsigned optim_grid(long n) {
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 128) blocks = 256 * 128;
    return (unsigned)blocks;
}

}  // namespace pe

extern "C" int32_t pe_adam_step(float* p, float* m, float* v, const float* grad, uint16_t* p_bf16,
                                int64_t n, float beta1, float beta2, float eps, float step_size,
                                float denom_scale, float grad_scale, pe_stream_t stream) {
    using namespace pe;
    PE_REQUIRE(n >= 0, PE_E_INVALID, "pe_adam_step: negative n");
    if (n == 0) return PE_OK;
    PE_REQUIRE(p && m && v && grad, PE_E_INVALID, "pe_adam_step: NULL arena pointer");
    PE_REQUIRE(aligned(p, 16) && aligned(m, 16) && aligned(v, 16) && aligned(grad, 16) &&
                   aligned(p_bf16, 8),
               PE_E_INVALID, "pe_adam_step: arenas must be 16-byte aligned (bf16 copy: 8)");
    AdamHyper h;
    h.beta1 = beta1;
    h.beta2 = beta2;
    h.omb1 = (float)(1.0 - (double)beta1);
    h.omb2 = (float)(1.0 - (double)beta2);
    h.eps = eps;
    h.step_size = step_size;
    h.denom_scale = denom_scale;
    h.grad_scale = grad_scale;
    pe::launch(adam_kernel, optim_grid(n), 256, 0, as_stream(stream), p, m, v, grad, p_bf16, (long)n, h);
    return check_launch("pe_adam_step");
}

extern "C" int32_t pe_sgd_step(float* p, float* mom, const float* grad, uint16_t* p_bf16,
                               int64_t n, float lr, float momentum, float grad_scale,
                               pe_stream_t stream) {
    using namespace pe;
    PE_REQUIRE(n >= 0, PE_E_INVALID, "pe_sgd_step: negative n");
    if (n == 0) return PE_OK;
    PE_REQUIRE(p && mom && grad, PE_E_INVALID, "pe_sgd_step: NULL arena pointer");
    PE_REQUIRE(aligned(p, 16) && aligned(mom, 16) && aligned(grad, 16) && aligned(p_bf16, 8),
               PE_E_INVALID, "pe_sgd_step: arenas must be 16-byte aligned (bf16 copy: 8)");
    pe::launch(sgd_kernel, optim_grid(n), 256, 0, as_stream(stream), p, mom, grad, p_bf16, (long)n, lr,
                                                            momentum, grad_scale);
    return check_launch("pe_sgd_step");
}
