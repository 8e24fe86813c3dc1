// Internal helpers shared by the HIP translation units of libpe_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "pe_api.h"

namespace pe {

// Records the message returned by pe_last_error() on this thread.
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

constexpr int kWave = 64;  // gfx950 wavefront width

struct GaussTable {
    float g[PE_MAX_RADIUS + 1];
};

// Fills t.g[0..radius] exactly as pe_gaussian_table_f32 does; entries past radius are 0.
void fill_gauss_table(float sigma, int radius, GaussTable* t);

// Stage 2 of the two-stage loss reductions (pe_mse.hip): partial[b*K + k] -> per_map[k] =
// per_map_scale * sum_b, loss = loss_scale * sum_k sum_b, in a fixed order (K <= 4096).
int32_t launch_map_sum_reduce(const float* partial, int B, int K, float per_map_scale,
                              float loss_scale, float* per_map, float* loss, hipStream_t st,
                              const char* what);

inline hipStream_t as_stream(pe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that was just enqueued; returns PE_OK or PE_E_HIP.
int32_t check_launch(const char* what);

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace pe

#define PE_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            pe::set_error(__VA_ARGS__);      \
            return (code);                   \
        }                                    \
    } while (0)

#ifdef __HIPCC__
#include <atomic>
#include <tuple>
#include <utility>

namespace pe {

// ---- launch tape --------------------------------------------------------------------------
// Every kernel of the library is enqueued through pe::launch.  While a tape is recording on
// the calling thread (pe_tape_begin .. pe_tape_end) each launch is also stored with a copy of
// its arguments, so that the same sequence can be enqueued again by pe_tape_replay without any
// of the host work that led to it (argument checks, plan choice, the caller's own code).
struct TapeRec {
    virtual void fire() = 0;
    virtual ~TapeRec() {}
};
bool tape_recording();
void tape_push(TapeRec* r);

template <class... P>
struct KernelRec final : TapeRec {
    void (*k)(P...);
    dim3 g, b;
    unsigned lds;
    hipStream_t st;
    std::tuple<P...> a;
    void* ptr[sizeof...(P) ? sizeof...(P) : 1];
    template <class... A>
    KernelRec(void (*k_)(P...), dim3 g_, dim3 b_, unsigned l_, hipStream_t s_, A&&... x)
        : k(k_), g(g_), b(b_), lds(l_), st(s_), a(std::forward<A>(x)...) {
        fill(std::index_sequence_for<P...>());
    }
    template <size_t... I>
    void fill(std::index_sequence<I...>) {
        ((ptr[I] = const_cast<void*>(static_cast<const void*>(&std::get<I>(a)))), ...);
    }
    void fire() override {
        (void)hipLaunchKernel(reinterpret_cast<const void*>(k), g, b, ptr, lds, st);
    }
};

struct MemsetRec final : TapeRec {
    void* dst;
    int value;
    size_t bytes;
    hipStream_t st;
    MemsetRec(void* d, int v, size_t n, hipStream_t s) : dst(d), value(v), bytes(n), st(s) {}
    void fire() override { (void)hipMemsetAsync(dst, value, bytes, st); }
};

template <class... P, class... A>
inline void launch(void (*k)(P...), dim3 g, dim3 b, size_t lds, hipStream_t st, A&&... a) {
    if (tape_recording()) tape_push(new KernelRec<P...>(k, g, b, (unsigned)lds, st, a...));
    k<<<g, b, lds, st>>>(a...);
}

// Opt-in to more than 64 KB of dynamic LDS for `Kernel`.  The attribute belongs to the
// (kernel, device) pair, so it is set once per device ordinal of the process, on the calling
// thread's current device, and the result is kept: a failure is reported on every launch
// attempt on that device, not only the first.
template <auto Kernel>
inline hipError_t ensure_dynamic_lds(int bytes) {
    constexpr int kMaxDev = 64;
    static std::atomic<int> state[kMaxDev];   // 0 = not set, 1 = set, 2 + hipError_t = failed
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
    const int s = state[dev].load(std::memory_order_acquire);
    if (s == 1) return hipSuccess;
    if (s >= 2) return static_cast<hipError_t>(s - 2);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    state[dev].store(e == hipSuccess ? 1 : 2 + static_cast<int>(e), std::memory_order_release);
    return e;
}

inline hipError_t memset_async(void* dst, int value, size_t bytes, hipStream_t st) {
    if (tape_recording()) tape_push(new MemsetRec(dst, value, bytes, st));
    return hipMemsetAsync(dst, value, bytes, st);
}

// Native 16-byte vector types: the non-temporal load/store builtins accept only these.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// (value, index) arg-max combine with numpy.argmax tie-breaking: the larger value wins and,
// between equal values, the lower index.
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    const bool take = (ov > v) || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
}

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        argmax_combine(v, i, ov, oi);
    }
}

// Butterfly sum: every lane ends with the same value, and the order of additions is fixed.
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }

// Two floats -> two bf16 (round to nearest even) packed lo | hi << 16: one v_cvt_pk_bf16_f32
// on gfx950 instead of about a dozen integer instructions.
__device__ __forceinline__ uint32_t pack_bf16x2_bits(float lo, float hi) {
    typedef float pe_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pe_bf16x2 __attribute__((ext_vector_type(2)));
    const pe_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pe_bf16x2));
}

__device__ __forceinline__ uint32_t float_to_bf16_bits(float f) {
    return pack_bf16x2_bits(f, 0.0f) & 0xffffu;
}

}  // namespace pe
#endif
