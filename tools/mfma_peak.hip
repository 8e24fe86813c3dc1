// Attainable bf16 MFMA rate on this GPU: every wave runs a register-only loop of independent
// v_mfma_f32_16x16x32_bf16 (or 32x32x16) chains, so the number printed is the ceiling any GEMM
// kernel on the chip can approach, to set beside the 2.5 PFLOP/s data-sheet peak.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_bin/mfma_peak && tools/_bin/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 4) void mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(seed + threadIdx.x * 1e-3f + i);
        b[i] = (__bf16)(seed - i);
    }
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) sum += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    } else {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 16; ++k) sum += acc[j][k];
    }
    if (sum == 12345.678f) out[0] = sum;  // keeps the loop alive without a store per thread
}

template <int SHAPE>
static double run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    mfma_loop<SHAPE><<<blocks, 256>>>(out, iters / 10, 1.0f);  // warm-up
    CHECK(hipEventRecord(e0));
    mfma_loop<SHAPE><<<blocks, 256>>>(out, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wave_iter = SHAPE == 16 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;
    const double flops = per_wave_iter * iters * 4.0 * blocks;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float* out;
    CHECK(hipMalloc(&out, 4));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.gcnArchName, cus,
           prop.clockRate / 1000);
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        const int blocks = cus * wg_per_cu;
        printf(", \"mfma16x16x32_bf16_tflops_%dwg\": %.1f", wg_per_cu, run<16>(blocks, 20000, out));
        printf(", \"mfma32x32x16_bf16_tflops_%dwg\": %.1f", wg_per_cu, run<32>(blocks, 20000, out));
    }
    printf("}\n");
    CHECK(hipFree(out));
    return 0;
}
