/* A plain-C consumer of libpe_hip.so: no Python, no torch.  It allocates device memory with the
 * HIP runtime, calls pe_render_targets_f32 -> pe_decode_f32 -> pe_mse_fwd_bwd through the C ABI
 * declared in include/pe_api.h and checks known answers:
 *   - a rendered joint decodes back to its centre cell with confidence 1,
 *   - a joint whose centre is off the map has weight 0 and an empty map,
 *   - the loss of a prediction equal to its own target is exactly 0 with an all-zero gradient,
 *   - an invalid argument returns PE_E_INVALID and sets pe_last_error().
 * Built by __graft_entry__.build() with hipcc (C mode, host only); run by tests/test_c_abi_program.py.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pe_api.h"

#define CHECK_HIP(x)                                                          \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                         \
        }                                                                     \
    } while (0)
#define CHECK_PE(x)                                                           \
    do {                                                                      \
        int32_t rc_ = (x);                                                    \
        if (rc_ != PE_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pe_last_error());      \
            return 3;                                                         \
        }                                                                     \
    } while (0)
#define EXPECT(c)                                                             \
    do {                                                                      \
        if (!(c)) {                                                           \
            fprintf(stderr, "expectation failed: %s (line %d)\n", #c, __LINE__); \
            return 4;                                                         \
        }                                                                     \
    } while (0)

int main(void) {
    enum { B = 2, K = 3, H = 64, W = 48 };
    const float stride = 4.0f, sigma = 2.0f;
    const int radius = 6;
    /* (x, y, v) in input pixels: centres (12, 10), (47, 63), off the map, unlabelled, ... */
    const float joints[B * K * 3] = {48.0f, 40.0f, 2.0f, 189.9f, 253.0f, 1.0f, 192.0f, 100.0f, 2.0f,
                                     20.0f, 20.0f, 0.0f, 100.0f, 128.0f, 2.0f, 2.0f, 2.0f, 1.0f};
    if (pe_version() < 1000 || pe_device_count() < 1) {
        fprintf(stderr, "no device or bad library version\n");
        return 1;
    }
    float *d_j, *d_t, *d_w, *d_xy, *d_conf, *d_loss, *d_pj, *d_g;
    int32_t* d_idx;
    void* d_ws;
    const size_t nmap = (size_t)B * K * H * W;
    CHECK_HIP(hipMalloc((void**)&d_j, sizeof(joints)));
    CHECK_HIP(hipMalloc((void**)&d_t, nmap * 4));
    CHECK_HIP(hipMalloc((void**)&d_g, nmap * 4));
    CHECK_HIP(hipMalloc((void**)&d_w, B * K * 4));
    CHECK_HIP(hipMalloc((void**)&d_xy, B * K * 8));
    CHECK_HIP(hipMalloc((void**)&d_conf, B * K * 4));
    CHECK_HIP(hipMalloc((void**)&d_idx, B * K * 4));
    CHECK_HIP(hipMalloc((void**)&d_loss, 4));
    CHECK_HIP(hipMalloc((void**)&d_pj, K * 4));
    CHECK_HIP(hipMalloc(&d_ws, pe_mse_workspace_bytes(B, K)));
    CHECK_HIP(hipMemcpy(d_j, joints, sizeof(joints), hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    CHECK_PE(pe_render_targets_f32(d_j, B, K, H, W, stride, sigma, radius, d_t, (int64_t)K * H * W,
                                   (int64_t)H * W, W, 1, d_w, st));
    CHECK_PE(pe_decode_f32(d_t, B, K, H, W, (int64_t)K * H * W, (int64_t)H * W, W, 1, 1, d_xy, d_conf,
                           d_idx, st));
    CHECK_PE(pe_mse_fwd_bwd(d_t, PE_DTYPE_F32, d_t, NULL, d_w, NULL, B, K, H, W, stride, sigma, radius,
                            d_loss, d_pj, d_g, d_ws, st));
    CHECK_HIP(hipStreamSynchronize(st));

    float w[B * K], xy[B * K * 2], conf[B * K], loss;
    int32_t idx[B * K];
    float* g = (float*)malloc(nmap * 4);
    CHECK_HIP(hipMemcpy(w, d_w, sizeof(w), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(xy, d_xy, sizeof(xy), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(conf, d_conf, sizeof(conf), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(idx, d_idx, sizeof(idx), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&loss, d_loss, 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(g, d_g, nmap * 4, hipMemcpyDeviceToHost));

    EXPECT(w[0] == 1.0f && idx[0] == 10 * W + 12 && conf[0] == 1.0f && xy[0] == 12.0f && xy[1] == 10.0f);
    EXPECT(w[1] == 1.0f && idx[1] == 63 * W + 47 && conf[1] == 1.0f);
    EXPECT(w[2] == 0.0f && conf[2] == 0.0f && xy[4] == 0.0f); /* centre column 48 of 48: not drawn */
    EXPECT(w[3] == 0.0f && conf[3] == 0.0f);                   /* v == 0 */
    EXPECT(w[4] == 1.0f && idx[4] == 32 * W + 25);
    EXPECT(w[5] == 1.0f && idx[5] == 1 * W + 1);
    EXPECT(loss == 0.0f);
    for (size_t i = 0; i < nmap; ++i) EXPECT(g[i] == 0.0f);

    /* error path: negative width */
    EXPECT(pe_decode_f32(d_t, B, K, H, -1, 0, 0, 0, 1, 1, d_xy, d_conf, d_idx, st) == PE_E_INVALID);
    EXPECT(strlen(pe_last_error()) > 0);
    printf("c_abi_smoke ok: render, decode and loss through the C ABI from plain C\n");
    free(g);
    return 0;
}
