// Error reporting, version and host-only helpers of libpe_hip.so.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "pe_common.h"

namespace pe {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

int32_t check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PE_E_HIP;
    }
    return PE_OK;
}

void fill_gauss_table(float sigma, int radius, GaussTable* t) {
    const double s = (double)sigma;
    for (int i = 0; i <= PE_MAX_RADIUS; ++i) {
        t->g[i] = (i <= radius) ? (float)exp(-((double)i * (double)i) / (2.0 * s * s)) : 0.0f;
    }
}

}  // namespace pe

extern "C" {

int32_t pe_version(void) { return 1000; }

const char* pe_last_error(void) { return pe::g_err; }

int32_t pe_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        pe::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PE_E_HIP;
    }
    return n;
}

int32_t pe_gaussian_table_f32(float sigma, int32_t radius, float* table) {
    PE_REQUIRE(table != nullptr, PE_E_INVALID, "pe_gaussian_table_f32: table is NULL");
    PE_REQUIRE(radius >= 0 && radius <= PE_MAX_RADIUS, PE_E_UNSUPPORTED,
               "pe_gaussian_table_f32: radius %d outside [0, %d]", radius, PE_MAX_RADIUS);
    PE_REQUIRE(sigma > 0.0f && isfinite(sigma), PE_E_INVALID,
               "pe_gaussian_table_f32: sigma must be positive and finite");
    pe::GaussTable t;
    pe::fill_gauss_table(sigma, radius, &t);
    memcpy(table, t.g, sizeof(float) * (size_t)(radius + 1));
    return PE_OK;
}

}  // extern "C"
