// Error reporting, version, host-only helpers and the launch tape of libpe_hip.so.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <memory>
#include <vector>

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

// ---- launch tape --------------------------------------------------------------------------
struct Tape {
    std::vector<std::unique_ptr<TapeRec>> recs;
    std::vector<hipEvent_t> events;
    ~Tape() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

// cross-stream order inside a tape: record an event on `waited`, make `waiter` wait for it
struct WaitRec final : TapeRec {
    hipEvent_t ev;
    hipStream_t waiter, waited;
    WaitRec(hipEvent_t e, hipStream_t a, hipStream_t b) : ev(e), waiter(a), waited(b) {}
    void fire() override {
        (void)hipEventRecord(ev, waited);
        (void)hipStreamWaitEvent(waiter, ev, 0);
    }
};

static thread_local Tape* g_tape = nullptr;   // the tape recording on this thread, if any

bool tape_recording() { return g_tape != nullptr; }
void tape_push(TapeRec* r) { g_tape->recs.emplace_back(r); }

}  // namespace pe

extern "C" {

int32_t pe_tape_begin(void) {
    PE_REQUIRE(pe::g_tape == nullptr, PE_E_INVALID, "pe_tape_begin: a tape is already recording");
    pe::g_tape = new pe::Tape();
    return PE_OK;
}

int64_t pe_tape_length(void) {
    return pe::g_tape ? (int64_t)pe::g_tape->recs.size() : -1;
}

int32_t pe_tape_end(pe_tape_t* tape) {
    PE_REQUIRE(tape != nullptr, PE_E_INVALID, "pe_tape_end: tape is NULL");
    PE_REQUIRE(pe::g_tape != nullptr, PE_E_INVALID, "pe_tape_end: no tape is recording");
    *tape = pe::g_tape;
    pe::g_tape = nullptr;
    return PE_OK;
}

int32_t pe_stream_wait_stream(pe_stream_t waiter, pe_stream_t waited) {
    if (waiter == waited) return PE_OK;
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, pe::as_stream(waited));
    if (e == hipSuccess) e = hipStreamWaitEvent(pe::as_stream(waiter), ev, 0);
    if (e != hipSuccess) {
        pe::set_error("pe_stream_wait_stream: %s", hipGetErrorString(e));
        return PE_E_HIP;
    }
    if (pe::g_tape) {   // the tape owns the event and re-records it at every replay
        pe::g_tape->events.push_back(ev);
        pe::tape_push(new pe::WaitRec(ev, pe::as_stream(waiter), pe::as_stream(waited)));
    } else {
        (void)hipEventDestroy(ev);   // released by the runtime once the recorded work is done
    }
    return PE_OK;
}

int32_t pe_tape_replay(pe_tape_t tape, int64_t first, int64_t count) {
    PE_REQUIRE(tape != nullptr, PE_E_INVALID, "pe_tape_replay: tape is NULL");
    PE_REQUIRE(pe::g_tape == nullptr, PE_E_INVALID, "pe_tape_replay: a tape is recording");
    pe::Tape* t = static_cast<pe::Tape*>(tape);
    const int64_t n = (int64_t)t->recs.size();
    PE_REQUIRE(first >= 0 && count >= 0 && first + count <= n, PE_E_INVALID,
               "pe_tape_replay: [%lld, %lld) outside the tape's %lld records", (long long)first,
               (long long)(first + count), (long long)n);
    (void)hipGetLastError();
    for (int64_t i = first; i < first + count; ++i) t->recs[(size_t)i]->fire();
    return pe::check_launch("pe_tape_replay");
}

int32_t pe_tape_destroy(pe_tape_t tape) {
    PE_REQUIRE(tape != pe::g_tape || tape == nullptr, PE_E_INVALID,
               "pe_tape_destroy: the tape is still recording");
    delete static_cast<pe::Tape*>(tape);
    return PE_OK;
}

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
