// Gradient exchange without a host framework: a communicator over RCCL (xGMI inside a node)
// and a bucket all-reduce that folds the 1/world averaging in.
//
// RCCL is loaded at the first pe_comm_* call (dlopen), so the library itself carries no link
// dependency on it and loads on hosts without a GPU.  One communicator per process / GPU; the
// 128-byte unique id is made by one rank (pe_comm_unique_id) and handed to the others by
// whatever channel launched them (environment, file, socket).
#include <dlfcn.h>
#include <string.h>

#include "pe_common.h"

namespace pe {

// the slice of the RCCL interface used here (values from rccl.h)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
constexpr int kRcclSum = 0, kRcclFloat = 7, kRcclBfloat16 = 9;

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

static RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) return nullptr;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.GetErrorString =
        reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        dlclose(api.handle);
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

static const char* rccl_msg(const RcclApi* a, int rc) {
    return a->GetErrorString ? a->GetErrorString(rc) : "RCCL error";
}

template <bool BF16>
__global__ __launch_bounds__(256) void scale_kernel(void* buf, long n, float s) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (BF16) {
            uint16_t* p = static_cast<uint16_t*>(buf);
            p[i] = (uint16_t)float_to_bf16_bits(bf16_bits_to_float(p[i]) * s);
        } else {
            static_cast<float*>(buf)[i] *= s;
        }
    }
}

}  // namespace pe

using namespace pe;

struct pe_comm {
    rcclComm_t comm;
    int32_t rank, world;
};

extern "C" int32_t pe_comm_unique_id(uint8_t id[128]) {
    PE_REQUIRE(id != nullptr, PE_E_INVALID, "pe_comm_unique_id: NULL id");
    RcclApi* a = rccl_api();
    PE_REQUIRE(a != nullptr, PE_E_UNSUPPORTED, "pe_comm_unique_id: librccl could not be loaded");
    rcclUniqueId u;
    const int rc = a->GetUniqueId(&u);
    PE_REQUIRE(rc == 0, PE_E_HIP, "pe_comm_unique_id: %s", rccl_msg(a, rc));
    memcpy(id, u.internal, 128);
    return PE_OK;
}

extern "C" int32_t pe_comm_create(int32_t rank, int32_t world, const uint8_t id[128],
                                  pe_comm** out) {
    PE_REQUIRE(out != nullptr && id != nullptr, PE_E_INVALID, "pe_comm_create: NULL pointer");
    PE_REQUIRE(world >= 1 && rank >= 0 && rank < world, PE_E_INVALID,
               "pe_comm_create: rank %d of %d", rank, world);
    RcclApi* a = rccl_api();
    PE_REQUIRE(a != nullptr, PE_E_UNSUPPORTED, "pe_comm_create: librccl could not be loaded");
    rcclUniqueId u;
    memcpy(u.internal, id, 128);
    rcclComm_t c = nullptr;
    const int rc = a->CommInitRank(&c, world, u, rank);   // on the calling thread's current GPU
    PE_REQUIRE(rc == 0, PE_E_HIP, "pe_comm_create: %s", rccl_msg(a, rc));
    *out = new pe_comm{c, rank, world};
    return PE_OK;
}

extern "C" int32_t pe_allreduce_bucket(pe_comm* comm, void* buf, int64_t n, int32_t dtype,
                                       float scale, pe_stream_t stream) {
    PE_REQUIRE(comm != nullptr && buf != nullptr && n > 0, PE_E_INVALID,
               "pe_allreduce_bucket: NULL pointer or empty bucket");
    PE_REQUIRE(dtype == PE_DTYPE_F32 || dtype == PE_DTYPE_BF16, PE_E_INVALID,
               "pe_allreduce_bucket: dtype must be PE_DTYPE_F32 or PE_DTYPE_BF16");
    RcclApi* a = rccl_api();
    PE_REQUIRE(a != nullptr, PE_E_UNSUPPORTED, "pe_allreduce_bucket: librccl could not be loaded");
    hipStream_t st = as_stream(stream);
    const int rc = a->AllReduce(buf, buf, (size_t)n, dtype == PE_DTYPE_F32 ? kRcclFloat : kRcclBfloat16,
                                kRcclSum, comm->comm, st);
    PE_REQUIRE(rc == 0, PE_E_HIP, "pe_allreduce_bucket: %s", rccl_msg(a, rc));
    if (scale != 1.0f) {
        long blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (dtype == PE_DTYPE_F32)
            scale_kernel<false><<<(unsigned)blocks, 256, 0, st>>>(buf, n, scale);
        else
            scale_kernel<true><<<(unsigned)blocks, 256, 0, st>>>(buf, n, scale);
        return check_launch("pe_allreduce_bucket(scale)");
    }
    return PE_OK;
}

extern "C" int32_t pe_comm_destroy(pe_comm* comm) {
    if (comm == nullptr) return PE_OK;
    RcclApi* a = rccl_api();
    if (a != nullptr && comm->comm != nullptr) a->CommDestroy(comm->comm);
    delete comm;
    return PE_OK;
}
