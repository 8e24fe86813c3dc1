// Gradient exchange without a host framework: a communicator over RCCL (xGMI inside a node)
// and a bucket all-reduce.  A scale other than 1 is folded into the reduction itself (RCCL's
// pre-multiplied sum: every rank's contribution is multiplied as it is read), so no second pass
// touches the bucket; the training path passes 1 and lets the optimizer kernel apply 1/world.
//
// RCCL is loaded at the first pe_comm_* call (dlopen), so the library itself carries no link
// dependency on it and loads on hosts without a GPU.  One communicator per process / GPU; the
// 128-byte unique id is made by one rank (pe_comm_unique_id) and handed to the others by
// whatever channel launched them (environment, file, socket).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "pe_common.h"

namespace pe {

// the slice of the RCCL interface used here (values from rccl.h)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
constexpr int kRcclSum = 0, kRcclFloat = 7, kRcclBfloat16 = 9, kRcclScalarHostImmediate = 1;

// float -> bf16 bits, round to nearest even (host side; finite inputs)
static uint16_t host_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static float host_bf16_float(float f) {   // f rounded to bf16, as a float
    const uint32_t u = (uint32_t)host_bf16_bits(f) << 16;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*RedOpCreatePreMulSum)(int*, void*, int, int, rcclComm_t) = nullptr;
    int (*RedOpDestroy)(int, rcclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static void rccl_load() {
    RcclApi& api = g_rccl;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.RedOpCreatePreMulSum = reinterpret_cast<decltype(api.RedOpCreatePreMulSum)>(
        dlsym(api.handle, "ncclRedOpCreatePreMulSum"));
    api.RedOpDestroy =
        reinterpret_cast<decltype(api.RedOpDestroy)>(dlsym(api.handle, "ncclRedOpDestroy"));
    api.GetErrorString =
        reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        dlclose(api.handle);
        api.handle = nullptr;
    }
}

// Loaded once, whichever thread makes the first pe_comm_* call.
static RcclApi* rccl_api() {
    std::call_once(g_rccl_once, rccl_load);
    return g_rccl.handle ? &g_rccl : nullptr;
}

static const char* rccl_msg(const RcclApi* a, int rc) {
    return a->GetErrorString ? a->GetErrorString(rc) : "RCCL error";
}

// Gradient transport in bf16: 8 elements per thread (32 B in, 16 B out, or the reverse), a
// scalar tail.  HBM-bound: 6 bytes per element each way.
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src,
                                                        uint16_t* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        u32x4 o;
        o.x = pack_bf16x2_bits(a.x, a.y);
        o.y = pack_bf16x2_bits(a.z, a.w);
        o.z = pack_bf16x2_bits(b.x, b.y);
        o.w = pack_bf16x2_bits(b.z, b.w);
        reinterpret_cast<u32x4*>(dst)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const long i = (n8 << 3) + threadIdx.x;
        dst[i] = (uint16_t)float_to_bf16_bits(src[i]);
    }
}

__global__ __launch_bounds__(256) void unpack_bf16_kernel(const uint16_t* __restrict__ src,
                                                          float* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const u32x4 v = reinterpret_cast<const u32x4*>(src)[i];
        const f32x4 a = {bf16_bits_to_float(v.x & 0xffffu), bf16_bits_to_float(v.x >> 16),
                         bf16_bits_to_float(v.y & 0xffffu), bf16_bits_to_float(v.y >> 16)};
        const f32x4 b = {bf16_bits_to_float(v.z & 0xffffu), bf16_bits_to_float(v.z >> 16),
                         bf16_bits_to_float(v.w & 0xffffu), bf16_bits_to_float(v.w >> 16)};
        reinterpret_cast<f32x4*>(dst)[2 * i] = a;
        reinterpret_cast<f32x4*>(dst)[2 * i + 1] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const long i = (n8 << 3) + threadIdx.x;
        dst[i] = bf16_bits_to_float(src[i]);
    }
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

static unsigned cast_grid(long n) {
    long blocks = ((n >> 3) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)blocks;
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_grad_pack_bf16(const float* src, void* dst_bf16, int64_t n,
                                     pe_stream_t stream) {
    if (n == 0) return PE_OK;
    PE_REQUIRE(n > 0 && src && dst_bf16, PE_E_INVALID, "pe_grad_pack_bf16: NULL pointer or n < 0");
    PE_REQUIRE(aligned(src, 16) && aligned(dst_bf16, 16), PE_E_INVALID,
               "pe_grad_pack_bf16: pointers must be 16-byte aligned");
    pe::launch(pack_bf16_kernel, cast_grid(n), 256, 0, as_stream(stream), src, static_cast<uint16_t*>(dst_bf16), n);
    return check_launch("pe_grad_pack_bf16");
}

extern "C" int32_t pe_grad_unpack_bf16(const void* src_bf16, float* dst, int64_t n,
                                       pe_stream_t stream) {
    if (n == 0) return PE_OK;
    PE_REQUIRE(n > 0 && src_bf16 && dst, PE_E_INVALID,
               "pe_grad_unpack_bf16: NULL pointer or n < 0");
    PE_REQUIRE(aligned(src_bf16, 16) && aligned(dst, 16), PE_E_INVALID,
               "pe_grad_unpack_bf16: pointers must be 16-byte aligned");
    pe::launch(unpack_bf16_kernel, cast_grid(n), 256, 0, as_stream(stream), static_cast<const uint16_t*>(src_bf16), dst, n);
    return check_launch("pe_grad_unpack_bf16");
}

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
    // RCCL enqueues the collective itself, so a launch tape would hold at most this entry
    // point's own scale kernel: replaying that rescales the bucket without reducing it
    PE_REQUIRE(!pe::tape_recording(), PE_E_UNSUPPORTED,
               "pe_allreduce_bucket: a collective cannot be recorded on a launch tape; end the "
               "tape before the gradient exchange");
    RcclApi* a = rccl_api();
    PE_REQUIRE(a != nullptr, PE_E_UNSUPPORTED, "pe_allreduce_bucket: librccl could not be loaded");
    hipStream_t st = as_stream(stream);
    const int dt = dtype == PE_DTYPE_F32 ? kRcclFloat : kRcclBfloat16;
    int op = kRcclSum;
    if (scale != 1.0f && comm->world == 1) {
        // one rank: nothing is exchanged, so there is no reduction for the scale to ride in
        // (RCCL's one-rank pre-multiply also leaves the last elements of some float32 buckets
        // unscaled); the bucket is scaled in place
        const int rc1 = a->AllReduce(buf, buf, (size_t)n, dt, kRcclSum, comm->comm, st);
        PE_REQUIRE(rc1 == 0, PE_E_HIP, "pe_allreduce_bucket: %s", rccl_msg(a, rc1));
        long blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (dtype == PE_DTYPE_F32)
            pe::launch(scale_kernel<false>, (unsigned)blocks, 256, 0, st, buf, n, scale);
        else
            pe::launch(scale_kernel<true>, (unsigned)blocks, 256, 0, st, buf, n, host_bf16_float(scale));
        return check_launch("pe_allreduce_bucket(scale)");
    }
    if (scale != 1.0f) {
        PE_REQUIRE(a->RedOpCreatePreMulSum && a->RedOpDestroy, PE_E_UNSUPPORTED,
                   "pe_allreduce_bucket: this RCCL has no pre-multiplied sum; pass scale = 1");
        // the scalar is read before the call returns (host-immediate), in the bucket's dtype
        float sf = scale;
        uint16_t sb = host_bf16_bits(scale);
        const int rc0 = a->RedOpCreatePreMulSum(&op, dtype == PE_DTYPE_F32 ? (void*)&sf : (void*)&sb,
                                                dt, kRcclScalarHostImmediate, comm->comm);
        PE_REQUIRE(rc0 == 0, PE_E_HIP, "pe_allreduce_bucket: %s", rccl_msg(a, rc0));
    }
    const int rc = a->AllReduce(buf, buf, (size_t)n, dt, op, comm->comm, st);
    if (op != kRcclSum) a->RedOpDestroy(op, comm->comm);   // safe once the collective is enqueued
    PE_REQUIRE(rc == 0, PE_E_HIP, "pe_allreduce_bucket: %s", rccl_msg(a, rc));
    return PE_OK;
}

extern "C" int32_t pe_comm_destroy(pe_comm* comm) {
    if (comm == nullptr) return PE_OK;
    RcclApi* a = rccl_api();
    if (a != nullptr && comm->comm != nullptr) a->CommDestroy(comm->comm);
    delete comm;
    return PE_OK;
}
