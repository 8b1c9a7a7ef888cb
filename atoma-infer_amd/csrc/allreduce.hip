// Tensor-parallel sum all-reduce over RCCL / xGMI.
//
// Replaces /root/reference/models/src/multi_gpu.rs:141-179 (`AllReduce::cuda_fwd` ->
// cudarc `ncclAllReduce(sum)`, out-of-place, bf16/f16/f32) and the communicator bootstrap of
// /root/reference/backends/vllm/src/model_executor.rs:413,436-439 (`Id::new()` once,
// `Comm::from_rank(dev, rank, n, id)` per GPU thread).  Same shape: one communicator per GPU
// (one process or thread per GPU), a 128-byte unique id handed to every rank out of band.
//
// librccl is opened lazily (dlopen) so that the attention / cache entry points carry no
// collective-library dependency; every failure goes through atoma_last_error().
#include "common.h"

#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace atoma {

// the slice of the NCCL/RCCL C API used here (rccl.h is the same ABI as nccl.h)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0 };
enum { NCCL_FLOAT16 = 6, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    });
    if (!r.handle || !r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        set_error("all-reduce: librccl.so could not be loaded");
        return nullptr;
    }
    return &r;
}

static bool check_nccl(Rccl *r, int code, const char *what) {
    if (code == 0) return true;
    set_error(std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "rccl error"));
    return false;
}

struct Comm {
    nccl_comm_t comm;
    int rank, world, device;
};

}  // namespace atoma

extern "C" {

int atoma_comm_unique_id(void *id128_out) {
    atoma::clear_error();
    atoma::Rccl *r = atoma::rccl();
    if (!r) return -1;
    atoma::nccl_unique_id id;
    if (!atoma::check_nccl(r, r->GetUniqueId(&id), "ncclGetUniqueId")) return -1;
    memcpy(id128_out, id.internal, 128);
    return 0;
}

int atoma_comm_init(void **comm_out, int rank, int world_size, const void *id128, int device) {
    atoma::clear_error();
    atoma::Rccl *r = atoma::rccl();
    if (!r) return -1;
    if (world_size < 1 || rank < 0 || rank >= world_size) { atoma::set_error("atoma_comm_init: bad rank/world_size"); return -1; }
    if (!atoma::check_hip(hipSetDevice(device), "hipSetDevice")) return -1;
    atoma::nccl_unique_id id;
    memcpy(id.internal, id128, 128);
    auto *c = new atoma::Comm{nullptr, rank, world_size, device};
    if (!atoma::check_nccl(r, r->CommInitRank(&c->comm, world_size, id, rank), "ncclCommInitRank")) {
        delete c;
        return -1;
    }
    *comm_out = c;
    return 0;
}

// multi_gpu.rs:141-179: out-of-place sum over the tensor-parallel ranks, input contiguous.
int atoma_allreduce_sum(void *comm, const void *in, void *out, int64_t count, int dtype, void *stream) {
    atoma::clear_error();
    atoma::Rccl *r = atoma::rccl();
    if (!r) return -1;
    if (!comm) { atoma::set_error("atoma_allreduce_sum: null communicator"); return -1; }
    int nd;
    switch (dtype) {
        case ATOMA_F16: nd = atoma::NCCL_FLOAT16; break;
        case ATOMA_BF16: nd = atoma::NCCL_BFLOAT16; break;
        case ATOMA_F32: nd = atoma::NCCL_FLOAT32; break;
        default: atoma::set_error("atoma_allreduce_sum: dtype must be f16, bf16 or f32"); return -1;
    }
    if (count <= 0) return 0;
    auto *c = static_cast<atoma::Comm *>(comm);
    return atoma::check_nccl(r, r->AllReduce(in, out, (size_t)count, nd, atoma::NCCL_SUM, c->comm,
                                             static_cast<hipStream_t>(stream)), "ncclAllReduce") ? 0 : -1;
}

int atoma_comm_destroy(void *comm) {
    atoma::clear_error();
    if (!comm) return 0;
    atoma::Rccl *r = atoma::rccl();
    auto *c = static_cast<atoma::Comm *>(comm);
    int rc = 0;
    if (r && !atoma::check_nccl(r, r->CommDestroy(c->comm), "ncclCommDestroy")) rc = -1;
    delete c;
    return rc;
}

}  // extern "C"
