// Tensor-parallel sum all-reduce: the communicator (RCCL bootstrap, as the reference's NCCL one) and the dispatch between
// RCCL's ncclAllReduce and the direct xGMI kernels of allreduce_xgmi.hip.
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
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
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
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
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

// Which engine atoma_allreduce_sum uses.  The direct xGMI kernels (allreduce_xgmi.hip) are bootstrapped over the RCCL
// communicator itself (an all-gather of the 128-byte staging handles), so the caller's bootstrap stays the reference's:
// one unique id, one Comm::from_rank per GPU.  Default: RCCL -- the direct path has only run with all ranks on ONE device
// so far (no multi-GPU box in the build pool); ATOMA_ALLREDUCE=auto|xgmi or atoma_comm_set_mode opt in.
enum { AR_RCCL = 0, AR_XGMI = 1, AR_AUTO = 2 };
struct Comm {
    nccl_comm_t comm;
    int rank, world, device;
    void *xgmi = nullptr;        // connected atoma_xgmi communicator, or null (see xgmi_note)
    int mode = AR_RCCL;
    int64_t xgmi_auto_max = 8 << 20;
    std::string xgmi_note;       // why the direct path is unavailable, if it is
    bool xgmi_tried = false;     // comm_setup_xgmi ran (it is collective: once per communicator, on every rank)
    bool xgmi_allowed = true;    // ATOMA_XGMI_SETUP != 0
    unsigned char *xchg = nullptr;   // 128 + world x 128 bytes of device memory for comm_setup_xgmi's two collectives, allocated by atoma_comm_init
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

int atoma_xgmi_create(void **out, int rank, int world_size, int device, int64_t max_bytes);
int atoma_xgmi_handle(void *xg, void *handle128_out);
int atoma_xgmi_connect(void *xg, const void *handles);
int atoma_xgmi_allreduce_sum(void *xg, const void *in, void *out, int64_t count, int dtype, void *stream);
int64_t atoma_xgmi_capacity(void *xg);
int atoma_xgmi_destroy(void *xg);

// Bring up the direct xGMI path of a communicator: staging region, handle all-gather over RCCL, peer mapping, and an
// agreement round (the path is used only if EVERY rank mapped every peer).  Never fails the communicator: the reason
// lands in atoma_comm_info().
static void comm_setup_xgmi(atoma::Rccl *r, atoma::Comm *c) {
    using namespace atoma;
    c->xgmi_tried = true;
    const char *mb = getenv("ATOMA_XGMI_MAX_BYTES");
    const int64_t cap = mb ? atoll(mb) : (int64_t)(8 << 20);
    c->xgmi_auto_max = cap;
    if (c->world > 8) { c->xgmi_note = "direct path supports up to 8 ranks"; return; }
    if (!r->AllGather) { c->xgmi_note = "ncclAllGather not found"; return; }
    // The two exchanges below are COLLECTIVE: every rank that got this far takes part in both, whatever happened to it locally --
    // a local failure travels in the payload (byte 127 of the handle / the agreement flag), never as a skipped call that would
    // leave the peers blocked inside RCCL.
    // The buffers of the exchange were allocated by atoma_comm_init (a rank without them never gets here: its init failed), so nothing
    // between here and the second collective can leave early.
    unsigned char *dsend = c->xchg, *drecv = c->xchg + 128;
    std::vector<unsigned char> all((size_t)c->world * 128, 0);
    const bool zeroed = hipMemset(dsend, 0, 128) == hipSuccess;
    if (!zeroed) (void)hipGetLastError();
    void *xg = nullptr;
    int ok = zeroed && atoma_xgmi_create(&xg, c->rank, c->world, c->device, cap) == 0;
    std::string why = ok ? "" : (zeroed ? atoma_last_error() : "could not clear the handle exchange buffer");
    unsigned char mine[128] = {0};
    if (ok && atoma_xgmi_handle(xg, mine) != 0) { ok = 0; why = atoma_last_error(); }
    // exchange the handles (and, in byte 127, whether this rank is still healthy) through RCCL; dsend holds zeros (= unhealthy)
    // until the upload succeeds
    mine[127] = ok ? 1 : 0;
    if (hipMemcpy(dsend, mine, 128, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); ok = 0; if (why.empty()) why = "handle upload failed"; }
    bool xok = r->AllGather(dsend, drecv, 128, 0 /* ncclInt8 */, c->comm, nullptr) == 0;
    xok = hipStreamSynchronize(nullptr) == hipSuccess && xok;
    xok = xok && hipMemcpy(all.data(), drecv, all.size(), hipMemcpyDeviceToHost) == hipSuccess;
    if (!xok) { (void)hipGetLastError(); ok = 0; if (why.empty()) why = "handle all-gather over RCCL failed"; }
    for (int q = 0; q < c->world && xok; ++q)
        if (!all[(size_t)q * 128 + 127]) { ok = 0; if (why.empty()) why = "rank " + std::to_string(q) + " could not create its staging region"; }
    for (int q = 0; q < c->world; ++q) all[(size_t)q * 128 + 127] = 0;
    if (ok && atoma_xgmi_connect(xg, all.data()) != 0) { ok = 0; why = atoma_last_error(); }
    // agreement: sum of the ranks' ok flags must equal the world size (dsend still holds a zero in its first word if the upload fails)
    float flag = ok ? 1.f : 0.f, total = 0.f;
    (void)hipMemset(dsend, 0, 4);
    (void)hipMemcpy(dsend, &flag, 4, hipMemcpyHostToDevice);
    bool aok = r->AllReduce(dsend, drecv, 1, NCCL_FLOAT32, NCCL_SUM, c->comm, nullptr) == 0;
    aok = hipStreamSynchronize(nullptr) == hipSuccess && aok;
    aok = aok && hipMemcpy(&total, drecv, 4, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipGetLastError();
    if (!aok || (int)(total + 0.5f) != c->world) {
        if (why.empty()) why = "another rank could not map its peers";
        if (xg) atoma_xgmi_destroy(xg);
        c->xgmi_note = why;
        clear_error();
        return;
    }
    c->xgmi = xg;
    clear_error();
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
    if (!atoma::check_hip(hipMalloc(reinterpret_cast<void **>(&c->xchg), 128 + (size_t)world_size * 128), "atoma_comm_init: exchange buffers")) {
        (void)r->CommDestroy(c->comm);
        delete c;
        return -1;
    }
    const char *m = getenv("ATOMA_ALLREDUCE");
    c->mode = (m && !strcmp(m, "xgmi")) ? atoma::AR_XGMI : ((m && !strcmp(m, "auto")) ? atoma::AR_AUTO : atoma::AR_RCCL);
    // The direct path (a 2 x cap uncached staging region + two RCCL collectives to exchange and agree on the handles) is built
    // only when it will be used: at init when the environment selects it (ATOMA_ALLREDUCE=xgmi|auto, or ATOMA_XGMI_SETUP=1),
    // otherwise on the first atoma_comm_set_mode(xgmi | auto) -- which every rank must call alike (it is collective then).
    // A plain RCCL communicator pays nothing.  ATOMA_XGMI_SETUP=0: never.
    const char *setup = getenv("ATOMA_XGMI_SETUP");
    c->xgmi_allowed = !(setup && atoi(setup) == 0);
    if (c->xgmi_allowed && (c->mode != atoma::AR_RCCL || (setup && atoi(setup) == 1))) comm_setup_xgmi(r, c);
    else c->xgmi_note = c->xgmi_allowed ? "not built yet (atoma_comm_set_mode builds it)" : "disabled by ATOMA_XGMI_SETUP=0";
    *comm_out = c;
    return 0;
}

// 0 = RCCL, 1 = direct xGMI kernels (error when unavailable), 2 = auto: direct for messages up to ATOMA_XGMI_MAX_BYTES
// that meet its alignment rules, RCCL otherwise.  All ranks must choose the same mode.
int atoma_comm_set_mode(void *comm, int mode) {
    atoma::clear_error();
    auto *c = static_cast<atoma::Comm *>(comm);
    if (!c || mode < 0 || mode > 2) { atoma::set_error("atoma_comm_set_mode: bad argument"); return -1; }
    if (mode != atoma::AR_RCCL && !c->xgmi && !c->xgmi_tried && c->xgmi_allowed) {   // lazy, collective build of the direct path
        atoma::Rccl *r = atoma::rccl();
        if (!r) return -1;
        if (!atoma::check_hip(hipSetDevice(c->device), "hipSetDevice")) return -1;
        comm_setup_xgmi(r, c);
    }
    if (mode == atoma::AR_XGMI && !c->xgmi) { atoma::set_error("atoma_comm_set_mode: the direct xGMI path is unavailable: " + c->xgmi_note); return -1; }
    c->mode = mode;
    return 0;
}

// "xgmi: ready" or "xgmi: unavailable (<reason>)"; valid until the communicator is destroyed
const char *atoma_comm_info(void *comm) {
    auto *c = static_cast<atoma::Comm *>(comm);
    if (!c) return "";
    if (c->xgmi) c->xgmi_note = "xgmi: ready";
    else if (c->xgmi_note.rfind("xgmi:", 0) != 0) c->xgmi_note = "xgmi: unavailable (" + c->xgmi_note + ")";
    return c->xgmi_note.c_str();
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
    if (c->mode != atoma::AR_RCCL) {
        const int64_t bytes = count * (dtype == ATOMA_F32 ? 4 : 2);
        const bool fits = c->xgmi && bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
        if (c->mode == atoma::AR_XGMI || (fits && bytes <= c->xgmi_auto_max)) {
            if (!c->xgmi) { atoma::set_error("atoma_allreduce_sum: the direct xGMI path is unavailable: " + c->xgmi_note); return -1; }
            return atoma_xgmi_allreduce_sum(c->xgmi, in, out, count, dtype, stream);
        }
    }
    return atoma::check_nccl(r, r->AllReduce(in, out, (size_t)count, nd, atoma::NCCL_SUM, c->comm,
                                             static_cast<hipStream_t>(stream)), "ncclAllReduce") ? 0 : -1;
}

// llama_nccl.rs:139 -> llama.rs:404,408 (and :195 -> :409, next layer's :402): x_out = residual + allreduce(in), norm_out = RMSNorm(x_out) * weight.
// Direct engine: ONE launch (atoma_xgmi_allreduce_add_rms_norm); RCCL engine: ncclAllReduce into norm_out (scratch), then atoma_add_rms_norm.
// Both engines end with the bits of "all-reduce, then atoma_add_rms_norm" of that engine.  in / x_out contiguous [rows, hidden].
int atoma_allreduce_add_rms_norm(void *comm, const void *in, const void *residual, const void *weight, void *x_out, void *norm_out, int64_t rows,
                                 int64_t hidden, float eps, int dtype, void *stream) {
    atoma::clear_error();
    if (!comm) { atoma::set_error("atoma_allreduce_add_rms_norm: null communicator"); return -1; }
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { atoma::set_error("atoma_allreduce_add_rms_norm: dtype must be f16 or bf16"); return -1; }
    if (rows <= 0) return 0;
    auto *c = static_cast<atoma::Comm *>(comm);
    const int64_t bytes = rows * hidden * 2;
    // aliasing contract (atoma_hip.h): x_out may be `residual` (the natural in-place x += allreduce(partial)) and may be `in`; norm_out is
    // written last and must not overlap `in`, `residual` or x_out
    auto overlaps = [&](const void *a, const void *b) {
        const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
        return a && b && x < y + (uintptr_t)bytes && y < x + (uintptr_t)bytes;
    };
    if (overlaps(norm_out, in) || overlaps(norm_out, residual) || overlaps(norm_out, x_out)) {
        atoma::set_error("atoma_allreduce_add_rms_norm: norm_out must not overlap in, residual or x_out");
        return -1;
    }
    // the direct engine moves 16-byte pieces: anything it cannot take in AUTO mode goes to RCCL, as in atoma_allreduce_sum
    const bool fits = c->xgmi && bytes % 16 == 0 && bytes <= atoma_xgmi_capacity(c->xgmi) &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(weight) |
                        reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(norm_out)) & 15u) == 0;
    if (c->mode == atoma::AR_XGMI || (c->mode != atoma::AR_RCCL && fits && bytes <= c->xgmi_auto_max)) {
        if (!c->xgmi) { atoma::set_error("atoma_allreduce_add_rms_norm: the direct xGMI path is unavailable: " + c->xgmi_note); return -1; }
        return atoma_xgmi_allreduce_add_rms_norm(c->xgmi, in, residual, weight, x_out, norm_out, rows, hidden, hidden, hidden, hidden, eps, dtype, 0, stream);
    }
    // RCCL engine: the sum lands in norm_out (scratch until the norm overwrites it row by row, each row read before it is written), so that
    // `residual` is still intact when x_out == residual (reducing into x_out first silently turned x += sum into x = 2 * sum)
    if (atoma_allreduce_sum(comm, in, norm_out, rows * hidden, dtype, stream) != 0) return -1;
    return atoma_add_rms_norm(residual, norm_out, weight, x_out, norm_out, rows, hidden, hidden, hidden, hidden, hidden, eps, dtype, stream);
}

int atoma_comm_destroy(void *comm) {
    atoma::clear_error();
    if (!comm) return 0;
    atoma::Rccl *r = atoma::rccl();
    auto *c = static_cast<atoma::Comm *>(comm);
    int rc = 0;
    if (c->xgmi) atoma_xgmi_destroy(c->xgmi);
    if (c->xchg) (void)hipFree(c->xchg);
    if (r && !atoma::check_nccl(r, r->CommDestroy(c->comm), "ncclCommDestroy")) rc = -1;
    delete c;
    return rc;
}

}  // extern "C"
