// Linear layer at any batch: y[b][n] = sum_k x[b][k] * w[n][k]  (SURVEY.md 8f item 1).
//
// candle_nn::Linear::forward on [T, hidden] activations (/root/reference/models/src/llama.rs:269-271,311,364-365) is a
// cuBLAS GEMM in the reference.  A plain GEMM goes to the vendor library, hipBLASLt (bf16 / f16 inputs, fp32
// accumulation, one rounding); the hand-written weight-streaming kernel (linear_decode.hip) keeps the batches where it is
// level with the library and its fused epilogues save launches (1..4 rows; measured crossover: at 16 rows hipBLASLt
// is 15-30 % faster).  GEMM layout: in column-major terms y^T[N x B] = W[N x K] . x^T[K x B], i.e. the "TN" layout with
// A = the weight memory (K x N, ld = w_row_stride), B = the activation memory (K x B, ld = x_row_stride),
// C = the output memory (N x B, ld = y_row_stride).  hipBLASLt is loaded on first use (dlopen), one handle, one
// 64 MiB workspace and a cache of (problem -> algorithm) per device.
// Parity: unpinned (Candle / cuBLAS are not in the tree); the oracle is the f64-accumulated product rounded once.
#include "common.h"
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <tuple>
#include <vector>

extern "C" int atoma_linear_decode(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                                   int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream);

namespace atoma {

struct LtApi {
    void *lib = nullptr;
    decltype(&hipblasLtCreate) create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
    decltype(&hipblasLtMatmul) matmul = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) desc_destroy = nullptr;        // optional: a library without them only leaks on error paths
    decltype(&hipblasLtMatrixLayoutDestroy) layout_destroy = nullptr;
    decltype(&hipblasLtDestroy) destroy = nullptr;                       // optional: the per-stream handles go with the workspaces
    bool ok = false;
};

static LtApi &lt_api() {
    static LtApi api = [] {
        LtApi a;
        a.lib = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) a.lib = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) return a;
#define ATOMA_LT_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, #name))
        ATOMA_LT_SYM(create, hipblasLtCreate);
        ATOMA_LT_SYM(desc_create, hipblasLtMatmulDescCreate);
        ATOMA_LT_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
        ATOMA_LT_SYM(layout_create, hipblasLtMatrixLayoutCreate);
        ATOMA_LT_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
        ATOMA_LT_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
        ATOMA_LT_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
        ATOMA_LT_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
        ATOMA_LT_SYM(matmul, hipblasLtMatmul);
        ATOMA_LT_SYM(desc_destroy, hipblasLtMatmulDescDestroy);
        ATOMA_LT_SYM(layout_destroy, hipblasLtMatrixLayoutDestroy);
        ATOMA_LT_SYM(destroy, hipblasLtDestroy);
#undef ATOMA_LT_SYM
        a.ok = a.create && a.desc_create && a.desc_set && a.layout_create && a.pref_create && a.pref_set && a.pref_destroy &&
               a.heuristic && a.matmul;
        return a;
    }();
    return api;
}

struct GemmPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
};
// a plan that does not end up in the cache gives its descriptor and layouts back (a shape first seen during a
// capture leaked them on every call until an eager call tuned it; so did every error path of the plan builder)
struct PlanGuard {
    LtApi &api;
    GemmPlan &pl;
    bool keep = false;
    ~PlanGuard() {
        if (keep) return;
        if (pl.desc && api.desc_destroy) api.desc_destroy(pl.desc);
        for (hipblasLtMatrixLayout_t l : {pl.a, pl.b, pl.c})
            if (l && api.layout_destroy) api.layout_destroy(l);
    }
};
struct LtDevice {
    // One library handle PER STREAM, like the workspace: a handle owns device-side synchronisation state of its own (the split-accumulation
    // kernels' flags), and two products issued through ONE handle on two streams that run at the same time race on it -- found in round 5 with
    // eight tensor-parallel ranks on one device (tools/tp_step.py --virtual-ranks 8 --prefill 4096): wrong sums in one run, a GEMM that never
    // finished in the next.  One stream per device (the reference's one thread per GPU) never met it.
    std::map<hipStream_t, hipblasLtHandle_t> handles;
    std::map<hipStream_t, void *> workspaces;   // one per stream: two GEMMs on different streams may run concurrently
    std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t>, GemmPlan> plans;
};
static const size_t LT_WORKSPACE_BYTES = 64u << 20;
static const int LT_MAX_CANDIDATES = 128, LT_TIMED_REPS = 10;
// how many of the library's candidate algorithms are timed on first use of a problem (ATOMA_LINEAR_CANDIDATES, 1..128)
static const int LT_CANDIDATES = [] { const char *e = getenv("ATOMA_LINEAR_CANDIDATES"); const int n = e ? atoi(e) : 16; return n < 1 ? 1 : (n > LT_MAX_CANDIDATES ? LT_MAX_CANDIDATES : n); }();
// time the library's candidate algorithms on first use of a problem (ATOMA_LINEAR_AUTOTUNE=0: take the heuristic's first choice)
static const int linear_autotune = getenv("ATOMA_LINEAR_AUTOTUNE") ? atoi(getenv("ATOMA_LINEAR_AUTOTUNE")) : 1;
static std::mutex *g_lt_mu = new std::mutex;
static std::map<int, LtDevice> *g_lt_devices = new std::map<int, LtDevice>;

static const int lt_debug = getenv("ATOMA_LINEAR_DEBUG") ? atoi(getenv("ATOMA_LINEAR_DEBUG")) : 0;
#define LT_DBG(...) do { if (lt_debug) { fprintf(stderr, "[lt] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
static bool lt_ok(hipblasStatus_t st, const char *what) {
    if (st == HIPBLAS_STATUS_SUCCESS) return true;
    set_error(std::string("linear: hipBLASLt ") + what + " failed with status " + std::to_string((int)st));
    return false;
}

// y^T[n x batch] = W[n x k] . x^T[k x batch]
static int gemm_tn(const void *x, const void *w, void *y, int64_t batch, int64_t k, int64_t n, int64_t ldx, int64_t ldw, int64_t ldy,
                   int dtype, hipStream_t stream) {
    LtApi &api = lt_api();
    if (!api.ok) { set_error("linear: libhipblaslt.so could not be loaded (needed for batches above the weight-streaming kernel's range)"); return -1; }
    int dev = 0;
    if (!check_hip(hipGetDevice(&dev), "hipGetDevice")) return -1;
    // The lock covers the bookkeeping (handle, workspace, plan cache, first-use tuning), NOT the launch of a cached plan: a launch can block while
    // the stream's queue is full, and with several ranks driven from one process a full queue waits for an all-reduce that waits for a peer
    // whose thread would be waiting for this lock (round 5: the 80-layer prefill chunk of 8 virtual ranks timed out exactly so).
    std::unique_lock<std::mutex> lock(*g_lt_mu);
    LtDevice &d = (*g_lt_devices)[dev];
    hipblasLtHandle_t &handle = d.handles[stream];
    if (!handle) {
        if (!lt_ok(api.create(&handle), "hipblasLtCreate")) { handle = nullptr; return -1; }
    }
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &capturing);
    void *&lt_ws = d.workspaces[stream];
    if (!lt_ws) {
        if (capturing != hipStreamCaptureStatusNone) {
            set_error("linear: the hipBLASLt workspace of this stream cannot be allocated during hipGraph capture: run the same call once eagerly first");
            return -1;
        }
        if (!check_hip(hipMalloc(&lt_ws, LT_WORKSPACE_BYTES), "linear: workspace hipMalloc")) { lt_ws = nullptr; return -1; }
        // Zero it: recycled device memory holds other buffers' bytes, and some of the library's algorithms (stream-K style
        // split accumulation) keep synchronisation words in the workspace -- observed as a GEMM that never finishes when
        // the per-stream workspace came out of recycled memory.
        if (!check_hip(hipMemsetAsync(lt_ws, 0, LT_WORKSPACE_BYTES, stream), "linear: workspace memset")) return -1;
    }
    const auto key = std::make_tuple(dtype, batch, k, n, ldx, ldw, ldy);
    auto it = d.plans.find(key);
    LT_DBG("gemm batch=%lld k=%lld n=%lld stream=%p plan %s", (long long)batch, (long long)k, (long long)n, (void *)stream, it == d.plans.end() ? "MISS" : "hit");
    if (it == d.plans.end()) {
        GemmPlan pl;
        PlanGuard guard{api, pl};
        const hipDataType t = dtype == ATOMA_BF16 ? HIP_R_16BF : HIP_R_16F;
        if (!lt_ok(api.desc_create(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F), "MatmulDescCreate")) return -1;
        const int32_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
        if (!lt_ok(api.desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof op_t), "MatmulDescSetAttribute(TRANSA)")) return -1;
        if (!lt_ok(api.desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof op_n), "MatmulDescSetAttribute(TRANSB)")) return -1;
        if (!lt_ok(api.layout_create(&pl.a, t, (uint64_t)k, (uint64_t)n, ldw), "MatrixLayoutCreate(W)")) return -1;
        if (!lt_ok(api.layout_create(&pl.b, t, (uint64_t)k, (uint64_t)batch, ldx), "MatrixLayoutCreate(x)")) return -1;
        if (!lt_ok(api.layout_create(&pl.c, t, (uint64_t)n, (uint64_t)batch, ldy), "MatrixLayoutCreate(y)")) return -1;
        hipblasLtMatmulPreference_t pref = nullptr;
        if (!lt_ok(api.pref_create(&pref), "MatmulPreferenceCreate")) return -1;
        const uint64_t ws = LT_WORKSPACE_BYTES;
        bool ok = lt_ok(api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws), "MatmulPreferenceSetAttribute");
        // The heuristic's first choice is often not the fastest kernel for a skinny problem: take its candidates, time
        // each once on this problem (first use only; y is overwritten with the same product every time) and keep the best.
        // candidate 0 = the library's own single choice; the longer list it returns is ordered differently and need not contain it
        std::vector<hipblasLtMatmulHeuristicResult_t> res((size_t)LT_CANDIDATES + 1);
        int found = 0, more = 0;
        ok = ok && lt_ok(api.heuristic(handle, pl.desc, pl.a, pl.b, pl.c, pl.c, pref, 1, res.data(), &found), "MatmulAlgoGetHeuristic");
        if (ok && found == 1 && linear_autotune &&
            api.heuristic(handle, pl.desc, pl.a, pl.b, pl.c, pl.c, pref, LT_CANDIDATES, res.data() + 1, &more) == HIPBLAS_STATUS_SUCCESS)
            found += more;
        api.pref_destroy(pref);
        if (!ok) return -1;
        if (found < 1) { set_error("linear: hipBLASLt has no algorithm for this problem"); return -1; }
        int best = 0;
        // The timing runs synchronise with the stream (hipEventSynchronize) -- WITHOUT the lock (with several ranks driven from one
        // process a rank's stream may be waiting for an all-reduce whose peer's thread wants this lock: the deadlock the cached-plan path was
        // cured of in round 5).  The handle and the workspace are this stream's own and the plan is not in the cache yet; two threads that tune
        // the same problem at once both finish and the second insert below is dropped.
        hipblasLtHandle_t tune_handle = handle;
        void *tune_ws = lt_ws;
        lock.unlock();
        if (found > 1 && capturing == hipStreamCaptureStatusNone) {
            hipblasLtHandle_t handle = tune_handle;   // (shadows the map references: nothing below touches the maps)
            void *lt_ws = tune_ws;
            const float alpha = 1.f, beta = 0.f;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            auto run = [&](int c, int reps, float *ms) -> bool {          // `reps` back-to-back launches of candidate c
                LT_DBG("  time candidate %d of %d x %d", c, found, reps);
                bool good = true;
                (void)hipEventRecord(e0, stream);
                for (int rep = 0; rep < reps && good; ++rep)
                    good = api.matmul(handle, pl.desc, &alpha, w, pl.a, x, pl.b, &beta, y, pl.c, y, pl.c, &res[c].algo, lt_ws,
                                      LT_WORKSPACE_BYTES, stream) == HIPBLAS_STATUS_SUCCESS;
                (void)hipEventRecord(e1, stream);
                if (hipEventSynchronize(e1) != hipSuccess || !good) { (void)hipGetLastError(); return false; }
                (void)hipEventElapsedTime(ms, e0, e1);
                return true;
            };
            // clocks first: a cold GPU makes whichever candidate is timed first look slow
            float warm = 0.f, ms = 0.f;
            for (int i = 0; i < 40 && warm < 5.f; ++i) {
                if (!run(0, 10, &ms)) break;
                warm += ms;
            }
            // launches of less than ~30 us cannot be ranked reliably from the host (clock and dispatch jitter exceed the
            // differences between candidates, measured): keep the library's choice there; otherwise time windows of >= 1 ms
            float per_launch = 1e30f;
            if (run(0, 20, &ms)) per_launch = ms / 20.f;
            const int reps = per_launch < 0.03f ? 0 : std::max(LT_TIMED_REPS, std::min(50, (int)(1.f / per_launch)));
            std::vector<float> t((size_t)found, 1e30f);
            for (int round = 0; round < 3 && reps > 0; ++round)
                for (int c = 0; c < found; ++c) {
                    if (res[c].state != HIPBLAS_STATUS_SUCCESS || res[c].workspaceSize > LT_WORKSPACE_BYTES) continue;
                    if (round == 0 && !run(c, 2, &ms)) { res[c].state = HIPBLAS_STATUS_NOT_SUPPORTED; continue; }   // code load
                    if (run(c, reps, &ms)) t[(size_t)c] = std::min(t[(size_t)c], ms);
                    else res[c].state = HIPBLAS_STATUS_NOT_SUPPORTED;
                }
            for (int c = 1; c < found; ++c)
                if (t[(size_t)c] < t[(size_t)best]) best = c;
            if (best != 0 && t[(size_t)best] > 0.97f * t[0]) best = 0;   // keep the heuristic's choice unless another is clearly faster
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        lock.lock();
        pl.algo = res[best].algo;
        pl.workspace = res[best].workspaceSize;
        if (found > 1 && capturing != hipStreamCaptureStatusNone) {
            // first seen during capture: the candidates could not be timed -- use the heuristic's choice for this launch only,
            // so that a later eager call still tunes the problem
            const float alpha1 = 1.f, beta0 = 0.f;
            return lt_ok(api.matmul(handle, pl.desc, &alpha1, w, pl.a, x, pl.b, &beta0, y, pl.c, y, pl.c, &pl.algo, lt_ws,
                                    LT_WORKSPACE_BYTES, stream), "Matmul") ? 0 : -1;
        }
        auto ins = d.plans.emplace(key, pl);
        guard.keep = ins.second;                 // lost the race against another thread tuning the same problem: its plan stays, ours is released
        it = ins.first;
    }
    const GemmPlan pl = it->second;             // descriptors and layouts are immutable once cached; the handle and the workspace are this stream's own
    hipblasLtHandle_t h = handle;
    void *ws = lt_ws;
    lock.unlock();
    const float alpha = 1.f, beta = 0.f;
    if (!lt_ok(api.matmul(h, pl.desc, &alpha, w, pl.a, x, pl.b, &beta, y, pl.c, y, pl.c, &pl.algo, ws,
                          LT_WORKSPACE_BYTES, stream), "Matmul"))
        return -1;
    return 0;
}

// atoma_release_workspaces(): the per-stream 64 MiB workspaces of the vendor GEMM go with the library's own scratch
// (stream handles are recycled by the runtime, so a long-lived process that creates and destroys streams would otherwise
// accumulate one workspace per handle value ever seen)
int release_gemm_workspaces() {
    std::lock_guard<std::mutex> lock(*g_lt_mu);
    int rc = 0;
    for (auto &dev : *g_lt_devices) {
        for (auto &ws : dev.second.workspaces)
            if (ws.second && hipFree(ws.second) != hipSuccess) { (void)hipGetLastError(); rc = -1; }
        dev.second.workspaces.clear();
        // ... and the per-stream library handles (a process that creates and destroys streams would otherwise keep one per handle value ever seen);
        // the caller has made the streams idle (atoma_release_workspaces' contract)
        if (!dev.second.handles.empty()) {      // (only then has the library been loaded)
            LtApi &api = lt_api();
            for (auto &h : dev.second.handles)
                if (h.second && api.destroy) (void)api.destroy(h.second);
            dev.second.handles.clear();
        }
    }
    return rc;
}

// batches up to this many rows take the weight-streaming kernel (measured crossover, tools/bench_kernels.py linear)
static const int linear_stream_max_batch = getenv("ATOMA_LINEAR_STREAM_MAX_BATCH") ? atoi(getenv("ATOMA_LINEAR_STREAM_MAX_BATCH")) : 4;

}  // namespace atoma

extern "C" int atoma_linear(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                            int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream) {
    using namespace atoma;
    clear_error();
    if (batch >= 0 && batch <= std::min(linear_stream_max_batch, 64))
        return atoma_linear_decode(x, w, y, batch, in_features, out_features, x_row_stride, w_row_stride, y_row_stride, dtype, stream);
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("linear: dtype must be f16 or bf16"); return -1; }
    if (batch < 0 || batch > (1 << 20)) { set_error("linear: batch out of range"); return -1; }
    if (in_features <= 0 || in_features % 8 != 0) { set_error("linear: in_features must be a positive multiple of 8"); return -1; }
    if (out_features <= 0 || out_features % 8 != 0) { set_error("linear: out_features must be a positive multiple of 8"); return -1; }
    if (x_row_stride < in_features || w_row_stride < in_features || y_row_stride < out_features) { set_error("linear: row strides must cover a row"); return -1; }
    if (x_row_stride % 8 || w_row_stride % 8 || y_row_stride % 8) { set_error("linear: row strides must be multiples of 8 elements"); return -1; }
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15u) {
        set_error("linear: x, w and y must be 16-byte aligned");
        return -1;
    }
    return gemm_tn(x, w, y, batch, in_features, out_features, x_row_stride, w_row_stride, y_row_stride, dtype, static_cast<hipStream_t>(stream));
}
