// Error slot, device queries and the reference's host-side split heuristic.
#include "common.h"
#include "sync_ticket.h"

#include <algorithm>
#include <math.h>
#include <mutex>

namespace atoma {

static thread_local std::string g_error;

void set_error(const std::string &msg) { g_error = msg; }
void clear_error() { g_error.clear(); }
bool has_error() { return !g_error.empty(); }

bool set_decode_option(const std::string &name, int value);  // paged_decode.hip
bool set_generic_attn_option(const std::string &name, int value);  // attn_generic.hip
bool set_linear_tile_option(const std::string &name, int value);  // linear_tile.hip
bool set_linear_wide_option(const std::string &name, int value);  // linear_wide.hip
bool set_xgmi_option(const std::string &name, int value);         // allreduce_xgmi.hip

int device_num_cus() {
    // The reference queries cudaDeviceGetAttribute on EVERY attention call
    // (/root/reference/csrc/src/lib.rs:1545,2201-2233); cache it per device instead.
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// ---- library-owned scratch: one grow-only block per (device, stream) ---------------------------------------------
// Split-KV / balanced-mode partials of the decode kernel, the K-split partials of the projections.  The reference hands
// caller scratch to the kernel and may drop it right after the async launch (/root/reference/csrc/src/lib.rs:1023-1042,
// 1100); owning the scratch here removes that race.  Rules that make it safe for the reference's one-thread-per-GPU
// callers and for hipGraphs (SURVEY 8b says "callee is stateless"; this is the one piece of state, so it is explicit):
//   * a block that was ever handed out is NEVER freed behind the caller's back: when a later call needs more, a new block
//     of at least twice the size is allocated and the old one is RETIRED (kept alive), because a captured hipGraph may
//     have its address baked into kernel arguments;
//   * growth cannot happen during stream capture (hipMalloc is illegal there): the call fails with a message that names
//     atoma_warmup / atoma_reserve_workspace, which size the block up front;
//   * atoma_release_workspaces() frees everything (live and retired) when the caller knows no graph and no in-flight
//     work refers to them (engine shutdown, or after destroying its graphs).
struct Workspace {
    int device;
    hipStream_t stream;
    void *ptr;
    size_t bytes;
};
static std::vector<Workspace> g_ws;
static std::vector<void *> g_ws_retired;
static std::mutex *g_ws_mu = new std::mutex;

static bool stream_is_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

void *workspace(hipStream_t stream, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(*g_ws_mu);
    Workspace *slot = nullptr;
    for (auto &w : g_ws)
        if (w.device == dev && w.stream == stream) { slot = &w; break; }
    if (slot && slot->bytes >= bytes) return slot->ptr;
    if (stream_is_capturing(stream)) {
        set_error("the split / merge scratch of this stream must grow, which is not possible during hipGraph capture: call "
                  "atoma_warmup (or atoma_reserve_workspace) for this stream before capturing, or run the same call once eagerly first");
        return nullptr;
    }
    const size_t want = std::max(bytes, slot ? slot->bytes * 2 : (size_t)0);
    void *fresh = nullptr;
    if (!check_hip(hipMalloc(&fresh, want), "workspace hipMalloc")) return nullptr;
    if (!slot) {
        g_ws.push_back(Workspace{dev, stream, nullptr, 0});
        slot = &g_ws.back();
    } else if (slot->ptr) {
        g_ws_retired.push_back(slot->ptr);   // a captured graph may still point at it
    }
    slot->ptr = fresh;
    slot->bytes = want;
    return fresh;
}

// Arrival words of the kernels that merge their own split-K / split-KV pieces (the last workgroup to arrive at an item reduces it):
// SYNC_COUNTERS 64-bit words per (device, stream), allocated once.  A word is EPOCH-TAGGED (sync_ticket.h: epoch = the launch's AQL
// dispatch id): a launch ignores whatever an earlier launch left in it, nobody resets anything, a captured graph replays without a memset
// node, and an inconsistent episode (a graph replayed beside eager calls on its stream, metadata changed under a running launch) cannot
// corrupt a LATER launch -- the callee is stateless again (csrc/src/ffi.rs:3-102).  Kernels of one stream run one after the other and
// share the words.  The pointer is baked into captured graphs (like the scratch of `workspace`): atoma_release_workspaces frees the
// words and thereby invalidates every graph captured before it.  atoma_reset_sync_counters is kept for callers of round 4's contract;
// nothing requires it any more.
constexpr size_t SYNC_COUNTERS = 8192;
struct SyncWords { int device; hipStream_t stream; sync_word_t *ptr; };
static std::vector<SyncWords> g_sync;
sync_word_t *sync_counters(hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(*g_ws_mu);
    for (auto &w : g_sync)
        if (w.device == dev && w.stream == stream) return w.ptr;
    if (stream_is_capturing(stream)) {
        set_error("the arrival counters of this stream do not exist yet and cannot be allocated during hipGraph capture: call atoma_warmup "
                  "for this stream before capturing, or run the same call once eagerly first");
        return nullptr;
    }
    sync_word_t *ptr = nullptr;
    if (!check_hip(hipMalloc(reinterpret_cast<void **>(&ptr), SYNC_COUNTERS * sizeof(sync_word_t)), "sync counters hipMalloc")) return nullptr;
    if (!check_hip(hipMemset(ptr, 0, SYNC_COUNTERS * sizeof(sync_word_t)), "sync counters hipMemset") || !check_hip(hipDeviceSynchronize(), "sync counters")) {
        (void)hipFree(ptr);
        return nullptr;
    }
    g_sync.push_back(SyncWords{dev, stream, ptr});
    return ptr;
}

// 0 = ok (also when the stream has no counters yet)
int reset_sync_counters(hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    sync_word_t *ptr = nullptr;
    {
        std::lock_guard<std::mutex> lock(*g_ws_mu);
        for (auto &w : g_sync)
            if (w.device == dev && w.stream == stream) ptr = w.ptr;
    }
    if (!ptr) return 0;
    if (stream_is_capturing(stream)) { set_error("atoma_reset_sync_counters: not inside a hipGraph capture"); return -1; }
    return check_hip(hipMemsetAsync(ptr, 0, SYNC_COUNTERS * sizeof(sync_word_t), stream), "atoma_reset_sync_counters") ? 0 : -1;
}

size_t decode_workspace_bound(int max_b, int h, int h_k, int d, int max_seqlen_k);   // paged_decode.hip
const char *last_decode_kernel();                                                    // paged_decode.hip
bool linear_tile_prepare();                                                          // linear_tile.hip
bool linear_wide_prepare();                                                          // linear_wide.hip
bool prefill_asm_prepare();                                                          // prefill_asm.hip: the kernels' 160 KiB LDS opt-in on this device
size_t prefill_asm_workspace_bound(int64_t max_seqlen_q, int64_t seqs, int64_t heads);     // prefill_asm.hip: plan table of the persistent prefill kernel
int release_gemm_workspaces();                                                       // linear_gemm.hip

// /root/reference/csrc/src/lib.rs:2122-2167, f32 arithmetic as there.
int num_splits_heuristic(int64_t batch_nheads_mblocks, int64_t num_sms, int64_t num_n_blocks, int64_t max_splits) {
    if ((float)batch_nheads_mblocks >= 0.8f * (float)num_sms) return 1;
    max_splits = std::min(max_splits, std::min(num_sms, num_n_blocks));
    float max_eff = 0.f;
    std::vector<float> eff;
    eff.reserve((size_t)(max_splits > 0 ? max_splits : 0));
    auto eligible = [&](int64_t s) { return s == 1 || cdiv(num_n_blocks, s) != cdiv(num_n_blocks, s - 1); };
    for (int64_t s = 1; s <= max_splits; ++s) {
        if (!eligible(s)) { eff.push_back(0.f); continue; }
        const float n_waves = (float)(batch_nheads_mblocks * s) / (float)num_sms;
        const float e = n_waves / ceilf(n_waves);
        if (e > max_eff) max_eff = e;
        eff.push_back(e);
    }
    for (int64_t s = 1; s <= max_splits; ++s)
        if (eligible(s) && eff[(size_t)s - 1] >= 0.85f * max_eff) return (int)s;
    return 1;
}

}  // namespace atoma

extern "C" {

const char *atoma_last_error(void) { return atoma::g_error.c_str(); }
void atoma_clear_error(void) { atoma::clear_error(); }

int atoma_num_splits_heuristic(int64_t batch_nheads_mblocks, int64_t num_sms, int64_t num_n_blocks,
                               int64_t max_splits) {
    return atoma::num_splits_heuristic(batch_nheads_mblocks, num_sms, num_n_blocks, max_splits);
}

// /root/reference/csrc/src/lib.rs:2169-2199 with CUs in place of SMs.
int atoma_compute_num_splits(int64_t batch_size, int64_t num_heads, int64_t head_size, int64_t max_seqlen_k,
                             int64_t max_seqlen_q, int num_cus) {
    const int64_t block_n = head_size <= 64 ? 256 : (head_size <= 128 ? 128 : 64);
    const int64_t n_blocks = atoma::cdiv(max_seqlen_k, block_n);
    const int64_t m_blocks = atoma::cdiv(max_seqlen_q, 64);
    if (num_cus <= 0) num_cus = atoma::device_num_cus();
    return atoma::num_splits_heuristic(batch_size * num_heads * m_blocks, (int64_t)num_cus * 2, n_blocks, 128);
}

int atoma_set_option(const char *name, int value) {
    atoma::clear_error();
    if (name && atoma::set_decode_option(name, value)) return 0;
    if (name && atoma::set_linear_tile_option(name, value)) return 0;
    if (name && atoma::set_linear_wide_option(name, value)) return 0;
    if (name && atoma::set_xgmi_option(name, value)) return 0;
    if (name && atoma::set_generic_attn_option(name, value)) return 0;
    atoma::set_error(std::string("atoma_set_option: unknown option ") + (name ? name : "(null)"));
    return -1;
}

int atoma_reserve_workspace(void *stream, int64_t bytes) {
    atoma::clear_error();
    if (bytes < 0) { atoma::set_error("atoma_reserve_workspace: negative size"); return -1; }
    if (bytes == 0) return 0;
    return atoma::workspace(static_cast<hipStream_t>(stream), (size_t)bytes) ? 0 : -1;
}

int atoma_warmup(void *stream, int64_t max_batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim, int64_t max_seqlen_k,
                 int64_t extra_bytes) {
    atoma::clear_error();
    if (max_batch <= 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads || max_seqlen_k <= 0 || extra_bytes < 0) {
        atoma::set_error("atoma_warmup: invalid shape");
        return -1;
    }
    (void)atoma::device_num_cus();
    const size_t need = std::max(atoma::decode_workspace_bound((int)std::min<int64_t>(max_batch, 1 << 20), (int)num_heads, (int)num_kv_heads,
                                                               (int)head_dim, (int)std::min<int64_t>(max_seqlen_k, 1 << 30)),
                                 (size_t)extra_bytes);
    if (!atoma::sync_counters(static_cast<hipStream_t>(stream)) || !atoma::linear_tile_prepare() || !atoma::linear_wide_prepare() || !atoma::prefill_asm_prepare()) return -1;
    if (need == 0) return 0;
    return atoma::workspace(static_cast<hipStream_t>(stream), need) ? 0 : -1;
}

int atoma_warmup_prefill(void *stream, int64_t max_seqlen_q, int64_t max_seqs, int64_t num_heads) {
    atoma::clear_error();
    if (max_seqlen_q <= 0 || max_seqs <= 0 || num_heads <= 0) {
        atoma::set_error("atoma_warmup_prefill: invalid shape");
        return -1;
    }
    if (!atoma::prefill_asm_prepare()) return -1;
    return atoma::workspace(static_cast<hipStream_t>(stream), atoma::prefill_asm_workspace_bound(max_seqlen_q, max_seqs, num_heads)) ? 0 : -1;
}

namespace atoma {
__global__ void debug_epoch_kernel(unsigned long long *out) {
    if (threadIdx.x == 0) out[0] = sync_epoch();
}
}  // namespace atoma

int atoma_debug_launch_epoch(void *stream, void *epoch_out_device) {
    atoma::clear_error();
    if (!epoch_out_device) { atoma::set_error("atoma_debug_launch_epoch: null output"); return -1; }
    hipLaunchKernelGGL(atoma::debug_epoch_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long *>(epoch_out_device));
    return ATOMA_CHECK_LAUNCH("debug_epoch_kernel") ? 0 : -1;
}

int atoma_debug_sync_words(void *stream, void **words_out, int64_t *count_out) {
    atoma::clear_error();
    atoma::sync_word_t *p = atoma::sync_counters(static_cast<hipStream_t>(stream));
    if (!p) return -1;
    if (words_out) *words_out = p;
    if (count_out) *count_out = (int64_t)atoma::SYNC_COUNTERS;
    return 0;
}

// tests: the stream's current scratch block (split / merge partials, slabs, plan tables) -- to fill it with poison between launches
int atoma_debug_workspace(void *stream, void **ptr_out, int64_t *bytes_out) {
    atoma::clear_error();
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(*atoma::g_ws_mu);
    for (auto &w : atoma::g_ws)
        if (w.device == dev && w.stream == static_cast<hipStream_t>(stream)) {
            if (ptr_out) *ptr_out = w.ptr;
            if (bytes_out) *bytes_out = (int64_t)w.bytes;
            return 0;
        }
    if (ptr_out) *ptr_out = nullptr;
    if (bytes_out) *bytes_out = 0;
    return 0;
}

int atoma_reset_sync_counters(void *stream) {
    atoma::clear_error();
    return atoma::reset_sync_counters(static_cast<hipStream_t>(stream));
}

int atoma_release_workspaces(void) {
    atoma::clear_error();
    std::lock_guard<std::mutex> lock(*atoma::g_ws_mu);
    int rc = 0;
    for (auto &w : atoma::g_ws)
        if (w.ptr && hipFree(w.ptr) != hipSuccess) { (void)hipGetLastError(); rc = -1; }
    for (void *p : atoma::g_ws_retired)
        if (hipFree(p) != hipSuccess) { (void)hipGetLastError(); rc = -1; }
    for (auto &w : atoma::g_sync)
        if (w.ptr && hipFree(w.ptr) != hipSuccess) { (void)hipGetLastError(); rc = -1; }
    atoma::g_ws.clear();
    atoma::g_ws_retired.clear();
    atoma::g_sync.clear();
    if (atoma::release_gemm_workspaces() != 0) rc = -1;
    if (rc) atoma::set_error("atoma_release_workspaces: hipFree failed");
    return rc;
}

// Name and configuration of the decode kernel the dispatcher chose for the last decode call of this thread ("" before any)
const char *atoma_last_decode_kernel(void) { return atoma::last_decode_kernel(); }

int atoma_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int atoma_num_cus(int device) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
