// KV-cache maintenance kernels for gfx950: reshape_and_cache_flash, copy_blocks, swap_blocks.
//
// All three are pure byte movement and are bit-exact by construction.  They replace
//   /root/reference/csrc/kernels/cache_manager.cu:15-37   (copy_blocks_kernel, 2-byte scalar copies)
//   /root/reference/csrc/kernels/cache_manager.cu:139-170 (reshape_and_cache_flash_kernel, 2-byte scalar)
//   /root/reference/csrc/src/cache_manager.rs:18-128 + csrc/src/ops.rs (one memcpy per page)
// with 16-byte-per-lane coalesced moves (a K or V row of one token is num_heads*head_size*2 B
// contiguous, a page is block_size such rows), falling back to 2-byte moves only when the
// caller's pointers/strides are not 16-byte aligned.  HBM-bound: bytes = 2x what is moved.
#include "common.h"
#include <algorithm>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <emmintrin.h>
#include <sched.h>

namespace atoma {

// ------------------------------------------------------------------------------------------
// reshape_and_cache_flash: one workgroup per token, K row then V row, 16 B per lane.
// ------------------------------------------------------------------------------------------
template <bool VEC16>
__global__ void __launch_bounds__(256)
reshape_and_cache_flash_kernel(const uint16_t *__restrict__ key, const uint16_t *__restrict__ value,
                               uint16_t *__restrict__ key_cache, uint16_t *__restrict__ value_cache,
                               const int64_t *__restrict__ slot_mapping, int64_t block_stride,
                               int64_t key_stride, int64_t value_stride, int n, int block_size) {
    const int64_t token = blockIdx.x;
    const int64_t slot = slot_mapping[token];
    if (slot < 0) return;  // padding token (cache_manager.cu:152-155)
    const int64_t dst = (slot / block_size) * block_stride + (slot % block_size) * (int64_t)n;
    const uint16_t *ksrc = key + token * key_stride;
    const uint16_t *vsrc = value + token * value_stride;
    if constexpr (VEC16) {
        const int nv = n >> 3;  // 16-byte vectors per row
        const uint4 *k4 = reinterpret_cast<const uint4 *>(ksrc);
        const uint4 *v4 = reinterpret_cast<const uint4 *>(vsrc);
        uint4 *kd = reinterpret_cast<uint4 *>(key_cache + dst);
        uint4 *vd = reinterpret_cast<uint4 *>(value_cache + dst);
        for (int i = threadIdx.x; i < 2 * nv; i += blockDim.x) {
            if (i < nv) kd[i] = k4[i];
            else vd[i - nv] = v4[i - nv];
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            key_cache[dst + i] = ksrc[i];
            value_cache[dst + i] = vsrc[i];
        }
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

void launch_reshape_and_cache_flash(const void *key, const void *value, void *key_cache, void *value_cache,
                                    const int64_t *slot_mapping, int64_t block_stride, int64_t num_tokens,
                                    int64_t num_heads, int64_t head_size, int64_t block_size,
                                    int64_t key_stride, int64_t value_stride, hipStream_t stream) {
    if (num_tokens <= 0) return;
    const int64_t n = num_heads * head_size;
    const bool vec = (n % 8 == 0) && (key_stride % 8 == 0) && (value_stride % 8 == 0) && (block_stride % 8 == 0) &&
                     aligned16(key) && aligned16(value) && aligned16(key_cache) && aligned16(value_cache);
    const int64_t work = vec ? 2 * (n / 8) : n;
    int threads = (int)((work + 63) / 64 * 64);
    threads = threads > 256 ? 256 : (threads < 64 ? 64 : threads);
    dim3 grid((unsigned)num_tokens);
    auto k16 = static_cast<const uint16_t *>(key);
    auto v16 = static_cast<const uint16_t *>(value);
    auto kc16 = static_cast<uint16_t *>(key_cache);
    auto vc16 = static_cast<uint16_t *>(value_cache);
    if (vec)
        hipLaunchKernelGGL(reshape_and_cache_flash_kernel<true>, grid, dim3(threads), 0, stream, k16, v16, kc16,
                           vc16, slot_mapping, block_stride, key_stride, value_stride, (int)n, (int)block_size);
    else
        hipLaunchKernelGGL(reshape_and_cache_flash_kernel<false>, grid, dim3(threads), 0, stream, k16, v16, kc16,
                           vc16, slot_mapping, block_stride, key_stride, value_stride, (int)n, (int)block_size);
    ATOMA_CHECK_LAUNCH("reshape_and_cache_flash");
}

// ------------------------------------------------------------------------------------------
// copy_blocks: grid (layer, pair, span); each workgroup moves one 16 KiB span of the K page
// and of the V page (4 x 16 B per lane, all loads issued before the stores).
// ------------------------------------------------------------------------------------------
constexpr int COPY_THREADS = 256;
constexpr int COPY_UNROLL = 4;
constexpr int COPY_SPAN_VECS = COPY_THREADS * COPY_UNROLL;  // 1024 x 16 B = 16 KiB

__device__ __forceinline__ void copy_span16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t nvec,
                                            int64_t span) {
    const int64_t base = span * COPY_SPAN_VECS + threadIdx.x;
    uint4 r[COPY_UNROLL];
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * COPY_THREADS;
        if (i < nvec) r[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * COPY_THREADS;
        if (i < nvec) dst[i] = r[u];
    }
}

// A span of the K page and the same span of the V page together: all eight 16-byte loads of a lane leave before its first store (32 KiB in
// flight per workgroup), every byte is touched once -> non-temporal on both sides (round 5: 4.85 -> see profiles/r05 K5; the copied pages are
// read by the NEXT decode step from HBM anyway).
typedef unsigned int copy_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void copy_span16_kv(const copy_u32x4 *__restrict__ ks, copy_u32x4 *__restrict__ kd, const copy_u32x4 *__restrict__ vs,
                                               copy_u32x4 *__restrict__ vd, int64_t nvec, int64_t span) {
    const int64_t base = span * COPY_SPAN_VECS + threadIdx.x;
    copy_u32x4 rk[COPY_UNROLL], rv[COPY_UNROLL];
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * COPY_THREADS;
        if (i < nvec) { rk[u] = __builtin_nontemporal_load(ks + i); rv[u] = __builtin_nontemporal_load(vs + i); }
    }
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * COPY_THREADS;
        if (i < nvec) { __builtin_nontemporal_store(rk[u], kd + i); __builtin_nontemporal_store(rv[u], vd + i); }
    }
}

__global__ void __launch_bounds__(COPY_THREADS)
copy_blocks_kernel(const int64_t *__restrict__ key_cache_ptrs, const int64_t *__restrict__ value_cache_ptrs,
                   const int64_t *__restrict__ block_mapping, int64_t numel_per_block) {
    const int layer = blockIdx.x, pair = blockIdx.y;
    const int64_t span = blockIdx.z;
    uint16_t *kc = reinterpret_cast<uint16_t *>(key_cache_ptrs[layer]);
    uint16_t *vc = reinterpret_cast<uint16_t *>(value_cache_ptrs[layer]);
    const int64_t src = block_mapping[2 * pair] * numel_per_block;
    const int64_t dst = block_mapping[2 * pair + 1] * numel_per_block;
    const bool vec = (numel_per_block % 8 == 0) && ((reinterpret_cast<uintptr_t>(kc) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(vc) & 15u) == 0);
    if (vec) {  // workgroup-uniform branch
        const int64_t nvec = numel_per_block >> 3;
        copy_span16_kv(reinterpret_cast<const copy_u32x4 *>(kc + src), reinterpret_cast<copy_u32x4 *>(kc + dst),
                       reinterpret_cast<const copy_u32x4 *>(vc + src), reinterpret_cast<copy_u32x4 *>(vc + dst), nvec, span);
    } else {
        const int64_t lo = span * COPY_SPAN_VECS * 8;
        int64_t hi = lo + (int64_t)COPY_SPAN_VECS * 8;
        if (hi > numel_per_block) hi = numel_per_block;
        for (int64_t i = lo + threadIdx.x; i < hi; i += COPY_THREADS) {
            kc[dst + i] = kc[src + i];
            vc[dst + i] = vc[src + i];
        }
    }
}

void launch_copy_blocks(const void *key_cache_ptrs, const void *value_cache_ptrs, const void *block_mapping,
                        int64_t num_layers, int64_t num_pairs, int64_t numel_per_block, hipStream_t stream) {
    if (num_layers <= 0 || num_pairs <= 0 || numel_per_block <= 0) return;
    const int64_t spans = cdiv(numel_per_block, (int64_t)COPY_SPAN_VECS * 8);
    // gridDim.y/z are limited to 65535: walk the pair list in slices if it is longer.
    for (int64_t p0 = 0; p0 < num_pairs; p0 += 65535) {
        const int64_t np = (num_pairs - p0) < 65535 ? (num_pairs - p0) : 65535;
        dim3 grid((unsigned)num_layers, (unsigned)np, (unsigned)spans);
        hipLaunchKernelGGL(copy_blocks_kernel, grid, dim3(COPY_THREADS), 0, stream,
                           static_cast<const int64_t *>(key_cache_ptrs),
                           static_cast<const int64_t *>(value_cache_ptrs),
                           static_cast<const int64_t *>(block_mapping) + 2 * p0, numel_per_block);
        if (!ATOMA_CHECK_LAUNCH("copy_blocks")) return;
    }
}

// ------------------------------------------------------------------------------------------
// swap_blocks: one gather/scatter launch moves up to SWAP_MAX_PAIRS pages of up to
// SWAP_MAX_TENSORS tensors (every layer's K and V).  Page numbers and base pointers travel
// in the kernel arguments, so no device-side mapping upload is needed.  The same kernel
// serves gpu->gpu and, when the host side is pinned + device-addressable, cpu<->gpu over
// PCIe (the GPU reads/writes host memory directly, 16 B per lane).
// ------------------------------------------------------------------------------------------
constexpr int SWAP_MAX_PAIRS = 192;
constexpr int SWAP_MAX_TENSORS = 64;
struct SwapArgs {
    const char *src[SWAP_MAX_TENSORS];
    char *dst[SWAP_MAX_TENSORS];
    int32_t pairs[SWAP_MAX_PAIRS][2];
};
static_assert(sizeof(SwapArgs) <= 3072, "kernarg budget");

__global__ void __launch_bounds__(COPY_THREADS) swap_blocks_kernel(SwapArgs a, int64_t block_bytes) {
    const int pair = blockIdx.x, tensor = blockIdx.y;
    const int64_t span = blockIdx.z;
    const char *src = a.src[tensor] + (int64_t)a.pairs[pair][0] * block_bytes;
    char *dst = a.dst[tensor] + (int64_t)a.pairs[pair][1] * block_bytes;
    const bool vec = (block_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    if (vec) {
        copy_span16(reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst), block_bytes >> 4, span);
    } else {
        const int64_t lo = span * COPY_SPAN_VECS * 16;
        int64_t hi = lo + (int64_t)COPY_SPAN_VECS * 16;
        if (hi > block_bytes) hi = block_bytes;
        for (int64_t i = lo + threadIdx.x; i < hi; i += COPY_THREADS) dst[i] = src[i];
    }
}

// Device-visible alias of a host pointer, or nullptr when the memory is pageable.
static void *device_alias(const void *host_ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host_ptr) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error for us
        return nullptr;
    }
    if (attr.type != hipMemoryTypeHost) return nullptr;
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void *>(host_ptr), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}

// Pageable host memory (the reference's CPU cache is ordinary Candle tensors: backends/vllm/src/worker.rs:570-598; its swap is one
// cudaMemcpyAsync per page, csrc/src/ops.rs:158-166,205-216 -- which the runtime stages page by page: 3 GB/s here).  Pages travel
// through a pinned, device-addressable bounce ring instead: BOUNCE_SLOTS slots of 16 MiB per device; the gather / scatter kernel moves a
// slot's worth of pages -- of several tensors at once: a chunk is (tensors x pages) -- between the cache and a slot in one launch, a
// persistent team of host threads copies between the slot and the caller's pages (non-temporal stores on the way out: a 32 KiB page is
// below the size from which memcpy stops reading the destination lines first), and the device runs up to BOUNCE_SLOTS - 1 chunks ahead of
// the host copies.  Round 6 (the unchanged reference hands over pageable tensors): 26-36 -> see DESIGN.md 4.3 (round 5:
// two 8 MiB slots, one tensor per chunk, threads spawned per chunk).  Like a pageable hipMemcpy, the call returns when the host side is
// done (GPU -> CPU: the data is in the caller's pages; CPU -> GPU: the caller's pages have been read); the device side stays ordered on
// `stream`.
namespace {
constexpr size_t BOUNCE_SLOT = 16u << 20;
constexpr int BOUNCE_SLOTS = 4;
struct Bounce {
    int device = -1;
    std::mutex mu;                               // one swap at a time per DEVICE (the slots are the device's): other devices' threads do not wait
    char *host[BOUNCE_SLOTS] = {};
    char *dev[BOUNCE_SLOTS] = {};
    hipEvent_t done[BOUNCE_SLOTS] = {};          // the kernel that last touched the slot
    bool busy[BOUNCE_SLOTS] = {};
};
std::mutex g_bounce_mu;                          // guards the list only
std::vector<std::unique_ptr<Bounce>> g_bounce;

Bounce *bounce_for_device() {
    int dev = 0;
    if (!check_hip(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
    std::lock_guard<std::mutex> lock(g_bounce_mu);
    for (auto &b : g_bounce)
        if (b->device == dev) return b.get();
    auto b = std::make_unique<Bounce>();
    b->device = dev;
    bool ok = true;
    for (int i = 0; i < BOUNCE_SLOTS && ok; ++i) {
        void *h = nullptr, *d = nullptr;
        ok = check_hip(hipHostMalloc(&h, BOUNCE_SLOT, hipHostMallocMapped | hipHostMallocPortable), "swap_blocks bounce hipHostMalloc");
        b->host[i] = static_cast<char *>(h);
        ok = ok && check_hip(hipHostGetDevicePointer(&d, h, 0), "swap_blocks bounce alias");
        b->dev[i] = static_cast<char *>(d);
        ok = ok && check_hip(hipEventCreateWithFlags(&b->done[i], hipEventDisableTiming), "swap_blocks bounce event");
    }
    if (!ok) {                                   // nothing of a half-built ring is kept (a retry starts from scratch and leaks nothing)
        for (int i = 0; i < BOUNCE_SLOTS; ++i) {
            if (b->done[i]) (void)hipEventDestroy(b->done[i]);
            if (b->host[i]) (void)hipHostFree(b->host[i]);
        }
        return nullptr;
    }
    g_bounce.push_back(std::move(b));
    return g_bounce.back().get();
}

// 16-byte-aligned page copy with non-temporal stores (SSE2: the x86-64 baseline)
inline void copy_page_stream(char *dst, const char *src, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15u) != 0) { memcpy(dst, src, bytes); return; }
    const __m128i *s = reinterpret_cast<const __m128i *>(src);
    __m128i *d = reinterpret_cast<__m128i *>(dst);
    size_t n = bytes >> 4, i = 0;
    for (; i + 4 <= n; i += 4) {
        const __m128i a = _mm_load_si128(s + i), b = _mm_load_si128(s + i + 1), c = _mm_load_si128(s + i + 2), e = _mm_load_si128(s + i + 3);
        _mm_stream_si128(d + i, a); _mm_stream_si128(d + i + 1, b); _mm_stream_si128(d + i + 2, c); _mm_stream_si128(d + i + 3, e);
    }
    for (; i < n; ++i) _mm_stream_si128(d + i, _mm_load_si128(s + i));
}

// A persistent team of copy threads (created on first use, ATOMA_SWAP_THREADS, default min(12, the CPUs this process may run on)): a job is
// a list of page copies handed out in batches through an atomic cursor; the calling thread works too.
class CopyTeam {
  public:
    struct Job { char *slot; char *const *tensors; const int64_t *pages; int64_t nt, np, block_bytes; bool to_slot; };
    static CopyTeam &get() { static CopyTeam *t = new CopyTeam; return *t; }      // (never destroyed: the threads idle on a condition variable)
    void run(const Job &j) {
        std::lock_guard<std::mutex> whole(run_mu_);        // one job at a time: swaps of DIFFERENT devices (one host thread per GPU) share the team
        std::unique_lock<std::mutex> lk(mu_);
        job_ = j;
        next_.store(0);
        total_ = j.nt * j.np;
        pending_ = (int)workers_.size();
        ++gen_;
        lk.unlock();
        cv_.notify_all();
        work();
        _mm_sfence();
        lk.lock();
        done_cv_.wait(lk, [&] { return pending_ == 0; });
    }

  private:
    CopyTeam() {
        const char *e = getenv("ATOMA_SWAP_THREADS");
        cpu_set_t set;
        int cpus = 8;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
        int n = e ? atoi(e) : std::min(12, std::max(1, cpus));
        n = std::max(1, std::min(n, 32));
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
        for (auto &w : workers_) w.detach();
    }
    void work() {
        const int64_t batch = 4;
        for (;;) {
            const int64_t lo = next_.fetch_add(batch);
            if (lo >= total_) return;
            const int64_t hi = std::min(lo + batch, total_);
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t t = i / job_.np, k = i - t * job_.np;
                char *pg = job_.tensors[t] + job_.pages[k] * job_.block_bytes, *sl = job_.slot + i * job_.block_bytes;
                if (job_.to_slot) memcpy(sl, pg, (size_t)job_.block_bytes);
                else copy_page_stream(pg, sl, (size_t)job_.block_bytes);
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            lk.unlock();
            work();
            _mm_sfence();
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    Job job_{};
    std::atomic<int64_t> next_{0};
    int64_t total_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
};
}  // namespace

static int swap_blocks_pageable(const void *const *srcs, void *const *dsts, int64_t num_tensors, const int64_t *mapping, int64_t num_pairs,
                                int64_t block_bytes, int kind, hipStream_t stream) {
    if ((size_t)block_bytes > BOUNCE_SLOT) {     // pages larger than a slot: the plain per-page copy
        const hipMemcpyKind mk = kind == ATOMA_SWAP_CPU_TO_GPU ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        for (int64_t t = 0; t < num_tensors; ++t)
            for (int64_t p = 0; p < num_pairs; ++p) {
                const char *sp = static_cast<const char *>(srcs[t]) + mapping[2 * p] * block_bytes;
                char *dp = static_cast<char *>(dsts[t]) + mapping[2 * p + 1] * block_bytes;
                if (!check_hip(hipMemcpyAsync(dp, sp, (size_t)block_bytes, mk, stream), "swap_blocks memcpy")) return -1;
            }
        return 0;
    }
    // The bounce route BLOCKS the host (like a pageable hipMemcpy) and waits on events: it cannot be recorded into a graph
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("swap_blocks: pageable host tensors cannot be swapped inside a hipGraph capture (the call waits for the device); pin them "
                  "(atoma_host_register) or swap outside the capture");
        return -1;
    }
    (void)hipGetLastError();
    Bounce *bn = bounce_for_device();
    if (!bn) return -1;
    std::lock_guard<std::mutex> lock(bn->mu);
    CopyTeam &team = CopyTeam::get();
    const bool out = kind == ATOMA_SWAP_GPU_TO_CPU;
    // a chunk = nt tensors x np pages (the same page list for every tensor), laid out [tensor][page] in its slot
    const int64_t slot_pages = (int64_t)(BOUNCE_SLOT / (size_t)block_bytes);
    const int64_t np_max = std::min<int64_t>(std::min<int64_t>(slot_pages, SWAP_MAX_PAIRS), num_pairs);
    const int64_t nt_max = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(slot_pages / np_max, SWAP_MAX_TENSORS), num_tensors));
    const int64_t spans = cdiv(block_bytes, (int64_t)COPY_SPAN_VECS * 16);
    struct Chunk { int64_t t0, nt, p0, np; };
    Chunk in_slot[BOUNCE_SLOTS] = {};
    std::vector<int64_t> host_pages((size_t)np_max);
    std::vector<char *> host_tensors((size_t)nt_max);
    auto host_side = [&](int slot, const Chunk &c) {       // copy a chunk between its slot and the caller's pages (the whole team)
        for (int64_t k = 0; k < c.np; ++k) host_pages[(size_t)k] = mapping[2 * (c.p0 + k) + (out ? 1 : 0)];
        for (int64_t t = 0; t < c.nt; ++t) host_tensors[(size_t)t] = out ? static_cast<char *>(dsts[c.t0 + t]) : const_cast<char *>(static_cast<const char *>(srcs[c.t0 + t]));
        team.run(CopyTeam::Job{bn->host[slot], host_tensors.data(), host_pages.data(), c.nt, c.np, block_bytes, !out});
    };
    auto retire = [&](int slot) -> bool {                  // the slot's last kernel is done; GPU -> CPU: its pages go out to the caller
        if (!bn->busy[slot]) return true;
        if (!check_hip(hipEventSynchronize(bn->done[slot]), "swap_blocks bounce wait")) return false;
        bn->busy[slot] = false;
        if (out && in_slot[slot].np) { host_side(slot, in_slot[slot]); in_slot[slot].np = 0; }
        return true;
    };
    int slot = 0;
    for (int64_t t0 = 0; t0 < num_tensors; t0 += nt_max)
        for (int64_t p0 = 0; p0 < num_pairs; p0 += np_max, slot = (slot + 1) % BOUNCE_SLOTS) {
            const Chunk c{t0, std::min(nt_max, num_tensors - t0), p0, std::min(np_max, num_pairs - p0)};
            if (!retire(slot)) return -1;         // (a slot left busy by an earlier call holds no pending host copy: in_slot starts empty)
            if (!out) host_side(slot, c);         // CPU -> GPU: fill the slot while the device scatters the chunks before this one
            SwapArgs a;
            for (int64_t k = 0; k < c.np; ++k) {
                a.pairs[k][0] = out ? (int32_t)mapping[2 * (p0 + k)] : (int32_t)k;
                a.pairs[k][1] = out ? (int32_t)k : (int32_t)mapping[2 * (p0 + k) + 1];
            }
            for (int64_t t = 0; t < c.nt; ++t) {
                char *in_slot_t = bn->dev[slot] + t * c.np * block_bytes;
                a.src[t] = out ? static_cast<const char *>(srcs[t0 + t]) : in_slot_t;
                a.dst[t] = out ? in_slot_t : static_cast<char *>(dsts[t0 + t]);
            }
            hipLaunchKernelGGL(swap_blocks_kernel, dim3((unsigned)c.np, (unsigned)c.nt, (unsigned)spans), dim3(COPY_THREADS), 0, stream, a, block_bytes);
            if (!ATOMA_CHECK_LAUNCH("swap_blocks (bounce)")) return -1;
            if (!check_hip(hipEventRecord(bn->done[slot], stream), "swap_blocks bounce record")) return -1;
            bn->busy[slot] = true;
            if (out) in_slot[slot] = c;           // GPU -> CPU: copied out when the slot comes round again (the device is BOUNCE_SLOTS - 1 chunks ahead) or at the end
        }
    if (out)
        for (int i = 0; i < BOUNCE_SLOTS; ++i, slot = (slot + 1) % BOUNCE_SLOTS)      // oldest first
            if (!retire(slot)) return -1;
    return 0;
}

int swap_blocks_multi(const void *const *srcs, void *const *dsts, int64_t num_tensors, const int64_t *mapping,
                      int64_t num_pairs, int64_t block_bytes, int kind, hipStream_t stream) {
    if (kind < ATOMA_SWAP_GPU_TO_GPU || kind > ATOMA_SWAP_GPU_TO_CPU) {
        // csrc/src/cache_manager.rs:122-124
        set_error("swap_blocks: Either src and dst are on the same cuda device, or src and dst are on cpu and "
                  "cuda devices, alternately");
        return -1;
    }
    if (num_tensors <= 0 || num_pairs <= 0 || block_bytes <= 0) return 0;
    for (int64_t p = 0; p < num_pairs; ++p)
        if (mapping[2 * p] < 0 || mapping[2 * p + 1] < 0 || mapping[2 * p] > INT32_MAX || mapping[2 * p + 1] > INT32_MAX) {
            set_error("swap_blocks: block number out of range");
            return -1;
        }
    // Resolve device-visible aliases of the host side; any pageable tensor forces the memcpy path.
    bool kernel_path = true;
    std::vector<const char *> s(num_tensors);
    std::vector<char *> d(num_tensors);
    for (int64_t t = 0; t < num_tensors; ++t) {
        s[t] = static_cast<const char *>(srcs[t]);
        d[t] = static_cast<char *>(dsts[t]);
        if (kind == ATOMA_SWAP_CPU_TO_GPU) {
            void *a = device_alias(srcs[t]);
            if (!a) kernel_path = false; else s[t] = static_cast<const char *>(a);
        } else if (kind == ATOMA_SWAP_GPU_TO_CPU) {
            void *a = device_alias(dsts[t]);
            if (!a) kernel_path = false; else d[t] = static_cast<char *>(a);
        }
    }
    if (!kernel_path) return swap_blocks_pageable(srcs, dsts, num_tensors, mapping, num_pairs, block_bytes, kind, stream);
    const int64_t spans = cdiv(block_bytes, (int64_t)COPY_SPAN_VECS * 16);
    for (int64_t t0 = 0; t0 < num_tensors; t0 += SWAP_MAX_TENSORS) {
        const int nt = (int)((num_tensors - t0) < SWAP_MAX_TENSORS ? (num_tensors - t0) : SWAP_MAX_TENSORS);
        for (int64_t p0 = 0; p0 < num_pairs; p0 += SWAP_MAX_PAIRS) {
            const int np = (int)((num_pairs - p0) < SWAP_MAX_PAIRS ? (num_pairs - p0) : SWAP_MAX_PAIRS);
            SwapArgs a;
            for (int t = 0; t < nt; ++t) { a.src[t] = s[t0 + t]; a.dst[t] = d[t0 + t]; }
            for (int p = 0; p < np; ++p) {
                a.pairs[p][0] = (int32_t)mapping[2 * (p0 + p)];
                a.pairs[p][1] = (int32_t)mapping[2 * (p0 + p) + 1];
            }
            hipLaunchKernelGGL(swap_blocks_kernel, dim3(np, nt, (unsigned)spans), dim3(COPY_THREADS), 0, stream, a,
                               block_bytes);
            if (!ATOMA_CHECK_LAUNCH("swap_blocks")) return -1;
        }
    }
    return 0;
}

}  // namespace atoma

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

// csrc/src/ffi.rs:86-101
void reshape_and_cache_flash(void *key, void *value, void *key_cache, void *value_cache, int64_t *slot_mapping,
                             int64_t block_stride, int64_t num_tokens, int64_t num_heads, int64_t head_size,
                             int64_t block_size, int64_t key_stride, int64_t value_stride, uint32_t dtype,
                             void *stream) {
    atoma::clear_error();
    if (dtype > 1) {  // the reference silently launches nothing (cache_manager.cu:236-241)
        atoma::set_error("reshape_and_cache_flash: dtype must be 0 (f16) or 1 (bf16)");
        return;
    }
    atoma::launch_reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, block_stride, num_tokens,
                                          num_heads, head_size, block_size, key_stride, value_stride,
                                          static_cast<hipStream_t>(stream));
}

// csrc/src/ffi.rs:66-84 -- both dtypes move 16-bit elements (cache_manager.cu:40-41)
void copy_blocks_f16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int64_t num_layers,
                     int64_t num_pairs, int64_t numel_per_block, void *stream) {
    atoma::clear_error();
    atoma::launch_copy_blocks(key_cache_ptrs, value_cache_ptrs, block_mapping, num_layers, num_pairs,
                              numel_per_block, static_cast<hipStream_t>(stream));
}
void copy_blocks_bf16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int64_t num_layers,
                      int64_t num_pairs, int64_t numel_per_block, void *stream) {
    atoma::clear_error();
    atoma::launch_copy_blocks(key_cache_ptrs, value_cache_ptrs, block_mapping, num_layers, num_pairs,
                              numel_per_block, static_cast<hipStream_t>(stream));
}

int atoma_swap_blocks_multi(const void *const *srcs, void *const *dsts, int64_t num_tensors, const int64_t *mapping,
                            int64_t num_pairs, int64_t block_size_in_bytes, int kind, void *stream) {
    atoma::clear_error();
    return atoma::swap_blocks_multi(srcs, dsts, num_tensors, mapping, num_pairs, block_size_in_bytes, kind,
                                    static_cast<hipStream_t>(stream));
}
int atoma_swap_blocks(const void *src, void *dst, const int64_t *mapping, int64_t num_pairs,
                      int64_t block_size_in_bytes, int kind, void *stream) {
    return atoma_swap_blocks_multi(&src, &dst, 1, mapping, num_pairs, block_size_in_bytes, kind, stream);
}

void *atoma_host_alloc(size_t bytes) {
    atoma::clear_error();
    void *p = nullptr;
    if (!atoma::check_hip(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable), "atoma_host_alloc"))
        return nullptr;
    return p;
}
void atoma_host_free(void *p) {
    if (p) atoma::check_hip(hipHostFree(p), "atoma_host_free");
}
// Pin an allocation the caller already owns (the reference's CPU cache tensors, worker.rs:570-598) and map it into the device's
// address space: swap_blocks then moves its pages with the gather / scatter kernel at the PCIe rate instead of through the bounce ring.
int atoma_host_register(void *p, size_t bytes) {
    atoma::clear_error();
    return atoma::check_hip(hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable), "atoma_host_register") ? 0 : -1;
}
int atoma_host_unregister(void *p) {
    atoma::clear_error();
    return atoma::check_hip(hipHostUnregister(p), "atoma_host_unregister") ? 0 : -1;
}

}  // extern "C"
