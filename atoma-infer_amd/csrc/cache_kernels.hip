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
// through a pinned, device-addressable bounce ring instead: two 8 MiB slots per device; the gather / scatter kernel moves up to a
// slot's worth of pages between the cache and a slot in one launch, a small team of host threads copies between the slot and the
// caller's pages, and the two slots alternate so that the PCIe transfer of one chunk runs beside the host copy of the other.  Like
// a pageable hipMemcpy, the call returns when the host side is done (GPU -> CPU: the data is in the caller's pages; CPU -> GPU: the
// caller's pages have been read); the device side stays ordered on `stream`.
namespace {
constexpr size_t BOUNCE_SLOT = 8u << 20;
struct Bounce {
    int device = -1;
    std::mutex mu;                               // one swap at a time per DEVICE (the two slots are the device's): other devices' threads do not wait
    char *host[2] = {nullptr, nullptr};
    char *dev[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};     // the kernel that last touched the slot
    bool busy[2] = {false, false};
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
    for (int i = 0; i < 2 && ok; ++i) {
        void *h = nullptr, *d = nullptr;
        ok = check_hip(hipHostMalloc(&h, BOUNCE_SLOT, hipHostMallocMapped | hipHostMallocPortable), "swap_blocks bounce hipHostMalloc");
        b->host[i] = static_cast<char *>(h);
        ok = ok && check_hip(hipHostGetDevicePointer(&d, h, 0), "swap_blocks bounce alias");
        b->dev[i] = static_cast<char *>(d);
        ok = ok && check_hip(hipEventCreateWithFlags(&b->done[i], hipEventDisableTiming), "swap_blocks bounce event");
    }
    if (!ok) {                                   // nothing of a half-built ring is kept (a retry starts from scratch and leaks nothing)
        for (int i = 0; i < 2; ++i) {
            if (b->done[i]) (void)hipEventDestroy(b->done[i]);
            if (b->host[i]) (void)hipHostFree(b->host[i]);
        }
        return nullptr;
    }
    g_bounce.push_back(std::move(b));
    return g_bounce.back().get();
}

// n pages between a packed slot and the caller's pageable pages, split over a few threads (one memcpy stream does ~10 GB/s)
void host_copy_pages(char *slot, char *tensor, const int64_t *pages, int64_t n, int64_t block_bytes, bool to_slot) {
    const int64_t bytes = n * block_bytes;
    const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(4, bytes >> 20));
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            char *pg = tensor + pages[k] * block_bytes, *sl = slot + k * block_bytes;
            if (to_slot) memcpy(sl, pg, (size_t)block_bytes); else memcpy(pg, sl, (size_t)block_bytes);
        }
    };
    if (nthreads == 1) { work(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, n * t / nthreads, n * (t + 1) / nthreads);
    work(0, n / nthreads);
    for (auto &t : th) t.join();
}
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
    const bool out = kind == ATOMA_SWAP_GPU_TO_CPU;
    const int64_t per_slot = std::min<int64_t>((int64_t)(BOUNCE_SLOT / (size_t)block_bytes), SWAP_MAX_PAIRS);
    const int64_t spans = cdiv(block_bytes, (int64_t)COPY_SPAN_VECS * 16);
    std::vector<int64_t> host_pages((size_t)per_slot);
    struct Pending { int slot; int64_t t, p0, n; };
    Pending pend{-1, 0, 0, 0};                   // GPU -> CPU: the chunk whose gather is in flight
    auto drain = [&](const Pending &c) -> bool {  // wait for its gather, copy the slot out to the caller's pages
        if (!check_hip(hipEventSynchronize(bn->done[c.slot]), "swap_blocks bounce wait")) return false;
        bn->busy[c.slot] = false;
        for (int64_t k = 0; k < c.n; ++k) host_pages[(size_t)k] = mapping[2 * (c.p0 + k) + 1];
        host_copy_pages(bn->host[c.slot], static_cast<char *>(dsts[c.t]), host_pages.data(), c.n, block_bytes, false);
        return true;
    };
    int slot = 0;
    for (int64_t t = 0; t < num_tensors; ++t)
        for (int64_t p0 = 0; p0 < num_pairs; p0 += per_slot, slot ^= 1) {
            const int64_t n = std::min(per_slot, num_pairs - p0);
            if (bn->busy[slot]) {                 // the kernel that last read / wrote this slot (this call or an earlier one)
                if (out && pend.slot == slot) { if (!drain(pend)) return -1; pend.slot = -1; }
                else if (!check_hip(hipEventSynchronize(bn->done[slot]), "swap_blocks bounce wait")) return -1;
                bn->busy[slot] = false;
            }
            SwapArgs a;
            for (int64_t k = 0; k < n; ++k) {
                a.pairs[k][0] = out ? (int32_t)mapping[2 * (p0 + k)] : (int32_t)k;
                a.pairs[k][1] = out ? (int32_t)k : (int32_t)mapping[2 * (p0 + k) + 1];
            }
            if (out) {
                a.src[0] = static_cast<const char *>(srcs[t]);
                a.dst[0] = bn->dev[slot];
            } else {
                for (int64_t k = 0; k < n; ++k) host_pages[(size_t)k] = mapping[2 * (p0 + k)];
                host_copy_pages(bn->host[slot], const_cast<char *>(static_cast<const char *>(srcs[t])), host_pages.data(), n, block_bytes, true);
                a.src[0] = bn->dev[slot];
                a.dst[0] = static_cast<char *>(dsts[t]);
            }
            hipLaunchKernelGGL(swap_blocks_kernel, dim3((unsigned)n, 1, (unsigned)spans), dim3(COPY_THREADS), 0, stream, a, block_bytes);
            if (!ATOMA_CHECK_LAUNCH("swap_blocks (bounce)")) return -1;
            if (!check_hip(hipEventRecord(bn->done[slot], stream), "swap_blocks bounce record")) return -1;
            bn->busy[slot] = true;
            if (out) {                            // the previous chunk's host copy runs beside this chunk's gather
                if (pend.slot >= 0 && !drain(pend)) return -1;
                pend = Pending{slot, t, p0, n};
            }
        }
    if (out && pend.slot >= 0 && !drain(pend)) return -1;
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
