// Direct tensor-parallel sum all-reduce over peer-mapped memory (xGMI between the 8 GPUs of a node) -- the exchange step of
// /root/reference/models/src/multi_gpu.rs:141-179 (`AllReduce::cuda_fwd` -> ncclAllReduce) for the messages a decode
// step produces: [tokens, hidden] bf16 after o_proj and after down_proj (llama_nccl.rs:139,195), 16 KiB .. a few MiB,
// 160 times per 80-layer step.  A ring/tree collective pays several hops of launch + latency per message; on a fully
// connected xGMI mesh every pair of GPUs has its own link, so small messages want ONE hop:
//
//   one-shot  (<= `oneshot_max` bytes): every rank PUSHES its whole input into slot[rank] of every peer's staging
//             region (W - 1 posted remote writes per 16-byte vector, no remote reads), raises one flag per peer and block,
//             waits for the W - 1 flags addressed to it, then sums the W contributions locally.
//   two-shot  (above): reduce-scatter + all-gather, each one hop: rank r owns chunk r; everybody pushes its piece of
//             chunk r to rank r (stage 0), rank r sums it and pushes the result to everybody (stage 1).  2.S.(W-1)/W bytes
//             leave each GPU instead of S.(W-1), spread evenly over the W - 1 links.
//
// Numerics: contributions are added in fp32 in RANK ORDER 0..W-1 on every rank and rounded once, so all ranks hold
// bit-identical results (NCCL/RCCL's ring gives each rank a different summation order only for the chunk it owns, the
// result is identical across ranks there too; the VALUE may differ from RCCL's in the last bit -- tested as <= 1 ulp).
//
// Protocol (no grid-wide barrier, no host involvement, capturable in a hipGraph):
//   * staging regions are allocated uncached (hipDeviceMallocUncached: fine-grained, never resident in a non-coherent
//     L2) and additionally accessed with system-scope (sc0 sc1) loads / stores;
//   * a call's sequence number lives in device memory -- in the uncached region itself -- (read by every block at kernel
//     start, advanced by the last block to leave), so a captured graph replays correctly; its parity selects one of two staging halves: a rank can only
//     start call n + 1 after every peer's flags of call n arrived, i.e. after every peer finished READING the staging
//     half of call n - 1, which is the half call n + 1 overwrites -- no end-of-call barrier is needed;
//   * block b of every rank works on the same element ranges and talks only to block b of its peers (flag per
//     (parity, stage, source rank, block)); all ranks must issue the same sequence of calls with the same counts, on one
//     stream per communicator -- the usual collective contract;
//   * every spin is bounded (default 30 s, ATOMA_XGMI_TIMEOUT_MS): a lost peer turns into an error word the host can
//     read (atoma_xgmi_status), never into a hung GPU.
//
// Setup mirrors the ncclUniqueId bootstrap: each rank creates its region, exports a 128-byte handle (pid, device, pointer,
// hipIpcMemHandle), the caller all-gathers the handles out of band (the Rust engine: over its thread channels -- there all
// ranks live in one process, model_executor.rs:428, and peers are reached through plain peer access; bench / tests:
// torch.distributed gloo or RCCL itself) and every rank connects.  Same-process peers use the raw pointer, other
// processes hipIpcOpenMemHandle (dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0).
#include "common.h"
#include "norm_shared.h"

#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string>
#include <string.h>
#include <unistd.h>

namespace atoma {

constexpr int XGMI_MAX_WORLD = 8;
constexpr int XGMI_MAX_BLOCKS = 64;
constexpr int XGMI_THREADS = 512;
constexpr uint32_t XGMI_MAGIC = 0x31475841u;   // "AXG1"

typedef unsigned int xu32x4 __attribute__((ext_vector_type(4)));

struct XgmiParams {
    char *peer[XGMI_MAX_WORLD];   // base of every rank's staging region as mapped in THIS process (peer[rank] = own)
    const char *in;
    char *out;
    uint32_t *seq;                // device: [0] = sequence number of the last completed call, [1] = blocks that left the current one, [2] = a wait timed out
    uint32_t *status;             // host-mapped: != 0 after a timed-out wait
    int64_t nvec;                 // 16-byte vectors in the message
    int64_t chunk_vec;            // two-shot: vectors per rank chunk (ceil(nvec / world))
    int64_t slot_bytes;           // bytes between two source slots of the scatter / gather area
    int64_t off_scatter, off_gather, half_bytes;   // area offsets inside a parity half, size of a half
    int64_t off_flags;
    long long timeout_ticks;      // wall_clock64 ticks (100 MHz)
    int rank, world;
};

// flags: [parity 2][stage 2][source rank 8][block 64] uint32
__device__ __forceinline__ uint32_t *xgmi_flag(char *base, const XgmiParams &p, int parity, int stage, int src, int block) {
    return reinterpret_cast<uint32_t *>(base + p.off_flags) + (((parity * 2 + stage) * XGMI_MAX_WORLD + src) * XGMI_MAX_BLOCKS + block);
}

// 16-byte system-scope (sc0 sc1) accesses through a raw buffer descriptor based at the (wave-uniform) area pointer
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xgmi_rsrc(char *base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
}
constexpr int XGMI_AUX_SYS = 1 | 16;   // sc0 | sc1
__device__ __forceinline__ void store_sys(__amdgpu_buffer_rsrc_t r, int64_t byte_off, xu32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, XGMI_AUX_SYS);
}
__device__ __forceinline__ xu32x4 load_sys(__amdgpu_buffer_rsrc_t r, int64_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, XGMI_AUX_SYS);
}

template <typename T> struct XgmiAcc {   // fp32 accumulator of one 16-byte vector
    static constexpr int N = 8;
    float a[8];
    __device__ __forceinline__ void set(const xu32x4 &v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] = lo_to_f32<T>(v[e]); a[2 * e + 1] = hi_to_f32<T>(v[e]); }
    }
    __device__ __forceinline__ void add(const xu32x4 &v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] += lo_to_f32<T>(v[e]); a[2 * e + 1] += hi_to_f32<T>(v[e]); }
    }
    __device__ __forceinline__ xu32x4 round() const {
        xu32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = pack2<T>(a[2 * e], a[2 * e + 1]);
        return r;
    }
};
template <> struct XgmiAcc<float> {
    float a[4];
    __device__ __forceinline__ void set(const xu32x4 &v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = __uint_as_float(v[e]);
    }
    __device__ __forceinline__ void add(const xu32x4 &v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += __uint_as_float(v[e]);
    }
    __device__ __forceinline__ xu32x4 round() const {
        xu32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = __float_as_uint(a[e]);
        return r;
    }
};

// Poll back-off: a waiting lane polls a system-scope flag; polled at a fixed ~60 ns period, the waiters of eight ranks that
// share ONE device (the virtual-rank rig) kept the fabric busy enough to starve the vendor GEMMs beside them.  The period grows from ~60 ns
// over ~0.25 us to ~1 us (s_sleep counts 64-clock units; its operand is an immediate): a flag that is about to flip is still seen at once,
// a long wait costs 1/16 of the polls.
__device__ __forceinline__ void xgmi_backoff(int &polls) {
    if (polls < 8) __builtin_amdgcn_s_sleep(2);
    else if (polls < 32) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(32);
    ++polls;
}

// Release this block's remote stores, then tell block `blockIdx.x` of every peer; then wait for theirs.
__device__ __forceinline__ void xgmi_signal_and_wait(const XgmiParams &p, int parity, int stage, uint32_t seq) {
    __threadfence_system();                    // every thread: its stores are ordered before the flag (system-scope release)
    __syncthreads();
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        __hip_atomic_store(xgmi_flag(p.peer[t], p, parity, stage, p.rank, blockIdx.x), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t *mine = xgmi_flag(p.peer[p.rank], p, parity, stage, t, blockIdx.x);
        const long long t0 = wall_clock64();
        // once a wait has timed out (device-side copy of the status in seq[2]) the communicator is broken: later waits give
        // up at once, so a captured graph of 160 calls costs one timeout, not 160
        const long long limit = __hip_atomic_load(p.seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 0 : p.timeout_ticks;
        int polls = 0;
        while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            xgmi_backoff(polls);
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(p.status, 1u + (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // which peer never arrived
                __hip_atomic_store(p.seq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

// The last block to leave publishes the new sequence number (kernel boundary = visibility for the next call's blocks).
__device__ __forceinline__ void xgmi_leave(const XgmiParams &p, uint32_t seq) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = __hip_atomic_fetch_add(p.seq + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (n == gridDim.x - 1) {
            __hip_atomic_store(p.seq + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(XGMI_THREADS) xgmi_oneshot_kernel(const XgmiParams p) {
    const uint32_t seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    const int parity = (int)(seq & 1u);
    const int64_t half = (int64_t)parity * p.half_bytes;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const xu32x4 *in = reinterpret_cast<const xu32x4 *>(p.in);
    // push: slot[rank] of every peer, peers visited in rotation so that the W - 1 links are loaded at the same time
    for (int r = 1; r < p.world; ++r) {
        const int q = (p.rank + r) % p.world;
        const __amdgpu_buffer_rsrc_t dst = xgmi_rsrc(p.peer[q] + half + p.off_scatter + (int64_t)p.rank * p.slot_bytes);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.nvec; i += stride) store_sys(dst, i * 16, in[i]);
    }
    xgmi_signal_and_wait(p, parity, 0, seq);
    char *mine = p.peer[p.rank] + half + p.off_scatter;
    xu32x4 *out = reinterpret_cast<xu32x4 *>(p.out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.nvec; i += stride) {
        XgmiAcc<T> acc;
        for (int q = 0; q < p.world; ++q) {
            const xu32x4 v = q == p.rank ? in[i] : load_sys(xgmi_rsrc(mine + (int64_t)q * p.slot_bytes), i * 16);
            if (q == 0) acc.set(v); else acc.add(v);
        }
        out[i] = acc.round();
    }
    xgmi_leave(p, seq);
}

template <typename T>
__global__ void __launch_bounds__(XGMI_THREADS) xgmi_twoshot_kernel(const XgmiParams p) {
    const uint32_t seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    const int parity = (int)(seq & 1u);
    const int64_t half = (int64_t)parity * p.half_bytes;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const xu32x4 *in = reinterpret_cast<const xu32x4 *>(p.in);
    xu32x4 *out = reinterpret_cast<xu32x4 *>(p.out);
    auto chunk_len = [&](int c) -> int64_t {
        const int64_t left = p.nvec - (int64_t)c * p.chunk_vec;
        return left <= 0 ? 0 : (left < p.chunk_vec ? left : p.chunk_vec);
    };
    // stage 0 (reduce-scatter): my piece of chunk c goes to slot[rank] of rank c
    for (int r = 1; r < p.world; ++r) {
        const int c = (p.rank + r) % p.world;
        const int64_t n = chunk_len(c), base = (int64_t)c * p.chunk_vec;
        const __amdgpu_buffer_rsrc_t dst = xgmi_rsrc(p.peer[c] + half + p.off_scatter + (int64_t)p.rank * p.slot_bytes);
        for (int64_t i = first; i < n; i += stride) store_sys(dst, i * 16, in[base + i]);
    }
    xgmi_signal_and_wait(p, parity, 0, seq);
    // sum my chunk in rank order, keep it, and push it to slot[rank] of every peer's gather area (stage 1: all-gather)
    {
        const int64_t n = chunk_len(p.rank), base = (int64_t)p.rank * p.chunk_vec;
        char *mine = p.peer[p.rank] + half + p.off_scatter;
        for (int64_t i = first; i < n; i += stride) {
            XgmiAcc<T> acc;
            for (int q = 0; q < p.world; ++q) {
                const xu32x4 v = q == p.rank ? in[base + i] : load_sys(xgmi_rsrc(mine + (int64_t)q * p.slot_bytes), i * 16);
                if (q == 0) acc.set(v); else acc.add(v);
            }
            const xu32x4 r = acc.round();
            out[base + i] = r;
            for (int rr = 1; rr < p.world; ++rr) {
                const int q = (p.rank + rr) % p.world;
                store_sys(xgmi_rsrc(p.peer[q] + half + p.off_gather + (int64_t)p.rank * p.slot_bytes), i * 16, r);
            }
        }
    }
    xgmi_signal_and_wait(p, parity, 1, seq);
    char *gathered = p.peer[p.rank] + half + p.off_gather;
    for (int r = 1; r < p.world; ++r) {
        const int c = (p.rank + r) % p.world;
        const int64_t n = chunk_len(c), base = (int64_t)c * p.chunk_vec;
        const __amdgpu_buffer_rsrc_t src = xgmi_rsrc(gathered + (int64_t)c * p.slot_bytes);
        for (int64_t i = first; i < n; i += stride) out[base + i] = load_sys(src, i * 16);
    }
    xgmi_leave(p, seq);
}

// ------------------------------------------------------------------------------------------
// all-reduce + residual add + RMSNorm in ONE launch per rank
// ------------------------------------------------------------------------------------------
// What follows both all-reduces of a tensor-parallel decoder layer (llama_nccl.rs:139 -> llama.rs:404,408; :195 -> :409 and the next
// layer's :402): x' = x + allreduce(partial), then RMSNorm(x').  Every rank holds the whole summed row after the exchange, so the rank
// that sums it can add the residual and normalise it while the row is in registers: two launches per layer fewer in the TP step (the
// all-reduce kernel's tail replaces `atoma_add_rms_norm`), bit-identical to atoma_xgmi_allreduce_sum followed by atoma_add_rms_norm --
// the sum is rounded to the storage type first (what the all-reduce writes), the add and the norm are norm_shared.h's arithmetic in
// rms_norm_kernel's thread layout: 256 threads per row, thread t owns the row's 16-byte vectors t, t + 256, ...
// Blocks of NORM_THREADS threads; a block owns the rows blockIdx.x, blockIdx.x + gridDim.x, ...

// the signal half of xgmi_signal_and_wait: after this block's stores, one flag per peer (and, `self` set, one in this rank's own region:
// the rows other blocks of this rank will read)
__device__ __forceinline__ void xgmi_signal(const XgmiParams &p, int parity, int stage, uint32_t seq, bool self) {
    __threadfence_system();
    __syncthreads();
    const int t = threadIdx.x;
    if (t < p.world && (self || t != p.rank))
        __hip_atomic_store(xgmi_flag(p.peer[t], p, parity, stage, p.rank, blockIdx.x), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait for the flags of EVERY block of every rank (this rank's own included when `self`): a row of the result was produced by several
// blocks of its owner
__device__ __forceinline__ void xgmi_wait_all_blocks(const XgmiParams &p, int parity, int stage, uint32_t seq, bool self) {
    const int n = p.world * (int)gridDim.x;
    const long long limit = __hip_atomic_load(p.seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 0 : p.timeout_ticks;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int src = i / (int)gridDim.x, blk = i - src * (int)gridDim.x;
        if (src == p.rank && !self) continue;
        uint32_t *mine = xgmi_flag(p.peer[p.rank], p, parity, stage, src, blk);
        const long long t0 = wall_clock64();
        int polls = 0;
        while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            xgmi_backoff(polls);
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(p.status, 1u + (uint32_t)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(p.seq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

struct XgmiNormParams {
    const uint16_t *residual, *weight;
    uint16_t *x_out, *norm_out;
    int64_t res_stride, x_stride, norm_stride;   // elements
    int rows, hidden;
    float eps;
};

// One row, its summed vectors in sv[ITERS] (already rounded to T): x' = residual + sum -> x_out, RMSNorm(x') -> norm_out
template <typename T, int ITERS>
__device__ __forceinline__ void xgmi_add_norm_row(const XgmiNormParams &n, int64_t row, uint4 (&sv)[ITERS], float *red) {
    const int nvec = n.hidden >> 3;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = it * NORM_THREADS + threadIdx.x;
        if (i < nvec) {
            float fa[8], fb[8];
            unpack8<T>(reinterpret_cast<const uint4 *>(n.residual + row * n.res_stride)[i], fa);
            unpack8<T>(sv[it], fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) fa[e] += fb[e];
            sv[it] = pack8<T>(fa);
            reinterpret_cast<uint4 *>(n.x_out + row * n.x_stride)[i] = sv[it];
        } else {
            sv[it] = make_uint4(0, 0, 0, 0);
        }
        ss = norm_sumsq8<T>(sv[it], ss);
    }
    ss = norm_wave_sum(ss);
    __syncthreads();                                   // (red is reused from the previous row)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float redr[NORM_THREADS / 64];
#pragma unroll
    for (int i = 0; i < NORM_THREADS / 64; ++i) redr[i] = red[i];
    const float scale = norm_scale(redr, n.hidden, n.eps);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = it * NORM_THREADS + threadIdx.x;
        if (i < nvec) reinterpret_cast<uint4 *>(n.norm_out + row * n.norm_stride)[i] = norm_apply8<T>(sv[it], reinterpret_cast<const uint4 *>(n.weight)[i], scale);
    }
}

template <typename T, int ITERS>
__global__ void __launch_bounds__(NORM_THREADS) xgmi_oneshot_add_norm_kernel(const XgmiParams p, const XgmiNormParams n) {
    __shared__ float red[NORM_THREADS / 64];
    const uint32_t seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    const int parity = (int)(seq & 1u);
    const int64_t half = (int64_t)parity * p.half_bytes;
    const int nvec = n.hidden >> 3;                    // vectors per row; the message is rows x nvec vectors, row-major (p.in contiguous)
    const xu32x4 *in = reinterpret_cast<const xu32x4 *>(p.in);
    // push this block's rows into slot[rank] of every peer
    for (int r = 1; r < p.world; ++r) {
        const int q = (p.rank + r) % p.world;
        const __amdgpu_buffer_rsrc_t dst = xgmi_rsrc(p.peer[q] + half + p.off_scatter + (int64_t)p.rank * p.slot_bytes);
        for (int64_t row = blockIdx.x; row < n.rows; row += gridDim.x)
            for (int i = threadIdx.x; i < nvec; i += NORM_THREADS) store_sys(dst, (row * nvec + i) * 16, in[row * nvec + i]);
    }
    xgmi_signal_and_wait(p, parity, 0, seq);           // block b of every rank owns the same rows: the per-block flags suffice
    char *mine = p.peer[p.rank] + half + p.off_scatter;
    for (int64_t row = blockIdx.x; row < n.rows; row += gridDim.x) {
        uint4 sv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = it * NORM_THREADS + threadIdx.x;
            sv[it] = make_uint4(0, 0, 0, 0);
            if (i < nvec) {
                XgmiAcc<T> acc;
                for (int q = 0; q < p.world; ++q) {
                    const xu32x4 v = q == p.rank ? in[row * nvec + i] : load_sys(xgmi_rsrc(mine + (int64_t)q * p.slot_bytes), (row * nvec + i) * 16);
                    if (q == 0) acc.set(v); else acc.add(v);
                }
                sv[it] = __builtin_bit_cast(uint4, acc.round());
            }
        }
        xgmi_add_norm_row<T, ITERS>(n, row, sv, red);
    }
    xgmi_leave(p, seq);
}

template <typename T, int ITERS>
__global__ void __launch_bounds__(NORM_THREADS) xgmi_twoshot_add_norm_kernel(const XgmiParams p, const XgmiNormParams n) {
    __shared__ float red[NORM_THREADS / 64];
    const uint32_t seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    const int parity = (int)(seq & 1u);
    const int64_t half = (int64_t)parity * p.half_bytes;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const xu32x4 *in = reinterpret_cast<const xu32x4 *>(p.in);
    auto chunk_len = [&](int c) -> int64_t {
        const int64_t left = p.nvec - (int64_t)c * p.chunk_vec;
        return left <= 0 ? 0 : (left < p.chunk_vec ? left : p.chunk_vec);
    };
    // stage 0 (reduce-scatter), as xgmi_twoshot_kernel
    for (int r = 1; r < p.world; ++r) {
        const int c = (p.rank + r) % p.world;
        const int64_t len = chunk_len(c), base = (int64_t)c * p.chunk_vec;
        const __amdgpu_buffer_rsrc_t dst = xgmi_rsrc(p.peer[c] + half + p.off_scatter + (int64_t)p.rank * p.slot_bytes);
        for (int64_t i = first; i < len; i += stride) store_sys(dst, i * 16, in[base + i]);
    }
    xgmi_signal_and_wait(p, parity, 0, seq);
    // stage 1: my chunk summed in rank order, rounded, into slot[rank] of EVERYBODY's gather area -- my own too: the rows are put together from there
    {
        const int64_t len = chunk_len(p.rank), base = (int64_t)p.rank * p.chunk_vec;
        char *mine = p.peer[p.rank] + half + p.off_scatter;
        for (int64_t i = first; i < len; i += stride) {
            XgmiAcc<T> acc;
            for (int q = 0; q < p.world; ++q) {
                const xu32x4 v = q == p.rank ? in[base + i] : load_sys(xgmi_rsrc(mine + (int64_t)q * p.slot_bytes), i * 16);
                if (q == 0) acc.set(v); else acc.add(v);
            }
            const xu32x4 r = acc.round();
            for (int rr = 0; rr < p.world; ++rr) {
                const int q = (p.rank + rr) % p.world;
                store_sys(xgmi_rsrc(p.peer[q] + half + p.off_gather + (int64_t)p.rank * p.slot_bytes), i * 16, r);
            }
        }
    }
    xgmi_signal(p, parity, 1, seq, true);
    xgmi_wait_all_blocks(p, parity, 1, seq, true);     // a row's vectors were summed by several blocks of its owner(s)
    char *gathered = p.peer[p.rank] + half + p.off_gather;
    const int nvec = n.hidden >> 3;
    for (int64_t row = blockIdx.x; row < n.rows; row += gridDim.x) {
        uint4 sv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = it * NORM_THREADS + threadIdx.x;
            sv[it] = make_uint4(0, 0, 0, 0);
            if (i < nvec) {
                const int64_t gi = row * nvec + i;
                const int c = (int)(gi / p.chunk_vec);
                sv[it] = __builtin_bit_cast(uint4, load_sys(xgmi_rsrc(gathered + (int64_t)c * p.slot_bytes), (gi - (int64_t)c * p.chunk_vec) * 16));
            }
        }
        xgmi_add_norm_row<T, ITERS>(n, row, sv, red);
    }
    xgmi_leave(p, seq);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct XgmiHandleBlob {          // 128 bytes, exchanged out of band like an ncclUniqueId
    uint32_t magic, world;
    int32_t rank, device;
    int64_t pid;
    uint64_t ptr, bytes, capacity;
    hipIpcMemHandle_t ipc;       // 64 bytes
    char pad[128 - 48 - sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(XgmiHandleBlob) == 128, "handle blob must be 128 bytes");

struct Xgmi {
    int rank, world, device;
    size_t capacity;             // largest message served by one launch (bytes)
    size_t oneshot_max;
    size_t region_bytes, half_bytes, off_flags, off_scatter, off_gather, slot_bytes, off_seq;
    char *region = nullptr;
    char *peer[XGMI_MAX_WORLD] = {};
    bool ipc_opened[XGMI_MAX_WORLD] = {};
    bool connected = false;
    uint32_t *seq = nullptr;      // inside the uncached region (off_seq)
    uint32_t *status = nullptr;   // host-mapped
    uint32_t *status_dev = nullptr;
    long long timeout_ticks;
};

static size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

static void xgmi_layout(Xgmi &x) {
    // a parity half = scatter area [world slots] + gather area [world slots]; a slot holds a whole one-shot message or one
    // rank chunk of the largest two-shot message, whichever is larger
    const size_t chunk = round_up((x.capacity + x.world - 1) / x.world, 16);
    x.slot_bytes = round_up(std::max(chunk, std::min(x.oneshot_max, x.capacity)), 256);
    x.off_flags = 0;
    // the flag words, then this rank's own call counter / ticket / broken marker (x.off_seq): they live in the UNCACHED region
    // too -- consecutive kernels of a stream (and of a replayed graph) read and advance the counter from different XCDs, and an
    // L2-served load of ordinary device memory can return the previous call's value (seen as ranks drifting apart by one call)
    const size_t flag_words = (size_t)2 * 2 * XGMI_MAX_WORLD * XGMI_MAX_BLOCKS * sizeof(uint32_t);
    x.off_seq = flag_words;
    const size_t flags_bytes = round_up(flag_words + 256, 4096);
    x.off_scatter = 0;                                  // offsets inside a half
    x.off_gather = x.slot_bytes * x.world;
    x.half_bytes = 2 * x.slot_bytes * x.world;
    x.region_bytes = flags_bytes + 2 * x.half_bytes;
    // the halves start after the flags: fold that into the area offsets so that a half is addressed as base + parity * half_bytes + off
    x.off_scatter += flags_bytes;
    x.off_gather += flags_bytes;
}

}  // namespace atoma

extern "C" {

int atoma_xgmi_create(void **out, int rank, int world_size, int device, int64_t max_bytes) {
    using namespace atoma;
    clear_error();
    if (!out || world_size < 1 || world_size > XGMI_MAX_WORLD || rank < 0 || rank >= world_size || max_bytes < 16) {
        set_error("atoma_xgmi_create: need 1 <= world_size <= 8, 0 <= rank < world_size, max_bytes >= 16");
        return -1;
    }
    if (!check_hip(hipSetDevice(device), "hipSetDevice")) return -1;
    auto *x = new Xgmi();
    x->rank = rank; x->world = world_size; x->device = device;
    x->capacity = round_up((size_t)max_bytes, 16);
    const char *om = getenv("ATOMA_XGMI_ONESHOT_MAX");
    x->oneshot_max = om ? (size_t)atoll(om) : (size_t)512 << 10;
    const char *tm = getenv("ATOMA_XGMI_TIMEOUT_MS");
    x->timeout_ticks = (long long)(tm ? atoll(tm) : 30000) * 100000ll;   // wall_clock64: 100 MHz
    xgmi_layout(*x);
    void *mem = nullptr;
    if (!check_hip(hipExtMallocWithFlags(&mem, x->region_bytes, hipDeviceMallocUncached), "xgmi staging region (hipExtMallocWithFlags uncached)")) { delete x; return -1; }
    x->region = static_cast<char *>(mem);
    bool ok = check_hip(hipMemset(x->region, 0, x->region_bytes), "xgmi region memset");
    x->seq = reinterpret_cast<uint32_t *>(x->region + x->off_seq);
    ok = ok && check_hip(hipHostMalloc(reinterpret_cast<void **>(&x->status), sizeof(uint32_t), hipHostMallocMapped), "xgmi status word");
    ok = ok && check_hip(hipDeviceSynchronize(), "xgmi create sync");
    if (!ok) { (void)hipFree(x->region); delete x; return -1; }
    *x->status = 0;
    x->status_dev = x->status;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&x->status_dev), x->status, 0) != hipSuccess) { (void)hipGetLastError(); x->status_dev = x->status; }
    x->peer[rank] = x->region;
    if (world_size == 1) x->connected = true;
    *out = x;
    return 0;
}

}  // extern "C" (reopened below)
namespace atoma {
// Fault injection for the set-up calls that have only ever run on one device: atoma_set_option("xgmi_fault", n) makes the
// next set-ups behave as if 1 = hipIpcOpenMemHandle, 2 = hipDeviceEnablePeerAccess, 3 = hipIpcGetMemHandle had failed (0 = off) -- the error
// path of a first multi-GPU contact (clean message, nothing leaked, the handle reusable or destroyable, RCCL still there) is then testable here.
static std::atomic<int> xgmi_fault{0};
bool set_xgmi_option(const std::string &name, int value) {
    if (name != "xgmi_fault") return false;
    xgmi_fault = value;
    return true;
}
static bool xgmi_injected(int which, const char *what) {
    if (xgmi_fault.load() != which) return false;
    set_error(std::string(what) + ": hipErrorInvalidValue (injected by the xgmi_fault option)");
    return true;
}
}  // namespace atoma
extern "C" {

int atoma_xgmi_handle(void *xg, void *handle128_out) {
    using namespace atoma;
    clear_error();
    auto *x = static_cast<Xgmi *>(xg);
    if (!x || !handle128_out) { set_error("atoma_xgmi_handle: null argument"); return -1; }
    XgmiHandleBlob b;
    memset(&b, 0, sizeof b);
    b.magic = XGMI_MAGIC; b.world = (uint32_t)x->world; b.rank = x->rank; b.device = x->device;
    b.pid = (int64_t)getpid();
    b.ptr = (uint64_t)reinterpret_cast<uintptr_t>(x->region);
    b.bytes = x->region_bytes; b.capacity = x->capacity;
    if (xgmi_injected(3, "hipIpcGetMemHandle (xgmi staging region)")) return -1;
    if (!check_hip(hipIpcGetMemHandle(&b.ipc, x->region), "hipIpcGetMemHandle (xgmi staging region)")) return -1;
    memcpy(handle128_out, &b, sizeof b);
    return 0;
}

int atoma_xgmi_connect(void *xg, const void *handles /* world_size x 128 bytes, rank order */) {
    using namespace atoma;
    clear_error();
    auto *x = static_cast<Xgmi *>(xg);
    if (!x || !handles) { set_error("atoma_xgmi_connect: null argument"); return -1; }
    if (!check_hip(hipSetDevice(x->device), "hipSetDevice")) return -1;
    const auto *blobs = static_cast<const XgmiHandleBlob *>(handles);
    for (int q = 0; q < x->world; ++q) {
        const XgmiHandleBlob &b = blobs[q];
        if (b.magic != XGMI_MAGIC || (int)b.world != x->world || b.rank != q || b.bytes != x->region_bytes || b.capacity != x->capacity) {
            set_error("atoma_xgmi_connect: handle " + std::to_string(q) + " does not describe rank " + std::to_string(q) +
                      " of this communicator (all ranks must be created with the same world_size and max_bytes)");
            return -1;
        }
        if (q == x->rank || x->peer[q]) continue;   // (a peer mapped by an earlier, failed attempt stays mapped: connect can be retried)
        if (b.pid == (int64_t)getpid()) {          // same process (one thread per GPU, as the reference runs): plain peer access
            if (xgmi_injected(2, "hipDeviceEnablePeerAccess")) return -1;
            if (b.device != x->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { check_hip(e, "hipDeviceEnablePeerAccess"); return -1; }
                (void)hipGetLastError();
            }
            x->peer[q] = reinterpret_cast<char *>((uintptr_t)b.ptr);
        } else {
            void *mapped = nullptr;
            if (xgmi_injected(1, "hipIpcOpenMemHandle (peer staging region)")) return -1;
            if (!check_hip(hipIpcOpenMemHandle(&mapped, b.ipc, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (peer staging region)")) return -1;
            x->peer[q] = static_cast<char *>(mapped);
            x->ipc_opened[q] = true;
        }
    }
    x->connected = true;
    return 0;
}

// 0 = healthy; n > 0: a wait for rank n - 1 timed out in some call (results of that call and later ones are invalid)
int atoma_xgmi_status(void *xg) {
    auto *x = static_cast<atoma::Xgmi *>(xg);
    return x ? (int)*reinterpret_cast<volatile uint32_t *>(x->status) : -1;
}

int64_t atoma_xgmi_capacity(void *xg) {
    auto *x = static_cast<atoma::Xgmi *>(xg);
    return x ? (int64_t)x->capacity : -1;
}

// mode: 0 = by size (one-shot up to oneshot_max), 1 = force one-shot (message must fit a slot), 2 = force two-shot
int atoma_xgmi_allreduce_sum_mode(void *xg, const void *in, void *out, int64_t count, int dtype, int mode, void *stream) {
    using namespace atoma;
    clear_error();
    auto *x = static_cast<Xgmi *>(xg);
    if (!x) { set_error("atoma_xgmi_allreduce_sum: null communicator"); return -1; }
    if (!x->connected) { set_error("atoma_xgmi_allreduce_sum: atoma_xgmi_connect has not been called"); return -1; }
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16 && dtype != ATOMA_F32) { set_error("atoma_xgmi_allreduce_sum: dtype must be f16, bf16 or f32"); return -1; }
    if (count < 0) { set_error("atoma_xgmi_allreduce_sum: negative count"); return -1; }
    if (count == 0) return 0;
    const int64_t bytes = count * (dtype == ATOMA_F32 ? 4 : 2);
    if (bytes % 16 || (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) {
        set_error("atoma_xgmi_allreduce_sum: the message must be a multiple of 16 bytes and in / out 16-byte aligned");
        return -1;
    }
    if (*reinterpret_cast<volatile uint32_t *>(x->status) != 0) { set_error("atoma_xgmi_allreduce_sum: an earlier call timed out waiting for a peer (atoma_xgmi_status)"); return -1; }
    const auto s = static_cast<hipStream_t>(stream);
    if (x->world == 1) {
        if (in != out && !check_hip(hipMemcpyAsync(out, in, (size_t)bytes, hipMemcpyDeviceToDevice, s), "xgmi world-of-one copy")) return -1;
        return 0;
    }
    // messages beyond the capacity go out in pieces (every rank cuts the same way)
    for (int64_t off = 0; off < bytes; off += (int64_t)x->capacity) {
        const int64_t piece = std::min<int64_t>((int64_t)x->capacity, bytes - off);
        XgmiParams p{};
        for (int q = 0; q < x->world; ++q) p.peer[q] = x->peer[q];
        p.in = static_cast<const char *>(in) + off;
        p.out = static_cast<char *>(out) + off;
        p.seq = x->seq;
        p.status = x->status_dev;
        p.nvec = piece / 16;
        p.chunk_vec = (p.nvec + x->world - 1) / x->world;
        p.slot_bytes = (int64_t)x->slot_bytes;
        p.off_scatter = (int64_t)x->off_scatter; p.off_gather = (int64_t)x->off_gather; p.half_bytes = (int64_t)x->half_bytes;
        p.off_flags = (int64_t)x->off_flags;
        p.timeout_ticks = x->timeout_ticks;
        p.rank = x->rank; p.world = x->world;
        const bool fits_slot = (size_t)piece <= x->slot_bytes;
        const bool oneshot = mode == 1 ? true : (mode == 2 ? false : (size_t)piece <= x->oneshot_max && fits_slot);
        if (oneshot && !fits_slot) { set_error("atoma_xgmi_allreduce_sum: message too large for the one-shot slots"); return -1; }
        const int64_t work = oneshot ? p.nvec : p.chunk_vec;
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(XGMI_MAX_BLOCKS, cdiv(work, XGMI_THREADS)));
#define ATOMA_XG(K, TT) hipLaunchKernelGGL((K<TT>), dim3(blocks), dim3(XGMI_THREADS), 0, s, p)
        if (oneshot) {
            if (dtype == ATOMA_BF16) ATOMA_XG(xgmi_oneshot_kernel, bf16_t); else if (dtype == ATOMA_F16) ATOMA_XG(xgmi_oneshot_kernel, f16_t); else ATOMA_XG(xgmi_oneshot_kernel, float);
        } else {
            if (dtype == ATOMA_BF16) ATOMA_XG(xgmi_twoshot_kernel, bf16_t); else if (dtype == ATOMA_F16) ATOMA_XG(xgmi_twoshot_kernel, f16_t); else ATOMA_XG(xgmi_twoshot_kernel, float);
        }
#undef ATOMA_XG
        if (!ATOMA_CHECK_LAUNCH("xgmi all-reduce kernel")) return -1;
    }
    return 0;
}

// all-reduce(in) + residual -> x_out, RMSNorm(x_out) * weight -> norm_out, one launch (see the kernels above).  in: [rows, hidden] contiguous;
// mode as atoma_xgmi_allreduce_sum_mode.  The message must fit ONE launch (rows * hidden * 2 <= capacity) -- a decode step's always does.
int atoma_xgmi_allreduce_add_rms_norm(void *xg, const void *in, const void *residual, const void *weight, void *x_out, void *norm_out, int64_t rows,
                                      int64_t hidden, int64_t residual_row_stride, int64_t x_row_stride, int64_t norm_row_stride, float eps, int dtype,
                                      int mode, void *stream) {
    using namespace atoma;
    clear_error();
    auto *x = static_cast<Xgmi *>(xg);
    if (!x) { set_error("atoma_xgmi_allreduce_add_rms_norm: null communicator"); return -1; }
    if (!x->connected) { set_error("atoma_xgmi_allreduce_add_rms_norm: atoma_xgmi_connect has not been called"); return -1; }
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("atoma_xgmi_allreduce_add_rms_norm: dtype must be f16 or bf16"); return -1; }
    if (rows < 0 || hidden <= 0 || hidden % 8 || hidden > 8 * 8 * NORM_THREADS) { set_error("atoma_xgmi_allreduce_add_rms_norm: hidden must be a multiple of 8, at most 16384"); return -1; }
    if (residual_row_stride % 8 || x_row_stride % 8 || norm_row_stride % 8) { set_error("atoma_xgmi_allreduce_add_rms_norm: row strides must be multiples of 8 elements"); return -1; }
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(x_out) |
         reinterpret_cast<uintptr_t>(norm_out)) & 15u) { set_error("atoma_xgmi_allreduce_add_rms_norm: tensors must be 16-byte aligned"); return -1; }
    if (rows == 0) return 0;
    const int64_t bytes = rows * hidden * 2;
    if ((size_t)bytes > x->capacity) { set_error("atoma_xgmi_allreduce_add_rms_norm: the message exceeds the communicator's capacity (one launch only)"); return -1; }
    if (*reinterpret_cast<volatile uint32_t *>(x->status) != 0) { set_error("atoma_xgmi_allreduce_add_rms_norm: an earlier call timed out waiting for a peer (atoma_xgmi_status)"); return -1; }
    const auto s = static_cast<hipStream_t>(stream);
    if (x->world == 1)       // nothing to exchange: the sum is the input
        return atoma_add_rms_norm(residual, in, weight, x_out, norm_out, rows, hidden, residual_row_stride, hidden, x_row_stride, norm_row_stride, eps, dtype, stream);
    XgmiParams p{};
    for (int q = 0; q < x->world; ++q) p.peer[q] = x->peer[q];
    p.in = static_cast<const char *>(in);
    p.out = nullptr;
    p.seq = x->seq;
    p.status = x->status_dev;
    p.nvec = bytes / 16;
    p.chunk_vec = (p.nvec + x->world - 1) / x->world;
    p.slot_bytes = (int64_t)x->slot_bytes;
    p.off_scatter = (int64_t)x->off_scatter; p.off_gather = (int64_t)x->off_gather; p.half_bytes = (int64_t)x->half_bytes;
    p.off_flags = (int64_t)x->off_flags;
    p.timeout_ticks = x->timeout_ticks;
    p.rank = x->rank; p.world = x->world;
    XgmiNormParams n{static_cast<const uint16_t *>(residual), static_cast<const uint16_t *>(weight), static_cast<uint16_t *>(x_out), static_cast<uint16_t *>(norm_out),
                     residual_row_stride, x_row_stride, norm_row_stride, (int)rows, (int)hidden, eps};
    const bool fits_slot = (size_t)bytes <= x->slot_bytes;
    const bool oneshot = mode == 1 ? true : (mode == 2 ? false : (size_t)bytes <= x->oneshot_max && fits_slot);
    if (oneshot && !fits_slot) { set_error("atoma_xgmi_allreduce_add_rms_norm: message too large for the one-shot slots"); return -1; }
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(XGMI_MAX_BLOCKS, rows));
    const int64_t iters = cdiv(hidden / 8, NORM_THREADS);
#define ATOMA_XGN(K, TT, IT) hipLaunchKernelGGL((K<TT, IT>), dim3(blocks), dim3(NORM_THREADS), 0, s, p, n)
#define ATOMA_XGN_I(K, TT) do { if (iters <= 1) ATOMA_XGN(K, TT, 1); else if (iters <= 2) ATOMA_XGN(K, TT, 2); else if (iters <= 4) ATOMA_XGN(K, TT, 4); else ATOMA_XGN(K, TT, 8); } while (0)
    if (oneshot) { if (dtype == ATOMA_BF16) ATOMA_XGN_I(xgmi_oneshot_add_norm_kernel, bf16_t); else ATOMA_XGN_I(xgmi_oneshot_add_norm_kernel, f16_t); }
    else { if (dtype == ATOMA_BF16) ATOMA_XGN_I(xgmi_twoshot_add_norm_kernel, bf16_t); else ATOMA_XGN_I(xgmi_twoshot_add_norm_kernel, f16_t); }
#undef ATOMA_XGN_I
#undef ATOMA_XGN
    return ATOMA_CHECK_LAUNCH("xgmi all-reduce + add + rms_norm kernel") ? 0 : -1;
}

int atoma_xgmi_allreduce_sum(void *xg, const void *in, void *out, int64_t count, int dtype, void *stream) {
    return atoma_xgmi_allreduce_sum_mode(xg, in, out, count, dtype, 0, stream);
}

// The caller guarantees that no rank still has a call in flight (barrier + stream sync), as for ncclCommDestroy.
int atoma_xgmi_destroy(void *xg) {
    using namespace atoma;
    clear_error();
    auto *x = static_cast<Xgmi *>(xg);
    if (!x) return 0;
    (void)hipSetDevice(x->device);
    for (int q = 0; q < x->world; ++q)
        if (x->ipc_opened[q]) (void)hipIpcCloseMemHandle(x->peer[q]);
    if (x->status) (void)hipHostFree(x->status);
    if (x->region) (void)hipFree(x->region);
    (void)hipGetLastError();
    delete x;
    return 0;
}

}  // extern "C"
