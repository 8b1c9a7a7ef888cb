// Shared device-side pieces of the decode kernels (paged_decode.hip: 16-bit caches; paged_decode_fp8.hip: fp8 caches):
// parameters, the wavefront -> work mapping, the balanced mode, small lane helpers.  See paged_decode.hip for the design.
#pragma once
#include "attn_params.h"
#include "sync_ticket.h"
#include <type_traits>

#include <atomic>
#include <stdlib.h>

#ifndef DECODE_DEFAULT_P
#define DECODE_DEFAULT_P 3
#endif
#ifndef DECODE_DEFAULT_NT
#define DECODE_DEFAULT_NT 1
#endif

namespace atoma {

struct DecodeParams {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    float *lse;            // [b][h] or nullptr
    float *o_accum;        // [splits][b][h][D] fp32
    float *lse_accum;      // [splits][b][h]
    const int *block_table;
    const int *cu_seqlens_k;
    const int *seqused_k;
    const float *alibi_slopes;
    int64_t q_batch_stride, q_head_stride, o_batch_stride, o_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t block_table_batch_stride;
    int alibi_batch_stride;
    int page_size;         // tokens per page (multiple of 16); 0 = contiguous cache
    int b, h, h_k, g, gchunks;
    int seqlen_k;
    int is_seqlens_k_cumulative;
    int num_splits;        // KV splits per sequence (grid slots)
    int stream_waves;      // > 0: balanced mode available -- this many wavefronts share the batch's tiles evenly (decode_run_items)
    int group_tile;        // q heads per wavefront (the kernel's G)
    const float *k_scale, *v_scale;   // fp8 (e4m3fn) KV cache: per-kv-head dequantisation scales [h_k]; null for 16-bit caches
    int wg_splits;         // > 0: workgroup-merged split mode (paged_decode_wg_kernel): KV pieces per sequence = wg_splits x wavefronts per workgroup
    sync_word_t *counters; // ... and its arrival word per (sequence, kv head, q-head chunk): epoch-tagged (sync_ticket.h), no state is carried between launches
    int head_major;        // 1: workgroup id -> (kv head, chunk) slowest, (split, sequence) fastest (decode_map_work)
    int stream_force;      // the balanced line also for uniform resident batches (decode_plan_launch says when)
    int fp8_klines;        // fp8 matrix-core kernel, K in full 128-byte lines: 0 never, 1 where it pays (not on the balanced line, not with a single kv head: contiguous rows), 2 always
    int line_merge;        // balanced mode: the cut pieces of a sequence are merged by the LAST wavefront to arrive at it (counters), no combine launch
    int *plan;             // balanced mode: [0] = tiles of the whole batch, [1] = balanced mode taken, [2 .. 2+b] = exclusive prefix of tiles per sequence (b + 3 ints)
    float scale, scale_log2;
};

// all-reduce over the LPR adjacent lanes that hold one row (DPP, no LDS)
template <int LPR> __device__ __forceinline__ float row_allreduce(float x) {
    if constexpr (LPR >= 2) x += __builtin_amdgcn_update_dpp(0.f, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    if constexpr (LPR >= 4) x += __builtin_amdgcn_update_dpp(0.f, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8) x += __builtin_amdgcn_update_dpp(0.f, x, 0x141, 0xf, 0xf, true);  // row_half_mirror
    if constexpr (LPR >= 16) x += __builtin_amdgcn_update_dpp(0.f, x, 0x140, 0xf, 0xf, true); // row_mirror
    if constexpr (LPR >= 32) x += __shfl_xor(x, 16, 64);
    return x;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ uint32_t pack_pair(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack_pair<bf16_t>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));  // v_cvt_pk_bf16_f32
}
template <> __device__ __forceinline__ uint32_t pack_pair<f16_t>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_v));   // v_cvt_pk_f16_f32 (RNE)
}

// Read-only metadata (block table, lengths) through the constant address space: a wave-uniform index then always
// becomes a scalar load (s_load), also inside the segment loop of the balanced mode where the compiler would
// otherwise fall back to vector loads because output stores of the previous segment precede them.
template <typename X> __device__ __forceinline__ X load_ro(const X *ptr) {
    return *(const X __attribute__((address_space(4))) *)ptr;
}
// sequence length of batch entry b: /root/reference/csrc/kernels/block_info.h:16-23
__device__ __forceinline__ int decode_seq_len(const DecodeParams &p, int b) {
    if (p.seqused_k) return load_ro(p.seqused_k + b);
    if (p.cu_seqlens_k == nullptr) return p.seqlen_k;
    if (p.is_seqlens_k_cumulative) return load_ro(p.cu_seqlens_k + b + 1) - load_ro(p.cu_seqlens_k + b);
    return load_ro(p.cu_seqlens_k + b);
}

// Workgroup (= one wavefront) -> (sequence, kv head, q-head chunk of the group, KV split) and the tile range it
// owns.  The split index is slowest: the dispatcher places workgroups on CUs round-robin by index, so
// wavefronts that exit at once must not be interleaved with the working ones -- measured 2.5x slower with the
// split index in the middle (only every 4th CU of an XCD got work).
constexpr int DECODE_PAIR_MAX_B = 512;   // paged_decode_pair_kernel ranks the batch's lengths in LDS
struct DecodeWork {
    int b, hk, gc, split;
    int L, n_tiles, t0, t1;
    int64_t kv_row0;   // first row of this sequence in a varlen (cumulative) K/V tensor
    int64_t prow;      // row of q head hq0's fp32 partial in o_accum / lse_accum (head hq0 + i: prow + i)
    bool partial;      // write fp32 partials for the combine kernel (true) or the final output (false)
    bool balanced;     // this piece is a range of the balanced line (the fp8 kernel fetches K differently there)
    float *sink_o, *sink_lse;   // SINK variants of the item functions: this wavefront's normalised O [heads][D] and LSE [heads] go here (LDS)
};
__device__ __forceinline__ void decode_map_work(const DecodeParams &p, int id, DecodeWork &w, bool allow_head_major = true) {
    const int hk_chunks = p.h_k * p.gchunks;
    int hkc;
    if (allow_head_major && p.head_major) {
        // kv-head major, as the balanced line: the workgroups an XCD receives back to back are the same head of consecutive pieces, so
        // the wavefronts that share a CU are the kv heads of ONE piece (exactly so when pieces / 8 is a multiple of the 32 CUs of an XCD)
        const int pieces = p.b * p.num_splits;
        hkc = id / pieces;
        id -= hkc * pieces;
    } else {
        hkc = id % hk_chunks;
        id /= hk_chunks;
    }
    w.b = id % p.b;
    w.split = id / p.b;
    w.hk = hkc / p.gchunks;
    w.gc = hkc % p.gchunks;
    w.L = decode_seq_len(p, w.b);
    w.kv_row0 = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? load_ro(p.cu_seqlens_k + w.b) : 0;
    w.n_tiles = (w.L + 15) >> 4;
    const int per = (w.n_tiles + p.num_splits - 1) / p.num_splits;
    w.t0 = w.split * per;
    w.t1 = min(w.t0 + per, w.n_tiles);
    w.partial = p.num_splits > 1;
    w.balanced = false;
    w.prow = ((int64_t)w.split * p.b + w.b) * p.h + w.hk * p.g + w.gc * p.group_tile;
}

// Balanced ("stream") mode for batches that fill the chip without KV splitting (b . h_k >= resident wavefronts / 2)
// and whose lengths live on the device.  Why: the bandwidth of this kernel against the number of ACTIVE wavefronts
// is concave (tools/probes/decode_curve.py: 1024 wavefronts reach 82 % of what 2048 do, 512 reach 52 %), so one
// wavefront per (sequence, kv head) spends the second half of a ragged launch below the HBM rate, a single long
// straggler streams alone at ~6 GB/s, and a batch of 1.2 x the resident wavefronts takes two rounds.  Instead all
// tiles of the batch are laid on one line -- position = (kv head, q-head chunk) . T + prefix[b] + tile, T = tiles of
// the batch -- and each of W wavefronts takes the same number of consecutive tiles (ceil(total / W), at least
// DECODE_MIN_SHARE).  A wavefront's range covers the end of one sequence, whole sequences, and the beginning of
// one more: whole sequences are written directly, the (at most two) cut pieces go to fp32 partial slots
// [wavefront][first / last] and decode_combine_kernel merges the pieces of each cut sequence.  No atomics, no
// queue, deterministic; every wavefront finishes at the same time by construction.
// Uniform batches that are resident at once keep the one-wavefront-per-sequence path (same kernel, no partials).
#define DECODE_STREAM_MAX_B 1024
#define DECODE_MIN_SHARE 8
struct DecodePlan {
    int T;          // tiles of the batch (one kv head)
    int share;      // tiles per wavefront
    bool stream;    // balanced mode taken
};
// Prefix of tiles per sequence into LDS (cum[b], cum[p.b] = T); every wavefront computes the same plan.
__device__ __forceinline__ DecodePlan decode_make_plan(const DecodeParams &p, int *cum) {
    const int lane = threadIdx.x & 63;
    const int per = (p.b + 63) >> 6;
    int local = 0, mx = 0;
    for (int j = 0; j < per; ++j) {
        const int b = lane * per + j;
        const int n = b < p.b ? (decode_seq_len(p, b) + 15) >> 4 : 0;
        local += n;
        mx = max(mx, n);
    }
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
        mx = max(mx, __shfl_xor(mx, off, 64));
    }
    int run = incl - local;
    for (int j = 0; j < per; ++j) {
        const int b = lane * per + j;
        if (b < p.b) {
            cum[b] = run;
            run += (decode_seq_len(p, b) + 15) >> 4;
        }
    }
    mx = __builtin_amdgcn_readfirstlane(mx);   // every lane holds the maximum: make it a scalar for the compiler
    DecodePlan pl;
    pl.T = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) cum[p.b] = pl.T;
    const int hk_chunks = p.h_k * p.gchunks;
    const int64_t total = (int64_t)pl.T * hk_chunks;
    pl.share = (int)max((total + p.stream_waves - 1) / p.stream_waves, (int64_t)DECODE_MIN_SHARE);
    // When do equal shares (= cut sequences) pay?  A cut costs ~4.5 % of the launch whatever the spread (two start-up chains, fp32 partials, a
    // merge per sequence); one wavefront per (sequence, kv head) in the same kv-head-major order costs the idle tail of the short
    // sequences.  Measured round 5 (profiles/r05_decode_narrow_spread_ab.txt, B = 256, 8 kv heads): lengths U[3800,4096] (idle share
    // 1 - mean/max = 4 %) 0.648 ms cut / 0.627 whole; U[2048,2560) (10 %, the contexts of the configs[2] step) 0.397 / 0.385;
    // U[2048,4096] (25 %) 0.521 / 0.552.  So: cut from 15 % idle share on, or when the batch has more (sequence, kv head) items than
    // resident wavefronts; an exactly uniform batch is the same launch either way (no share boundary falls inside a sequence).
    const bool ragged = (int64_t)mx * p.b * 85 > (int64_t)pl.T * 100;
    const bool uniform = (int64_t)mx * p.b == (int64_t)pl.T;
    pl.stream = ragged || p.b * hk_chunks > p.stream_waves || p.stream_force == 2 || (p.stream_force && uniform);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS writes above
    __builtin_amdgcn_wave_barrier();
    return pl;
}

// fp32 partials (split-KV pieces, cut pieces of the balanced line) are published WRITE-THROUGH (agent-scope stores: past the XCD's
// L2), so that a wavefront on another XCD -- the last arriver of decode_line_merge, or the combine kernel -- reads them with
// agent-scope loads and no fence.
// One 16-byte write-through store (sc1, what the compiler emits for an agent-scope atomic store, at twice the width the atomics allow): the
// counters of round 5 showed 34 MiB written per ragged launch for ~10 MiB of pieces -- every 8-byte store its own 32-byte write request
// (profiles/r05_decode_counters.json).  The readers' accesses are unchanged (8-byte agent-scope loads of the same bytes).
__device__ __forceinline__ void partial_store(float *dst, float a, float b, float c, float d) {
    typedef float f32x4_st __attribute__((ext_vector_type(4)));
    const f32x4_st v = {a, b, c, d};
    // (s_nop 1: a VMEM store of more than 64 bits reads its data registers late -- a VALU write to them needs a wait state in between, and the
    //  compiler's hazard recognizer does not look inside an asm string; without it the first GPU run stored overwritten registers)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void partial_store(float *dst, float a) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(dst), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float partial_load(const float *src) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Balanced line, merge inside the launch (the uniform headline launched a combine kernel that found nothing to merge, a
// ragged batch paid a second launch per layer).  A sequence cut between the wavefronts w0 .. w0 + nsp - 1 of the line has one arrival
// counter, index w0 (a wavefront's range ends inside at most one sequence, so w0 names the sequence); every piece is published write-through, the wavefront takes a ticket, and the one that arrives LAST merges the pieces
// in line order with decode_combine_kernel's arithmetic (flash_fwd_kernel.h:1204-1236) and writes the output.  Nobody waits; the result
// does not depend on the order of arrival; the arrival word is epoch-tagged (sync_ticket.h: nothing to reset, nothing inherited).  Not inlined: the call sits in the segment loop, whose scalar
// registers are the tight resource of these kernels.
template <typename T, int D>
__device__ __attribute__((noinline)) void decode_line_merge(const DecodeParams *pp, int b, int hkc, int w0, int nsp, int first_slot) {
    const DecodeParams &p = *pp;
    const int lane = threadIdx.x & 63;
    const int hk = hkc / p.gchunks, gc = hkc - hk * p.gchunks;
    const int nq = min(p.group_tile, p.g - gc * p.group_tile), hq0 = hk * p.g + gc * p.group_tile;
    auto prow = [&](int gq, int s) -> int64_t { return ((int64_t)(w0 + s) * 2 + (s == 0 ? first_slot : 0)) * p.group_tile + gq; };
    // The merge is three dependent trips to memory (ticket, LSEs, rows) on the critical path of the launch's last wavefronts, so every trip
    // fetches as much as it can: four q heads at a time; LSEs: 16 lanes per head, one piece each; rows: D / 4 lanes per row (four floats
    // each) and 64 / (D / 4) pieces side by side, summed in a fixed order (pieces sub, sub + PP, .. per lane group, then the groups).
    constexpr int LPR = D / 4, PP = 64 / LPR;
    const int col = lane % LPR, sub = lane / LPR;
    for (int g0 = 0; g0 < nq; g0 += 4) {
        const int gql = min(g0 + (lane >> 4), nq - 1), sl = lane & 15;
        float mx = -INFINITY;
        for (int s = sl; s < nsp; s += 16) mx = fmaxf(mx, partial_load(p.lse_accum + prow(gql, s)));
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float ms = mx == -INFINITY ? 0.f : mx;
        float tot = 0.f;
        for (int s = sl; s < nsp; s += 16) tot += __expf(partial_load(p.lse_accum + prow(gql, s)) - ms);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) tot += __shfl_xor(tot, off, 64);
        const float lse_l = tot > 0.f ? __logf(tot) + ms : INFINITY;
        float lse[4], acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lse[u] = __shfl(lse_l, 16 * u, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
        }
        for (int s0 = 0; s0 < nsp; s0 += PP) {
            const int sp = s0 + sub;
            const bool on = sp < nsp;
            float pl[4];
            unsigned long long v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                  // all twelve loads leave before the first is needed
                const int64_t row = prow(min(g0 + u, nq - 1), on ? sp : 0);
                pl[u] = partial_load(p.lse_accum + row);
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.o_accum + row * D + col * 4);
                v[u][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[u][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = (on && lse[u] != INFINITY) ? __expf(pl[u] - lse[u]) : 0.f;
                acc[u][0] += w * __uint_as_float((unsigned)v[u][0]);
                acc[u][1] += w * __uint_as_float((unsigned)(v[u][0] >> 32));
                acc[u][2] += w * __uint_as_float((unsigned)v[u][1]);
                acc[u][3] += w * __uint_as_float((unsigned)(v[u][1] >> 32));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][e] += __shfl_xor(acc[u][e], off, 64);
            if (g0 + u < nq && sub == 0) {
                const int hq = hq0 + g0 + u;
                uint2 o;
                o.x = pack2<T>(acc[u][0], acc[u][1]);
                o.y = pack2<T>(acc[u][2], acc[u][3]);
                *reinterpret_cast<uint2 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + col * 4) = o;
                if (p.lse && col == 0) p.lse[(int64_t)b * p.h + hq] = lse[u];
            }
        }
    }
}
// ... and the sequences without a single tile, which no wavefront of the line ever sees (flash_fwd_kernel.h:543-582: O = 0, LSE = +inf)
template <int D> __device__ __attribute__((noinline)) void decode_line_zero(const DecodeParams *pp, int b) {
    const DecodeParams &p = *pp;
    const int lane = threadIdx.x & 63;
    for (int hq = 0; hq < p.h; ++hq) {
        uint16_t *dst = p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride;
        for (int e = lane; e < D; e += 64) dst[e] = 0;
        if (p.lse && lane == 0) p.lse[(int64_t)b * p.h + hq] = INFINITY;
    }
}
struct DecodeNoMerge {
    __device__ __forceinline__ void merge(const DecodeParams *, int, int, int, int, int) const {}
    __device__ __forceinline__ void zero(const DecodeParams *, int) const {}
};
template <typename T, int D> struct DecodeLineMerge {
    __device__ __forceinline__ void merge(const DecodeParams *p, int b, int hkc, int w0, int nsp, int first_slot) const { decode_line_merge<T, D>(p, b, hkc, w0, nsp, first_slot); }
    __device__ __forceinline__ void zero(const DecodeParams *p, int b) const { decode_line_zero<D>(p, b); }
};

// `item(p, wk)` is inlined exactly once.  In the balanced variant the kernel arguments are re-read through a pointer
// the compiler cannot see through at the top of every segment: hoisting every field of DecodeParams out of the
// segment loop costs ~20 SGPRs more than the 102 there are and the spills (v_readlane in the tile loop) cost 5 %.
// NWG wavefronts per workgroup (default 1): wavefront index = blockIdx.x * NWG + wave.  Consecutive indices are the kv heads
// of ONE sequence, so a workgroup of NWG wavefronts reads NWG adjacent head slices of every token row from one CU at about
// the same time -- with 128-byte slices (fp8 cache, or d = 64 at 16 bits) a lone wavefront fetches half of a 256-byte
// DRAM granule and its neighbour, dispatched to another XCD (block index % 8), fetches the other half some time later.
template <bool STREAM, int NWG = 1, typename F, typename M = DecodeNoMerge> __device__ __forceinline__ void decode_run_items(const DecodeParams &p0, F &&item, M merger = M{}) {
    DecodeWork wk;
    const int wid = NWG == 1 ? (int)blockIdx.x : (int)blockIdx.x * NWG + (int)(threadIdx.x >> 6);
    if (NWG > 1 && (int64_t)wid >= (int64_t)p0.b * p0.num_splits * p0.h_k * p0.gchunks) return;
    if constexpr (!STREAM) {
        decode_map_work(p0, wid, wk, NWG == 1);
        item(p0, wk);
    } else {
        __shared__ int cum[DECODE_STREAM_MAX_B + 1];
        __shared__ sync_word_t s_epoch;     // this launch's epoch for the arrival tickets; parked in LDS: scalar registers are the tight resource here
        if ((threadIdx.x & 63) == 0) s_epoch = sync_epoch();
        const DecodePlan pl = decode_make_plan(p0, cum);
        if (wid == 0) {   // for the combine kernel
            for (int i = threadIdx.x & 63; i <= p0.b; i += 64) p0.plan[2 + i] = cum[i];
            if ((threadIdx.x & 63) == 0) { p0.plan[0] = pl.T; p0.plan[1] = pl.stream ? 1 : 0; }
        }
        const bool stream = pl.stream;
        // position of this wavefront on the line.  The line is kv-head major, so wavefronts w and w + W / 8 walk the same
        // sequences of adjacent heads: with 8 wavefronts per workgroup, give the 8 of a workgroup those ranges (W is a multiple of 8)
        if (stream && wid >= p0.stream_waves) return;      // the line is shared by stream_waves wavefronts; the grid may hold more
        // The out-of-line merge / zero routines take the kernel arguments BY ADDRESS: always the kernarg segment's, never &p0 -- the
        // address of the by-value parameter makes the compiler copy all 288 bytes of DecodeParams to scratch in the prologue of EVERY
        // wavefront (18 KiB per wavefront, 36 MiB of HBM writes per headline launch and scratch loads on the start-up chain: the
        // round-4 regression 0.83 -> 0.80, found in round 5 from WRITE_SIZE and `private_segment_fixed_size: 304` in the metadata).
        typedef const DecodeParams __attribute__((address_space(4))) *KernArg;
        KernArg kp = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
        if (stream && p0.line_merge == 1) {
            for (int e = wid; e < p0.b; e += p0.stream_waves)
                if (__builtin_amdgcn_readfirstlane(cum[e]) == __builtin_amdgcn_readfirstlane(cum[e + 1])) merger.zero((const DecodeParams *)kp, e);
        }
        const int lw = NWG == 1 ? wid : (wid % NWG) * (p0.stream_waves / NWG) + wid / NWG;
        int pos = 0, end = 1, hkc = 0, r = 0, b = 0;       // host guarantees total < 2^31
        bool first = true;
        if (stream) {
            const int64_t total = (int64_t)pl.T * p0.h_k * p0.gchunks;
            const int64_t start = (int64_t)lw * pl.share;
            if (start >= total) return;
            pos = (int)start;
            end = (int)min(start + pl.share, total);
            hkc = pos / pl.T;
            r = pos - hkc * pl.T;
            int lo = 0, hi = p0.b;               // largest b with cum[b] <= r (the last of equal entries: empty sequences own no tile)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (__builtin_amdgcn_readfirstlane(cum[mid]) <= r) lo = mid; else hi = mid;   // LDS values are wave-uniform here
            }
            b = lo;
        }
        for (;;) {
            asm volatile("" : "+s"(kp));
            const DecodeParams &p = *(const DecodeParams *)kp;
            int seg = 1;
            if (stream) {
                const int c0 = __builtin_amdgcn_readfirstlane(cum[b]), c1 = __builtin_amdgcn_readfirstlane(cum[b + 1]);
                seg = min(c1 - r, end - pos);
                wk.b = b;
                wk.hk = hkc / p.gchunks;
                wk.gc = hkc % p.gchunks;
                wk.split = 0;
                wk.L = decode_seq_len(p, b);
                wk.n_tiles = c1 - c0;
                wk.t0 = r - c0;
                wk.t1 = wk.t0 + seg;
                wk.kv_row0 = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? load_ro(p.cu_seqlens_k + b) : 0;
                wk.partial = seg != wk.n_tiles;
                wk.balanced = true;
                wk.prow = ((int64_t)lw * 2 + (first ? 0 : 1)) * p.group_tile;
            } else {
                decode_map_work(p, wid, wk, NWG == 1);   // one wavefront per (sequence, kv head), final output
            }
            item(p, wk);
            if (!stream) break;
            if (wk.partial && p.line_merge == 1) {
                // w0 / w1: the wavefronts of the line that hold the first / last tile of this sequence
                const int c0 = __builtin_amdgcn_readfirstlane(cum[b]);
                const int64_t s0 = (int64_t)hkc * pl.T + c0;
                const int w0 = (int)(s0 / pl.share), w1 = (int)((s0 + wk.n_tiles - 1) / pl.share);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this piece is out
                unsigned t = 0;
                if ((threadIdx.x & 63) == 0) t = sync_arrive(p.counters + w0, s_epoch, (unsigned)(w1 - w0 + 1));
                t = __builtin_amdgcn_readfirstlane(t);
                if (t == (unsigned)(w1 - w0)) merger.merge(&p, b, hkc, w0, w1 - w0 + 1, s0 > (int64_t)w0 * pl.share ? 1 : 0);
            }
            first = false;
            pos += seg;
            r += seg;
            if (pos >= end) break;
            while (r == __builtin_amdgcn_readfirstlane(cum[b + 1])) {   // next sequence that owns tiles (or the next kv head's line)
                if (++b == p.b) { b = 0; r = 0; ++hkc; }
            }
        }
    }
}

template <int I, int N, typename F> __device__ __forceinline__ void decode_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        decode_static_for<I + 1, N>(f);
    }
}
// value of lane `H` of this lane's 16-lane row
template <int H> __device__ __forceinline__ uint32_t row_bcast(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + H, 0xf, 0xf, false);   // row_newbcast:H
}
template <int H> __device__ __forceinline__ float row_bcastf(float x) { return __uint_as_float(row_bcast<H>(__float_as_uint(x))); }

typedef __attribute__((ext_vector_type(4))) float f32x4_v;
template <typename T> __device__ __forceinline__ f32x4_v mfma16(const u32x4 &a, const u32x4 &b, f32x4_v c);
template <> __device__ __forceinline__ f32x4_v mfma16<bf16_t>(const u32x4 &a, const u32x4 &b, f32x4_v c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_v mfma16<f16_t>(const u32x4 &a, const u32x4 &b, f32x4_v c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}

// v_mfma_f32_16x16x16: lane (grp, col) supplies A[row col][k = 4 grp ..+3] and B[k = 4 grp ..+3][column col], holds D[row 4 grp + i][column col]
template <typename T> __device__ __forceinline__ f32x4_v mfma16k16(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f32x4_v c);
template <> __device__ __forceinline__ f32x4_v mfma16k16<bf16_t>(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f32x4_v c) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u2{a0, a1}), __builtin_bit_cast(s16x4, u2{b0, b1}), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_v mfma16k16<f16_t>(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f32x4_v c) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, u2{a0, a1}), __builtin_bit_cast(h16x4, u2{b0, b1}), c, 0, 0, 0);
}
// max over the four lanes (grp = 0..3) that share `col`
__device__ __forceinline__ float col_max4(float x) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

}  // namespace atoma
