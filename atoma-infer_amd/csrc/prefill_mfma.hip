// FlashAttention-2 style prefill for gfx950 on the MFMA matrix cores: dense / varlen / paged
// K-V, causal or not, GQA, head_dim 64 or 128, bf16 or f16.
//
// Replaces /root/reference/csrc/kernels/flash_fwd_kernel.h:56-500 (compute_attn_1rowblock, the
// CUTLASS sm80 kernel behind csrc::flash_attn_varlen, csrc/src/lib.rs:606-1103) and the
// seqlen_q > 1 use of the paged split-KV kernel (flash_fwd_kernel.h:504-1092, prefix / chunked
// prefill: csrc/src/lib.rs:1392-1420).  Same numerics contract: exp2-domain online softmax,
// fp32 accumulation, P rounded to the storage dtype before P.V (softmax.h:65-185).
//
// Shape of the computation (wave64, v_mfma_f32_32x32x16_{bf16,f16}):
//  * workgroup = W waves (4: 128 query rows, two workgroups per CU; or 8: 256 rows, one per CU) of one
//    (sequence, q head); wave w owns 32 rows;
//  * "swapped" products: S^T = K.Q^T and O^T = V^T.P^T, so a lane's accumulator registers all
//    belong to ONE query row (column l&31): the online softmax (row max / sum / rescale of O)
//    is lane-local, with a single lane <-> lane+32 exchange per K/V tile for the row max;
//  * the C layout of S^T (keys (r&3)+8(r>>2)+4(l>>5)) is used directly as the k-slot order of
//    the P^T operand, and the V^T operand is fetched with ds_read_b64_tr_b16 in that same key
//    order -- P never leaves registers and needs no lane permutation;
//  * K/V tiles of 64 keys are staged through LDS (shared by the waves) in a 2- or 3-deep ring by
//    direct global->LDS DMA (global_load_lds_dwordx4): later tiles are in flight during the MFMAs
//    of tile t, counted vmcnt wait, one barrier per tile; 16-byte chunks are XOR-swizzled per row (on the
//    DMA's source address) so that the K reads (ds_read_b128, one key row per lane) and the V
//    transpose reads are bank-conflict free;
//  * causal: K/V tiles above the diagonal are never loaded; a wave skips tiles that are
//    entirely masked for its own 32 rows; masking code runs only on diagonal / tail tiles;
//  * workgroups are issued longest-first (last query block first) for causal balance, and all
//    workgroups that share a kv head's K/V are steered to the same XCD (L2 reuse).
// MFMA-bound: 4*Lq*Lk*d flops per (sequence, head) (half of it when causal).
#include "attn_params.h"
#include "prefill_map.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <type_traits>

namespace atoma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_v;
typedef __attribute__((ext_vector_type(4))) short short4_v;
typedef __attribute__((ext_vector_type(16))) float f32x16_v;
typedef __attribute__((ext_vector_type(2))) float f32x2_v;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_v;

template <typename T> __device__ __forceinline__ f32x16_v mfma32(const uint4 &a, const uint4 &b, f32x16_v c);
template <> __device__ __forceinline__ f32x16_v mfma32<bf16_t>(const uint4 &a, const uint4 &b, f32x16_v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_v mfma32<f16_t>(const uint4 &a, const uint4 &b, f32x16_v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
}
// Lanes l and l + 32 hold the two halves of a query row's keys.  v_permlane32_swap hands every lane both members of
// its pair in the VALU (tools/probes/permlane_probe.hip); __shfl_xor(x, 32) would go through the LDS (ds_bpermute) and
// its lgkmcnt(0) wait once per tile.
__device__ __forceinline__ float pair_max(float x) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <typename T> __device__ __forceinline__ uint32_t cvt_pk(float lo, float hi);
template <> __device__ __forceinline__ uint32_t cvt_pk<bf16_t>(float lo, float hi) {
    f32x2_v v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}
template <> __device__ __forceinline__ uint32_t cvt_pk<f16_t>(float lo, float hi) {
    f32x2_v v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_v));
}

#ifndef PREFILL_DEFAULT_CFG
#define PREFILL_DEFAULT_CFG 4
#endif
constexpr int PF_BN = 64;              // keys per K/V tile
// The running row max is only raised when the new tile's max exceeds it by more than PF_DEFER (log2 units):
// P = exp2(S - m) then stays <= 2^PF_DEFER (exact in the f32 sums, representable in f16 / bf16) and the
// rescale of O and of the row sums becomes rare instead of per-tile.
constexpr float PF_DEFER = 8.f;
// -DPF_TIMING: per-phase cycle accounting (s_memtime) of wave 0 of every workgroup, written to p.lse as
// [workgroup][8] floats (wait+barrier, dma issue, qk, softmax, pv, tiles, total, -) -- tools/probes/prefill_phases.py
#ifdef PF_TIMING
#ifndef PF_TIMING_TID
#define PF_TIMING_TID 0
#endif
#define PF_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += (float)(now_ - tlast); tlast = now_; } while (0)
#else
#define PF_T(i) do {} while (0)
#endif

template <int D> struct PfSwz {
    static constexpr int CPR = D / 8;  // 16-byte chunks per row
    // K tile, read one key row per lane with ds_read_b128: 16 consecutive rows must hit 16
    // different 16-byte slots of the 256-byte bank row.
    __device__ static __forceinline__ int k(int row, int chunk) {
        return D == 128 ? (chunk ^ (row & 15)) : (chunk ^ ((row >> 1) & 7));
    }
    // V tile, read with ds_read_b64_tr_b16: a 32-lane group touches 4 consecutive key rows x 64 bytes.
    __device__ static __forceinline__ int v(int row, int chunk) {
        return D == 128 ? (chunk ^ ((row & 3) << 2)) : (chunk ^ (((row >> 1) & 1) << 2));
    }
};

// One 1 KiB global->LDS DMA (global_load_lds_dwordx4): LDS destination = wave-uniform byte
// address in M0 + lane*16, per-lane global source.  Issued from inline asm so that hipcc does
// not treat every later ds_read as dependent on it (it would drain the DMA with vmcnt(0) before
// the first LDS read of the tile being computed); the consumer side waits explicitly with
// dma_wait_all() ahead of the workgroup barrier.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}
// same with a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset (no per-lane 64-bit math)
__device__ __forceinline__ void glds16_saddr(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's DMAs are still in flight (they complete in issue order)
template <int N> __device__ __forceinline__ void dma_wait_keep() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// K/V tile loader: direct global -> LDS DMA (1 KiB per wave instruction, no staging registers).
// The DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle is applied on the
// SOURCE side: lane -> (row, slot) of the LDS image, chunk = slot ^ f(row) of the global row.
// Everything that depends only on the lane is computed once (init); a tile of a contiguous K/V
// tensor that lies fully inside the sequence costs no vector ALU at all (uniform base in SGPRs).
template <int D, int W> struct PfLoader {
    static constexpr int CPR = D / 8;
    static constexpr int ROWB = D * 2;
    static constexpr int TILEB = PF_BN * ROWB;
    static constexpr int NDMA = TILEB / 1024 / W;  // DMA instructions per wave per tile (K or V)
    const uint16_t *kbase, *vbase;
    const int *bt;
    int64_t k_page, k_row, v_page, v_row;
    int page_size, page_shift, last_key, wave, lane;
    uint32_t kfast[NDMA], vfast[NDMA];     // byte offset from the tile base, contiguous layout (the common case: kept in registers)

    // DMA piece u of this wave: the 16-byte unit of the LDS image this lane fills -> tile row and the element
    // offset of the (de-swizzled) source chunk inside the row (XOR is its own inverse); recomputed where needed
    // rather than held in registers across the tile loop
    __device__ __forceinline__ int row_of(int u) const { return ((wave * NDMA + u) * 64 + lane) / CPR; }
    __device__ __forceinline__ uint32_t kchunk_of(int u) const { return PfSwz<D>::k(row_of(u), ((wave * NDMA + u) * 64 + lane) % CPR) * 8; }
    __device__ __forceinline__ uint32_t vchunk_of(int u) const { return PfSwz<D>::v(row_of(u), ((wave * NDMA + u) * 64 + lane) % CPR) * 8; }

    __device__ __forceinline__ void init(int lane_) {
        lane = lane_;
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            kfast[u] = (uint32_t)(row_of(u) * k_row * 2) + kchunk_of(u) * 2;
            vfast[u] = (uint32_t)(row_of(u) * v_row * 2) + vchunk_of(u) * 2;
        }
    }

    // Issue the DMAs of K tile `ktile` -> LDS byte address k_lds and of V tile `vtile` -> v_lds
    // (either may be < 0: skipped; the conditions are workgroup-uniform).
    __device__ __forceinline__ void issue2(int ktile, uint32_t k_lds, int vtile, uint32_t v_lds) const {
        const bool do_k = ktile >= 0, do_v = vtile >= 0;
        const int kkey0 = ktile * PF_BN, vkey0 = vtile * PF_BN;
        if (!bt && (!do_k || kkey0 + PF_BN - 1 <= last_key) && (!do_v || vkey0 + PF_BN - 1 <= last_key)) {
            // contiguous tensor, full tiles: uniform base in SGPRs + the lane's constant offset
            const uint64_t kb = uniform64((uint64_t)(kbase + (int64_t)kkey0 * k_row));
            const uint64_t vb = uniform64((uint64_t)(vbase + (int64_t)vkey0 * v_row));
#pragma unroll
            for (int u = 0; u < NDMA; ++u) {
                const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + u) * 1024));
                if (do_k) glds16_saddr(kb, kfast[u], k_lds + off);
                if (do_v) glds16_saddr(vb, vfast[u], v_lds + off);
            }
            return;
        }
        if (bt && page_shift >= 0 && (!do_k || kkey0 + PF_BN - 1 <= last_key) && (!do_v || vkey0 + PF_BN - 1 <= last_key)) {
            // paged cache, full tiles: a 1 KiB piece covers 64 / CPR consecutive rows, which never straddle a page
            // (pages are multiples of 16 tokens), so its page id is wave-uniform: scalar loads for the block table
            // (lgkmcnt -- the DMA queue is not drained) and, as above, a uniform base + the lane's constant offset
            constexpr int RPP = 64 / CPR;   // rows per piece
            uint64_t kb[NDMA], vb[NDMA];
            if (page_shift >= 6) {          // pages of 64+ tokens: the whole 64-key tile sits in one page
                const int kpi = max(kkey0, 0) >> page_shift, vpi = max(vkey0, 0) >> page_shift;
                const int kpg = __builtin_amdgcn_readfirstlane(bt[kpi]), vpg = __builtin_amdgcn_readfirstlane(bt[vpi]);
                const uint64_t kb0 = uniform64((uint64_t)(kbase + (int64_t)kpg * k_page + (int64_t)(kkey0 - (kpi << page_shift)) * k_row));
                const uint64_t vb0 = uniform64((uint64_t)(vbase + (int64_t)vpg * v_page + (int64_t)(vkey0 - (vpi << page_shift)) * v_row));
#pragma unroll
                for (int u = 0; u < NDMA; ++u) { kb[u] = kb0; vb[u] = vb0; }
            } else {
#pragma unroll
                for (int u = 0; u < NDMA; ++u) {
                    const int row0 = __builtin_amdgcn_readfirstlane((wave * NDMA + u) * RPP);
                    if (do_k) {
                        const int pi = (kkey0 + row0) >> page_shift;
                        const int pg = __builtin_amdgcn_readfirstlane(bt[pi]);
                        kb[u] = uniform64((uint64_t)(kbase + (int64_t)pg * k_page + (int64_t)(kkey0 - (pi << page_shift)) * k_row));
                    }
                    if (do_v) {
                        const int pi = (vkey0 + row0) >> page_shift;
                        const int pg = __builtin_amdgcn_readfirstlane(bt[pi]);
                        vb[u] = uniform64((uint64_t)(vbase + (int64_t)pg * v_page + (int64_t)(vkey0 - (pi << page_shift)) * v_row));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NDMA; ++u) {
                const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + u) * 1024));
                if (do_k) glds16_saddr(kb[u], kfast[u], k_lds + off);
                if (do_v) glds16_saddr(vb[u], vfast[u], v_lds + off);
            }
            return;
        }
        const uint16_t *ksrc[NDMA], *vsrc[NDMA];
        // all block-table lookups first, then the DMAs back to back (a lookup's vmcnt wait
        // would otherwise drain the DMA issued just before it)
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int kkey = min(max(kkey0, 0) + row_of(u), last_key);  // never read past the sequence
            const int vkey = min(max(vkey0, 0) + row_of(u), last_key);
            int64_t koff, voff;
            if (bt) {
                const int kpi = page_shift >= 0 ? kkey >> page_shift : kkey / page_size;
                const int vpi = page_shift >= 0 ? vkey >> page_shift : vkey / page_size;
                const int kpg = bt[kpi], vpg = bt[vpi];
                koff = (int64_t)kpg * k_page + (int64_t)(kkey - kpi * page_size) * k_row;
                voff = (int64_t)vpg * v_page + (int64_t)(vkey - vpi * page_size) * v_row;
            } else {
                koff = (int64_t)kkey * k_row;
                voff = (int64_t)vkey * v_row;
            }
            ksrc[u] = kbase + koff + kchunk_of(u);
            vsrc[u] = vbase + voff + vchunk_of(u);
        }
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + u) * 1024));
            if (do_k) glds16(ksrc[u], k_lds + off);
            if (do_v) glds16(vsrc[u], v_lds + off);
        }
    }
    // The same two fast routes, split into "bases now, pieces later": plan2() computes the wave-uniform source base of every
    // piece of K tile `ktile` / V tile `vtile` (scalar work + block-table lookups) and says whether the tiles qualify;
    // piece_k<U>() / piece_v<U>() then issue ONE DMA each, wherever the caller's instruction stream has room for it.
    __device__ __forceinline__ bool plan2(int ktile, int vtile, uint64_t (&kb)[NDMA], uint64_t (&vb)[NDMA]) const {
        const bool do_k = ktile >= 0, do_v = vtile >= 0;
        const int kkey0 = ktile * PF_BN, vkey0 = vtile * PF_BN;
        if ((do_k && kkey0 + PF_BN - 1 > last_key) || (do_v && vkey0 + PF_BN - 1 > last_key)) return false;
        if (!bt) {
            const uint64_t kb0 = uniform64((uint64_t)(kbase + (int64_t)kkey0 * k_row));
            const uint64_t vb0 = uniform64((uint64_t)(vbase + (int64_t)vkey0 * v_row));
#pragma unroll
            for (int u = 0; u < NDMA; ++u) { kb[u] = kb0; vb[u] = vb0; }
            return true;
        }
        if (page_shift < 0) return false;
        constexpr int RPP = 64 / CPR;   // rows per piece
        if (page_shift >= 6) {
            const int kpi = max(kkey0, 0) >> page_shift, vpi = max(vkey0, 0) >> page_shift;
            const int kpg = __builtin_amdgcn_readfirstlane(bt[kpi]), vpg = __builtin_amdgcn_readfirstlane(bt[vpi]);
            const uint64_t kb0 = uniform64((uint64_t)(kbase + (int64_t)kpg * k_page + (int64_t)(kkey0 - (kpi << page_shift)) * k_row));
            const uint64_t vb0 = uniform64((uint64_t)(vbase + (int64_t)vpg * v_page + (int64_t)(vkey0 - (vpi << page_shift)) * v_row));
#pragma unroll
            for (int u = 0; u < NDMA; ++u) { kb[u] = kb0; vb[u] = vb0; }
            return true;
        }
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int row0 = __builtin_amdgcn_readfirstlane((wave * NDMA + u) * RPP);
            kb[u] = vb[u] = 0;
            if (do_k) {
                const int pi = (kkey0 + row0) >> page_shift;
                const int pg = __builtin_amdgcn_readfirstlane(bt[pi]);
                kb[u] = uniform64((uint64_t)(kbase + (int64_t)pg * k_page + (int64_t)(kkey0 - (pi << page_shift)) * k_row));
            }
            if (do_v) {
                const int pi = (vkey0 + row0) >> page_shift;
                const int pg = __builtin_amdgcn_readfirstlane(bt[pi]);
                vb[u] = uniform64((uint64_t)(vbase + (int64_t)pg * v_page + (int64_t)(vkey0 - (pi << page_shift)) * v_row));
            }
        }
        return true;
    }
    template <int U> __device__ __forceinline__ void piece_k(uint64_t base, uint32_t k_lds) const {
        glds16_saddr(base, kfast[U], k_lds + __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + U) * 1024)));
    }
    template <int U> __device__ __forceinline__ void piece_v(uint64_t base, uint32_t v_lds) const {
        glds16_saddr(base, vfast[U], v_lds + __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + U) * 1024)));
    }
    // contiguous K / V tensor, tile fully inside the sequence: piece u of this wave's share, from a
    // wave-uniform tile base (no per-lane address arithmetic)
    __device__ __forceinline__ bool fast_tile(int tile) const { return !bt && tile * PF_BN + PF_BN - 1 <= last_key; }
    __device__ __forceinline__ uint64_t k_tile_base(int tile) const { return uniform64((uint64_t)(kbase + (int64_t)tile * PF_BN * k_row)); }
    __device__ __forceinline__ uint64_t v_tile_base(int tile) const { return uniform64((uint64_t)(vbase + (int64_t)tile * PF_BN * v_row)); }
    template <int U> __device__ __forceinline__ void fast_piece(bool is_v, uint64_t tile_base, uint32_t lds) const {
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + U) * 1024));
        glds16_saddr(tile_base, is_v ? vfast[U] : kfast[U], lds + off);
    }
    // K and V of the same tile into one [K tile | V tile] buffer
    __device__ __forceinline__ void issue(int tile, char *kt) const {
        const uint32_t k_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)kt;
        issue2(tile, k_lds, tile, k_lds + TILEB);
    }
};

// W waves per workgroup (32 query rows each), NB LDS tile buffers (prefetch distance NB - 1).
template <typename T, int D, bool CAUSAL, int W, int NB, bool PP = false>
__global__ void __launch_bounds__(64 * W, 2) prefill_mfma_kernel(const AttnParams p) {
    const int pf_pp_pair = PP ? p.pp_pair : 0;
    constexpr int PF_BM = 32 * W;
    constexpr int NDMA2 = 2 * PfLoader<D, W>::NDMA;   // DMA instructions per wave per tile (K and V)
    constexpr int ROWB = D * 2;                     // bytes per tile row
    constexpr int TILEB = PF_BN * ROWB;             // bytes per K (or V) tile
    constexpr int NJ = D / 16;                      // MFMA k-steps over d for S^T = K.Q^T
    constexpr int NDB = D / 32;                     // 32-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NB][K tile | V tile]

    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (scalar branches below)
    PfWork wk;
    if (!pf_map_workgroup(p, PF_BM, wk)) return;
    const int b = wk.b, hq = wk.hq, mblk = wk.mblk;
    const int hk = hq / (p.h / p.h_k);
    const SeqInfo si(p, b);
    const int m0 = mblk * PF_BM;
    if (m0 >= si.len_q) return;
    const int shift = si.len_k - si.len_q;          // mask.h:170: key <= row + seqlen_k - seqlen_q
    const int mw0 = m0 + wave * 32;                 // first query row of this wave
    const int my_q = mw0 + lq;                      // this lane's query row

    // keys the workgroup / this wave can see at all
    int n_end = si.len_k;
    if (CAUSAL) n_end = min(n_end, m0 + PF_BM + shift);
    int n_end_w = si.len_k;
    if (CAUSAL) n_end_w = min(n_end_w, mw0 + 32 + shift);
    const int n_tiles = n_end > 0 ? (n_end + PF_BN - 1) / PF_BN : 0;

    // ---- Q^T fragments: lane = (query lq, d half hi); B operand of S^T = K.Q^T ----
    uint4 qf[NJ];
    {
        const int qrow = min(my_q, si.len_q - 1);
        const uint16_t *qp = p.q + si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)qrow * p.q_row_stride +
                             (int64_t)hq * p.q_head_stride + hi * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) qf[j] = *reinterpret_cast<const uint4 *>(qp + j * 16);
        // Make hipcc retire these loads HERE: its vmcnt bookkeeping does not see the asm DMAs below,
        // and a counted wait for q inside the tile loop would drain the DMA queue every iteration.
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(qf[j].x), "+v"(qf[j].y), "+v"(qf[j].z), "+v"(qf[j].w));
    }

    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const uint16_t *kbase = p.k + (int64_t)hk * p.k_head_stride + (paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b));
    const uint16_t *vbase = p.v + (int64_t)hk * p.v_head_stride + (paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b));
    PfLoader<D, W> ld;
    ld.kbase = kbase; ld.vbase = vbase; ld.bt = bt;
    ld.k_page = p.k_batch_stride; ld.k_row = p.k_row_stride; ld.v_page = p.v_batch_stride; ld.v_row = p.v_row_stride;
    ld.page_size = p.page_size; ld.last_key = si.len_k - 1; ld.wave = wave;
    ld.page_shift = (p.page_size > 0 && (p.page_size & (p.page_size - 1)) == 0) ? __builtin_ctz(p.page_size) : -1;
    ld.init(lane);
    // ---- per-lane LDS read offsets (tile-relative), computed once ----
    // K: lane reads key row lq (+32 per 32-key half) at chunk 2j+hi, swizzled by the row.
    uint32_t koff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) koff[j] = (uint32_t)(lq * ROWB + PfSwz<D>::k(lq, 2 * j + hi) * 16);
    // V^T via ds_read_b64_tr_b16: 16-lane group g2 = lane>>4 reads a [4 keys][16 d] block transposed;
    // lanes 4j+c of the group supply the address of V[key j][16-col block, 4c..4c+3]; afterwards the
    // group's lanes hold, for d column 16*(g2&1) + (lane&15), the 4 keys.  Row part 4hi+jrow and the
    // swizzle depend only on the lane; 32-key half, 16-key k-step and +8 rows are immediates.
    const int g2 = lane >> 4, jrow = (lane & 15) >> 2, cc = lane & 3;
    uint32_t voff[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const int vrow = 4 * hi + jrow;
        const int dcol = db * 32 + 16 * (g2 & 1) + 4 * cc;  // element index, 4 contiguous
        voff[db] = (uint32_t)(vrow * ROWB + PfSwz<D>::v(vrow, dcol >> 3) * 16 + (dcol & 7) * 2);
    }

    f32x16_v oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;   // running max in the scaled log2 domain (both halves agree); this lane's part of the row sum
    const float sl2 = p.scale_log2;

#ifdef PF_TIMING
    float tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // ---- S^T[key][query] = K.Q^T of one tile, the two 32-key halves ----
    // LDS fragments are fetched one k-step ahead of the MFMAs that use them and the order is pinned with
    // sched_barrier(0): left alone, hipcc issues each ds_read right in front of its MFMA and the LDS
    // latency shows once per MFMA pair.
    auto qk_tile = [&](uint32_t kt, f32x16_v (&s)[2]) {
        // Per-tile read bases = buffer address + the lane's constant offset.  They are made opaque
        // so that the remaining constants (32-key half, 16-key step, +8 rows) fold into the ds_read
        // `offset:` immediates instead of costing one v_add per LDS read (the reassociator would
        // otherwise pair the constant with the uniform buffer address).
        uint32_t kb_[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { kb_[j] = kt + koff[j]; asm volatile("" : "+v"(kb_[j])); }
        auto read_k = [&](uint32_t addr) {
            const u32x4_v av = *(const __attribute__((address_space(3))) u32x4_v *)(uintptr_t)addr;
            uint4 a;
            a.x = av[0]; a.y = av[1]; a.z = av[2]; a.w = av[3];
            return a;
        };
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
        constexpr int KLA = 2;          // k-steps of lookahead (an LDS read takes 100+ cycles under load, a k-step 64)
        uint4 kf[KLA + 1][2];           // [k-step % (KLA + 1)][32-key half]
#pragma unroll
        for (int j = 0; j < KLA && j < NJ; ++j) {
            kf[j][0] = read_k(kb_[j]);
            kf[j][1] = read_k(kb_[j] + 32 * ROWB);
        }
        // the two 32-key halves are independent accumulators: alternate them so that no MFMA
        // waits for the previous one's result
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + KLA < NJ) {
                kf[(j + KLA) % (KLA + 1)][0] = read_k(kb_[j + KLA]);
                kf[(j + KLA) % (KLA + 1)][1] = read_k(kb_[j + KLA] + 32 * ROWB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) s[blk] = mfma32<T>(kf[j % (KLA + 1)][blk], qf[j], s[blk]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- mask (diagonal / tail tiles only), online softmax in the exp2 domain; rescales O ----
    auto softmax_tile = [&](f32x16_v (&s)[2], int kv0, uint4 (&pp)[2][2]) {
        const bool need_mask = (kv0 + PF_BN > si.len_k) || (CAUSAL && kv0 + PF_BN > mw0 + shift + 1);
        if (need_mask) {  // wave-uniform
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < si.len_k && (!CAUSAL || key <= my_q + shift);
                    s[blk][r] = ok ? s[blk][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;   // raw-domain max (scale > 0 commutes with max)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = pair_max(mx);
        const float m_cand = fmaxf(m_run, mx * sl2);
        const float m_new = m_cand > m_run + PF_DEFER ? m_cand : m_run;   // deferred raise (PF_DEFER)
        const float ms = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - ms);
        m_run = m_new;
        // packed f32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction) around
        // the 32 v_exp_f32; four independent partial sums keep the add chain short
        f32x2_v ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        const f32x2_v sl2v = {sl2, sl2}, nmsv = {-ms, -ms};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_v x = {s[blk][r], s[blk][r + 1]};
                const f32x2_v y = __builtin_elementwise_fma(x, sl2v, nmsv);   // s*scale*log2e - max
                f32x2_v ev;
                ev[0] = __builtin_amdgcn_exp2f(y[0]);
                ev[1] = __builtin_amdgcn_exp2f(y[1]);
                e[r] = ev[0];
                e[r + 1] = ev[1];
                ps2[(r >> 1) & 1] += ev;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                pp[blk][kk].x = cvt_pk<T>(e[8 * kk + 0], e[8 * kk + 1]);
                pp[blk][kk].y = cvt_pk<T>(e[8 * kk + 2], e[8 * kk + 3]);
                pp[blk][kk].z = cvt_pk<T>(e[8 * kk + 4], e[8 * kk + 5]);
                pp[blk][kk].w = cvt_pk<T>(e[8 * kk + 6], e[8 * kk + 7]);
            }
        }
        const f32x2_v pst = ps2[0] + ps2[1];
        const float psum = pst[0] + pst[1];
        l_part = l_part * alpha + psum;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
    };

    // ---- O^T[d][query] += V^T . P^T of one tile ----
    auto pv_tile = [&](uint32_t vt, const uint4 (&pp)[2][2]) {
        uint32_t vb_[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) { vb_[db] = vt + voff[db]; asm volatile("" : "+v"(vb_[db])); }
        auto read_vt = [&](uint32_t addr) {   // two transposed 8-byte reads = one A operand
            const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_v *)(uintptr_t)addr));
            const uint2 hi2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_v *)(uintptr_t)(addr + 8 * ROWB)));
            uint4 a;
            a.x = lo.x; a.y = lo.y; a.z = hi2.x; a.w = hi2.y;
            return a;
        };
        // (32-key half, 16-key step) outer, the NDB independent accumulators inner; operands one step ahead
        constexpr int NA = 4 * NDB, VLA = 4;   // operands per tile; operands (= MFMAs, 32 cycles each) of lookahead
        auto vaddr = [&](int n) { const int q = n / NDB; return vb_[n % NDB] + ((q >> 1) * 32 + (q & 1) * 16) * ROWB; };
        uint4 vf[VLA + 1];
#pragma unroll
        for (int n = 0; n < VLA && n < NA; ++n) vf[n] = read_vt(vaddr(n));
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            const int q = n / NDB, db = n % NDB, blk = q >> 1, kk = q & 1;
            if (n + VLA < NA) vf[(n + VLA) % (VLA + 1)] = read_vt(vaddr(n + VLA));
            __builtin_amdgcn_sched_barrier(0);
            oacc[db] = mfma32<T>(vf[n % (VLA + 1)], pp[blk][kk], oacc[db]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (PP) {
        // ---- ping-pong (W = 8, NB = 4): the two wavefronts of a SIMD (waves w and w + 4: rows 0..127 and 128..255 of the block)
        // alternate -- while one issues the 32 MFMAs of a tile (P.V of the previous tile, then Q.K^T of this one), the other
        // runs the softmax VALU stream of its tile; a workgroup barrier after every half step keeps them in opposite phases, so
        // the matrix pipe of the SIMD sees the two wavefronts' MFMA bursts back to back instead of two wavefronts that
        // drift into the same phase (measured on the 2 x 4-wave layout: MFMA time and VALU time of a SIMD add up).
        //   half step hs:  set 0: even -> M(hs / 2), odd -> V((hs - 1) / 2);   set 1 one half step later.
        //   M(t) = PV(t - 1) + QK(t);  V(t) = mask + online softmax of tile t (rescales O).
        // K/V tile t lives in ring slot t % 4 from half step 2t to 2t + 3; tile hs / 2 + 2 is requested at even hs (its slot
        // was released by the barrier of half step hs - 1), i.e. two tile times ahead of its first use.
        static_assert(!PP || (W == 8 && NB == 4), "ping-pong layout");
        // which two wavefronts share a SIMD depends on how the dispatcher deals a workgroup's wavefronts to the SIMDs:
        // pp_pair 0 -> (w, w + 4), 1 -> (2k, 2k + 1) (measured: see DESIGN.md 4.2)
        const int set = pf_pp_pair ? (wave & 1) : (wave >> 2);
        f32x16_v s[2];
        uint4 pp[2][2];
        if (0 < n_tiles) ld.issue(0, smem);
        if (1 < n_tiles) ld.issue(1, smem + 2 * TILEB);
        if (n_tiles > 1) dma_wait_keep<NDMA2>(); else dma_wait_all();
        __syncthreads();
        const int last_hs = 2 * n_tiles + 1;
        for (int hs = 0; hs <= last_hs; ++hs) {
            if ((hs & 1) == 0) {
                const int nt = (hs >> 1) + 2;
                if (nt < n_tiles) ld.issue(nt, smem + (nt & 3) * 2 * TILEB);
            }
            const int my = hs - set;                         // this set's own half-step counter
            if (my >= 0 && my <= 2 * n_tiles) {
                const int t = my >> 1;
                if ((my & 1) == 0) {                         // M phase of tile t
                    if (t >= 1 && (t - 1) * PF_BN < n_end_w) pv_tile(smem_lds + ((t - 1) & 3) * 2 * TILEB + TILEB, pp);
                    __builtin_amdgcn_sched_barrier(0);       // keep the K fragments of QK out of PV's register budget
                    if (t < n_tiles && t * PF_BN < n_end_w) qk_tile(smem_lds + (t & 3) * 2 * TILEB, s);
                } else if (t < n_tiles && t * PF_BN < n_end_w) {   // V phase of tile t
                    softmax_tile(s, t * PF_BN, pp);
                }
            }
            // tile (hs + 1) / 2 is first read in the next half step when that one is even: its DMAs (requested two tiles ago)
            // must have landed; the one requested after it may stay in flight
            if (hs & 1) {
                const int need = (hs + 1) >> 1;
                if (need < n_tiles) {
                    if (need + 1 < n_tiles) dma_wait_keep<NDMA2>(); else dma_wait_all();
                }
            }
            __syncthreads();
        }
    } else {
    // LDS ring of NB tiles, prefetch distance NB - 1, ONE barrier per tile:
    //   wait for this wave's pieces of tile t (later tiles may stay in flight) -> barrier (everyone's
    //   pieces of tile t are in LDS, and everyone is done reading tile t-1) -> issue tile t+NB-1 into the
    //   buffer tile t-1 used -> MFMAs of tile t.
#pragma unroll
    for (int s0 = 0; s0 < NB - 1; ++s0)
        if (s0 < n_tiles) ld.issue(s0, smem + s0 * 2 * TILEB);
    int buf = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (NB > 2 && t + 1 < n_tiles) dma_wait_keep<NDMA2>();   // NB == 3: tile t+1 stays in flight
        else dma_wait_all();
        __syncthreads();
        PF_T(0);
        if (t + NB - 1 < n_tiles) {
            const int nb = buf + NB - 1 >= NB ? buf - 1 : buf + NB - 1;
            ld.issue(t + NB - 1, smem + nb * 2 * TILEB);
        }
        PF_T(1);
        const int kv0 = t * PF_BN;
        if (kv0 < n_end_w) {  // wave-uniform: tile not entirely masked for this wave's rows
            const uint32_t kt = smem_lds + buf * 2 * TILEB;
            f32x16_v s[2];
            uint4 pp[2][2];  // P^T operands: [32-key half][16-key k-step]
            qk_tile(kt, s);
#ifdef PF_TIMING
            asm volatile("" :: "v"(s[0][0]), "v"(s[1][15]));
#endif
            PF_T(2);
            softmax_tile(s, kv0, pp);
#ifdef PF_TIMING
            asm volatile("" :: "v"(pp[0][0].x), "v"(pp[1][1].w));
#endif
            PF_T(3);
            pv_tile(kt + TILEB, pp);
#ifdef PF_TIMING
            asm volatile("" :: "v"(oacc[0][0]), "v"(oacc[NDB - 1][15]));
#endif
            PF_T(4);
        }
        buf = buf + 1 == NB ? 0 : buf + 1;
    }

    }
#ifdef PF_TIMING
    if (p.lse && tid == 0) {
        tacc[5] = (float)n_tiles;
        tacc[6] = (float)(__builtin_readcyclecounter() - tstart);
        for (int i2 = 0; i2 < 8; ++i2) p.lse[(int64_t)blockIdx.x * 8 + i2] = tacc[i2];
    }
    if (p.lse) return;
#endif
    // ---- epilogue: total row sum = own part + partner lane's part (same running max) ----
    const float l_tot = pair_sum(l_part);
    if (my_q >= si.len_q) return;
    const bool empty = !(l_tot > 0.f);
    const float inv = empty ? 0.f : 1.f / l_tot;
    uint16_t *op = p.o + si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)my_q * p.o_row_stride +
                   (int64_t)hq * p.o_head_stride;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            uint2 w;
            w.x = pack2<T>(oacc[db][4 * r4 + 0] * inv, oacc[db][4 * r4 + 1] * inv);
            w.y = pack2<T>(oacc[db][4 * r4 + 2] * inv, oacc[db][4 * r4 + 3] * inv);
            *reinterpret_cast<uint2 *>(op + db * 32 + 8 * r4 + 4 * hi) = w;  // d = 32db + (r&3) + 8(r>>2) + 4hi
        }
    if (p.lse && hi == 0) {
        const float lse = empty ? INFINITY : (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
        if (p.unpadded_lse && p.cu_seqlens_q) p.lse[(int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q + my_q] = lse;
        else p.lse[((int64_t)b * p.h + hq) * p.seqlen_q + my_q] = lse;
    }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined kernels: 4 waves per workgroup, RB 32-row query blocks per wave.
//   RB = 1: 128-row workgroup, two workgroups per CU (two waves per SIMD, 256 registers each);
//   RB = 2: 256-row workgroup, ONE wave per SIMD with the whole 512-register file; every K / V^T
//           fragment read from LDS feeds two MFMAs.
// The matrix pipe and the VALU are separate issue ports of a SIMD but a wave issues in order, so the two
// instruction streams are interleaved in program order and pinned with sched_barrier(0):
//   phase B(t):  S(t+1) = K(t+1).Q^T  (MFMA + K ds_read_b128)   ||  P(t) = exp2(S(t) - m), row sums, pack  (VALU)
//   phase C(t):  O += V(t)^T.P(t)^T   (MFMA + V^T tr reads)     ||  row max of S(t+1), rescale decision  (VALU)
// The running max is only raised when some row's new max exceeds it by more than PF_DEFER (log2 units;
// P <= 2^PF_DEFER stays exact in the f32 sums and representable in f16 / bf16), so the rescale of O is
// a rare, separate block and phases B and C are straight-line code.
// LDS: rings of NS K tiles and NS V tiles (NS = 2 at RB = 1, 3 at RB = 2); tile t lives in slot t % NS.
// Prefetch groups {K0}, {K1, V0}, .. {K(NS-1), V(NS-2)} before the loop, then {K(t+NS), V(t+NS-1)} at the
// top of iteration t, into the slots of K(t) and V(t-1), which are free once the barrier is passed.
//
// Register files.  VALU instructions only see the arch VGPRs; hipcc's own MFMA selection puts every D/C
// operand in the accumulator file once a kernel may exceed 256 registers and then shuffles tuples
// between the files (v_accvgpr_* by the hundred per tile).  So the accumulator registers below PF_NACC
// are owned by the asm statements, with literal register numbers:  O^T[rb][db] = a[16(rb.NDB+db) ..+15],
// Q^T[rb][j] = a[PF_QA + 4(rb.NJ+j) ..+3]; S (read by the softmax VALU), P, and the K / V^T operands are
// ordinary variables in arch VGPRs.  Every statement lists the owned registers as clobbered, so hipcc
// keeps out of them (it spills to the accumulator registers above).  hipcc pads nothing inside an asm
// string: `s_nop 1` covers a just-written VGPR (or v_accvgpr_write) -> MFMA operand read; readers of an
// MFMA's D other than the next MFMA of its accumulation chain come after an `s_nop 11` (8-pass MFMA D
// -> any other access: 12 wait states).
#define PF_ACC_CLOBBERS_96 \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", \
    "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", \
    "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", \
    "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", \
    "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", \
    "a92", "a93", "a94", "a95"
#define PF_ACC_CLOBBERS_192 \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", \
    "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", \
    "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", \
    "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", \
    "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", \
    "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", \
    "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", \
    "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", \
    "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", \
    "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", \
    "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", \
    "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", \
    "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NACC = number of asm-owned accumulator registers (96: RB = 1, 192: RB = 2)
template <typename T, int NACC> struct PfAcc;
#define ATOMA_PF_ACC(TT, MN, NACC, CLOB)                                                                                  \
    template <> struct PfAcc<TT, NACC> {                                                                                  \
        /* s (+)= k.Q^T[q]: D (tied: the tuple keeps its registers across the tile loop), A in VGPRs (K fragment     */    \
        /* straight from LDS), B = a[q..q+3] */                                                                           \
        template <bool FIRST, int Q>                                                                                      \
        static __device__ __forceinline__ void qk(f32x16_v &s, const u32x4_v &k) {                                        \
            if constexpr (FIRST)                                                                                          \
                asm volatile(MN " %0, %1, a[%2:%3], 0" : "+v"(s) : "v"(k), "n"(Q), "n"(Q + 3) : CLOB);                     \
            else                                                                                                          \
                asm volatile(MN " %0, %1, a[%2:%3], %0" : "+v"(s) : "v"(k), "n"(Q), "n"(Q + 3) : CLOB);                    \
        }                                                                                                                 \
        /* a[o..o+15] += vt.p (p written by VALU: two wait states first) */                                                \
        template <int O>                                                                                                  \
        static __device__ __forceinline__ void pv(const u32x4_v &vt, const u32x4_v &p) {                                   \
            asm volatile("s_nop 1\n\t" MN " a[%2:%3], %0, %1, a[%2:%3]" :: "v"(vt), "v"(p), "n"(O), "n"(O + 15) : CLOB);    \
        }                                                                                                                 \
        template <int A> static __device__ __forceinline__ void write4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) { \
            asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\t"                              \
                         "v_accvgpr_write_b32 a[%6], %2\n\tv_accvgpr_write_b32 a[%7], %3"                                  \
                         :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "n"(A), "n"(A + 1), "n"(A + 2), "n"(A + 3) : CLOB);        \
        }                                                                                                                 \
        template <int A> static __device__ __forceinline__ void read4(float &x0, float &x1, float &x2, float &x3) {        \
            asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\t"                                \
                         "v_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"                                    \
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "n"(A), "n"(A + 1), "n"(A + 2), "n"(A + 3) : CLOB);    \
        }                                                                                                                 \
        template <int A> static __device__ __forceinline__ void scale4(float alpha) { /* a[A..A+3] *= alpha */             \
            float t0, t1, t2, t3;                                                                                         \
            asm volatile("v_accvgpr_read_b32 %0, a[%5]\n\tv_accvgpr_read_b32 %1, a[%6]\n\t"                                \
                         "v_accvgpr_read_b32 %2, a[%7]\n\tv_accvgpr_read_b32 %3, a[%8]\n\t"                                \
                         "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t" \
                         "v_accvgpr_write_b32 a[%5], %0\n\tv_accvgpr_write_b32 a[%6], %1\n\t"                              \
                         "v_accvgpr_write_b32 a[%7], %2\n\tv_accvgpr_write_b32 a[%8], %3"                                  \
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)                                                     \
                         : "v"(alpha), "n"(A), "n"(A + 1), "n"(A + 2), "n"(A + 3) : CLOB);                                 \
        }                                                                                                                 \
        /* MFMA D (accumulator file) -> v_accvgpr_read */                                                                 \
        static __device__ __forceinline__ void fence() { asm volatile("s_nop 11" ::: CLOB); }                              \
    }
ATOMA_PF_ACC(bf16_t, "v_mfma_f32_32x32x16_bf16", 96, PF_ACC_CLOBBERS_96);
ATOMA_PF_ACC(f16_t, "v_mfma_f32_32x32x16_f16", 96, PF_ACC_CLOBBERS_96);
ATOMA_PF_ACC(bf16_t, "v_mfma_f32_32x32x16_bf16", 192, PF_ACC_CLOBBERS_192);
ATOMA_PF_ACC(f16_t, "v_mfma_f32_32x32x16_f16", 192, PF_ACC_CLOBBERS_192);
#undef ATOMA_PF_ACC

#ifndef PF_PIPE_VLA
#define PF_PIPE_VLA 2
#endif
template <typename T, int D, bool CAUSAL, int RB>
__global__ void __launch_bounds__(256, RB == 2 ? 1 : 2) prefill_pipe_kernel(const AttnParams p) {
    constexpr int W = 4, PF_BM = 32 * RB * W;
    constexpr int NDMA = PfLoader<D, W>::NDMA;
    constexpr int ROWB = D * 2, TILEB = PF_BN * ROWB, NJ = D / 16, NDB = D / 32;
    constexpr int NS = RB == 2 ? 3 : 2;              // LDS ring depth (tiles of K, and of V)
    constexpr int PF_QA = RB * NDB * 16;             // first accumulator register of Q^T
    constexpr int NACC = RB * 96;
    static_assert(PF_QA + RB * NJ * 4 <= NACC, "accumulator map exceeds the clobber list");
    PfAcc<T, NACC> acc;
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // K tiles [NS] | V tiles [NS]

    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PfWork wk;   // workgroup -> (sequence, q head, query block): XCD / shader-engine aware order (pf_map_workgroup)
    if (!pf_map_workgroup(p, PF_BM, wk)) return;
    const int b = wk.b, hq = wk.hq, mblk = wk.mblk;
    const int hk = hq / (p.h / p.h_k);
    const SeqInfo si(p, b);
    const int m0 = mblk * PF_BM;
    if (m0 >= si.len_q) return;
    const int shift = si.len_k - si.len_q;
    // Row block rb of wave w = rows m0 + rb.RSTEP + 32w ..+31: the waves' blocks are interleaved (not 64
    // consecutive rows per wave), so that under a causal mask all four waves need (almost) the same
    // number of K/V tiles and nobody idles at the per-tile barrier.
    constexpr int RSTEP = PF_BM / RB;
    const int mw0 = m0 + wave * 32;                  // first query row of this wave's block 0
    const int my_q0 = mw0 + lq;                      // this lane's query row in row block 0 (+RSTEP per block)
    const bool pf_ride = (p.pp_pair & 1) != 0;       // DMA pieces inside phase C (launch_pf)

    int n_end = si.len_k, n_end_w = si.len_k;
    if (CAUSAL) {
        n_end = min(n_end, m0 + PF_BM + shift);
        n_end_w = min(n_end_w, mw0 + (RB - 1) * RSTEP + 32 + shift);   // the wave's last block decides
    }
    const int n_tiles = n_end > 0 ? (n_end + PF_BN - 1) / PF_BN : 0;

    // ---- Q^T fragments -> accumulator file; O^T = 0 ----
    static_for<0, RB>([&](auto RBc) {
        constexpr int rb = decltype(RBc)::value;
        const int qrow = min(my_q0 + RSTEP * rb, si.len_q - 1);
        const uint16_t *qp = p.q + si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)qrow * p.q_row_stride +
                             (int64_t)hq * p.q_head_stride + hi * 8;
        u32x4_v qv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) qv[j] = *reinterpret_cast<const u32x4_v *>(qp + j * 16);
        static_for<0, NJ>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            acc.template write4<PF_QA + (rb * NJ + j) * 4>(qv[j][0], qv[j][1], qv[j][2], qv[j][3]);
        });
    });
    static_for<0, RB * NDB * 4>([&](auto Ic) { acc.template write4<decltype(Ic)::value * 4>(0u, 0u, 0u, 0u); });

    const bool paged = p.block_table != nullptr;
    PfLoader<D, W> ld;
    ld.kbase = p.k + (int64_t)hk * p.k_head_stride + (paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b));
    ld.vbase = p.v + (int64_t)hk * p.v_head_stride + (paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b));
    ld.bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    ld.k_page = p.k_batch_stride; ld.k_row = p.k_row_stride; ld.v_page = p.v_batch_stride; ld.v_row = p.v_row_stride;
    ld.page_size = p.page_size; ld.last_key = si.len_k - 1; ld.wave = wave;
    ld.page_shift = (p.page_size > 0 && (p.page_size & (p.page_size - 1)) == 0) ? __builtin_ctz(p.page_size) : -1;
    ld.init(lane);

    uint32_t koff[NJ], voff[NDB];
#pragma unroll
    for (int j = 0; j < NJ; ++j) koff[j] = (uint32_t)(lq * ROWB + PfSwz<D>::k(lq, 2 * j + hi) * 16);
    {
        const int g2 = lane >> 4, jrow = (lane & 15) >> 2, cc = lane & 3, vrow = 4 * hi + jrow;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int dcol = db * 32 + 16 * (g2 & 1) + 4 * cc;
            voff[db] = (uint32_t)(vrow * ROWB + PfSwz<D>::v(vrow, dcol >> 3) * 16 + (dcol & 7) * 2);
        }
    }

    float m_run[RB], l_part[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { m_run[rb] = -INFINITY; l_part[rb] = 0.f; }
    const float sl2 = p.scale_log2;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

#ifdef PF_TIMING
    float tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    f32x16_v s_a[RB][2], s_b[RB][2];   // raw scores of two consecutive tiles (roles alternate)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_a[rb][blk][r] = s_b[rb][blk][r] = 0.f;
    u32x4_v pp[RB][2][2];              // P^T operands of the current tile: [row block][32-key half][16-key k-step]

    auto read_k = [&](uint32_t addr) { return *(const __attribute__((address_space(3))) u32x4_v *)(uintptr_t)addr; };
    auto read_vt = [&](uint32_t addr) {   // two transposed 8-byte reads = one A operand (16 keys x this lane's d column)
        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) short4_v *)(uintptr_t)addr));
        const uint2 hi2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) short4_v *)(uintptr_t)(addr + 8 * ROWB)));
        const u32x4_v a = {lo.x, lo.y, hi2.x, hi2.y};
        return a;
    };
    auto fence_s = [&](f32x16_v (&s)[RB][2]) {   // S (MFMA D in VGPRs) is read by VALU code next
        if constexpr (RB == 2) asm volatile("s_nop 11" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[RB - 1][0]), "+v"(s[RB - 1][1]));
    };

    // ---- phase B: [s_out = K.Q^T of the tile at kt]  ||  [P = exp2(s_in*scale - m), row sums, pack] ----
    // The instruction stream is written in issue order and pinned with sched_barrier(0): one MFMA, then
    // its share of the softmax VALU work (the matrix pipe is busy 32 cycles per MFMA; a wave issues in
    // order), K fragments fetched from LDS one k-step ahead.
    auto phase_b = [&](auto with_qk, auto with_sm, uint32_t kt, f32x16_v (&s_in)[RB][2], f32x16_v (&s_out)[RB][2]) {
        constexpr bool QK = decltype(with_qk)::value, SM = decltype(with_sm)::value;
        constexpr int NM = NJ * 2 * RB, EPM = RB * 32 / NM;   // MFMAs per phase, score elements handled beside each
        uint32_t kb_[NJ];
        u32x4_v kf[2][2];                                   // K fragments [k-step parity][32-key half]
        if (QK) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { kb_[j] = kt + koff[j]; asm volatile("" : "+v"(kb_[j])); }
            kf[0][0] = read_k(kb_[0]);
            kf[0][1] = read_k(kb_[0] + 32 * ROWB);
        }
        float nms[RB], ps[RB][2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            nms[rb] = m_run[rb] == -INFINITY ? 0.f : -m_run[rb];
            ps[rb][0] = ps[rb][1] = 0.f;
        }
        static_for<0, NM>([&](auto Mc) {
            constexpr int m = decltype(Mc)::value, j = m / (2 * RB), blk = (m / RB) & 1, rb = m % RB;
            if constexpr (QK) {
                if constexpr (m % (2 * RB) == 0 && j + 1 < NJ) {   // prefetch the next k-step's fragments
                    kf[(j + 1) & 1][0] = read_k(kb_[j + 1]);
                    kf[(j + 1) & 1][1] = read_k(kb_[j + 1] + 32 * ROWB);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc.template qk<j == 0, PF_QA + (rb * NJ + j) * 4>(s_out[rb][blk], kf[j & 1][blk]);
            }
            if constexpr (SM) {
#pragma unroll
                for (int n = m * EPM; n < (m + 1) * EPM; ++n) {
                    const int erb = n >> 5, eblk = (n >> 4) & 1, r = n & 15;
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s_in[erb][eblk][r], sl2, nms[erb]));
                    s_in[erb][eblk][r] = e;
                    ps[erb][r & 1] += e;
                    asm volatile("" : "+v"(ps[erb][r & 1]));   // the sum is taken HERE, beside this MFMA (IR-level sinking: see below)
                }
                if constexpr (((m + 1) * EPM) % 8 == 0) {     // a group of 8 is complete: pack it
                    constexpr int g = ((m + 1) * EPM) / 8 - 1, grb = g >> 2, gblk = (g >> 1) & 1, kk = g & 1;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        pp[grb][gblk][kk][c] = cvt_pk<T>(s_in[grb][gblk][8 * kk + 2 * c], s_in[grb][gblk][8 * kk + 2 * c + 1]);
                    // materialise P HERE: LLVM's IR-level sinking otherwise moves the fma / exp / cvt chains of the later groups down
                    // to their first use in phase C (sched_barrier only binds the machine scheduler), where nothing covers them
                    asm volatile("" : "+v"(pp[grb][gblk][kk]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (SM) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) l_part[rb] += ps[rb][0] + ps[rb][1];
        }
        if (QK) fence_s(s_out);
    };

    auto mask_tile = [&](f32x16_v (&s)[RB][2], int kv0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < si.len_k && (!CAUSAL || key <= my_q0 + RSTEP * rb + shift);
                    s[rb][blk][r] = ok ? s[rb][blk][r] : -INFINITY;
                }
    };
    // deferred raise of the running max; placed after every P.V MFMA of the previous tile
    auto raise_max = [&](const float (&mx)[RB]) {
        bool need = false;
        float m_new[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float m = pair_max(mx[rb]) * sl2;
            m_new[rb] = m;
            need = need || (m > m_run[rb] + PF_DEFER);
        }
        if (__any(need)) {   // rare after the first tiles: rescale O and the row sums to the new reference
            acc.fence();
            static_for<0, RB>([&](auto RBc) {
                constexpr int rb = decltype(RBc)::value;
                const float mt = fmaxf(m_run[rb], m_new[rb]);
                const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - (mt == -INFINITY ? 0.f : mt));
                m_run[rb] = mt;
                l_part[rb] *= alpha;
                static_for<0, NDB * 4>([&](auto Ic) { acc.template scale4<rb * NDB * 16 + decltype(Ic)::value * 4>(alpha); });
            });
        }
    };

    // ---- phase C: [O += V^T.P^T of the tile at vt]  ||  [row max of s_new] ----
    // Optional (ATOMA_PREFILL_RIDE=1): the DMA pieces of the NEXT ring slots ride in this phase (dma.on), one every NM / (2 NDMA)
    // MFMAs, K and V alternating, instead of going out as a burst right behind the barrier.  Measured: the burst shrinks by 330
    // cycles per tile and this phase grows by 410 -- a wave whose VMEM issue is held back (the LDS-DMA path moves about 1 KiB
    // per 100 cycles per wave) cannot issue its next MFMA either, so the pieces cost the same wherever they stand.
    struct DmaRide { bool on, do_k; uint64_t kb[NDMA], vb[NDMA]; uint32_t k_lds, v_lds; };
    auto phase_c = [&](auto with_max, uint32_t vt, f32x16_v (&s_new)[RB][2], float (&mx)[RB], const DmaRide &dma) {
        constexpr bool MX = decltype(with_max)::value;
        constexpr int NA = 4 * NDB, EPM = 32 / NA;           // V^T operands per tile; score elements per MFMA
        uint32_t vb_[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) { vb_[db] = vt + voff[db]; asm volatile("" : "+v"(vb_[db])); }
        constexpr int VLA = PF_PIPE_VLA; // operands of lookahead (2 or 4 measure the same: the phase does not wait on these reads)
        auto vaddr = [&](int n) { const int q = n / NDB; return vb_[n % NDB] + ((q >> 1) * 32 + (q & 1) * 16) * ROWB; };
        u32x4_v vf[VLA + 1];
#pragma unroll
        for (int n = 0; n < VLA; ++n) vf[n] = read_vt(vaddr(n));
        if (MX) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) mx[rb] = s_new[rb][0][0];
        }
        static_for<0, NA>([&](auto Nc) {
            constexpr int n = decltype(Nc)::value, q = n / NDB, db = n % NDB, blk = q >> 1, kk = q & 1;
            if constexpr (n + VLA < NA) {
                vf[(n + VLA) % (VLA + 1)] = read_vt(vaddr(n + VLA));
                __builtin_amdgcn_sched_barrier(0);
            }
            static_for<0, RB>([&](auto Rc) {
                constexpr int rb = decltype(Rc)::value;
                acc.template pv<(rb * NDB + db) * 16>(vf[n % (VLA + 1)], pp[rb][blk][kk]);
                if constexpr (MX) {
                    constexpr int e0 = (n * RB + rb) * EPM;   // 0 .. 32.RB: elements of the flattened [rb][blk][r]
#pragma unroll
                    for (int e = e0; e < e0 + EPM; ++e) mx[e >> 5] = fmaxf(mx[e >> 5], s_new[e >> 5][(e >> 4) & 1][e & 15]);
                    asm volatile("" : "+v"(mx[e0 >> 5]));      // keep the max chain spread over the phase
                }
                constexpr int mi = n * RB + rb, STEP = NA * RB / (2 * NDMA);   // MFMA index in the phase; MFMAs per DMA piece
                if constexpr (mi % STEP == STEP / 2) {
                    constexpr int pi = mi / STEP, u = pi >> 1;
                    if (dma.on) {
                        if constexpr ((pi & 1) == 0) { if (dma.do_k) ld.template piece_k<u>(dma.kb[u], dma.k_lds); }
                        else ld.template piece_v<u>(dma.vb[u], dma.v_lds);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    auto tile_max = [&](f32x16_v (&s)[RB][2], float (&mx)[RB]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float m = -INFINITY;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, s[rb][blk][r]);
            mx[rb] = m;
        }
    };
    auto need_mask = [&](int kv0) { return (kv0 + PF_BN > si.len_k) || (CAUSAL && kv0 + PF_BN > mw0 + shift + 1); };
    const std::true_type yes;
    const std::false_type no;
    auto k_slot = [&](int t) { return smem_lds + (uint32_t)(t % NS) * TILEB; };
    auto v_slot = [&](int t) { return smem_lds + (uint32_t)(NS + t % NS) * TILEB; };

    // one tile: s_c holds the raw scores of tile t (row max already folded into m_run), s_n receives tile t+1
    auto iteration = [&](int t, f32x16_v (&s_c)[RB][2], f32x16_v (&s_n)[RB][2]) {
        // K(t+1), V(t) must have landed; the NS - 2 younger groups may stay in flight (all of them are
        // complete groups as long as K(t+NS-1) exists)
        if (NS > 2 && t + NS - 1 < n_tiles) dma_wait_keep<(NS - 2) * 2 * NDMA>();
        else dma_wait_all();
        __syncthreads();
        PF_T(0);
        const int kv0 = t * PF_BN;
        const bool cur = kv0 < n_end_w, nxt = t + 1 < n_tiles && kv0 + PF_BN < n_end_w;   // wave-uniform; nxt implies cur
        DmaRide dma;
        dma.on = false;
        if (t + NS - 1 < n_tiles) {
            const int kti = t + NS < n_tiles ? t + NS : -1, vti = t + NS - 1;
            dma.do_k = kti >= 0; dma.k_lds = k_slot(t + NS); dma.v_lds = v_slot(t + NS - 1);
            if (pf_ride && nxt && ld.plan2(kti, vti, dma.kb, dma.vb)) dma.on = true;   // pieces go out inside phase C
            else ld.issue2(kti, dma.k_lds, vti, dma.v_lds);
        }
        PF_T(1);
        float mx[RB];
        if (nxt) {
            phase_b(yes, yes, k_slot(t + 1), s_c, s_n);
            PF_T(2);
            if (need_mask(kv0 + PF_BN)) mask_tile(s_n, kv0 + PF_BN);
            phase_c(yes, v_slot(t), s_n, mx, dma);
            PF_T(3);
            raise_max(mx);
            PF_T(4);
        } else if (cur) {
            phase_b(no, yes, 0, s_c, s_n);
            phase_c(no, v_slot(t), s_n, mx, dma);
        }
    };

    if (n_tiles > 0) {
        // prefetch groups {K0}, {K1, V0}, .. ; K(0) must have landed, everything issued after it may stay in flight
        ld.issue2(0, k_slot(0), -1, 0);
        int younger = 0;   // pieces issued after K(0)
#pragma unroll
        for (int g = 1; g < NS; ++g) {
            if (g - 1 < n_tiles) {
                ld.issue2(g < n_tiles ? g : -1, k_slot(g), g - 1, v_slot(g - 1));
                younger += (g < n_tiles ? 2 : 1) * NDMA;
            }
        }
        if (younger == (2 * NS - 2) * NDMA) dma_wait_keep<(2 * NS - 2) * NDMA>();
        else if (younger == NDMA) dma_wait_keep<NDMA>();
        else if (NS == 3 && younger == 3 * NDMA) dma_wait_keep<3 * NDMA>();
        else dma_wait_all();
        __syncthreads();
        if (0 < n_end_w) {
            float mx[RB];
            phase_b(yes, no, k_slot(0), s_b, s_a);
            if (need_mask(0)) mask_tile(s_a, 0);
            tile_max(s_a, mx);
            raise_max(mx);
        }
    }
    for (int t = 0; t < n_tiles; t += 2) {
        iteration(t, s_a, s_b);
        if (t + 1 < n_tiles) iteration(t + 1, s_b, s_a);
    }

#ifdef PF_TIMING
    if (p.lse && tid == PF_TIMING_TID) {
        tacc[5] = (float)n_tiles;
        tacc[6] = (float)(__builtin_readcyclecounter() - tstart);
        for (int i2 = 0; i2 < 8; ++i2) p.lse[(int64_t)blockIdx.x * 8 + i2] = tacc[i2];
    }
    if (p.lse) return;
#endif
    // ---- epilogue ----
    acc.fence();
    static_for<0, RB>([&](auto RBc) {
        constexpr int rb = decltype(RBc)::value;
        const int my_q = my_q0 + RSTEP * rb;
        const float l_tot = pair_sum(l_part[rb]);
        const bool empty = !(l_tot > 0.f);
        const float inv = empty ? 0.f : 1.f / l_tot;
        uint16_t *op = p.o + si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)my_q * p.o_row_stride +
                       (int64_t)hq * p.o_head_stride;
        static_for<0, NDB * 4>([&](auto Ic) {
            constexpr int db = decltype(Ic)::value >> 2, r4 = decltype(Ic)::value & 3;
            float x0, x1, x2, x3;
            acc.template read4<(rb * NDB + db) * 16 + 4 * r4>(x0, x1, x2, x3);
            uint2 w;
            w.x = pack2<T>(x0 * inv, x1 * inv);
            w.y = pack2<T>(x2 * inv, x3 * inv);
            if (my_q < si.len_q) *reinterpret_cast<uint2 *>(op + db * 32 + 8 * r4 + 4 * hi) = w;
        });
        if (p.lse && hi == 0 && my_q < si.len_q) {
            const float lse = empty ? INFINITY : (m_run[rb] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
            if (p.unpadded_lse && p.cu_seqlens_q) p.lse[(int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q + my_q] = lse;
            else p.lse[((int64_t)b * p.h + hq) * p.seqlen_q + my_q] = lse;
        }
    });
}

bool prefill_mfma_supported(const AttnParams &p) {
    return (p.d == 64 || p.d == 128) && p.alibi_slopes == nullptr && p.seqlen_q > 1;
}

// The dynamic-LDS opt-in of a kernel, once per DEVICE and per kernel (one host thread per GPU calls in: model_executor.rs:428-440; a process-wide
// flag served only the device that came first -- ADVICE r4 on prefill_asm.hip, the same pattern here), return code checked.
static bool pf_lds_opt_in(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;       // (kernel, device)
    int dev = 0;
    if (!check_hip(hipGetDevice(&dev), "hipGetDevice")) return false;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : done)
        if (e.first == kernel && e.second == dev) return true;
    if (!check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "prefill: dynamic LDS opt-in")) return false;
    done.emplace_back(kernel, dev);
    return true;
}

template <typename T, int D, bool CAUSAL, int W, int NB, bool PP = false>
static void launch_pf_cfg(const AttnParams &p, hipStream_t stream) {
#ifdef PF_TIMING   // ATOMA_PF_ONE_WG=1: pad the LDS request so that only one workgroup fits a CU (occupancy experiment)
    static const int pad = getenv("ATOMA_PF_ONE_WG") ? 96 * 1024 - NB * 2 * PF_BN * D * 2 : 0;
    const int smem = NB * 2 * PF_BN * D * 2 + pad;
#else
    constexpr int smem = NB * 2 * PF_BN * D * 2;
#endif
    if (!pf_lds_opt_in(reinterpret_cast<const void *>(&prefill_mfma_kernel<T, D, CAUSAL, W, NB, PP>), smem)) return;   // up to 96 KiB: above the default dynamic-LDS limit
    const int64_t m_blocks = cdiv(p.seqlen_q, 32 * W), n_units = (int64_t)p.b * p.h;
    const int64_t nu_max = n_units / 8 + (n_units % 8 ? 1 : 0);
    dim3 grid((unsigned)(8 * nu_max * m_blocks));   // padded: see the mapping comment in the kernel
    hipLaunchKernelGGL((prefill_mfma_kernel<T, D, CAUSAL, W, NB, PP>), grid, dim3(64 * W), smem, stream, p);
    ATOMA_CHECK_LAUNCH("prefill_mfma_kernel");
}

template <typename T, int D, bool CAUSAL, int RB>
static void launch_pf_pipe(const AttnParams &p, hipStream_t stream) {
    constexpr int smem = (RB == 2 ? 6 : 4) * PF_BN * D * 2;   // K and V rings of 3 (RB = 2) or 2 tiles
    if (!pf_lds_opt_in(reinterpret_cast<const void *>(&prefill_pipe_kernel<T, D, CAUSAL, RB>), smem)) return;
    const int64_t m_blocks = cdiv(p.seqlen_q, 128 * RB), n_units = (int64_t)p.b * p.h;
    const int64_t nu_max = n_units / 8 + (n_units % 8 ? 1 : 0);
    dim3 grid((unsigned)(8 * nu_max * m_blocks));
    hipLaunchKernelGGL((prefill_pipe_kernel<T, D, CAUSAL, RB>), grid, dim3(256), smem, stream, p);
    ATOMA_CHECK_LAUNCH("prefill_pipe_kernel");
}

// atoma_set_option("prefill_cfg", n): 4 = the hand-scheduled kernel of prefill_asm.hip (head_dim 128: persistent, 4 waves x 64
// rows, one wave per SIMD), the default -- shapes it does not take (head_dim 64, ALiBi, ..) run on 0; 0 = tile-sequential loop
// (4 waves x 32 rows, two workgroups per CU); 2 = software-pipelined loop, 4 waves x 64 rows, one wave per SIMD (experimental: correct, but its
// un-overlapped barrier / LDS-DMA issue time makes it slower than 0 -- DESIGN.md).  RB = 1 of the pipelined
// kernel is not instantiated: hipcc splits a 256-register budget 128 / 128 between the two register files
// and the arch half spills.  The environment variable ATOMA_PREFILL_CFG overrides the option.
std::atomic<int> prefill_cfg{PREFILL_DEFAULT_CFG};
static int prefill_cfg_effective() {
    static const int env = [] { const char *e = getenv("ATOMA_PREFILL_CFG"); return e ? atoi(e) : -1; }();
    return env >= 0 ? env : prefill_cfg.load();   // the environment wins (lets the test suite run against a variant)
}

bool prefill_asm_supported(const AttnParams &p);                                   // prefill_asm.hip
int launch_prefill_asm(const AttnParams &p, bool is_bf16, hipStream_t stream);      // 0 / -1 (error in atoma_last_error)

template <typename T, int D, bool CAUSAL>
static void launch_pf(const AttnParams &p, hipStream_t stream) {
    switch (prefill_cfg_effective()) {
        case 4:                                                                    // hand-scheduled kernel (prefill_asm.hip); shapes it does not take fall through
            if (prefill_asm_supported(p)) { (void)launch_prefill_asm(p, std::is_same<T, bf16_t>::value, stream); break; }   // a failure is in the error slot: run_mha is void (ffi.rs), the host layer returns it
            launch_pf_cfg<T, D, CAUSAL, 4, 2>(p, stream);
            break;
        case 2: {
            static const int ride = [] { const char *e = getenv("ATOMA_PREFILL_RIDE"); return e ? atoi(e) : 0; }();   // 1 = DMA pieces inside phase C instead of a burst behind the barrier (measured: no gain)
            AttnParams q = p;
            q.pp_pair = ride;
            launch_pf_pipe<T, D, CAUSAL, 2>(q, stream);
            break;
        }
        case 3: {                                                              // 8-wave ping-pong (see the kernel)
            static const int pair = [] { const char *e = getenv("ATOMA_PREFILL_PP_PAIR"); return e ? atoi(e) : 0; }();
            AttnParams q = p;
            q.pp_pair = pair;
            launch_pf_cfg<T, D, CAUSAL, 8, 4, true>(q, stream);
            break;
        }
        default: launch_pf_cfg<T, D, CAUSAL, 4, 2>(p, stream); break;
    }
}

void launch_prefill_mfma(const AttnParams &p, bool is_bf16, hipStream_t stream) {
    if (p.b <= 0 || p.h <= 0 || p.seqlen_q <= 0) return;
#define ATOMA_PF(TT, DD)                                         \
    do {                                                         \
        if (p.is_causal) launch_pf<TT, DD, true>(p, stream);     \
        else launch_pf<TT, DD, false>(p, stream);                \
    } while (0)
    if (is_bf16) { if (p.d == 128) ATOMA_PF(bf16_t, 128); else ATOMA_PF(bf16_t, 64); }
    else { if (p.d == 128) ATOMA_PF(f16_t, 128); else ATOMA_PF(f16_t, 64); }
#undef ATOMA_PF
}

}  // namespace atoma
