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
#define PREFILL_DEFAULT_CFG 0
#endif
constexpr int PF_BN = 64;              // keys per K/V tile
// -DPF_TIMING: per-phase cycle accounting (s_memtime) of wave 0 of every workgroup, written to p.lse as
// [workgroup][8] floats (wait+barrier, dma issue, qk, softmax, pv, tiles, total, -) -- tools/probes/prefill_phases.py
#ifdef PF_TIMING
#define PF_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += (float)(now_ - tlast); tlast = now_; } while (0)
#else
#define PF_T(i) do {} while (0)
#endif
constexpr int PF_SGU = 8;              // (sequence, q head) units scheduled together on an XCD (2 kv groups at g = 4)

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
    static_assert(N == 2 || N == 4 || N == 8, "extend the switch");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
    int page_size, page_shift, last_key, wave;
    int row[NDMA];                    // tile row this lane fetches in DMA piece u
    uint32_t kchunk[NDMA], vchunk[NDMA];   // element offset of the (de-swizzled) 16-byte chunk inside the row
    uint32_t kfast[NDMA], vfast[NDMA];     // byte offset from the tile base, contiguous layout

    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int L = (wave * NDMA + u) * 64 + lane;   // 16-byte unit in the LDS image
            row[u] = L / CPR;
            const int slot = L % CPR;
            kchunk[u] = PfSwz<D>::k(row[u], slot) * 8;      // XOR is its own inverse
            vchunk[u] = PfSwz<D>::v(row[u], slot) * 8;
            kfast[u] = (uint32_t)(row[u] * k_row * 2) + kchunk[u] * 2;
            vfast[u] = (uint32_t)(row[u] * v_row * 2) + vchunk[u] * 2;
        }
    }

    __device__ __forceinline__ void issue(int tile, char *kt) const {
        char *vt = kt + TILEB;
        const uint32_t k_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)kt;
        const uint32_t v_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)vt;
        const int key0 = tile * PF_BN;
        if (!bt && key0 + PF_BN - 1 <= last_key) {  // workgroup-uniform: contiguous tensor, full tile
            const uint64_t kb = uniform64((uint64_t)(kbase + (int64_t)key0 * k_row));
            const uint64_t vb = uniform64((uint64_t)(vbase + (int64_t)key0 * v_row));
#pragma unroll
            for (int u = 0; u < NDMA; ++u) {
                const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + u) * 1024));
                glds16_saddr(kb, kfast[u], k_lds + off);
                glds16_saddr(vb, vfast[u], v_lds + off);
            }
            return;
        }
        const uint16_t *ksrc[NDMA], *vsrc[NDMA];
        // all block-table lookups first, then the DMAs back to back (a lookup's vmcnt wait
        // would otherwise drain the DMA issued just before it)
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int key = min(key0 + row[u], last_key);  // never read past the sequence
            int64_t koff, voff;
            if (bt) {
                const int pi = page_shift >= 0 ? key >> page_shift : key / page_size;
                const int r = key - pi * page_size;
                const int pg = bt[pi];
                koff = (int64_t)pg * k_page + (int64_t)r * k_row;
                voff = (int64_t)pg * v_page + (int64_t)r * v_row;
            } else {
                koff = (int64_t)key * k_row;
                voff = (int64_t)key * v_row;
            }
            ksrc[u] = kbase + koff + kchunk[u];
            vsrc[u] = vbase + voff + vchunk[u];
        }
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * NDMA + u) * 1024));
            glds16(ksrc[u], k_lds + off);
            glds16(vsrc[u], v_lds + off);
        }
    }
};

// W waves per workgroup (32 query rows each), NB LDS tile buffers (prefetch distance NB - 1).
template <typename T, int D, bool CAUSAL, int W, int NB, bool PRIO>
__global__ void __launch_bounds__(64 * W, 2) prefill_mfma_kernel(const AttnParams p) {
    constexpr int PF_BM = 32 * W;
    constexpr int NDMA2 = 2 * PfLoader<D, W>::NDMA;   // DMA instructions per wave per tile (K and V)
    constexpr int ROWB = D * 2;                     // bytes per tile row
    constexpr int TILEB = PF_BN * ROWB;             // bytes per K (or V) tile
    constexpr int NJ = D / 16;                      // MFMA k-steps over d for S^T = K.Q^T
    constexpr int NDB = D / 32;                     // 32-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NB][K tile | V tile]

    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (scalar branches below)
    // Workgroup -> (sequence, q head, query block).  Measured dispatch policy of the chip
    // (tools/probes/dispatch_probe.hip): workgroup L runs on XCD L % 8 and shader engine (L / 8) % 4 --
    // both fixed by the index -- and on the first free CU of that engine, in index order.  So:
    //  * units (sequence, q head) are dealt to XCDs in contiguous slices: all query blocks of all q heads
    //    of a kv head read the same K/V (1 MiB at S = 2048) and stay behind one 4 MiB L2;
    //  * inside an XCD, PF_SGU units at a time, blocks are ordered longest-first (causal: last block
    //    first) across those units, and dealt to the 4 shader engines in snake order, so every engine
    //    gets the same amount of work and its CUs take it in LPT order.
    // The grid is padded to 8 * (largest slice) * m_blocks; surplus workgroups exit here.
    const int L = (int)blockIdx.x, xcd = L & 7, i = L >> 3;
    const int m_blocks = (p.seqlen_q + PF_BM - 1) / PF_BM;
    const int n_units = p.b * p.h, uq = n_units >> 3, ur = n_units & 7;
    const int nu_x = uq + (xcd < ur ? 1 : 0);                                  // units of this XCD
    const int u0_x = xcd < ur ? xcd * (uq + 1) : ur * (uq + 1) + (xcd - ur) * uq;
    const int count_x = nu_x * m_blocks;
    if (i >= count_x) return;
    int rank = i;
    {
        const int q4 = i >> 2, s4 = i & 3;
        if (q4 * 4 + 4 <= count_x) rank = q4 * 4 + ((q4 & 1) ? 3 - s4 : s4);     // snake over the 4 shader engines
    }
    const int sg_items = PF_SGU * m_blocks, sg = rank / sg_items, rr = rank - sg * sg_items;
    const int units_in_sg = min(PF_SGU, nu_x - sg * PF_SGU);
    const int mpos = rr / units_in_sg, uu = rr - mpos * units_in_sg;
    const int unit = u0_x + sg * PF_SGU + uu;
    const int b = unit / p.h, hq = unit - b * p.h;                               // q heads of a kv group are adjacent
    const int mblk = m_blocks - 1 - mpos;                                        // longest (most keys) first
    const int hk_ = hq / (p.h / p.h_k);
    const SeqInfo si(p, b);
    const int m0 = mblk * PF_BM;
    if (m0 >= si.len_q) return;
    const int hk = hk_;
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

    // LDS ring of NB tiles, prefetch distance NB - 1, ONE barrier per tile:
    //   wait for this wave's pieces of tile t (later tiles may stay in flight) -> barrier (everyone's
    //   pieces of tile t are in LDS, and everyone is done reading tile t-1) -> issue tile t+NB-1 into the
    //   buffer tile t-1 used -> MFMAs of tile t.
#pragma unroll
    for (int s0 = 0; s0 < NB - 1; ++s0)
        if (s0 < n_tiles) ld.issue(s0, smem + s0 * 2 * TILEB);
#ifdef PF_TIMING
    float tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
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
            const uint32_t kt = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(smem + buf * 2 * TILEB);
            const uint32_t vt = kt + TILEB;
            // Per-tile read bases = buffer address + the lane's constant offset.  They are made opaque
            // so that the remaining constants (32-key half, 16-key step, +8 rows) fold into the ds_read
            // `offset:` immediates instead of costing one v_add per LDS read (the reassociator would
            // otherwise pair the constant with the uniform buffer address).
            uint32_t kb_[NJ], vb_[NDB];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { kb_[j] = kt + koff[j]; asm volatile("" : "+v"(kb_[j])); }
#pragma unroll
            for (int db = 0; db < NDB; ++db) { vb_[db] = vt + voff[db]; asm volatile("" : "+v"(vb_[db])); }
            // ---- S^T[key][query] for the two 32-key halves ----
            f32x16_v s[2];
            if (PRIO) __builtin_amdgcn_s_setprio(1);   // MFMA clusters outrank the partner wave's softmax VALU
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
            // the two 32-key halves are independent accumulators: alternate them so that no MFMA
            // waits for the previous one's result
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const u32x4_v av = *(const __attribute__((address_space(3))) u32x4_v *)(uintptr_t)(kb_[j] + blk * 32 * ROWB);
                    uint4 a;
                    a.x = av[0]; a.y = av[1]; a.z = av[2]; a.w = av[3];
                    s[blk] = mfma32<T>(a, qf[j], s[blk]);
                }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef PF_TIMING
            asm volatile("" :: "v"(s[0][0]), "v"(s[1][15]));
#endif
            PF_T(2);
            // ---- mask (diagonal / tail tiles only), online softmax in the exp2 domain ----
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
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * sl2);
            const float ms = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - ms);
            m_run = m_new;
            uint4 pp[2][2];  // P^T operands: [32-key half][16-key k-step]
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
#ifdef PF_TIMING
            asm volatile("" :: "v"(pp[0][0].x), "v"(pp[1][1].w));
#endif
            PF_T(3);
            // ---- O^T[d][query] += V^T . P^T ----
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            // (32-key half, 16-key step) outer, the NDB independent accumulators inner
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        uint4 a;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const uint32_t addr = vb_[db] + (blk * 32 + kk * 16 + 8 * half) * ROWB;
                            const short4_v r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) short4_v *)(uintptr_t)addr);
                            const uint2 r2 = __builtin_bit_cast(uint2, r4);
                            if (half == 0) { a.x = r2.x; a.y = r2.y; } else { a.z = r2.x; a.w = r2.y; }
                        }
                        oacc[db] = mfma32<T>(a, pp[blk][kk], oacc[db]);
                    }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef PF_TIMING
            asm volatile("" :: "v"(oacc[0][0]), "v"(oacc[NDB - 1][15]));
#endif
            PF_T(4);
        }
        buf = buf + 1 == NB ? 0 : buf + 1;
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
    const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
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

bool prefill_mfma_supported(const AttnParams &p) {
    return (p.d == 64 || p.d == 128) && p.alibi_slopes == nullptr && p.seqlen_q > 1;
}

template <typename T, int D, bool CAUSAL, int W, int NB, bool PRIO>
static void launch_pf_cfg(const AttnParams &p, hipStream_t stream) {
    constexpr int smem = NB * 2 * PF_BN * D * 2;
    static bool attr_set = false;  // up to 96 KiB: above the default dynamic-LDS limit
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_mfma_kernel<T, D, CAUSAL, W, NB, PRIO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    const int64_t m_blocks = cdiv(p.seqlen_q, 32 * W), n_units = (int64_t)p.b * p.h;
    const int64_t nu_max = n_units / 8 + (n_units % 8 ? 1 : 0);
    dim3 grid((unsigned)(8 * nu_max * m_blocks));   // padded: see the mapping comment in the kernel
    hipLaunchKernelGGL((prefill_mfma_kernel<T, D, CAUSAL, W, NB, PRIO>), grid, dim3(64 * W), smem, stream, p);
    ATOMA_CHECK_LAUNCH("prefill_mfma_kernel");
}

// atoma_set_option("prefill_cfg", bits): bit 0 = 8 waves x 3 buffers (else 4 waves x 2 buffers, two
// workgroups per CU), bit 1 = s_setprio around the MFMA clusters.
int prefill_cfg = PREFILL_DEFAULT_CFG;

template <typename T, int D, bool CAUSAL>
static void launch_pf(const AttnParams &p, hipStream_t stream) {
    switch (prefill_cfg & 3) {
        case 0: launch_pf_cfg<T, D, CAUSAL, 4, 2, false>(p, stream); break;
        case 1: launch_pf_cfg<T, D, CAUSAL, 8, 3, false>(p, stream); break;
        case 2: launch_pf_cfg<T, D, CAUSAL, 4, 2, true>(p, stream); break;
        default: launch_pf_cfg<T, D, CAUSAL, 8, 3, true>(p, stream); break;
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
