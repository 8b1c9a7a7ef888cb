// Projections of a decode step at 17..64 rows: the LDS-DMA tile kernel (llama.rs:269-271,311,364-365 at the batch
// sizes of the configs[2] tails and of one tensor-parallel rank of configs[3], where the four projections of a layer are 16-117 MB
// streams and every launch, ramp and merge pass costs as much as the stream itself).
//
// What bounds these products was measured first (tools/probes/gemm64_probe.hip, tools/probes/ks_ab.py):
//   * a CU keeps only ~40-50 KB of vector-memory requests outstanding, whatever the program asks for: HBM requests (~2 us) and the L2
//     hits of the x operand (~0.6 us) take their turns in the same queue, so their times ADD (x-only 15 us + W-only 25 us = 40 us for
//     gate/up with 16-row tiles) -- the x operand must be re-read as rarely as possible, i.e. many weight rows per workgroup;
//   * 8 wavefronts per CU stream 25 % faster than 4, and every CU must have a workgroup (224 of 256: -12 %);
//   * full 128-byte lines per row (256 B per row and instruction) beat the 64-byte pieces of the MFMA operand layout by 15 %.
// Hence: a workgroup of 8 wavefronts owns NW = 32 / 64 / 128 weight rows (PAIR: NW/2 gate + the NW/2 matching up rows) x all 64 batch
// rows over a K range; BOTH operands arrive in LDS by the global->LDS DMA in 4 rows x 256 B pieces (no staging registers, no
// ds_write; image [row][16-byte slot ^ (row & 15)]: the swizzle sits on the per-lane SOURCE address, fragment reads are
// conflict-free), ring of 3-4 chunks of 128 inputs, one barrier per chunk, counted vmcnt so that the ring never drains;
// v_mfma_f32_16x16x32 on ds_read_b128 fragments: wavefront w multiplies batch tile w >> 1 with half of the row groups.
// K is split over 1 .. 8 workgroups so that the grid fills the CUs; a split is merged INSIDE the launch: every workgroup publishes
// its fp32 tile (write-through stores), the one that arrives LAST adds the others' in split order and runs the epilogue (arrival counter
// per tile, agent scope; no spinning: the earlier arrivers just leave).  More splits (matrices with a handful of 64-row tiles) go to fp32
// partials for the caller's reduce / RoPE kernel.  fp32 accumulation in K order (split 0 + split 1 + ..), one rounding, then the epilogue
// with the reference's rounding points -- the same for the paired and the plain kernel on the same matrix.  Epilogues: none / residual
// add / SiLU.up (MODE 1) / RoPE + KV-cache write (MODE 2, atoma_linear_decode_qkv_rope_cache).
#include "linear_params.h"
#include "sync_ticket.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string>
#include <type_traits>

namespace atoma {

sync_word_t *sync_counters(hipStream_t stream);   // runtime.hip: epoch-tagged arrival words (sync_ticket.h)

// One 1 KiB global->LDS DMA: LDS destination = wave-uniform byte address (M0) + lane * 16, source = wave-uniform 64-bit base + 32-bit
// lane offset.  Inline asm: hipcc neither counts it nor drains it; the consumer waits with vm_wait<N>() ahead of the barrier.
__device__ __forceinline__ void tile_dma(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void tile_dma_nt(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ uint64_t tile_uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ void tile_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 2 (the q/k/v projection with RoPE and the KV-cache write as its epilogue, atoma_linear_decode_qkv_rope_cache)
struct TileRope {
    const uint16_t *cos_t, *sin_t;       // [table rows][head_dim / 2]
    const int64_t *positions, *slot_mapping;
    uint16_t *k_cache, *v_cache;
    int64_t block_stride, table_rows;    // elements; rope_pos() clamp
    int heads_q, heads_kv, head_dim, page_size, per_op;
};
struct TileParams {
    LinearParams p;
    float *slabs;                // in-launch merge: [tile][split][wave][group][lane] float4
    sync_word_t *counters;       // arrival word per tile (epoch-tagged: any state left by earlier launches is ignored -- sync_ticket.h)
    int chunks_per_split;        // chunks of 128 inputs
    int merge_splits;            // 2..8: K split this many ways and merged here by the last workgroup to arrive; 0: no in-launch merge
    TileRope rope;
    int w_plain;                 // probe (ATOMA_LINEAR_TILE_W_NT=0): weight pieces without the non-temporal hint
};
constexpr int TILE_PLAIN = 0, TILE_GATE_UP = 1, TILE_ROPE = 2;

// MODE: TILE_PLAIN rows n0 .. n0 + NW; TILE_GATE_UP (PAIR) NW/2 gate rows + the NW/2 matching up rows (p.n / 2 further down), SiLU.up
// epilogue; TILE_ROPE the same pairing at a distance of head_dim / 2 inside every head's block of rows -- a lane then holds both partners
// of a rotation -- with RoPE (q, k heads) and the KV-cache write (k, v heads) as the epilogue.
template <typename T, int NW, int MODE>
__global__ void __launch_bounds__(512, 1) linear_tile_kernel(const TileParams tp) {
    constexpr bool PAIR = MODE != TILE_PLAIN;
    const LinearParams &p = tp.p;
    constexpr int WAVES = 8;
    constexpr int NSLOT = NW == 128 ? 3 : 4;
    constexpr int WT = NW * 256, XT = 64 * 256, SLOT = WT + XT;
    constexpr int PW = NW / 4, P = PW + 16, PPW = P / WAVES;      // DMA pieces (1 KiB = 4 rows x 256 B) per chunk / per wavefront
    constexpr int G = NW / 16, GPW = G / 2;                       // 16-row groups of the tile; groups per wavefront (one batch tile each)
    static_assert(P % WAVES == 0 && GPW >= 1, "tile shape");
    // PAIR: a wavefront holds gate groups and the matching up groups -- except at NW = 32 (one gate and one up group per batch tile):
    // there the up wavefront hands its tile to the gate wavefront through LDS before the epilogue
    static_assert(!PAIR || GPW % 2 == 0 || NW == 32, "PAIR tile shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = lane >> 4, col = lane & 15;
    const int out_n = MODE == TILE_GATE_UP ? p.n / 2 : p.n;
    const int tiles = p.n / NW;
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    // tile row r -> W row: plain n0 + r; PAIR: the first NW/2 rows are rows n0 + r, the others their partners pair_dist further down
    // (gate / up: p.n / 2; RoPE: head_dim / 2 inside the head's block of rows -- tiles per head = head_dim / NW)
    const int half = MODE == TILE_ROPE ? tp.rope.head_dim / 2 : 0;
    const int pair_dist = MODE == TILE_GATE_UP ? out_n : half;
    int n0 = tile * NW;
    if (MODE == TILE_GATE_UP) n0 = tile * (NW / 2);
    if (MODE == TILE_ROPE) { const int tph = half / (NW / 2); n0 = (tile / tph) * tp.rope.head_dim + (tile % tph) * (NW / 2); }
    const int chunks_all = p.k >> 7;
    const int c0 = split * tp.chunks_per_split, c1 = min(c0 + tp.chunks_per_split, chunks_all), chunks = c1 - c0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // piece q of a chunk: q < PW: tile rows 4q..4q+3 of W, else x rows 4(q - PW)..; the lane fills (row, slot) = (4q' + (lane >> 4), lane & 15)
    // from the de-swizzled source piece (lane & 15) ^ (row & 15); x rows beyond the batch re-read the last row (their columns are never stored)
    uint32_t voff[PPW], dst[PPW];
    bool isw[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        isw[i] = q < PW;
        const int row = 4 * (isw[i] ? q : q - PW) + (lane >> 4);
        const int64_t src_row = isw[i] ? (PAIR && row >= NW / 2 ? (int64_t)pair_dist + row - NW / 2 : (int64_t)row) : (int64_t)min(row, p.batch - 1);
        voff[i] = (uint32_t)(src_row * (isw[i] ? p.w_row_stride : p.x_row_stride) * 2 + ((lane & 15) ^ (row & 15)) * 16);
        dst[i] = (isw[i] ? 0 : WT) + 4 * (isw[i] ? q : q - PW) * 256;
    }
    const uint64_t wb = tile_uniform64((uint64_t)(p.w + (int64_t)n0 * p.w_row_stride)) + (uint64_t)c0 * 256;
    const uint64_t xb = tile_uniform64((uint64_t)p.x) + (uint64_t)c0 * 256;
    auto issue = [&](int chunk, int slot) {
        const uint32_t sl = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (isw[i]) { if (tp.w_plain) tile_dma(wb + (uint64_t)chunk * 256, voff[i], sl + dst[i]); else tile_dma_nt(wb + (uint64_t)chunk * 256, voff[i], sl + dst[i]); }
            else tile_dma(xb + (uint64_t)chunk * 256, voff[i], sl + dst[i]);
        }
    };
    // wavefront -> batch tile ct and GPW row groups: plain: groups h.GPW ..; PAIR: gate groups h.GPW/2 .. and the matching up groups
    const int ct = wave >> 1, h = wave & 1;
    int grow[GPW];                                                 // first tile row of each of the wavefront's groups
#pragma unroll
    for (int a = 0; a < GPW; ++a) {
        if constexpr (PAIR && GPW >= 2) grow[a] = (a < GPW / 2 ? 0 : NW / 2) + 16 * (h * (GPW / 2) + a % (GPW / 2));
        else grow[a] = 16 * (h * GPW + a);                         // (PAIR at NW = 32: h = 0 gate rows, h = 1 up rows)
    }
    lf32x4 acc[GPW];
#pragma unroll
    for (int a = 0; a < GPW; ++a) acc[a] = lf32x4{0.f, 0.f, 0.f, 0.f};
    const bool live = 16 * ct < p.batch;                           // batch tiles beyond the batch: no arithmetic (the wavefront still moves its pieces)
    const int b_off = WT + (16 * ct + col) * 256;
    auto compute = [&](int slot) {
        const char *base = smem + slot * SLOT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sw = ((4 * s + grp) ^ col) * 16;
            const lu32x4 b = *reinterpret_cast<const lu32x4 *>(base + b_off + sw);
#pragma unroll
            for (int a = 0; a < GPW; ++a) {
                const lu32x4 av = *reinterpret_cast<const lu32x4 *>(base + (grow[a] + col) * 256 + sw);
                acc[a] = lin_mfma<T>(av, b, acc[a]);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
        if (s < chunks) issue(s, s);
    int slot = 0;
    for (int c = 0; c < chunks; ++c) {
        // this wavefront's pieces of chunk c have landed (pieces of later chunks stay in flight), then everybody's
        if (c + NSLOT - 2 < chunks) tile_vm_wait<PPW * (NSLOT - 2)>(); else tile_vm_wait<0>();
        __builtin_amdgcn_s_barrier();                              // ... and every wavefront is past its reads of chunk c - 1: its slot is free
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        if (c + NSLOT - 1 < chunks) issue(c + NSLOT - 1, pslot);
        if (live) compute(slot);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    const int brow = 16 * ct + col;
    // lane holds y^T[row of group a + 4.grp + i][batch row brow]
    if (p.partial) {                                               // more than two K splits: fp32 partials for the caller's reduce kernel
        if (brow >= p.batch) return;
#pragma unroll
        for (int a = 0; a < GPW; ++a) {
            const int r = grow[a] + 4 * grp;
            const int n = PAIR ? (r >= NW / 2 ? pair_dist + n0 + r - NW / 2 : n0 + r) : n0 + r;
            *reinterpret_cast<float4 *>(p.partial + ((int64_t)split * p.batch + brow) * p.n + n) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        }
        return;
    }
    if (tp.merge_splits > 1) {
        // publish this workgroup's fp32 tile write-through (sc1: straight to memory, no release fence), drain, take a ticket; the
        // LAST arriver reads the other splits' tiles (sc1 loads: past its own L1 / L2) and finishes.  The sum runs in split order
        // (its own tile taken from registers at its own place), so the result does not depend on who came last.
        const int S = tp.merge_splits;
        float *mine = tp.slabs + ((int64_t)(tile * S + split) * WAVES + wave) * GPW * 256;
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(mine, 0, GPW * 1024, 0x00020000);
#pragma unroll
        for (int a = 0; a < GPW; ++a)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lu32x4, acc[a]), sr, (a * 64 + lane) * 16, 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *ticket = reinterpret_cast<unsigned *>(smem);    // the ring is idle now (every DMA was waited for)
        if (tid == 0) *ticket = sync_arrive(tp.counters + tile, sync_epoch(), (unsigned)S);
        __syncthreads();
        if (*ticket + 1 != (unsigned)S) return;                    // not the last: somebody else finishes the tile
        lf32x4 tot[GPW];
        for (int sp = 0; sp < S; ++sp) {
            float *theirs = tp.slabs + ((int64_t)(tile * S + sp) * WAVES + wave) * GPW * 256;
            const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, GPW * 1024, 0x00020000);
#pragma unroll
            for (int a = 0; a < GPW; ++a) {
                lf32x4 o = acc[a];
                if (sp != split) o = __builtin_bit_cast(lf32x4, __builtin_amdgcn_raw_buffer_load_b128(tr, (a * 64 + lane) * 16, 0, 16 /* sc1 */));
                tot[a] = sp == 0 ? o : tot[a] + o;
            }
        }
#pragma unroll
        for (int a = 0; a < GPW; ++a) acc[a] = tot[a];
        __syncthreads();                                           // (the ticket word is reused below)
    }
    if constexpr (MODE == TILE_ROPE) {
        // ---- RoPE (q and k heads) + KV-cache write (k and v heads): rope_cache_kernel's arithmetic on the projection's ROUNDED output ----
        const TileRope &rp = tp.rope;
        lf32x4 other[GPW == 1 ? 1 : GPW];
        if constexpr (GPW == 1) {                                  // NW = 32: the partner group lives in the neighbouring wavefront (h ^ 1)
            lf32x4 *xch = reinterpret_cast<lf32x4 *>(smem + 1024);
            __syncthreads();
            xch[(ct * 2 + h) * 64 + lane] = acc[0];
            __syncthreads();
            other[0] = xch[(ct * 2 + (h ^ 1)) * 64 + lane];
        }
        if (brow >= p.batch) return;
        const int tph = half / (NW / 2);
        const int head = tile / tph;
        const bool is_v = head >= rp.heads_q + rp.heads_kv, is_k = !is_v && head >= rp.heads_q;
        const int64_t slot_ix = rp.slot_mapping[brow];
        int64_t pos = rp.positions[brow];
        pos = rp.table_rows > 0 ? (pos < 0 ? 0 : (pos >= rp.table_rows ? rp.table_rows - 1 : pos)) : pos;      // rope_pos() of norm_rope.hip
        const int64_t crow = slot_ix >= 0 ? (slot_ix / rp.page_size) * rp.block_stride + (slot_ix % rp.page_size) * (int64_t)rp.heads_kv * rp.head_dim : 0;
        constexpr int NP = GPW == 1 ? 1 : GPW / 2;                 // partner pairs held by this lane
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            // x1 = first-half element, x2 = its partner half further; this lane finishes `mine` of them (both when it holds both)
            const lf32x4 &g1 = GPW == 1 ? (h == 0 ? acc[0] : other[0]) : acc[a];
            const lf32x4 &g2 = GPW == 1 ? (h == 0 ? other[0] : acc[0]) : acc[a + GPW / 2];
            const int j0 = (GPW == 1 ? 0 : grow[a]) + (tile % tph) * (NW / 2) + 4 * grp;   // index inside the half: 0 .. half - 1
            float x1[4], x2[4], y1[4], y2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { x1[i] = round_through<T>(g1[i]); x2[i] = round_through<T>(g2[i]); }
            if (is_v) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { y1[i] = x1[i]; y2[i] = x2[i]; }
            } else {
#pragma clang fp contract(off)
                const uint2 cw = *reinterpret_cast<const uint2 *>(rp.cos_t + pos * half + j0), sw = *reinterpret_cast<const uint2 *>(rp.sin_t + pos * half + j0);
                const float cs[4] = {lo_to_f32<T>(cw.x), hi_to_f32<T>(cw.x), lo_to_f32<T>(cw.y), hi_to_f32<T>(cw.y)};
                const float sn[4] = {lo_to_f32<T>(sw.x), hi_to_f32<T>(sw.x), lo_to_f32<T>(sw.y), hi_to_f32<T>(sw.y)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (rp.per_op) {
                        y1[i] = round_through<T>(x1[i] * cs[i]) - round_through<T>(x2[i] * sn[i]);
                        y2[i] = round_through<T>(x1[i] * sn[i]) + round_through<T>(x2[i] * cs[i]);
                    } else {
                        y1[i] = x1[i] * cs[i] - x2[i] * sn[i];
                        y2[i] = x1[i] * sn[i] + x2[i] * cs[i];
                    }
                }
            }
            const int64_t col0 = (int64_t)head * rp.head_dim + j0;
            const int hk = is_v ? head - rp.heads_q - rp.heads_kv : head - rp.heads_q;
            uint16_t *cache = is_v ? rp.v_cache : rp.k_cache;
            auto put = [&](const float (&y)[4], int64_t c) {
                uint2 o;
                o.x = pack2<T>(y[0], y[1]);
                o.y = pack2<T>(y[2], y[3]);
                *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + c) = o;
                if ((is_k || is_v) && slot_ix >= 0) *reinterpret_cast<uint2 *>(cache + crow + (int64_t)hk * rp.head_dim + (c - (int64_t)head * rp.head_dim)) = o;
            };
            if (GPW > 1 || h == 0) put(y1, col0);
            if (GPW > 1 || h == 1) put(y2, col0 + half);
        }
        return;
    }
    if constexpr (MODE == TILE_GATE_UP && GPW == 1) {              // NW = 32: up tile -> LDS -> the gate wavefront of the same batch tile
        lf32x4 *xch = reinterpret_cast<lf32x4 *>(smem + 1024);    // (the ring is idle; the first KiB may hold the merge ticket)
        __syncthreads();
        if (h == 1) xch[ct * 64 + lane] = acc[0];
        __syncthreads();
        if (h == 1 || brow >= p.batch) return;
        const lf32x4 up = xch[ct * 64 + lane];
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = round_through<T>(acc[0][i]);
            v[i] = round_through<T>(g / (1.f + __expf(-g))) * round_through<T>(up[i]);
        }
        uint2 o;
        o.x = pack2<T>(v[0], v[1]);
        o.y = pack2<T>(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n0 + 4 * grp) = o;
        return;
    }
    if (brow >= p.batch) return;
    if constexpr (MODE == TILE_GATE_UP && GPW >= 2) {              // rounding points as in linear_reduce_kernel
#pragma unroll
        for (int a = 0; a < GPW / 2; ++a) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = round_through<T>(acc[a][i]);
                v[i] = round_through<T>(g / (1.f + __expf(-g))) * round_through<T>(acc[a + GPW / 2][i]);
            }
            uint2 o;
            o.x = pack2<T>(v[0], v[1]);
            o.y = pack2<T>(v[2], v[3]);
            *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n0 + grow[a] + 4 * grp) = o;
        }
    } else if constexpr (MODE == TILE_PLAIN) {
#pragma unroll
        for (int a = 0; a < GPW; ++a) {
            const int n = n0 + grow[a] + 4 * grp;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = round_through<T>(acc[a][i]);
            if (p.epilogue == 1) {
                const uint2 rr = *reinterpret_cast<const uint2 *>(p.aux + (int64_t)brow * p.aux_row_stride + n);
                v[0] += lo_to_f32<T>(rr.x); v[1] += hi_to_f32<T>(rr.x); v[2] += lo_to_f32<T>(rr.y); v[3] += hi_to_f32<T>(rr.y);
            }
            uint2 o;
            o.x = pack2<T>(v[0], v[1]);
            o.y = pack2<T>(v[2], v[3]);
            *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n) = o;
        }
    }
}

static int tile_env_or(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
// knobs: environment at load time, atoma_set_option("linear_tile*") at run time (A/B runs inside one process)
static std::atomic<int> linear_tile_on{tile_env_or("ATOMA_LINEAR_TILE", 1)};          // 0: the older kernels serve 17..64 rows
static std::atomic<int> linear_tile_nw{tile_env_or("ATOMA_LINEAR_TILE_NW", 0)};       // weight rows per workgroup: 0 = by shape
static std::atomic<int> linear_tile_splits{tile_env_or("ATOMA_LINEAR_TILE_SPLITS", 0)};   // K splits: 0 = by shape
static std::atomic<int> linear_tile_max_splits{tile_env_or("ATOMA_LINEAR_TILE_MAX_SPLITS", 8)};   // largest split the plan considers (A/B: 4 = round 3's plans)
bool set_linear_tile_option(const std::string &name, int value) {
    if (name == "linear_tile") linear_tile_on = value;
    else if (name == "linear_tile_nw") linear_tile_nw = value;
    else if (name == "linear_tile_splits") linear_tile_splits = value;
    else if (name == "linear_tile_max_splits") linear_tile_max_splits = value;
    else return false;
    return true;
}

// Workgroup shape and K split from the SHAPE OF W alone (p.n rows, p.k inputs), so that the stacked gate / up launch with its SiLU.up
// epilogue, the residual epilogue, the RoPE epilogue and the plain projection of the same matrix split K alike and stay bit-identical
// to projection + separate op.  Candidates: 128 / 64 / 32 rows per workgroup x 1 .. 8 K splits (a split merges inside the launch),
// priced with the measured model of DESIGN.md 4.8c -- a CU's time = (weight KB + 0.3 x KB of x) x 0.04 us, plus 1 us + 0.05 us per KB
// the last arriver reads back for an in-launch merge, times the rounds of workgroups over the CUs -- e.g. (64 rows, 2 splits) for the
// 70B shard's gate/up and down, (32 rows, 1 split) for its o projection, (32 rows, 4 splits) for its 1280-row q/k/v shard.  A matrix with
// so few rows that even the best candidate leaves more than 40 % of the CUs idle is split further over K (powers of two) and leaves
// fp32 partials for the caller's reduce / RoPE kernel.
constexpr int TILE_MAX_MERGE = 8;
static void tile_plan(int64_t n, int64_t k, int cus, int *nw_out, int *splits_out) {
    const int64_t chunks = k / 128;
    int best_nw = 0, best_s = 1;
    int64_t best_wgs = 0;
    double best_t = 1e30;
    for (int s : {1, 2, 3, 4, 5, 6, 8})      // (uneven splits are fine: the last workgroup of a tile takes the remainder)
        for (int nw : {128, 64, 32}) {
            if (n % nw || chunks / s < 4 || n / nw > 4096 || s > linear_tile_max_splits) continue;
            const int64_t wgs = n / nw * s;
            const double kb = (double)(k / s) * 2.0 / 1024.0;
            const double merge = s == 1 ? 0.0 : 1.0 + 0.05 * (nw * 64 * 4 / 1024.0) * (s - 1);
            const double t = (double)cdiv(wgs, cus) * ((nw + 0.3 * 64) * kb * 0.04 + merge);
            if (t < best_t) { best_t = t; best_wgs = wgs; best_nw = nw; best_s = s; }
        }
    if (best_nw && best_wgs * 10 >= (int64_t)cus * 6) { *nw_out = best_nw; *splits_out = best_s; return; }
    // few rows: 64-row tiles, K split in powers of two up to one round of the CUs
    const int nw = 64;
    if (n % nw) { *nw_out = 0; return; }
    int64_t s = 1;
    while (n / nw * s * 2 <= cus && chunks / (s * 2) >= 4) s *= 2;
    *nw_out = nw;
    *splits_out = (int)s;
}

// nw = 0: not served
static void tile_route(const LinearParams &p, int *nw_out, int *splits_out) {
    *nw_out = 0;
    *splits_out = 1;
    // p.n % 64: the paired kernels need 64-row tiles at least once, and the plain kernel must take the same route for the same matrix
    if (!linear_tile_on || p.k % 128 || p.n % 64 || p.batch > 64 || p.batch < 1) return;
    int nw = 0, splits = 1;
    tile_plan(p.n, p.k, device_num_cus(), &nw, &splits);
    if (linear_tile_nw > 0 && p.n % linear_tile_nw == 0) nw = linear_tile_nw;
    if (linear_tile_splits > 0) splits = linear_tile_splits;
    if ((nw != 32 && nw != 64 && nw != 128) || p.n % nw) return;
    const int64_t chunks = p.k / 128;
    splits = (int)cdiv(chunks, cdiv(chunks, splits));
    if (splits > 1 && splits <= TILE_MAX_MERGE && p.n / nw > 4096) return;   // one arrival counter per tile
    *nw_out = nw;
    *splits_out = splits;
}

bool linear_tile_leaves_partials(const LinearParams &p, int dtype) {
    (void)dtype;
    int nw, splits;
    tile_route(p, &nw, &splits);
    return nw != 0 && splits > TILE_MAX_MERGE;
}
// the RoPE epilogue can ride on this product: served, merged inside the launch (or not split), and a tile holds both halves of a head
bool linear_tile_can_rope(const LinearParams &p, int head_dim) {
    int nw, splits;
    tile_route(p, &nw, &splits);
    return nw != 0 && splits <= TILE_MAX_MERGE && head_dim % 32 == 0 && nw / 2 <= head_dim / 2 && (head_dim / 2) % (nw / 2) == 0;
}

template <typename T> static int launch_linear_tile_t(LinearParams &p, hipStream_t stream, const TileRope *rope) {
    int nw, splits;
    tile_route(p, &nw, &splits);
    if (nw == 0) return 1;
    const int64_t chunks = p.k / 128;
    TileParams tp{};
    tp.chunks_per_split = (int)cdiv(chunks, splits);
    splits = (int)cdiv(chunks, tp.chunks_per_split);
    p.splits = splits;
    p.partial = nullptr;
    const int64_t tiles = p.n / nw;
    if (splits > 1 && splits <= TILE_MAX_MERGE) {
        tp.merge_splits = splits;
        tp.slabs = static_cast<float *>(workspace(stream, (size_t)tiles * splits * nw * 64 * sizeof(float)));
        tp.counters = sync_counters(stream);
        if (!tp.slabs || !tp.counters) return -1;
    } else if (splits > TILE_MAX_MERGE) {
        if (rope) return 1;
        p.partial = static_cast<float *>(workspace(stream, (size_t)splits * p.batch * p.n * sizeof(float)));
        if (!p.partial) return -1;
    }
    tp.p = p;
    static const int w_nt = tile_env_or("ATOMA_LINEAR_TILE_W_NT", 1);
    tp.w_plain = w_nt ? 0 : 1;
    if (rope) tp.rope = *rope;
    const int mode = rope ? TILE_ROPE : (p.epilogue == 2 ? TILE_GATE_UP : TILE_PLAIN);
    const dim3 grid((unsigned)(tiles * splits)), block(512);
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
#define ATOMA_TILE_M(NW_, MODE_) do { \
        const size_t lds = (size_t)((NW_) == 128 ? 3 : 4) * ((NW_) * 256 + 64 * 256); \
        static std::atomic<bool> once[64];   /* per device: the attribute belongs to the device's code object */ \
        if (!once[dev].load()) { if (!check_hip(hipFuncSetAttribute((const void *)linear_tile_kernel<T, NW_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "linear_tile LDS")) return -1; once[dev] = true; } \
        hipLaunchKernelGGL((linear_tile_kernel<T, NW_, MODE_>), grid, block, lds, stream, tp); } while (0)
#define ATOMA_TILE(NW_) do { if (mode == TILE_ROPE) ATOMA_TILE_M(NW_, TILE_ROPE); else if (mode == TILE_GATE_UP) ATOMA_TILE_M(NW_, TILE_GATE_UP); else ATOMA_TILE_M(NW_, TILE_PLAIN); } while (0)
    if (nw == 128) ATOMA_TILE(128); else if (nw == 64) ATOMA_TILE(64); else ATOMA_TILE(32);
#undef ATOMA_TILE
#undef ATOMA_TILE_M
    return ATOMA_CHECK_LAUNCH("linear_tile_kernel") ? 0 : -1;
}

// atoma_warmup: the kernels ask for more LDS than the default limit -- raise it for every variant on the current device now, so that a
// hipGraph capture can be the first call (hipFuncSetAttribute is not a stream operation, but it has no business inside a capture)
template <typename T, int NW, int MODE> static bool tile_prepare_one() {
    const size_t lds = (size_t)(NW == 128 ? 3 : 4) * (NW * 256 + 64 * 256);
    return check_hip(hipFuncSetAttribute((const void *)linear_tile_kernel<T, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "linear_tile LDS");
}
template <typename T> static bool tile_prepare_t() {
    return tile_prepare_one<T, 32, TILE_PLAIN>() && tile_prepare_one<T, 64, TILE_PLAIN>() && tile_prepare_one<T, 128, TILE_PLAIN>() &&
           tile_prepare_one<T, 32, TILE_GATE_UP>() && tile_prepare_one<T, 64, TILE_GATE_UP>() && tile_prepare_one<T, 128, TILE_GATE_UP>() &&
           tile_prepare_one<T, 32, TILE_ROPE>() && tile_prepare_one<T, 64, TILE_ROPE>() && tile_prepare_one<T, 128, TILE_ROPE>();
}
bool linear_tile_prepare() { return tile_prepare_t<bf16_t>() && tile_prepare_t<f16_t>(); }

// 0 = launched (p.partial set when more than TILE_MAX_MERGE K splits left fp32 partials: the caller runs linear_reduce_kernel), 1 = shape
// not served, -1 = error
int launch_linear_tile(LinearParams &p, int dtype, hipStream_t stream) {
    return dtype == ATOMA_BF16 ? launch_linear_tile_t<bf16_t>(p, stream, nullptr) : launch_linear_tile_t<f16_t>(p, stream, nullptr);
}
// the q/k/v projection with RoPE + KV-cache write as its epilogue (one launch); only when linear_tile_can_rope
int launch_linear_tile_rope(LinearParams &p, int dtype, hipStream_t stream, const uint16_t *cos_t, const uint16_t *sin_t, const int64_t *positions,
                            const int64_t *slot_mapping, uint16_t *k_cache, uint16_t *v_cache, int64_t block_stride, int64_t table_rows, int heads_q,
                            int heads_kv, int head_dim, int page_size, int per_op) {
    TileRope r{cos_t, sin_t, positions, slot_mapping, k_cache, v_cache, block_stride, table_rows, heads_q, heads_kv, head_dim, page_size, per_op};
    return dtype == ATOMA_BF16 ? launch_linear_tile_t<bf16_t>(p, stream, &r) : launch_linear_tile_t<f16_t>(p, stream, &r);
}

}  // namespace atoma
