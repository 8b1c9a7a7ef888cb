// Projections of a decode step at 65..256 rows (continuous batching at bs <= 256, BASELINE configs[2]; SURVEY.md 8f item 1;
// llama.rs:269-271,311,364-365): the LDS-DMA tile kernel of linear_tile.hip grown to a 128 / 256-row batch tile.
//
// What bounds these products (re-measured for 256 rows in round 6): 2.B = 512 flop per weight byte is the machine
// balance, but the wall is neither HBM nor the matrix pipe -- it is the CU's vector-memory path.  A CU keeps ~45 KB of requests
// outstanding; a weight byte (HBM, ~2 us) and a byte of the x operand (re-read from L2 by every workgroup, ~0.6 us) queue in the same
// path, so a workgroup's time ~ (W bytes x 2.0 + x bytes x 0.6) / 45 KB.  For gate/up at 256 rows (235 MB of W, 2 MB of x per
// workgroup) that is ~70 us on every decomposition that fills the chip once -- where the vendor GEMM sits too (61-65 us) -- against 30 us of
// MFMA time.  So the kernel is built around the memory path, not the matrix pipe:
//   * a workgroup of 8 wavefronts (two per SIMD: one wavefront's DMA issue stalls hide under the other's MFMAs) owns NW = 64 / 128
//     weight rows (PAIR: NW/2 gate + the NW/2 matching up rows; RoPE: both halves of a head) x BR = 128 / 192 / 256 batch rows over a K range;
//   * both operands arrive by the global->LDS DMA in full 128-byte lines (a piece = 8 rows x 128 B; image [row][16-byte slot ^ (row & 7)],
//     the swizzle on the per-lane SOURCE address, fragment reads conflict-free), chunks of 64 inputs, ring of 3-4 chunks, ONE barrier
//     per chunk, counted vmcnt so that two chunks stay in flight across every barrier; W non-temporal, x default policy (L2-resident);
//   * wavefront (h, bq): half of the tile's 16-row groups x a quarter of the batch rows: 8 fragment reads feed 16 v_mfma_f32_16x16x32;
//   * K is split over 1..8 workgroups so that the grid fills the CUs once (q/k/v, o, down: few rows, long K); the splits of a tile sit on
//     different XCDs and every XCD sees ONE K range of x (its L2 holds 256 x K/S instead of 256 x K: down's x is 7.3 MB); a split is
//     merged inside the launch by the last arriver (sync_ticket.h), in split order: the result does not depend on who came last.
// fp32 accumulation in K order, one rounding, then the epilogue with the reference's rounding points (none / residual add / SiLU.up /
// RoPE + KV-cache write) -- bit-identical to the plain projection followed by the separate op, because the plan (rows per workgroup, K
// split) comes from the shape of W alone.
#include "linear_params.h"
#include "sync_ticket.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string>
#include <type_traits>

namespace atoma {

sync_word_t *sync_counters(hipStream_t stream);   // runtime.hip

__device__ __forceinline__ void wide_dma(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void wide_dma_nt(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ uint64_t wide_uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ void wide_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct WideRope {                        // MODE 2: as TileRope of linear_tile.hip
    const uint16_t *cos_t, *sin_t;
    const int64_t *positions, *slot_mapping;
    uint16_t *k_cache, *v_cache;
    int64_t block_stride, table_rows;
    int heads_q, heads_kv, head_dim, page_size, per_op;
};
struct WideParams {
    LinearParams p;
    float *slabs;                // in-launch merge: [tile][split][wave][fragment][lane] float4
    sync_word_t *counters;       // arrival word per tile
    int chunks_per_split;        // chunks of 64 inputs
    int splits;                  // 1..8
    int xcd_map;                 // 1: workgroup b -> split (b % 8) % splits (a K range per XCD); 0: split = b / tiles
    int var;                     // probe bits (atoma_set_option("linear_wide_var")): 1 = W pieces ahead of the x pieces, 2 = W without the non-temporal hint, 4 = x non-temporal
    WideRope rope;
};
constexpr int WIDE_PLAIN = 0, WIDE_GATE_UP = 1, WIDE_ROPE = 2;

template <typename T, int NW, int BR, int MODE>
__global__ void __launch_bounds__(512, 1) linear_wide_kernel(const WideParams tp) {
    constexpr bool PAIR = MODE != WIDE_PLAIN;
    const LinearParams &p = tp.p;
    constexpr int WAVES = 8;
    constexpr int WT = NW * 128, XT = BR * 128, SLOT = WT + XT;
    constexpr int NSLOT = 4 * SLOT + 1024 <= 160 * 1024 ? 4 : 3;
    constexpr int PWW = NW / 64, PXW = BR / 64, PPW = PWW + PXW;   // DMA pieces (1 KiB = 8 rows x 128 B) per chunk and wavefront: W, x
    constexpr int GPW = NW / 32;                                   // 16-row groups of W per wavefront (half of the tile's)
    constexpr int BTW = BR / 64;                                   // 16-row batch tiles per wavefront (a quarter of the batch rows)
    static_assert(NW % 64 == 0 && BR % 64 == 0 && GPW >= 2 && GPW % 2 == 0, "tile shape");
    static_assert(NSLOT * SLOT <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = lane >> 4, col = lane & 15;
    const int out_n = MODE == WIDE_GATE_UP ? p.n / 2 : p.n;
    const int tiles = p.n / NW;
    int tile, split;
    if (tp.xcd_map) {            // 8 consecutive workgroups = one from every XCD: S of them (one per K range) x 8 / S tiles
        const int S = tp.splits, per8 = 8 / S, b8 = blockIdx.x >> 3, x = blockIdx.x & 7;
        split = x % S;
        tile = b8 * per8 + x / S;
    } else {
        tile = blockIdx.x % tiles;
        split = blockIdx.x / tiles;
    }
    if (tile >= tiles) return;
    // batch block: the launch's y dimension cuts the batch into blocks of BR rows (two blocks of 128 for a 256-row batch halve a small product's
    // slabs and x traffic per workgroup at the price of a second, L2-served pass over W: the plan decides); item = (tile, batch block)
    const int b0 = (int)blockIdx.y * BR, item = tile + tiles * (int)blockIdx.y;
    const int half = MODE == WIDE_ROPE ? tp.rope.head_dim / 2 : 0;
    const int pair_dist = MODE == WIDE_GATE_UP ? out_n : half;
    int n0 = tile * NW;
    if (MODE == WIDE_GATE_UP) n0 = tile * (NW / 2);
    if (MODE == WIDE_ROPE) { const int tph = half / (NW / 2); n0 = (tile / tph) * tp.rope.head_dim + (tile % tph) * (NW / 2); }
    const int chunks_all = p.k >> 6;
    const int c0 = split * tp.chunks_per_split, c1 = min(c0 + tp.chunks_per_split, chunks_all), chunks = c1 - c0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // every wavefront moves PWW pieces of W (tile rows 8q..8q+7, q = wave.PWW + i) and PXW pieces of x per chunk -- which piece is which is
    // known at compile time (no branch in the loop); the lane fills (row, slot) = (8q + (lane >> 3), lane & 7) from the de-swizzled source
    // piece (lane & 7) ^ (row & 7); x rows beyond the batch re-read the last row (their columns are never stored)
    uint32_t voff[PPW], dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const bool isw = i < PWW;
        const int q = isw ? wave * PWW + i : wave * PXW + (i - PWW);
        const int row = 8 * q + (lane >> 3);
        const int64_t src_row = isw ? (PAIR && row >= NW / 2 ? (int64_t)pair_dist + row - NW / 2 : (int64_t)row) : (int64_t)min(b0 + row, p.batch - 1);
        voff[i] = (uint32_t)(src_row * (isw ? p.w_row_stride : p.x_row_stride) * 2 + ((lane & 7) ^ (row & 7)) * 16);
        dst[i] = (isw ? 0 : WT) + 8 * q * 128;
    }
    const uint64_t wb = wide_uniform64((uint64_t)(p.w + (int64_t)n0 * p.w_row_stride)) + (uint64_t)c0 * 128;
    const uint64_t xb = wide_uniform64((uint64_t)p.x) + (uint64_t)c0 * 128;
    const int var = tp.var;
    auto issue = [&](int chunk, int slot) {
        const uint32_t sl = lds0 + slot * SLOT;
        auto wpieces = [&]() {
#pragma unroll
            for (int i = 0; i < PWW; ++i) { if (var & 2) wide_dma(wb + (uint64_t)chunk * 128, voff[i], sl + dst[i]); else wide_dma_nt(wb + (uint64_t)chunk * 128, voff[i], sl + dst[i]); }
        };
        auto xpieces = [&]() {
#pragma unroll
            for (int i = PWW; i < PPW; ++i) { if (var & 4) wide_dma_nt(xb + (uint64_t)chunk * 128, voff[i], sl + dst[i]); else wide_dma(xb + (uint64_t)chunk * 128, voff[i], sl + dst[i]); }
        };
        // x (L2, ~0.6 us) ahead of W (HBM, ~2 us): a wavefront's loads retire in order, so its x pieces would otherwise wait behind its W pieces
        // (measured: 1-3 % on all four 8B projections; x non-temporal: -12..-17 %; W without the hint: level)
        if (var & 1) { wpieces(); xpieces(); } else { xpieces(); wpieces(); }
    };
    // wavefront (h, bq): plain: groups h.GPW ..; PAIR: first-half groups h.GPW/2 .. and the matching partner groups; batch tiles bq.BTW ..
    const int h = wave & 1, bq = wave >> 1;
    int grow[GPW];
#pragma unroll
    for (int a = 0; a < GPW; ++a) {
        if constexpr (PAIR) grow[a] = (a < GPW / 2 ? 0 : NW / 2) + 16 * (h * (GPW / 2) + a % (GPW / 2));
        else grow[a] = 16 * (h * GPW + a);
    }
    lf32x4 acc[GPW][BTW];
#pragma unroll
    for (int a = 0; a < GPW; ++a)
#pragma unroll
        for (int b = 0; b < BTW; ++b) acc[a][b] = lf32x4{0.f, 0.f, 0.f, 0.f};
    const int bt0 = bq * BTW;                                      // first batch tile of the wavefront
    // RoPE epilogue: the token's cache slot and (clamped) position are fetched NOW, ahead of the K loop -- the epilogue runs in the last
    // arriver only, behind the merge, where a chain slot -> position -> table row -> store would be pure latency (measured: 7.5 us of a 38 us launch)
    int64_t rope_slot[BTW], rope_pos[BTW];
    if constexpr (MODE == WIDE_ROPE) {
#pragma unroll
        for (int b = 0; b < BTW; ++b) {
            const int brow = min(b0 + 16 * (bt0 + b) + col, p.batch - 1);
            rope_slot[b] = tp.rope.slot_mapping[brow];
            const int64_t pos = tp.rope.positions[brow];
            rope_pos[b] = tp.rope.table_rows > 0 ? (pos < 0 ? 0 : (pos >= tp.rope.table_rows ? tp.rope.table_rows - 1 : pos)) : pos;      // rope_pos() of norm_rope.hip
        }
    }
    int live = 0;                                                  // batch tiles of this wavefront that hold rows of the batch
#pragma unroll
    for (int b = 0; b < BTW; ++b) live += b0 + 16 * (bt0 + b) < p.batch ? 1 : 0;
    auto compute = [&](int slot) {
        const char *base = smem + slot * SLOT;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = ((4 * s + grp) ^ (col & 7)) * 16;
            lu32x4 bf[BTW], af[GPW];
#pragma unroll
            for (int b = 0; b < BTW; ++b) bf[b] = *reinterpret_cast<const lu32x4 *>(base + WT + (16 * (bt0 + b) + col) * 128 + sw);
#pragma unroll
            for (int a = 0; a < GPW; ++a) af[a] = *reinterpret_cast<const lu32x4 *>(base + (grow[a] + col) * 128 + sw);
#pragma unroll
            for (int b = 0; b < BTW; ++b)                       // (a wavefront with ANY live batch tile multiplies all of its tiles: no branch per tile)
#pragma unroll
                for (int a = 0; a < GPW; ++a) acc[a][b] = lin_mfma<T>(af[a], bf[b], acc[a][b]);
        }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
        if (s < chunks) issue(s, s);
    int slot = 0;
    for (int c = 0; c < chunks; ++c) {
        if (c + NSLOT - 2 < chunks) wide_vm_wait<PPW * (NSLOT - 2)>(); else wide_vm_wait<0>();
        __builtin_amdgcn_s_barrier();                              // chunk c has landed for everybody; everybody is past its reads of chunk c - 1
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        if (c + NSLOT - 1 < chunks) issue(c + NSLOT - 1, pslot);
        if (live) compute(slot);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    // lane holds y^T[tile row grow[a] + 4.grp + i][batch row 16.(bt0 + b) + col]
    // RoPE epilogue: ALL table reads of the lane up front (the epilogue's stores may alias the tables as far as the compiler knows, so reads
    // left inside the store loop wait one round trip per (pair, batch tile): 8 x ~0.9 us, measured); their latency rides under the merge
    uint2 rope_cw[GPW / 2][BTW], rope_sw[GPW / 2][BTW];
    if constexpr (MODE == WIDE_ROPE) {
        const int tph = half / (NW / 2);
#pragma unroll
        for (int b = 0; b < BTW; ++b)
#pragma unroll
            for (int a = 0; a < GPW / 2; ++a) {
                const int j0 = grow[a] + (tile % tph) * (NW / 2) + 4 * grp;
                rope_cw[a][b] = *reinterpret_cast<const uint2 *>(tp.rope.cos_t + rope_pos[b] * half + j0);
                rope_sw[a][b] = *reinterpret_cast<const uint2 *>(tp.rope.sin_t + rope_pos[b] * half + j0);
            }
    }
    if (tp.splits > 1) {
        // publish this workgroup's fp32 tile write-through, drain, take a ticket; the LAST arriver adds the tiles in split order and finishes -- linear_tile.hip's merge
        const int S = tp.splits;
        constexpr int F = GPW * BTW;
        float *mine = tp.slabs + ((int64_t)(item * S + split) * WAVES + wave) * F * 256;
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(mine, 0, F * 1024, 0x00020000);
#pragma unroll
        for (int a = 0; a < GPW; ++a)
#pragma unroll
            for (int b = 0; b < BTW; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lu32x4, acc[a][b]), sr, ((a * BTW + b) * 64 + lane) * 16, 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *ticket = reinterpret_cast<unsigned *>(smem);    // the ring is idle now (every DMA was waited for)
        if (tid == 0) *ticket = sync_arrive(tp.counters + item, sync_epoch(), (unsigned)S);
        __syncthreads();
        if (*ticket + 1 != (unsigned)S) return;
        // (its own tile comes back from memory like the others: the fp32 registers are free for the sum, and the order is split order anyway)
        // two splits' loads in flight together (the second descriptor is empty past the last split: its loads return zeros, no branch)
        for (int sp = 0; sp < S; sp += 2) {
            float *t0 = tp.slabs + ((int64_t)(item * S + sp) * WAVES + wave) * F * 256;
            const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(t0, 0, F * 1024, 0x00020000);
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(t0 + WAVES * F * 256, 0, sp + 1 < S ? F * 1024 : 0, 0x00020000);
            lf32x4 o0[GPW][BTW], o1[GPW][BTW];
#pragma unroll
            for (int a = 0; a < GPW; ++a)
#pragma unroll
                for (int b = 0; b < BTW; ++b) {
                    o0[a][b] = __builtin_bit_cast(lf32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, ((a * BTW + b) * 64 + lane) * 16, 0, 16 /* sc1 */));
                    o1[a][b] = __builtin_bit_cast(lf32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, ((a * BTW + b) * 64 + lane) * 16, 0, 16 /* sc1 */));
                }
#pragma unroll
            for (int a = 0; a < GPW; ++a)
#pragma unroll
                for (int b = 0; b < BTW; ++b) acc[a][b] = (sp == 0 ? o0[a][b] : acc[a][b] + o0[a][b]) + o1[a][b];
        }
    }
#pragma unroll
    for (int b = 0; b < BTW; ++b) {
        const int brow = b0 + 16 * (bt0 + b) + col;
        if (brow >= p.batch) continue;
        if constexpr (MODE == WIDE_ROPE) {
            // RoPE (q and k heads) + KV-cache write (k and v heads): rope_cache_kernel's arithmetic on the projection's ROUNDED output
            const WideRope &rp = tp.rope;
            const int tph = half / (NW / 2);
            const int head = tile / tph;
            const bool is_v = head >= rp.heads_q + rp.heads_kv, is_k = !is_v && head >= rp.heads_q;
            const int64_t slot_ix = rope_slot[b];
            const int64_t crow = slot_ix >= 0 ? (slot_ix / rp.page_size) * rp.block_stride + (slot_ix % rp.page_size) * (int64_t)rp.heads_kv * rp.head_dim : 0;
#pragma unroll
            for (int a = 0; a < GPW / 2; ++a) {
                const lf32x4 &g1 = acc[a][b], &g2 = acc[a + GPW / 2][b];
                const int j0 = grow[a] + (tile % tph) * (NW / 2) + 4 * grp;   // index inside the half: 0 .. half - 1
                float x1[4], x2[4], y1[4], y2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { x1[i] = round_through<T>(g1[i]); x2[i] = round_through<T>(g2[i]); }
                if (is_v) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { y1[i] = x1[i]; y2[i] = x2[i]; }
                } else {
#pragma clang fp contract(off)
                    const uint2 cw = rope_cw[a][b], sw = rope_sw[a][b];
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
                put(y1, col0);
                put(y2, col0 + half);
            }
        } else if constexpr (MODE == WIDE_GATE_UP) {               // rounding points as in linear_reduce_kernel
#pragma unroll
            for (int a = 0; a < GPW / 2; ++a) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g = round_through<T>(acc[a][b][i]);
                    v[i] = round_through<T>(g / (1.f + __expf(-g))) * round_through<T>(acc[a + GPW / 2][b][i]);
                }
                uint2 o;
                o.x = pack2<T>(v[0], v[1]);
                o.y = pack2<T>(v[2], v[3]);
                *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n0 + grow[a] + 4 * grp) = o;
            }
        } else {
#pragma unroll
            for (int a = 0; a < GPW; ++a) {
                const int n = n0 + grow[a] + 4 * grp;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = round_through<T>(acc[a][b][i]);
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
}

// ---- gate/up + SiLU.up with 112-row tiles: every CU of the chip gets a workgroup ------------------------------------------------------------------------
// The stacked gate / up matrix of the Llama MLPs has 2 x 14336 (8B) or 2 x 28672 (70B) rows: 128-row tiles give 224 / 448 workgroups -- 224 leave 32 of the
// 256 CUs idle, and on this kernel's memory-path bound a workgroup's time is its own bytes, so the launch takes as long with 224 as it would with 256.
// 112-row tiles (56 gate + 56 up rows) give 256 workgroups of 7/8 the weight bytes each (model: 72.8 -> 67 us at 256 batch rows).  56 is not a multiple of
// the 16-row MFMA group, so the tile INTERLEAVES: group g (of 7) holds gate rows 8g..8g+7 in its first eight rows and their up partners in its last eight;
// a DMA piece (8 rows x 128 B) is still eight consecutive rows of one of the two blocks.  In the 16 x 16 output a lane holds rows 4.(lane >> 4) + i: gate
// for lanes 0..31, up for lanes 32..63 of the SAME batch column -- one cross-half exchange per group pairs them (the low half finishes batch tile 0 of the
// wavefront, the high half batch tile 1: no idle lanes).  K is never split here (one round of workgroups): every output is the same chain of MFMA
// accumulations as in the 128-row kernel, so the result is bit-identical to the plain projection + atoma_silu_mul whatever tile either used.
template <typename T>
__global__ void __launch_bounds__(512, 1) linear_wide_gu112_kernel(const WideParams tp) {
    const LinearParams &p = tp.p;
    constexpr int NW = 112, BR = 256, G = 7, NSLOT = 3;
    constexpr int WT = NW * 128, XT = BR * 128, SLOT = WT + XT;
    static_assert(NSLOT * SLOT <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = lane >> 4, col = lane & 15;
    const int out_n = p.n / 2, tile = blockIdx.x, n0 = tile * (NW / 2);
    const int chunks = p.k >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    // W: 14 pieces per chunk (piece q = tile rows 8q..8q+7: even q = gate rows n0 + 4q.., odd q = up rows out_n + n0 + 4(q - 1)..): wavefronts 0..5 move
    // two each, 6 and 7 one; x: 32 pieces, four per wavefront.  The lane fills (row, slot) = (8q + (lane >> 3), lane & 7) from source piece (lane & 7) ^ (row & 7).
    const int nwp = wave < 6 ? 2 : 1, q0 = wave < 6 ? 2 * wave : 12 + (wave - 6);
    uint32_t woff[2], wdst[2], xoff[4], xdst[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = min(q0 + i, 13), row = 8 * q + (lane >> 3);
        const int64_t src_row = (q & 1 ? (int64_t)out_n : 0) + 8 * (q >> 1) + (lane >> 3);
        woff[i] = (uint32_t)(src_row * p.w_row_stride * 2 + ((lane & 7) ^ (row & 7)) * 16);
        wdst[i] = 8 * q * 128;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 4 * wave + i, row = 8 * q + (lane >> 3);
        xoff[i] = (uint32_t)((int64_t)min(row, p.batch - 1) * p.x_row_stride * 2 + ((lane & 7) ^ (row & 7)) * 16);
        xdst[i] = WT + 8 * q * 128;
    }
    const uint64_t wb = wide_uniform64((uint64_t)(p.w + (int64_t)n0 * p.w_row_stride));
    const uint64_t xb = wide_uniform64((uint64_t)p.x);
    auto issue = [&](int chunk, int slot) {
        const uint32_t sl = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) wide_dma(xb + (uint64_t)chunk * 128, xoff[i], sl + xdst[i]);
        wide_dma_nt(wb + (uint64_t)chunk * 128, woff[0], sl + wdst[0]);
        if (nwp == 2) wide_dma_nt(wb + (uint64_t)chunk * 128, woff[1], sl + wdst[1]);
    };
    lf32x4 acc[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc[g][0] = lf32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = lf32x4{0.f, 0.f, 0.f, 0.f}; }
    const bool live = 32 * wave < p.batch;
    auto compute = [&](int slot) {
        const char *base = smem + slot * SLOT;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = ((4 * s + grp) ^ (col & 7)) * 16;
            lu32x4 bf[2], af[G];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const lu32x4 *>(base + WT + (16 * (2 * wave + b) + col) * 128 + sw);
#pragma unroll
            for (int g = 0; g < G; ++g) af[g] = *reinterpret_cast<const lu32x4 *>(base + (16 * g + col) * 128 + sw);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][b] = lin_mfma<T>(af[g], bf[b], acc[g][b]);
        }
    };
    for (int s = 0; s < NSLOT - 1; ++s)
        if (s < chunks) issue(s, s);
    int slot = 0;
    for (int c = 0; c < chunks; ++c) {
        // this wavefront's pieces of chunk c have landed (5 or 6 per chunk: the count is the wavefront's own), then everybody's
        if (c + NSLOT - 2 < chunks) { if (nwp == 2) wide_vm_wait<6 * (NSLOT - 2)>(); else wide_vm_wait<5 * (NSLOT - 2)>(); } else wide_vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        if (c + NSLOT - 1 < chunks) issue(c + NSLOT - 1, pslot);
        if (live) compute(slot);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    // lanes 0..31 hold gate rows 8g + 4.(grp & 1) + i, lanes 32..63 the matching up rows, of batch columns 16.(2 wave + b) + col.  The halves swap one
    // accumulator per group: the low half gets the up values of batch tile 0, the high half the gate values of batch tile 1 -- each finishes one tile.
    const bool hi = lane >= 32;
    const int brow = 16 * (2 * wave + (hi ? 1 : 0)) + col;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const lf32x4 give = hi ? acc[g][0] : acc[g][1];
        lf32x4 got;
#pragma unroll
        for (int i = 0; i < 4; ++i) got[i] = __shfl_xor(give[i], 32, 64);
        if (brow >= p.batch) continue;
        const lf32x4 gate = hi ? got : acc[g][0], up = hi ? acc[g][1] : got;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                  // rounding points as in linear_reduce_kernel / the 128-row kernel
            const float x = round_through<T>(gate[i]);
            v[i] = round_through<T>(x / (1.f + __expf(-x))) * round_through<T>(up[i]);
        }
        uint2 o;
        o.x = pack2<T>(v[0], v[1]);
        o.y = pack2<T>(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n0 + 8 * g + 4 * (grp & 1)) = o;
    }
}

static int wide_env_or(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
// knobs: environment at load time, atoma_set_option("linear_wide*") at run time (A/B runs inside one process)
static std::atomic<int> linear_wide_on{wide_env_or("ATOMA_LINEAR_WIDE", 1)};           // 0: linear_big_kernel serves 65..256 rows
static std::atomic<int> linear_wide_nw{wide_env_or("ATOMA_LINEAR_WIDE_NW", 0)};        // weight rows per workgroup: 0 = by shape
static std::atomic<int> linear_wide_splits{wide_env_or("ATOMA_LINEAR_WIDE_SPLITS", 0)};   // K splits: 0 = by shape
static std::atomic<int> linear_wide_var{wide_env_or("ATOMA_LINEAR_WIDE_VAR", 0)};      // probe bits, see WideParams::var
static std::atomic<int> linear_wide_bb{wide_env_or("ATOMA_LINEAR_WIDE_BB", 0)};        // batch blocks: 0 = by shape, 1 = one tile over the batch, 2 = blocks of 128 rows
static std::atomic<int> linear_wide_gu112{wide_env_or("ATOMA_LINEAR_WIDE_GU112", 1)};  // 0: gate/up + SiLU.up always on the 128-row tiles
static std::atomic<int> linear_wide_xcd{wide_env_or("ATOMA_LINEAR_WIDE_XCD", 0)};      // 1: a K range per XCD (splits 2 / 4 / 8) -- measured slower (o 25.2 vs 23.7 us, down 53.4 vs 52.0): off
bool set_linear_wide_option(const std::string &name, int value) {
    if (name == "linear_wide") linear_wide_on = value;
    else if (name == "linear_wide_nw") linear_wide_nw = value;
    else if (name == "linear_wide_splits") linear_wide_splits = value;
    else if (name == "linear_wide_xcd") linear_wide_xcd = value;
    else if (name == "linear_wide_var") linear_wide_var = value;
    else if (name == "linear_wide_bb") linear_wide_bb = value;
    else if (name == "linear_wide_gu112") linear_wide_gu112 = value;
    else return false;
    return true;
}

// Rows per workgroup, K split and batch blocks from the SHAPE of the product alone (n, k, batch rows), priced with the queue model in the header: a
// workgroup's time = (weight KB x 2.0 us [x 0.6 for the passes a sibling batch block already pulled into L2] + x KB x 0.6 us) / 45 KB, plus ~1.5 us + the
// slabs the last arriver reads back at ~50 KB/us for an in-launch merge, plus the one-workgroup epilogue tail, times the rounds of workgroups over the
// CUs.  8B at 256 rows (measured, profiles/r06_linear256_ab.jsonl): gate/up (128 rows, 1 split, one 256-row tile: 224 workgroups) 72-75 us; down (128 rows,
// 4 splits, two 128-row batch blocks: 256 workgroups) 49 (one 256-row tile, 64 rows, 4 splits: 52-54); o (64, 2, two blocks: 256) 21.9 (23.6-24.5); q/k/v
// (64, 1, two blocks: 192 workgroups, no merge) 32.7 with the RoPE epilogue (34.4).  The plan must NOT depend on the epilogue: the fused entries are
// bit-identical to projection + separate op only because both split K alike.
constexpr int WIDE_MAX_MERGE = 8;
struct WidePlan { int nw = 0, splits = 1, br = 256, bblocks = 1; };
// candidates: 128 / 64 rows x 1..8 K splits x (the whole batch in one tile | blocks of 128 batch rows).  Batch blocks: W is streamed once from HBM
// and (bblocks - 1) more times from L2 by the sibling blocks; x and the slabs per workgroup shrink with the tile.
static WidePlan wide_plan(int64_t n, int64_t k, int batch, int cus) {
    const int64_t chunks = k / 64;
    const int br_one = batch <= 128 ? 128 : (batch <= 192 ? 192 : 256);
    WidePlan best;
    double best_t = 1e30;
    for (int bb : {1, 2}) {
        if (bb == 2 && batch <= 128) continue;
        const int br = bb == 1 ? br_one : 128;
        for (int s : {1, 2, 3, 4, 5, 6, 8})
            for (int nw : {128, 64}) {
                if (n % nw || chunks / s < 4) continue;
                const int64_t wgs = n / nw * s * bb;
                if (s > 1 && n / nw * bb > 8192) continue;                  // one arrival counter per (tile, batch block)
                const double kb = (double)cdiv(chunks, s) * 128.0 / 1024.0;
                const double merge = s == 1 ? 0.0 : 1.5 + (double)(s - 1) * (nw * br * 4 / 1024.0) / 50.0;
                const double tail = nw * br / 8192.0;      // the epilogue of a tile runs in ONE workgroup (measured: the RoPE epilogue of a 128 x 256 tile ~8 us, of a 64 x 256 tile ~3)
                const double w_cost = nw * kb * (2.0 / bb + 0.6 * (bb - 1) / bb);
                const double t = (double)cdiv(wgs, cus) * ((w_cost + br * kb * 0.6) / 45.0 + merge + tail);
                if (t < best_t) { best_t = t; best = WidePlan{nw, s, br, bb}; }
            }
    }
    return best;
}

// nw = 0: not served
static WidePlan wide_route(const LinearParams &p) {
    WidePlan none;
    if (!linear_wide_on || p.k % 64 || p.n % 64 || p.batch > 256 || p.batch <= 64) return none;
    WidePlan pl = wide_plan(p.n, p.k, p.batch, device_num_cus());
    if (linear_wide_nw > 0 && p.n % linear_wide_nw == 0) pl.nw = linear_wide_nw;
    if (linear_wide_splits > 0) pl.splits = linear_wide_splits;
    if (linear_wide_bb == 1) { pl.bblocks = 1; pl.br = p.batch <= 128 ? 128 : (p.batch <= 192 ? 192 : 256); }
    if (linear_wide_bb == 2 && p.batch > 128) { pl.bblocks = 2; pl.br = 128; }
    if ((pl.nw != 64 && pl.nw != 128) || p.n % pl.nw) return none;
    const int64_t chunks = p.k / 64;
    int splits = (int)std::min<int64_t>(std::min<int64_t>(pl.splits, WIDE_MAX_MERGE), std::max<int64_t>(chunks / 4, 1));
    pl.splits = (int)cdiv(chunks, cdiv(chunks, splits));
    if (pl.splits > 1 && p.n / pl.nw * pl.bblocks > 8192) return none;
    return pl;
}

// the RoPE epilogue can ride on this product: served, and a tile holds both halves of a head
bool linear_wide_can_rope(const LinearParams &p, int head_dim) {
    const int nw = wide_route(p).nw;
    return nw != 0 && head_dim % 32 == 0 && nw / 2 <= head_dim / 2 && (head_dim / 2) % (nw / 2) == 0;
}

template <typename T, int NW, int BR, int MODE> static constexpr size_t wide_lds() {
    constexpr int SLOT = NW * 128 + BR * 128;
    return (size_t)(4 * SLOT + 1024 <= 160 * 1024 ? 4 : 3) * SLOT;
}

template <typename T> static int launch_linear_wide_t(LinearParams &p, hipStream_t stream, const WideRope *rope) {
    const WidePlan pl = wide_route(p);
    const int nw = pl.nw;
    int splits = pl.splits;
    if (nw == 0) return 1;
    const int64_t chunks = p.k / 64;
    WideParams tp{};
    tp.chunks_per_split = (int)cdiv(chunks, splits);
    splits = (int)cdiv(chunks, tp.chunks_per_split);
    tp.splits = splits;
    p.splits = splits;
    p.partial = nullptr;
    const int br = pl.br;
    const int64_t tiles = p.n / nw;
    if (splits > 1) {
        tp.slabs = static_cast<float *>(workspace(stream, (size_t)tiles * pl.bblocks * splits * nw * br * sizeof(float)));
        tp.counters = sync_counters(stream);
        if (!tp.slabs || !tp.counters) return -1;
    }
    tp.xcd_map = linear_wide_xcd && (splits == 2 || splits == 4 || splits == 8) ? 1 : 0;
    tp.p = p;
    tp.var = linear_wide_var;
    if (rope) tp.rope = *rope;
    const int mode = rope ? WIDE_ROPE : (p.epilogue == 2 ? WIDE_GATE_UP : WIDE_PLAIN);
    // gate/up + SiLU.up, unsplit, one 256-row batch tile: 112-row tiles when they fill more CUs in the same number of rounds (8B / 70B: 256 workgroups
    // instead of 224) -- bit-identical to any other unsplit tiling (linear_wide_gu112_kernel's header)
    if (mode == WIDE_GATE_UP && splits == 1 && pl.bblocks == 1 && br == 256 && linear_wide_gu112 && (p.n / 2) % 56 == 0) {
        const int64_t t112 = p.n / 112, t128 = p.n / 128, cus = device_num_cus();
        if (cdiv(t112, cus) <= cdiv(t128, cus)) {
            tp.p = p;
            tp.var = linear_wide_var;
            tp.splits = 1;
            int dev112 = 0;
            (void)hipGetDevice(&dev112);
            dev112 = dev112 < 0 || dev112 >= 64 ? 0 : dev112;
            constexpr size_t lds112 = 3 * (112 + 256) * 128;
            static std::atomic<bool> once112[64];
            if (!once112[dev112].load()) { if (!check_hip(hipFuncSetAttribute((const void *)linear_wide_gu112_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds112), "linear_wide_gu112 LDS")) return -1; once112[dev112] = true; }
            hipLaunchKernelGGL((linear_wide_gu112_kernel<T>), dim3((unsigned)t112), dim3(512), lds112, stream, tp);
            return ATOMA_CHECK_LAUNCH("linear_wide_gu112_kernel") ? 0 : -1;
        }
    }
    // xcd_map: groups of 8 workgroups cover 8 / S tiles: round the tile count up to a whole group (surplus workgroups leave at once)
    const int64_t wgs = tp.xcd_map ? cdiv(tiles, 8 / splits) * 8 : tiles * splits;
    const dim3 grid((unsigned)wgs, (unsigned)pl.bblocks), block(512);
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
#define ATOMA_WIDE_M(NW_, BR_, MODE_) do { \
        const size_t lds = wide_lds<T, NW_, BR_, MODE_>(); \
        static std::atomic<bool> once[64]; \
        if (!once[dev].load()) { if (!check_hip(hipFuncSetAttribute((const void *)linear_wide_kernel<T, NW_, BR_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "linear_wide LDS")) return -1; once[dev] = true; } \
        hipLaunchKernelGGL((linear_wide_kernel<T, NW_, BR_, MODE_>), grid, block, lds, stream, tp); } while (0)
#define ATOMA_WIDE_B(NW_, BR_) do { if (mode == WIDE_ROPE) ATOMA_WIDE_M(NW_, BR_, WIDE_ROPE); else if (mode == WIDE_GATE_UP) ATOMA_WIDE_M(NW_, BR_, WIDE_GATE_UP); else ATOMA_WIDE_M(NW_, BR_, WIDE_PLAIN); } while (0)
#define ATOMA_WIDE(NW_) do { if (br == 256) ATOMA_WIDE_B(NW_, 256); else if (br == 192) ATOMA_WIDE_B(NW_, 192); else ATOMA_WIDE_B(NW_, 128); } while (0)
    if (nw == 128) ATOMA_WIDE(128); else ATOMA_WIDE(64);
#undef ATOMA_WIDE
#undef ATOMA_WIDE_B
#undef ATOMA_WIDE_M
    return ATOMA_CHECK_LAUNCH("linear_wide_kernel") ? 0 : -1;
}

template <typename T, int NW, int BR, int MODE> static bool wide_prepare_one() {
    return check_hip(hipFuncSetAttribute((const void *)linear_wide_kernel<T, NW, BR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds<T, NW, BR, MODE>()), "linear_wide LDS");
}
template <typename T, int MODE> static bool wide_prepare_m() {
    return wide_prepare_one<T, 64, 128, MODE>() && wide_prepare_one<T, 128, 128, MODE>() && wide_prepare_one<T, 64, 192, MODE>() && wide_prepare_one<T, 128, 192, MODE>() &&
           wide_prepare_one<T, 64, 256, MODE>() && wide_prepare_one<T, 128, 256, MODE>();
}
template <typename T> static bool wide_prepare_t() { return wide_prepare_m<T, WIDE_PLAIN>() && wide_prepare_m<T, WIDE_GATE_UP>() && wide_prepare_m<T, WIDE_ROPE>(); }
// atoma_warmup: raise the LDS limit of every variant on the current device (a hipGraph capture can then be the first call)
bool linear_wide_prepare() {
    constexpr int lds112 = 3 * (112 + 256) * 128;
    return wide_prepare_t<bf16_t>() && wide_prepare_t<f16_t>() &&
           check_hip(hipFuncSetAttribute((const void *)linear_wide_gu112_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds112), "linear_wide_gu112 LDS") &&
           check_hip(hipFuncSetAttribute((const void *)linear_wide_gu112_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds112), "linear_wide_gu112 LDS");
}

// 0 = launched, 1 = shape not served (the caller falls back to linear_big_kernel), -1 = error
int launch_linear_wide(LinearParams &p, int dtype, hipStream_t stream) {
    return dtype == ATOMA_BF16 ? launch_linear_wide_t<bf16_t>(p, stream, nullptr) : launch_linear_wide_t<f16_t>(p, stream, nullptr);
}
int launch_linear_wide_rope(LinearParams &p, int dtype, hipStream_t stream, const uint16_t *cos_t, const uint16_t *sin_t, const int64_t *positions,
                            const int64_t *slot_mapping, uint16_t *k_cache, uint16_t *v_cache, int64_t block_stride, int64_t table_rows, int heads_q,
                            int heads_kv, int head_dim, int page_size, int per_op) {
    WideRope r{cos_t, sin_t, positions, slot_mapping, k_cache, v_cache, block_stride, table_rows, heads_q, heads_kv, head_dim, page_size, per_op};
    return dtype == ATOMA_BF16 ? launch_linear_wide_t<bf16_t>(p, stream, &r) : launch_linear_wide_t<f16_t>(p, stream, &r);
}

}  // namespace atoma
