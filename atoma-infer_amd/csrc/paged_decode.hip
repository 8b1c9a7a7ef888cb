// Paged-KV decode attention for gfx950 ("paged_attention v1/v2"): seqlen_q == 1.
//
// Replaces the reference's split-KV FlashAttention kernel when it is used for decode
//   /root/reference/csrc/kernels/flash_fwd_kernel.h:504-1092 (compute_attn_1rowblock_splitkv, paged K/V)
//   /root/reference/csrc/kernels/flash_fwd_kernel.h:1131-1313 (combine_attn_seqk_parallel)
// reached through csrc::flash_attn_kv_cache_full (csrc/src/lib.rs:1521-1855).
//
// The reference runs a 64-row MMA tile for ONE query row and gives every q head of a GQA
// group its own CTA (K/V re-read h/h_k times).  Decode is HBM-bound (h/h_k FLOP per byte),
// so this kernel is organised around the byte stream instead:
//
//  * one 64-lane wavefront per (sequence, kv head, KV split); all q heads of the GQA group
//    are processed together, so every K/V byte is fetched exactly once;
//  * a 16-token tile of one kv head is 16 rows of D*2 bytes; a row is read by D/8 adjacent
//    lanes with one 16-byte load each (full 128-byte lines, 1 KiB per wave instruction);
//    P tiles (8 KiB each for D=128) are kept in flight per wave in registers;
//  * block-table entries travel through the scalar cache (wave-uniform s_load, 64-byte lines =
//    16 page ids per memory access), fetched 2P tiles ahead of use, so the table is never on
//    the critical path and is never read beyond the pages the sequence owns (the reference
//    does, SURVEY B/Q6);
//  * q.k: v_dot2c_f32_{bf16,f16} on the lane's 8 elements, then an all-reduce over the
//    D/8 lanes of the row with DPP adds (no LDS);
//  * each group of D/8 lanes keeps its own online-softmax state (running max, sum, O) for
//    the rows it owns, so there is no cross-lane traffic in the loop; the groups are
//    merged once at the end with the usual LSE rescale;
//  * P is rounded to the storage dtype before P.V (as the reference: softmax.h + the
//    bf16 MMA) and P.V is again v_dot2c on token pairs;
//  * KV splits write fp32 partial O / LSE to a workspace and a small combine kernel merges
//    them (same math as the reference's combine kernel).
// (That is the first kernel, paged_decode_kernel -- today it serves MHA, d = 64 and split-KV launches of small groups.  GQA groups at
// d = 128 run on paged_decode_mqk_kernel further down: both products on the matrix cores; and WHICH wavefronts share a CU -- the
// launch order, decode_plan_launch / decode_run_items -- turned out to be worth more than either kernel's inner loop: DESIGN.md 4.1.)
//
// Algorithmic HBM bytes per call: 2*B*S*h_k*D*2 (K,V once) + 2*B*h*D*2 + 4*B*ceil(S/page) + 4*B.
#include "paged_decode.h"

namespace atoma {

template <typename T, int D, int G, int P, bool NT, bool SINK = false>
__device__ __forceinline__ void paged_decode_item(const DecodeParams &p, const DecodeWork &wk) {
    constexpr int LPR = D / 8;     // lanes per row
    constexpr int RPI = 64 / LPR;  // rows per load instruction
    constexpr int IPP = 16 / RPI;  // load instructions per 16-token tile
    static_assert(IPP >= 2 && IPP % 2 == 0, "token pairs for the P.V dot2");
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, dc = lane % LPR;

    const int b = wk.b, hk = wk.hk, gc = wk.gc, L = wk.L, t0 = wk.t0, t1 = wk.t1;
    const int64_t kv_row0 = wk.kv_row0;
    const bool partial = wk.partial;
    const int hq0 = hk * p.g + gc * G;
    const int nq = min(G, p.g - gc * G);

    const float sl2 = p.scale_log2;
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        m[gq] = -INFINITY;
        l[gq] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[gq][e] = 0.f;
    }

    if (t0 < t1) {
        // ---- q: 8 elements per lane per head, replicated over the RPI row groups ----
        uint4 qv[G];
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            qv[gq] = make_uint4(0, 0, 0, 0);
            if (gq < nq)
                qv[gq] = *reinterpret_cast<const uint4 *>(p.q + (int64_t)b * p.q_batch_stride +
                                                          (int64_t)(hq0 + gq) * p.q_head_stride + dc * 8);
        }
        float alibi[G];
        const bool has_alibi = p.alibi_slopes != nullptr;
#pragma unroll
        for (int gq = 0; gq < G; ++gq)
            alibi[gq] = (has_alibi && gq < nq) ? p.alibi_slopes[b * p.alibi_batch_stride + hq0 + gq] * 1.4426950408889634f : 0.f;

        // ---- loader ----
        // Page ids come through the scalar cache (s_load_dword on a wave-uniform address), one
        // per tile, fetched 2P tiles ahead of use: they ride lgkmcnt, so the K/V stream's vmcnt
        // accounting stays exact, and the 64-byte scalar-cache line makes the table read coalesced
        // (16 ids per HBM/L2 access).  The index is clamped to the pages the sequence owns.
        const bool paged = p.block_table != nullptr;
        const uint32_t tpp = paged ? (uint32_t)(p.page_size >> 4) : 1u;  // tiles per page
        const uint32_t tpp_magic = tpp > 1 ? (uint32_t)((0x100000000ull + tpp - 1) / tpp) : 0u;
        const int last_pg = paged ? (L + p.page_size - 1) / p.page_size - 1 : 0;
        const int *bt_row = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
        const char *kbase = reinterpret_cast<const char *>(p.k + (int64_t)hk * p.k_head_stride);
        const char *vbase = reinterpret_cast<const char *>(p.v + (int64_t)hk * p.v_head_stride);
        if (!paged) {
            const bool cum = p.cu_seqlens_k && p.is_seqlens_k_cumulative;
            kbase += (cum ? kv_row0 * p.k_row_stride : (int64_t)b * p.k_batch_stride) * 2;
            vbase += (cum ? kv_row0 * p.v_row_stride : (int64_t)b * p.v_batch_stride) * 2;
        }
        const int64_t k_row_bytes = p.k_row_stride * 2, v_row_bytes = p.v_row_stride * 2;
        const int64_t k_page_bytes = p.k_batch_stride * 2, v_page_bytes = p.v_batch_stride * 2;
        // per-lane byte offset inside a tile (row `sub` of each RPI-row slab, 16-byte chunk `dc`)
        const uint32_t k_lane_off = (uint32_t)(sub * k_row_bytes + dc * 16);
        const uint32_t v_lane_off = (uint32_t)(sub * v_row_bytes + dc * 16);

        auto page_of = [&](int tile, uint32_t &tip) -> int {  // wave-uniform
            if (tpp == 1) { tip = 0; return tile; }
            const uint32_t pg = __umulhi((uint32_t)tile, tpp_magic);
            tip = (uint32_t)tile - pg * tpp;
            return (int)pg;
        };
        auto fetch_pid = [&](int tile) -> int {  // scalar load, always in bounds
            if (!paged) return 0;
            uint32_t tip;
            const int pg = min(page_of(tile, tip), last_pg);
            return load_ro(bt_row + pg);
        };
        auto tile_bases = [&](int tile, int pid, const char *&kt, const char *&vt) {
            if (paged) {
                uint32_t tip;
                (void)page_of(tile, tip);
                kt = kbase + (int64_t)pid * k_page_bytes + (int64_t)(tip << 4) * k_row_bytes;
                vt = vbase + (int64_t)pid * v_page_bytes + (int64_t)(tip << 4) * v_row_bytes;
            } else {
                kt = kbase + (int64_t)(tile << 4) * k_row_bytes;
                vt = vbase + (int64_t)(tile << 4) * v_row_bytes;
            }
        };
        // Buffer loads: a 128-bit descriptor in SGPRs whose base is the (wave-uniform) tile address,
        // a 32-bit per-lane byte offset that never changes, and the row-slab offset in soffset --
        // no per-load 64-bit VGPR address math, and out-of-range rows would read zeros, not fault.
        // NT: K/V bytes are consumed exactly once per call -> non-temporal (streaming) cache policy.
        constexpr int AUX = NT ? 2 : 0;
        auto tile_rsrc = [&](const char *base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7fffffff, 0x00020000);
        };
        // full tile: fixed lane offsets
        auto issue_fast = [&](u32x4 (&kb)[IPP], u32x4 (&vb)[IPP], int tile, int pid) {
            const char *kt, *vt;
            tile_bases(tile, pid, kt, vt);
            const __amdgpu_buffer_rsrc_t kr = tile_rsrc(kt), vr = tile_rsrc(vt);
#pragma unroll
            for (int r = 0; r < IPP; ++r)
                kb[r] = __builtin_amdgcn_raw_buffer_load_b128(kr, k_lane_off, (int)(r * RPI * k_row_bytes), AUX);
#pragma unroll
            for (int r = 0; r < IPP; ++r)
                vb[r] = __builtin_amdgcn_raw_buffer_load_b128(vr, v_lane_off, (int)(r * RPI * v_row_bytes), AUX);
        };
        // any tile: rows clamped to the last row of the sequence: a contiguous (non-paged) cache is never read past its end, and the
        // never-written slots of a paged cache's last page (NaN patterns included) never meet p = 0 in P.V (the reference zero-fills
        // out-of-range V rows, flash_fwd_kernel.h:903)
        auto issue_tail = [&](u32x4 (&kb)[IPP], u32x4 (&vb)[IPP], int tile, int pid) {
            const char *kt, *vt;
            tile_bases(tile, pid, kt, vt);
            const __amdgpu_buffer_rsrc_t kr = tile_rsrc(kt), vr = tile_rsrc(vt);
            const int lastrow = min(15, L - 1 - (tile << 4));
#pragma unroll
            for (int r = 0; r < IPP; ++r) {
                const int row = min(r * RPI + sub, lastrow);
                kb[r] = __builtin_amdgcn_raw_buffer_load_b128(kr, (uint32_t)(row * k_row_bytes + dc * 16), 0, AUX);
            }
#pragma unroll
            for (int r = 0; r < IPP; ++r) {
                const int row = min(r * RPI + sub, lastrow);
                vb[r] = __builtin_amdgcn_raw_buffer_load_b128(vr, (uint32_t)(row * v_row_bytes + dc * 16), 0, AUX);
            }
        };

        auto compute = [&](const u32x4 (&kb)[IPP], const u32x4 (&vb)[IPP], int tile) {
            float s[IPP][G];
#pragma unroll
            for (int r = 0; r < IPP; ++r)
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    float a = dot2<T>(kb[r].x, qv[gq].x, 0.f);
                    a = dot2<T>(kb[r].y, qv[gq].y, a);
                    a = dot2<T>(kb[r].z, qv[gq].z, a);
                    a = dot2<T>(kb[r].w, qv[gq].w, a);
                    s[r][gq] = row_allreduce<LPR>(a) * sl2;  // log2 domain
                }
            const int tok0 = (tile << 4) + sub;
            if (has_alibi) {  // wave-uniform
#pragma unroll
                for (int r = 0; r < IPP; ++r)
#pragma unroll
                    for (int gq = 0; gq < G; ++gq)
                        s[r][gq] -= alibi[gq] * (float)(L - 1 - (tok0 + r * RPI));  // mask.h:183, row 0 of 1
            }
            if ((tile << 4) + 16 > L) {  // wave-uniform: ragged last tile
#pragma unroll
                for (int r = 0; r < IPP; ++r)
                    if (tok0 + r * RPI >= L) {
#pragma unroll
                        for (int gq = 0; gq < G; ++gq) s[r][gq] = -INFINITY;
                    }
            }
            float mnew[G];
            bool changed = false;
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
                float mx = s[0][gq];
#pragma unroll
                for (int r = 1; r < IPP; ++r) mx = fmaxf(mx, s[r][gq]);
                mnew[gq] = fmaxf(m[gq], mx);
                changed |= mnew[gq] > m[gq];
            }
            if (__any(changed)) {  // rescale only when some running max moved
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    const float ms = mnew[gq] == -INFINITY ? 0.f : mnew[gq];
                    const float alpha = __builtin_amdgcn_exp2f(m[gq] - ms);
                    l[gq] *= alpha;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[gq][e] *= alpha;
                    m[gq] = mnew[gq];
                }
            }
            uint32_t pp[IPP / 2][G];
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
                const float ms = m[gq] == -INFINITY ? 0.f : m[gq];
                float pr[IPP];
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < IPP; ++r) {
                    pr[r] = __builtin_amdgcn_exp2f(s[r][gq] - ms);
                    sum += pr[r];
                }
                l[gq] += sum;
#pragma unroll
                for (int c = 0; c < IPP / 2; ++c) pp[c][gq] = pack_pair<T>(pr[2 * c], pr[2 * c + 1]);
            }
            // P.V on token pairs: (V[r0][e], V[r1][e]) . (p[r0], p[r1])
#pragma unroll
            for (int c = 0; c < IPP / 2; ++c) {
                const uint32_t a[4] = {vb[2 * c].x, vb[2 * c].y, vb[2 * c].z, vb[2 * c].w};
                const uint32_t bb[4] = {vb[2 * c + 1].x, vb[2 * c + 1].y, vb[2 * c + 1].z, vb[2 * c + 1].w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t lo = __builtin_amdgcn_perm(bb[w], a[w], 0x05040100u);  // (a.lo, b.lo)
                    const uint32_t hi = __builtin_amdgcn_perm(bb[w], a[w], 0x07060302u);  // (a.hi, b.hi)
#pragma unroll
                    for (int gq = 0; gq < G; ++gq) {
                        o[gq][2 * w] = dot2<T>(lo, pp[c][gq], o[gq][2 * w]);
                        o[gq][2 * w + 1] = dot2<T>(hi, pp[c][gq], o[gq][2 * w + 1]);
                    }
                }
            }
        };

        // ---- software pipeline: P tiles in flight per wave ----
        // The steady-state loop has no conditional loads, so the compiler's vmcnt waits are exact
        // (tile t is consumed while tiles t+1 .. t+P-1 and the just-issued t+P stay in flight);
        // the last < 2P tiles go through the conditional tail.
        u32x4 kb[P][IPP], vb[P][IPP];
        int pid[P];
        int t = t0;
        const int steady_end = min(t1, L >> 4);  // full tiles load without clamping; a ragged last tile goes through the clamped tail (paged too: never-written slots)
        if (t0 + 2 * P <= steady_end) {
            // long sequence: unconditional prologue, so the loop header sees ONE load history
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                issue_fast(kb[s], vb[s], t0 + s, pid[s]);
                pid[s] = fetch_pid(t0 + s + P);
            }
            for (; t + 2 * P <= steady_end; t += P) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    compute(kb[s], vb[s], t + s);
                    issue_fast(kb[s], vb[s], t + s + P, pid[s]);
                    pid[s] = fetch_pid(t + s + 2 * P);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s)
                if (t0 + s < t1) {
                    issue_tail(kb[s], vb[s], t0 + s, pid[s]);
                    pid[s] = fetch_pid(t0 + s + P);
                }
        }
        for (; t < t1; t += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                if (t + s < t1) {
                    compute(kb[s], vb[s], t + s);
                    if (t + s + P < t1) {
                        issue_tail(kb[s], vb[s], t + s + P, pid[s]);
                        pid[s] = fetch_pid(t + s + 2 * P);
                    }
                }
            }
        }
    }

    // ---- merge the RPI row groups (each has its own m, l, o) ----
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        float mt = m[gq];
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) mt = fmaxf(mt, __shfl_xor(mt, off, 64));
        const float ms = mt == -INFINITY ? 0.f : mt;
        const float w = __builtin_amdgcn_exp2f(m[gq] - ms);
        float lt = l[gq] * w;
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) lt += __shfl_xor(lt, off, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = o[gq][e] * w;
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
            o[gq][e] = x;
        }
        m[gq] = mt;
        l[gq] = lt;
    }

    if (sub != 0) return;
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        if (gq >= nq) continue;
        const int hq = hq0 + gq;
        const bool empty = !(l[gq] > 0.f);  // no visible key (L == 0 or empty split)
        const float inv = empty ? 0.f : 1.f / l[gq];
        // natural-log LSE: m*scale + ln(l); m is already scaled by scale*log2e
        const float lse = empty ? INFINITY : (m[gq] + __builtin_amdgcn_logf(l[gq])) * 0.6931471805599453f;
        if constexpr (SINK) {   // workgroup-merged split mode: the piece's result stays on the CU
            float4 *dst = reinterpret_cast<float4 *>(wk.sink_o + gq * D + dc * 8);
            dst[0] = make_float4(o[gq][0] * inv, o[gq][1] * inv, o[gq][2] * inv, o[gq][3] * inv);
            dst[1] = make_float4(o[gq][4] * inv, o[gq][5] * inv, o[gq][6] * inv, o[gq][7] * inv);
            if (dc == 0) wk.sink_lse[gq] = empty ? -INFINITY : lse;
        } else if (!partial) {
            uint4 w4;
            w4.x = pack2<T>(o[gq][0] * inv, o[gq][1] * inv);
            w4.y = pack2<T>(o[gq][2] * inv, o[gq][3] * inv);
            w4.z = pack2<T>(o[gq][4] * inv, o[gq][5] * inv);
            w4.w = pack2<T>(o[gq][6] * inv, o[gq][7] * inv);
            *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + dc * 8) = w4;
            if (p.lse && dc == 0) p.lse[(int64_t)b * p.h + hq] = lse;
        } else {
            const int64_t row = wk.prow + gq;
            float *dst = p.o_accum + row * D + dc * 8;
            partial_store(dst, o[gq][0] * inv, o[gq][1] * inv, o[gq][2] * inv, o[gq][3] * inv);
            partial_store(dst + 4, o[gq][4] * inv, o[gq][5] * inv, o[gq][6] * inv, o[gq][7] * inv);
            if (dc == 0) partial_store(p.lse_accum + row, empty ? -INFINITY : lse);  // flash_fwd_kernel.h:543-582
        }
    }
}

template <typename T, int D, int G, int P, int MINW, bool NT, bool STREAM>
__global__ void __launch_bounds__(64, MINW) paged_decode_kernel(const DecodeParams p) {
    decode_run_items<STREAM>(p, [](const DecodeParams &pp, const DecodeWork &wk) { paged_decode_item<T, D, G, P, NT>(pp, wk); }, DecodeLineMerge<T, D>{});
}

// ------------------------------------------------------------------------------------------
// d = 128 variant with q.K^T on the matrix cores.  With 8 q heads per kv head (Llama-70B) the dot2 kernel
// above does 8 flop per K/V byte on the VALU and is compute-bound (43 % of the HBM peak, one 512-register
// wavefront per SIMD).  The scores of a 16-token tile for up to 16 q heads are one small matrix product,
// S^T[token][head] = K[16 x 128] . Q^T[128 x 16]: four v_mfma_f32_16x16x32 per tile instead of
// 16.G v_dot2c + 16.G DPP reduction steps.  Layout (lane = 16.grp + col):
//   * K operand (A): lane reads 16 bytes of token `col`, d chunk 4s + grp, for k-step s = 0..3 -- the same
//     4 x 1 KiB wave loads per tile as before, rows and chunks assigned differently;
//   * Q^T operand (B): lane holds q[head col][d chunk 4s + grp] (zero for col >= heads), loaded once;
//   * result: lane holds S^T[token 4.grp + i][head col], i = 0..3 -- one head per lane, so the online
//     softmax state is two scalars per lane;
//   * P.V (round 3; until then v_dot2c over token pairs with one DPP row_newbcast per head): O[head][d] = P[head][token] . V[token][d]
//     as eight v_mfma_f32_16x16x16 per tile.  The A operand of lane (grp, col) is P[head col][tokens 4.grp ..+3] -- the four
//     probabilities the lane has just computed, packed; the B operand is V[tokens 4.grp ..+3][d = 8.col + n], which is what the
//     lane's four V loads (rows 4.grp + r, 16-byte d chunk `col`) hold, re-paired along the tokens with v_perm; MFMA n accumulates
//     O[head 4.grp + i][d = 8.col + n] -- 32 accumulator registers for ANY group size.  All 16 tokens of a tile enter one
//     accumulator, so the running max of a head is common to its 4 lane groups (v_permlane16_swap + v_permlane32_swap per tile);
//     the row sums stay per lane group and are added at the end; the rare rescale fetches the heads' factors with ds_bpermute.
//     Why: at one wavefront per SIMD (the TP rank's split-KV launches) the kernel was issue-bound -- 50 % of its cycles issuing, 11 %
//     waiting for memory, 306 VALU instructions per tile at 8 heads (tools/probes/headline_counters.sh).
// VALU work per tile: ~120 at any G (round 2: ~100 + 16.G; the dot2 kernel: ~70.G); 2 wavefronts per SIMD with 3 tiles in flight.
// ------------------------------------------------------------------------------------------
// PAIR64 (round 4): head_dim 64 through this kernel.  A kv head's slice of a token row is then 128 bytes -- half of what a wavefront fetches
// per row here -- and its neighbour's half arrives from another wavefront some time later: the dot2 kernel stays at 0.63-0.73 of HBM on the
// Llama-3.2-1B shape and loses on the kv-head-major line.  So ONE wavefront takes TWO adjacent kv heads: the launcher hands
// over a view with h_k / 2 kv heads of 128 "dims" (the two heads' 64 + 64, contiguous in the reference's cache layout) and 2 g q heads per
// group; the q heads of the first kv head carry zeros in the upper 64 dims of the Q^T operand, those of the second in the lower 64, so
// S = K . Q^T is each head's own 64-dim product; O = P . V comes out 128 wide, of which a head keeps the half of its own kv head
// (fp32 partial rows are 64 floats).  Half of the matrix-core work is wasted; the kernel waits for memory either way.
template <typename T, int G, int P, bool NT, bool SINK = false, bool PAIR64 = false>
__device__ __forceinline__ void paged_decode_mqk_item(const DecodeParams &p, const DecodeWork &wk) {
    constexpr int D = 128;
    static_assert(!(PAIR64 && SINK), "the workgroup-merged route does not take kv-head pairs");
    const int lane = threadIdx.x & 63, grp = lane >> 4, col = lane & 15;

    const int b = wk.b, hk = wk.hk, gc = wk.gc, L = wk.L, t0 = wk.t0, t1 = wk.t1;
    const int64_t kv_row0 = wk.kv_row0;
    const bool partial = wk.partial;
    const int hq0 = hk * p.g + gc * G;
    const int nq = min(G, p.g - gc * G);

    const float sl2 = p.scale_log2;
    float m = -INFINITY, l = 0.f;        // head `col`: running max over ALL tokens seen (common to its 4 lane groups), sum over this lane group's tokens
    f32x4_v o[8];                        // o[n][i] = O[head 4 grp + i][d = 8 col + n]  (P.V on the matrix cores: 32 registers for any group size)
#pragma unroll
    for (int n = 0; n < 8; ++n) o[n] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    const int head_lane4 = ((lane & 48) + 4 * grp) << 2;     // ds_bpermute address of lane (grp, 4 grp): the state of head 4 grp + i sits in lane (grp, 4 grp + i)

    if (t0 < t1) {
        u32x4 qb[4];                     // Q^T operand, k-step s: q[head col][d = 32s + 8.grp ..+7]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            qb[s4] = u32x4{0, 0, 0, 0};
            if constexpr (PAIR64) {      // chunk c = 4 s4 + grp of the 128-wide row: chunks 0..7 belong to the first kv head of the pair, 8..15 to the second
                const int c = 4 * s4 + grp;
                if (col < nq && (c >> 3) == (col >= (p.g >> 1) ? 1 : 0))
                    qb[s4] = *reinterpret_cast<const u32x4 *>(p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hq0 + col) * p.q_head_stride + (c & 7) * 8);
            } else if (col < nq)
                qb[s4] = *reinterpret_cast<const u32x4 *>(p.q + (int64_t)b * p.q_batch_stride +
                                                          (int64_t)(hq0 + col) * p.q_head_stride + (4 * s4 + grp) * 8);
        }
        const bool has_alibi = p.alibi_slopes != nullptr;
        const float alibi = (has_alibi && col < nq) ? p.alibi_slopes[b * p.alibi_batch_stride + hq0 + col] * 1.4426950408889634f : 0.f;

        // ---- loader (page ids through the scalar cache, buffer loads: see paged_decode_kernel) ----
        const bool paged = p.block_table != nullptr;
        const uint32_t tpp = paged ? (uint32_t)(p.page_size >> 4) : 1u;
        const uint32_t tpp_magic = tpp > 1 ? (uint32_t)((0x100000000ull + tpp - 1) / tpp) : 0u;
        const int last_pg = paged ? (L + p.page_size - 1) / p.page_size - 1 : 0;
        const int *bt_row = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
        const char *kbase = reinterpret_cast<const char *>(p.k + (int64_t)hk * p.k_head_stride);
        const char *vbase = reinterpret_cast<const char *>(p.v + (int64_t)hk * p.v_head_stride);
        if (!paged) {
            const bool cum = p.cu_seqlens_k && p.is_seqlens_k_cumulative;
            kbase += (cum ? kv_row0 * p.k_row_stride : (int64_t)b * p.k_batch_stride) * 2;
            vbase += (cum ? kv_row0 * p.v_row_stride : (int64_t)b * p.v_batch_stride) * 2;
        }
        const int64_t k_row_bytes = p.k_row_stride * 2, v_row_bytes = p.v_row_stride * 2;
        const int64_t k_page_bytes = p.k_batch_stride * 2, v_page_bytes = p.v_batch_stride * 2;
        const uint32_t k_lane_off = (uint32_t)(col * k_row_bytes + grp * 16);        // token col, chunk grp (+ 64 bytes per k-step)
        const uint32_t v_lane_off = (uint32_t)(4 * grp * v_row_bytes + col * 16);    // token 4.grp (+ one row per load), chunk col

        auto page_of = [&](int tile, uint32_t &tip) -> int {
            if (tpp == 1) { tip = 0; return tile; }
            const uint32_t pg = __umulhi((uint32_t)tile, tpp_magic);
            tip = (uint32_t)tile - pg * tpp;
            return (int)pg;
        };
        auto fetch_pid = [&](int tile) -> int {
            if (!paged) return 0;
            uint32_t tip;
            const int pg = min(page_of(tile, tip), last_pg);
            return load_ro(bt_row + pg);
        };
        auto tile_bases = [&](int tile, int pid, const char *&kt, const char *&vt) {
            if (paged) {
                uint32_t tip;
                (void)page_of(tile, tip);
                kt = kbase + (int64_t)pid * k_page_bytes + (int64_t)(tip << 4) * k_row_bytes;
                vt = vbase + (int64_t)pid * v_page_bytes + (int64_t)(tip << 4) * v_row_bytes;
            } else {
                kt = kbase + (int64_t)(tile << 4) * k_row_bytes;
                vt = vbase + (int64_t)(tile << 4) * v_row_bytes;
            }
        };
        constexpr int AUX = NT ? 2 : 0;
        auto tile_rsrc = [&](const char *base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7fffffff, 0x00020000);
        };
        auto issue_fast = [&](u32x4 (&kb)[4], u32x4 (&vb)[4], int tile, int pid) {
            const char *kt, *vt;
            tile_bases(tile, pid, kt, vt);
            const __amdgpu_buffer_rsrc_t kr = tile_rsrc(kt), vr = tile_rsrc(vt);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kb[s4] = __builtin_amdgcn_raw_buffer_load_b128(kr, k_lane_off, s4 * 64, AUX);
#pragma unroll
            for (int r = 0; r < 4; ++r) vb[r] = __builtin_amdgcn_raw_buffer_load_b128(vr, v_lane_off, (int)(r * v_row_bytes), AUX);
        };
        auto issue_tail = [&](u32x4 (&kb)[4], u32x4 (&vb)[4], int tile, int pid) {   // rows clamped to the sequence's last row
            const char *kt, *vt;
            tile_bases(tile, pid, kt, vt);
            const __amdgpu_buffer_rsrc_t kr = tile_rsrc(kt), vr = tile_rsrc(vt);
            const int lastrow = min(15, L - 1 - (tile << 4));
            const uint32_t koff = (uint32_t)(min(col, lastrow) * k_row_bytes + grp * 16);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kb[s4] = __builtin_amdgcn_raw_buffer_load_b128(kr, koff, s4 * 64, AUX);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vb[r] = __builtin_amdgcn_raw_buffer_load_b128(vr, (uint32_t)(min(4 * grp + r, lastrow) * v_row_bytes + col * 16), 0, AUX);
        };

        auto compute = [&](const u32x4 (&kb)[4], const u32x4 (&vb)[4], int tile) {
            f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = mfma16<T>(kb[s4], qb[s4], acc);
            float s[4];
            const int tok0 = (tile << 4) + 4 * grp;
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = acc[i] * sl2;                       // log2 domain
            if (has_alibi) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] -= alibi * (float)(L - 1 - (tok0 + i));   // mask.h:183, row 0 of 1
            }
            if ((tile << 4) + 16 > L) {  // wave-uniform: ragged last tile
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (tok0 + i >= L) s[i] = -INFINITY;
            }
            // all 16 tokens of the tile enter ONE accumulator, so the max of head col is taken over its four lane groups
            const float mnew = fmaxf(col_max4(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]))), m);
            if (__any(mnew > m)) {  // rescale only when some running max moved
                const float ms = mnew == -INFINITY ? 0.f : mnew;
                const float alpha = __builtin_amdgcn_exp2f(m - ms);
                l *= alpha;
                m = mnew;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ai = __uint_as_float(__builtin_amdgcn_ds_bpermute(head_lane4 + 4 * i, __float_as_uint(alpha)));
#pragma unroll
                    for (int n = 0; n < 8; ++n) o[n][i] *= ai;
                }
            }
            const float ms = m == -INFINITY ? 0.f : m;
            float pr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pr[i] = __builtin_amdgcn_exp2f(s[i] - ms);
            l += (pr[0] + pr[1]) + (pr[2] + pr[3]);
            // P.V as O[head][d] = P[head][token] . V[token][d] on v_mfma_f32_16x16x16: the A operand of lane (grp, col) is P[head col][tokens
            // 4 grp ..+3] -- the four probabilities this lane has just computed; the B operand is V[tokens 4 grp ..+3][d = 8 col + n]: the
            // lane's four V loads are exactly those rows and columns, paired along d, re-paired along the tokens with v_perm
            // (16 instructions instead of 16 + 16 G v_dot2c and G row broadcasts)
            const uint32_t pk01 = pack_pair<T>(pr[0], pr[1]), pk23 = pack_pair<T>(pr[2], pr[3]);
            const uint32_t x0[4] = {vb[0].x, vb[0].y, vb[0].z, vb[0].w}, x1[4] = {vb[1].x, vb[1].y, vb[1].z, vb[1].w};
            const uint32_t x2[4] = {vb[2].x, vb[2].y, vb[2].z, vb[2].w}, x3[4] = {vb[3].x, vb[3].y, vb[3].z, vb[3].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t e01 = __builtin_amdgcn_perm(x1[q], x0[q], 0x05040100u), e23 = __builtin_amdgcn_perm(x3[q], x2[q], 0x05040100u);
                const uint32_t o01 = __builtin_amdgcn_perm(x1[q], x0[q], 0x07060302u), o23 = __builtin_amdgcn_perm(x3[q], x2[q], 0x07060302u);
                o[2 * q] = mfma16k16<T>(pk01, pk23, e01, e23, o[2 * q]);
                o[2 * q + 1] = mfma16k16<T>(pk01, pk23, o01, o23, o[2 * q + 1]);
            }
        };

        // ---- software pipeline: P tiles in flight per wave (see paged_decode_kernel) ----
        u32x4 kb[P][4], vb[P][4];
        int pid[P];
        int t = t0;
        const int steady_end = min(t1, L >> 4);   // a ragged last tile takes the clamped loads (see paged_decode_item)
        if (t0 + 2 * P <= steady_end) {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                issue_fast(kb[s], vb[s], t0 + s, pid[s]);
                pid[s] = fetch_pid(t0 + s + P);
            }
            for (; t + 2 * P <= steady_end; t += P) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    compute(kb[s], vb[s], t + s);
                    issue_fast(kb[s], vb[s], t + s + P, pid[s]);
                    pid[s] = fetch_pid(t + s + 2 * P);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s)
                if (t0 + s < t1) {
                    issue_tail(kb[s], vb[s], t0 + s, pid[s]);
                    pid[s] = fetch_pid(t0 + s + P);
                }
        }
        for (; t < t1; t += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                if (t + s < t1) {
                    compute(kb[s], vb[s], t + s);
                    if (t + s + P < t1) {
                        issue_tail(kb[s], vb[s], t + s + P, pid[s]);
                        pid[s] = fetch_pid(t + s + 2 * P);
                    }
                }
            }
        }
    }

    // ---- the max is already common; the row sums of head col over the 4 lane groups; then each lane fetches its 4 heads' (m, l) ----
    float lt = l;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lh = __uint_as_float(__builtin_amdgcn_ds_bpermute(head_lane4 + 4 * i, __float_as_uint(lt)));
        const float mh = __uint_as_float(__builtin_amdgcn_ds_bpermute(head_lane4 + 4 * i, __float_as_uint(m)));
        const int h = 4 * grp + i;
        if (h >= nq) continue;
        const int hq = hq0 + h;
        const bool empty = !(lh > 0.f);
        const float inv = empty ? 0.f : 1.f / lh;
        const float lse = empty ? INFINITY : (mh + __builtin_amdgcn_logf(lh)) * 0.6931471805599453f;
        if constexpr (SINK) {   // workgroup-merged split mode: the piece's result stays on the CU
            float4 *dst = reinterpret_cast<float4 *>(wk.sink_o + h * D + col * 8);
            dst[0] = make_float4(o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv);
            dst[1] = make_float4(o[4][i] * inv, o[5][i] * inv, o[6][i] * inv, o[7][i] * inv);
            if (col == 0) wk.sink_lse[h] = empty ? -INFINITY : lse;
        } else if constexpr (PAIR64) {
            // this lane's eight outputs are dims 8 col ..+7 of the 128-wide row: the head's own when they lie in its kv head's half
            if ((col >> 3) != (h >= (p.g >> 1) ? 1 : 0)) continue;
            const int dc = (col & 7) * 8;
            if (!partial) {
                uint4 w4;
                w4.x = pack2<T>(o[0][i] * inv, o[1][i] * inv);
                w4.y = pack2<T>(o[2][i] * inv, o[3][i] * inv);
                w4.z = pack2<T>(o[4][i] * inv, o[5][i] * inv);
                w4.w = pack2<T>(o[6][i] * inv, o[7][i] * inv);
                *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + dc) = w4;
                if (p.lse && dc == 0) p.lse[(int64_t)b * p.h + hq] = lse;
            } else {
                const int64_t row = wk.prow + h;
                float *dst = p.o_accum + row * 64 + dc;
                partial_store(dst, o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv);
                partial_store(dst + 4, o[4][i] * inv, o[5][i] * inv, o[6][i] * inv, o[7][i] * inv);
                if (dc == 0) partial_store(p.lse_accum + row, empty ? -INFINITY : lse);
            }
        } else if (!partial) {
            uint4 w4;
            w4.x = pack2<T>(o[0][i] * inv, o[1][i] * inv);
            w4.y = pack2<T>(o[2][i] * inv, o[3][i] * inv);
            w4.z = pack2<T>(o[4][i] * inv, o[5][i] * inv);
            w4.w = pack2<T>(o[6][i] * inv, o[7][i] * inv);
            *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + col * 8) = w4;
            if (p.lse && col == 0) p.lse[(int64_t)b * p.h + hq] = lse;
        } else {
#ifdef ATOMA_DECODE_CUT_PROBE
            if (p.line_merge == 3) continue;      // TIMING PROBE ONLY: the cut pieces are not even stored
            if (p.line_merge == 5 && (wk.prow / p.group_tile) % 2 == 1) continue;   // ... only a wavefront's FIRST piece is stored (slot 0)
            if (p.line_merge == 6 && (wk.prow / p.group_tile) % 2 == 0) continue;   // ... only its LAST piece (slot 1)
            if (p.line_merge == 4) {              // ... stored with plain (write-back) stores
                float4 *d4 = reinterpret_cast<float4 *>(p.o_accum + (wk.prow + h) * D + col * 8);
                d4[0] = make_float4(o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv);
                d4[1] = make_float4(o[4][i] * inv, o[5][i] * inv, o[6][i] * inv, o[7][i] * inv);
                if (col == 0) p.lse_accum[wk.prow + h] = empty ? -INFINITY : lse;
                continue;
            }
#endif
            const int64_t row = wk.prow + h;
            float *dst = p.o_accum + row * D + col * 8;
            partial_store(dst, o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv);
            partial_store(dst + 4, o[4][i] * inv, o[5][i] * inv, o[6][i] * inv, o[7][i] * inv);
            if (col == 0) partial_store(p.lse_accum + row, empty ? -INFINITY : lse);
        }
    }
}

template <typename T, int G, int P, bool NT, bool STREAM, bool PAIR64 = false>
__global__ void __launch_bounds__(64, 2) paged_decode_mqk_kernel(const DecodeParams p) {
    decode_run_items<STREAM>(p, [](const DecodeParams &pp, const DecodeWork &wk) { paged_decode_mqk_item<T, G, P, NT, false, PAIR64>(pp, wk); },
                             DecodeLineMerge<T, PAIR64 ? 64 : 128>{});
}

// LSE-weighted merge of the split partials: /root/reference/csrc/kernels/flash_fwd_kernel.h:1204-1236.
// One wavefront per (b, q head); lane i owns D/64 output pairs.
// ---- split-KV for batches that do not fill the chip, merged INSIDE the launch ----
// The split kernels above leave one fp32 partial per (wavefront, q head) in HBM and decode_combine_kernel -- a second launch with a
// chain of dependent loads -- merges them: 4.3-11 us on top of a 15-35 us kernel (batch 1..64, the tensor-parallel rank's h_k = 1).
// Here a workgroup of NWG = 2 / 4 / 8 wavefronts owns NWG consecutive pieces of ONE (sequence, kv head, q-head chunk): every
// wavefront streams its piece exactly as before (the same item functions), leaves its normalised O and LSE in LDS, and the workgroup
// merges the NWG pieces there with the combine kernel's arithmetic (LSE weights, pieces in order).  A sequence that spans several
// workgroups (wg_splits > 1, at most a handful) goes through fp32 partials once more: every workgroup publishes its merged piece
// write-through (agent-scope 8-byte stores), takes a ticket on the sequence's arrival counter, and the LAST one to arrive reads them all
// back (agent-scope loads: past its L1 / L2), merges in piece order and writes the output -- nobody waits for anybody, so the launch
// needs no co-residency, and the result does not depend on who came last.  Counters return to zero (graph replays need no memset).
template <typename T, int D, int G, int P, bool NT, bool MQK, int NWG>
__global__ void __launch_bounds__(64 * NWG, 2) paged_decode_wg_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) float s_o[NWG][G][D];
    __shared__ float s_lse[NWG][G];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk_chunks = p.h_k * p.gchunks;
    int id = blockIdx.x;                                   // (kv head, q-head chunk) fastest, then the sequence, the piece group slowest
    const int hkc = id % hk_chunks;
    id /= hk_chunks;
    DecodeWork wk;
    wk.b = id % p.b;
    const int wg_split = id / p.b;
    wk.hk = hkc / p.gchunks;
    wk.gc = hkc % p.gchunks;
    wk.split = wg_split * NWG + wave;
    wk.L = decode_seq_len(p, wk.b);
    wk.kv_row0 = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? load_ro(p.cu_seqlens_k + wk.b) : 0;
    wk.n_tiles = (wk.L + 15) >> 4;
    const int per = (wk.n_tiles + p.num_splits - 1) / p.num_splits;
    wk.t0 = wk.split * per;
    wk.t1 = min(wk.t0 + per, wk.n_tiles);
    wk.partial = true;
    wk.prow = 0;
    wk.sink_o = &s_o[wave][0][0];
    wk.sink_lse = &s_lse[wave][0];
    if constexpr (MQK) paged_decode_mqk_item<T, G, P, NT, true>(p, wk);
    else paged_decode_item<T, D, G, P, NT, true>(p, wk);
    __syncthreads();

    const int hq0 = wk.hk * p.g + wk.gc * G, nq = min(G, p.g - wk.gc * G);
    const bool last_level = p.wg_splits <= 1;
    // merge of n pieces of head h, elements 2.d2 and 2.d2 + 1: decode_combine_kernel's arithmetic, pieces in order
    auto merge = [&](int n, auto lse_of, auto o_of, float &lse_out, float &o0, float &o1) {
        float mx = -INFINITY;
        for (int i = 0; i < n; ++i) mx = fmaxf(mx, lse_of(i));
        const float ms = mx == -INFINITY ? 0.f : mx;
        float tot = 0.f;
        for (int i = 0; i < n; ++i) tot += __expf(lse_of(i) - ms);
        const bool empty = !(tot > 0.f);
        const float lse = empty ? INFINITY : __logf(tot) + ms;
        o0 = o1 = 0.f;
        for (int i = 0; i < n; ++i) {
            const float w = empty ? 0.f : __expf(lse_of(i) - lse);
            const float2 v = o_of(i);
            o0 += w * v.x;
            o1 += w * v.y;
        }
        lse_out = lse;
    };
    auto store_final = [&](int h, int d2, float lse, float o0, float o1) {
        const int hq = hq0 + h;
        *reinterpret_cast<uint32_t *>(p.o + (int64_t)wk.b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + 2 * d2) = pack2<T>(o0, o1);
        if (p.lse && d2 == 0) p.lse[(int64_t)wk.b * p.h + hq] = lse;
    };
    for (int idx = tid; idx < nq * (D / 2); idx += 64 * NWG) {
        const int h = idx / (D / 2), d2 = idx - h * (D / 2);
        float lse, o0, o1;
        merge(NWG, [&](int i) { return s_lse[i][h]; }, [&](int i) { return *reinterpret_cast<const float2 *>(&s_o[i][h][2 * d2]); }, lse, o0, o1);
        if (last_level) {
            store_final(h, d2, lse, o0, o1);
        } else {      // this workgroup's piece: fp32, write-through, for the last arriver (LSE of an empty piece: -inf, as the split kernels write it)
            const int64_t row = ((int64_t)wg_split * p.b + wk.b) * p.h + hq0 + h;
            const unsigned long long bits = ((unsigned long long)__float_as_uint(o1) << 32) | __float_as_uint(o0);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(p.o_accum + row * D + 2 * d2), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d2 == 0) __hip_atomic_store(reinterpret_cast<unsigned *>(p.lse_accum + row), __float_as_uint(lse == INFINITY ? -INFINITY : lse), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (last_level) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wavefront drains, then ONE ticket per workgroup
    __syncthreads();
    if (tid == 0) s_ticket = sync_arrive(p.counters + ((int64_t)wk.b * hk_chunks + hkc), sync_epoch(), (unsigned)p.wg_splits);
    __syncthreads();
    if (s_ticket + 1 != (unsigned)p.wg_splits) return;
    for (int idx = tid; idx < nq * (D / 2); idx += 64 * NWG) {
        const int h = idx / (D / 2), d2 = idx - h * (D / 2);
        const int64_t row0 = (int64_t)wk.b * p.h + hq0 + h, step = (int64_t)p.b * p.h;
        float lse, o0, o1;
        merge(p.wg_splits,
              [&](int i) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p.lse_accum + row0 + i * step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); },
              [&](int i) {
                  const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p.o_accum + (row0 + i * step) * D + 2 * d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  return make_float2(__uint_as_float((unsigned)bits), __uint_as_float((unsigned)(bits >> 32)));
              },
              lse, o0, o1);
        store_final(h, d2, lse, o0, o1);
    }
}

// ---- narrow ragged batches: two sequences per workgroup (round 6 probe, option decode_pair) ----------------------------------------------------------
// BASELINE configs[2] mid-trace: 256 sequences of U[2048, 2560) tokens x 8 kv heads = 2048 (sequence, kv head) units for 2048 resident wavefronts --
// one unit each, so the launch lasts as long as its LONGEST sequence (0.383 ms against 0.356 for a uniform batch of the same bytes) and cutting
// sequences between wavefronts costs more than it returns below ~15 % idle share.  Here a workgroup of 8 wavefronts takes a PAIR of sequences --
// the i-th shortest with the i-th longest (each workgroup ranks the batch's lengths itself: 256 compares per thread) -- and 4 kv heads: wavefront
// (head j, half k) streams half k of sequence A's tiles, the two halves of a head merge in LDS (paged_decode_wg_kernel's arithmetic), then the same
// for sequence B.  Every wavefront of the launch then streams (len_A + len_B) / 2 tokens: balanced to the pairing's residue, no partial ever leaves the CU.
template <typename T, int G, int P, bool NT>
__global__ void __launch_bounds__(512, 2) paged_decode_pair_kernel(const DecodeParams p) {
    constexpr int D = 128, NW = 8;
    __shared__ __attribute__((aligned(16))) float s_o[NW][G][D];
    __shared__ float s_lse[NW][G];
    __shared__ int s_len[DECODE_PAIR_MAX_B];
    __shared__ short s_order[DECODE_PAIR_MAX_B];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int s_lo[NW], s_hi[NW];
    const int hk_chunks = p.h_k * p.gchunks, hgroups = hk_chunks / 4;
    int lo = 0x7fffffff, hi = -1;
    for (int b = tid; b < p.b; b += 64 * NW) { const int l = decode_seq_len(p, b); s_len[b] = l; lo = min(lo, l); hi = max(hi, l); }
#pragma unroll
    for (int off = 32; off; off >>= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
    if ((tid & 63) == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
    // A batch of EQUAL lengths has nothing to pair: there the workgroup is the 8 (kv head, q chunk) units of one sequence or of two, one
    // wavefront each, written straight to the output -- the kv-head-major order of the balanced line by construction (same grid: B . units / 8
    // workgroups) -- and nothing is ranked.
    if (lo == hi && (p.b * hk_chunks) % 8 == 0) {
        const int unit = blockIdx.x * 8 + wave;            // sequence-major: the 8 wavefronts of a workgroup = consecutive kv heads of one sequence
        DecodeWork wk;
        wk.b = unit / hk_chunks;
        if (wk.b >= p.b) return;                           // (an odd batch: the grid is rounded up to whole pairs)
        const int c = unit - wk.b * hk_chunks;
        wk.hk = c / p.gchunks;
        wk.gc = c % p.gchunks;
        wk.split = 0;
        wk.L = lo;
        wk.kv_row0 = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? load_ro(p.cu_seqlens_k + wk.b) : 0;
        wk.n_tiles = (wk.L + 15) >> 4;
        wk.t0 = 0;
        wk.t1 = wk.n_tiles;
        wk.partial = false;
        wk.balanced = false;
        wk.prow = 0;
        wk.sink_o = nullptr;
        wk.sink_lse = nullptr;
        paged_decode_mqk_item<T, G, P, NT, false>(p, wk);
        return;
    }
    for (int b = tid; b < p.b; b += 64 * NW) {
        const int lb = s_len[b];
        int rank = 0;
        for (int j = 0; j < p.b; ++j) { const int lj = s_len[j]; rank += (lj < lb || (lj == lb && j < b)) ? 1 : 0; }
        s_order[rank] = (short)b;
    }
    __syncthreads();
    const int pair = blockIdx.x / hgroups, hg = blockIdx.x - pair * hgroups;
    const int seq[2] = {s_order[pair], s_order[p.b - 1 - pair]};
    const int nseq = seq[0] == seq[1] ? 1 : 2;
    const int hkc = hg * 4 + (wave >> 1), half = wave & 1;
    for (int s = 0; s < nseq; ++s) {
        DecodeWork wk;
        wk.b = seq[s];
        wk.hk = hkc / p.gchunks;
        wk.gc = hkc % p.gchunks;
        wk.split = half;
        wk.L = s_len[wk.b];
        wk.kv_row0 = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? load_ro(p.cu_seqlens_k + wk.b) : 0;
        wk.n_tiles = (wk.L + 15) >> 4;
        const int per = (wk.n_tiles + 1) >> 1;
        wk.t0 = half * per;
        wk.t1 = min(wk.t0 + per, wk.n_tiles);
        wk.partial = true;
        wk.balanced = false;
        wk.prow = 0;
        wk.sink_o = &s_o[wave][0][0];
        wk.sink_lse = &s_lse[wave][0];
        paged_decode_mqk_item<T, G, P, NT, true>(p, wk);
        __syncthreads();
        // the two halves of every (kv head, q head): decode_combine_kernel's arithmetic, pieces in order
        for (int idx = tid; idx < 4 * G * (D / 2); idx += 64 * NW) {
            const int u = idx / (G * (D / 2)), r = idx - u * (G * (D / 2)), h = r / (D / 2), d2 = r - h * (D / 2);
            const int c = hg * 4 + u, hk = c / p.gchunks, gc = c % p.gchunks;
            if (h >= min(G, p.g - gc * G)) continue;
            const float l0 = s_lse[2 * u][h], l1 = s_lse[2 * u + 1][h];
            const float mx = fmaxf(l0, l1), ms = mx == -INFINITY ? 0.f : mx;
            const float tot = __expf(l0 - ms) + __expf(l1 - ms);
            const bool empty = !(tot > 0.f);
            const float lse = empty ? INFINITY : __logf(tot) + ms;
            const float w0 = empty ? 0.f : __expf(l0 - lse), w1 = empty ? 0.f : __expf(l1 - lse);
            const float2 a = *reinterpret_cast<const float2 *>(&s_o[2 * u][h][2 * d2]), b2 = *reinterpret_cast<const float2 *>(&s_o[2 * u + 1][h][2 * d2]);
            const int hq = hk * p.g + gc * G + h;
            *reinterpret_cast<uint32_t *>(p.o + (int64_t)wk.b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + 2 * d2) = pack2<T>(w0 * a.x + w1 * b2.x, w0 * a.y + w1 * b2.y);
            if (p.lse && d2 == 0) p.lse[(int64_t)wk.b * p.h + hq] = lse;
        }
        __syncthreads();
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(64) decode_combine_kernel(const DecodeParams p) {
    const int lane = threadIdx.x;
    if (p.stream_waves > 0 && p.plan[1] == 0) return;    // balanced mode, but one wavefront per sequence was taken: outputs are final
    const int64_t stride = (int64_t)p.b * p.h;
    // grid-stride over (b, q head): a uniform batch launches this kernel only to find nothing to merge, so keep the grid small
    for (int64_t bh = blockIdx.x; bh < stride; bh += gridDim.x) [&] {
    const int b = (int)(bh / p.h), hq = (int)(bh % p.h);
    int nsp = p.num_splits;
    int64_t row0 = bh, row_step = stride, row1 = bh;   // partial row of piece s: s == 0 ? row0 : row1 + s * row_step
    if (p.stream_waves > 0) {   // balanced mode: merge the pieces of a sequence that was cut between wavefronts
        const int tiles = p.plan[0], c0 = p.plan[2 + b], n = p.plan[3 + b] - c0;
        uint16_t *dst = p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride;
        if (n == 0) {                                 // empty sequence: no wavefront saw it (flash_fwd_kernel.h:543-582)
            for (int e = lane; e < D; e += 64) dst[e] = 0;
            if (p.lse && lane == 0) p.lse[bh] = INFINITY;
            return;
        }
        const int hk = hq / p.g, in_group = hq - hk * p.g;
        const int hkc = hk * p.gchunks + in_group / p.group_tile, gq = in_group % p.group_tile;
        const int64_t total = (int64_t)tiles * p.h_k * p.gchunks;
        const int64_t share = max((total + p.stream_waves - 1) / p.stream_waves, (int64_t)DECODE_MIN_SHARE);
        const int64_t s0 = (int64_t)hkc * tiles + c0, w0 = s0 / share, w1 = (s0 + n - 1) / share;
        if (w0 == w1) return;                         // whole sequence inside one wavefront's range: written directly
        nsp = (int)(w1 - w0 + 1);
        row0 = (w0 * 2 + (s0 > w0 * share ? 1 : 0)) * p.group_tile + gq;   // the first piece is that wavefront's last segment unless it starts its range
        row_step = 2 * p.group_tile;
        row1 = w0 * 2 * p.group_tile + gq;
    }
    auto prow = [&](int s) -> int64_t { return s == 0 ? row0 : row1 + s * row_step; };
    // LSE pass: lane s owns piece s (pieces beyond 64 wrap around)
    float mx = -INFINITY;
    for (int s = lane; s < nsp; s += 64) mx = fmaxf(mx, p.lse_accum[prow(s)]);
#pragma unroll
    for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float ms = mx == -INFINITY ? 0.f : mx;
    float tot = 0.f;
    for (int s = lane; s < nsp; s += 64) tot += __expf(p.lse_accum[prow(s)] - ms);
#pragma unroll
    for (int off = 32; off; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const bool empty = !(tot > 0.f);
    const float lse = empty ? INFINITY : __logf(tot) + ms;
    // O pass: 16 lanes per partial row (EPL floats each), four rows per step and four steps unrolled, so 16 independent
    // row loads are in flight per wavefront -- merging 32 pieces one dependent load at a time took 8-16 us, longer than
    // the decode kernel itself at batch 1.
    constexpr int EPL = D / 16;
    const int grp = lane >> 4, col = lane & 15;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int c0 = 0; c0 < nsp; c0 += 64) {
        const int mine = c0 + lane;
        const float wl = (mine < nsp && !empty) ? __expf(p.lse_accum[prow(mine)] - lse) : 0.f;   // weight of piece c0 + lane
        const int chunk = min(64, nsp - c0);
        for (int j0 = 0; j0 < chunk; j0 += 16) {
            float4 v[4][EPL / 4];
            float wgt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = j0 + 4 * u + grp;                       // piece (within the chunk) of this lane group
                wgt[u] = __shfl(wl, sl & 63, 64);
                const bool on = sl < chunk;
                if (!on) wgt[u] = 0.f;
                const float4 *src = reinterpret_cast<const float4 *>(p.o_accum + prow(on ? c0 + sl : c0) * D + col * EPL);
#pragma unroll
                for (int q = 0; q < EPL / 4; ++q) v[u][q] = src[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < EPL / 4; ++q) {
                    acc[4 * q + 0] += wgt[u] * v[u][q].x;
                    acc[4 * q + 1] += wgt[u] * v[u][q].y;
                    acc[4 * q + 2] += wgt[u] * v[u][q].z;
                    acc[4 * q + 3] += wgt[u] * v[u][q].w;
                }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if (grp == 0) {
        uint16_t *dst = p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + col * EPL;
#pragma unroll
        for (int e = 0; e < EPL; e += 2) *reinterpret_cast<uint32_t *>(dst + e) = pack2<T>(acc[e], acc[e + 1]);
    }
    if (p.lse && lane == 0) p.lse[bh] = lse;
    }();
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
void *workspace(hipStream_t stream, size_t bytes);   // runtime.hip: grow-only scratch per (device, stream), never freed under a graph
sync_word_t *sync_counters(hipStream_t stream);      // runtime.hip: epoch-tagged arrival words per (device, stream)
constexpr int64_t DECODE_WG_MAX_COUNTERS = 8192;     // = SYNC_COUNTERS of runtime.hip

// Split count for THIS kernel: enough wavefronts to fill the resident slots of every CU (8, or 4 for the
// 512-register G = 8 / d = 128 variant), never fewer than `min_tiles` 16-token tiles per split.  (The reference's heuristic, lib.rs:2122-2199,
// is tuned for 128-thread CTAs of a 64-row tile; it is restated as atoma_compute_num_splits.)
int decode_num_splits(int64_t waves_per_split, int max_seqlen_k, int waves_per_cu, int min_tiles) {
    const int64_t target = (int64_t)device_num_cus() * waves_per_cu;
    // at exactly half the resident slots (the 2-way kv-head shard of the headline: 256 sequences x 4 kv heads = 1024 wavefronts on 2048 slots)
    // one piece per sequence on the balanced line beats two pieces + combine: 0.370 -> 0.325 ms
    if (waves_per_split * 2 >= target) return 1;
    const int64_t n_tiles = cdiv(max_seqlen_k, 16);
    auto clamp = [&](int64_t s, int64_t tiles_per_split) {
        const int64_t max_s = std::max<int64_t>(1, n_tiles / std::max<int64_t>(1, tiles_per_split));
        return std::max<int64_t>(1, std::min<int64_t>(std::min(s, max_s), 128));
    };
    // Measured (tools/probes): wavefronts of 32+ tiles want every resident slot filled (the bandwidth against
    // active wavefronts curve), shorter ones pay more for their ramp and their partials than the extra wavefronts bring --
    // then one wavefront per SIMD with pieces of at least `min_tiles` is 3-9 % faster than two.
    int64_t s = clamp(cdiv(target, waves_per_split), std::max(min_tiles, 32));
    if (waves_per_split * s * 2 < target) s = clamp(cdiv(target / 2, waves_per_split), min_tiles);
    return (int)s;
}

// Tuning knobs (atoma_set_option / environment, for A/B runs and tests):
//   decode_p            ATOMA_DECODE_P            tiles in flight per wave (2..4)
//   decode_nt           ATOMA_DECODE_NT           0/1 non-temporal K/V loads
//   decode_stream       ATOMA_DECODE_STREAM       balanced mode for large batches with device-side lengths (decode_run_items): 0 off, 1 default, 2 always, 3 ragged only
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
// One OS thread per GPU calls into the library (model_executor.rs:428): the knobs are relaxed atomics, initialised once
// (function-local static), so a set_option from one thread never tears a read on another.
typedef std::atomic<int> opt_int;
struct DecodeOptions {
    opt_int p{env_int("ATOMA_DECODE_P", DECODE_DEFAULT_P)};
    opt_int nt{env_int("ATOMA_DECODE_NT", DECODE_DEFAULT_NT)};
    opt_int stream{env_int("ATOMA_DECODE_STREAM", 1)};
    opt_int stream_waves_per_cu{env_int("ATOMA_DECODE_STREAM_WAVES_PER_CU", 0)};   // 0 = resident capacity
    opt_int waves_per_cu{env_int("ATOMA_DECODE_WAVES_PER_CU", 0)};   // 0 = resident capacity
    opt_int min_tiles{env_int("ATOMA_DECODE_MIN_TILES", 8)};
    opt_int fp8_wg{env_int("ATOMA_DECODE_FP8_WG", 0)};     // fp8 KV cache: 8 wavefronts (the kv heads of a sequence) per workgroup: 0 never (default since the kv-head-major order does the same for free), 1 split-KV launches, 2 always
    opt_int fp8_mqk{env_int("ATOMA_DECODE_FP8_MQK", 1)};   // fp8 KV cache: q.K^T of the converted K on the matrix cores (1) or v_dot2c (0)
    opt_int pair{env_int("ATOMA_DECODE_PAIR", 0)};         // two sequences per workgroup (paged_decode_pair_kernel): 0 = when atoma_prepare_inputs packed a RAGGED batch of this size last (the hint above) and the launch gives every resident wavefront about one unit, 1 = for every such launch, 2 = whenever applicable, -1 = never
    opt_int wg_merge{env_int("ATOMA_DECODE_WG_MERGE", 1)};   // split-KV merged inside the launch (paged_decode_wg_kernel) instead of split kernel + combine kernel
    opt_int pair64{env_int("ATOMA_DECODE_PAIR64", 1)};   // head_dim 64 with an even number of kv heads and groups of 1 / 2 / 4 q heads: two kv heads per wavefront on the matrix-core kernel (1) or the dot2 kernel (0)
    opt_int line_merge{env_int("ATOMA_DECODE_LINE_MERGE", 1)};   // balanced line: cut sequences merged by the last wavefront to arrive (1) or by decode_combine_kernel (0)
    opt_int fp8_klines{env_int("ATOMA_DECODE_FP8_KLINES", 1)};   // fp8 matrix-core kernel, K in full 128-byte lines: 0 never, 1 where it pays, 2 always
    opt_int head_major{env_int("ATOMA_DECODE_HEAD_MAJOR", 1)};   // workgroup order of the non-balanced launches: kv head slowest (1) or fastest (0)
    opt_int mqk{env_int("ATOMA_DECODE_MQK", 29)};   // q.K^T on the matrix cores at d = 128: bit 0 = groups of more than 4 q heads, bit 1 = all smaller groups, bit 2 = groups of 2..4 at tiny batches, bit 3 = groups of 2..4 on the balanced line, bit 4 = groups of 2..4 on split-KV launches
};
// Lengths of the decode sequences of the batch atoma_prepare_inputs last packed on this device (one host thread per GPU: model_executor.rs:428) -- the
// one thing the dispatcher cannot see for itself (the lengths live on the device) and the paired kernel's choice depends on: exactly uniform batches lose on
// it, ragged ones win.  A HINT: it changes which kernel runs, never a result; a caller that does not prepare its batches through the library never sets it.
struct DecodeLengthHint { std::atomic<int64_t> min_len{0}, max_len{0}, count{0}; };
static DecodeLengthHint *decode_length_hints() { static DecodeLengthHint h[64]; return h; }
void note_decode_lengths(int64_t min_len, int64_t max_len, int64_t count) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return; }
    DecodeLengthHint &h = decode_length_hints()[dev];
    h.min_len = min_len; h.max_len = max_len; h.count = count;
}
}  // namespace atoma
// The same hint from a caller that packs its batches itself: the shortest / longest decode sequence of the batch about to be decoded and their number on
// the current device (0, 0, 0 forgets it).
extern "C" int atoma_hint_decode_lengths(int64_t min_len, int64_t max_len, int64_t count) {
    atoma::clear_error();
    if (min_len < 0 || max_len < min_len || count < 0) { atoma::set_error("atoma_hint_decode_lengths: need 0 <= min_len <= max_len and count >= 0"); return -1; }
    atoma::note_decode_lengths(min_len, max_len, count);
    return 0;
}
namespace atoma {
// 1: the batch packed last on this device was ragged and had `b` decode sequences
static bool decode_hint_says_ragged(int b) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    const DecodeLengthHint &h = decode_length_hints()[dev];
    return h.count.load() == b && h.max_len.load() > h.min_len.load();
}
static DecodeOptions &decode_options() {
    static DecodeOptions o;
    return o;
}
extern std::atomic<int> prefill_cfg;  // prefill_mfma.hip
extern std::atomic<int> prefill_exact_keys, prefill_simple;  // prefill_asm.hip
extern std::atomic<int64_t> rope_table_rows;  // norm_rope.hip
bool set_decode_option(const std::string &name, int value) {
    DecodeOptions &o = decode_options();
    if (name == "prefill_cfg") { prefill_cfg = value; return true; }
    if (name == "prefill_exact_keys") { prefill_exact_keys = value; return true; }
    if (name == "prefill_simple") { prefill_simple = value; return true; }
    if (name == "rope_table_rows") { rope_table_rows = value < 0 ? 0 : value; return true; }
    if (name == "decode_p") o.p = value;
    else if (name == "decode_nt") o.nt = value;
    else if (name == "decode_stream") o.stream = value;
    else if (name == "decode_stream_waves_per_cu") o.stream_waves_per_cu = value;
    else if (name == "decode_waves_per_cu") o.waves_per_cu = value;
    else if (name == "decode_min_tiles") o.min_tiles = value;
    else if (name == "decode_mqk") o.mqk = value;
    else if (name == "decode_wg_merge") o.wg_merge = value;
    else if (name == "decode_pair") o.pair = value;
    else if (name == "decode_line_merge") o.line_merge = value;
    else if (name == "decode_pair64") o.pair64 = value;
    else if (name == "decode_fp8_mqk") o.fp8_mqk = value;
    else if (name == "decode_fp8_wg") o.fp8_wg = value;
    else if (name == "decode_head_major") o.head_major = value;
    else if (name == "decode_fp8_klines") o.fp8_klines = value;
    else return false;
    return true;
}

// Combine grid: one wavefront merges the pieces of one (sequence, q head) -- a chain of dependent loads (plan -> LSEs -> partial rows,
// ~3 us).  In a ragged batch nearly every (sequence, kv head) is cut once, so all b.h rows merge: with 4 workgroups per CU the
// 8192 rows of the 8B step took 8 rounds = 25.6 us per layer (rocprofv3 on the decode step); 32 per CU take one or two.
static int combine_grid_cap() {
    static const int per_cu = env_int("ATOMA_DECODE_COMBINE_WG_PER_CU", 32);
    return device_num_cus() * std::max(1, per_cu);
}

// Balanced mode: as many wavefronts share the batch as THIS kernel keeps resident (its register count decides: 8 per CU
// for the 4-head variants, 12 for one head per wavefront); the workspace was sized for the upper bound of 16.
#define DECODE_STREAM_MAX_WAVES_PER_CU 16
template <typename K> static int resident_waves_per_cu(K kernel) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 64, 0) != hipSuccess || n <= 0) n = 8;
    return std::min(n, DECODE_STREAM_MAX_WAVES_PER_CU);
}
// ... and, with the number of wavefronts on the line known, whether its cut sequences are merged inside the launch (decode_line_merge)
static void set_stream_waves(DecodeParams &p, int occupancy, hipStream_t stream) {
    const int opt = decode_options().stream_waves_per_cu;
    const int wpc = opt > 0 ? std::min(opt, DECODE_STREAM_MAX_WAVES_PER_CU) : occupancy;
    p.stream_waves = (int)std::min<int64_t>((int64_t)p.b * p.h_k * p.gchunks, (int64_t)device_num_cus() * wpc);
    p.line_merge = 0;
    if (decode_options().line_merge && p.stream_waves <= DECODE_WG_MAX_COUNTERS) {
        p.counters = sync_counters(stream);
        p.line_merge = p.counters ? 1 : 0;
        if (!p.counters) clear_error();           // (a capture without warm-up: the combine kernel serves)
    }
#ifdef ATOMA_DECODE_CUT_PROBE   // TIMING PROBE ONLY (make cutprobe; results are wrong): decode_line_merge = 2 -> nobody merges the cut pieces
    if (decode_options().line_merge >= 2) p.line_merge = decode_options().line_merge;
#endif
}

// Which decode kernel the dispatcher took last on this thread (bench.py labels its roofline line with it instead of a literal)
static thread_local std::string g_last_decode_kernel;
template <typename T> static const char *decode_tname() { return std::is_same<T, bf16_t>::value ? "bf16" : "f16"; }
static void note_decode_kernel(const char *kernel, const char *t, int d, int g, int p, bool nt, const char *mode) {
    g_last_decode_kernel = std::string(kernel) + "<" + t + ",D=" + std::to_string(d) + ",G=" + std::to_string(g) + ",P=" + std::to_string(p) +
                           (nt ? ",nt" : "") + "," + mode + ">";
}

template <typename T, int D, int G, int P, int MINW, bool NT>
static void launch_decode_cfg(DecodeParams &p, hipStream_t stream) {
    const int64_t blocks = (int64_t)p.b * p.num_splits * p.h_k * p.gchunks;
    if (p.stream_waves > 0) {
        static const int occ = resident_waves_per_cu(paged_decode_kernel<T, D, G, P, MINW, NT, true>);
        set_stream_waves(p, occ, stream);
    }
    note_decode_kernel("paged_decode_kernel", decode_tname<T>(), D, G, P, NT,
                       p.stream_waves > 0 ? "balanced" : (p.num_splits > 1 ? (std::to_string(p.num_splits) + " KV splits + combine").c_str() : "one wavefront per (sequence, kv head)"));
    if (p.stream_waves > 0) hipLaunchKernelGGL((paged_decode_kernel<T, D, G, P, MINW, NT, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((paged_decode_kernel<T, D, G, P, MINW, NT, false>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
}

// Split-KV merged inside the launch: workgroups of NWG = 8 / 4 / 2 wavefronts, wg_splits workgroups per sequence (p.num_splits is
// rounded down to a multiple of NWG).  False when this launch must take the split + combine kernels instead.
static int decode_wg_waves(const DecodeParams &p) {
    // wavefronts per workgroup: as many as keep one workgroup on every CU (a CU holds only ~40-50 KB of requests in flight whatever its
    // wavefronts ask for, DESIGN.md 4.8c: 128 workgroups of 8 wavefronts stream slower than 1024 single wavefronts on 256 CUs)
    const int64_t waves = (int64_t)p.b * p.h_k * p.gchunks * p.num_splits;
    int nwg = 2;
    while (nwg < 8 && nwg * 2 <= p.num_splits && waves / (nwg * 2) >= device_num_cus()) nwg *= 2;
    return nwg;
}
// Option decode_wg_merge: 1 (default) = only launches whose pieces all fit ONE workgroup per sequence (the merge never leaves the CU:
// the 70B TP = 8 shard at batch 256: 95.5 -> 87.8 us); 2 = also with a last-arriver merge across workgroups -- measured level with the
// combine kernel (shard at batch 64: 33.1 vs 31.8 us; 16 x 8192: 96.6 vs 97.9), so the two-kernel route keeps those; 0 = never.
static bool decode_wg_applicable(const DecodeParams &p) {
    const int opt = decode_options().wg_merge;
    if (p.num_splits <= 1 || p.stream_waves != 0 || opt == 0 || p.k_scale || (int64_t)p.b * p.h_k * p.gchunks > DECODE_WG_MAX_COUNTERS) return false;
    if ((int64_t)p.b * p.h_k * p.gchunks * p.num_splits / 2 < device_num_cus() && opt < 3) return false;   // too few wavefronts to pair them up
    return opt >= 2 || p.num_splits / decode_wg_waves(p) <= 1;
}
template <typename T, int D, int G, int P, bool NT, bool MQK>
static bool launch_decode_wg(DecodeParams &p, hipStream_t stream) {
    const int nwg = decode_wg_waves(p);
    p.wg_splits = p.num_splits / nwg;
    p.num_splits = p.wg_splits * nwg;
    if (p.wg_splits > 1) {
        p.counters = sync_counters(stream);
        if (!p.counters) return false;
    }
    const dim3 grid((unsigned)((int64_t)p.b * p.h_k * p.gchunks * p.wg_splits));
    note_decode_kernel(MQK ? "paged_decode_wg_kernel(matrix-core scores)" : "paged_decode_wg_kernel", decode_tname<T>(), D, G, P, NT,
                       (std::to_string(nwg) + " wavefronts x " + std::to_string(p.wg_splits) + " workgroups per sequence, merged in the launch").c_str());
    if (nwg == 8) hipLaunchKernelGGL((paged_decode_wg_kernel<T, D, G, P, NT, MQK, 8>), grid, dim3(512), 0, stream, p);
    else if (nwg == 4) hipLaunchKernelGGL((paged_decode_wg_kernel<T, D, G, P, NT, MQK, 4>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((paged_decode_wg_kernel<T, D, G, P, NT, MQK, 2>), grid, dim3(128), 0, stream, p);
    return ATOMA_CHECK_LAUNCH("paged_decode_wg_kernel");
}

// d = 128: scores on the matrix cores (paged_decode_mqk_kernel), any G at two wavefronts per SIMD
template <typename T, int G, bool PAIR64 = false>
static void launch_decode_mqk(DecodeParams &p, hipStream_t stream) {
    // 3 tiles in flight for every group size since P.V runs on the matrix cores (32 accumulator registers whatever G; option decode_mqk_p8:
    // tiles in flight at 8 q heads per wavefront, A/B)
    static const int p8 = env_int("ATOMA_DECODE_MQK_P8", 3);
    const bool p3 = decode_options().p >= 3 && (G <= 4 || p8 >= 3), nt = decode_options().nt != 0;
    if constexpr (!PAIR64) {
        if (decode_wg_applicable(p)) {
            if (nt) { if (p3) launch_decode_wg<T, 128, G, 3, true, true>(p, stream); else launch_decode_wg<T, 128, G, 2, true, true>(p, stream); }
            else { if (p3) launch_decode_wg<T, 128, G, 3, false, true>(p, stream); else launch_decode_wg<T, 128, G, 2, false, true>(p, stream); }
            return;
        }
    }
    if constexpr (!PAIR64) {
        // two sequences per workgroup: lengths on the device, no KV split, groups of 4 (kv head, q chunk) units, at least one unit per resident wavefront
        const int pair_opt = decode_options().pair;
        const int64_t units = (int64_t)p.b * p.h_k * p.gchunks;
        if (pair_opt >= 0 && p.stream_waves > 0 && p.num_splits == 1 && !p.k_scale && (p.h_k * p.gchunks) % 4 == 0 && p.b >= 2 && p.b <= DECODE_PAIR_MAX_B &&
            (pair_opt == 2 || (units >= (int64_t)device_num_cus() * 6 && units <= (int64_t)device_num_cus() * 8 && (pair_opt == 1 || decode_hint_says_ragged(p.b))))) {
            p.stream_waves = 0;
            const dim3 grid((unsigned)(((int64_t)p.b + 1) / 2 * (p.h_k * p.gchunks / 4)));
            note_decode_kernel("paged_decode_pair_kernel", decode_tname<T>(), 128, G, p3 ? 3 : 2, nt, "two sequences per workgroup, halves merged in LDS");
            if (nt) { if (p3) hipLaunchKernelGGL((paged_decode_pair_kernel<T, G, 3, true>), grid, dim3(512), 0, stream, p); else hipLaunchKernelGGL((paged_decode_pair_kernel<T, G, 2, true>), grid, dim3(512), 0, stream, p); }
            else { if (p3) hipLaunchKernelGGL((paged_decode_pair_kernel<T, G, 3, false>), grid, dim3(512), 0, stream, p); else hipLaunchKernelGGL((paged_decode_pair_kernel<T, G, 2, false>), grid, dim3(512), 0, stream, p); }
            ATOMA_CHECK_LAUNCH("paged_decode_pair_kernel");
            return;
        }
    }
    const int64_t blocks = (int64_t)p.b * p.num_splits * p.h_k * p.gchunks;
    if (p.stream_waves > 0) set_stream_waves(p, 8, stream);   // __launch_bounds__(64, 2)
    note_decode_kernel(PAIR64 ? "paged_decode_mqk_kernel(kv-head pairs)" : "paged_decode_mqk_kernel", decode_tname<T>(), PAIR64 ? 64 : 128, G, p3 ? 3 : 2, nt,
                       p.stream_waves > 0 ? "balanced" : (p.num_splits > 1 ? (std::to_string(p.num_splits) + " KV splits + combine").c_str() : "one wavefront per (sequence, kv head)"));
#define ATOMA_MQK(P_, NT_, S_) hipLaunchKernelGGL((paged_decode_mqk_kernel<T, G, P_, NT_, S_, PAIR64>), dim3((unsigned)blocks), dim3(64), 0, stream, p)
#define ATOMA_MQK_S(P_, NT_) do { if (p.stream_waves > 0) ATOMA_MQK(P_, NT_, true); else ATOMA_MQK(P_, NT_, false); } while (0)
    // (4 and 5 tiles in flight for the 8-head groups of split-KV launches, 8 / 16 / 32 splits, in-launch merge or combine kernel: all within
    // 29.3-32 us on the 70B shard's B = 64 call -- profiles/r04_decode_b64_hk1_depth_probe.txt; launch, ramp and merge bound it, not depth)
    if (nt) { if (p3) ATOMA_MQK_S(3, true); else ATOMA_MQK_S(2, true); }
    else { if (p3) ATOMA_MQK_S(3, false); else ATOMA_MQK_S(2, false); }
#undef ATOMA_MQK_S
#undef ATOMA_MQK
    if (!ATOMA_CHECK_LAUNCH("paged_decode_mqk_kernel")) return;
    if (p.num_splits > 1 || (p.stream_waves > 0 && !p.line_merge)) {   // (cut probe: line_merge = 2 skips this launch too)
        hipLaunchKernelGGL((decode_combine_kernel<T, PAIR64 ? 64 : 128>), dim3((unsigned)std::min<int64_t>((int64_t)p.b * p.h, combine_grid_cap())), dim3(64), 0, stream, p);
        ATOMA_CHECK_LAUNCH("decode_combine_kernel");
    }
}

template <typename T, int D, int G>
static void launch_decode_tdg(DecodeParams &p, hipStream_t stream) {
    // tiles in flight per wave / waves per SIMD the register budget is capped for: G = 8 at
    // D = 128 needs more than 256 VGPRs (64 for O, 32 for q, 64 per K+V pair in flight).
    constexpr int MINW = (G >= 8 && D >= 128) ? 1 : 2;
    const int cfg_p = decode_options().p, cfg_nt = decode_options().nt;
    const int P = (G >= 8 && cfg_p > 2) ? 2 : cfg_p;
    if constexpr (MINW == 2) {                   // (the 8-head dot2 variant needs the 512 registers of one wavefront per SIMD)
        if (decode_wg_applicable(p)) {
            if (cfg_nt) { if (P >= 3) launch_decode_wg<T, D, G, 3, true, false>(p, stream); else launch_decode_wg<T, D, G, 2, true, false>(p, stream); }
            else { if (P >= 3) launch_decode_wg<T, D, G, 3, false, false>(p, stream); else launch_decode_wg<T, D, G, 2, false, false>(p, stream); }
            return;
        }
    }
    if (cfg_nt) {
        if (P >= 4) launch_decode_cfg<T, D, G, 4, MINW, true>(p, stream);
        else if (P == 3) launch_decode_cfg<T, D, G, 3, MINW, true>(p, stream);
        else launch_decode_cfg<T, D, G, 2, MINW, true>(p, stream);
    } else {
        if (P >= 4) launch_decode_cfg<T, D, G, 4, MINW, false>(p, stream);
        else if (P == 3) launch_decode_cfg<T, D, G, 3, MINW, false>(p, stream);
        else launch_decode_cfg<T, D, G, 2, MINW, false>(p, stream);
    }
    if (!ATOMA_CHECK_LAUNCH("paged_decode_kernel")) return;
    if (p.num_splits > 1 || (p.stream_waves > 0 && !p.line_merge)) {
        hipLaunchKernelGGL((decode_combine_kernel<T, D>), dim3((unsigned)std::min<int64_t>((int64_t)p.b * p.h, combine_grid_cap())), dim3(64), 0, stream, p);
        ATOMA_CHECK_LAUNCH("decode_combine_kernel");
    }
}

// The launch decisions that do not depend on the element type: q heads per wavefront (G), matrix-core scores or not,
// KV splits, balanced mode, and the fp32 scratch rows they need.  Shared by the launcher and by atoma_warmup's sizing.
struct DecodeLaunchPlan {
    int G;
    bool use_mqk;
    size_t rows;          // fp32 partial rows of D floats (+ 1 LSE each); 0 = no scratch
    size_t bytes;         // scratch bytes including the plan ints
};
int decode_fp8_mma_group();   // paged_decode_fp8.hip
static DecodeLaunchPlan decode_plan_launch(DecodeParams &p, int D, bool fp8 = false) {
    const int g = p.g;
    // matrix-core scores: d = 128, selected groups (option decode_mqk: bit 0 = groups of 5..8+ q heads, bit 1 = smaller ones, bit 2 below)
    const int mqk_opt = decode_options().mqk;
    // bit 2: groups of 2..4 when the batch is tiny (b * h_k <= 64): every wavefront is latency-bound there and the
    // matrix-core kernel's instruction stream per tile is 1.5x shorter (B=1: 32 -> 27 us, 70B TP=8 shard B=64: -6 %);
    // at larger batches the two kernels tie or the dot2 kernel wins by 2-3 %, and MHA always prefers dot2.
    const bool tiny = g >= 2 && g <= 4 && (int64_t)p.b * p.h_k <= 64;
    const bool use_mqk = !fp8 && D == 128 && ((g > 4 && (mqk_opt & 1)) || (g <= 4 && (mqk_opt & 2)) || (tiny && (mqk_opt & 4)));
    // fp8: the matrix-core kernel takes up to 16 q heads per wavefront in one pass, the dot2 kernel 4
    const int G = fp8 ? (decode_options().fp8_mqk != 0 ? decode_fp8_mma_group() : (g > 2 ? 4 : (g == 2 ? 2 : 1))) : (g >= 8 ? 8 : (g > 4 && use_mqk ? 8 : (g > 2 ? 4 : (g == 2 ? 2 : 1))));
    p.gchunks = (int)cdiv(g, G);
    if (p.num_splits <= 0) {
        const int64_t waves = (int64_t)p.b * p.h_k * p.gchunks;
        const int cap = (!fp8 && G >= 8 && D >= 128 && !use_mqk) ? 4 : 8;   // wavefronts per CU to split for: the 512-register dot2 variant keeps 4 resident, all others 8+
        const int wpc_opt = decode_options().waves_per_cu;
        const int wpc = wpc_opt > 0 ? wpc_opt : cap;
        p.num_splits = decode_num_splits(waves, p.seqlen_k, wpc, std::max(1, decode_options().min_tiles.load()));
    }
    p.group_tile = G;
    // The balanced line also for uniform batches that are resident at once (option decode_stream: 0 = no balanced mode at all, 1 = ragged
    // batches and, where measured faster, uniform ones, 2 = always, 3 = ragged batches only).  The line is kv-head major: the wavefronts an
    // XCD receives back to back are the SAME head of 32 sequences, so the 8 wavefronts that land on one CU are the 8 kv heads of ONE
    // sequence and walk the same token rows together (with the kv head fastest, a CU holds 8 unrelated sequences of one head).  Measured
    // (tools/probes/fp8_modes.sh, stream_force_ab.sh, bench.py): the headline workload 6.12 -> 6.41 TB/s (0.77 -> 0.81 of HBM); fp8 C2a
    // 0.400 -> 0.347 ms; the bf16 matrix-core kernel on the 70B shape 0.746 -> 0.677 ms; MHA (32 kv heads) level; d = 64 loses 8 %
    // (0.385 -> 0.42 ms), so d = 64 keeps the old order.  Neither rotating the kv head by the sequence index nor dealing the sequences
    // of a block of 8 to the 8 XCDs (all heads of a sequence behind one L2) reproduces it: it is the CU, not the XCD, that matters.
    const int st_opt = decode_options().stream;
    p.stream_force = st_opt == 2 ? 2 : ((st_opt == 1 && p.h_k > 1 && D == 128) ? 1 : 0);   // 2: the cut line whatever the lengths; 1: the line for exactly uniform batches (the order), cuts from 15 % idle share on
    p.fp8_klines = decode_options().fp8_klines;
    p.head_major = decode_options().head_major != 0 && p.h_k > 1;
    p.stream_waves = 0;
    const int64_t hk_chunks = (int64_t)p.h_k * p.gchunks, max_tiles = cdiv(p.seqlen_k, 16);
    // (Round 5 re-measured round 3's question with the in-launch merge in place: small batches -- the pieces the split heuristic cuts -- laid on the
    // line and merged by their last wavefront instead of split + combine.  Slower on every shape: the 70B TP = 8 shard at B = 64 30.5 -> 42.2 us,
    // B = 1 12.5 -> 20.2, 16 x 8192 90.2 -> 99.0, the 4-head shard at B = 256 83.9 -> 99.7 (profiles/r05_decode_line_small_ab.txt): the pieces of
    // one sequence sit on consecutive line positions = 8 different XCDs, and one wavefront merges 16-32 pieces through three dependent trips.)
    if (p.num_splits == 1 && st_opt != 0 && (p.cu_seqlens_k || p.seqused_k) && p.b <= DECODE_STREAM_MAX_B &&
        p.b * hk_chunks * max_tiles < (int64_t)1 << 31) {
        // enough wavefronts without splitting and the lengths are on the device: the kernel balances ragged batches itself
        p.stream_waves = (int)std::min<int64_t>(p.b * hk_chunks, (int64_t)device_num_cus() * DECODE_STREAM_MAX_WAVES_PER_CU);   // upper bound, set per kernel at launch
    }
    // bit 3 of decode_mqk: groups of 2..4 q heads take the matrix-core kernel when the launch runs on the balanced line.  In the
    // per-sequence order the two kernels tied (or dot2 won by 2-3 %); on the line the matrix-core kernel's shorter instruction stream
    // shows: headline 6.35 -> 6.6 TB/s, C2a 0.703 -> 0.662 ms, ragged 0.567 -> 0.535, B = 256 x 1024 0.190 -> 0.175; split-KV launches
    // (+3 %) and MHA (+17 %) keep dot2 (tools/probes/mqk_ab.sh).  Same G, same scratch: only the kernel changes.
    bool mqk_line = false;
    // (only when uniform batches take the line too -- stream_force: in the per-sequence order the matrix-core kernel's half-line K
    // requests cost 6 %: 0.744 against 0.702 ms)
    if (!fp8 && !use_mqk && D == 128 && g >= 2 && g <= 4 && p.stream_waves > 0 && p.stream_force && (mqk_opt & 8)) mqk_line = true;
    // bit 4: ... and on split-KV launches (since P.V moved to the matrix cores: 16 x 8192 tokens 0.0958 -> 0.0893 ms)
    if (!fp8 && !use_mqk && D == 128 && g >= 2 && g <= 4 && p.num_splits > 1 && (mqk_opt & 16)) mqk_line = true;
    DecodeLaunchPlan lp{G, use_mqk || mqk_line, 0, 0};
    if (p.num_splits > 1 || p.stream_waves > 0) {
        lp.rows = p.stream_waves > 0 ? (size_t)p.stream_waves * 2 * G : (size_t)p.num_splits * p.b * p.h;
        // plan: [0], [1], then b + 1 prefix entries (decode_run_items writes plan[2 + i] for i = 0..b, the combine kernel reads plan[3 + b])
        lp.bytes = lp.rows * (D + 1) * sizeof(float) + ((size_t)p.b + 3) * sizeof(int);
    }
    return lp;
}

static bool decode_pair64_view(const DecodeParams &p, DecodeParams &v);
// Largest scratch any decode call with batch <= max_b, these head counts and contexts <= max_seqlen_k can ask for
// (atoma_warmup reserves it up front, so that a hipGraph capture never meets a growing workspace).
size_t decode_workspace_bound(int max_b, int h, int h_k, int d, int max_seqlen_k) {
    size_t worst = 0;
    if (h_k <= 0 || h % h_k) return 0;
    const int dummy = 0;
    for (int b = 1; b <= max_b; ++b) {
        DecodeParams p{};
        p.b = b; p.h = h; p.h_k = h_k; p.g = h / h_k;
        p.seqlen_k = max_seqlen_k;
        p.cu_seqlens_k = &dummy;   // lengths on the device: the balanced mode is reachable
        p.num_splits = 0;
        worst = std::max(worst, decode_plan_launch(p, d).bytes);
        if (d == 64) {    // the kv-head-pair view of the same call
            DecodeParams v;
            p.k_head_stride = p.v_head_stride = 64;
            if (decode_pair64_view(p, v)) { v.num_splits = 0; worst = std::max(worst, decode_plan_launch(v, 128).bytes); }
        }
        if (d == 128) {   // the same heads over an fp8 cache: up to 16 q heads per wavefront, so more partial rows on the balanced line
            DecodeParams p8 = p;
            p8.num_splits = 0;
            worst = std::max(worst, decode_plan_launch(p8, d, true).bytes);
        }
    }
    return worst;
}

// head_dim 64 as pairs of kv heads on the matrix-core kernel (paged_decode_mqk_item<.., PAIR64>): the view the kernel works on, or false
static bool decode_pair64_view(const DecodeParams &p, DecodeParams &v) {
    if (!decode_options().pair64 || p.k_scale || p.h_k % 2 || !(p.g == 1 || p.g == 2 || p.g == 4) || p.k_head_stride != 64 || p.v_head_stride != 64) return false;
    v = p;
    v.h_k = p.h_k / 2;
    v.g = 2 * p.g;
    v.k_head_stride = v.v_head_stride = 128;
    return true;
}
template <typename T> static void launch_decode_pair64(DecodeParams &v, hipStream_t stream) {
    const DecodeLaunchPlan lp = decode_plan_launch(v, 128);      // the decisions of a 128-wide launch with these head counts (its scratch bound covers the 64-float rows used here)
    if (lp.rows) {
        float *ws = static_cast<float *>(workspace(stream, lp.bytes));
        if (!ws) return;
        v.o_accum = ws;
        v.lse_accum = ws + lp.rows * 64;
        v.plan = reinterpret_cast<int *>(ws + lp.rows * 65);
    }
    switch (lp.G) {
        case 2: launch_decode_mqk<T, 2, true>(v, stream); break;
        case 4: launch_decode_mqk<T, 4, true>(v, stream); break;
        default: launch_decode_mqk<T, 8, true>(v, stream); break;
    }
}

template <typename T, int D>
static void launch_decode_td(DecodeParams &p, hipStream_t stream) {
    if constexpr (D == 64) {
        DecodeParams v;
        if (decode_pair64_view(p, v)) { launch_decode_pair64<T>(v, stream); return; }
    }
    const DecodeLaunchPlan lp = decode_plan_launch(p, D);
    const int G = lp.G;
    const bool use_mqk = lp.use_mqk;
    if (lp.rows) {
        float *ws = static_cast<float *>(workspace(stream, lp.bytes));
        if (!ws) return;
        p.o_accum = ws;
        p.lse_accum = ws + lp.rows * D;
        p.plan = reinterpret_cast<int *>(ws + lp.rows * (D + 1));
    }
    if (D == 128 && use_mqk) {
        switch (G) {
            case 1: launch_decode_mqk<T, 1>(p, stream); break;
            case 2: launch_decode_mqk<T, 2>(p, stream); break;
            case 4: launch_decode_mqk<T, 4>(p, stream); break;
            default: launch_decode_mqk<T, 8>(p, stream); break;
        }
        return;
    }
    switch (G) {
        case 1: launch_decode_tdg<T, D, 1>(p, stream); break;
        case 2: launch_decode_tdg<T, D, 2>(p, stream); break;
        case 4: launch_decode_tdg<T, D, 4>(p, stream); break;
        default: launch_decode_tdg<T, D, 8>(p, stream); break;
    }
}

// fp8 KV cache (d = 128): same planning (splits / balanced mode / scratch) as the 16-bit path, G <= 4 heads per wavefront
// the fp8 kernels live in paged_decode_fp8.hip
void launch_fp8_kernels(const DecodeParams &p, int G, bool is_bf16, bool nt, bool mqk, bool wg8, int64_t blocks, hipStream_t stream);
int decode_fp8_tiles_in_flight();
template <typename T>
static void launch_decode_fp8_g(DecodeParams &p, int G, hipStream_t stream) {
    const int64_t blocks = (int64_t)p.b * p.num_splits * p.h_k * p.gchunks;
    if (p.stream_waves > 0) set_stream_waves(p, 8, stream);   // __launch_bounds__(64, 2)
    const bool nt = decode_options().nt != 0;
    const bool mqk = decode_options().fp8_mqk != 0;
    // 8 wavefronts per workgroup = the 8 kv-head slices (128 B each) of every token row read from one CU (round 2: +20 % on split-KV
    // launches).  Round 3's kv-head-major workgroup order puts the same 8 wavefronts on one CU without tying them into a workgroup and
    // is 4 % faster still (16 x 8192: 0.0503 -> 0.0483 ms), so this is opt-in now (decode_fp8_wg).
    const int wg_opt = decode_options().fp8_wg;
    const bool wg8 = mqk && ((int64_t)p.h_k * p.gchunks) % 8 == 0 && (wg_opt >= 2 || (wg_opt == 1 && p.stream_waves == 0));
    note_decode_kernel(mqk ? "paged_decode_fp8_mma_kernel" : "paged_decode_fp8_kernel", decode_tname<T>(), 128, G, decode_fp8_tiles_in_flight(), nt,
                       p.stream_waves > 0 ? "balanced" : (p.num_splits > 1 ? "KV splits + combine" : (wg8 ? "8 wavefronts per workgroup" : "one wavefront per (sequence, kv head)")));
    launch_fp8_kernels(p, G, std::is_same<T, bf16_t>::value, nt, mqk, wg8, blocks, stream);
    if (!ATOMA_CHECK_LAUNCH("paged_decode_fp8_kernel")) return;
    if (p.num_splits > 1 || (p.stream_waves > 0 && !p.line_merge)) {
        hipLaunchKernelGGL((decode_combine_kernel<T, 128>), dim3((unsigned)std::min<int64_t>((int64_t)p.b * p.h, combine_grid_cap())), dim3(64), 0, stream, p);
        ATOMA_CHECK_LAUNCH("decode_combine_kernel");
    }
}
template <typename T>
static void launch_decode_fp8(DecodeParams &p, hipStream_t stream) {
    const DecodeLaunchPlan lp = decode_plan_launch(p, 128, true);
    if (lp.rows) {
        float *ws = static_cast<float *>(workspace(stream, lp.bytes));
        if (!ws) return;
        p.o_accum = ws;
        p.lse_accum = ws + lp.rows * 128;
        p.plan = reinterpret_cast<int *>(ws + lp.rows * 129);
    }
    launch_decode_fp8_g<T>(p, lp.G, stream);
}
void launch_paged_decode_fp8(DecodeParams &p, bool is_bf16, hipStream_t stream) {
    if (is_bf16) launch_decode_fp8<bf16_t>(p, stream);
    else launch_decode_fp8<f16_t>(p, stream);
}

bool decode_supported(int d) { return d == 64 || d == 128; }
const char *last_decode_kernel() { return g_last_decode_kernel.c_str(); }

void launch_paged_decode(DecodeParams &p, int d, bool is_bf16, hipStream_t stream) {
    if (is_bf16) {
        if (d == 64) launch_decode_td<bf16_t, 64>(p, stream);
        else launch_decode_td<bf16_t, 128>(p, stream);
    } else {
        if (d == 64) launch_decode_td<f16_t, 64>(p, stream);
        else launch_decode_td<f16_t, 128>(p, stream);
    }
}

// AttnParams (run_mha's view) -> DecodeParams.  seqlen_q == 1, so q/o row strides drop out.
void launch_paged_decode_from_attn(const AttnParams &a, bool is_bf16, int num_splits_hint, hipStream_t stream) {
    (void)num_splits_hint;  // the caller's split count is tuned for the reference's CTA shape; see decode_num_splits
    DecodeParams p{};
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o;
    p.lse = a.lse;
    p.block_table = a.block_table;
    p.cu_seqlens_k = a.cu_seqlens_k;
    p.seqused_k = a.seqused_k;
    p.alibi_slopes = a.alibi_slopes;
    p.q_batch_stride = a.q_batch_stride; p.q_head_stride = a.q_head_stride;
    p.o_batch_stride = a.o_batch_stride; p.o_head_stride = a.o_head_stride;
    p.k_batch_stride = a.k_batch_stride; p.k_row_stride = a.k_row_stride; p.k_head_stride = a.k_head_stride;
    p.v_batch_stride = a.v_batch_stride; p.v_row_stride = a.v_row_stride; p.v_head_stride = a.v_head_stride;
    p.block_table_batch_stride = a.block_table_batch_stride;
    p.alibi_batch_stride = a.alibi_batch_stride;
    p.page_size = a.page_size;
    p.b = a.b; p.h = a.h; p.h_k = a.h_k; p.g = a.h / a.h_k;
    p.seqlen_k = a.seqlen_k;
    p.is_seqlens_k_cumulative = a.is_seqlens_k_cumulative;
    p.num_splits = 0;  // 0 = let decode_num_splits choose
    p.scale = a.scale; p.scale_log2 = a.scale_log2;
    launch_paged_decode(p, a.d, is_bf16, stream);
}

}  // namespace atoma

// Paged decode attention (seqlen_q = 1) over an fp8 e4m3fn KV cache [nb, page, h_k, 128] with per-kv-head scales; the
// 16-bit path's semantics otherwise (flash_attn_kv_cache_full, csrc/src/lib.rs:1521-1855: per-sequence lengths, block table,
// empty sequence -> zeros).  Strides in elements (= bytes for the caches).
extern "C" int atoma_paged_decode_fp8(const void *q, const void *k_cache, const void *v_cache, void *o, const float *k_scale, const float *v_scale,
                                      const int32_t *block_table, const int32_t *seqlens_k, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
                                      int64_t head_dim, int64_t block_table_batch_stride, int64_t page_size, int64_t q_batch_stride,
                                      int64_t q_head_stride, int64_t o_batch_stride, int64_t o_head_stride, int64_t cache_block_stride,
                                      int64_t cache_row_stride, int64_t cache_head_stride, float softmax_scale, int dtype, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("paged_decode_fp8: q / o dtype must be f16 or bf16"); return -1; }
    if (head_dim != 128) { set_error("paged_decode_fp8: head_dim must be 128"); return -1; }
    if (batch < 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads) { set_error("paged_decode_fp8: invalid head counts"); return -1; }
    if (page_size <= 0 || page_size % 16) { set_error("paged_decode_fp8: page_size must be a positive multiple of 16"); return -1; }
    if (!q || !k_cache || !v_cache || !o || !k_scale || !v_scale || !block_table || !seqlens_k) { set_error("paged_decode_fp8: null tensor"); return -1; }
    if (q_batch_stride % 8 || q_head_stride % 8 || o_batch_stride % 8 || o_head_stride % 8 || cache_block_stride % 16 || cache_row_stride % 16 || cache_head_stride % 16) {
        set_error("paged_decode_fp8: q / o strides must be multiples of 8 elements, cache strides multiples of 16 bytes");
        return -1;
    }
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache)) & 15u) {
        set_error("paged_decode_fp8: tensors must be 16-byte aligned");
        return -1;
    }
    if (batch == 0) return 0;
    DecodeParams p{};
    p.q = static_cast<const uint16_t *>(q);
    p.k = static_cast<const uint16_t *>(k_cache);
    p.v = static_cast<const uint16_t *>(v_cache);
    p.o = static_cast<uint16_t *>(o);
    p.k_scale = k_scale; p.v_scale = v_scale;
    p.block_table = block_table;
    p.cu_seqlens_k = seqlens_k;
    p.is_seqlens_k_cumulative = 0;
    p.q_batch_stride = q_batch_stride; p.q_head_stride = q_head_stride;
    p.o_batch_stride = o_batch_stride; p.o_head_stride = o_head_stride;
    p.k_batch_stride = p.v_batch_stride = cache_block_stride;
    p.k_row_stride = p.v_row_stride = cache_row_stride;
    p.k_head_stride = p.v_head_stride = cache_head_stride;
    p.block_table_batch_stride = block_table_batch_stride;
    p.page_size = (int)page_size;
    p.b = (int)batch; p.h = (int)num_heads; p.h_k = (int)num_kv_heads; p.g = (int)(num_heads / num_kv_heads);
    p.seqlen_k = (int)(block_table_batch_stride * page_size);
    p.num_splits = 0;
    p.scale = softmax_scale; p.scale_log2 = softmax_scale * 1.4426950408889634f;
    launch_paged_decode_fp8(p, dtype == ATOMA_BF16, static_cast<hipStream_t>(stream));
    return has_error() ? -1 : 0;
}
