// Decode attention over an fp8 (e4m3fn) KV cache: the kernels (own translation unit: paged_decode.hip takes minutes to compile).
// Planning, scratch, the combine kernel and the C entry point are in paged_decode.hip (launch_decode_fp8).
#include "paged_decode.h"

namespace atoma {

// ------------------------------------------------------------------------------------------
// fp8 (OCP e4m3fn) KV cache, d = 128 (/root/reference/README.md:35 roadmap "quantization").  The cache
// keeps the reference's layout [nb, page, h_k, d] with ONE byte per element and a per-kv-head dequantisation scale
// (value = e4m3 * scale[hk]); q and o stay bf16 / f16.  Decode is HBM-bound, so halving the K/V bytes is the lever:
//   * a token row of one kv head is 128 bytes: 8 adjacent lanes read it with one 16-byte load each, a wave instruction
//     covers 8 rows = 8 full 128-byte lines (1 KiB, as in the 16-bit kernel); a 16-token tile is 2 + 2 load instructions;
//   * fp8 -> bf16 is exact: v_cvt_scalef32_pk_bf16_fp8 with scale 1.0 turns two bytes into one packed bf16 pair (8 per
//     16-byte load), after which q.k and P.V are the same v_dot2c streams as in the 16-bit kernel;
//   * the K scale folds into the softmax scale (scores = k_scale * q.k_q), the V scale into the final 1/l -- nothing per element.
// Same work mapping, split-KV / balanced modes and combine kernel as the 16-bit path.  This first kernel (v_dot2c for both products,
// option decode_fp8_mqk = 0) runs groups of more than 4 q heads in chunks of 4 (K/V re-read per chunk); the default is the matrix-core
// kernel further down (paged_decode_fp8_mma_kernel), which takes up to 16 q heads in one pass.
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ uint32_t fp8x2_to_pair(uint32_t word, bool hi);
template <> __device__ __forceinline__ uint32_t fp8x2_to_pair<bf16_t>(uint32_t word, bool hi) {
    return hi ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, true))
              : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, false));
}
template <> __device__ __forceinline__ uint32_t fp8x2_to_pair<f16_t>(uint32_t word, bool hi) {
    return hi ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, 1.0f, true))
              : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, 1.0f, false));
}
// 16 fp8 bytes -> 8 packed 16-bit pairs, element order preserved
template <typename T> __device__ __forceinline__ void fp8x16_to_pairs(const u32x4 &v, uint32_t (&out)[8]) {
    out[0] = fp8x2_to_pair<T>(v.x, false); out[1] = fp8x2_to_pair<T>(v.x, true);
    out[2] = fp8x2_to_pair<T>(v.y, false); out[3] = fp8x2_to_pair<T>(v.y, true);
    out[4] = fp8x2_to_pair<T>(v.z, false); out[5] = fp8x2_to_pair<T>(v.z, true);
    out[6] = fp8x2_to_pair<T>(v.w, false); out[7] = fp8x2_to_pair<T>(v.w, true);
}

template <typename T, int G, int P, bool NT>
__device__ __forceinline__ void paged_decode_fp8_item(const DecodeParams &p, const DecodeWork &wk) {
    constexpr int D = 128;
    constexpr int LPR = 8;         // lanes per 128-byte row
    constexpr int RPI = 8;         // rows per load instruction
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, dc = lane % LPR;   // row of the 8-row slab, 16-element chunk of the row

    const int b = wk.b, hk = wk.hk, gc = wk.gc, L = wk.L, t0 = wk.t0, t1 = wk.t1;
    const bool partial = wk.partial;
    const int hq0 = hk * p.g + gc * G;
    const int nq = min(G, p.g - gc * G);

    const float sl2 = p.scale_log2 * load_ro(p.k_scale + hk);   // scores = k_scale * (q . k_q)
    float m[G], l[G], o[G][16];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        m[gq] = -INFINITY;
        l[gq] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[gq][e] = 0.f;
    }

    if (t0 < t1) {
        uint32_t qv[G][8];   // q[head][16.dc ..+15] as 8 packed pairs, replicated over the 8 row groups
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            uint4 a = make_uint4(0, 0, 0, 0), c = a;
            if (gq < nq) {
                const uint4 *src = reinterpret_cast<const uint4 *>(p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hq0 + gq) * p.q_head_stride + dc * 16);
                a = src[0];
                c = src[1];
            }
            qv[gq][0] = a.x; qv[gq][1] = a.y; qv[gq][2] = a.z; qv[gq][3] = a.w;
            qv[gq][4] = c.x; qv[gq][5] = c.y; qv[gq][6] = c.z; qv[gq][7] = c.w;
        }
        // ---- loader: as paged_decode_item, strides in BYTES (one byte per element) ----
        const uint32_t tpp = (uint32_t)(p.page_size >> 4);
        const uint32_t tpp_magic = tpp > 1 ? (uint32_t)((0x100000000ull + tpp - 1) / tpp) : 0u;
        const int last_pg = (L + p.page_size - 1) / p.page_size - 1;
        const int *bt_row = p.block_table + (int64_t)b * p.block_table_batch_stride;
        const char *kbase = reinterpret_cast<const char *>(p.k) + (int64_t)hk * p.k_head_stride;
        const char *vbase = reinterpret_cast<const char *>(p.v) + (int64_t)hk * p.v_head_stride;
        const int64_t k_row_bytes = p.k_row_stride, v_row_bytes = p.v_row_stride;
        const int64_t k_page_bytes = p.k_batch_stride, v_page_bytes = p.v_batch_stride;
        const uint32_t k_lane_off = (uint32_t)(sub * k_row_bytes + dc * 16);
        const uint32_t v_lane_off = (uint32_t)(sub * v_row_bytes + dc * 16);
        auto page_of = [&](int tile, uint32_t &tip) -> int {
            if (tpp == 1) { tip = 0; return tile; }
            const uint32_t pg = __umulhi((uint32_t)tile, tpp_magic);
            tip = (uint32_t)tile - pg * tpp;
            return (int)pg;
        };
        auto fetch_pid = [&](int tile) -> int {
            uint32_t tip;
            const int pg = min(page_of(tile, tip), last_pg);
            return load_ro(bt_row + pg);
        };
        constexpr int AUX = NT ? 2 : 0;
        auto issue = [&](u32x4 (&kb)[2], u32x4 (&vb)[2], int tile, int pid) {   // paged tiles always own their 16 rows
            uint32_t tip;
            (void)page_of(tile, tip);
            const char *kt = kbase + (int64_t)pid * k_page_bytes + (int64_t)(tip << 4) * k_row_bytes;
            const char *vt = vbase + (int64_t)pid * v_page_bytes + (int64_t)(tip << 4) * v_row_bytes;
            const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(kt), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vt), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int r = 0; r < 2; ++r) kb[r] = __builtin_amdgcn_raw_buffer_load_b128(kr, k_lane_off, (int)(r * RPI * k_row_bytes), AUX);
#pragma unroll
            for (int r = 0; r < 2; ++r) vb[r] = __builtin_amdgcn_raw_buffer_load_b128(vr, v_lane_off, (int)(r * RPI * v_row_bytes), AUX);
        };
        auto compute = [&](const u32x4 (&kb)[2], const u32x4 (&vb)[2], int tile) {
            float s[2][G];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                uint32_t kk[8];
                fp8x16_to_pairs<T>(kb[r], kk);
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) a = dot2<T>(kk[i], qv[gq][i], a);
                    s[r][gq] = row_allreduce<LPR>(a) * sl2;   // log2 domain
                }
            }
            const int tok0 = (tile << 4) + sub;
            if ((tile << 4) + 16 > L) {   // wave-uniform: ragged last tile
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (tok0 + r * RPI >= L) {
#pragma unroll
                        for (int gq = 0; gq < G; ++gq) s[r][gq] = -INFINITY;
                    }
            }
            float mnew[G];
            bool changed = false;
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
                mnew[gq] = fmaxf(m[gq], fmaxf(s[0][gq], s[1][gq]));
                changed |= mnew[gq] > m[gq];
            }
            if (__any(changed)) {
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    const float ms = mnew[gq] == -INFINITY ? 0.f : mnew[gq];
                    const float alpha = __builtin_amdgcn_exp2f(m[gq] - ms);
                    l[gq] *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[gq][e] *= alpha;
                    m[gq] = mnew[gq];
                }
            }
            uint32_t pp[G];
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
                const float ms = m[gq] == -INFINITY ? 0.f : m[gq];
                const float p0 = __builtin_amdgcn_exp2f(s[0][gq] - ms), p1 = __builtin_amdgcn_exp2f(s[1][gq] - ms);
                l[gq] += p0 + p1;
                pp[gq] = pack_pair<T>(p0, p1);   // tokens (sub, 8 + sub)
            }
            uint32_t va[8], vc[8];
            fp8x16_to_pairs<T>(vb[0], va);
            fp8x16_to_pairs<T>(vb[1], vc);
            if ((tile << 4) + 16 > L) {   // wave-uniform: slots of the last page behind the sequence were never written -- whatever they hold
                                          // (NaN codes 0x7f / 0xff included) must not meet p = 0 in the products (flash_fwd_kernel.h:903 zero-fills them)
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    va[w] = tok0 >= L ? 0u : va[w];
                    vc[w] = tok0 + RPI >= L ? 0u : vc[w];
                }
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const uint32_t lo = __builtin_amdgcn_perm(vc[w], va[w], 0x05040100u);  // (row sub, row 8 + sub) of element 2w
                const uint32_t hi = __builtin_amdgcn_perm(vc[w], va[w], 0x07060302u);  // ... of element 2w + 1
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    o[gq][2 * w] = dot2<T>(lo, pp[gq], o[gq][2 * w]);
                    o[gq][2 * w + 1] = dot2<T>(hi, pp[gq], o[gq][2 * w + 1]);
                }
            }
        };
        // ---- software pipeline: P tiles in flight (every paged tile is complete: one code path, unconditional steady state) ----
        u32x4 kb[P][2], vb[P][2];
        int pid[P];
        int t = t0;
        if (t0 + 2 * P <= t1) {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                issue(kb[s], vb[s], t0 + s, pid[s]);
                pid[s] = fetch_pid(t0 + s + P);
            }
            for (; t + 2 * P <= t1; t += P) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    compute(kb[s], vb[s], t + s);
                    issue(kb[s], vb[s], t + s + P, pid[s]);
                    pid[s] = fetch_pid(t + s + 2 * P);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s)
                if (t0 + s < t1) {
                    issue(kb[s], vb[s], t0 + s, pid[s]);
                    pid[s] = fetch_pid(t0 + s + P);
                }
        }
        for (; t < t1; t += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                if (t + s < t1) {
                    compute(kb[s], vb[s], t + s);
                    if (t + s + P < t1) {
                        issue(kb[s], vb[s], t + s + P, pid[s]);
                        pid[s] = fetch_pid(t + s + 2 * P);
                    }
                }
            }
        }
    }

    // ---- merge the 8 row groups ----
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
        for (int e = 0; e < 16; ++e) {
            float x = o[gq][e] * w;
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
            o[gq][e] = x;
        }
        m[gq] = mt;
        l[gq] = lt;
    }
    if (sub != 0) return;
    const float vs = load_ro(p.v_scale + hk);
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        if (gq >= nq) continue;
        const int hq = hq0 + gq;
        const bool empty = !(l[gq] > 0.f);
        const float inv = empty ? 0.f : vs / l[gq];      // O = v_scale * sum(p v_q) / sum(p)
        const float lse = empty ? INFINITY : (m[gq] + __builtin_amdgcn_logf(l[gq])) * 0.6931471805599453f;
        if (!partial) {
            uint4 w4[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                w4[hlf].x = pack2<T>(o[gq][8 * hlf + 0] * inv, o[gq][8 * hlf + 1] * inv);
                w4[hlf].y = pack2<T>(o[gq][8 * hlf + 2] * inv, o[gq][8 * hlf + 3] * inv);
                w4[hlf].z = pack2<T>(o[gq][8 * hlf + 4] * inv, o[gq][8 * hlf + 5] * inv);
                w4[hlf].w = pack2<T>(o[gq][8 * hlf + 6] * inv, o[gq][8 * hlf + 7] * inv);
            }
            uint4 *dst = reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + dc * 16);
            dst[0] = w4[0];
            dst[1] = w4[1];
            if (p.lse && dc == 0) p.lse[(int64_t)b * p.h + hq] = lse;
        } else {
            const int64_t row = wk.prow + gq;
            float *dst = p.o_accum + row * D + dc * 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                partial_store(dst + 4 * q4, o[gq][4 * q4] * inv, o[gq][4 * q4 + 1] * inv, o[gq][4 * q4 + 2] * inv, o[gq][4 * q4 + 3] * inv);
            if (dc == 0) partial_store(p.lse_accum + row, empty ? -INFINITY : lse);
        }
    }
}

template <typename T, int G, int P, bool NT, bool STREAM>
__global__ void __launch_bounds__(64, 2) paged_decode_fp8_kernel(const DecodeParams p) {
    decode_run_items<STREAM>(p, [](const DecodeParams &pp, const DecodeWork &wk) { paged_decode_fp8_item<T, G, P, NT>(pp, wk); }, DecodeLineMerge<T, 128>{});
}

// fp8 KV cache with BOTH products on the matrix cores (round 3; the v_dot2c variant above stays behind option decode_fp8_mqk = 0).
// What the counters said (profiles/r03_fp8_decode_counters.json): the first matrix-core version (q.K^T only) asked the L2 for every K line
// twice -- the MFMA operand layout gives a token 4 lanes x 16 B = 64 B per instruction -- and a CU's request slots, not bytes, are what
// runs out (0.098 requests per CU and cycle, the same as the bf16 kernel at 0.76 of HBM); and with that fixed the kernel is bound by its
// own instruction stream (188 VALU instructions per 4 KiB tile at two wavefronts per SIMD), half of it the P.V dot2 pairs.  So:
//   * K in FULL 128-byte lines: instruction jj fetches tokens 8 jj + (col & 7), the lane's 16 bytes are chunk grp + 4 (col >> 3) of the
//     row (8 rows x 128 B per wave instruction).  MFMA row r < 8 then holds the d < 64 half of token 8 jj + r and row r + 8 the other
//     half of the SAME token: the product with the first half of Q^T is right in rows 0..7 (lanes 0..31), with the second half in rows
//     8..15 (lanes 32..63); the halves that sit in the wrong lanes change sides with one v_permlane32_swap per register and enter the
//     other product as its C operand.  Result as before: S^T[token 4 grp + i][head col].  (On the balanced line the request format
//     makes no difference -- timing probe 0.65 either way -- so its pieces keep the half-line fetch, 4 MFMAs and 8 VALU fewer.)
//   * P.V as O[head][d] = P[head][token] . V[token][d] with v_mfma_f32_16x16x16: the A operand of lane (grp, col) is P[head col][tokens
//     4 grp ..+3] -- exactly the four probabilities the lane has just computed, packed; the B operand is V[tokens 4 grp ..+3][one d]:
//     the lane loads 8 bytes (d = 8 col ..+7) of each of its 4 token rows (a wave instruction = 4 rows x 128 B), converts them to pairs
//     along d and re-pairs them along the tokens with v_perm (16 + 16 instructions instead of 16 + 16 + 64 v_dot2c at 4 heads);
//     MFMA n = 0..7 accumulates O[head 4 grp + i][d = 8 col + n].  All 16 tokens of a tile enter one accumulator, so the running max
//     of a head is common to its four lane groups (v_permlane16_swap + v_permlane32_swap per tile), the row sums stay per lane group
//     and are added once at the end; the rare rescale fetches the heads' factors with ds_bpermute.
//   * Any group size up to 16 q heads runs in ONE pass with the same 32 accumulator registers (the dot2 layouts needed 16 per head:
//     groups of 8 took two passes over K / V).
#ifndef ATOMA_FP8_KLINES
#define ATOMA_FP8_KLINES 1
#endif
constexpr bool FP8_KLINES = ATOMA_FP8_KLINES != 0;    // -DATOMA_FP8_KLINES=0: the half-line K fetch everywhere, kept for A/B
constexpr int FP8_MMA_G = 16;                         // q heads per wavefront of the matrix-core kernel (p.group_tile)

template <typename T, int P, bool NT>
__device__ __forceinline__ void paged_decode_fp8_mma_item(const DecodeParams &p, const DecodeWork &wk) {
    constexpr int D = 128, G = FP8_MMA_G;
    const int lane = threadIdx.x & 63, grp = lane >> 4, col = lane & 15;
    const int b = wk.b, hk = wk.hk, gc = wk.gc, L = wk.L, t0 = wk.t0, t1 = wk.t1;
    const bool partial = wk.partial;
    const int hq0 = hk * p.g + gc * G;
    const int nq = min(G, p.g - gc * G);
    const float sl2 = p.scale_log2 * load_ro(p.k_scale + hk);
    float m = -INFINITY, l = 0.f;        // head `col`: running max over ALL tokens seen, sum over this lane group's 4 tokens per tile
    f32x4_v o[8];                        // o[n][i] = O[head 4 grp + i][d = 8 col + n]
#pragma unroll
    for (int n = 0; n < 8; ++n) o[n] = f32x4_v{0.f, 0.f, 0.f, 0.f};
    // lane (grp, 4 grp + i) of this 16-lane row holds the state of head 4 grp + i
    const int head_lane4 = ((lane & 48) + 4 * grp) << 2;     // ds_bpermute byte address of i = 0

    if (t0 < t1) {
        u32x4 qb[4];                     // Q^T operand of k-step s = 2j + u: q[head col][64 j + 16 grp + 8 u ..+7]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            qb[s4] = u32x4{0, 0, 0, 0};
            if (col < nq)
                qb[s4] = *reinterpret_cast<const u32x4 *>(p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hq0 + col) * p.q_head_stride +
                                                          64 * (s4 >> 1) + 16 * grp + 8 * (s4 & 1));
        }
        const uint32_t tpp = (uint32_t)(p.page_size >> 4);
        const uint32_t tpp_magic = tpp > 1 ? (uint32_t)((0x100000000ull + tpp - 1) / tpp) : 0u;
        const int last_pg = (L + p.page_size - 1) / p.page_size - 1;
        const int *bt_row = p.block_table + (int64_t)b * p.block_table_batch_stride;
        const char *kbase = reinterpret_cast<const char *>(p.k) + (int64_t)hk * p.k_head_stride;
        const char *vbase = reinterpret_cast<const char *>(p.v) + (int64_t)hk * p.v_head_stride;
        const int64_t k_row_bytes = p.k_row_stride, v_row_bytes = p.v_row_stride;
        const int64_t k_page_bytes = p.k_batch_stride, v_page_bytes = p.v_batch_stride;
        const bool klines = FP8_KLINES && (p.fp8_klines >= 2 || (p.fp8_klines == 1 && !wk.balanced && p.h_k > 1));
        const uint32_t k_lane_off = klines ? (uint32_t)((col & 7) * k_row_bytes + (grp + 4 * (col >> 3)) * 16)   // + 8 jj rows
                                           : (uint32_t)(col * k_row_bytes + grp * 16);                          // + 64 j
        const int k_step = klines ? (int)(8 * k_row_bytes) : 64;
        const uint32_t v_lane_off = (uint32_t)(4 * grp * v_row_bytes + col * 8);                                 // + i rows
        auto page_of = [&](int tile, uint32_t &tip) -> int {
            if (tpp == 1) { tip = 0; return tile; }
            const uint32_t pg = __umulhi((uint32_t)tile, tpp_magic);
            tip = (uint32_t)tile - pg * tpp;
            return (int)pg;
        };
        auto fetch_pid = [&](int tile) -> int {
            uint32_t tip;
            const int pg = min(page_of(tile, tip), last_pg);
            return load_ro(bt_row + pg);
        };
        constexpr int AUX = NT ? 2 : 0;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        auto issue = [&](u32x4 (&kb)[2], u32x2 (&vb)[4], int tile, int pid) {
            uint32_t tip;
            (void)page_of(tile, tip);
            const char *kt = kbase + (int64_t)pid * k_page_bytes + (int64_t)(tip << 4) * k_row_bytes;
            const char *vt = vbase + (int64_t)pid * v_page_bytes + (int64_t)(tip << 4) * v_row_bytes;
            const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(kt), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vt), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) kb[j] = __builtin_amdgcn_raw_buffer_load_b128(kr, k_lane_off, j * k_step, AUX);
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[i] = __builtin_amdgcn_raw_buffer_load_b64(vr, v_lane_off, (int)(i * v_row_bytes), AUX);
        };
        auto compute = [&](const u32x4 (&kb)[2], const u32x2 (&vb)[4], int tile) {
            f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
            uint32_t k0[8], k1[8];
            fp8x16_to_pairs<T>(kb[0], k0);
            fp8x16_to_pairs<T>(kb[1], k1);
            if (klines) {
                const f32x4_v zero = {0.f, 0.f, 0.f, 0.f};
                f32x4_v hi0 = mfma16<T>(u32x4{k0[0], k0[1], k0[2], k0[3]}, qb[2], zero);     // high lanes: tokens 0..7, d >= 64
                f32x4_v lo1 = mfma16<T>(u32x4{k1[0], k1[1], k1[2], k1[3]}, qb[0], zero);     // low lanes: tokens 8..15, d < 64
                hi0 = mfma16<T>(u32x4{k0[4], k0[5], k0[6], k0[7]}, qb[3], hi0);
                lo1 = mfma16<T>(u32x4{k1[4], k1[5], k1[6], k1[7]}, qb[1], lo1);
                f32x4_v c_lo, c_hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
                    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(hi0[i]), __float_as_uint(lo1[i]), false, false);
                    c_hi[i] = __uint_as_float(r[0]);     // high lanes: lo1 of lane - 32
                    c_lo[i] = __uint_as_float(r[1]);     // low lanes: hi0 of lane + 32
                }
                f32x4_v lo0 = mfma16<T>(u32x4{k0[0], k0[1], k0[2], k0[3]}, qb[0], c_lo);     // low lanes: tokens 0..7 complete
                f32x4_v hi1 = mfma16<T>(u32x4{k1[0], k1[1], k1[2], k1[3]}, qb[2], c_hi);     // high lanes: tokens 8..15 complete
                lo0 = mfma16<T>(u32x4{k0[4], k0[5], k0[6], k0[7]}, qb[1], lo0);
                hi1 = mfma16<T>(u32x4{k1[4], k1[5], k1[6], k1[7]}, qb[3], hi1);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = lane < 32 ? lo0[i] : hi1[i];
            } else {
                acc = mfma16<T>(u32x4{k0[0], k0[1], k0[2], k0[3]}, qb[0], acc);
                acc = mfma16<T>(u32x4{k0[4], k0[5], k0[6], k0[7]}, qb[1], acc);
                acc = mfma16<T>(u32x4{k1[0], k1[1], k1[2], k1[3]}, qb[2], acc);
                acc = mfma16<T>(u32x4{k1[4], k1[5], k1[6], k1[7]}, qb[3], acc);
            }
            float s[4];
            const int tok0 = (tile << 4) + 4 * grp;
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = acc[i] * sl2;
            if ((tile << 4) + 16 > L) {  // wave-uniform: ragged last tile
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (tok0 + i >= L) s[i] = -INFINITY;
            }
            const float mnew = fmaxf(col_max4(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]))), m);
            if (__any(mnew > m)) {
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
            const uint32_t pk01 = pack_pair<T>(pr[0], pr[1]), pk23 = pack_pair<T>(pr[2], pr[3]);   // A operand: P[head col][tokens 4 grp ..+3]
            uint32_t c[4][4];            // c[i][q]: token 4 grp + i, the pair d = 8 col + 2q, + 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i][0] = fp8x2_to_pair<T>(vb[i].x, false); c[i][1] = fp8x2_to_pair<T>(vb[i].x, true);
                c[i][2] = fp8x2_to_pair<T>(vb[i].y, false); c[i][3] = fp8x2_to_pair<T>(vb[i].y, true);
            }
            if ((tile << 4) + 16 > L) {  // wave-uniform: never-written slots behind the sequence (NaN codes included) must not meet p = 0
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (tok0 + i >= L) c[i][0] = c[i][1] = c[i][2] = c[i][3] = 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t e01 = __builtin_amdgcn_perm(c[1][q], c[0][q], 0x05040100u), e23 = __builtin_amdgcn_perm(c[3][q], c[2][q], 0x05040100u);
                const uint32_t o01 = __builtin_amdgcn_perm(c[1][q], c[0][q], 0x07060302u), o23 = __builtin_amdgcn_perm(c[3][q], c[2][q], 0x07060302u);
                o[2 * q] = mfma16k16<T>(pk01, pk23, e01, e23, o[2 * q]);
                o[2 * q + 1] = mfma16k16<T>(pk01, pk23, o01, o23, o[2 * q + 1]);
            }
        };
        u32x4 kb[P][2];
        u32x2 vb[P][4];
        int pid[P];
        int t = t0;
        if (t0 + 2 * P <= t1) {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                issue(kb[s], vb[s], t0 + s, pid[s]);
                pid[s] = fetch_pid(t0 + s + P);
            }
            for (; t + 2 * P <= t1; t += P) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    compute(kb[s], vb[s], t + s);
                    issue(kb[s], vb[s], t + s + P, pid[s]);
                    pid[s] = fetch_pid(t + s + 2 * P);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < P; ++s) pid[s] = fetch_pid(t0 + s);
#pragma unroll
            for (int s = 0; s < P; ++s)
                if (t0 + s < t1) {
                    issue(kb[s], vb[s], t0 + s, pid[s]);
                    pid[s] = fetch_pid(t0 + s + P);
                }
        }
        for (; t < t1; t += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                if (t + s < t1) {
                    compute(kb[s], vb[s], t + s);
                    if (t + s + P < t1) {
                        issue(kb[s], vb[s], t + s + P, pid[s]);
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
    const float vs = load_ro(p.v_scale + hk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lh = __uint_as_float(__builtin_amdgcn_ds_bpermute(head_lane4 + 4 * i, __float_as_uint(lt)));
        const float mh = __uint_as_float(__builtin_amdgcn_ds_bpermute(head_lane4 + 4 * i, __float_as_uint(m)));
        const int h = 4 * grp + i;
        if (h >= nq) continue;
        const int hq = hq0 + h;
        const bool empty = !(lh > 0.f);
        const float inv = empty ? 0.f : vs / lh;
        const float lse = empty ? INFINITY : (mh + __builtin_amdgcn_logf(lh)) * 0.6931471805599453f;
        if (!partial) {
            uint4 w4;
            w4.x = pack2<T>(o[0][i] * inv, o[1][i] * inv);
            w4.y = pack2<T>(o[2][i] * inv, o[3][i] * inv);
            w4.z = pack2<T>(o[4][i] * inv, o[5][i] * inv);
            w4.w = pack2<T>(o[6][i] * inv, o[7][i] * inv);
            *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)hq * p.o_head_stride + col * 8) = w4;
            if (p.lse && col == 0) p.lse[(int64_t)b * p.h + hq] = lse;
        } else {
            const int64_t row = wk.prow + h;
            float *dst = p.o_accum + row * D + col * 8;
            partial_store(dst, o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv);
            partial_store(dst + 4, o[4][i] * inv, o[5][i] * inv, o[6][i] * inv, o[7][i] * inv);
            if (col == 0) partial_store(p.lse_accum + row, empty ? -INFINITY : lse);
        }
    }
}

template <typename T, int P, bool NT, bool STREAM, int NWG = 1>
__global__ void __launch_bounds__(64 * NWG, 2) paged_decode_fp8_mma_kernel(const DecodeParams p) {
    decode_run_items<STREAM, NWG>(p, [](const DecodeParams &pp, const DecodeWork &wk) { paged_decode_fp8_mma_item<T, P, NT>(pp, wk); }, DecodeLineMerge<T, 128>{});
}

// ------------------------------------------------------------------------------------------
// host side: the launches only (planning, scratch and the combine kernel: launch_decode_fp8 in paged_decode.hip)
// ------------------------------------------------------------------------------------------
#ifndef ATOMA_FP8_P
#define ATOMA_FP8_P 4      // 16-token tiles (4 KiB) in flight per wavefront (3: ragged batches -8 %; 5, 6, 8: level); -DATOMA_FP8_P=.. builds are probe variants (make fp8p)
#endif
int decode_fp8_tiles_in_flight() { return ATOMA_FP8_P; }
int decode_fp8_mma_group() { return FP8_MMA_G; }

template <typename T, int G, bool NT, bool STREAM>
static void launch_fp8_dot2(const DecodeParams &p, int64_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL((paged_decode_fp8_kernel<T, G, ATOMA_FP8_P, NT, STREAM>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
}
template <typename T, bool NT, bool STREAM>
static void launch_fp8_mma(const DecodeParams &p, bool wg8, int64_t blocks, hipStream_t stream) {
    if (wg8) hipLaunchKernelGGL((paged_decode_fp8_mma_kernel<T, ATOMA_FP8_P, NT, STREAM, 8>), dim3((unsigned)cdiv(blocks, 8)), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((paged_decode_fp8_mma_kernel<T, ATOMA_FP8_P, NT, STREAM>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
}
template <typename T, bool NT, bool STREAM>
static void launch_fp8_tns(const DecodeParams &p, int G, bool mqk, bool wg8, int64_t blocks, hipStream_t stream) {
    if (mqk) return launch_fp8_mma<T, NT, STREAM>(p, wg8, blocks, stream);
    switch (G) {
        case 1: launch_fp8_dot2<T, 1, NT, STREAM>(p, blocks, stream); break;
        case 2: launch_fp8_dot2<T, 2, NT, STREAM>(p, blocks, stream); break;
        default: launch_fp8_dot2<T, 4, NT, STREAM>(p, blocks, stream); break;
    }
}
template <typename T>
static void launch_fp8_t(const DecodeParams &p, int G, bool nt, bool mqk, bool wg8, int64_t blocks, hipStream_t stream) {
    const bool balanced = p.stream_waves > 0;
    if (nt) { if (balanced) launch_fp8_tns<T, true, true>(p, G, mqk, wg8, blocks, stream); else launch_fp8_tns<T, true, false>(p, G, mqk, wg8, blocks, stream); }
    else { if (balanced) launch_fp8_tns<T, false, true>(p, G, mqk, wg8, blocks, stream); else launch_fp8_tns<T, false, false>(p, G, mqk, wg8, blocks, stream); }
}
void launch_fp8_kernels(const DecodeParams &p, int G, bool is_bf16, bool nt, bool mqk, bool wg8, int64_t blocks, hipStream_t stream) {
    if (is_bf16) launch_fp8_t<bf16_t>(p, G, nt, mqk, wg8, blocks, stream);
    else launch_fp8_t<f16_t>(p, G, nt, mqk, wg8, blocks, stream);
}

}  // namespace atoma
