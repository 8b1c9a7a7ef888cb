// Shape-generic attention forward for gfx950: any head_dim <= 256 that is a multiple of 8,
// dense / varlen / paged K-V, causal, ALiBi, GQA.  One wavefront per (batch, q head, query row).
//
// This is the library's coverage kernel, not its fast path: run_mha routes here only the
// shapes the tuned kernels (paged_decode.hip, prefill_mfma.hip) do not take -- the head sizes
// the reference instantiates besides 64/128 (/root/reference/csrc/build.rs:7-74: 32..256 step 32),
// ALiBi slopes (Llama passes none, backends/vllm/src/models/llama.rs:125-127), and the tiny
// d=8 tensors of the reference's own golden tests (csrc/tests/flash_attn_tests.rs:31-93).
// Same numerics contract as the reference kernel: exp2-domain softmax, fp32 accumulation,
// P rounded to the storage dtype before P.V (csrc/kernels/softmax.h:65-185).
#include "attn_params.h"
#include <atomic>
#include <cstring>
#include <string>
#include <type_traits>
#include <stdlib.h>

namespace atoma {

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

template <typename T>
__global__ void __launch_bounds__(64) attn_generic_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t q_sm[128];  // one q row, packed pairs
    __shared__ float p_sm[64];
    const int lane = threadIdx.x;
    const int mrow = blockIdx.x, hq = blockIdx.y, b = blockIdx.z;
    const SeqInfo si(p, b);
    if (mrow >= si.len_q) return;
    const int D = p.d;
    const int hk = hq / (p.h / p.h_k);
    const uint16_t *qrow = p.q + si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)mrow * p.q_row_stride +
                           (int64_t)hq * p.q_head_stride;
    uint16_t *orow = p.o + si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)mrow * p.o_row_stride +
                     (int64_t)hq * p.o_head_stride;
    const int shift = si.len_k - si.len_q;  // mask.h:170
    int hi = p.is_causal ? min(si.len_k, mrow + shift + 1) : si.len_k;
    if (hi < 0) hi = 0;

    for (int i = lane; i < D / 2; i += 64) q_sm[i] = reinterpret_cast<const uint32_t *>(qrow)[i];
    __syncthreads();

    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int64_t koff = paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b);
    const int64_t voff = paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b);
    const float slope_l2 = p.alibi_slopes ? p.alibi_slopes[b * p.alibi_batch_stride + hq] * 1.4426950408889634f : 0.f;

    float o[4] = {0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l = 0.f;
    const bool owns = lane * 4 < D;
    for (int j0 = 0; j0 < hi; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < hi) {
            const uint16_t *krow = paged ? p.k + (int64_t)bt[j / p.page_size] * p.k_batch_stride +
                                               (int64_t)(j % p.page_size) * p.k_row_stride
                                         : p.k + koff + (int64_t)j * p.k_row_stride;
            krow += (int64_t)hk * p.k_head_stride;
            float acc = 0.f;
            for (int c = 0; c < D / 8; ++c) {
                const uint4 kk = *reinterpret_cast<const uint4 *>(krow + c * 8);
                const uint4 qq = *reinterpret_cast<const uint4 *>(&q_sm[c * 4]);
                acc = dot2<T>(kk.x, qq.x, acc);
                acc = dot2<T>(kk.y, qq.y, acc);
                acc = dot2<T>(kk.z, qq.z, acc);
                acc = dot2<T>(kk.w, qq.w, acc);
            }
            s = acc * p.scale_log2;
            if (p.alibi_slopes) s -= slope_l2 * fabsf((float)(mrow + shift - j));  // mask.h:179-186
        }
        const float m_new = fmaxf(m_run, wave_max(s));
        const float ms = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - ms);
        const float pj = j < hi ? __builtin_amdgcn_exp2f(s - ms) : 0.f;
        l = l * alpha + wave_sum(pj);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= alpha;
        m_run = m_new;
        p_sm[lane] = round_through<T>(pj);
        __syncthreads();
        const int n = min(64, hi - j0);
        for (int jj = 0; jj < n; ++jj) {
            const int t = j0 + jj;
            const float pv = p_sm[jj];
            const uint16_t *vrow = paged ? p.v + (int64_t)bt[t / p.page_size] * p.v_batch_stride +
                                               (int64_t)(t % p.page_size) * p.v_row_stride
                                         : p.v + voff + (int64_t)t * p.v_row_stride;
            vrow += (int64_t)hk * p.v_head_stride;
            if (owns) {
                const uint2 vv = *reinterpret_cast<const uint2 *>(vrow + lane * 4);
                o[0] += pv * lo_to_f32<T>(vv.x);
                o[1] += pv * hi_to_f32<T>(vv.x);
                o[2] += pv * lo_to_f32<T>(vv.y);
                o[3] += pv * hi_to_f32<T>(vv.y);
            }
        }
        __syncthreads();
    }
    const bool empty = !(l > 0.f);  // no visible key: O = 0, LSE = +inf (flash_fwd_kernel.h:97-133)
    const float inv = empty ? 0.f : 1.f / l;
    if (owns) {
        uint2 w;
        w.x = pack2<T>(o[0] * inv, o[1] * inv);
        w.y = pack2<T>(o[2] * inv, o[3] * inv);
        *reinterpret_cast<uint2 *>(orow + lane * 4) = w;
    }
    if (p.lse && lane == 0) {
        const float lse = empty ? INFINITY : (m_run + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
        // varlen: [h, total_q] with total_q = cu_seqlens_q[b] (run_mha is not told total_q: SURVEY B/Q4)
        if (p.unpadded_lse && p.cu_seqlens_q) p.lse[(int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q + mrow] = lse;
        else p.lse[((int64_t)b * p.h + hq) * p.seqlen_q + mrow] = lse;
    }
}

// ------------------------------------------------------------------------------------------
// Decode (one query row per sequence) for the OTHER head sizes the reference instantiates (csrc/build.rs:7-74: 32, 96, 160, 192, 224,
// 256; any multiple of 8 up to 256): a streaming kernel in the shape of paged_decode.hip's, without its tuning.  One wavefront per
// (sequence, kv head, chunk of up to 4 q heads of the GQA group): every K / V byte is fetched once for the chunk (the row-per-lane kernel above
// reads a K row per lane, 16 bytes at a time, 64 rows apart -- and once per q head).  A row's D / 8 16-byte chunks sit on CP adjacent lanes
// (CP = the next power of two: lanes beyond D / 8 idle), 64 / CP rows per load instruction, 16-token tiles; q.k by v_dot2 + an xor-shuffle
// reduction over the row's lanes; one online-softmax state per q head (running max common to the wavefront, row sums and O kept per row group and
// folded once at the end); P rounded to the storage type before P.V (softmax.h:65-185).  No KV split: small batches leave CUs idle here.
// ------------------------------------------------------------------------------------------
template <typename T, int CP>
__global__ void __launch_bounds__(64) attn_decode_anyd_kernel(const AttnParams p, const int gchunks) {
    constexpr int R = 64 / CP, NP = 16 / R;    // rows per load instruction, load instructions per 16-token tile
    constexpr int GQ = 4;
    const int lane = threadIdx.x, r = lane / CP, c = lane % CP;
    const int g = p.h / p.h_k;
    const int gc = blockIdx.x % gchunks, hk = (blockIdx.x / gchunks) % p.h_k, b = blockIdx.x / (gchunks * p.h_k);
    const int hq0 = hk * g + gc * GQ, nq = min(GQ, g - gc * GQ);
    const SeqInfo si(p, b);
    const int L = si.len_k, C = p.d >> 3;
    const bool act = c < C;
    uint4 qv[GQ];
    float m[GQ], l[GQ], o[GQ][8], slope[GQ];
#pragma unroll
    for (int gq = 0; gq < GQ; ++gq) {
        qv[gq] = make_uint4(0, 0, 0, 0);
        if (act && gq < nq) qv[gq] = *reinterpret_cast<const uint4 *>(p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hq0 + gq) * p.q_head_stride + c * 8);
        m[gq] = -INFINITY;
        l[gq] = 0.f;
        slope[gq] = (p.alibi_slopes && gq < nq) ? p.alibi_slopes[b * p.alibi_batch_stride + hq0 + gq] * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[gq][e] = 0.f;
    }
    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int64_t koff = paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b);
    const int64_t voff = paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b);
    for (int t0 = 0; t0 < L; t0 += 16) {
        uint4 kk[NP], vv[NP];
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const int tc = min(t0 + pi * R + r, L - 1);         // rows behind the sequence re-read its last row (never-written slots may hold anything) and get p = 0
            int64_t ko, vo;
            if (paged) {
                const int64_t pg = bt[tc / p.page_size];
                ko = pg * p.k_batch_stride + (int64_t)(tc % p.page_size) * p.k_row_stride;
                vo = pg * p.v_batch_stride + (int64_t)(tc % p.page_size) * p.v_row_stride;
            } else {
                ko = koff + (int64_t)tc * p.k_row_stride;
                vo = voff + (int64_t)tc * p.v_row_stride;
            }
            kk[pi] = make_uint4(0, 0, 0, 0);
            vv[pi] = make_uint4(0, 0, 0, 0);
            if (act) {
                kk[pi] = *reinterpret_cast<const uint4 *>(p.k + ko + (int64_t)hk * p.k_head_stride + c * 8);
                vv[pi] = *reinterpret_cast<const uint4 *>(p.v + vo + (int64_t)hk * p.v_head_stride + c * 8);
            }
        }
#pragma unroll
        for (int gq = 0; gq < GQ; ++gq) {
            if (gq >= nq) continue;                              // wave-uniform
            float s[NP], mx = -INFINITY;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                float acc = dot2<T>(kk[pi].x, qv[gq].x, 0.f);
                acc = dot2<T>(kk[pi].y, qv[gq].y, acc);
                acc = dot2<T>(kk[pi].z, qv[gq].z, acc);
                acc = dot2<T>(kk[pi].w, qv[gq].w, acc);
#pragma unroll
                for (int off = 1; off < CP; off <<= 1) acc += __shfl_xor(acc, off, 64);
                const int tok = t0 + pi * R + r;
                s[pi] = tok < L ? acc * p.scale_log2 - slope[gq] * (float)(L - 1 - tok) : -INFINITY;   // ALiBi: mask.h:179-186 with one query row at position L - 1
                mx = fmaxf(mx, s[pi]);
            }
#pragma unroll
            for (int off = CP; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float m_new = fmaxf(m[gq], mx);
            const float ms = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m[gq] - ms);
            m[gq] = m_new;
            l[gq] *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[gq][e] *= alpha;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                const float pj = __builtin_amdgcn_exp2f(s[pi] - ms);    // exp2(-inf) = 0 for the rows behind the sequence
                l[gq] += pj;
                const float pr = round_through<T>(pj);
                o[gq][0] += pr * lo_to_f32<T>(vv[pi].x); o[gq][1] += pr * hi_to_f32<T>(vv[pi].x);
                o[gq][2] += pr * lo_to_f32<T>(vv[pi].y); o[gq][3] += pr * hi_to_f32<T>(vv[pi].y);
                o[gq][4] += pr * lo_to_f32<T>(vv[pi].z); o[gq][5] += pr * hi_to_f32<T>(vv[pi].z);
                o[gq][6] += pr * lo_to_f32<T>(vv[pi].w); o[gq][7] += pr * hi_to_f32<T>(vv[pi].w);
            }
        }
    }
#pragma unroll
    for (int gq = 0; gq < GQ; ++gq) {
        if (gq >= nq) continue;
        float lt = l[gq];
#pragma unroll
        for (int off = CP; off < 64; off <<= 1) {
            lt += __shfl_xor(lt, off, 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[gq][e] += __shfl_xor(o[gq][e], off, 64);
        }
        const bool empty = !(lt > 0.f);                          // no key: O = 0, LSE = +inf (flash_fwd_kernel.h:97-133)
        const float inv = empty ? 0.f : 1.f / lt;
        if (act && r == 0) {
            uint4 w;
            w.x = pack2<T>(o[gq][0] * inv, o[gq][1] * inv); w.y = pack2<T>(o[gq][2] * inv, o[gq][3] * inv);
            w.z = pack2<T>(o[gq][4] * inv, o[gq][5] * inv); w.w = pack2<T>(o[gq][6] * inv, o[gq][7] * inv);
            *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)(hq0 + gq) * p.o_head_stride + c * 8) = w;
        }
        if (p.lse && lane == 0) p.lse[(int64_t)b * p.h + hq0 + gq] = empty ? INFINITY : (m[gq] + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------
// Prefill (several query rows per sequence) for the other head sizes that are multiples of 16 (32, 96, 160, 192, 224, 256): one wavefront per block
// of 16 query rows of one (sequence, q head), K / V tiles of 16 keys shared by the 16 rows (the row-per-wavefront kernel above re-reads all of K
// and V for every row), both products on v_mfma_f32_16x16x16 in the swapped form of the tuned kernels:
//     S^T[key][q] = K . Q^T   -- A = K  (lane (grp, col): K[key0 + col][16 c + 4 grp ..+3], 8 bytes straight from the row),
//                                B = Q^T (Q[q0 + col][16 c + 4 grp ..+3], held in registers for the whole block); result: S^T[4 grp + i][col]
//     O^T[d][q]  += V^T . P^T  -- B = the lane's own four probabilities (keys 4 grp ..+3 of row col), A = V^T from the tile staged in LDS.
// So a lane owns ONE query row (col) and four keys per tile: online softmax state per lane (max shared over the 4 lanes of a column), P rounded to
// the storage type before P.V (softmax.h:65-185), O^T[16 c + 4 grp + i][col] in D / 4 accumulator registers.  Untuned (no pipelining, V through
// a row-major LDS tile read 2 bytes at a time): the point is 16 x less K / V traffic, not the matrix pipe.
// ------------------------------------------------------------------------------------------
// two f32 -> one packed pair of the storage type, round to nearest even in hardware (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <typename T> __device__ __forceinline__ uint32_t gpack(float lo, float hi);
template <> __device__ __forceinline__ uint32_t gpack<bf16_t>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_v));
}
template <> __device__ __forceinline__ uint32_t gpack<f16_t>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2_v));
}

// max over the four lanes (grp = 0 .. 3) that hold the same query row: two VALU lane swaps instead of two LDS round trips
__device__ __forceinline__ float gcol_max4(float x) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef __attribute__((ext_vector_type(4))) float gf32x4;
template <typename T> __device__ __forceinline__ gf32x4 gmfma16(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, gf32x4 c);
template <> __device__ __forceinline__ gf32x4 gmfma16<bf16_t>(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, gf32x4 c) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u2{a0, a1}), __builtin_bit_cast(s16x4, u2{b0, b1}), c, 0, 0, 0);
}
template <> __device__ __forceinline__ gf32x4 gmfma16<f16_t>(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, gf32x4 c) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, u2{a0, a1}), __builtin_bit_cast(h16x4, u2{b0, b1}), c, 0, 0, 0);
}

template <typename T, int NCMAX, bool EXACT>                     // head_dim / 16 <= NCMAX (EXACT: == NCMAX, no predicated chunks)
__global__ void __launch_bounds__(64) attn_prefill_tile16_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t v_sm[16 * 16 * NCMAX];
    const int lane = threadIdx.x, grp = lane >> 4, col = lane & 15;
    const int mblk = blockIdx.x, hq = blockIdx.y, b = blockIdx.z;
    const SeqInfo si(p, b);
    const int q0 = mblk * 16;
    if (q0 >= si.len_q) return;
    const int D = EXACT ? 16 * NCMAX : p.d, nc = D >> 4;
    const int hk = hq / (p.h / p.h_k);
    const int shift = si.len_k - si.len_q;                        // mask.h:170
    const int qrow = q0 + col;                                   // this lane's query row
    const bool qvalid = qrow < si.len_q;
    const int qr = min(qrow, si.len_q - 1);
    int hi_q = p.is_causal ? min(si.len_k, qr + shift + 1) : si.len_k;   // keys this row sees
    if (hi_q < 0) hi_q = 0;
    const int last = min(q0 + 15, si.len_q - 1);
    int hi_blk = p.is_causal ? min(si.len_k, last + shift + 1) : si.len_k;
    if (hi_blk < 0) hi_blk = 0;
    const uint16_t *qptr = p.q + si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)qr * p.q_row_stride + (int64_t)hq * p.q_head_stride + 4 * grp;
    uint32_t qreg[NCMAX][2];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
        qreg[c][0] = qreg[c][1] = 0;
        if (c < nc) {
            const uint2 t = *reinterpret_cast<const uint2 *>(qptr + 16 * c);
            qreg[c][0] = t.x; qreg[c][1] = t.y;
        }
    }
    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int64_t koff = paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b);
    const int64_t voff = paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b);
    const float slope_l2 = p.alibi_slopes ? p.alibi_slopes[b * p.alibi_batch_stride + hq] * 1.4426950408889634f : 0.f;
    auto row_off = [&](int t, bool is_v) -> int64_t {             // element offset of token t's row of this kv head
        if (paged) {
            const int64_t pg = bt[t / p.page_size];
            return is_v ? pg * p.v_batch_stride + (int64_t)(t % p.page_size) * p.v_row_stride + (int64_t)hk * p.v_head_stride
                        : pg * p.k_batch_stride + (int64_t)(t % p.page_size) * p.k_row_stride + (int64_t)hk * p.k_head_stride;
        }
        return is_v ? voff + (int64_t)t * p.v_row_stride + (int64_t)hk * p.v_head_stride : koff + (int64_t)t * p.k_row_stride + (int64_t)hk * p.k_head_stride;
    };
    gf32x4 o[NCMAX];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) o[c] = gf32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l = 0.f;
    const int vpr = D >> 3;                                       // 16-byte pieces per V row
    for (int j0 = 0; j0 < hi_blk; j0 += 16) {
        // ---- V tile -> LDS (row-major [key][D]); rows behind the sequence re-read its last row and meet p = 0
        for (int idx = lane; idx < 16 * vpr; idx += 64) {
            const int kr = idx / vpr, ch = idx - kr * vpr;
            const int t = min(j0 + kr, si.len_k - 1);
            *reinterpret_cast<uint4 *>(&v_sm[kr * D + ch * 8]) = *reinterpret_cast<const uint4 *>(p.v + row_off(t, true) + ch * 8);
        }
        // ---- S^T = K . Q^T
        const int tk = min(j0 + col, si.len_k - 1);
        const uint16_t *kptr = p.k + row_off(tk, false) + 4 * grp;
        gf32x4 sacc = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCMAX; ++c) {
            if (c < nc) {
                const uint2 kk = *reinterpret_cast<const uint2 *>(kptr + 16 * c);
                sacc = gmfma16<T>(kk.x, kk.y, qreg[c][0], qreg[c][1], sacc);
            }
        }
        float sv[4], mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = j0 + 4 * grp + i;
            float x = sacc[i] * p.scale_log2;
            if (p.alibi_slopes) x -= slope_l2 * fabsf((float)(qr + shift - key));   // mask.h:179-186
            sv[i] = key < hi_q ? x : -INFINITY;
            mx = fmaxf(mx, sv[i]);
        }
        mx = gcol_max4(mx);
        const float m_new = fmaxf(m_run, mx);
        const float ms = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - ms);
        m_run = m_new;
        float pr[4], psum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pj = __builtin_amdgcn_exp2f(sv[i] - ms);
            psum += pj;
            pr[i] = pj;
        }
        l = l * alpha + psum;
        const uint32_t b0 = gpack<T>(pr[0], pr[1]), b1 = gpack<T>(pr[2], pr[3]);
        __syncthreads();                                          // the V tile is in LDS
#pragma unroll
        for (int c = 0; c < NCMAX; ++c) {
            if (c < nc) {
                const uint16_t *vp = &v_sm[(4 * grp) * D + 16 * c + col];
                const uint32_t a0 = (uint32_t)vp[0] | ((uint32_t)vp[D] << 16), a1 = (uint32_t)vp[2 * D] | ((uint32_t)vp[3 * D] << 16);
                gf32x4 acc = o[c];
                acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
                o[c] = gmfma16<T>(a0, a1, b0, b1, acc);
            }
        }
        __syncthreads();                                          // before the next tile overwrites it
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const bool empty = !(l > 0.f);                                // no visible key: O = 0, LSE = +inf (flash_fwd_kernel.h:97-133)
    const float inv = empty ? 0.f : 1.f / l;
    if (qvalid) {
        uint16_t *orow = p.o + si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)qrow * p.o_row_stride + (int64_t)hq * p.o_head_stride + 4 * grp;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c) {
            if (c < nc) {
                uint2 w;
                w.x = gpack<T>(o[c][0] * inv, o[c][1] * inv);
                w.y = gpack<T>(o[c][2] * inv, o[c][3] * inv);
                *reinterpret_cast<uint2 *>(orow + 16 * c) = w;
            }
        }
        if (p.lse && grp == 0) {
            const float lse = empty ? INFINITY : (m_run + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
            if (p.unpadded_lse && p.cu_seqlens_q) p.lse[(int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q + qrow] = lse;
            else p.lse[((int64_t)b * p.h + hq) * p.seqlen_q + qrow] = lse;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same products for head sizes that are multiples of 32 (every one the reference instantiates: 32 ... 256) and more than 16 query rows:
// a workgroup of 4 wavefronts = 64 query rows of one (sequence, q head), K / V tiles of KT keys staged ONCE per workgroup through LDS
// (register prefetch of tile t + 1 while tile t is computed), v_mfma_f32_16x16x32 for both products:
//     S^T sub-tile h of key group g: A = K[32 g + 16 h + col][32 c + 8 grp ..+7] (one ds_read_b128; rows padded to 16 * odd bytes: 16 rows on 16
//                                    different 16-byte slots), B = Q[q col][32 c + 8 grp ..+7] in registers; the lane gets keys 32 g + 16 h + 4 grp + i
//     O^T chunk dc (16 head-dim columns): B = the lane's 8 probabilities of group g in k order (h, i) = keys {32 g + 4 grp ..+3, 32 g + 16 + 4 grp ..+3},
//                                    A = V^T for those 8 keys = two ds_read_b64_tr_b16 of the row-major V tile (rows padded to 32 * odd bytes).
// The 4 wavefronts of a causal block end at different keys: a wavefront skips the tiles behind its last row, the loads and barriers are the
// workgroup's.  Workgroups are numbered so that the q heads of one kv head, and neighbouring row blocks, run on ONE XCD (its L2 then serves the
// K / V re-reads), heaviest (last) row blocks first.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short gshort4;
typedef __attribute__((ext_vector_type(4))) unsigned int gu32x4;
template <typename T> __device__ __forceinline__ gf32x4 gmfma32(const gu32x4 &a, const gu32x4 &b, gf32x4 c);
template <> __device__ __forceinline__ gf32x4 gmfma32<bf16_t>(const gu32x4 &a, const gu32x4 &b, gf32x4 c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ gf32x4 gmfma32<f16_t>(const gu32x4 &a, const gu32x4 &b, gf32x4 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}

template <typename T, int NC, int KT, int ST, int RQ>   // RQ 16-row blocks per wavefront: a workgroup covers 64 RQ query rows
__global__ void __launch_bounds__(256, 2) attn_prefill_tile64_kernel(const AttnParams p, const int mblocks) {
    constexpr int D = 32 * NC, KRB = 2 * D + 16, VRB = 2 * D + 32, KG = KT / 32, CPR = D / 8;
    constexpr int PIECES = KT * CPR, NPC = (PIECES + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[KT * (KRB + VRB)];
    const uint32_t k_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem, v_lds = k_lds + KT * KRB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, col = lane & 15;
    // ---- workgroup -> (row block, q head): consecutive ids of one XCD walk the q heads first, then the row blocks from the last one down
    const int N = gridDim.x, full = (N >> 3) << 3;
    int w = blockIdx.x;
    if (w < full) w = (w & 7) * (N >> 3) + (w >> 3);
    const int hq = w % p.h, mblk = mblocks - 1 - w / p.h, b = blockIdx.z;
    const SeqInfo si(p, b);
    const int q0 = mblk * (64 * RQ);
    if (q0 >= si.len_q) return;
    const int hk = hq / (p.h / p.h_k);
    const int shift = si.len_k - si.len_q;                        // mask.h:170
    const int wq0 = q0 + 16 * RQ * wave;                          // this wavefront's rows: RQ blocks of 16, block s at wq0 + 16 s
    int qrow[RQ], qr[RQ], hi_q[RQ], hi_first[RQ];
    bool qvalid[RQ];
#pragma unroll
    for (int s = 0; s < RQ; ++s) {
        qrow[s] = wq0 + 16 * s + col;
        qvalid[s] = qrow[s] < si.len_q;
        qr[s] = min(qrow[s], si.len_q - 1);
        hi_q[s] = max(0, p.is_causal ? min(si.len_k, qr[s] + shift + 1) : si.len_k);
        hi_first[s] = p.is_causal ? min(si.len_k, wq0 + 16 * s + shift + 1) : si.len_k;   // keys EVERY row of block s sees
    }
    const int hi_wave = wq0 < si.len_q ? max(0, p.is_causal ? min(si.len_k, min(wq0 + 16 * RQ - 1, si.len_q - 1) + shift + 1) : si.len_k) : 0;
    const int hi_wg = max(0, p.is_causal ? min(si.len_k, min(q0 + 64 * RQ - 1, si.len_q - 1) + shift + 1) : si.len_k);
    gu32x4 qreg[RQ][NC];
#pragma unroll
    for (int s = 0; s < RQ; ++s) {
        const uint16_t *qptr = p.q + si.q_offset(p.q_batch_stride, p.q_row_stride, b) + (int64_t)qr[s] * p.q_row_stride + (int64_t)hq * p.q_head_stride + 8 * grp;
#pragma unroll
        for (int c = 0; c < NC; ++c) qreg[s][c] = *reinterpret_cast<const gu32x4 *>(qptr + 32 * c);
    }
    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int pshift = (paged && (p.page_size & (p.page_size - 1)) == 0) ? __builtin_ctz(p.page_size) : -1;
    const int64_t kbase = (paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b)) + (int64_t)hk * p.k_head_stride;
    const int64_t vbase = (paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b)) + (int64_t)hk * p.v_head_stride;
    const float slope_raw = p.alibi_slopes ? p.alibi_slopes[b * p.alibi_batch_stride + hq] / p.scale : 0.f;   // slope.log2(e) / scale_log2
    gu32x4 kreg0[NPC], vreg0[NPC], kreg1[NPC], vreg1[NPC];       // ST register stages (separate arrays: a [ST][NPC] array of this size is left in scratch): tile t + ST is requested while tile t is computed
    // Addresses: 32-bit arithmetic inside a page / a row, ONE 32 x 32 -> 64-bit multiply-add per tensor and piece (the strides arrive as u32 over
    // the FFI: ffi.rs:3-102).  Paged: the page numbers of tile t + 1's rows are requested right after tile t's K / V loads, so the block-table
    // latency never sits in front of a K / V load.
    const uint32_t krs = (uint32_t)p.k_row_stride, vrs = (uint32_t)p.v_row_stride, kbs = (uint32_t)p.k_batch_stride, vbs = (uint32_t)p.v_batch_stride;
    const uint16_t *kg = p.k + kbase, *vg = p.v + vbase;
    const uint32_t pmask = (uint32_t)p.page_size - 1u;
    int pg[NPC];
    auto piece_row = [&](int i, int j0) { return min(j0 + (tid + 256 * i) / CPR, si.len_k - 1); };   // rows behind the sequence re-read its last row and meet p = 0
    auto load_pages = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPC; ++i)
            if (PIECES % 256 == 0 || tid + 256 * i < PIECES) {
                const int t = piece_row(i, j0);
                pg[i] = bt[pshift >= 0 ? t >> pshift : t / p.page_size];
            }
    };
    auto load_tile = [&](auto sel, int j0) __attribute__((always_inline)) {
        constexpr int SG = decltype(sel)::value;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int idx = tid + 256 * i;
            if (PIECES % 256 == 0 || idx < PIECES) {
                const uint32_t ch8 = (uint32_t)(idx % CPR) * 8u;
                const int t = piece_row(i, j0);
                uint64_t ko, vo;
                if (paged) {
                    const uint32_t r = pshift >= 0 ? (uint32_t)t & pmask : (uint32_t)(t % p.page_size);
                    ko = (uint64_t)(uint32_t)pg[i] * kbs + (__umul24(r, krs) + ch8);
                    vo = (uint64_t)(uint32_t)pg[i] * vbs + (__umul24(r, vrs) + ch8);
                } else {
                    ko = (uint64_t)(uint32_t)t * krs + ch8;
                    vo = (uint64_t)(uint32_t)t * vrs + ch8;
                }
                const gu32x4 kx = *reinterpret_cast<const gu32x4 *>(kg + ko), vx = *reinterpret_cast<const gu32x4 *>(vg + vo);
                if constexpr (SG == 0) { kreg0[i] = kx; vreg0[i] = vx; } else { kreg1[i] = kx; vreg1[i] = vx; }
            }
        }
        if (paged && j0 + KT < hi_wg) load_pages(j0 + KT);
    };
    auto store_tile = [&](auto sel) __attribute__((always_inline)) {
        constexpr int SG = decltype(sel)::value;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int idx = tid + 256 * i;
            if (PIECES % 256 == 0 || idx < PIECES) {
                const int kr = idx / CPR, ch = idx - kr * CPR;
                *reinterpret_cast<gu32x4 *>(smem + kr * KRB + ch * 16) = SG == 0 ? kreg0[i] : kreg1[i];
                *reinterpret_cast<gu32x4 *>(smem + KT * KRB + kr * VRB + ch * 16) = SG == 0 ? vreg0[i] : vreg1[i];
            }
        }
    };
    gf32x4 o[RQ][2 * NC];
    float m_run[RQ], l[RQ];
#pragma unroll
    for (int s = 0; s < RQ; ++s) {
        m_run[s] = -INFINITY;
        l[s] = 0.f;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) o[s][c] = gf32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t k_rd = k_lds + col * KRB + grp * 16;                                  // + (32 g + 16 h) rows + 64 c
    const uint32_t v_rd = v_lds + (4 * grp + (col >> 2)) * VRB + (col & 3) * 8;          // + (32 g + 16 h) rows + 32 dc
    auto compute_tile = [&](int j0) __attribute__((always_inline)) {
        // ---- S^T = K . Q^T: one K operand read feeds the RQ row blocks.  Head-dim step outermost: consecutive MFMAs then belong to different
        // accumulators (a chain of dependent MFMAs issues at half rate)
        gf32x4 sacc[RQ][KG][2];
#pragma unroll
        for (int s = 0; s < RQ; ++s)
#pragma unroll
            for (int g = 0; g < KG; ++g) { sacc[s][g][0] = gf32x4{0.f, 0.f, 0.f, 0.f}; sacc[s][g][1] = gf32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const gu32x4 a = *(const __attribute__((address_space(3))) gu32x4 *)(uintptr_t)(k_rd + (32 * g + 16 * hh) * KRB + 64 * c);
#pragma unroll
                    for (int s = 0; s < RQ; ++s) sacc[s][g][hh] = gmfma32<T>(a, qreg[s][c], sacc[s][g][hh]);
                }
        // ---- softmax per row block.  Scores stay raw: p = exp2(s * scale_log2 - m) is one fma + one exp; the mask / ALiBi pass only runs on tiles that reach
        // past the block's first row's last key (the diagonal, the sequence's tail) or when slopes are given (softmax.h:65-185 in the exp2 domain; scale_log2 > 0)
        gu32x4 pb[RQ][KG];
#pragma unroll
        for (int s = 0; s < RQ; ++s) {
            const bool plain = !p.alibi_slopes && j0 + KT <= hi_first[s];
            float mx = -INFINITY;
            if (plain) {
#pragma unroll
                for (int g = 0; g < KG; ++g)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sacc[s][g][hh][i]);
                mx *= p.scale_log2;
            } else {
#pragma unroll
                for (int g = 0; g < KG; ++g)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int key = j0 + 32 * g + 16 * hh + 4 * grp + i;
                            float x = sacc[s][g][hh][i];
                            if (p.alibi_slopes) x -= slope_raw * fabsf((float)(qr[s] + shift - key));   // mask.h:179-186, in units of 1 / scale_log2
                            x = key < hi_q[s] ? x : -INFINITY;
                            sacc[s][g][hh][i] = x;
                            mx = fmaxf(mx, x * p.scale_log2);
                        }
            }
            mx = gcol_max4(mx);
            const float m_new = fmaxf(m_run[s], mx);
            const float ms = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[s] - ms);
            m_run[s] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                float pr[2][4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pr[hh][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[s][g][hh][i], p.scale_log2, -ms));
                        psum += pr[hh][i];
                    }
                pb[s][g] = gu32x4{gpack<T>(pr[0][0], pr[0][1]), gpack<T>(pr[0][2], pr[0][3]), gpack<T>(pr[1][0], pr[1][1]), gpack<T>(pr[1][2], pr[1][3])};
            }
            l[s] = l[s] * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // (the running max settles after the first tiles of a row block)
#pragma unroll
                for (int dc = 0; dc < 2 * NC; ++dc) { o[s][dc][0] *= alpha; o[s][dc][1] *= alpha; o[s][dc][2] *= alpha; o[s][dc][3] *= alpha; }
            }
        }
        // ---- O^T += V^T . P^T: one V^T operand (two transposing reads) feeds the RQ row blocks
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int dc = 0; dc < 2 * NC; ++dc) {
                const uint32_t a0 = v_rd + (32 * g) * VRB + 32 * dc;
                const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gshort4 *)(uintptr_t)a0));
                const uint2 hi2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gshort4 *)(uintptr_t)(a0 + 16 * VRB)));
                const gu32x4 a = gu32x4{lo.x, lo.y, hi2.x, hi2.y};
#pragma unroll
                for (int s = 0; s < RQ; ++s) o[s][dc] = gmfma32<T>(a, pb[s][g], o[s][dc]);
            }
    };
    typedef std::integral_constant<int, 0> Stage0;
    typedef std::integral_constant<int, ST - 1> Stage1;
    auto step = [&](auto sel, int j0) __attribute__((always_inline)) {
        store_tile(sel);
        __syncthreads();
        if (j0 + ST * KT < hi_wg) load_tile(sel, j0 + ST * KT);
        if (j0 < hi_wave) compute_tile(j0);
        __syncthreads();
    };
    if (paged && hi_wg > 0) load_pages(0);
    if (hi_wg > 0) load_tile(Stage0(), 0);
    if (ST == 2 && KT < hi_wg) load_tile(Stage1(), KT);
    for (int j0 = 0; j0 < hi_wg; j0 += ST * KT) {
        step(Stage0(), j0);
        if (ST == 2 && j0 + KT < hi_wg) step(Stage1(), j0 + KT);
    }
#pragma unroll
    for (int s = 0; s < RQ; ++s) {
        float lt = l[s];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const bool empty = !(lt > 0.f);                           // no visible key: O = 0, LSE = +inf (flash_fwd_kernel.h:97-133)
        const float inv = empty ? 0.f : 1.f / lt;
        if (qvalid[s]) {
            uint16_t *orow = p.o + si.q_offset(p.o_batch_stride, p.o_row_stride, b) + (int64_t)qrow[s] * p.o_row_stride + (int64_t)hq * p.o_head_stride + 4 * grp;
#pragma unroll
            for (int dc = 0; dc < 2 * NC; ++dc) {
                uint2 wv;
                wv.x = gpack<T>(o[s][dc][0] * inv, o[s][dc][1] * inv);
                wv.y = gpack<T>(o[s][dc][2] * inv, o[s][dc][3] * inv);
                *reinterpret_cast<uint2 *>(orow + 16 * dc) = wv;
            }
            if (p.lse && grp == 0) {
                const float lse = empty ? INFINITY : (m_run[s] + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
                if (p.unpadded_lse && p.cu_seqlens_q) p.lse[(int64_t)hq * p.cu_seqlens_q[p.b] + si.sum_q + qrow[s]] = lse;
                else p.lse[((int64_t)b * p.h + hq) * p.seqlen_q + qrow[s]] = lse;
            }
        }
    }
}

// Knobs of this file: environment at load time, atoma_set_option("generic_*") at run time (A/B runs and tests inside one process) -- no getenv per call.
static int generic_env_or(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
static std::atomic<int> generic_prefill_tile{generic_env_or("ATOMA_GENERIC_PREFILL_TILE", 64)};   // 64: the tiled kernels; 16: only the 16-row one; 0: the row-per-wavefront kernel
static std::atomic<int> generic_prefill_kt{generic_env_or("ATOMA_GENERIC_PREFILL_KT", 0)};        // keys per LDS tile of the 64-row kernel: 0 = by head size, 32, 64
static std::atomic<int> generic_prefill_rq{generic_env_or("ATOMA_GENERIC_PREFILL_RQ", 0)};        // 16-row blocks per wavefront: 0 = by head size, 1, 2
static std::atomic<int> generic_decode_stream{generic_env_or("ATOMA_GENERIC_DECODE_STREAM", 2)};  // 2: the streaming decode kernel (second version); 1: its first version; 0: the row-per-lane kernel
static std::atomic<int> generic_decode_waves{generic_env_or("ATOMA_GENERIC_DECODE_WAVES", 0)};    // wavefronts per unit: 0 = by shape, 1 / 2 / 4 / 8
bool set_generic_attn_option(const std::string &name, int value) {
    if (name == "generic_prefill_tile") generic_prefill_tile = value;
    else if (name == "generic_prefill_kt") generic_prefill_kt = value;
    else if (name == "generic_prefill_rq") generic_prefill_rq = value;
    else if (name == "generic_decode_stream") generic_decode_stream = value;
    else if (name == "generic_decode_waves") generic_decode_waves = value;
    else return false;
    return true;
}

template <typename T>
static void launch_prefill_tile64(const AttnParams &p, hipStream_t stream) {
    int kt = p.d == 192 ? 32 : 64;                               // measured per head size: profiles/r05_generic_prefill_cfg.json
    if (const int v = generic_prefill_kt.load()) kt = v == 32 ? 32 : 64;   // A/B runs
    // two 16-row blocks per wavefront (128-row workgroups: every K / V operand read from LDS feeds two MFMAs) up to head size 128, where the registers allow it
    int rq = (p.d <= 128 && p.seqlen_q > 64) ? 2 : 1;
    if (const int v = generic_prefill_rq.load()) rq = (v == 2 && p.d <= 128) ? 2 : 1;   // A/B runs
    const int mblocks = (p.seqlen_q + 64 * rq - 1) / (64 * rq);
    const dim3 grid((unsigned)(mblocks * p.h), 1, (unsigned)p.b);
    // (two register prefetch stages -- tile t + 2 requested while tile t is computed -- were measured again with everything in registers and never won:
    // profiles/r05_generic_prefill_cfg.json; ST stays 1)
#define ATOMA_T64B(NC_, KT_, RQ_) hipLaunchKernelGGL((attn_prefill_tile64_kernel<T, NC_, KT_, 1, RQ_>), grid, dim3(256), 0, stream, p, mblocks)
#define ATOMA_T64S(NC_) case NC_: if (rq == 2) { if (kt == 64) ATOMA_T64B(NC_, 64, 2); else ATOMA_T64B(NC_, 32, 2); } \
                                  else { if (kt == 64) ATOMA_T64B(NC_, 64, 1); else ATOMA_T64B(NC_, 32, 1); } break
#define ATOMA_T64(NC_) case NC_: if (kt == 64) ATOMA_T64B(NC_, 64, 1); else ATOMA_T64B(NC_, 32, 1); break
    switch (p.d / 32) {
        ATOMA_T64S(1); ATOMA_T64S(2); ATOMA_T64S(3); ATOMA_T64S(4); ATOMA_T64(5); ATOMA_T64(6); ATOMA_T64(7); ATOMA_T64(8);
    }
#undef ATOMA_T64
#undef ATOMA_T64S
#undef ATOMA_T64B
}

// ------------------------------------------------------------------------------------------
// attn_decode_anyd_kernel, second version (round 5): same decomposition -- one wavefront per (sequence, kv head, chunk of NQ q heads), a row's 16-byte chunks
// on CP adjacent lanes, 16-token tiles -- with what the counters of the prefill kernel above taught:
//  * a 16-token tile lies in ONE page (pages are multiples of 16 tokens: lib.rs:778-785), so its page number is wave-uniform: a scalar load, fetched two tiles
//    ahead of the tile's K / V loads; the tile's base address is scalar, the lane's offset inside a tile is 32-bit (one 24-bit multiply for the tail clamp);
//  * two register sets: tile t + 1's K / V loads are in flight while tile t is computed (NQ is a template parameter -- MHA, Phi-3's case, keeps one softmax
//    state, not four -- so the second set still fits three wavefronts per SIMD at d = 96);
//  * the q.k reduction over a row's lanes and the row maximum by DPP / lane swaps instead of LDS shuffles; P rounded by v_cvt_pk.
// ------------------------------------------------------------------------------------------
template <int CP> __device__ __forceinline__ float grow_sum(float x) {                         // all-reduce over the CP adjacent lanes of a row
    if constexpr (CP >= 2) x += __builtin_amdgcn_update_dpp(0.f, x, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    if constexpr (CP >= 4) x += __builtin_amdgcn_update_dpp(0.f, x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    if constexpr (CP >= 8) x += __builtin_amdgcn_update_dpp(0.f, x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    if constexpr (CP >= 16) x += __builtin_amdgcn_update_dpp(0.f, x, 0x140, 0xf, 0xf, true);  // row_mirror
    if constexpr (CP >= 32) {
        typedef __attribute__((ext_vector_type(2))) unsigned int u2;
        const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return x;
}
template <int CP> __device__ __forceinline__ float gwave_max_over_rows(float x) {              // x is uniform over each row's CP lanes: max over the 64 / CP rows
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    if constexpr (CP <= 4) x = fmaxf(x, __builtin_amdgcn_update_dpp(-INFINITY, x, 0x141, 0xf, 0xf, false));
    if constexpr (CP <= 8) x = fmaxf(x, __builtin_amdgcn_update_dpp(-INFINITY, x, 0x140, 0xf, 0xf, false));
    if constexpr (CP <= 16) {
        const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <typename T, int CP, int NQ, int NW>                    // NW wavefronts share one unit's tiles (batches that do not fill the chip) and merge in LDS
__global__ void __launch_bounds__(64 * NW) attn_decode_anyd2_kernel(const AttnParams p, const int gchunks) {
    constexpr int R = 64 / CP, NP = 16 / R;    // rows per load instruction, load instructions per 16-token tile
    constexpr bool DB = CP <= 16;              // two register sets up to head size 128; one above (measured: a second set of 64 registers does not pay there)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane / CP, c = lane % CP;
    const int g = p.h / p.h_k;
    // workgroup -> unit: consecutive units (the kv heads of one token row share 128-byte lines when a head's row is not a multiple of 128 bytes) on ONE XCD
    const int N = gridDim.x, full = (N >> 3) << 3;
    int w = blockIdx.x;
    if (w < full) w = (w & 7) * (N >> 3) + (w >> 3);
    const int gc = w % gchunks, hk = (w / gchunks) % p.h_k, b = w / (gchunks * p.h_k);
    const int hq0 = hk * g + gc * NQ, nq = min(NQ, g - gc * NQ);
    const SeqInfo si(p, b);
    const int L = si.len_k, C = p.d >> 3;
    const bool act = c < C;
    uint4 qv[NQ];
    float m[NQ], l[NQ], o[NQ][8], slope[NQ];
#pragma unroll
    for (int gq = 0; gq < NQ; ++gq) {
        qv[gq] = make_uint4(0, 0, 0, 0);
        if (act && gq < nq) qv[gq] = *reinterpret_cast<const uint4 *>(p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hq0 + gq) * p.q_head_stride + c * 8);
        m[gq] = -INFINITY;
        l[gq] = 0.f;
        slope[gq] = (p.alibi_slopes && gq < nq) ? p.alibi_slopes[b * p.alibi_batch_stride + hq0 + gq] * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[gq][e] = 0.f;
    }
    const bool paged = p.block_table != nullptr;
    const int *bt = paged ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int pshift = (paged && (p.page_size & (p.page_size - 1)) == 0) ? __builtin_ctz(p.page_size) : -1;
    const uint32_t krs = (uint32_t)p.k_row_stride, vrs = (uint32_t)p.v_row_stride;
    const uint16_t *kg = p.k + (paged ? 0 : si.k_offset(p.k_batch_stride, p.k_row_stride, b)) + (int64_t)hk * p.k_head_stride;
    const uint16_t *vg = p.v + (paged ? 0 : si.k_offset(p.v_batch_stride, p.v_row_stride, b)) + (int64_t)hk * p.v_head_stride;
    auto page_of = [&](int t0) -> int {                          // wave-uniform: the page of tile t0 (tiles behind the sequence: its last page)
        if (!paged) return 0;
        const int t = min(t0, L - 1);
        return __builtin_amdgcn_readfirstlane(bt[pshift >= 0 ? t >> pshift : t / p.page_size]);
    };
    uint4 kk[DB ? 2 : 1][NP], vv[DB ? 2 : 1][NP];
    auto load_tile = [&](auto sel, int t0, int pg) {
        constexpr int SG = decltype(sel)::value;
        const uint16_t *kb, *vb;                                  // scalar: the tile's first row
        if (paged) {
            const uint32_t r0 = pshift >= 0 ? (uint32_t)t0 & ((uint32_t)p.page_size - 1u) : (uint32_t)(t0 % p.page_size);
            kb = kg + (int64_t)pg * p.k_batch_stride + (uint64_t)r0 * krs;
            vb = vg + (int64_t)pg * p.v_batch_stride + (uint64_t)r0 * vrs;
        } else {
            kb = kg + (uint64_t)(uint32_t)t0 * krs;
            vb = vg + (uint64_t)(uint32_t)t0 * vrs;
        }
        const uint32_t last = (uint32_t)(L - 1 - t0);            // rows behind the sequence re-read its last row (never-written slots may hold anything) and get p = 0
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const uint32_t row = min((uint32_t)(pi * R + r), last);
            kk[SG][pi] = make_uint4(0, 0, 0, 0);
            vv[SG][pi] = make_uint4(0, 0, 0, 0);
            if (act) {
                kk[SG][pi] = *reinterpret_cast<const uint4 *>(kb + (__umul24(row, krs) + (uint32_t)c * 8u));
                vv[SG][pi] = *reinterpret_cast<const uint4 *>(vb + (__umul24(row, vrs) + (uint32_t)c * 8u));
            }
        }
    };
    auto compute_tile = [&](auto sel, int t0) {
        constexpr int SG = decltype(sel)::value;
        constexpr int NPP = (NP + 1) / 2;
        uint32_t pk[NQ][NPP];                                     // the tile's probabilities, rounded to the storage type (softmax.h:65-185), two per register
#pragma unroll
        for (int gq = 0; gq < NQ; ++gq) {
#pragma unroll
            for (int i = 0; i < NPP; ++i) pk[gq][i] = 0;
            if (gq >= nq) continue;                              // wave-uniform
            float sc[NP], mx = -INFINITY;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                float acc = dot2<T>(kk[SG][pi].x, qv[gq].x, 0.f);
                acc = dot2<T>(kk[SG][pi].y, qv[gq].y, acc);
                acc = dot2<T>(kk[SG][pi].z, qv[gq].z, acc);
                acc = dot2<T>(kk[SG][pi].w, qv[gq].w, acc);
                acc = grow_sum<CP>(acc);
                const int tok = t0 + pi * R + r;
                sc[pi] = tok < L ? acc * p.scale_log2 - slope[gq] * (float)(L - 1 - tok) : -INFINITY;   // ALiBi: mask.h:179-186 with one query row at position L - 1
                mx = fmaxf(mx, sc[pi]);
            }
            mx = gwave_max_over_rows<CP>(mx);
            const float m_new = fmaxf(m[gq], mx);
            const float ms = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m[gq] - ms);
            m[gq] = m_new;
            l[gq] *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[gq][e] *= alpha;
#pragma unroll
            for (int pi = 0; pi < NP; pi += 2) {
                const float p0 = __builtin_amdgcn_exp2f(sc[pi] - ms);                       // exp2(-inf) = 0 for the rows behind the sequence
                const float p1 = pi + 1 < NP ? __builtin_amdgcn_exp2f(sc[pi + 1 < NP ? pi + 1 : pi] - ms) : 0.f;
                l[gq] += p0 + p1;
                pk[gq][pi / 2] = gpack<T>(p0, p1);
            }
        }
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {                        // V unpacked once per load instruction, used by every q head of the chunk
            const float v0 = lo_to_f32<T>(vv[SG][pi].x), v1 = hi_to_f32<T>(vv[SG][pi].x), v2 = lo_to_f32<T>(vv[SG][pi].y), v3 = hi_to_f32<T>(vv[SG][pi].y);
            const float v4 = lo_to_f32<T>(vv[SG][pi].z), v5 = hi_to_f32<T>(vv[SG][pi].z), v6 = lo_to_f32<T>(vv[SG][pi].w), v7 = hi_to_f32<T>(vv[SG][pi].w);
#pragma unroll
            for (int gq = 0; gq < NQ; ++gq) {
                const float pr = (pi & 1) ? hi_to_f32<T>(pk[gq][pi / 2]) : lo_to_f32<T>(pk[gq][pi / 2]);
                o[gq][0] += pr * v0; o[gq][1] += pr * v1; o[gq][2] += pr * v2; o[gq][3] += pr * v3;
                o[gq][4] += pr * v4; o[gq][5] += pr * v5; o[gq][6] += pr * v6; o[gq][7] += pr * v7;
            }
        }
    };
    typedef std::integral_constant<int, 0> SetA;
    typedef std::integral_constant<int, DB ? 1 : 0> SetB;
    // this wavefront's tiles, in load order: DB -- pairs of neighbouring tiles, a pair every 32 NW tokens; else every NW-th tile.  The page of load k + 2 is
    // requested when load k is issued (pg_cur: the page of the next load, pg_nxt: of the one after)
    auto tile_at = [&](int kq) { return DB ? 32 * wv + 32 * NW * (kq >> 1) + 16 * (kq & 1) : 16 * (wv + NW * kq); };
    if (tile_at(0) < L) {
        int pg_cur = page_of(tile_at(0)), pg_nxt = page_of(tile_at(1));
        if (DB) {
            load_tile(SetA(), tile_at(0), pg_cur);
            pg_cur = pg_nxt; pg_nxt = page_of(tile_at(2));
            for (int kq = 0; tile_at(kq) < L; kq += 2) {
                const int t0 = tile_at(kq);
                if (t0 + 16 < L) { load_tile(SetB(), t0 + 16, pg_cur); pg_cur = pg_nxt; pg_nxt = page_of(tile_at(kq + 3)); }
                compute_tile(SetA(), t0);
                if (t0 + 16 < L) {
                    const int tn = tile_at(kq + 2);
                    if (tn < L) { load_tile(SetA(), tn, pg_cur); pg_cur = pg_nxt; pg_nxt = page_of(tile_at(kq + 4)); }
                    compute_tile(SetB(), t0 + 16);
                }
            }
        } else {
            for (int kq = 0; tile_at(kq) < L; ++kq) {
                load_tile(SetA(), tile_at(kq), pg_cur);
                pg_cur = pg_nxt; pg_nxt = page_of(tile_at(kq + 2));
                compute_tile(SetA(), tile_at(kq));
            }
        }
    }
    __shared__ float s_part[NW > 1 ? NW - 1 : 1][NQ][CP][10];     // the other wavefronts' pieces: running max, row sum, 8 output elements per chunk lane
#pragma unroll
    for (int gq = 0; gq < NQ; ++gq) {
        if (gq >= nq) continue;
        float lt = l[gq];
#pragma unroll
        for (int off = CP; off < 64; off <<= 1) {
            lt += __shfl_xor(lt, off, 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[gq][e] += __shfl_xor(o[gq][e], off, 64);
        }
        float mt = m[gq];
        if (NW > 1) {                                            // pieces merged in wavefront order by wavefront 0 (LSE-weighted: flash_fwd_kernel.h:1204-1236 in the exp2 domain)
            if (wv > 0 && r == 0) {
                float *dst = s_part[wv - 1][gq][c];
                dst[0] = mt; dst[1] = lt;
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[2 + e] = o[gq][e];
            }
            __syncthreads();
            if (wv == 0) {
                float mm = mt;
#pragma unroll
                for (int w2 = 0; w2 < NW - 1; ++w2) mm = fmaxf(mm, s_part[w2][gq][c][0]);
                const float ms = mm == -INFINITY ? 0.f : mm;
                const float f0 = __builtin_amdgcn_exp2f(mt - ms);
                lt *= f0;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[gq][e] *= f0;
#pragma unroll
                for (int w2 = 0; w2 < NW - 1; ++w2) {
                    const float *src = s_part[w2][gq][c];
                    const float fw = __builtin_amdgcn_exp2f(src[0] - ms);
                    lt += fw * src[1];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[gq][e] += fw * src[2 + e];
                }
                mt = mm;
            }
            __syncthreads();                                     // (s_part is reused by the next q head)
        }
        const bool empty = !(lt > 0.f);                          // no key: O = 0, LSE = +inf (flash_fwd_kernel.h:97-133)
        const float inv = empty ? 0.f : 1.f / lt;
        if (act && r == 0 && wv == 0) {
            uint4 w;
            w.x = gpack<T>(o[gq][0] * inv, o[gq][1] * inv); w.y = gpack<T>(o[gq][2] * inv, o[gq][3] * inv);
            w.z = gpack<T>(o[gq][4] * inv, o[gq][5] * inv); w.w = gpack<T>(o[gq][6] * inv, o[gq][7] * inv);
            *reinterpret_cast<uint4 *>(p.o + (int64_t)b * p.o_batch_stride + (int64_t)(hq0 + gq) * p.o_head_stride + c * 8) = w;
        }
        if (p.lse && threadIdx.x == 0) p.lse[(int64_t)b * p.h + hq0 + gq] = empty ? INFINITY : (mt + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
    }
}

template <typename T, int CP>
static void launch_decode_anyd2(const AttnParams &p, hipStream_t stream) {
    const int g = p.h / p.h_k, nq = g == 1 ? 1 : (g == 2 ? 2 : 4), gchunks = (g + nq - 1) / nq;
    const int64_t units = (int64_t)p.b * p.h_k * gchunks;
    // Several wavefronts per unit pay even when one per unit would fill the chip (B = 256 x 32 heads at d = 96: 0.69 -> 0.75 of HBM with 4): the workgroup's
    // wavefronts walk neighbouring tiles of ONE sequence, and small batches get their parallelism back (B = 8: 0.17 -> 0.41) -- profiles/r05_generic_decode_waves_ab.txt
    int nw = p.seqlen_k >= 1024 ? 4 : (p.seqlen_k >= 256 ? 2 : 1);
    if (p.seqlen_k >= 2048 && units < 768) nw = 8;               // (B = 8 x 32 heads x 4096: 0.40 -> 0.68)
    if (const int v = generic_decode_waves.load()) nw = v == 8 ? 8 : (v == 4 ? 4 : (v == 2 ? 2 : 1));   // A/B runs
    const dim3 grid((unsigned)units);
#define ATOMA_AD2(NQ_) do { if (nw == 1) hipLaunchKernelGGL((attn_decode_anyd2_kernel<T, CP, NQ_, 1>), grid, dim3(64), 0, stream, p, gchunks); \
                            else if (nw == 2) hipLaunchKernelGGL((attn_decode_anyd2_kernel<T, CP, NQ_, 2>), grid, dim3(128), 0, stream, p, gchunks); \
                            else if (nw == 4) hipLaunchKernelGGL((attn_decode_anyd2_kernel<T, CP, NQ_, 4>), grid, dim3(256), 0, stream, p, gchunks); \
                            else hipLaunchKernelGGL((attn_decode_anyd2_kernel<T, CP, NQ_, 8>), grid, dim3(512), 0, stream, p, gchunks); } while (0)
    if (nq == 1) ATOMA_AD2(1); else if (nq == 2) ATOMA_AD2(2); else ATOMA_AD2(4);
#undef ATOMA_AD2
}

static int attn_prefill_tile_choice() {                          // generic_prefill_tile: 64 = the tiled kernels; 16 = only the 16-row one; anything else = the row-per-wavefront kernel
    const int v = generic_prefill_tile.load();
    return v == 64 ? 64 : (v == 16 ? 16 : 0);
}

static bool attn_prefill_tile16_applicable(const AttnParams &p) {
    const int64_t strides = p.q_head_stride | p.k_head_stride | p.v_head_stride | p.o_head_stride | p.q_row_stride | p.o_row_stride | p.k_row_stride | p.v_row_stride |
                            p.q_batch_stride | p.o_batch_stride | p.k_batch_stride | p.v_batch_stride;
    return p.seqlen_q > 1 && p.d >= 16 && p.d <= 256 && p.d % 16 == 0 && p.h % p.h_k == 0 && strides % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(p.q) | reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v) | reinterpret_cast<uintptr_t>(p.o)) & 15u) == 0 &&
           attn_prefill_tile_choice() != 0 &&
           // the tiled kernels keep scores raw and fold the scale into the exp2 argument (max(raw) * scale): positive finite scales only; a
           // zero, negative or non-finite softmax_scale takes the row-per-wavefront kernel, which handles any finite scale as the reference does
           p.scale_log2 > 0.f && p.scale_log2 < INFINITY;
}

static bool attn_decode_anyd_applicable(const AttnParams &p) {
    const int64_t strides = p.q_head_stride | p.k_head_stride | p.v_head_stride | p.o_head_stride | p.k_row_stride | p.v_row_stride | p.q_batch_stride |
                            p.o_batch_stride | p.k_batch_stride | p.v_batch_stride;
    return p.seqlen_q == 1 && p.cu_seqlens_q == nullptr && p.d >= 8 && p.d <= 256 && p.d % 8 == 0 && p.h % p.h_k == 0 && strides % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(p.q) | reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v) | reinterpret_cast<uintptr_t>(p.o)) & 15u) == 0 &&
           generic_decode_stream.load() != 0;                    // (0: the row-per-lane kernel, for A/B runs)
}

void launch_attn_generic(const AttnParams &p, bool is_bf16, hipStream_t stream) {
    if (p.b <= 0 || p.h <= 0 || p.seqlen_q <= 0) return;
    if (attn_decode_anyd_applicable(p)) {
        const int C = p.d / 8;
        if (generic_decode_stream.load() != 1) {                 // 1: the first version of the streaming kernel (A/B runs)
#define ATOMA_ANYD2(CP_) do { if (is_bf16) launch_decode_anyd2<bf16_t, CP_>(p, stream); else launch_decode_anyd2<f16_t, CP_>(p, stream); } while (0)
            if (C <= 4) ATOMA_ANYD2(4); else if (C <= 8) ATOMA_ANYD2(8); else if (C <= 16) ATOMA_ANYD2(16); else ATOMA_ANYD2(32);
#undef ATOMA_ANYD2
            ATOMA_CHECK_LAUNCH("attn_decode_anyd2_kernel");
            return;
        }
        const int g = p.h / p.h_k, gchunks = (g + 3) / 4;
        const dim3 grid((unsigned)((int64_t)p.b * p.h_k * gchunks));
#define ATOMA_ANYD(CP_) do { if (is_bf16) hipLaunchKernelGGL((attn_decode_anyd_kernel<bf16_t, CP_>), grid, dim3(64), 0, stream, p, gchunks); \
                             else hipLaunchKernelGGL((attn_decode_anyd_kernel<f16_t, CP_>), grid, dim3(64), 0, stream, p, gchunks); } while (0)
        if (C <= 4) ATOMA_ANYD(4); else if (C <= 8) ATOMA_ANYD(8); else if (C <= 16) ATOMA_ANYD(16); else ATOMA_ANYD(32);
#undef ATOMA_ANYD
        ATOMA_CHECK_LAUNCH("attn_decode_anyd_kernel");
        return;
    }
    if (attn_prefill_tile16_applicable(p) && p.d % 32 == 0 && p.seqlen_q > 16 && attn_prefill_tile_choice() == 64) {
        if (is_bf16) launch_prefill_tile64<bf16_t>(p, stream);
        else launch_prefill_tile64<f16_t>(p, stream);
        ATOMA_CHECK_LAUNCH("attn_prefill_tile64_kernel");
        return;
    }
    if (attn_prefill_tile16_applicable(p)) {
        const dim3 grid((unsigned)((p.seqlen_q + 15) / 16), (unsigned)p.h, (unsigned)p.b);
#define ATOMA_T16(N_) do { if (p.d == 16 * N_) { if (is_bf16) hipLaunchKernelGGL((attn_prefill_tile16_kernel<bf16_t, N_, true>), grid, dim3(64), 0, stream, p); \
                                                 else hipLaunchKernelGGL((attn_prefill_tile16_kernel<f16_t, N_, true>), grid, dim3(64), 0, stream, p); } \
                           else { if (is_bf16) hipLaunchKernelGGL((attn_prefill_tile16_kernel<bf16_t, N_, false>), grid, dim3(64), 0, stream, p); \
                                  else hipLaunchKernelGGL((attn_prefill_tile16_kernel<f16_t, N_, false>), grid, dim3(64), 0, stream, p); } } while (0)
        switch ((p.d + 31) / 32) {
        case 1: ATOMA_T16(2); break; case 2: ATOMA_T16(4); break; case 3: ATOMA_T16(6); break; case 4: ATOMA_T16(8); break;
        case 5: ATOMA_T16(10); break; case 6: ATOMA_T16(12); break; case 7: ATOMA_T16(14); break; default: ATOMA_T16(16); break;
        }
#undef ATOMA_T16
        ATOMA_CHECK_LAUNCH("attn_prefill_tile16_kernel");
        return;
    }
    // gridDim.y/z <= 65535: heads and batch are far below that in every caller of this path
    dim3 grid((unsigned)p.seqlen_q, (unsigned)p.h, (unsigned)p.b);
    if (is_bf16) hipLaunchKernelGGL(attn_generic_kernel<bf16_t>, grid, dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(attn_generic_kernel<f16_t>, grid, dim3(64), 0, stream, p);
    ATOMA_CHECK_LAUNCH("attn_generic_kernel");
}

}  // namespace atoma
