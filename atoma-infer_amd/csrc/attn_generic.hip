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

static bool attn_decode_anyd_applicable(const AttnParams &p) {
    const int64_t strides = p.q_head_stride | p.k_head_stride | p.v_head_stride | p.o_head_stride | p.k_row_stride | p.v_row_stride | p.q_batch_stride |
                            p.o_batch_stride | p.k_batch_stride | p.v_batch_stride;
    return p.seqlen_q == 1 && p.cu_seqlens_q == nullptr && p.d >= 8 && p.d <= 256 && p.d % 8 == 0 && p.h % p.h_k == 0 && strides % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(p.q) | reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v) | reinterpret_cast<uintptr_t>(p.o)) & 15u) == 0 &&
           getenv("ATOMA_GENERIC_DECODE_STREAM") == nullptr;     // (set to anything: the row-per-lane kernel, for A/B runs)
}

void launch_attn_generic(const AttnParams &p, bool is_bf16, hipStream_t stream) {
    if (p.b <= 0 || p.h <= 0 || p.seqlen_q <= 0) return;
    if (attn_decode_anyd_applicable(p)) {
        const int g = p.h / p.h_k, gchunks = (g + 3) / 4, C = p.d / 8;
        const dim3 grid((unsigned)((int64_t)p.b * p.h_k * gchunks));
#define ATOMA_ANYD(CP_) do { if (is_bf16) hipLaunchKernelGGL((attn_decode_anyd_kernel<bf16_t, CP_>), grid, dim3(64), 0, stream, p, gchunks); \
                             else hipLaunchKernelGGL((attn_decode_anyd_kernel<f16_t, CP_>), grid, dim3(64), 0, stream, p, gchunks); } while (0)
        if (C <= 4) ATOMA_ANYD(4); else if (C <= 8) ATOMA_ANYD(8); else if (C <= 16) ATOMA_ANYD(16); else ATOMA_ANYD(32);
#undef ATOMA_ANYD
        ATOMA_CHECK_LAUNCH("attn_decode_anyd_kernel");
        return;
    }
    // gridDim.y/z <= 65535: heads and batch are far below that in every caller of this path
    dim3 grid((unsigned)p.seqlen_q, (unsigned)p.h, (unsigned)p.b);
    if (is_bf16) hipLaunchKernelGGL(attn_generic_kernel<bf16_t>, grid, dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(attn_generic_kernel<f16_t>, grid, dim3(64), 0, stream, p);
    ATOMA_CHECK_LAUNCH("attn_generic_kernel");
}

}  // namespace atoma
