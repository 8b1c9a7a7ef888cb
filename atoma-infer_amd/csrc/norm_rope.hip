// RMSNorm and RoPE for gfx950 -- the two small per-layer ops next to attention.
//
// The reference does not own these kernels: it calls Candle
//   RMSNorm  /root/reference/models/src/llama.rs:402,408,474  -> candle_nn::ops::rms_norm
//   RoPE     /root/reference/models/src/llama.rs:218-251      -> candle_nn::rotary_emb::rope
// (candle 0.9.2-alpha.1, not vendored; semantics restated in oracle/norm_rope_oracle.py).
// Both are HBM-bound streaming ops: 16-byte vector access, one pass over the data.
//   RMSNorm bytes = 2*T*hidden*2 + hidden*2 ;  RoPE bytes = 2*T*(h+h_k)*d*2 + 2*T*(d/2)*2
// The reference wraps RoPE in 4 transposes + contiguous() copies per layer
// (llama.rs:273-303) because Candle's kernel wants [1, h, T, d]; this kernel works on the
// [T, h, d] layout the projections produce and fuses the cos/sin index_select.
#include "common.h"
#include "norm_shared.h"
#include "linear_params.h"

#include <atomic>
#include <math.h>
#include <string.h>

namespace atoma {

// ------------------------------------------------------------------------------------------
// RMSNorm: one 256-thread workgroup per row; the row stays in registers between the
// sum-of-squares pass and the scale pass (ITERS x 8 elements per thread).  The arithmetic itself
// lives in norm_shared.h (shared with the projection kernel that normalises its own input).
// ------------------------------------------------------------------------------------------

// ADD: x = round(a + b) first (the residual add that precedes every RMSNorm of a decoder layer, llama.rs:404,409 -> 402,408),
// written to `sum` as the separate add kernel would and normalised from the rounded values: bit-identical to the two ops.
template <typename T, int ITERS, bool ADD = false>
__global__ void __launch_bounds__(NORM_THREADS)
rms_norm_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, uint16_t *__restrict__ y,
                int hidden, int64_t x_row_stride, int64_t y_row_stride, float eps, const uint16_t *__restrict__ b = nullptr,
                uint16_t *__restrict__ sum = nullptr, int64_t b_row_stride = 0, int64_t sum_row_stride = 0) {
    __shared__ float red[NORM_THREADS / 64];
    const int64_t row = blockIdx.x;
    const uint16_t *xr = x + row * x_row_stride;
    uint16_t *yr = y + row * y_row_stride;
    const int nvec = hidden >> 3;
    uint4 xv[ITERS];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = it * NORM_THREADS + threadIdx.x;
        xv[it] = make_uint4(0, 0, 0, 0);
        if (i < nvec) xv[it] = reinterpret_cast<const uint4 *>(xr)[i];
        if constexpr (ADD) {
            if (i < nvec) {
                float fa[8], fb[8];
                unpack8<T>(xv[it], fa);
                unpack8<T>(reinterpret_cast<const uint4 *>(b + row * b_row_stride)[i], fb);
#pragma unroll
                for (int e = 0; e < 8; ++e) fa[e] += fb[e];
                xv[it] = pack8<T>(fa);
                reinterpret_cast<uint4 *>(sum + row * sum_row_stride)[i] = xv[it];
            }
        }
        ss = norm_sumsq8<T>(xv[it], ss);
    }
    ss = norm_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float redr[NORM_THREADS / 64];
#pragma unroll
    for (int i = 0; i < NORM_THREADS / 64; ++i) redr[i] = red[i];
    const float scale = norm_scale(redr, hidden, eps);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = it * NORM_THREADS + threadIdx.x;
        if (i < nvec) reinterpret_cast<uint4 *>(yr)[i] = norm_apply8<T>(xv[it], reinterpret_cast<const uint4 *>(w)[i], scale);
    }
}

// any hidden size / alignment: two passes over the row through L2
template <typename T>
__global__ void __launch_bounds__(NORM_THREADS)
rms_norm_generic_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, uint16_t *__restrict__ y,
                        int hidden, int64_t x_row_stride, int64_t y_row_stride, float eps) {
    __shared__ float red[NORM_THREADS / 64];
    const int64_t row = blockIdx.x;
    const uint16_t *xr = x + row * x_row_stride;
    uint16_t *yr = y + row * y_row_stride;
    float ss = 0.f;
    for (int i = threadIdx.x; i < hidden; i += NORM_THREADS) {
        const float f = lo_to_f32<T>(xr[i]);
        ss += f * f;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_THREADS / 64; ++i) tot += red[i];
    const float scale = 1.0f / sqrtf(tot / (float)hidden + eps);
    for (int i = threadIdx.x; i < hidden; i += NORM_THREADS)
        yr[i] = (uint16_t)f32_to_bits<T>((scale * lo_to_f32<T>(xr[i])) * lo_to_f32<T>(w[i]));
}

template <typename T>
static void launch_rms_norm(const void *x, const void *w, void *y, int64_t rows, int64_t hidden, int64_t xs,
                            int64_t ys, float eps, hipStream_t stream) {
    auto x16 = static_cast<const uint16_t *>(x);
    auto w16 = static_cast<const uint16_t *>(w);
    auto y16 = static_cast<uint16_t *>(y);
    const bool vec = hidden % 8 == 0 && xs % 8 == 0 && ys % 8 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
    const int64_t iters = cdiv(hidden / 8, NORM_THREADS);
    dim3 grid((unsigned)rows), block(NORM_THREADS);
    if (vec && iters <= 1)
        hipLaunchKernelGGL((rms_norm_kernel<T, 1>), grid, block, 0, stream, x16, w16, y16, (int)hidden, xs, ys, eps);
    else if (vec && iters <= 2)
        hipLaunchKernelGGL((rms_norm_kernel<T, 2>), grid, block, 0, stream, x16, w16, y16, (int)hidden, xs, ys, eps);
    else if (vec && iters <= 4)
        hipLaunchKernelGGL((rms_norm_kernel<T, 4>), grid, block, 0, stream, x16, w16, y16, (int)hidden, xs, ys, eps);
    else if (vec && iters <= 8)
        hipLaunchKernelGGL((rms_norm_kernel<T, 8>), grid, block, 0, stream, x16, w16, y16, (int)hidden, xs, ys, eps);
    else
        hipLaunchKernelGGL((rms_norm_generic_kernel<T>), grid, block, 0, stream, x16, w16, y16, (int)hidden, xs, ys, eps);
    ATOMA_CHECK_LAUNCH("rms_norm");
}

// atoma_set_option("rope_table_rows", n): the number of rows of the cos / sin tables the caller built (atoma_rope_table's
// max_pos).  The reference's index_select fails on a position beyond the table; these kernels cannot raise from the device,
// so with the option set they read the table's LAST row for such a position instead of memory behind it (0 = unchecked,
// the default: the FFI carries no table length).
std::atomic<int64_t> rope_table_rows{0};
__device__ __forceinline__ int64_t rope_pos(int64_t pos, int64_t table_rows) {
    return table_rows > 0 ? (pos < 0 ? 0 : (pos >= table_rows ? table_rows - 1 : pos)) : pos;
}

// ------------------------------------------------------------------------------------------
// RoPE (rotate-half): thread = (token, head, 8-element chunk of the first half).
// PER_OP = Candle's arithmetic in the tensor dtype: every product and the sum are rounded.
// ------------------------------------------------------------------------------------------
template <typename T, bool PER_OP>
__device__ __forceinline__ void rope_chunk_vals(const float (&x1)[8], const float (&x2)[8], uint16_t *y, const uint16_t *cosr, const uint16_t *sinr,
                                                int half, int c, uint16_t *y_copy = nullptr) {
    // No contraction here: for f16 the compiler narrows the f32 expressions below to half
    // arithmetic (legitimately -- f32 carries 2p+2 bits) and would then fuse mul+sub into one
    // v_fma_f16, i.e. drop exactly the intermediate rounding PER_OP exists to reproduce.
#pragma clang fp contract(off)
    float cs[8], sn[8], y1[8], y2[8];
    unpack8<T>(*reinterpret_cast<const uint4 *>(cosr + c * 8), cs);
    unpack8<T>(*reinterpret_cast<const uint4 *>(sinr + c * 8), sn);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if constexpr (PER_OP) {
            y1[e] = round_through<T>(x1[e] * cs[e]) - round_through<T>(x2[e] * sn[e]);
            y2[e] = round_through<T>(x1[e] * sn[e]) + round_through<T>(x2[e] * cs[e]);
        } else {
            y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
            y2[e] = x1[e] * sn[e] + x2[e] * cs[e];
        }
    }
    const uint4 lo = pack8<T>(y1), hi = pack8<T>(y2);
    *reinterpret_cast<uint4 *>(y + c * 8) = lo;
    *reinterpret_cast<uint4 *>(y + half + c * 8) = hi;
    if (y_copy) {   // second destination of the same rotated chunk (the KV cache page)
        *reinterpret_cast<uint4 *>(y_copy + c * 8) = lo;
        *reinterpret_cast<uint4 *>(y_copy + half + c * 8) = hi;
    }
}
template <typename T, bool PER_OP>
__device__ __forceinline__ void rope_chunk(const uint16_t *x, uint16_t *y, const uint16_t *cosr, const uint16_t *sinr,
                                           int half, int c, uint16_t *y_copy = nullptr) {
    float x1[8], x2[8];
    unpack8<T>(*reinterpret_cast<const uint4 *>(x + c * 8), x1);
    unpack8<T>(*reinterpret_cast<const uint4 *>(x + half + c * 8), x2);
    rope_chunk_vals<T, PER_OP>(x1, x2, y, cosr, sinr, half, c, y_copy);
}

// Two tensors (q and k) in one launch; nb heads == 0 disables the second.
template <typename T, bool PER_OP>
__global__ void __launch_bounds__(256)
rope_kernel(const uint16_t *__restrict__ xa, uint16_t *__restrict__ ya, int heads_a, int64_t xa_ts, int64_t xa_hs,
            int64_t ya_ts, int64_t ya_hs, const uint16_t *__restrict__ xb, uint16_t *__restrict__ yb, int heads_b,
            int64_t xb_ts, int64_t xb_hs, int64_t yb_ts, int64_t yb_hs, const uint16_t *__restrict__ cos_t,
            const uint16_t *__restrict__ sin_t, const int64_t *__restrict__ positions, int head_dim, int64_t table_rows) {
    const int64_t t = blockIdx.x;
    const int half = head_dim >> 1;
    const int cpr = half >> 3;  // chunks per (token, head)
    const int64_t pos = rope_pos(positions[t], table_rows);
    const uint16_t *cosr = cos_t + pos * half, *sinr = sin_t + pos * half;
    const int total = (heads_a + heads_b) * cpr;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int head = i / cpr, c = i - head * cpr;
        if (head < heads_a)
            rope_chunk<T, PER_OP>(xa + t * xa_ts + head * xa_hs, ya + t * ya_ts + head * ya_hs, cosr, sinr, half, c);
        else
            rope_chunk<T, PER_OP>(xb + t * xb_ts + (head - heads_a) * xb_hs, yb + t * yb_ts + (head - heads_a) * yb_hs,
                                  cosr, sinr, half, c);
    }
}

static int launch_rope(const void *xa, void *ya, int64_t ha, int64_t xa_ts, int64_t xa_hs, int64_t ya_ts, int64_t ya_hs,
                       const void *xb, void *yb, int64_t hb, int64_t xb_ts, int64_t xb_hs, int64_t yb_ts, int64_t yb_hs,
                       const void *cos_t, const void *sin_t, const int64_t *positions, int64_t T, int64_t d, int dtype,
                       int per_op, hipStream_t stream) {
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("rope: dtype must be f16 or bf16"); return -1; }
    if (d % 16 != 0) { set_error("rope: head_dim must be a multiple of 16"); return -1; }
    const int64_t strides[] = {xa_ts, xa_hs, ya_ts, ya_hs, xb_ts, xb_hs, yb_ts, yb_hs};
    for (int64_t s : strides)
        if (s % 8 != 0) { set_error("rope: strides must be multiples of 8 elements"); return -1; }
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(xa) | reinterpret_cast<uintptr_t>(ya) | reinterpret_cast<uintptr_t>(xb) |
                           reinterpret_cast<uintptr_t>(yb) | reinterpret_cast<uintptr_t>(cos_t) | reinterpret_cast<uintptr_t>(sin_t);
    if (ptrs & 15u) { set_error("rope: tensors must be 16-byte aligned"); return -1; }
    if (T <= 0 || ha + hb <= 0) return 0;
    const int64_t work = (ha + hb) * (d / 16);
    int threads = (int)(cdiv(work, 64) * 64);
    threads = threads > 256 ? 256 : threads;
#define ATOMA_ROPE_LAUNCH(TT, PO)                                                                                     \
    hipLaunchKernelGGL((rope_kernel<TT, PO>), dim3((unsigned)T), dim3(threads), 0, stream,                            \
                       static_cast<const uint16_t *>(xa), static_cast<uint16_t *>(ya), (int)ha, xa_ts, xa_hs, ya_ts,  \
                       ya_hs, static_cast<const uint16_t *>(xb), static_cast<uint16_t *>(yb), (int)hb, xb_ts, xb_hs,  \
                       yb_ts, yb_hs, static_cast<const uint16_t *>(cos_t), static_cast<const uint16_t *>(sin_t),      \
                       positions, (int)d, rope_table_rows.load())
    if (dtype == ATOMA_BF16) { if (per_op) ATOMA_ROPE_LAUNCH(bf16_t, true); else ATOMA_ROPE_LAUNCH(bf16_t, false); }
    else { if (per_op) ATOMA_ROPE_LAUNCH(f16_t, true); else ATOMA_ROPE_LAUNCH(f16_t, false); }
#undef ATOMA_ROPE_LAUNCH
    return ATOMA_CHECK_LAUNCH("rope") ? 0 : -1;
}

// ------------------------------------------------------------------------------------------
// Fused RoPE(q, k) + reshape_and_cache_flash(k, v): one pass over a step's q / k / v instead of two
// launches.  q and k are rotated in place (the prefill attention reads them as
// tensors, llama.rs:273-303), the rotated k and v also go to their cache slot; slot < 0 = padding token
// (cache_kernels.cu:306-310).  Block = one token; a thread handles one 16-byte chunk pair (RoPE) or one
// 16-byte chunk (v copy).
// ------------------------------------------------------------------------------------------
template <typename T, bool PER_OP>
__global__ void __launch_bounds__(256)
rope_cache_kernel(uint16_t *__restrict__ q, uint16_t *__restrict__ k, const uint16_t *__restrict__ v,
                  uint16_t *__restrict__ k_cache, uint16_t *__restrict__ v_cache, const int64_t *__restrict__ slot_mapping,
                  const uint16_t *__restrict__ cos_t, const uint16_t *__restrict__ sin_t,
                  const int64_t *__restrict__ positions, int heads_q, int heads_kv, int head_dim, int64_t q_ts, int64_t k_ts,
                  int64_t v_ts, int64_t block_stride, int page_size, int64_t table_rows) {
    const int64_t t = blockIdx.x;
    const int half = head_dim >> 1, cpr = half >> 3, vpr = head_dim >> 3;
    const int64_t pos = rope_pos(positions[t], table_rows), slot = slot_mapping[t];
    const uint16_t *cosr = cos_t + pos * half, *sinr = sin_t + pos * half;
    const int64_t row = slot >= 0 ? (slot / page_size) * block_stride + (slot % page_size) * (int64_t)heads_kv * head_dim : 0;
    const int n_rope = (heads_q + heads_kv) * cpr, total = n_rope + (slot >= 0 ? heads_kv * vpr : 0);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        if (i < n_rope) {
            const int head = i / cpr, c = i - head * cpr;
            if (head < heads_q) {
                uint16_t *x = q + t * q_ts + (int64_t)head * head_dim;
                rope_chunk<T, PER_OP>(x, x, cosr, sinr, half, c);
            } else {
                const int hk = head - heads_q;
                uint16_t *x = k + t * k_ts + (int64_t)hk * head_dim;
                rope_chunk<T, PER_OP>(x, x, cosr, sinr, half, c, slot >= 0 ? k_cache + row + (int64_t)hk * head_dim : nullptr);
            }
        } else {
            const int j = i - n_rope, hk = j / vpr, c = j - hk * vpr;
            *reinterpret_cast<uint4 *>(v_cache + row + (int64_t)hk * head_dim + c * 8) =
                *reinterpret_cast<const uint4 *>(v + t * v_ts + (int64_t)hk * head_dim + c * 8);
        }
    }
}

// The same pass fed by the fp32 K-split partials of the q/k/v projection instead of its rounded output: the merge kernel of the
// projection, RoPE and the cache write in ONE launch (the tensor-parallel rank's 1280-row q/k/v shard is split 8 ways over K; its
// merge used to be a launch of its own in front of this one).  Sum over the splits in split order, ONE rounding to the storage
// dtype -- exactly what linear_reduce_kernel writes -- then the arithmetic of rope_cache_kernel on those values: bit-identical to
// projection + atoma_rope_qk_cache.  partial [splits][tokens][width] fp32, width = (heads_q + 2 heads_kv) head_dim; out [tokens, out_ts].
template <typename T, bool PER_OP>
__global__ void __launch_bounds__(256)
qkv_partials_rope_cache_kernel(const float *__restrict__ partial, int splits, int64_t tokens, uint16_t *__restrict__ out, int64_t out_ts,
                               uint16_t *__restrict__ k_cache, uint16_t *__restrict__ v_cache, const int64_t *__restrict__ slot_mapping,
                               const uint16_t *__restrict__ cos_t, const uint16_t *__restrict__ sin_t, const int64_t *__restrict__ positions,
                               int heads_q, int heads_kv, int head_dim, int64_t block_stride, int page_size, int64_t table_rows) {
    const int64_t t = blockIdx.x;
    const int half = head_dim >> 1, cpr = half >> 3, vpr = head_dim >> 3;
    const int width = (heads_q + 2 * heads_kv) * head_dim;
    const int64_t pos = rope_pos(positions[t], table_rows), slot = slot_mapping[t];
    const uint16_t *cosr = cos_t + pos * half, *sinr = sin_t + pos * half;
    const int64_t row = slot >= 0 ? (slot / page_size) * block_stride + (slot % page_size) * (int64_t)heads_kv * head_dim : 0;
    const int n_rope = (heads_q + heads_kv) * cpr, total = n_rope + heads_kv * vpr;
    const int64_t plane = tokens * width;
    auto sum8 = [&](int n, float (&v)[8]) {        // 8 consecutive outputs of token t: splits added in order, rounded once
        const float *src = partial + t * width + n;
        float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
        for (int s0 = 1; s0 < splits; s0 += 8) {   // up to 8 splits' loads in flight, then the adds in split order
            float4 c[8], d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < splits) {
                    c[j] = *reinterpret_cast<const float4 *>(src + (s0 + j) * plane);
                    d[j] = *reinterpret_cast<const float4 *>(src + (s0 + j) * plane + 4);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < splits) {
                    a.x += c[j].x; a.y += c[j].y; a.z += c[j].z; a.w += c[j].w;
                    b.x += d[j].x; b.y += d[j].y; b.z += d[j].z; b.w += d[j].w;
                }
        }
        v[0] = round_through<T>(a.x); v[1] = round_through<T>(a.y); v[2] = round_through<T>(a.z); v[3] = round_through<T>(a.w);
        v[4] = round_through<T>(b.x); v[5] = round_through<T>(b.y); v[6] = round_through<T>(b.z); v[7] = round_through<T>(b.w);
    };
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        if (i < n_rope) {
            const int head = i / cpr, c = i - head * cpr;          // heads_q q heads, then the k heads: consecutive in the projection's output
            float x1[8], x2[8];
            sum8(head * head_dim + c * 8, x1);
            sum8(head * head_dim + half + c * 8, x2);
            uint16_t *y = out + t * out_ts + (int64_t)head * head_dim;
            const int hk = head - heads_q;
            rope_chunk_vals<T, PER_OP>(x1, x2, y, cosr, sinr, half, c, hk >= 0 && slot >= 0 ? k_cache + row + (int64_t)hk * head_dim : nullptr);
        } else {
            const int j = i - n_rope, hk = j / vpr, c = j - hk * vpr;
            float v[8];
            const int n = (heads_q + heads_kv + hk) * head_dim + c * 8;
            sum8(n, v);
            const uint4 packed = pack8<T>(v);
            *reinterpret_cast<uint4 *>(out + t * out_ts + n) = packed;
            if (slot >= 0) *reinterpret_cast<uint4 *>(v_cache + row + (int64_t)hk * head_dim + c * 8) = packed;
        }
    }
}

// host-side f32 -> storage rounding for the table builder
static uint16_t host_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static uint16_t host_f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}

}  // namespace atoma

extern "C" {

int atoma_rms_norm(const void *x, const void *weight, void *y, int64_t rows, int64_t hidden, int64_t x_row_stride,
                   int64_t y_row_stride, float eps, int dtype, void *stream) {
    atoma::clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { atoma::set_error("rms_norm: dtype must be f16 or bf16"); return -1; }
    if (rows <= 0 || hidden <= 0) return 0;
    auto s = static_cast<hipStream_t>(stream);
    if (dtype == ATOMA_BF16) atoma::launch_rms_norm<atoma::bf16_t>(x, weight, y, rows, hidden, x_row_stride, y_row_stride, eps, s);
    else atoma::launch_rms_norm<atoma::f16_t>(x, weight, y, rows, hidden, x_row_stride, y_row_stride, eps, s);
    return atoma::has_error() ? -1 : 0;
}

int atoma_add_rms_norm(const void *a, const void *b, const void *weight, void *sum, void *y, int64_t rows, int64_t hidden,
                       int64_t a_row_stride, int64_t b_row_stride, int64_t sum_row_stride, int64_t y_row_stride, float eps, int dtype,
                       void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("add_rms_norm: dtype must be f16 or bf16"); return -1; }
    if (hidden <= 0 || hidden % 8 || hidden > 8 * 8 * NORM_THREADS) { set_error("add_rms_norm: hidden must be a multiple of 8, at most 16384"); return -1; }
    if (a_row_stride % 8 || b_row_stride % 8 || sum_row_stride % 8 || y_row_stride % 8) { set_error("add_rms_norm: row strides must be multiples of 8 elements"); return -1; }
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(sum) |
         reinterpret_cast<uintptr_t>(y)) & 15u) { set_error("add_rms_norm: tensors must be 16-byte aligned"); return -1; }
    if (rows <= 0) return 0;
    auto s = static_cast<hipStream_t>(stream);
    auto a16 = static_cast<const uint16_t *>(a), b16 = static_cast<const uint16_t *>(b), w16 = static_cast<const uint16_t *>(weight);
    auto s16 = static_cast<uint16_t *>(sum), y16 = static_cast<uint16_t *>(y);
    const int64_t iters = cdiv(hidden / 8, NORM_THREADS);
    const dim3 grid((unsigned)rows), block(NORM_THREADS);
#define ATOMA_ARN(T_, IT_) hipLaunchKernelGGL((rms_norm_kernel<T_, IT_, true>), grid, block, 0, s, a16, w16, y16, (int)hidden, a_row_stride, y_row_stride, eps, b16, s16, b_row_stride, sum_row_stride)
#define ATOMA_ARN_T(T_) do { if (iters <= 1) ATOMA_ARN(T_, 1); else if (iters <= 2) ATOMA_ARN(T_, 2); else if (iters <= 4) ATOMA_ARN(T_, 4); else ATOMA_ARN(T_, 8); } while (0)
    if (dtype == ATOMA_BF16) ATOMA_ARN_T(bf16_t); else ATOMA_ARN_T(f16_t);
#undef ATOMA_ARN_T
#undef ATOMA_ARN
    return ATOMA_CHECK_LAUNCH("add_rms_norm") ? 0 : -1;
}

int atoma_rope(const void *x, void *y, const void *cos_table, const void *sin_table, const int64_t *positions,
               int64_t num_tokens, int64_t num_heads, int64_t head_dim, int64_t x_token_stride, int64_t x_head_stride,
               int64_t y_token_stride, int64_t y_head_stride, int dtype, int per_op_rounding, void *stream) {
    atoma::clear_error();
    return atoma::launch_rope(x, y, num_heads, x_token_stride, x_head_stride, y_token_stride, y_head_stride, nullptr,
                              nullptr, 0, 0, 0, 0, 0, cos_table, sin_table, positions, num_tokens, head_dim, dtype,
                              per_op_rounding, static_cast<hipStream_t>(stream));
}

int atoma_rope_qk(void *q, void *k, const void *cos_table, const void *sin_table, const int64_t *positions,
                  int64_t num_tokens, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                  int64_t q_token_stride, int64_t k_token_stride, int dtype, int per_op_rounding, void *stream) {
    atoma::clear_error();
    return atoma::launch_rope(q, q, num_q_heads, q_token_stride, head_dim, q_token_stride, head_dim, k, k, num_kv_heads,
                              k_token_stride, head_dim, k_token_stride, head_dim, cos_table, sin_table, positions,
                              num_tokens, head_dim, dtype, per_op_rounding, static_cast<hipStream_t>(stream));
}

int atoma_rope_qk_cache(void *q, void *k, const void *v, void *k_cache, void *v_cache, const int64_t *slot_mapping,
                        const void *cos_table, const void *sin_table, const int64_t *positions, int64_t num_tokens,
                        int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride,
                        int64_t k_token_stride, int64_t v_token_stride, int64_t block_stride, int64_t page_size, int dtype,
                        int per_op_rounding, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("rope_qk_cache: dtype must be f16 or bf16"); return -1; }
    if (head_dim % 16 != 0) { set_error("rope_qk_cache: head_dim must be a multiple of 16"); return -1; }
    if (page_size <= 0) { set_error("rope_qk_cache: page_size must be positive"); return -1; }
    for (int64_t st : {q_token_stride, k_token_stride, v_token_stride, block_stride})
        if (st % 8 != 0) { set_error("rope_qk_cache: strides must be multiples of 8 elements"); return -1; }
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                           reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) |
                           reinterpret_cast<uintptr_t>(cos_table) | reinterpret_cast<uintptr_t>(sin_table);
    if (ptrs & 15u) { set_error("rope_qk_cache: tensors must be 16-byte aligned"); return -1; }
    if (num_tokens <= 0 || num_q_heads + num_kv_heads <= 0) return 0;
    const int64_t work = (num_q_heads + num_kv_heads) * (head_dim / 16) + num_kv_heads * (head_dim / 8);
    int threads = (int)(cdiv(work, 64) * 64);
    threads = threads > 256 ? 256 : threads;
#define ATOMA_RC_LAUNCH(TT, PO)                                                                                          \
    hipLaunchKernelGGL((rope_cache_kernel<TT, PO>), dim3((unsigned)num_tokens), dim3(threads), 0,                       \
                       static_cast<hipStream_t>(stream), static_cast<uint16_t *>(q), static_cast<uint16_t *>(k),         \
                       static_cast<const uint16_t *>(v), static_cast<uint16_t *>(k_cache), static_cast<uint16_t *>(v_cache), \
                       slot_mapping, static_cast<const uint16_t *>(cos_table), static_cast<const uint16_t *>(sin_table), \
                       positions, (int)num_q_heads, (int)num_kv_heads, (int)head_dim, q_token_stride, k_token_stride,    \
                       v_token_stride, block_stride, (int)page_size, rope_table_rows.load())
    if (dtype == ATOMA_BF16) { if (per_op_rounding) ATOMA_RC_LAUNCH(bf16_t, true); else ATOMA_RC_LAUNCH(bf16_t, false); }
    else { if (per_op_rounding) ATOMA_RC_LAUNCH(f16_t, true); else ATOMA_RC_LAUNCH(f16_t, false); }
#undef ATOMA_RC_LAUNCH
    return ATOMA_CHECK_LAUNCH("rope_qk_cache") ? 0 : -1;
}

// q/k/v projection -> RoPE(q, k) -> KV-cache write (llama.rs:269-271 -> 273-303 -> cache_manager.rs:404-535) behind one entry:
// qkv_out [batch, out_row_stride] receives the projection's output with q and k rotated (what atoma_linear_decode followed by
// atoma_rope_qk_cache on q = qkv_out, k = qkv_out + h.d, v = qkv_out + (h + h_k).d leave there), bit for bit.  When the projection
// splits K over more than two workgroups (matrices with few rows: a tensor-parallel shard) its fp32 partials are merged by the
// RoPE / cache kernel itself: two launches instead of three.  Otherwise the two ops run one after the other.
int atoma_linear_decode_qkv_rope_cache(const void *x, const void *w_qkv, void *qkv_out, void *k_cache, void *v_cache, const int64_t *slot_mapping,
                                       const void *cos_table, const void *sin_table, const int64_t *positions, int64_t batch, int64_t in_features,
                                       int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t x_row_stride, int64_t w_row_stride,
                                       int64_t out_row_stride, int64_t block_stride, int64_t page_size, int dtype, int per_op_rounding, void *stream) {
    using namespace atoma;
    clear_error();
    const int64_t width = (num_q_heads + 2 * num_kv_heads) * head_dim;
    auto two_ops = [&]() {
        if (atoma_linear_decode(x, w_qkv, qkv_out, batch, in_features, width, x_row_stride, w_row_stride, out_row_stride, dtype, stream) != 0) return -1;
        auto *o = static_cast<uint16_t *>(qkv_out);
        return atoma_rope_qk_cache(o, o + num_q_heads * head_dim, o + (num_q_heads + num_kv_heads) * head_dim, k_cache, v_cache, slot_mapping, cos_table,
                                   sin_table, positions, batch, num_q_heads, num_kv_heads, head_dim, out_row_stride, out_row_stride, out_row_stride,
                                   block_stride, page_size, dtype, per_op_rounding, stream);
    };
    // the fused route: exactly the checks of the two entry points that matter for it; anything unusual takes the two ops (and their messages)
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_qkv) | reinterpret_cast<uintptr_t>(qkv_out) |
                           reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) | reinterpret_cast<uintptr_t>(cos_table) |
                           reinterpret_cast<uintptr_t>(sin_table);
    const bool plain = (dtype == ATOMA_F16 || dtype == ATOMA_BF16) && batch > 16 && batch <= 256 && num_q_heads > 0 && num_kv_heads > 0 &&
                       head_dim > 0 && head_dim % 16 == 0 && in_features > 0 && in_features % 128 == 0 && page_size > 0 && !(ptrs & 15u) &&
                       x_row_stride >= in_features && w_row_stride >= in_features && out_row_stride >= width &&
                       !(x_row_stride % 8 || w_row_stride % 8 || out_row_stride % 8 || block_stride % 8) && slot_mapping && positions &&
                       // a null pointer passes the alignment test above: the two ops have the messages for those; the tile
                       // kernel's cache write assumes the reference's page layout [page_size][heads_kv][head_dim]
                       x && w_qkv && qkv_out && k_cache && v_cache && cos_table && sin_table && block_stride >= page_size * num_kv_heads * head_dim;
    if (!plain) return two_ops();
    LinearParams p{};
    p.x = static_cast<const uint16_t *>(x);
    p.w = static_cast<const uint16_t *>(w_qkv);
    p.y = static_cast<uint16_t *>(qkv_out);
    p.x_row_stride = x_row_stride; p.w_row_stride = w_row_stride; p.y_row_stride = out_row_stride;
    p.batch = (int)batch; p.n = (int)width; p.k = (int)in_features;
    const auto s = static_cast<hipStream_t>(stream);
    if (batch > 64) {                                          // 65..256 rows: linear_wide_kernel, same epilogue
        if (!linear_wide_can_rope(p, (int)head_dim)) return two_ops();
        const int rc = launch_linear_wide_rope(p, dtype, s, static_cast<const uint16_t *>(cos_table), static_cast<const uint16_t *>(sin_table), positions,
                                               slot_mapping, static_cast<uint16_t *>(k_cache), static_cast<uint16_t *>(v_cache), block_stride,
                                               rope_table_rows.load(), (int)num_q_heads, (int)num_kv_heads, (int)head_dim, (int)page_size, per_op_rounding);
        return rc <= 0 ? rc : two_ops();
    }
    if (linear_tile_can_rope(p, (int)head_dim)) {              // ONE launch: RoPE and the cache write are the projection's epilogue
        const int rc = launch_linear_tile_rope(p, dtype, s, static_cast<const uint16_t *>(cos_table), static_cast<const uint16_t *>(sin_table), positions,
                                               slot_mapping, static_cast<uint16_t *>(k_cache), static_cast<uint16_t *>(v_cache), block_stride,
                                               rope_table_rows.load(), (int)num_q_heads, (int)num_kv_heads, (int)head_dim, (int)page_size, per_op_rounding);
        if (rc <= 0) return rc;
    }
    if (!linear_tile_leaves_partials(p, dtype)) return two_ops();
    const int rc = launch_linear_tile(p, dtype, s);
    if (rc != 0 || !p.partial) { if (rc < 0) return -1; set_error("linear_decode_qkv_rope_cache: the projection plan changed under the call"); return -1; }
#define ATOMA_QRC(TT, PO)                                                                                                                     \
    hipLaunchKernelGGL((qkv_partials_rope_cache_kernel<TT, PO>), dim3((unsigned)batch), dim3(256), 0, s, p.partial, p.splits, batch,          \
                       static_cast<uint16_t *>(qkv_out), out_row_stride, static_cast<uint16_t *>(k_cache), static_cast<uint16_t *>(v_cache),   \
                       slot_mapping, static_cast<const uint16_t *>(cos_table), static_cast<const uint16_t *>(sin_table), positions,           \
                       (int)num_q_heads, (int)num_kv_heads, (int)head_dim, block_stride, (int)page_size, rope_table_rows.load())
    if (dtype == ATOMA_BF16) { if (per_op_rounding) ATOMA_QRC(bf16_t, true); else ATOMA_QRC(bf16_t, false); }
    else { if (per_op_rounding) ATOMA_QRC(f16_t, true); else ATOMA_QRC(f16_t, false); }
#undef ATOMA_QRC
    return ATOMA_CHECK_LAUNCH("linear_decode_qkv_rope_cache") ? 0 : -1;
}

// models/src/llama.rs:146-200 (Cache::new): f32 arithmetic throughout, table rounded to the model dtype.
int atoma_rope_table(void *cos_out, void *sin_out, int64_t max_pos, int64_t head_dim, float rope_theta,
                     float rope_factor, float low_freq_factor, float high_freq_factor,
                     int64_t original_max_position_embeddings, int dtype) {
    atoma::clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { atoma::set_error("rope_table: dtype must be f16 or bf16"); return -1; }
    if (head_dim <= 0 || head_dim % 2) { atoma::set_error("rope_table: head_dim must be even"); return -1; }
    const int64_t half = head_dim / 2;
    std::vector<float> inv((size_t)half);
    for (int64_t j = 0; j < half; ++j) {
        float f = 1.0f / powf(rope_theta, (float)(2 * j) / (float)head_dim);
        if (rope_factor > 0.f) {  // Llama-3 wavelength-dependent scaling (llama.rs:163-186)
            const float orig = (float)original_max_position_embeddings;
            const float low_wl = orig / low_freq_factor, high_wl = orig / high_freq_factor;
            const float wavelen = 2.0f * 3.14159265358979323846f / f;
            if (wavelen < high_wl) {
            } else if (wavelen > low_wl) {
                f = f / rope_factor;
            } else {
                const float smooth = (orig / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor);
                f = (1.0f - smooth) * f / rope_factor + smooth * f;
            }
        }
        inv[(size_t)j] = f;
    }
    auto *c16 = static_cast<uint16_t *>(cos_out);
    auto *s16 = static_cast<uint16_t *>(sin_out);
    for (int64_t p = 0; p < max_pos; ++p)
        for (int64_t j = 0; j < half; ++j) {
            const float ang = (float)p * inv[(size_t)j];
            const float c = cosf(ang), s = sinf(ang);
            c16[p * half + j] = dtype == ATOMA_BF16 ? atoma::host_f32_to_bf16(c) : atoma::host_f32_to_f16(c);
            s16[p * half + j] = dtype == ATOMA_BF16 ? atoma::host_f32_to_bf16(s) : atoma::host_f32_to_f16(s);
        }
    return 0;
}

}  // extern "C"
