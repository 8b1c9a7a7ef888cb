/*
 * C restatement of the atoma-infer attention / KV-cache hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): built into oracle/_build/liboracle.so,
 * used by tests/ as a second, independent checker next to the numpy restatement and by
 * bench.py's "cpu_baseline" leg ("kind": "port").  Never linked into libatoma_hip.so.
 *
 * The reference has no CPU implementation of this path (every custom op's cpu_fwd bails:
 * /root/reference/csrc/src/lib.rs:350-360,1110-1120,1862-1872); its only CPU statement of
 * the result is the test helper fa_acausal (csrc/tests/flash_attn_tests.rs:19-29):
 *     att = softmax(f32(q) @ f32(k)^T * scale) ; out = dtype(att @ f32(v))
 * which is what oracle_attention() computes, with
 *   - sequence length / offset rules of csrc/kernels/block_info.h:11-39,
 *   - key j visible to query row r iff !causal or j <= r + Lk - Lq (csrc/kernels/mask.h:170-190),
 *   - paged rows at cache[block_table[j / page]][j % page]     (csrc/kernels/utils.h:296-314),
 *   - empty key range -> out = 0                               (csrc/kernels/flash_fwd_kernel.h:97-133).
 * Cache ops follow csrc/kernels/cache_manager.cu:15-37,139-170 and csrc/src/cache_manager.rs:18-128.
 *
 * PARITY: pinned through tests/test_oracle_golden.py (reference golden tables G1/G2 and
 * equivalence properties P1/P2/P3); _ref build of the reference: impossible here (Rust+CUDA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            u = s | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_f16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u);          /* overflow -> inf */
    if (a < 0x33000001u) return (uint16_t)s;                       /* underflow -> 0 (<= 2^-25) */
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                 /* bits dropped */
    uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
    uint32_t r = m >> shift;
    if (rest > half || (rest == half && (r & 1u))) ++r;
    if (e < -14) return (uint16_t)(s | r);                         /* subnormal (may carry into normal) */
    return (uint16_t)(s | (((uint32_t)(e + 15) << 10) + (r - 0x400u)));
}
static inline float ld(const uint16_t *p, int dtype) { return dtype ? bf16_to_f32(*p) : f16_to_f32(*p); }
static inline uint16_t st(float f, int dtype) { return dtype ? f32_to_bf16(f) : f32_to_f16(f); }

/*
 * One call covers the reference's three entry points (csrc/src/lib.rs): the arguments are
 * the subset of run_mha's (csrc/src/ffi.rs:4-64) that carries meaning, strides in ELEMENTS.
 *   cu_seqlens_q == NULL : fixed seqlen_q per batch, q at b * q_batch_stride
 *   cu_seqlens_k == NULL : fixed seqlen_k; else cumulative [b+1] or, when
 *   !is_seqlens_k_cumulative, per-sequence lengths [b] (kv-cache entry point)
 *   block_table != NULL  : k,v are paged caches, k_batch_stride is the page stride.
 */
void oracle_attention(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *o,
                      const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                      int is_seqlens_k_cumulative, int64_t q_batch_stride, int64_t k_batch_stride,
                      int64_t v_batch_stride, int64_t o_batch_stride, int64_t q_row_stride,
                      int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride,
                      int64_t o_head_stride, int b, int h, int h_k, int d, float softmax_scale,
                      const int32_t *block_table, int64_t block_table_batch_stride,
                      int page_block_size, int seqlen_q, int seqlen_k, int is_bf16, int is_causal,
                      int num_threads) {
    const int g = h / h_k;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int ib = 0; ib < b; ++ib) {
        for (int ih = 0; ih < h; ++ih) {
            const int sum_q = cu_seqlens_q ? cu_seqlens_q[ib] : -1;
            const int Lq = cu_seqlens_q ? cu_seqlens_q[ib + 1] - sum_q : seqlen_q;
            const int sum_k = (cu_seqlens_k && is_seqlens_k_cumulative) ? cu_seqlens_k[ib] : -1;
            const int Lk = !cu_seqlens_k ? seqlen_k
                           : (is_seqlens_k_cumulative ? cu_seqlens_k[ib + 1] - sum_k : cu_seqlens_k[ib]);
            const uint16_t *qb = q + (sum_q < 0 ? ib * q_batch_stride : sum_q * q_row_stride) + ih * q_head_stride;
            uint16_t *ob = o + (sum_q < 0 ? ib * o_batch_stride : sum_q * o_row_stride) + ih * o_head_stride;
            const int ihk = ih / g;
            const int64_t koff = block_table ? 0 : (sum_k < 0 ? ib * k_batch_stride : sum_k * k_row_stride);
            const int64_t voff = block_table ? 0 : (sum_k < 0 ? ib * v_batch_stride : sum_k * v_row_stride);
            float *sc = (float *)malloc(sizeof(float) * (size_t)(Lk > 0 ? Lk : 1));
            float *acc = (float *)malloc(sizeof(float) * (size_t)d);
            float *qf = (float *)malloc(sizeof(float) * (size_t)d);
            for (int r = 0; r < Lq; ++r) {
                int hi = is_causal ? r + Lk - Lq + 1 : Lk; /* visible keys: [0, hi) */
                if (hi > Lk) hi = Lk;
                uint16_t *orow = ob + (int64_t)r * o_row_stride;
                if (hi <= 0) {
                    for (int c = 0; c < d; ++c) orow[c] = 0;
                    continue;
                }
                for (int c = 0; c < d; ++c) qf[c] = ld(qb + (int64_t)r * q_row_stride + c, is_bf16);
                float m = -INFINITY;
                for (int j = 0; j < hi; ++j) {
                    const uint16_t *kr =
                        block_table ? k + (int64_t)block_table[ib * block_table_batch_stride + j / page_block_size] * k_batch_stride +
                                          (int64_t)(j % page_block_size) * k_row_stride
                                    : k + koff + (int64_t)j * k_row_stride;
                    kr += (int64_t)ihk * k_head_stride;
                    float s = 0.f;
                    for (int c = 0; c < d; ++c) s += qf[c] * ld(kr + c, is_bf16);
                    s *= softmax_scale;
                    sc[j] = s;
                    if (s > m) m = s;
                }
                float l = 0.f;
                for (int c = 0; c < d; ++c) acc[c] = 0.f;
                for (int j = 0; j < hi; ++j) {
                    const float p = expf(sc[j] - m);
                    l += p;
                    const uint16_t *vr =
                        block_table ? v + (int64_t)block_table[ib * block_table_batch_stride + j / page_block_size] * v_batch_stride +
                                          (int64_t)(j % page_block_size) * v_row_stride
                                    : v + voff + (int64_t)j * v_row_stride;
                    vr += (int64_t)ihk * v_head_stride;
                    for (int c = 0; c < d; ++c) acc[c] += p * ld(vr + c, is_bf16);
                }
                const float inv = 1.f / l;
                for (int c = 0; c < d; ++c) orow[c] = st(acc[c] * inv, is_bf16);
            }
            free(sc);
            free(acc);
            free(qf);
        }
    }
}

/*
 * The same result definition specialised for what bench.py times on the host cores ("cpu_baseline"): paged decode
 * (seqlen_q = 1, per-sequence lengths, bf16), written the way a CPU implementation of the path would be -- one task per
 * (sequence, KV head); a K / V row is converted to f32 ONCE and used by all h / h_k query heads of the group; the inner
 * loops over d are contiguous f32 loops the compiler vectorises (omp simd).  Two passes (scores, then exp / accumulate):
 * the arithmetic of fa_acausal, only the summation order inside a dot product differs from oracle_attention (checked by
 * tests/test_oracle_golden.py: identical after the one rounding except for boundary cases, never more than one unit).
 */
void oracle_decode_grouped(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *o, const int32_t *seqlens_k,
                           const int32_t *block_table, int64_t block_table_batch_stride, int page_block_size, int64_t page_stride,
                           int64_t row_stride, int64_t head_stride, int b, int h, int h_k, int d, float softmax_scale, int num_threads) {
    const int g = h / h_k;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int ib = 0; ib < b; ++ib) {
        for (int ihk = 0; ihk < h_k; ++ihk) {
            const int L = seqlens_k[ib];
            uint16_t *ob = o + ((int64_t)ib * h + (int64_t)ihk * g) * d;
            if (L <= 0) {
                for (int c = 0; c < g * d; ++c) ob[c] = 0;
                continue;
            }
            float *qf = (float *)malloc(sizeof(float) * (size_t)g * d);
            float *sc = (float *)malloc(sizeof(float) * (size_t)g * L);
            float *acc = (float *)calloc((size_t)g * d, sizeof(float));
            float *row = (float *)malloc(sizeof(float) * (size_t)d);
            float m[64], l[64];
            for (int c = 0; c < g * d; ++c) qf[c] = bf16_to_f32(q[((int64_t)ib * h + (int64_t)ihk * g) * d + c]);
            for (int a = 0; a < g; ++a) { m[a] = -INFINITY; l[a] = 0.f; }
            for (int j = 0; j < L; ++j) {
                const uint16_t *kr = k + (int64_t)block_table[ib * block_table_batch_stride + j / page_block_size] * page_stride +
                                     (int64_t)(j % page_block_size) * row_stride + (int64_t)ihk * head_stride;
#pragma omp simd
                for (int c = 0; c < d; ++c) row[c] = bf16_to_f32(kr[c]);
                for (int a = 0; a < g; ++a) {
                    float s = 0.f;
#pragma omp simd reduction(+ : s)
                    for (int c = 0; c < d; ++c) s += qf[a * d + c] * row[c];
                    s *= softmax_scale;
                    sc[(int64_t)a * L + j] = s;
                    if (s > m[a]) m[a] = s;
                }
            }
            for (int j = 0; j < L; ++j) {
                const uint16_t *vr = v + (int64_t)block_table[ib * block_table_batch_stride + j / page_block_size] * page_stride +
                                     (int64_t)(j % page_block_size) * row_stride + (int64_t)ihk * head_stride;
#pragma omp simd
                for (int c = 0; c < d; ++c) row[c] = bf16_to_f32(vr[c]);
                for (int a = 0; a < g; ++a) {
                    const float p = expf(sc[(int64_t)a * L + j] - m[a]);
                    l[a] += p;
                    float *ac = acc + a * d;
#pragma omp simd
                    for (int c = 0; c < d; ++c) ac[c] += p * row[c];
                }
            }
            for (int a = 0; a < g; ++a) {
                const float inv = 1.f / l[a];
                for (int c = 0; c < d; ++c) ob[a * d + c] = f32_to_bf16(acc[a * d + c] * inv);
            }
            free(qf); free(sc); free(acc); free(row);
        }
    }
}

/* csrc/kernels/cache_manager.cu:139-170 */
void oracle_reshape_and_cache_flash(const uint16_t *key, const uint16_t *value, uint16_t *key_cache,
                                    uint16_t *value_cache, const int64_t *slot_mapping,
                                    int64_t block_stride, int64_t num_tokens, int64_t num_heads,
                                    int64_t head_size, int64_t block_size, int64_t key_stride,
                                    int64_t value_stride) {
    const int64_t n = num_heads * head_size;
    for (int64_t t = 0; t < num_tokens; ++t) {
        const int64_t slot = slot_mapping[t];
        if (slot < 0) continue;
        const int64_t dst = (slot / block_size) * block_stride + (slot % block_size) * n;
        for (int64_t i = 0; i < n; ++i) {
            key_cache[dst + i] = key[t * key_stride + i];
            value_cache[dst + i] = value[t * value_stride + i];
        }
    }
}

/* csrc/kernels/cache_manager.cu:15-37 -- host pointers instead of device pointers */
void oracle_copy_blocks(uint16_t **key_caches, uint16_t **value_caches, const int64_t *block_mapping,
                        int64_t num_layers, int64_t num_pairs, int64_t numel_per_block) {
    for (int64_t l = 0; l < num_layers; ++l)
        for (int64_t p = 0; p < num_pairs; ++p) {
            const int64_t s = block_mapping[2 * p] * numel_per_block, t = block_mapping[2 * p + 1] * numel_per_block;
            memmove(key_caches[l] + t, key_caches[l] + s, (size_t)numel_per_block * 2);
            memmove(value_caches[l] + t, value_caches[l] + s, (size_t)numel_per_block * 2);
        }
}

/* csrc/src/cache_manager.rs:18-128: dst[d] = src[s], whole pages */
void oracle_swap_blocks(const uint8_t *src, uint8_t *dst, const int64_t *mapping, int64_t num_pairs,
                        int64_t block_size_in_bytes) {
    for (int64_t p = 0; p < num_pairs; ++p)
        memcpy(dst + mapping[2 * p + 1] * block_size_in_bytes, src + mapping[2 * p] * block_size_in_bytes,
               (size_t)block_size_in_bytes);
}

/* candle-kernels rmsnorm semantics (see oracle/norm_rope_oracle.py): f32, one rounding */
void oracle_rms_norm(const uint16_t *x, const uint16_t *w, uint16_t *y, int64_t rows, int64_t hidden,
                     int64_t x_row_stride, float eps, int is_bf16) {
#pragma omp parallel for
    for (int64_t r = 0; r < rows; ++r) {
        double ss = 0;
        for (int64_t c = 0; c < hidden; ++c) {
            const float f = ld(x + r * x_row_stride + c, is_bf16);
            ss += (double)f * f;
        }
        const float scale = (float)(1.0 / sqrt(ss / (double)hidden + (double)eps));
        for (int64_t c = 0; c < hidden; ++c)
            y[r * hidden + c] = st((scale * ld(x + r * x_row_stride + c, is_bf16)) * ld(w + c, is_bf16), is_bf16);
    }
}

/* candle rope semantics: per-op rounding in the tensor dtype; x [T, heads, d] with strides */
void oracle_rope(const uint16_t *x, uint16_t *y, const uint16_t *cos_t, const uint16_t *sin_t,
                 const int64_t *positions, int64_t T, int64_t heads, int64_t d, int64_t x_tok_stride,
                 int64_t x_head_stride, int64_t y_tok_stride, int64_t y_head_stride, int is_bf16) {
    const int64_t hd = d / 2;
#pragma omp parallel for
    for (int64_t t = 0; t < T; ++t)
        for (int64_t hh = 0; hh < heads; ++hh) {
            const uint16_t *xr = x + t * x_tok_stride + hh * x_head_stride;
            uint16_t *yr = y + t * y_tok_stride + hh * y_head_stride;
            for (int64_t i = 0; i < hd; ++i) {
                const float c = ld(cos_t + positions[t] * hd + i, is_bf16), s = ld(sin_t + positions[t] * hd + i, is_bf16);
                const float x1 = ld(xr + i, is_bf16), x2 = ld(xr + hd + i, is_bf16);
                const float a = ld(&(uint16_t){st(x1 * c, is_bf16)}, is_bf16), b = ld(&(uint16_t){st(x2 * s, is_bf16)}, is_bf16);
                const float e = ld(&(uint16_t){st(x1 * s, is_bf16)}, is_bf16), f = ld(&(uint16_t){st(x2 * c, is_bf16)}, is_bf16);
                yr[i] = st(a - b, is_bf16);
                yr[hd + i] = st(e + f, is_bf16);
            }
        }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
