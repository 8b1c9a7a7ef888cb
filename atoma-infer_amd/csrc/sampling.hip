// On-device greedy selection over the logits rows of a step.
//
// Replaces the per-sequence device->host copies of /root/reference/backends/vllm/src/model_executor.rs:
// 206-249 (`logits.i(idx)` -> `LogitsProcessor::sample` with Sampling::ArgMax -> `to_vec1::<f32>()[next_token]`):
// two 512 KB transfers per sequence per step at a 128k vocabulary become 8 bytes per sequence.  Index work:
// bit-exact against numpy argmax (smallest index among the maxima; NaNs are never selected).
// HBM-bound: rows * vocab * elt bytes read once.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace atoma {

// Every comparison below is an INTEGER comparison of an order-preserving key made from the float's bit pattern:
// this translation unit is built with -fno-honor-nans (Makefile), under which the compiler may fold `v != v` to false
// and give `v > b` any value for a NaN operand.  Larger float <=> larger key; -0.0 == +0.0; NaN (either sign) -> 0,
// below the key of -inf (0x007fffff), so a NaN is never selected and sorts last.
__device__ __forceinline__ uint32_t order_key(float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;   // NaN, from the bits
    if ((u & 0x7fffffffu) == 0u) u = 0u;              // -0.0 and +0.0 compare equal
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
constexpr uint32_t KEY_NEG_INF = 0x007fffffu;

struct Best {
    uint32_t v;   // order_key of the value
    int i;
};
// larger value wins; equal values: smaller index (so the result does not depend on the work split)
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ void take(Best &b, float v, int i) {
    const uint32_t k = order_key(v);
    if (k > b.v) { b.v = k; b.i = i; }   // strict: keeps the first occurrence inside a thread's ascending scan; NaN never wins
}

template <typename T> __device__ __forceinline__ float load1(const void *row, int64_t i);
template <> __device__ __forceinline__ float load1<float>(const void *row, int64_t i) { return static_cast<const float *>(row)[i]; }
template <> __device__ __forceinline__ float load1<bf16_t>(const void *row, int64_t i) {
    return __uint_as_float((uint32_t) static_cast<const uint16_t *>(row)[i] << 16);
}
template <> __device__ __forceinline__ float load1<f16_t>(const void *row, int64_t i) {
    return (float)__builtin_bit_cast(_Float16, static_cast<const uint16_t *>(row)[i]);
}

// one workgroup per row; 16-byte loads when the row allows it
template <typename T, bool VEC>
__global__ void __launch_bounds__(1024) argmax_rows_kernel(const void *__restrict__ logits, int64_t row_stride_bytes, int vocab,
                                                           int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    constexpr int EPV = std::is_same<T, float>::value ? 4 : 8;   // elements per 16-byte load
    const char *row = static_cast<const char *>(logits) + (int64_t)blockIdx.x * row_stride_bytes;
    Best best{KEY_NEG_INF, 0x7fffffff};
    if (VEC) {
        const int nvec = vocab / EPV;
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_v;
        const u32x4_v *rv = reinterpret_cast<const u32x4_v *>(row);
        for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
            const u32x4_v u = __builtin_nontemporal_load(rv + c);   // read once: keep the logits out of L2's way
            const uint32_t w[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (std::is_same<T, float>::value) {
                    take(best, __uint_as_float(w[e]), c * 4 + e);
                } else {
                    take(best, lo_to_f32<T>(w[e]), c * 8 + 2 * e);
                    take(best, hi_to_f32<T>(w[e]), c * 8 + 2 * e + 1);
                }
            }
        }
        for (int i = nvec * EPV + threadIdx.x; i < vocab; i += blockDim.x) take(best, load1<T>(row, i), i);
    } else {
        for (int i = threadIdx.x; i < vocab; i += blockDim.x) take(best, load1<T>(row, i), i);
    }
    // wave reduction, then across the waves through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o{(uint32_t)__shfl_xor((int)best.v, off, 64), __shfl_xor(best.i, off, 64)};
        best = better(best, o);
    }
    __shared__ uint32_t sv[16];
    __shared__ int si[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { sv[wave] = best.v; si[wave] = best.i; }
    __syncthreads();
    if (wave == 0) {
        Best b = lane < nw ? Best{sv[lane], si[lane]} : Best{KEY_NEG_INF, 0x7fffffff};
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            Best o{(uint32_t)__shfl_xor((int)b.v, off, 64), __shfl_xor(b.i, off, 64)};
            b = better(b, o);
        }
        if (lane == 0) {
            const int idx = b.i == 0x7fffffff ? 0 : b.i;   // nothing compared greater than -inf (all -inf / NaN): index 0, as numpy on -inf
            out_idx[blockIdx.x] = idx;
            if (out_val) out_val[blockIdx.x] = load1<T>(row, idx);
        }
    }
}

}  // namespace atoma

extern "C" int atoma_argmax_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, int32_t *out_idx,
                                 float *out_val, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16 && dtype != ATOMA_F32) { set_error("argmax_rows: dtype must be f16, bf16 or f32"); return -1; }
    if (rows < 0 || vocab <= 0 || vocab > 0x7ffffffe) { set_error("argmax_rows: invalid shape"); return -1; }
    if (row_stride < vocab) { set_error("argmax_rows: row_stride must be >= vocab"); return -1; }
    if (!out_idx) { set_error("argmax_rows: out_idx is required"); return -1; }
    if (rows == 0) return 0;
    const int elt = dtype == ATOMA_F32 ? 4 : 2;
    const int64_t stride_bytes = row_stride * elt;
    const bool vec = (reinterpret_cast<uintptr_t>(logits) & 15u) == 0 && stride_bytes % 16 == 0;
    const int threads = vocab >= 32768 ? 1024 : (vocab >= 4096 ? 256 : 64);
    const auto s = static_cast<hipStream_t>(stream);
#define ATOMA_ARGMAX(TT)                                                                                                   \
    do {                                                                                                                   \
        if (vec) hipLaunchKernelGGL((argmax_rows_kernel<TT, true>), dim3((unsigned)rows), dim3(threads), 0, s, logits,     \
                                    stride_bytes, (int)vocab, out_idx, out_val);                                           \
        else hipLaunchKernelGGL((argmax_rows_kernel<TT, false>), dim3((unsigned)rows), dim3(threads), 0, s, logits,        \
                                stride_bytes, (int)vocab, out_idx, out_val);                                               \
    } while (0)
    if (dtype == ATOMA_F32) ATOMA_ARGMAX(float);
    else if (dtype == ATOMA_BF16) ATOMA_ARGMAX(bf16_t);
    else ATOMA_ARGMAX(f16_t);
#undef ATOMA_ARGMAX
    return ATOMA_CHECK_LAUNCH("argmax_rows") ? 0 : -1;
}

// ------------------------------------------------------------------------------------------
// Top-k per row: the k largest logits with their indices, ordered by (value descending, index ascending), so that
// top-k / top-p sampling on the host needs k values per sequence instead of the vocabulary (model_executor.rs:206-249,
// candle_transformers LogitsProcessor top-k / top-p branches).  One workgroup per row:
//   1. histogram of the top 12 bits of an order-preserving key -> the bucket T that holds the k-th largest element;
//   2. every element of a bucket >= T goes to an LDS candidate list (k + one bucket's population: a few hundred for
//      real logits), which is sorted (bitonic, (key desc, index asc)) and cut at k;
//   3. rows whose candidates do not fit (thousands of values in one 1/4096 slice of the number line: constant rows,
//      all -inf) fall back to k rounds of "largest element after the previous one" -- exact, slow, rare.
// Index / order work: bit-exact against numpy lexsort.  Reads the row twice (HBM, then MALL / L2).
// ------------------------------------------------------------------------------------------
namespace atoma {

constexpr int TOPK_MAX = 1024, TOPK_CAP = 4096, TOPK_THREADS = 1024;

// sort key of a candidate: value key in the high word, inverted index in the low word -> plain descending order
__device__ __forceinline__ unsigned long long cand(uint32_t key, int idx) { return ((unsigned long long)key << 32) | (uint32_t)(0x7fffffff - idx); }

// sort the n <= TOPK_CAP candidates (bitonic, descending) and emit the first k; whole workgroup, n uniform
template <typename T>
__device__ __forceinline__ void topk_sort_emit(unsigned long long *cands, unsigned int n, int k, const char *row, float *ov, int32_t *oi) {
    const int tid = threadIdx.x;
    int np2 = 1;
    while (np2 < (int)n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += TOPK_THREADS) cands[i] = 0ull;   // padding sorts last
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < np2 / 2; i += TOPK_THREADS) {
                const int lo = (i / stride) * 2 * stride + (i % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cands[lo], b = cands[hi];
                if ((a < b) == desc) { cands[lo] = b; cands[hi] = a; }
            }
            __syncthreads();
        }
    for (int j = tid; j < k; j += TOPK_THREADS) {
        const int idx = 0x7fffffff - (int)(uint32_t)cands[j];
        oi[j] = idx;
        ov[j] = load1<T>(row, idx);
    }
}

struct TopkShared {
    unsigned int hist[4096];
    unsigned long long cands[TOPK_CAP];
    unsigned long long red[TOPK_THREADS / 64];
    unsigned int cnt[16];
    unsigned int n_cand, t_bucket;
};

// the histogram route over a row in memory (two passes over the row; any vocabulary, any k <= TOPK_MAX)
template <typename T>
__device__ void topk_row_hist(TopkShared &sh, const char *row, int vocab, int k, float *ov, int32_t *oi) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += TOPK_THREADS) sh.hist[i] = 0;
    if (tid == 0) sh.n_cand = 0;
    __syncthreads();
    for (int i = tid; i < vocab; i += TOPK_THREADS) atomicAdd(&sh.hist[order_key(load1<T>(row, i)) >> 20], 1u);
    __syncthreads();
    if (tid < 64) {   // one wavefront walks the histogram from the top: 64 buckets per step
        unsigned int above = 0;
        int found = -1;
        for (int base = 4096 - 64; base >= 0 && found < 0; base -= 64) {
            const unsigned int c = sh.hist[base + tid];
            // inclusive suffix sum over the 64 lanes (lane 63 = highest bucket)
            unsigned int suf = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int o = __shfl_down(suf, off, 64);
                if (tid + off < 64) suf += o;
            }
            const bool hit = above + suf >= (unsigned)k;                    // the k-th largest is in this bucket or a higher one
            const unsigned long long m = __ballot(hit);
            if (m) found = base + 63 - __builtin_clzll(m);                  // highest bucket that reaches k
            above += __shfl(suf, 0, 64);
        }
        if (tid == 0) sh.t_bucket = found < 0 ? 0u : (unsigned)found;
    }
    __syncthreads();
    const unsigned int T0 = sh.t_bucket;
    for (int i = tid; i < vocab; i += TOPK_THREADS) {
        const uint32_t key = order_key(load1<T>(row, i));
        if ((key >> 20) >= T0) {
            const unsigned int pos = atomicAdd(&sh.n_cand, 1u);
            if (pos < TOPK_CAP) sh.cands[pos] = cand(key, i);
        }
    }
    __syncthreads();
    const unsigned int n = sh.n_cand;
    if (n <= TOPK_CAP) { topk_sort_emit<T>(sh.cands, n, k, row, ov, oi); return; }
    // fallback: k rounds of "largest candidate strictly below the previous pick" (lexicographic on (key, -index))
    unsigned long long prev = ~0ull;
    for (int j = 0; j < k; ++j) {
        unsigned long long best = 0ull;
        for (int i = tid; i < vocab; i += TOPK_THREADS) {
            const unsigned long long c = cand(order_key(load1<T>(row, i)), i);
            if (c < prev && c > best) best = c;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        if ((tid & 63) == 0) sh.red[tid >> 6] = best;
        __syncthreads();
        if (tid < 64) {
            unsigned long long b = tid < TOPK_THREADS / 64 ? sh.red[tid] : 0ull;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(b, off, 64);
                b = o > b ? o : b;
            }
            if (tid == 0) sh.red[0] = b;
        }
        __syncthreads();
        prev = sh.red[0];
        __syncthreads();
        if (tid == 0) {
            const int idx = 0x7fffffff - (int)(uint32_t)prev;
            oi[j] = idx;
            ov[j] = load1<T>(row, idx);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_kernel(const void *__restrict__ logits, int64_t row_stride_bytes, int vocab, int k,
                                                                 float *__restrict__ out_val, int32_t *__restrict__ out_idx) {
    __shared__ TopkShared sh;
    const char *row = static_cast<const char *>(logits) + (int64_t)blockIdx.x * row_stride_bytes;
    topk_row_hist<T>(sh, row, vocab, k, out_val + (int64_t)blockIdx.x * k, out_idx + (int64_t)blockIdx.x * k);
}

// Rows of up to 131072 logits with k <= 256 (every real sampler setting): ONE pass over the row.  Each of the 1024 threads keeps
// a 16-bit prefix of the order keys of its 128 elements in 64 registers (16-byte non-temporal loads, element (1024 j + tid) * EPV
// + e).  Threshold: t = the k-th largest of the 1024 per-thread maxima (16 rounds of a ballot count, no sort): at least k
// elements have a prefix >= t, so every member of the exact top k has one too -- and only a few more do (k = 50 of 128256 random
// logits: 50-60 candidates).  Those are re-read with their full keys (a few hundred L2 hits), sorted and cut at k as before.
// No histogram, no LDS atomics on hot buckets; a row with too many candidates (constant rows, NaN-filled rows) takes the
// histogram route above.
constexpr int TOPK_REG_MAX_VOCAB = 131072, TOPK_REG_MAX_K = 256;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_v;
// 16-bit prefixes of the order, two at a time.  h = the top 16 bits of an f32 / the bf16 / the f16 pattern (sign, exponent,
// leading mantissa bits); exact prefix: NaN -> 0, -0.0 == +0.0, negative -> ~h, positive -> h | 0x8000 (monotone in the exact
// key).  The register kernel only needs an UPPER bound of it that is cheap on packed lanes: positive h + 0x8000, negative
// -h = ~h + 1 (one too many except for -0.0, where it is exact), NaNs left as they fall (>= 0 = their exact prefix).  The
// bound can only ADD candidates or raise the threshold; the latter is detected afterwards (fewer than k candidates whose exact
// prefix reaches the threshold) and sent down the histogram route.
typedef __attribute__((ext_vector_type(2))) short i16x2_v;
__device__ __forceinline__ uint32_t prefix_approx2(uint32_t h2) {
    const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2_v, h2) >> (short)15);   // 0xffff in the negative halves
    const uint32_t t = h2 ^ (m | 0x80008000u);                                                             // ~h or h | 0x8000
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_v, t) + __builtin_bit_cast(u16x2_v, m & 0x00010001u));
}
template <typename T> __device__ __forceinline__ uint32_t prefix_exact(const void *row, int idx) {
    if constexpr (std::is_same<T, f16_t>::value) {
        const uint32_t h = static_cast<const uint16_t *>(row)[idx];
        if ((h & 0x7fffu) > 0x7c00u) return 0u;
        if ((h & 0x7fffu) == 0u) return 0x8000u;
        return (h & 0x8000u) ? (~h & 0xffffu) : (h | 0x8000u);
    } else {
        return order_key(load1<T>(row, idx)) >> 16;
    }
}
template <typename T>
__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_reg_kernel(const void *__restrict__ logits, int64_t row_stride_bytes, int vocab, int k,
                                                                     float *__restrict__ out_val, int32_t *__restrict__ out_idx) {
    constexpr bool F32IN = std::is_same<T, float>::value;
    constexpr int EPV = F32IN ? 4 : 8, NV = TOPK_REG_MAX_VOCAB / TOPK_THREADS / EPV, PPV = EPV / 2, NP = NV * PPV;   // 64 packed registers
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_v;
    __shared__ TopkShared sh;
    const char *row = static_cast<const char *>(logits) + (int64_t)blockIdx.x * row_stride_bytes;
    float *ov = out_val + (int64_t)blockIdx.x * k;
    int32_t *oi = out_idx + (int64_t)blockIdx.x * k;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nvec = vocab / EPV;                                      // the launcher guarantees vocab % EPV == 0
    if (tid < 16) sh.cnt[tid] = 0;
    if (tid == 0) { sh.n_cand = 0; sh.t_bucket = 0; }
    uint32_t pk[NP];
    const u32x4_v *rv = reinterpret_cast<const u32x4_v *>(row);
    constexpr uint32_t PAD = 0xffffffffu;                              // beyond the row: a negative NaN, whose prefix bound is 1 -- below every number
    constexpr int BATCH = F32IN ? 4 : 8;                               // 16-byte loads in flight per thread (f32: 128 registers is the budget at 1024 threads)
#pragma unroll
    for (int j0 = 0; j0 < NV; j0 += BATCH) {
        u32x4_v raw[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int v = (j0 + j) * TOPK_THREADS + tid;
            raw[j] = v < nvec ? __builtin_nontemporal_load(rv + v) : u32x4_v{PAD, PAD, PAD, PAD};   // beyond the row
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if constexpr (F32IN) {
                pk[(j0 + j) * 2] = prefix_approx2(__builtin_amdgcn_perm(raw[j][1], raw[j][0], 0x07060302u));       // the top halves of two floats
                pk[(j0 + j) * 2 + 1] = prefix_approx2(__builtin_amdgcn_perm(raw[j][3], raw[j][2], 0x07060302u));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[(j0 + j) * 4 + e] = prefix_approx2(raw[j][e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                             // keep the batches apart: hoisting every load to the top spills
    }
    // per-thread maximum of the prefixes
    u16x2_v m2 = __builtin_bit_cast(u16x2_v, pk[0]);
#pragma unroll
    for (int i = 1; i < NP; ++i) m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2_v, pk[i]));
    const uint32_t mx = m2[0] > m2[1] ? m2[0] : m2[1];
    __syncthreads();                                                   // cnt / n_cand are zero
    // t = the k-th largest of the 1024 maxima: the largest p with count(mx >= p) >= k
    uint32_t t = 0;
#pragma unroll 1
    for (int bit = 15; bit >= 0; --bit) {
        const uint32_t p = t | (1u << bit);
        const unsigned int c = (unsigned)__builtin_popcountll(__ballot(mx >= p));
        if (lane == 0 && c) atomicAdd(&sh.cnt[bit], c);
        __syncthreads();
        if (sh.cnt[bit] >= (unsigned)k) t = p;
    }
    // candidates: every element whose prefix is >= t (padding slots are beyond nvec and skipped)
    const u16x2_v tm = u16x2_v{(unsigned short)(t ? t - 1 : 0), (unsigned short)(t ? t - 1 : 0)};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const u16x2_v kv = __builtin_bit_cast(u16x2_v, pk[i]);
        const u16x2_v mm = __builtin_elementwise_max(kv, tm);
        if (t == 0 || __builtin_bit_cast(uint32_t, mm) != __builtin_bit_cast(uint32_t, tm)) {   // some half is >= t
            const int v = (i / PPV) * TOPK_THREADS + tid;
            if (v < nvec) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (kv[h] >= t) {
                        const int idx = v * EPV + (i % PPV) * 2 + h;
                        const unsigned int pos = atomicAdd(&sh.n_cand, 1u);
                        if (pos < TOPK_CAP) sh.cands[pos] = cand(order_key(load1<T>(row, idx)), idx);
                        if (prefix_exact<T>(row, idx) >= t) atomicAdd(&sh.t_bucket, 1u);   // candidates the exact prefix would have admitted too
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned int n = sh.n_cand, n_exact = sh.t_bucket;
    if (n <= TOPK_CAP && n_exact >= (unsigned)k) { topk_sort_emit<T>(sh.cands, n, k, row, ov, oi); return; }
    __syncthreads();
    topk_row_hist<T>(sh, row, vocab, k, ov, oi);
}

}  // namespace atoma

static bool topk_reg_enabled() {   // ATOMA_TOPK_REG=0: always the two-pass histogram kernel (A/B runs)
    static const bool on = [] { const char *e = getenv("ATOMA_TOPK_REG"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int atoma_topk_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, int64_t k, float *out_val,
                               int32_t *out_idx, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16 && dtype != ATOMA_F32) { set_error("topk_rows: dtype must be f16, bf16 or f32"); return -1; }
    if (rows < 0 || vocab <= 0 || vocab > 0x7ffffffe) { set_error("topk_rows: invalid shape"); return -1; }
    if (k <= 0 || k > TOPK_MAX || k > vocab) { set_error("topk_rows: k must be in [1, min(vocab, 1024)]"); return -1; }
    if (row_stride < vocab) { set_error("topk_rows: row_stride must be >= vocab"); return -1; }
    if (!out_val || !out_idx) { set_error("topk_rows: out_val and out_idx are required"); return -1; }
    if (rows == 0) return 0;
    const int64_t stride_bytes = row_stride * (dtype == ATOMA_F32 ? 4 : 2);
    const auto s = static_cast<hipStream_t>(stream);
    const int epv = dtype == ATOMA_F32 ? 4 : 8;
    const bool reg = vocab <= TOPK_REG_MAX_VOCAB && k <= TOPK_REG_MAX_K && vocab % epv == 0 && vocab >= 4 * TOPK_THREADS &&
                     (reinterpret_cast<uintptr_t>(logits) & 15u) == 0 && stride_bytes % 16 == 0 && topk_reg_enabled();
#define ATOMA_TOPK(TT)                                                                                                                       \
    do {                                                                                                                                     \
        if (reg) hipLaunchKernelGGL((topk_rows_reg_kernel<TT>), dim3((unsigned)rows), dim3(TOPK_THREADS), 0, s, logits, stride_bytes, (int)vocab, (int)k, out_val, out_idx); \
        else hipLaunchKernelGGL((topk_rows_kernel<TT>), dim3((unsigned)rows), dim3(TOPK_THREADS), 0, s, logits, stride_bytes, (int)vocab, (int)k, out_val, out_idx);       \
    } while (0)
    if (dtype == ATOMA_F32) ATOMA_TOPK(float);
    else if (dtype == ATOMA_BF16) ATOMA_TOPK(bf16_t);
    else ATOMA_TOPK(f16_t);
#undef ATOMA_TOPK
    return ATOMA_CHECK_LAUNCH("topk_rows") ? 0 : -1;
}

// ------------------------------------------------------------------------------------------
// Stochastic token selection on the device: the non-ArgMax branches of candle_transformers' LogitsProcessor::sample, which
// /root/reference/backends/vllm/src/llm_service.rs:348-372 configures (All / TopK / TopP / TopKThenTopP with a temperature)
// and model_executor.rs:230-249 calls once per sequence on a logits row copied to the host.  Per row, in f32:
//   w_i = exp((x_i - max x) / temperature)            (softmax numerators; the denominator cancels in a weighted draw)
//   top_k > 0: only the k largest logits keep their weight; top_p < 1: in descending order, tokens keep their weight until the
//   cumulative probability reaches top_p (the token that crosses it is kept, as in Candle's sample_topp);
//   draw = u * sum(kept w); the token is the first kept one whose running sum exceeds draw.
// The order of the running sum is vocabulary order for plain multinomial sampling and (logit descending, index ascending) for
// top-k / top-p -- any fixed order gives the same distribution (Candle's own top-k order is unspecified: select_nth_unstable).
// The random stream stays with the caller: u[row] in [0, 1), one uniform per row (rand's WeightedIndex draws one uniform in
// [0, total)).  Two passes over the row for plain sampling (max, then 512-element chunk sums and a rescan of ONE chunk); the
// restricted variants run atoma_topk_rows first and then touch k values (+ one pass for the full-row denominator of top_p).
// ------------------------------------------------------------------------------------------
namespace atoma {

void *workspace(hipStream_t stream, size_t bytes);   // runtime.hip

constexpr int SAMPLE_THREADS = 1024, SAMPLE_CHUNK = 512, SAMPLE_MAX_CHUNKS = 4096;   // vocab <= 2 M

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
// finite-or-minus-infinity view of a logit: NaN never takes part (weight 0), from the bits (-fno-honor-nans)
__device__ __forceinline__ float sample_clean(float v) {
    return (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u ? -INFINITY : v;
}

template <typename T>
__global__ void __launch_bounds__(SAMPLE_THREADS) sample_full_kernel(const void *__restrict__ logits, int64_t row_stride_bytes, int vocab, float inv_temp,
                                                                     const float *__restrict__ u, int32_t *__restrict__ out_idx, float *__restrict__ out_logit) {
    __shared__ float csum[SAMPLE_MAX_CHUNKS];
    __shared__ float red[16];
    const char *row = static_cast<const char *>(logits) + (int64_t)blockIdx.x * row_stride_bytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pass 1: row max
    float mx = -INFINITY;
    for (int i = tid; i < vocab; i += SAMPLE_THREADS) mx = fmaxf(mx, sample_clean(load1<T>(row, i)));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    const float m = mx == -INFINITY ? 0.f : mx;
    // pass 2: weight sum of every 512-element chunk (lane: 8 consecutive elements, then a wave reduction)
    const int n_chunks = (vocab + SAMPLE_CHUNK - 1) / SAMPLE_CHUNK;
    auto weight = [&](int i) { return i < vocab ? __expf((sample_clean(load1<T>(row, i)) - m) * inv_temp) : 0.f; };
    for (int c = wave; c < n_chunks; c += 16) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += weight(c * SAMPLE_CHUNK + lane * 8 + e);
        s = wave_sum(s);
        if (lane == 0) csum[c] = s;
    }
    __syncthreads();
    // wave 0: total, draw, the chunk in which the running sum crosses the draw
    if (wave == 0) {
        float part = 0.f;
        for (int c = lane; c < n_chunks; c += 64) part += csum[c];
        const float total = wave_sum(part);
        const float draw = u[blockIdx.x] * total;
        // running sum over chunks in order, 64 at a time (inclusive scan over the lanes)
        float base = 0.f;
        int found = -1;
        float found_base = 0.f;
        for (int c0 = 0; c0 < n_chunks && found < 0; c0 += 64) {
            const float v = c0 + lane < n_chunks ? csum[c0 + lane] : 0.f;
            float inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float y = __shfl_up(inc, off, 64);
                if (lane >= off) inc += y;
            }
            const bool hit = c0 + lane < n_chunks && base + inc > draw;
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const int l0 = __builtin_ctzll(mask);
                found = c0 + l0;
                found_base = base + __shfl(inc - v, l0, 64);
            }
            base += __shfl(inc, 63, 64);
        }
        if (found < 0) { found = n_chunks - 1; found_base = base - csum[n_chunks - 1]; }   // draw == total after rounding: the last chunk
        // rescan that chunk: lane's 8 elements, inclusive scan over lanes, first element whose running sum exceeds the draw
        float wv[8], s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { wv[e] = weight(found * SAMPLE_CHUNK + lane * 8 + e); s += wv[e]; }
        float inc = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        float run = found_base + inc - s;
        int idx = -1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            run += wv[e];
            if (idx < 0 && wv[e] > 0.f && run > draw) idx = found * SAMPLE_CHUNK + lane * 8 + e;
        }
        const unsigned long long mask = __ballot(idx >= 0);
        int chosen;
        if (mask) chosen = __shfl(idx, __builtin_ctzll(mask), 64);
        else {   // rounding left the draw beyond the last weight: the last token of the chunk that has any weight
            int last = -1;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (wv[e] > 0.f) last = found * SAMPLE_CHUNK + lane * 8 + e;
            const unsigned long long m2 = __ballot(last >= 0);
            chosen = m2 ? __shfl(last, 63 - __builtin_clzll(m2), 64) : 0;
        }
        if (lane == 0) {
            out_idx[blockIdx.x] = chosen;
            if (out_logit) out_logit[blockIdx.x] = load1<T>(row, chosen);
        }
    }
}

// ---- full-row nucleus (top-p whose nucleus is larger than the sorted top-1024 list) ----
// Candle's sample_topp keeps EVERY token, most probable first, until the cumulative probability reaches top_p.  When the
// 1024 most probable tokens do not get there (high temperature / flat rows over a 128k vocabulary) the nucleus is found over
// the whole row without sorting it: a crossing position in the order (logit descending, index ascending) is
//   the largest key t whose tail weight W(t) = sum of the weights of all tokens with key >= t satisfies the predicate
//   (bisection over the 32-bit order key, one pass over the row per step, fixed summation order => every thread sees the same
//   sums), then the tokens that share that key, in index order, one equal weight at a time.
// Rare path: ~35 passes over an L2-resident row by 256 threads.
struct Crossing { int token; float cum; };          // cum = running weight including `token`

template <typename T>
__device__ float tail_weight(const char *row, int vocab, float m, float inv_temp, uint32_t t, float *red) {
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < vocab; i += 256) {
        const float x = sample_clean(load1<T>(row, i));
        if (order_key(x) >= t) s += __expf((x - m) * inv_temp);
    }
    s = wave_sum(s);
    __syncthreads();                                 // red is reused by every call
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// STRICT: first position whose running sum EXCEEDS target (the draw); else: first position whose running sum REACHES it (the
// nucleus cut).  If no position qualifies (target beyond the row's total after rounding) the last token with any weight.
template <typename T, bool STRICT>
__device__ Crossing sorted_crossing(const char *row, int vocab, float m, float inv_temp, float target, float *red, int *ired) {
    const int tid = threadIdx.x;
    auto pred = [&](float w) { return STRICT ? w > target : w >= target; };
    // keys of weighted tokens are > KEY_NEG_INF; lo: predicate holds (or nothing qualifies), hi: it does not
    unsigned long long lo = KEY_NEG_INF + 1ull, hi = 1ull << 32;
    float w_hi = 0.f;
    const bool any = pred(tail_weight<T>(row, vocab, m, inv_temp, (uint32_t)lo, red));
    if (!any) {                                      // the last weighted token in sorted order: smallest key, largest index
        Best b{0xffffffffu, -1};
        for (int i = tid; i < vocab; i += 256) {
            const uint32_t k = order_key(sample_clean(load1<T>(row, i)));
            if (k > KEY_NEG_INF && (k < b.v || (k == b.v && i > b.i))) { b.v = k; b.i = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            Best o{(uint32_t)__shfl_xor((int)b.v, off, 64), __shfl_xor(b.i, off, 64)};
            if (o.v < b.v || (o.v == b.v && o.i > b.i)) b = o;
        }
        __syncthreads();
        if ((tid & 63) == 0) { ired[tid >> 6] = (int)b.v; ired[4 + (tid >> 6)] = b.i; }
        __syncthreads();
        Best r{(uint32_t)ired[0], ired[4]};
        for (int w = 1; w < 4; ++w) {
            Best o{(uint32_t)ired[w], ired[4 + w]};
            if (o.v < r.v || (o.v == r.v && o.i > r.i)) r = o;
        }
        return Crossing{r.i < 0 ? 0 : r.i, tail_weight<T>(row, vocab, m, inv_temp, KEY_NEG_INF + 1u, red)};
    }
    while (hi - lo > 1) {
        const unsigned long long mid = (lo + hi) >> 1;
        const float w = tail_weight<T>(row, vocab, m, inv_temp, (uint32_t)mid, red);
        if (pred(w)) lo = mid; else { hi = mid; w_hi = w; }
    }
    // tokens with key == lo share one weight; position j (index order) has running sum w_hi + (j + 1) * wt
    const uint32_t key = (uint32_t)lo;
    const int per = (vocab + 255) / 256, i0 = tid * per, i1 = min(vocab, i0 + per);
    int cnt = 0;
    float wt = 0.f;
    for (int i = i0; i < i1; ++i) {
        const float x = sample_clean(load1<T>(row, i));
        if (order_key(x) == key) { ++cnt; wt = __expf((x - m) * inv_temp); }
    }
    // exclusive prefix of the counts over the 256 threads (contiguous ranges => index order) and the shared weight
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(inc, off, 64);
        if ((tid & 63) >= off) inc += y;
    }
    const float wmax = wave_max(wt);
    __syncthreads();
    if ((tid & 63) == 63) ired[tid >> 6] = inc;
    if ((tid & 63) == 0) red[tid >> 6] = wmax;
    __syncthreads();
    int before = inc - cnt;
    for (int w = 0; w < (tid >> 6); ++w) before += ired[w];
    const int n_ties = ired[0] + ired[1] + ired[2] + ired[3];
    wt = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int jl = 0, jh = n_ties - 1;                     // smallest j with pred(w_hi + (j + 1) wt); j = n_ties - 1 qualifies (= W(lo))
    while (jl < jh) {
        const int jm = (jl + jh) >> 1;
        if (pred(w_hi + (float)(jm + 1) * wt)) jh = jm; else jl = jm + 1;
    }
    __syncthreads();
    if (before <= jl && jl < before + cnt) {         // exactly one thread owns the jl-th tie
        int seen = before;
        for (int i = i0; i < i1; ++i)
            if (order_key(sample_clean(load1<T>(row, i))) == key && seen++ == jl) ired[8] = i;
    }
    __syncthreads();
    return Crossing{ired[8], w_hi + (float)(jl + 1) * wt};
}

// top-k / top-p: `vals`, `idxs` [rows][k] from atoma_topk_rows (value descending, index ascending); one wave per row
template <typename T>
__global__ void __launch_bounds__(256) sample_topk_kernel(const void *__restrict__ logits, int64_t row_stride_bytes, int vocab, const float *__restrict__ vals,
                                                          const int32_t *__restrict__ idxs, int k, float inv_temp, float top_p, const float *__restrict__ u,
                                                          int32_t *__restrict__ out_idx, float *__restrict__ out_logit, int full_row_nucleus) {
    __shared__ float red[4];
    __shared__ int ired[12];
    const char *row = static_cast<const char *>(logits) + (int64_t)blockIdx.x * row_stride_bytes;
    const float *v = vals + (int64_t)blockIdx.x * k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float m = sample_clean(v[0]) == -INFINITY ? 0.f : v[0];
    float denom = 0.f;      // full-row softmax denominator, needed only by top_p
    if (top_p < 1.f) {
        float s = 0.f;
        for (int i = tid; i < vocab; i += 256) s += __expf((sample_clean(load1<T>(row, i)) - m) * inv_temp);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        denom = red[0] + red[1] + red[2] + red[3];
    }
    // full_row_nucleus (top-p with no top-k): the sorted list is only the first 1024 tokens of the order; when their weight does
    // not reach top_p * denom the nucleus continues beyond it and all four wavefronts find it over the whole row
    if (wave != 0 && !full_row_nucleus) return;
    // kept prefix of the sorted list: all k, or up to (and including) the token whose cumulative probability reaches top_p
    int keep = k;
    float total = 0.f;
    bool cut = false;
    {
        float base = 0.f;
        for (int c0 = 0; c0 < k && !cut; c0 += 64) {
            const float w = c0 + lane < k ? __expf((sample_clean(v[c0 + lane]) - m) * inv_temp) : 0.f;
            float inc = w;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float y = __shfl_up(inc, off, 64);
                if (lane >= off) inc += y;
            }
            if (top_p < 1.f) {
                const bool reach = c0 + lane < k && (base + inc) >= top_p * denom;
                const unsigned long long mask = __ballot(reach);
                if (mask) {
                    const int l0 = __builtin_ctzll(mask);
                    keep = c0 + l0 + 1;
                    total = base + __shfl(inc, l0, 64);
                    cut = true;
                    break;
                }
            }
            base += __shfl(inc, 63, 64);
            total = base;
        }
    }
    if (full_row_nucleus) {
        // every wavefront ran the scan above on the same values: `cut` is uniform over the workgroup
        if (!cut && k < vocab) {
            const Crossing nucleus = sorted_crossing<T, false>(row, vocab, m, inv_temp, top_p * denom, red, ired);
            const Crossing pick = sorted_crossing<T, true>(row, vocab, m, inv_temp, u[blockIdx.x] * nucleus.cum, red, ired);
            if (tid == 0) {
                out_idx[blockIdx.x] = pick.token;
                if (out_logit) out_logit[blockIdx.x] = load1<T>(row, pick.token);
            }
            return;
        }
        if (wave != 0) return;
    }
    const float draw = u[blockIdx.x] * total;
    float base = 0.f;
    int chosen = -1;
    for (int c0 = 0; c0 < keep && chosen < 0; c0 += 64) {
        const float w = c0 + lane < keep ? __expf((sample_clean(v[c0 + lane]) - m) * inv_temp) : 0.f;
        float inc = w;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        const unsigned long long mask = __ballot(c0 + lane < keep && w > 0.f && base + inc > draw);
        if (mask) chosen = c0 + __builtin_ctzll(mask);
        base += __shfl(inc, 63, 64);
    }
    if (chosen < 0) chosen = keep - 1;
    if (lane == 0) {
        const int tok = idxs[(int64_t)blockIdx.x * k + chosen];
        out_idx[blockIdx.x] = tok;
        if (out_logit) out_logit[blockIdx.x] = load1<T>(row, tok);
    }
}

}  // namespace atoma

extern "C" int atoma_sample_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, float temperature, int64_t top_k,
                                 float top_p, const float *u, int32_t *out_idx, float *out_logit, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16 && dtype != ATOMA_F32) { set_error("sample_rows: dtype must be f16, bf16 or f32"); return -1; }
    if (rows < 0 || vocab <= 0 || vocab > (int64_t)SAMPLE_CHUNK * SAMPLE_MAX_CHUNKS) { set_error("sample_rows: invalid shape"); return -1; }
    if (row_stride < vocab) { set_error("sample_rows: row_stride must be >= vocab"); return -1; }
    if (!(temperature > 0.f)) { set_error("sample_rows: temperature must be positive (greedy selection is atoma_argmax_rows)"); return -1; }
    if (top_k < 0 || top_k > TOPK_MAX) { set_error("sample_rows: top_k must be in [0, 1024] (0 = no top-k)"); return -1; }
    if (!(top_p > 0.f)) { set_error("sample_rows: top_p must be in (0, 1] (1 = no top-p)"); return -1; }
    if (!u || !out_idx) { set_error("sample_rows: u and out_idx are required"); return -1; }
    if (rows == 0) return 0;
    const int64_t stride_bytes = row_stride * (dtype == ATOMA_F32 ? 4 : 2);
    const auto s = static_cast<hipStream_t>(stream);
    const float inv_temp = 1.f / temperature;
    const bool restricted = (top_k > 0 && top_k < vocab) || top_p < 1.f;
    if (!restricted) {
#define ATOMA_SF(TT) hipLaunchKernelGGL((sample_full_kernel<TT>), dim3((unsigned)rows), dim3(SAMPLE_THREADS), 0, s, logits, stride_bytes, (int)vocab, inv_temp, u, out_idx, out_logit)
        if (dtype == ATOMA_F32) ATOMA_SF(float); else if (dtype == ATOMA_BF16) ATOMA_SF(bf16_t); else ATOMA_SF(f16_t);
#undef ATOMA_SF
        return ATOMA_CHECK_LAUNCH("sample_rows") ? 0 : -1;
    }
    // top-p without top-k: the nucleus is searched among the 1024 most probable tokens first (it ends there for any peaked
    // distribution); a row whose 1024 most probable tokens do not reach top_p continues over the whole row (sorted_crossing)
    const bool have_k = top_k > 0 && top_k < vocab;
    const int64_t k = std::min<int64_t>(have_k ? top_k : TOPK_MAX, vocab);
    const int full_row = !have_k && top_p < 1.f;
    char *ws = static_cast<char *>(workspace(s, (size_t)rows * k * 8));
    if (!ws) return -1;
    float *vals = reinterpret_cast<float *>(ws);
    int32_t *idxs = reinterpret_cast<int32_t *>(ws + (size_t)rows * k * 4);
    if (atoma_topk_rows(logits, rows, vocab, row_stride, dtype, k, vals, idxs, stream) != 0) return -1;
#define ATOMA_ST(TT) hipLaunchKernelGGL((sample_topk_kernel<TT>), dim3((unsigned)rows), dim3(256), 0, s, logits, stride_bytes, (int)vocab, vals, idxs, (int)k, inv_temp, top_p, u, out_idx, out_logit, full_row)
    if (dtype == ATOMA_F32) ATOMA_ST(float); else if (dtype == ATOMA_BF16) ATOMA_ST(bf16_t); else ATOMA_ST(f16_t);
#undef ATOMA_ST
    return ATOMA_CHECK_LAUNCH("sample_rows (top-k / top-p)") ? 0 : -1;
}
