// On-device greedy selection over the logits rows of a step.
//
// Replaces the per-sequence device->host copies of /root/reference/backends/vllm/src/model_executor.rs:
// 206-249 (`logits.i(idx)` -> `LogitsProcessor::sample` with Sampling::ArgMax -> `to_vec1::<f32>()[next_token]`):
// two 512 KB transfers per sequence per step at a 128k vocabulary become 8 bytes per sequence.  Index work:
// bit-exact against numpy argmax (smallest index among the maxima; NaNs are never selected).
// HBM-bound: rows * vocab * elt bytes read once.
#include "common.h"
#include <type_traits>

namespace atoma {

struct Best {
    float v;
    int i;
};
// larger value wins; equal values: smaller index (so the result does not depend on the work split)
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ void take(Best &b, float v, int i) {
    if (v > b.v) { b.v = v; b.i = i; }   // strict: keeps the first occurrence inside a thread's ascending scan; NaN never wins
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
    Best best{-INFINITY, 0x7fffffff};
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
        Best o{__shfl_xor(best.v, off, 64), __shfl_xor(best.i, off, 64)};
        best = better(best, o);
    }
    __shared__ float sv[16];
    __shared__ int si[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { sv[wave] = best.v; si[wave] = best.i; }
    __syncthreads();
    if (wave == 0) {
        Best b = lane < nw ? Best{sv[lane], si[lane]} : Best{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            Best o{__shfl_xor(b.v, off, 64), __shfl_xor(b.i, off, 64)};
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
