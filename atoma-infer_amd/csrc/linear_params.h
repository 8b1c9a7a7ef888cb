// Shared by the projection kernels (linear_decode.hip, linear_tile.hip): operand types, the 16x16x32 MFMA wrapper, kernel parameters.
#pragma once
#include "common.h"

namespace atoma {

void *workspace(hipStream_t stream, size_t bytes);   // runtime.hip: grow-only fp32 scratch per (device, stream)

typedef unsigned int lu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) float lf32x4;

template <typename T> __device__ __forceinline__ lf32x4 lin_mfma(const lu32x4 &a, const lu32x4 &b, lf32x4 c);
template <> __device__ __forceinline__ lf32x4 lin_mfma<bf16_t>(const lu32x4 &a, const lu32x4 &b, lf32x4 c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ lf32x4 lin_mfma<f16_t>(const lu32x4 &a, const lu32x4 &b, lf32x4 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 v8;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}

struct LinearParams {
    const uint16_t *x, *w;
    uint16_t *y;
    float *partial;              // [splits][batch][n] fp32, or null when splits == 1
    int64_t x_row_stride, w_row_stride, y_row_stride;   // elements
    int batch, n, k, splits, chunks_per_split;           // chunk = 128 inputs
    int epilogue;                // 0 none; 1 out = round(y) + aux (residual add); 2 out[:, i] = silu(y[:, i]) * y[:, n/2 + i] (stacked gate / up)
    const uint16_t *aux;         // epilogue 1: the residual [batch, n]
    int64_t aux_row_stride;
    const uint16_t *norm_w;      // non-null: x is RMS-normalised with this weight [k] on the way in (linear_gemv_kernel<.., NORM>)
    float norm_eps;
};


// linear_tile.hip: the LDS-DMA tile kernel for 17..64 rows.  0 = launched (p.partial set when more than two K splits left fp32 partials:
// the caller runs linear_reduce_kernel), 1 = shape not served, -1 = error.  dtype: ATOMA_F16 / ATOMA_BF16.
int launch_linear_tile(LinearParams &p, int dtype, hipStream_t stream);
// true when launch_linear_tile would serve this product AND leave fp32 partials (more than two K splits)
bool linear_tile_leaves_partials(const LinearParams &p, int dtype);
// the q/k/v projection with RoPE and the KV-cache write as the tile kernel's epilogue (ONE launch): possible when the product is served,
// not split over K beyond what the launch merges itself, and a tile holds both halves of a head
bool linear_tile_can_rope(const LinearParams &p, int head_dim);
int launch_linear_tile_rope(LinearParams &p, int dtype, hipStream_t stream, const uint16_t *cos_t, const uint16_t *sin_t, const int64_t *positions,
                            const int64_t *slot_mapping, uint16_t *k_cache, uint16_t *v_cache, int64_t block_stride, int64_t table_rows, int heads_q,
                            int heads_kv, int head_dim, int page_size, int per_op);

// linear_wide.hip: the LDS-DMA tile kernel for 65..256 rows (K splits merged inside the launch).  0 = launched, 1 = shape not served, -1 = error.
int launch_linear_wide(LinearParams &p, int dtype, hipStream_t stream);
bool linear_wide_can_rope(const LinearParams &p, int head_dim);
int launch_linear_wide_rope(LinearParams &p, int dtype, hipStream_t stream, const uint16_t *cos_t, const uint16_t *sin_t, const int64_t *positions,
                            const int64_t *slot_mapping, uint16_t *k_cache, uint16_t *v_cache, int64_t block_stride, int64_t table_rows, int heads_q,
                            int heads_kv, int head_dim, int page_size, int per_op);

}  // namespace atoma
