// The three element-wise ops between the kernels of a Llama decode step (what a whole step on the
// device still needs around the attention path and the projections):
//   embedding gather      /root/reference/models/src/llama.rs:456-458  (candle_nn::Embedding::forward = index_select)
//   residual add          llama.rs:404,409                              ((attn + residual), (mlp + residual))
//   SiLU(gate) * up       llama.rs:364-365                              (candle_nn::ops::silu(c_fc1(x)) * c_fc2(x))
// Pure byte movement / one or two flops per element: HBM-bound, 16 bytes per lane, bit-exact or one rounding per op
// as Candle's kernels do (silu is evaluated in f32 and rounded, the product rounds again: two ops in the reference).
#include "common.h"

namespace atoma {

typedef unsigned int eu32x4 __attribute__((ext_vector_type(4)));

template <typename IDX>
__global__ void __launch_bounds__(256) embedding_kernel(const IDX *__restrict__ ids, const eu32x4 *__restrict__ table, eu32x4 *__restrict__ out,
                                                        int chunks_per_row, int64_t table_row_chunks, int64_t vocab) {
    const int64_t t = blockIdx.x;
    int64_t id = (int64_t)ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);    // never read outside the table (the reference would fault)
    for (int c = threadIdx.x; c < chunks_per_row; c += blockDim.x) out[t * chunks_per_row + c] = table[id * table_row_chunks + c];
}

template <typename T> __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {
    return pack2<T>(lo_to_f32<T>(a) + lo_to_f32<T>(b), hi_to_f32<T>(a) + hi_to_f32<T>(b));
}
template <typename T> __global__ void __launch_bounds__(256) add_kernel(const eu32x4 *__restrict__ a, const eu32x4 *__restrict__ b, eu32x4 *__restrict__ out, int64_t chunks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const eu32x4 x = a[i], y = b[i];
    out[i] = eu32x4{add2<T>(x[0], y[0]), add2<T>(x[1], y[1]), add2<T>(x[2], y[2]), add2<T>(x[3], y[3])};
}

template <typename T> __device__ __forceinline__ float silu_rounded(float g) {   // silu in f32, rounded to the storage dtype
    return round_through<T>(g / (1.f + __expf(-g)));
}
template <typename T> __device__ __forceinline__ uint32_t silu_mul2(uint32_t g, uint32_t u) {
    return pack2<T>(silu_rounded<T>(lo_to_f32<T>(g)) * lo_to_f32<T>(u), silu_rounded<T>(hi_to_f32<T>(g)) * hi_to_f32<T>(u));
}
// gate and up are rows of (possibly the same) wider tensors: row strides in 16-byte chunks
template <typename T> __global__ void __launch_bounds__(256)
silu_mul_kernel(const eu32x4 *__restrict__ gate, const eu32x4 *__restrict__ up, eu32x4 *__restrict__ out, int chunks_per_row,
                int64_t gate_row_chunks, int64_t up_row_chunks, int64_t out_row_chunks) {
    const int64_t t = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= chunks_per_row) return;
    const eu32x4 g = gate[t * gate_row_chunks + c], u = up[t * up_row_chunks + c];
    out[t * out_row_chunks + c] = eu32x4{silu_mul2<T>(g[0], u[0]), silu_mul2<T>(g[1], u[1]), silu_mul2<T>(g[2], u[2]), silu_mul2<T>(g[3], u[3])};
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace atoma

extern "C" {

int atoma_embedding(const void *ids, int ids_are_i64, const void *table, void *out, int64_t num_tokens, int64_t hidden, int64_t vocab,
                    int64_t table_row_stride, int dtype, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("embedding: dtype must be f16 or bf16"); return -1; }
    if (hidden <= 0 || hidden % 8 || table_row_stride % 8 || table_row_stride < hidden) { set_error("embedding: hidden and the table row stride must be multiples of 8 elements"); return -1; }
    if (vocab <= 0) { set_error("embedding: vocab must be positive"); return -1; }
    if (!aligned16(table) || !aligned16(out)) { set_error("embedding: table and out must be 16-byte aligned"); return -1; }
    if (num_tokens <= 0) return 0;
    const auto s = static_cast<hipStream_t>(stream);
    const int cpr = (int)(hidden / 8);
    const int threads = cpr >= 256 ? 256 : 64;
    if (ids_are_i64)
        hipLaunchKernelGGL((embedding_kernel<int64_t>), dim3((unsigned)num_tokens), dim3(threads), 0, s, static_cast<const int64_t *>(ids),
                           static_cast<const eu32x4 *>(table), static_cast<eu32x4 *>(out), cpr, table_row_stride / 8, vocab);
    else
        hipLaunchKernelGGL((embedding_kernel<int32_t>), dim3((unsigned)num_tokens), dim3(threads), 0, s, static_cast<const int32_t *>(ids),
                           static_cast<const eu32x4 *>(table), static_cast<eu32x4 *>(out), cpr, table_row_stride / 8, vocab);
    return ATOMA_CHECK_LAUNCH("embedding") ? 0 : -1;
}

int atoma_add(const void *a, const void *b, void *out, int64_t count, int dtype, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("add: dtype must be f16 or bf16"); return -1; }
    if (count < 0 || count % 8) { set_error("add: count must be a multiple of 8 elements"); return -1; }
    if (!aligned16(a) || !aligned16(b) || !aligned16(out)) { set_error("add: tensors must be 16-byte aligned"); return -1; }
    if (count == 0) return 0;
    const int64_t chunks = count / 8;
    const auto s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)cdiv(chunks, 256));
    if (dtype == ATOMA_BF16)
        hipLaunchKernelGGL((add_kernel<bf16_t>), grid, dim3(256), 0, s, static_cast<const eu32x4 *>(a), static_cast<const eu32x4 *>(b), static_cast<eu32x4 *>(out), chunks);
    else
        hipLaunchKernelGGL((add_kernel<f16_t>), grid, dim3(256), 0, s, static_cast<const eu32x4 *>(a), static_cast<const eu32x4 *>(b), static_cast<eu32x4 *>(out), chunks);
    return ATOMA_CHECK_LAUNCH("add") ? 0 : -1;
}

int atoma_silu_mul(const void *gate, const void *up, void *out, int64_t rows, int64_t width, int64_t gate_row_stride, int64_t up_row_stride,
                   int64_t out_row_stride, int dtype, void *stream) {
    using namespace atoma;
    clear_error();
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("silu_mul: dtype must be f16 or bf16"); return -1; }
    if (width <= 0 || width % 8 || gate_row_stride % 8 || up_row_stride % 8 || out_row_stride % 8) { set_error("silu_mul: width and row strides must be multiples of 8 elements"); return -1; }
    if (gate_row_stride < width || up_row_stride < width || out_row_stride < width) { set_error("silu_mul: row strides must cover a row"); return -1; }
    if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) { set_error("silu_mul: tensors must be 16-byte aligned"); return -1; }
    if (rows <= 0) return 0;
    const int cpr = (int)(width / 8);
    const dim3 grid((unsigned)cdiv(cpr, 256), (unsigned)rows);
    const auto s = static_cast<hipStream_t>(stream);
    if (dtype == ATOMA_BF16)
        hipLaunchKernelGGL((silu_mul_kernel<bf16_t>), grid, dim3(256), 0, s, static_cast<const eu32x4 *>(gate), static_cast<const eu32x4 *>(up),
                           static_cast<eu32x4 *>(out), cpr, gate_row_stride / 8, up_row_stride / 8, out_row_stride / 8);
    else
        hipLaunchKernelGGL((silu_mul_kernel<f16_t>), grid, dim3(256), 0, s, static_cast<const eu32x4 *>(gate), static_cast<const eu32x4 *>(up),
                           static_cast<eu32x4 *>(out), cpr, gate_row_stride / 8, up_row_stride / 8, out_row_stride / 8);
    return ATOMA_CHECK_LAUNCH("silu_mul") ? 0 : -1;
}

}  // extern "C"
