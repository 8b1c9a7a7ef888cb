// fp8 (OCP e4m3fn) KV cache: the cache-write side (/root/reference/README.md:35 roadmap "quantization").
//
// Same slot arithmetic as reshape_and_cache_flash (/root/reference/csrc/kernels/cache_manager.cu:139-170: slot -> page
// slot / block_size, row slot % block_size, padding slots < 0 skipped), same cache layout [nb, page, h_k, d], ONE byte per
// element: byte = e4m3fn(clamp(f32(x) * (1 / scale[head]), -448, 448)), round-to-nearest-even, where scale[head] is the
// per-kv-head DEQUANTISATION scale the decode kernel multiplies back (paged_decode_fp8_kernel).  Index work and a
// deterministic conversion: bit-exact against oracle/fp8_oracle.py.  copy_blocks / swap_blocks are byte movers and serve
// fp8 caches unchanged (pass the page size in bytes).
#include "common.h"

namespace atoma {

// 8 source elements (one 16-byte load) -> 8 fp8 bytes (one 8-byte store)
template <typename T> __device__ __forceinline__ uint2 quantize8(const uint4 &v, float inv_scale) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = lo_to_f32<T>(w[e]) * inv_scale;
        f[2 * e + 1] = hi_to_f32<T>(w[e]) * inv_scale;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], -448.f, 448.f);   // saturate: e4m3fn has no infinity
    uint32_t lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    return make_uint2(lo, hi);
}

// one workgroup per token: K row then V row; a thread converts 8 elements at a time
template <typename T>
__global__ void __launch_bounds__(256)
reshape_and_cache_flash_fp8_kernel(const uint16_t *__restrict__ key, const uint16_t *__restrict__ value, uint8_t *__restrict__ key_cache,
                                   uint8_t *__restrict__ value_cache, const int64_t *__restrict__ slot_mapping,
                                   const float *__restrict__ k_scale, const float *__restrict__ v_scale, int64_t block_stride,
                                   int64_t key_stride, int64_t value_stride, int heads, int head_size, int block_size) {
    const int64_t token = blockIdx.x;
    const int64_t slot = slot_mapping[token];
    if (slot < 0) return;  // padding token (cache_manager.cu:152-155)
    const int n = heads * head_size, nv = n >> 3;
    const int64_t dst = (slot / block_size) * block_stride + (slot % block_size) * (int64_t)n;
    for (int i = threadIdx.x; i < 2 * nv; i += blockDim.x) {
        const bool is_v = i >= nv;
        const int c = is_v ? i - nv : i;
        const int head = (c * 8) / head_size;
        const uint16_t *src = (is_v ? value + token * value_stride : key + token * key_stride) + c * 8;
        const float inv = 1.0f / (is_v ? v_scale : k_scale)[head];
        const uint2 q8 = quantize8<T>(*reinterpret_cast<const uint4 *>(src), inv);
        *reinterpret_cast<uint2 *>((is_v ? value_cache : key_cache) + dst + c * 8) = q8;
    }
}

static int launch_cache_fp8(const void *key, const void *value, void *key_cache, void *value_cache, const int64_t *slot_mapping,
                            const float *k_scale, const float *v_scale, int64_t block_stride, int64_t num_tokens, int64_t num_heads,
                            int64_t head_size, int64_t block_size, int64_t key_stride, int64_t value_stride, int src_dtype, hipStream_t stream,
                            const char *who) {
    const std::string w(who);
    if (src_dtype != ATOMA_F16 && src_dtype != ATOMA_BF16) { set_error(w + ": the source dtype must be f16 or bf16"); return -1; }
    if (num_heads <= 0 || head_size <= 0 || head_size % 8 || block_size <= 0) { set_error(w + ": head_size must be a positive multiple of 8, block_size positive"); return -1; }
    if (key_stride % 8 || value_stride % 8 || block_stride % 8) { set_error(w + ": strides must be multiples of 8 elements"); return -1; }
    if ((reinterpret_cast<uintptr_t>(key) | reinterpret_cast<uintptr_t>(value)) & 15u || (reinterpret_cast<uintptr_t>(key_cache) | reinterpret_cast<uintptr_t>(value_cache)) & 7u) {
        set_error(w + ": key / value must be 16-byte aligned, the caches 8-byte aligned");
        return -1;
    }
    if (!k_scale || !v_scale || !slot_mapping) { set_error(w + ": null scale or slot mapping"); return -1; }
    if (num_tokens <= 0) return 0;
    const int64_t work = 2 * (num_heads * head_size / 8);
    int threads = (int)(cdiv(work, 64) * 64);
    threads = threads > 256 ? 256 : threads;
    const dim3 grid((unsigned)num_tokens);
    auto k16 = static_cast<const uint16_t *>(key), v16 = static_cast<const uint16_t *>(value);
    auto kc = static_cast<uint8_t *>(key_cache), vc = static_cast<uint8_t *>(value_cache);
    if (src_dtype == ATOMA_BF16)
        hipLaunchKernelGGL((reshape_and_cache_flash_fp8_kernel<bf16_t>), grid, dim3(threads), 0, stream, k16, v16, kc, vc, slot_mapping, k_scale, v_scale,
                           block_stride, key_stride, value_stride, (int)num_heads, (int)head_size, (int)block_size);
    else
        hipLaunchKernelGGL((reshape_and_cache_flash_fp8_kernel<f16_t>), grid, dim3(threads), 0, stream, k16, v16, kc, vc, slot_mapping, k_scale, v_scale,
                           block_stride, key_stride, value_stride, (int)num_heads, (int)head_size, (int)block_size);
    return ATOMA_CHECK_LAUNCH(who) ? 0 : -1;
}

}  // namespace atoma

extern "C" {

int atoma_rope_qk(void *q, void *k, const void *cos_table, const void *sin_table, const int64_t *positions, int64_t num_tokens,
                  int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride, int64_t k_token_stride, int dtype,
                  int per_op_rounding, void *stream);

int atoma_reshape_and_cache_flash_fp8(const void *key, const void *value, void *key_cache, void *value_cache, const int64_t *slot_mapping,
                                      const float *k_scale, const float *v_scale, int64_t block_stride, int64_t num_tokens, int64_t num_heads,
                                      int64_t head_size, int64_t block_size, int64_t key_stride, int64_t value_stride, int src_dtype, void *stream) {
    atoma::clear_error();
    return atoma::launch_cache_fp8(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale, block_stride, num_tokens, num_heads, head_size,
                                   block_size, key_stride, value_stride, src_dtype, static_cast<hipStream_t>(stream), "reshape_and_cache_flash_fp8");
}

// RoPE(q, k) in place, then the rotated k and v quantised into the fp8 caches: atoma_rope_qk_cache for an fp8 cache.  One
// entry point, two launches (the rotation, then the conversion of the rotated rows) -- bit-identical to atoma_rope_qk followed
// by atoma_reshape_and_cache_flash_fp8 by construction.
int atoma_rope_qk_cache_fp8(void *q, void *k, const void *v, void *k_cache, void *v_cache, const int64_t *slot_mapping, const float *k_scale,
                            const float *v_scale, const void *cos_table, const void *sin_table, const int64_t *positions, int64_t num_tokens,
                            int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride, int64_t k_token_stride,
                            int64_t v_token_stride, int64_t block_stride, int64_t page_size, int dtype, int per_op_rounding, void *stream) {
    if (atoma_rope_qk(q, k, cos_table, sin_table, positions, num_tokens, num_q_heads, num_kv_heads, head_dim, q_token_stride, k_token_stride, dtype,
                      per_op_rounding, stream) != 0)
        return -1;
    return atoma::launch_cache_fp8(k, v, k_cache, v_cache, slot_mapping, k_scale, v_scale, block_stride, num_tokens, num_kv_heads, head_dim, page_size,
                                   k_token_stride, v_token_stride, dtype, static_cast<hipStream_t>(stream), "rope_qk_cache_fp8");
}

}  // extern "C"
