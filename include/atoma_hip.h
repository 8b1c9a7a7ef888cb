/*
 * atoma_hip.h -- C ABI of libatoma_hip.so, the MI355X (gfx950) drop-in for atoma-infer's
 * paged-attention hot path.
 *
 * SECTION 1 is the reference's own FFI boundary, symbol for symbol and argument for
 * argument: /root/reference/csrc/src/ffi.rs:3-102 (definitions: csrc/kernels/flash_api.cu:22-159,
 * csrc/kernels/cache_manager.cu:43-81,215-242).  A Rust build of the reference links this
 * library instead of `libflashattention.a` + `cudart` (csrc/build.rs:105-113) and needs no
 * source change in csrc/src/ffi.rs.  All strides are in ELEMENTS, all pointers are device
 * pointers unless stated, `void *stream` is a `hipStream_t`.
 *
 * SECTION 2 holds what the reference does NOT route through its C ABI (it delegates to
 * cudarc / Candle / NCCL) but which sits on the same hot path: swap_blocks, RMSNorm, RoPE,
 * the tensor-parallel all-reduce, and the error channel.  Each entry names the reference
 * interface it replaces.
 *
 * SECTION 3 is the host-side mirror of the reference's Candle operator layer
 * (csrc/src/lib.rs, csrc/src/cache_manager.rs, models/src/flash_attention.rs): same
 * function names, argument meaning and error strings, over a plain tensor descriptor.
 *
 * Error convention: the reference's entry points return void and call exit() on a launch
 * failure (csrc/kernels/flash_fwd_launch_template.h:25-31).  Here every entry point records
 * failures in a thread-local slot readable through atoma_last_error(); void signatures are
 * kept so the ABI stays identical.  int-returning extension functions return 0 on success.
 */
#ifndef ATOMA_HIP_H
#define ATOMA_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes: 0/1 are the reference's (csrc/src/cache_manager.rs:384-390) */
enum { ATOMA_F16 = 0, ATOMA_BF16 = 1, ATOMA_F32 = 2, ATOMA_U32 = 3, ATOMA_I64 = 4, ATOMA_U8 = 5, ATOMA_I32 = 6 };

/* ------------------------------------------------------------------------------------------
 * SECTION 1 -- the reference FFI (csrc/src/ffi.rs)
 * ------------------------------------------------------------------------------------------ */

/* csrc/src/ffi.rs:4-64 == csrc/kernels/flash_api.cu:22-159.  FlashAttention-2 forward:
 * dense / varlen prefill, paged split-KV decode ("paged_attention v1/v2") and split combine.
 * Launches on the NULL stream like the reference (flash_api.cu:157).
 *   num_splits <= 1 && !force_split_kernel -> prefill kernel, else split-KV kernel (+combine)
 *   block_table != NULL: k/v are paged caches [nb, page, h_k, d]; k_batch_stride is the page
 *   stride; cu_seqlens_k holds per-sequence lengths when !is_seqlens_k_cumulative.
 *   softmax_lseaccum_ptr / oaccum_ptr: caller scratch, fp32 [num_splits,b,h,seqlen_q] and
 *   [num_splits,b,h,seqlen_q,d_rounded]; only read/written when num_splits > 1. */
void run_mha(void *q_ptr, void *k_ptr, void *v_ptr, void *o_ptr, void *softmax_lse_ptr,
             void *alibi_slopes_ptr, int32_t *cu_seqlens_q_ptr, int32_t *cu_seqlens_k_ptr,
             bool is_seqlens_k_cumulative, uint32_t q_batch_stride, uint32_t k_batch_stride,
             uint32_t v_batch_stride, uint32_t o_batch_stride, uint32_t alibi_slopes_batch_stride,
             uint32_t q_row_stride, uint32_t k_row_stride, uint32_t v_row_stride,
             uint32_t o_row_stride, uint32_t q_head_stride, uint32_t k_head_stride,
             uint32_t v_head_stride, uint32_t o_head_stride, uint32_t num_splits, uint32_t b,
             uint32_t h, uint32_t h_k, uint32_t d, uint32_t d_rounded, float softmax_scale,
             float scale_softmax_log2, int *block_table, uint32_t block_table_batch_stride,
             int page_block_size, int *seqused_k, uint32_t seqlen_q, uint32_t seqlen_k,
             uint32_t seqlen_q_rounded, uint32_t seqlen_k_rounded, int is_bf16, int is_causal,
             int window_size_left, int window_size_right, float softcap, bool unpadded_lse,
             bool force_split_kernel, void *softmax_lseaccum_ptr, void *oaccum_ptr);

/* csrc/src/ffi.rs:66-84 == csrc/kernels/cache_manager.cu:43-81.  Copy-on-write page copies
 * for all layers in one launch: key/value_cache_ptrs are DEVICE arrays of int64 per-layer
 * base pointers, block_mapping a DEVICE int64 [num_pairs][2] of (src,dst) page numbers. */
void copy_blocks_f16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping,
                     int64_t num_layers, int64_t num_pairs, int64_t numel_per_block, void *stream);
void copy_blocks_bf16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping,
                      int64_t num_layers, int64_t num_pairs, int64_t numel_per_block, void *stream);

/* csrc/src/ffi.rs:86-101 == csrc/kernels/cache_manager.cu:215-242.  Scatter new K,V tokens
 * [num_tokens, num_heads, head_size] (row strides key_stride/value_stride) into the paged
 * caches [nb, block_size, num_heads, head_size] at slot_mapping[t] (< 0 = padding, skipped).
 * dtype: 0 = f16, 1 = bf16. */
void reshape_and_cache_flash(void *key, void *value, void *key_cache, void *value_cache,
                             int64_t *slot_mapping, int64_t block_stride, int64_t num_tokens,
                             int64_t num_heads, int64_t head_size, int64_t block_size,
                             int64_t key_stride, int64_t value_stride, uint32_t dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * SECTION 2 -- same hot path, not behind the reference's C ABI
 * ------------------------------------------------------------------------------------------ */

/* Error channel (replaces exit() in csrc/kernels/flash_fwd_launch_template.h:25-31).
 * Returns "" when the calling thread's last library call succeeded. */
const char *atoma_last_error(void);
void atoma_clear_error(void);

/* run_mha with an explicit stream (the reference hard-codes stream 0, flash_api.cu:157). */
void run_mha_stream(void *q_ptr, void *k_ptr, void *v_ptr, void *o_ptr, void *softmax_lse_ptr,
                    void *alibi_slopes_ptr, int32_t *cu_seqlens_q_ptr, int32_t *cu_seqlens_k_ptr,
                    bool is_seqlens_k_cumulative, uint32_t q_batch_stride, uint32_t k_batch_stride,
                    uint32_t v_batch_stride, uint32_t o_batch_stride,
                    uint32_t alibi_slopes_batch_stride, uint32_t q_row_stride,
                    uint32_t k_row_stride, uint32_t v_row_stride, uint32_t o_row_stride,
                    uint32_t q_head_stride, uint32_t k_head_stride, uint32_t v_head_stride,
                    uint32_t o_head_stride, uint32_t num_splits, uint32_t b, uint32_t h,
                    uint32_t h_k, uint32_t d, uint32_t d_rounded, float softmax_scale,
                    float scale_softmax_log2, int *block_table, uint32_t block_table_batch_stride,
                    int page_block_size, int *seqused_k, uint32_t seqlen_q, uint32_t seqlen_k,
                    uint32_t seqlen_q_rounded, uint32_t seqlen_k_rounded, int is_bf16,
                    int is_causal, int window_size_left, int window_size_right, float softcap,
                    bool unpadded_lse, bool force_split_kernel, void *softmax_lseaccum_ptr,
                    void *oaccum_ptr, void *stream);

/* Split count the library would pick (csrc/src/lib.rs:2122-2199, `num_splits_heuristic` and
 * `compute_num_splits`, with the CU count in place of the SM count).  num_cus <= 0: query
 * the current device. */
int atoma_num_splits_heuristic(int64_t batch_nheads_mblocks, int64_t num_sms, int64_t num_n_blocks,
                               int64_t max_splits);
int atoma_compute_num_splits(int64_t batch_size, int64_t num_heads, int64_t head_size,
                             int64_t max_seqlen_k, int64_t max_seqlen_q, int num_cus);

/* swap_blocks (csrc/src/cache_manager.rs:18-128 + csrc/src/ops.rs:14-220: one async memcpy per
 * page there).  dst[page d] = src[page s] for every (s,d) in `mapping` (HOST int64 [n][2]),
 * whole pages of block_size_in_bytes.  kind: 0 = gpu->gpu (same device), 1 = cpu->gpu,
 * 2 = gpu->cpu.  Host pointers that the device can address (atoma_host_alloc, atoma_host_register
 * or hipHostRegister'ed) are moved by one gather/scatter kernel over PCIe.  Pageable host memory
 * (what the unchanged reference passes) travels through a pinned bounce ring owned by the library:
 * two 8 MiB slots, the same kernel between cache and slot, host threads between slot and the
 * caller's pages, the two overlapped; like a pageable hipMemcpy the call then returns when the
 * host side is done.  Device side stream-ordered; the caller syncs the stream. */
enum { ATOMA_SWAP_GPU_TO_GPU = 0, ATOMA_SWAP_CPU_TO_GPU = 1, ATOMA_SWAP_GPU_TO_CPU = 2 };
int atoma_swap_blocks(const void *src, void *dst, const int64_t *mapping, int64_t num_pairs,
                      int64_t block_size_in_bytes, int kind, void *stream);
/* Same for many tensors at once (every layer's K and V: worker.rs:602-632 loops them). */
int atoma_swap_blocks_multi(const void *const *srcs, void *const *dsts, int64_t num_tensors,
                            const int64_t *mapping, int64_t num_pairs, int64_t block_size_in_bytes,
                            int kind, void *stream);
/* Pinned, device-addressable host memory for the CPU KV cache (the reference uses pageable
 * Candle CPU tensors, backends/vllm/src/worker.rs:570-598). */
void *atoma_host_alloc(size_t bytes);
void atoma_host_free(void *p);
/* Pin and map an allocation the caller already owns (one call per CPU cache tensor after it is
 * allocated): its pages then move at the PCIe rate without the bounce ring.  0 on success. */
int atoma_host_register(void *p, size_t bytes);
int atoma_host_unregister(void *p);

/* RMSNorm (models/src/llama.rs:402,408,474 -> candle_nn::ops::rms_norm): per row
 * y = T(rsqrt(mean(x^2) + eps) * x * w), f32 arithmetic, one rounding. */
int atoma_rms_norm(const void *x, const void *weight, void *y, int64_t rows, int64_t hidden,
                   int64_t x_row_stride, int64_t y_row_stride, float eps, int dtype, void *stream);
/* The residual add that precedes every RMSNorm of a decoder layer, fused (llama.rs:404,409 then 402,408,474):
 * sum = round(a + b), y = rms_norm(sum) * weight -- bit-identical to atoma_add followed by atoma_rms_norm.
 * hidden a multiple of 8 (<= 16384), row strides multiples of 8 elements, 16-byte aligned tensors. */
int atoma_add_rms_norm(const void *a, const void *b, const void *weight, void *sum, void *y, int64_t rows, int64_t hidden,
                       int64_t a_row_stride, int64_t b_row_stride, int64_t sum_row_stride, int64_t y_row_stride, float eps, int dtype,
                       void *stream);

/* RoPE, rotate-half (models/src/llama.rs:218-251 -> candle_nn::rotary_emb::rope), with the
 * reference's index_select of the cos/sin rows fused in: x,y [T, heads, d] (token / head
 * strides in elements, d contiguous), tables [max_pos, d/2] in the tensor dtype, positions
 * int64 [T].  per_op_rounding != 0 reproduces Candle's arithmetic in the tensor dtype
 * (every product and the sum rounded); 0 = f32 round-once.  y may alias x.  positions[t] must lie in [0, max_pos) of
 * the tables: they are device data and are not checked (Candle's index_select on CUDA does not check either). */
int atoma_rope(const void *x, void *y, const void *cos_table, const void *sin_table,
               const int64_t *positions, int64_t num_tokens, int64_t num_heads, int64_t head_dim,
               int64_t x_token_stride, int64_t x_head_stride, int64_t y_token_stride,
               int64_t y_head_stride, int dtype, int per_op_rounding, void *stream);
/* q and k in one launch (the reference calls rope twice per layer, llama.rs:296-297). */
int atoma_rope_qk(void *q, void *k, const void *cos_table, const void *sin_table,
                  const int64_t *positions, int64_t num_tokens, int64_t num_q_heads,
                  int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride,
                  int64_t k_token_stride, int dtype, int per_op_rounding, void *stream);
/* Fused RoPE(q, k) + KV-cache write: q and k rotated in place (as atoma_rope_qk), the rotated k and v also
 * stored at slot_mapping[t] of the paged caches (as reshape_and_cache_flash; slot < 0 = padding token, no cache
 * write).  One launch and one pass over k instead of models/src/llama.rs:273-303 (rope) followed by
 * csrc/src/cache_manager.rs:404-535 (reshape_and_cache_flash).  Strides in elements, multiples of 8. */
int atoma_rope_qk_cache(void *q, void *k, const void *v, void *k_cache, void *v_cache, const int64_t *slot_mapping,
                        const void *cos_table, const void *sin_table, const int64_t *positions, int64_t num_tokens,
                        int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride,
                        int64_t k_token_stride, int64_t v_token_stride, int64_t block_stride, int64_t page_size, int dtype,
                        int per_op_rounding, void *stream);

/* q/k/v projection -> RoPE(q, k) -> KV-cache write behind one entry (llama.rs:269-271 -> 273-303 -> cache_manager.rs:404-535):
 * qkv_out [batch, out_row_stride] = x . w_qkv^T with q (heads first) and k rotated in place, the rotated k and v also stored at
 * slot_mapping[t] -- bit for bit what atoma_linear_decode followed by atoma_rope_qk_cache(q = qkv_out, k = qkv_out + h.d,
 * v = qkv_out + (h + h_k).d) leaves.  For 17..64 rows and a matrix whose K is split over more than two workgroups (few rows: the
 * shard of a tensor-parallel rank) the fp32 partials are merged by the RoPE / cache kernel itself (two launches instead of three);
 * every other case runs the two ops. */
int atoma_linear_decode_qkv_rope_cache(const void *x, const void *w_qkv, void *qkv_out, void *k_cache, void *v_cache, const int64_t *slot_mapping,
                                       const void *cos_table, const void *sin_table, const int64_t *positions, int64_t batch, int64_t in_features,
                                       int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t x_row_stride, int64_t w_row_stride,
                                       int64_t out_row_stride, int64_t block_stride, int64_t page_size, int dtype, int per_op_rounding, void *stream);

/* ---- fp8 (OCP e4m3fn) KV cache (the reference's roadmap "quantization", README.md:35) ----
 * The cache keeps the reference's layout [num_blocks, block_size, h_k, d] with ONE byte per element; k_scale / v_scale are
 * DEVICE arrays f32[h_k] of per-kv-head dequantisation scales (value = e4m3 * scale).  copy_blocks_* / atoma_swap_blocks*
 * move bytes and serve these caches unchanged (sizes in bytes; copy_blocks: numel_per_block = page bytes / 2).
 * atoma_reshape_and_cache_flash_fp8: reshape_and_cache_flash (csrc/kernels/cache_manager.cu:139-170, same slot rule) with
 *   byte = e4m3fn(clamp(f32(x) * (1 / scale[head]), -448, 448)), round-to-nearest-even; src_dtype f16 / bf16; block_stride
 *   and key / value strides in elements (multiples of 8).
 * atoma_rope_qk_cache_fp8: atoma_rope_qk_cache for such a cache (q, k rotated in place; rotated k and v quantised).
 * atoma_paged_decode_fp8: flash_attn_kv_cache_full (csrc/src/lib.rs:1521-1855) for seqlen_q = 1 over such a cache:
 *   q [batch, h, 128] / o in f16 / bf16 (strides in elements), block_table int32 [batch, max_blocks], seqlens_k int32 [batch]
 *   on the device; cache strides in bytes; scores = softmax_scale * k_scale[hk] * (q . k_q), O = v_scale[hk] * softmax . v_q.
 *   head_dim 128 only; up to 16 q heads per kv head in one pass over K / V (larger groups in chunks of 16). */
int atoma_reshape_and_cache_flash_fp8(const void *key, const void *value, void *key_cache, void *value_cache, const int64_t *slot_mapping,
                                      const float *k_scale, const float *v_scale, int64_t block_stride, int64_t num_tokens, int64_t num_heads,
                                      int64_t head_size, int64_t block_size, int64_t key_stride, int64_t value_stride, int src_dtype, void *stream);
int atoma_rope_qk_cache_fp8(void *q, void *k, const void *v, void *k_cache, void *v_cache, const int64_t *slot_mapping, const float *k_scale,
                            const float *v_scale, const void *cos_table, const void *sin_table, const int64_t *positions, int64_t num_tokens,
                            int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t q_token_stride, int64_t k_token_stride,
                            int64_t v_token_stride, int64_t block_stride, int64_t page_size, int dtype, int per_op_rounding, void *stream);
int atoma_paged_decode_fp8(const void *q, const void *k_cache, const void *v_cache, void *o, const float *k_scale, const float *v_scale,
                           const int32_t *block_table, const int32_t *seqlens_k, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
                           int64_t head_dim, int64_t block_table_batch_stride, int64_t page_size, int64_t q_batch_stride,
                           int64_t q_head_stride, int64_t o_batch_stride, int64_t o_head_stride, int64_t cache_block_stride,
                           int64_t cache_row_stride, int64_t cache_head_stride, float softmax_scale, int dtype, void *stream);

/* ---- KV block container: a checksummed, self-describing image of a set of pages of EVERY layer, for
 * handing KV blocks to another engine or to disk.  The reference only swaps raw pages between two caches of one process
 * (csrc/src/cache_manager.rs:18-128, worker.rs:602-632); this is that swap with one contiguous destination.  Image =
 * header | int64 block ids [n] | (fp8 only) f32 scales [2][layers][h_k] | zeros up to a multiple of 256 bytes |
 * payload [layer][K|V][block][page bytes].  Format version 2: the checksum covers the header (checksum field zeroed) and
 * everything after it, and a reader re-derives page_bytes / payload_offset / total_bytes from the geometry (overflow-checked)
 * and refuses a header that states anything else -- nothing it dereferences comes from unchecked bytes.
 * k_caches / v_caches: HOST arrays of per-layer DEVICE pointers [nb, block_size, h_k, d]; ids on the HOST; dtype f16 / bf16 /
 * ATOMA_U8 (= fp8 e4m3fn cache).  pack synchronises the stream (it checksums the bytes); unpack is stream-ordered and
 * verifies magic, version, geometry and checksum first.  Pinned host memory (atoma_host_alloc) moves with one gather /
 * scatter launch, pageable memory page by page. */
typedef struct atoma_kv_block_header {
    char magic[8];                 /* "ATOMAKV1" */
    uint32_t version, dtype, num_layers, num_kv_heads, head_dim, block_size;
    uint64_t num_blocks, page_bytes, payload_offset, total_bytes, checksum;
    uint8_t reserved[56];
} atoma_kv_block_header;           /* 128 bytes */
int64_t atoma_kv_blocks_packed_size(int64_t num_layers, int64_t num_kv_heads, int64_t head_dim, int64_t block_size, int64_t num_blocks, int dtype);
int atoma_kv_pack_blocks(const void *const *k_caches, const void *const *v_caches, int64_t num_layers, int64_t num_kv_heads, int64_t head_dim,
                         int64_t block_size, const int64_t *block_ids, int64_t num_blocks, int dtype, const float *k_scales, const float *v_scales,
                         void *out, int64_t out_capacity, void *stream);
int atoma_kv_read_header(const void *packed, int64_t bytes, atoma_kv_block_header *header);
int atoma_kv_unpack_blocks(const void *packed, int64_t bytes, void *const *k_caches, void *const *v_caches, int64_t num_layers, int64_t num_kv_heads,
                           int64_t head_dim, int64_t block_size, int dtype, const int64_t *dst_block_ids, int64_t num_blocks, float *k_scales_out,
                           float *v_scales_out, void *stream);

/* cos/sin table of `Cache::new` (models/src/llama.rs:154-200) built on the HOST into
 * cos_out/sin_out [max_pos, head_dim/2] (storage dtype).  rope_factor <= 0: no Llama-3 scaling. */
int atoma_rope_table(void *cos_out, void *sin_out, int64_t max_pos, int64_t head_dim,
                     float rope_theta, float rope_factor, float low_freq_factor,
                     float high_freq_factor, int64_t original_max_position_embeddings, int dtype);

/* Linear layer of a decode step at small batch: y[b][n] = sum_k x[b][k] * w[n][k] for batch <= 64, fp32 accumulation,
 * one rounding to dtype.  Replaces candle_nn::Linear::forward (a cuBLAS GEMM with 1..64 rows) for the q/k/v/o and MLP
 * projections of models/src/llama.rs:269-271,311,364-365: the weights are streamed once at HBM rate, the arithmetic runs on
 * the matrix cores.  x [batch, in_features], w [out_features, in_features] (the nn.Linear layout), y [batch, out_features];
 * row strides in elements; in_features % 128 == 0, out_features % 16 == 0. */
int atoma_linear_decode(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream);

/* The same with the following op of the layer folded into the kernel that merges the K splits, keeping the reference's
 * rounding points (projection rounded, then the op, rounded again):  _residual: y = residual + x.w^T (o_proj / down_proj
 * followed by the residual add, llama.rs:311+404, 366+409);  _silu_mul: w_gate_up = gate rows then up rows
 * [2.intermediate, in_features], y[b][i] = silu((x.w^T)[b][i]) * (x.w^T)[b][intermediate + i] (llama.rs:364-365). */
int atoma_linear_decode_residual(const void *x, const void *w, const void *residual, void *y, int64_t batch, int64_t in_features,
                                 int64_t out_features, int64_t x_row_stride, int64_t w_row_stride, int64_t residual_row_stride,
                                 int64_t y_row_stride, int dtype, void *stream);
int atoma_linear_decode_silu_mul(const void *x, const void *w_gate_up, void *y, int64_t batch, int64_t in_features, int64_t intermediate,
                                 int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream);

/* The RMSNorm in FRONT of a projection folded into it (llama.rs:402 -> 269-271 for q/k/v, :408 -> 364-365 for gate/up):
 * y = rms_norm(x; norm_weight, eps) . w^T, and the _silu_mul form of the stacked gate/up projection on top of that.  On the rows the
 * VALU streaming kernel serves (1 by default, ATOMA_LINEAR_GEMV_MAX_BATCH up to 2) the projection kernel derives each row's scale itself with atoma_rms_norm's own arithmetic and
 * normalises its input on the way in: bit-identical to atoma_rms_norm followed by atoma_linear_decode[_silu_mul], one launch
 * less per norm.  Larger batches run exactly those two calls, through xn_scratch [batch, in_features] (16-byte aligned; may be
 * NULL when the batch never exceeds the limit).  norm_weight [in_features], 16-byte aligned; in_features <= 16384. */
int atoma_linear_decode_rmsnorm(const void *x, const void *norm_weight, float eps, const void *w, void *y, void *xn_scratch, int64_t batch,
                                int64_t in_features, int64_t out_features, int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride,
                                int dtype, void *stream);
int atoma_linear_decode_rmsnorm_silu_mul(const void *x, const void *norm_weight, float eps, const void *w_gate_up, void *y, void *xn_scratch,
                                         int64_t batch, int64_t in_features, int64_t intermediate, int64_t x_row_stride, int64_t w_row_stride,
                                         int64_t y_row_stride, int dtype, void *stream);

/* The same product at any batch (candle_nn::Linear::forward = a cuBLAS GEMM in the reference, llama.rs:269-271,311,364-365):
 * up to 4 rows (ATOMA_LINEAR_STREAM_MAX_BATCH) it is atoma_linear_decode, above that a plain TN GEMM in the vendor
 * library (hipBLASLt, loaded on first use; bf16 / f16 inputs, fp32 accumulation, one rounding).  Sizes and strides in
 * multiples of 8 elements on the GEMM route, 16-byte aligned tensors. */
int atoma_linear(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                 int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream);

/* The element-wise ops between the kernels of a decode step.  atoma_embedding: out[t] = table[ids[t]] (ids int32 or
 * int64, clamped to the table; models/src/llama.rs:456-458).  atoma_add: out = a + b, one rounding (residual adds,
 * llama.rs:404,409).  atoma_silu_mul: out[t] = silu(gate[t]) * up[t] with the reference's two roundings (llama.rs:364-365);
 * gate / up / out may be row slices of wider tensors (row strides in elements), e.g. the two halves of a fused gate_up
 * projection.  Sizes and strides in multiples of 8 elements, 16-byte aligned tensors. */
int atoma_embedding(const void *ids, int ids_are_i64, const void *table, void *out, int64_t num_tokens, int64_t hidden, int64_t vocab,
                    int64_t table_row_stride, int dtype, void *stream);
int atoma_add(const void *a, const void *b, void *out, int64_t count, int dtype, void *stream);
int atoma_silu_mul(const void *gate, const void *up, void *out, int64_t rows, int64_t width, int64_t gate_row_stride, int64_t up_row_stride,
                   int64_t out_row_stride, int dtype, void *stream);

/* Greedy token selection on the device: out_idx[r] = argmax over logits[r, 0..vocab) (smallest index among the
 * maxima, as numpy; NaNs are never selected), out_val[r] (optional) = that logit as f32.  Replaces the per-sequence
 * `logits.i(idx)` -> LogitsProcessor::sample (ArgMax) -> `to_vec1()[next_token]` device->host copies of
 * backends/vllm/src/model_executor.rs:206-249.  logits [rows, vocab] in dtype f16 / bf16 / f32, row_stride in elements. */
int atoma_argmax_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, int32_t *out_idx,
                      float *out_val, void *stream);

/* The k largest logits of every row with their indices, ordered by (value descending, index ascending; NaNs last):
 * what the top-k / top-p branches of the reference's sampler need from a row (model_executor.rs:206-249), k values per
 * sequence instead of the vocabulary.  out_val / out_idx [rows, k]; k <= 1024. */
int atoma_topk_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, int64_t k, float *out_val,
                    int32_t *out_idx, void *stream);

/* Stochastic token selection on the device: the All / TopK / TopP / TopKThenTopP branches of candle_transformers'
 * LogitsProcessor::sample (configured at backends/vllm/src/llm_service.rs:348-372, called per sequence at model_executor.rs:
 * 230-235 on a row copied to the host).  Weights w_i = exp((x_i - max) / temperature) in f32; top_k in [1, 1024] keeps the k
 * largest logits (0 = off); top_p in (0, 1) keeps, in descending order, the tokens up to and including the one whose
 * cumulative probability reaches top_p (1 = off; without top_k the nucleus is searched among the 1024 most probable tokens);
 * the token is the first kept one whose running weight sum exceeds u[row] * (sum of kept weights), in vocabulary order for
 * plain sampling and in (logit descending, index ascending) order otherwise.  The random stream stays with the caller:
 * u f32 [rows] on the device, uniform in [0, 1).  out_logit (optional) = the chosen token's logit (the reference's log-prob
 * read-back, model_executor.rs:249).  NaN logits have weight 0.  Greedy selection is atoma_argmax_rows. */
int atoma_sample_rows(const void *logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype, float temperature, int64_t top_k,
                      float top_p, const float *u, int32_t *out_idx, float *out_logit, void *stream);

/* Tensor-parallel sum all-reduce (models/src/multi_gpu.rs:141-179 `AllReduce::cuda_fwd`,
 * bootstrap backends/vllm/src/model_executor.rs:413,436-439 `Id::new` / `Comm::from_rank`).
 * One process (or thread) per GPU over RCCL/xGMI.  id128: 128-byte ncclUniqueId. */
int atoma_comm_unique_id(void *id128_out);
int atoma_comm_init(void **comm_out, int rank, int world_size, const void *id128, int device);
int atoma_allreduce_sum(void *comm, const void *in, void *out, int64_t count, int dtype, void *stream);
/* ... and through a communicator: the direct engine runs the fused launch above, the RCCL engine ncclAllReduce + atoma_add_rms_norm.
 * Aliasing: x_out may be `residual` (in-place x += allreduce(partial)) or `in`; norm_out must not overlap in / residual / x_out (checked:
 * the RCCL engine reduces into norm_out as scratch before the norm overwrites it).  In auto mode operands the direct engine cannot take
 * (not 16-byte aligned, larger than its staging region) go to RCCL, as in atoma_allreduce_sum. */
int atoma_allreduce_add_rms_norm(void *comm, const void *in, const void *residual, const void *weight, void *x_out, void *norm_out,
                                 int64_t rows, int64_t hidden, float eps, int dtype, void *stream);
int atoma_comm_destroy(void *comm);
/* Engine behind atoma_allreduce_sum: 0 = RCCL's ncclAllReduce (default), 1 = the direct xGMI kernels below (error when
 * they could not be brought up), 2 = auto (direct up to ATOMA_XGMI_MAX_BYTES, default 8 MiB, when the message is a
 * multiple of 16 bytes and 16-byte aligned; RCCL otherwise).  Initial value from ATOMA_ALLREDUCE=rccl|xgmi|auto.  All
 * ranks must choose the same mode, in the same order of calls: the direct path is built over the RCCL communicator (staging
 * region + all-gather of the handles + an agreement round, both collective and joined by every rank whatever failed locally)
 * only when it is first selected -- inside atoma_comm_init when ATOMA_ALLREDUCE=xgmi|auto (or ATOMA_XGMI_SETUP=1), else inside
 * the first atoma_comm_set_mode(1 | 2); never with ATOMA_XGMI_SETUP=0.  A communicator that stays on RCCL allocates nothing
 * and runs no extra collective.  atoma_comm_info says "xgmi: ready" or why not. */
int atoma_comm_set_mode(void *comm, int mode);
const char *atoma_comm_info(void *comm);

/* Direct sum all-reduce over peer-mapped staging memory (xGMI inside a node; same arithmetic contract as above:
 * models/src/multi_gpu.rs:141-179, out-of-place or in-place, f16 / bf16 / f32): one-shot push for messages up to
 * ATOMA_XGMI_ONESHOT_MAX (default 512 KiB), reduce-scatter + all-gather (two one-hop stages) above; fp32 accumulation in
 * RANK ORDER and one rounding, so every rank holds bit-identical results.  Usable without RCCL:
 *   atoma_xgmi_create   this rank's staging region (uncached device memory, sized for messages up to max_bytes per launch;
 *                       larger messages are cut into pieces), world_size <= 8;
 *   atoma_xgmi_handle   128 bytes to hand to every other rank out of band (like the ncclUniqueId of
 *                       backends/vllm/src/model_executor.rs:413, but one per rank: all-gather them);
 *   atoma_xgmi_connect  handles of ALL ranks in rank order [world_size][128]: peers in the same process (one thread per
 *                       GPU, as the reference runs) are reached through peer access, other processes through HIP IPC;
 *   atoma_xgmi_allreduce_sum  enqueue on `stream` (one stream per communicator; every rank issues the same calls with the
 *                       same counts); count * element size must be a multiple of 16, in / out 16-byte aligned; capturable in
 *                       a hipGraph (the call counter lives in device memory);
 *   atoma_xgmi_status   0, or 1 + the rank a wait timed out on (ATOMA_XGMI_TIMEOUT_MS, default 30 s): never a hung GPU;
 *   atoma_xgmi_destroy  after all ranks drained their streams. */
int atoma_xgmi_create(void **xgmi_out, int rank, int world_size, int device, int64_t max_bytes);
int atoma_xgmi_handle(void *xgmi, void *handle128_out);
int atoma_xgmi_connect(void *xgmi, const void *handles);
int atoma_xgmi_allreduce_sum(void *xgmi, const void *in, void *out, int64_t count, int dtype, void *stream);
/* The tail of both all-reduces of a tensor-parallel decoder layer in the all-reduce's own launch (llama_nccl.rs:139 -> llama.rs:404,408;
 * :195 -> :409 and the next layer's :402): x_out = residual + allreduce_sum(in) (one rounding each, as the separate calls), norm_out =
 * RMSNorm(x_out) * weight.  Bit-identical to atoma_xgmi_allreduce_sum followed by atoma_add_rms_norm; two launches per layer fewer.  in:
 * [rows, hidden] contiguous, rows * hidden * 2 bytes <= the communicator's capacity (one launch); strides in elements; mode as
 * atoma_xgmi_allreduce_sum_mode (0 = by size).  All ranks must call it with the same shapes, like every collective. */
int atoma_xgmi_allreduce_add_rms_norm(void *xgmi, const void *in, const void *residual, const void *weight, void *x_out, void *norm_out,
                                      int64_t rows, int64_t hidden, int64_t residual_row_stride, int64_t x_row_stride,
                                      int64_t norm_row_stride, float eps, int dtype, int mode, void *stream);
/* mode: 0 = by size, 1 = one-shot (message must fit a staging slot), 2 = two-shot -- for A/B measurements and tests */
int atoma_xgmi_allreduce_sum_mode(void *xgmi, const void *in, void *out, int64_t count, int dtype, int mode, void *stream);
int atoma_xgmi_status(void *xgmi);
int64_t atoma_xgmi_capacity(void *xgmi);
int atoma_xgmi_destroy(void *xgmi);

/* ---- host-side batch preparation (backends/vllm/src/worker.rs:224-460, ModelWorker::prepare_input_tensors) ----
 * One sequence of the step as the scheduler describes it (SequenceGroupMetadata / SequenceData): is_prompt,
 * length = sequence_data.length(), num_computed_tokens (prompts), token_chunk_size, all token ids of the sequence,
 * its block table (null for a prompt without chunked prefill), no_block_tables = the group's block-table map is
 * empty (memory profiling: slots are padded with -1). */
typedef struct atoma_seq_desc {
    int32_t is_prompt;
    int32_t no_block_tables;
    int64_t length;
    int64_t num_computed_tokens;
    int64_t token_chunk_size;
    const uint32_t *token_ids;
    const uint32_t *block_table;
    int64_t block_table_len;
} atoma_seq_desc;
/* Where each tensor of ModelInput / FlashAttentionMetadata lies in the packed buffer (byte offsets, 256-byte aligned)
 * and the scalars the reference keeps beside them.  input_tokens u32[num_tokens], input_positions i64[num_tokens],
 * slot_mapping i64[num_slots], seq_lens / context_lens u32[n], query_start_loc / seq_start_loc u32[n + 1],
 * block_tables u32[n][max_block_table_len] padded with 0. */
typedef struct atoma_batch_layout {
    int64_t num_sequences, num_tokens, num_slots, num_prefills, num_prefill_tokens, num_decode_tokens;
    int64_t max_query_len, max_prefill_seq_len, max_decode_seq_len, max_block_table_len;
    int64_t off_input_tokens, off_input_positions, off_slot_mapping, off_seq_lens, off_context_lens;
    int64_t off_query_start_loc, off_seq_start_loc, off_block_tables;
    int64_t total_bytes;
} atoma_batch_layout;
/* Builds all of the above in `host_staging` (pinned memory for an asynchronous copy) and sends them to
 * `device_buffer` with ONE hipMemcpyAsync on `stream` (the reference: one H2D copy per tensor plus one per sequence
 * for the padded block table, worker.rs:411-441,670-683).  host_staging == NULL: only fills `layout` (sizing query);
 * device_buffer == NULL: packs on the host only.  sliding_window 0 = none.  Returns 0, or -1 with the reference's
 * message for an empty decode sequence / a missing block table. */
int atoma_prepare_inputs(const atoma_seq_desc *seqs, int64_t num_sequences, int64_t block_size, int64_t sliding_window,
                         int enable_chunked_prefill, void *host_staging, int64_t host_capacity, void *device_buffer,
                         int64_t device_capacity, atoma_batch_layout *layout, void *stream);

/* Tuning knobs for A/B measurements and tests (returns 0, or -1 for an unknown name).  Decode: "decode_p" (K/V tiles
 * in flight per wavefront, 2..4), "decode_nt" (0/1 non-temporal K/V loads), "decode_stream" (the balanced, kv-head-major line for
 * batches with device-side lengths that fill the chip: 0 = never (one wavefront per (sequence, kv head)), 1 = ragged batches and
 * uniform head_dim-128 GQA batches [default], 2 = every such batch, 3 = ragged batches only), "decode_head_major" (workgroup order of
 * the other launches: kv head slowest 1 [default] / fastest 0), "decode_stream_waves_per_cu", "decode_waves_per_cu" /
 * "decode_min_tiles" (KV split heuristic), "decode_wg_merge" (split-KV merged inside the launch), "decode_line_merge" (the balanced line
 * merges the pieces of a cut sequence inside the launch, by the last wavefront to arrive: 1 [default]; 0 = a second launch of the combine
 * kernel), "decode_pair64" (head_dim 64 with an even number of kv heads and 1 / 2 / 4 q heads per kv head: two kv heads per wavefront on
 * the matrix-core kernel 1 [default] / the dot2 kernel 0), "decode_mqk" (both products on the
 * matrix cores at head_dim 128: bit 0 = groups of more than 4 q heads per kv head, bit 1 = all smaller groups, bit 2 = groups of 2..4
 * when b * h_k <= 64, bit 3 = groups of 2..4 on the line, bit 4 = groups of 2..4 on split-KV launches; default 29), "decode_fp8_mqk"
 * (fp8 KV cache: 1 = the matrix-core kernel [default], 0 = v_dot2c), "decode_fp8_klines" (fp8: K fetched in full 128-byte lines: 0
 * never, 1 where it pays [default], 2 always), "decode_fp8_wg" (fp8: the 8 kv-head wavefronts of a sequence in one workgroup: 0 never
 * [default], 1 split-KV launches, 2 always).  None of them changes a result bit except through the choice of kernel.  Projections at
 * 17..64 rows: "linear_tile" (0 = the older kernels), "linear_tile_nw" / "linear_tile_splits" (force the tile height / the K split),
 * "linear_tile_max_splits" (largest K split the plan considers, default 8; 4 = round 3's plans).  Prefill: "prefill_cfg" (4 = the hand-scheduled persistent
 * kernel of csrc/prefill_asm.hip [default; head_dim 128 -- other shapes run on 0], 0 = tile-sequential kernel, 2 = the
 * software-pipelined one-wave-per-SIMD kernel), "prefill_exact_keys" (kernel 4: query blocks whose first row sees fewer keys take the
 * arithmetic that leaves Q unrounded; default 512, 0 = never, 0x7fffffff = always), "prefill_simple" (A/B knob, default 0).  The
 * persistent kernel keeps its plan table (256 bytes per 256-row query block and wavefront = 1 KiB per block) in the stream's workspace:
 * count it in atoma_warmup's extra_bytes before capturing a graph.  RoPE: "rope_table_rows" (rows of the caller's cos / sin tables; positions beyond
 * them then read the last row instead of memory behind the table -- the FFI carries no table length; 0 = unchecked [default]).
 * "generic_prefill_tile" (64 [default] / 16 / 0), "generic_prefill_kt", "generic_prefill_rq", "generic_decode_stream" (2 [default] / 1 / 0),
 * "generic_decode_waves": kernel choice for head sizes other than 64 / 128 (csrc/attn_generic.hip), for A/B runs; defaults from ATOMA_GENERIC_*.
 * Defaults also come from ATOMA_DECODE_{P,NT,STREAM,WAVES_PER_CU,
 * MIN_TILES,MQK}; ATOMA_PREFILL_CFG overrides "prefill_cfg". */
int atoma_set_option(const char *name, int value);
/* Name and configuration of the decode kernel the dispatcher chose for the last decode call of the calling thread (measurement
 * tools label their numbers with it); "" before any call.  Valid until the thread's next decode call. */
const char *atoma_last_decode_kernel(void);

/* Library-owned scratch (the one piece of state behind the "stateless callee" of csrc/src/lib.rs -- the reference passes
 * caller scratch, lib.rs:1023-1042, and may drop it while the kernel still runs).  One grow-only block per (device, stream)
 * holds the split-KV / balanced-mode partials of the decode kernel and the K-split partials of the projections.
 *   - A block that was handed out is never freed behind the caller: growth allocates a larger block and RETIRES the old one
 *     (a captured hipGraph may have its address in kernel arguments).
 *   - Growth is impossible during stream capture; atoma_warmup sizes the block for every decode call with
 *     batch <= max_batch, these head counts, contexts <= max_seqlen_k, and at least extra_bytes (projection partials:
 *     splits * batch_rows * out_features * 4 with splits <= 8 and batch_rows = the batch rounded up to 64 (17..64 rows) / 128 / 256
 *     (65..256 rows: the K-split slabs of the tile kernels); 8 * 256 * the widest layer * 4 bytes covers every decode projection) --
 *     call it once per stream before capturing, instead of relying on an eager call.
 *   - atoma_release_workspaces frees live and retired blocks of ALL streams: only when no graph that used them will be
 *     replayed and the streams are idle.
 *   - The kernels that merge their own split-K / split-KV pieces count arrivals in 8192 64-bit words per (device, stream).  A word is
 *     tagged with the EPOCH of the launch that used it last (the launch's AQL dispatch id; csrc/sync_ticket.h) and a launch ignores
 *     words of other epochs: nothing has to be zero on entry, nothing is reset, and a launch that ended inconsistently (a graph
 *     replayed beside eager calls on its stream, lengths changed under a running launch) cannot make a LATER launch wrong -- the
 *     callee carries no state from call to call (csrc/src/ffi.rs:3-102).  The pointer is baked into captured graphs like the scratch
 *     block: replay a graph on the stream it was captured on.  ONE residual case (csrc/sync_ticket.h): dispatch ids are per hardware
 *     QUEUE, so a stream handle that is destroyed and created again may come back on a queue whose ids start lower; words that an
 *     inconsistent episode left non-zero then look like "future" epochs to it.  The last arriver zeroes its word, so this needs BOTH an
 *     inconsistent launch AND such a re-mapped stream; atoma_reset_sync_counters(stream) is the recovery after a failed or inconsistent
 *     launch (round 4's repair call, kept for exactly this), harmless otherwise.  atoma_debug_sync_words exposes the words to tests (which fill them with garbage and expect identical bits).
 *   - atoma_warmup_prefill: the hand-scheduled prefill kernel (head_dim 128) plans a call in a table of 1 KiB per 256 query rows and
 *     q head in the same scratch block, padded to sequences x the LONGEST sequence's blocks; size it for the largest prefill call
 *     (longest sequence's query rows, sequences, q heads) before capturing a graph that contains one.  A prefill whose table cannot be allocated launches nothing and says so in atoma_last_error().
 * The device is the calling thread's current device (hipSetDevice), as for every entry point. */
int atoma_warmup(void *stream, int64_t max_batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim, int64_t max_seqlen_k,
                 int64_t extra_bytes);
int atoma_reserve_workspace(void *stream, int64_t bytes);
int atoma_release_workspaces(void);
int atoma_reset_sync_counters(void *stream);
int atoma_warmup_prefill(void *stream, int64_t max_seqlen_q, int64_t max_seqs, int64_t num_heads);
/* Decode dispatch hint (round 6).  The lengths of a decode batch live on the device, so the dispatcher cannot see whether they are all equal; it matters for ONE
 * choice: a launch that gives every resident wavefront about one (sequence, kv head) unit runs 3-5.6 % faster on paged_decode_pair_kernel (two sequences per
 * workgroup, the i-th shortest with the i-th longest) when the batch is ragged and 1.8 % slower when its lengths are all equal.  atoma_prepare_inputs records the
 * shortest / longest decode sequence and their number for the current device whenever it packs a batch (worker.rs:224-460 has them on the host); a caller that
 * packs its own batches can say the same here (0, 0, 0 forgets it).  A hint changes which kernel runs, never a result; option "decode_pair": 0 (default) follows
 * the hint, 1 = every such launch is treated as ragged, 2 = whenever the kernel applies, -1 = never. */
int atoma_hint_decode_lengths(int64_t min_len, int64_t max_len, int64_t count);
int atoma_debug_sync_words(void *stream, void **words_out, int64_t *count_out);
/* tests: the stream's current scratch block (null / 0 when it has none yet): filled with poison between launches by the merge stress tests */
int atoma_debug_workspace(void *stream, void **ptr_out, int64_t *bytes_out);
/* Diagnostics of the arrival tickets: writes the epoch word of ONE tiny launch on `stream` (its AQL dispatch id + 1, shifted by 16) to
 * *epoch_out_device (8 bytes of device memory).  Tests use it to show that epochs grow from launch to launch, also under graph replay. */
int atoma_debug_launch_epoch(void *stream, void *epoch_out_device);

/* Device helpers used by the host layer, tests and bench (plain HIP runtime, no torch). */
int atoma_device_count(void);
int atoma_num_cus(int device);

/* ------------------------------------------------------------------------------------------
 * SECTION 3 -- host mirror of the reference's Candle operator layer
 * ------------------------------------------------------------------------------------------ */

/* A Candle `Tensor` as the ops see it (storage pointer already advanced by start_offset,
 * shape, strides in elements): candle_core::Layout.  device: -1 = host, >= 0 HIP ordinal. */
typedef struct atoma_tensor {
    void *data;
    int32_t dtype;
    int32_t device;
    int32_t rank;
    int32_t _pad;
    int64_t shape[5];
    int64_t stride[5];
} atoma_tensor;

/* csrc::flash_attn (csrc/src/lib.rs:392-411): q [b,sq,h,d], k,v [b,sk,hk,d] -> out [b,sq,h,d]
 * (contiguous, caller-allocated).  Returns 0 or -1 with the reference's error text in
 * atoma_last_error(). */
int atoma_flash_attn(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                     float softmax_scale, int causal, atoma_tensor *out);
/* The reference's other forms of the same op: csrc::flash_attn_windowed (lib.rs:432-450), flash_attn_alibi (:464-487),
 * flash_attn_alibi_windowed (:506-527), flash_attn_alibi_windowed_with_softcap (:552-572).  window_size_* are the reference's
 * Option<usize>: negative = None; (None, Some(0)) is the causal mask.  Sliding windows and softcap are compiled OUT of the reference's
 * kernels (csrc/kernels/static_switch.h:8-11,66-83): there as here only the causal combination acts and softcap is ignored.
 * alibi_slopes f32 [h]. */
int atoma_flash_attn_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, float softmax_scale,
                              int64_t window_size_left, int64_t window_size_right, atoma_tensor *out);
int atoma_flash_attn_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                           float softmax_scale, int causal, atoma_tensor *out);
int atoma_flash_attn_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                    float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out);
int atoma_flash_attn_alibi_windowed_with_softcap(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                                 const atoma_tensor *alibi_slopes, float softmax_scale, int64_t window_size_left,
                                                 int64_t window_size_right, float softcap, atoma_tensor *out);
/* csrc::flash_attn_varlen (lib.rs:1160-1188). q [total_q,h,d], k,v [total_k,hk,d],
 * seqlens_q/k u32 cumulative [B+1]. */
int atoma_flash_attn_varlen(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                            const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k,
                            int64_t max_seqlen_q, int64_t max_seqlen_k, float softmax_scale,
                            int causal, atoma_tensor *out);
/* csrc::flash_attn_varlen_with_block_table (lib.rs:1392-1420). k,v = paged caches
 * [nb,page,hk,d]; window_* < 0 = None; alibi_slopes / block_table may be NULL. */
int atoma_flash_attn_varlen_with_block_table(const atoma_tensor *q, const atoma_tensor *k,
                                             const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                             const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k,
                                             int64_t max_seqlen_q, int64_t max_seqlen_k,
                                             float softmax_scale, int64_t window_size_left,
                                             int64_t window_size_right, const atoma_tensor *block_table,
                                             atoma_tensor *out);
/* csrc::flash_attn_varlen_windowed (lib.rs:1218-1248), _varlen_alibi (:1268-1300), _varlen_alibi_windowed (:1328-1360),
 * flash_attn_varlen_full (:1464-1495; alibi_slopes, block_table, seqused_k [B] u32 may be NULL; softcap ignored as in the reference's build). */
int atoma_flash_attn_varlen_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *seqlens_q,
                                     const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k, float softmax_scale,
                                     int64_t window_size_left, int64_t window_size_right, atoma_tensor *out);
int atoma_flash_attn_varlen_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                  const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                  float softmax_scale, int causal, atoma_tensor *out);
int atoma_flash_attn_varlen_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                           const atoma_tensor *alibi_slopes, const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k,
                                           int64_t max_seqlen_q, int64_t max_seqlen_k, float softmax_scale, int64_t window_size_left,
                                           int64_t window_size_right, atoma_tensor *out);
int atoma_flash_attn_varlen_full(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                 const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                 float softmax_scale, int64_t window_size_left, int64_t window_size_right, const atoma_tensor *block_table,
                                 const atoma_tensor *seqused_k, float softcap, atoma_tensor *out);
/* csrc::flash_attn_kv_cache (lib.rs:1907-1925), _kv_cache_windowed (:1949-1966), _kv_cache_alibi (:1989-2007),
 * _kv_cache_alibi_windowed (:2036-2053): contiguous caches [B,sk,hk,d], seqlens_k u32 [B] or NULL. */
int atoma_flash_attn_kv_cache(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, float softmax_scale, int causal,
                              atoma_tensor *out);
int atoma_flash_attn_kv_cache_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *seqlens_k,
                                       float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out);
int atoma_flash_attn_kv_cache_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                    const atoma_tensor *seqlens_k, float softmax_scale, int causal, atoma_tensor *out);
int atoma_flash_attn_kv_cache_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                             const atoma_tensor *alibi_slopes, float softmax_scale, int64_t window_size_left,
                                             int64_t window_size_right, atoma_tensor *out);
/* csrc::flash_attn_kv_cache_full (lib.rs:2083-2105). q [B,sq,h,d]; caches [B_c,sk,hk,d] or
 * paged [nb,page,hk,d] with block_table [B,max_blocks] u32; seqlens_k u32 [B] or NULL. */
int atoma_flash_attn_kv_cache_full(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                   const atoma_tensor *alibi_slopes, float softmax_scale,
                                   const atoma_tensor *block_table, const atoma_tensor *seqlens_k,
                                   int causal, atoma_tensor *out);
/* csrc::reshape_and_cache_flash (csrc/src/cache_manager.rs:319-535). */
int atoma_reshape_and_cache_flash(const atoma_tensor *key, const atoma_tensor *value,
                                  const atoma_tensor *key_cache, const atoma_tensor *value_cache,
                                  const atoma_tensor *slot_mapping);
/* csrc::copy_blocks (cache_manager.rs:148-307): block_mapping i64 [num_pairs,2] on the device. */
int atoma_copy_blocks(const atoma_tensor *const *key_caches, int64_t num_key_caches,
                      const atoma_tensor *const *value_caches, int64_t num_value_caches,
                      const atoma_tensor *block_mapping);
/* csrc::swap_blocks (cache_manager.rs:18-128): mapping = HOST u32 pairs [(src,dst)...]. */
int atoma_swap_blocks_tensor(const atoma_tensor *src, atoma_tensor *dst, const uint32_t *mapping_pairs,
                             int64_t num_pairs);

/* models::FlashAttentionMetadata + FlashAttention::forward (models/src/flash_attention.rs:
 * 11-146,322-469).  NULL tensor pointers = None. */
typedef struct atoma_attn_metadata {
    const atoma_tensor *slot_mapping;               /* i64 [T] */
    int64_t num_prefill_tokens, num_decoding_tokens;
    /* prefill_metadata */
    int has_prefill;
    const atoma_tensor *prefill_block_tables;       /* u32 [Bp,max_blocks] or NULL */
    int64_t max_prefill_sequence_length;
    const atoma_tensor *query_start_locations;      /* u32 [Bp+1] */
    const atoma_tensor *sequence_start_locations;   /* u32 [Bp+1] */
    int64_t max_sequence_length_k;                  /* max(sequence_lengths) (flash_attention.rs:419) */
    /* decoding_metadata */
    int has_decoding;
    const atoma_tensor *decoding_block_tables;      /* u32 [Bd,max_blocks] */
    const atoma_tensor *decoding_sequence_lengths;  /* u32 [Bd] */
} atoma_attn_metadata;

typedef struct atoma_flash_attention {
    int64_t num_heads, num_kv_heads, head_dim;
    float softmax_scale;
    const atoma_tensor *alibi_slopes;               /* f32 [h] or NULL */
    int64_t sliding_window;                         /* < 0 = None */
    int32_t kv_cache_dtype, device;
} atoma_flash_attention;

/* FlashAttention::new checks (flash_attention.rs:198-230). */
int atoma_flash_attention_new(atoma_flash_attention *self, int64_t num_heads, int64_t num_kv_heads,
                              int64_t head_dim, float softmax_scale, const atoma_tensor *alibi_slopes,
                              int64_t sliding_window, int32_t kv_cache_dtype, int32_t device);
/* FlashAttention::forward: q [T,h,d], k,v [T,hk,d], kv_cache [2,nb,page,hk,d] -> out [T,h*d]. */
int atoma_flash_attention_forward(const atoma_flash_attention *self, const atoma_tensor *q,
                                  const atoma_tensor *k, const atoma_tensor *v,
                                  const atoma_tensor *kv_cache, const atoma_attn_metadata *meta,
                                  atoma_tensor *out);

#ifdef __cplusplus
}
#endif
#endif /* ATOMA_HIP_H */
