// run_mha: the reference's attention entry point (/root/reference/csrc/src/ffi.rs:4-64,
// definition csrc/kernels/flash_api.cu:22-159) and its dispatch (flash_api.cu:8-20).
#include "attn_params.h"

namespace atoma {

struct DecodeParams;
bool decode_supported(int d);
// defined in paged_decode.hip
void launch_paged_decode_from_attn(const AttnParams &a, bool is_bf16, int num_splits_hint, hipStream_t stream);

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static void run_mha_impl(void *q_ptr, void *k_ptr, void *v_ptr, void *o_ptr, void *softmax_lse_ptr,
                         void *alibi_slopes_ptr, int32_t *cu_seqlens_q_ptr, int32_t *cu_seqlens_k_ptr,
                         bool is_seqlens_k_cumulative, uint32_t q_batch_stride, uint32_t k_batch_stride,
                         uint32_t v_batch_stride, uint32_t o_batch_stride, uint32_t alibi_slopes_batch_stride,
                         uint32_t q_row_stride, uint32_t k_row_stride, uint32_t v_row_stride, uint32_t o_row_stride,
                         uint32_t q_head_stride, uint32_t k_head_stride, uint32_t v_head_stride,
                         uint32_t o_head_stride, uint32_t num_splits, uint32_t b, uint32_t h, uint32_t h_k, uint32_t d,
                         uint32_t d_rounded, float softmax_scale, float scale_softmax_log2, int *block_table,
                         uint32_t block_table_batch_stride, int page_block_size, int *seqused_k, uint32_t seqlen_q,
                         uint32_t seqlen_k, uint32_t seqlen_q_rounded, uint32_t seqlen_k_rounded, int is_bf16,
                         int is_causal, int window_size_left, int window_size_right, float softcap, bool unpadded_lse,
                         bool force_split_kernel, void *softmax_lseaccum_ptr, void *oaccum_ptr, hipStream_t stream) {
    (void)d_rounded; (void)seqlen_q_rounded; (void)seqlen_k_rounded;
    (void)softmax_lseaccum_ptr; (void)oaccum_ptr;  // split scratch is owned by the library (paged_decode.hip)
    // Sliding-window ("local") attention and softcap are compiled out of the reference
    // (csrc/kernels/static_switch.h:8-11,66-83): window sizes are ignored there, only is_causal acts.
    (void)window_size_left; (void)window_size_right; (void)force_split_kernel;
    clear_error();
    if (b == 0 || h == 0 || seqlen_q == 0) return;
    if (softcap != 0.f) { set_error("run_mha: softcap is not supported (compiled out in the reference)"); return; }
    if (h_k == 0 || h % h_k != 0) { set_error("run_mha: number of k/v heads must divide number of heads in query"); return; }
    if (d == 0 || d > 256) { set_error("run_mha: only supports head dimension at most 256"); return; }
    if (d % 8 != 0) { set_error("run_mha: only supports head sizes that are a multiple of 8"); return; }
    if (block_table && (page_block_size <= 0 || page_block_size % 16 != 0)) {
        set_error("run_mha: page_block_size must be a multiple of 16");
        return;
    }
    // 16-byte vector access everywhere (the reference's 128-bit cp.async has the same requirement)
    const uint32_t strides[] = {q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride, k_head_stride,
                                v_head_stride, o_head_stride, q_batch_stride, k_batch_stride, v_batch_stride,
                                o_batch_stride};
    for (uint32_t s : strides)
        if (s % 8 != 0) { set_error("run_mha: strides must be multiples of 8 elements (16-byte rows)"); return; }
    if (!aligned16(q_ptr) || !aligned16(k_ptr) || !aligned16(v_ptr) || !aligned16(o_ptr)) {
        set_error("run_mha: q/k/v/o must be 16-byte aligned");
        return;
    }

    AttnParams a{};
    a.q = static_cast<const uint16_t *>(q_ptr);
    a.k = static_cast<const uint16_t *>(k_ptr);
    a.v = static_cast<const uint16_t *>(v_ptr);
    a.o = static_cast<uint16_t *>(o_ptr);
    a.lse = static_cast<float *>(softmax_lse_ptr);
    a.alibi_slopes = static_cast<const float *>(alibi_slopes_ptr);
    a.cu_seqlens_q = cu_seqlens_q_ptr;
    a.cu_seqlens_k = cu_seqlens_k_ptr;
    a.seqused_k = seqused_k;
    a.block_table = block_table;
    a.q_batch_stride = q_batch_stride; a.k_batch_stride = k_batch_stride;
    a.v_batch_stride = v_batch_stride; a.o_batch_stride = o_batch_stride;
    a.q_row_stride = q_row_stride; a.k_row_stride = k_row_stride;
    a.v_row_stride = v_row_stride; a.o_row_stride = o_row_stride;
    a.q_head_stride = q_head_stride; a.k_head_stride = k_head_stride;
    a.v_head_stride = v_head_stride; a.o_head_stride = o_head_stride;
    a.block_table_batch_stride = block_table_batch_stride;
    a.alibi_batch_stride = (int)alibi_slopes_batch_stride;
    a.page_size = block_table ? page_block_size : 0;
    a.b = (int)b; a.h = (int)h; a.h_k = (int)h_k; a.d = (int)d;
    a.seqlen_q = (int)seqlen_q; a.seqlen_k = (int)seqlen_k;
    a.total_q = 0;
    a.is_seqlens_k_cumulative = is_seqlens_k_cumulative ? 1 : 0;
    a.is_causal = is_causal ? 1 : 0;
    a.unpadded_lse = unpadded_lse ? 1 : 0;
    a.scale = softmax_scale;
    a.scale_log2 = scale_softmax_log2;

    // Dispatch.  The reference picks by (num_splits, force_split_kernel) (flash_api.cu:13-17)
    // because both of its kernels are the same 64/128-row MMA tile; here the choice is by
    // regime: one query row per sequence -> the HBM-streaming decode kernel; several rows ->
    // the MFMA prefill kernel; everything else -> the shape-generic kernel.
    if (seqlen_q == 1 && cu_seqlens_q_ptr == nullptr && decode_supported((int)d)) {
        launch_paged_decode_from_attn(a, is_bf16 != 0, (int)num_splits, stream);
    } else if (prefill_mfma_supported(a)) {
        launch_prefill_mfma(a, is_bf16 != 0, stream);
    } else {
        launch_attn_generic(a, is_bf16 != 0, stream);
    }
}

}  // namespace atoma

extern "C" {

void run_mha_stream(void *q_ptr, void *k_ptr, void *v_ptr, void *o_ptr, void *softmax_lse_ptr, void *alibi_slopes_ptr,
                    int32_t *cu_seqlens_q_ptr, int32_t *cu_seqlens_k_ptr, bool is_seqlens_k_cumulative,
                    uint32_t q_batch_stride, uint32_t k_batch_stride, uint32_t v_batch_stride, uint32_t o_batch_stride,
                    uint32_t alibi_slopes_batch_stride, uint32_t q_row_stride, uint32_t k_row_stride,
                    uint32_t v_row_stride, uint32_t o_row_stride, uint32_t q_head_stride, uint32_t k_head_stride,
                    uint32_t v_head_stride, uint32_t o_head_stride, uint32_t num_splits, uint32_t b, uint32_t h,
                    uint32_t h_k, uint32_t d, uint32_t d_rounded, float softmax_scale, float scale_softmax_log2,
                    int *block_table, uint32_t block_table_batch_stride, int page_block_size, int *seqused_k,
                    uint32_t seqlen_q, uint32_t seqlen_k, uint32_t seqlen_q_rounded, uint32_t seqlen_k_rounded,
                    int is_bf16, int is_causal, int window_size_left, int window_size_right, float softcap,
                    bool unpadded_lse, bool force_split_kernel, void *softmax_lseaccum_ptr, void *oaccum_ptr,
                    void *stream) {
    atoma::run_mha_impl(q_ptr, k_ptr, v_ptr, o_ptr, softmax_lse_ptr, alibi_slopes_ptr, cu_seqlens_q_ptr,
                        cu_seqlens_k_ptr, is_seqlens_k_cumulative, q_batch_stride, k_batch_stride, v_batch_stride,
                        o_batch_stride, alibi_slopes_batch_stride, q_row_stride, k_row_stride, v_row_stride,
                        o_row_stride, q_head_stride, k_head_stride, v_head_stride, o_head_stride, num_splits, b, h,
                        h_k, d, d_rounded, softmax_scale, scale_softmax_log2, block_table, block_table_batch_stride,
                        page_block_size, seqused_k, seqlen_q, seqlen_k, seqlen_q_rounded, seqlen_k_rounded, is_bf16,
                        is_causal, window_size_left, window_size_right, softcap, unpadded_lse, force_split_kernel,
                        softmax_lseaccum_ptr, oaccum_ptr, static_cast<hipStream_t>(stream));
}

// csrc/src/ffi.rs:4-64: default (NULL) stream, as csrc/kernels/flash_api.cu:157.
void run_mha(void *q_ptr, void *k_ptr, void *v_ptr, void *o_ptr, void *softmax_lse_ptr, void *alibi_slopes_ptr,
             int32_t *cu_seqlens_q_ptr, int32_t *cu_seqlens_k_ptr, bool is_seqlens_k_cumulative,
             uint32_t q_batch_stride, uint32_t k_batch_stride, uint32_t v_batch_stride, uint32_t o_batch_stride,
             uint32_t alibi_slopes_batch_stride, uint32_t q_row_stride, uint32_t k_row_stride, uint32_t v_row_stride,
             uint32_t o_row_stride, uint32_t q_head_stride, uint32_t k_head_stride, uint32_t v_head_stride,
             uint32_t o_head_stride, uint32_t num_splits, uint32_t b, uint32_t h, uint32_t h_k, uint32_t d,
             uint32_t d_rounded, float softmax_scale, float scale_softmax_log2, int *block_table,
             uint32_t block_table_batch_stride, int page_block_size, int *seqused_k, uint32_t seqlen_q,
             uint32_t seqlen_k, uint32_t seqlen_q_rounded, uint32_t seqlen_k_rounded, int is_bf16, int is_causal,
             int window_size_left, int window_size_right, float softcap, bool unpadded_lse, bool force_split_kernel,
             void *softmax_lseaccum_ptr, void *oaccum_ptr) {
    atoma::run_mha_impl(q_ptr, k_ptr, v_ptr, o_ptr, softmax_lse_ptr, alibi_slopes_ptr, cu_seqlens_q_ptr,
                        cu_seqlens_k_ptr, is_seqlens_k_cumulative, q_batch_stride, k_batch_stride, v_batch_stride,
                        o_batch_stride, alibi_slopes_batch_stride, q_row_stride, k_row_stride, v_row_stride,
                        o_row_stride, q_head_stride, k_head_stride, v_head_stride, o_head_stride, num_splits, b, h,
                        h_k, d, d_rounded, softmax_scale, scale_softmax_log2, block_table, block_table_batch_stride,
                        page_block_size, seqused_k, seqlen_q, seqlen_k, seqlen_q_rounded, seqlen_k_rounded, is_bf16,
                        is_causal, window_size_left, window_size_right, softcap, unpadded_lse, force_split_kernel,
                        softmax_lseaccum_ptr, oaccum_ptr, nullptr);
}

}  // extern "C"
