// Parameter block shared by the attention kernels: the meaningful subset of run_mha's 47
// arguments (/root/reference/csrc/src/ffi.rs:4-64, flash_api.cu:78-155 fills Flash_fwd_params
// the same way).  Strides in ELEMENTS.
#pragma once
#include "common.h"

namespace atoma {

struct AttnParams {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    float *lse;                      // may be nullptr
    const float *alibi_slopes;       // may be nullptr
    const int *cu_seqlens_q;         // [b+1] or nullptr
    const int *cu_seqlens_k;         // [b+1] cumulative, [b] lengths, or nullptr
    const int *seqused_k;            // [b] or nullptr
    const int *block_table;          // [b, max_blocks] or nullptr
    int64_t q_batch_stride, k_batch_stride, v_batch_stride, o_batch_stride;
    int64_t q_row_stride, k_row_stride, v_row_stride, o_row_stride;
    int64_t q_head_stride, k_head_stride, v_head_stride, o_head_stride;
    int64_t block_table_batch_stride;
    int alibi_batch_stride;
    int page_size;
    int b, h, h_k, d;
    int seqlen_q, seqlen_k;          // maxima (or the fixed lengths when cu_seqlens_* is null)
    int total_q;                     // rows of q when varlen (for the unpadded LSE layout), else 0
    int is_seqlens_k_cumulative;
    int is_causal;
    int unpadded_lse;
    int pp_pair;                     // prefill ping-pong variant: which wavefronts of a workgroup are treated as SIMD partners
    float scale, scale_log2;
};

// /root/reference/csrc/kernels/block_info.h:11-39
struct SeqInfo {
    int sum_q, sum_k, len_q, len_k;
    __device__ __forceinline__ SeqInfo(const AttnParams &p, int b) {
        sum_q = p.cu_seqlens_q ? p.cu_seqlens_q[b] : -1;
        sum_k = (p.cu_seqlens_k && p.is_seqlens_k_cumulative) ? p.cu_seqlens_k[b] : -1;
        len_q = p.cu_seqlens_q ? p.cu_seqlens_q[b + 1] - sum_q : p.seqlen_q;
        const int cache = !p.cu_seqlens_k ? p.seqlen_k
                                          : (p.is_seqlens_k_cumulative ? p.cu_seqlens_k[b + 1] - sum_k : p.cu_seqlens_k[b]);
        len_k = p.seqused_k ? p.seqused_k[b] : cache;
    }
    __device__ __forceinline__ int64_t q_offset(int64_t batch_stride, int64_t row_stride, int b) const {
        return sum_q < 0 ? (int64_t)b * batch_stride : (int64_t)sum_q * row_stride;
    }
    __device__ __forceinline__ int64_t k_offset(int64_t batch_stride, int64_t row_stride, int b) const {
        return sum_k < 0 ? (int64_t)b * batch_stride : (int64_t)sum_k * row_stride;
    }
};

void launch_attn_generic(const AttnParams &p, bool is_bf16, hipStream_t stream);
bool prefill_mfma_supported(const AttnParams &p);
void launch_prefill_mfma(const AttnParams &p, bool is_bf16, hipStream_t stream);

}  // namespace atoma
