// Host-side mirror of the reference's Candle operator layer over the C ABI kernels:
//   /root/reference/csrc/src/lib.rs            flash_attn, flash_attn_varlen(+_with_block_table),
//                                              flash_attn_kv_cache_full  (CustomOp3::cuda_fwd_t bodies)
//   /root/reference/csrc/src/cache_manager.rs  swap_blocks, copy_blocks, reshape_and_cache_flash
//   /root/reference/models/src/flash_attention.rs  FlashAttention::{new, forward}, metadata
// Same checks in the same order with the same error text (what the reference's tests assert:
// csrc/tests/cache_manager_tests.rs:189-191,457,476,492,508,524), same argument derivation for
// run_mha (strides, causal/window canonicalisation, seqlen_k = max_blocks * page for the kv-cache
// entry point).  What is deliberately NOT mirrored: per-call cudaGetDeviceProperties (SURVEY
// B/Q9), per-call scratch allocation (the library owns split scratch), the forked stream +
// host wait per cache op, and the zeros + slice_set output assembly (outputs are written in place).
#include "common.h"

#include <string.h>

extern "C" void run_mha_stream(void *, void *, void *, void *, void *, void *, int32_t *, int32_t *, bool, uint32_t,
                               uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                               uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                               float, float, int *, uint32_t, int, int *, uint32_t, uint32_t, uint32_t, uint32_t, int,
                               int, int, int, float, bool, bool, void *, void *, void *);

namespace atoma {

void *workspace(hipStream_t stream, size_t bytes);  // paged_decode.hip

static int fail(const std::string &m) {
    set_error(m);
    return -1;
}
static std::string dims(const atoma_tensor *t) {
    std::string s = "[";
    for (int i = 0; i < t->rank; ++i) s += (i ? ", " : "") + std::to_string(t->shape[i]);
    return s + "]";
}
static std::string strides_str(const atoma_tensor *t) {
    std::string s = "[";
    for (int i = 0; i < t->rank; ++i) s += (i ? ", " : "") + std::to_string(t->stride[i]);
    return s + "]";
}
static const char *dtype_name(int d) {
    switch (d) {
        case ATOMA_F16: return "F16";
        case ATOMA_BF16: return "BF16";
        case ATOMA_F32: return "F32";
        case ATOMA_U32: return "U32";
        case ATOMA_I64: return "I64";
        case ATOMA_U8: return "U8";
        case ATOMA_I32: return "I32";
        default: return "?";
    }
}
static size_t dtype_size(int d) {
    switch (d) {
        case ATOMA_F16: case ATOMA_BF16: return 2;
        case ATOMA_F32: case ATOMA_U32: case ATOMA_I32: return 4;
        case ATOMA_I64: return 8;
        default: return 1;
    }
}
static bool is_index32(int d) { return d == ATOMA_U32 || d == ATOMA_I32; }
static int64_t round_multiple(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static bool contiguous(const atoma_tensor *t) {
    int64_t expect = 1;
    for (int i = t->rank - 1; i >= 0; --i) {
        if (t->shape[i] != 1 && t->stride[i] != expect) return false;
        expect *= t->shape[i];
    }
    return true;
}
static int64_t numel(const atoma_tensor *t) {
    int64_t n = 1;
    for (int i = 0; i < t->rank; ++i) n *= t->shape[i];
    return n;
}

struct MhaCall {  // the argument block the reference's wrappers assemble for ffi::run_mha
    const atoma_tensor *q, *k, *v;
    atoma_tensor *out;
    const atoma_tensor *alibi = nullptr, *seqlens_q = nullptr, *seqlens_k = nullptr, *block_table = nullptr, *seqused_k = nullptr;
    bool is_seqlens_k_cumulative = true;
    int64_t b = 0, h = 0, h_k = 0, d = 0, seqlen_q = 0, seqlen_k = 0, page = 0;
    uint32_t qs[3] = {0, 0, 0}, ks[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, os[3] = {0, 0, 0};  // batch,row,head
    int is_causal = 0, wl = -1, wr = -1;
    float scale = 1.f;
    bool unpadded_lse = true, force_split = false;
    uint32_t num_splits = 0;
};

static int launch(const MhaCall &c) {
    const float LOG2E = 1.4426950408889634f;
    uint32_t alibi_bs = 0;
    void *alibi_ptr = nullptr;
    if (c.alibi) {
        alibi_ptr = c.alibi->data;
        alibi_bs = c.alibi->rank == 2 ? (uint32_t)c.alibi->stride[0] : 0;
    }
    run_mha_stream(c.q->data, c.k->data, c.v->data, c.out->data, /*softmax_lse*/ nullptr, alibi_ptr,
                   c.seqlens_q ? static_cast<int32_t *>(c.seqlens_q->data) : nullptr,
                   c.seqlens_k ? static_cast<int32_t *>(c.seqlens_k->data) : nullptr, c.is_seqlens_k_cumulative,
                   c.qs[0], c.ks[0], c.vs[0], c.os[0], alibi_bs, c.qs[1], c.ks[1], c.vs[1], c.os[1], c.qs[2], c.ks[2],
                   c.vs[2], c.os[2], c.num_splits, (uint32_t)c.b, (uint32_t)c.h, (uint32_t)c.h_k, (uint32_t)c.d,
                   (uint32_t)round_multiple(c.d, 32), c.scale, c.scale * LOG2E,
                   c.block_table ? static_cast<int *>(c.block_table->data) : nullptr,
                   c.block_table ? (uint32_t)c.block_table->stride[0] : 0u, (int)c.page, c.seqused_k ? static_cast<int *>(c.seqused_k->data) : nullptr, (uint32_t)c.seqlen_q,
                   (uint32_t)c.seqlen_k, (uint32_t)round_multiple(c.seqlen_q, 128),
                   (uint32_t)round_multiple(c.seqlen_k, 128), c.q->dtype == ATOMA_BF16 ? 1 : 0, c.is_causal, c.wl, c.wr,
                   0.f, c.unpadded_lse, c.force_split, nullptr, nullptr, nullptr);
    return has_error() ? -1 : 0;
}

static int check_common(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const char *op) {
    if (q->dtype != k->dtype) return fail("query and key must have the same dtype");
    if (q->dtype != v->dtype) return fail("query and value must have the same dtype");
    if (q->dtype != ATOMA_F16 && q->dtype != ATOMA_BF16)
        return fail(std::string(op) + " is only supported for f16/bf16 (" + dtype_name(q->dtype) + ")");
    if (q->device < 0) return fail(std::string("no cpu support for ") + op);
    return 0;
}
static int check_head(int64_t head_size_og, int64_t num_heads, int64_t num_heads_k) {
    if (head_size_og > 256) return fail("only supports head dimension at most 256 (got " + std::to_string(head_size_og) + ")");
    if (head_size_og % 8 != 0)
        return fail("only supports head sizes that are a multiple of 8 (got " + std::to_string(head_size_og) + ")");
    if (num_heads_k == 0 || num_heads % num_heads_k != 0)
        return fail("number of k/v heads " + std::to_string(num_heads_k) + " must divide number of heads in query " +
                    std::to_string(num_heads));
    return 0;
}
static int check_alibi(const atoma_tensor *alibi, int64_t num_heads) {
    if (!alibi) return 0;
    if (alibi->dtype != ATOMA_F32)
        return fail(std::string("DType mismatch alibi_slopes ") + dtype_name(alibi->dtype) + ", expected F32");
    if (alibi->rank != 1 || alibi->shape[0] != num_heads)
        return fail("shape mismatch alibi_slopes " + dims(alibi) + ", expected " + std::to_string(num_heads));
    if (alibi->device < 0) return fail("alibi_slopes must be a cuda tensor");
    return 0;
}
static int check_out(const atoma_tensor *out, const atoma_tensor *q) {
    if (!out || !out->data) return fail("output tensor missing");
    if (out->dtype != q->dtype || numel(out) != numel(q) || !contiguous(out))
        return fail("output tensor must be contiguous with the shape and dtype of q");
    return 0;
}

// csrc/src/lib.rs:31-343 (FlashAttention::cuda_fwd_t) behind csrc::flash_attn (:392-411)
// window_left / window_right: the reference's Option<usize>, None = negative (causal = (None, Some(0)), lib.rs:399-400)
static int flash_attn(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi, float scale,
                      int64_t window_left, int64_t window_right, atoma_tensor *out) {
    if (int e = check_common(q, k, v, "flash-attn")) return e;
    if (q->rank != 4 || k->rank != 4 || v->rank != 4)
        return fail("flash-attn expects input tensors of rank 4 (q: " + std::to_string(q->rank) + ", k: " +
                    std::to_string(k->rank) + ", v: " + std::to_string(v->rank) + ")");
    if (q->stride[3] != 1) return fail("the last dim of q must be contiguous " + strides_str(q));
    if (k->stride[3] != 1) return fail("the last dim of k must be contiguous " + strides_str(k));
    if (v->stride[3] != 1) return fail("the last dim of v must be contiguous " + strides_str(v));
    const int64_t b = q->shape[0], sq = q->shape[1], h = q->shape[2], d = q->shape[3];
    const int64_t sk = k->shape[1], hk = k->shape[2];
    const int64_t want[4] = {b, sk, hk, d};
    if (memcmp(k->shape, want, sizeof(want))) return fail("shape mismatch q " + dims(q) + " and k " + dims(k));
    if (memcmp(v->shape, want, sizeof(want))) return fail("shape mismatch q " + dims(q) + " and v " + dims(v));
    if (int e = check_head(d, h, hk)) return e;
    if (int e = check_alibi(alibi, h)) return e;
    if (int e = check_out(out, q)) return e;
    MhaCall c;
    c.q = q; c.k = k; c.v = v; c.out = out; c.alibi = alibi;
    c.b = b; c.h = h; c.h_k = hk; c.d = d; c.seqlen_q = sq; c.seqlen_k = sk;
    c.scale = scale;
    // lib.rs:186-229: a window beyond seqlen_k (or None) -> -1; (None, Some(0)) == causal; seqlen_q == 1 without alibi switches it off;
    // a one-sided window gets seqlen_k on the other side
    int wl = (window_left >= 0 && window_left <= sk) ? (int)window_left : -1;
    int wr = (window_right >= 0 && window_right <= sk) ? (int)window_right : -1;
    c.is_causal = (wl < 0 && wr == 0) ? 1 : 0;
    if (sq == 1 && !alibi) c.is_causal = 0;
    if (wl < 0 && wr >= 0) wl = (int)sk;
    if (wl >= 0 && wr < 0) wr = (int)sk;
    c.wl = wl; c.wr = wr;
    c.qs[0] = (uint32_t)q->stride[0]; c.qs[1] = (uint32_t)q->stride[1]; c.qs[2] = (uint32_t)q->stride[2];
    c.ks[0] = (uint32_t)k->stride[0]; c.ks[1] = (uint32_t)k->stride[1]; c.ks[2] = (uint32_t)k->stride[2];
    c.vs[0] = (uint32_t)v->stride[0]; c.vs[1] = (uint32_t)v->stride[1]; c.vs[2] = (uint32_t)v->stride[2];
    c.os[0] = (uint32_t)(sq * h * d); c.os[1] = (uint32_t)(h * d); c.os[2] = (uint32_t)d;
    return launch(c);
}

// csrc/src/lib.rs:606-1103 (FlashAttentionVarLen::cuda_fwd_t)
static int flash_attn_varlen(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                             const atoma_tensor *alibi, const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k,
                             int64_t max_seqlen_q, int64_t max_seqlen_k, float scale, int64_t window_left,
                             int64_t window_right, const atoma_tensor *block_table, atoma_tensor *out, const atoma_tensor *seqused_k = nullptr) {
    if (int e = check_common(q, k, v, "flash-attn-varlen")) return e;
    if (seqused_k) {   // lib.rs:775-791
        if (seqused_k->device < 0 || !is_index32(seqused_k->dtype)) return fail("seqused_k must be a cuda tensor");
        if (!contiguous(seqused_k)) return fail("seqused_k has to be contiguous");
    }
    if (!seqlens_q || seqlens_q->device < 0 || !is_index32(seqlens_q->dtype)) return fail("seqlens_q must be a cuda tensor");
    if (!contiguous(seqlens_q)) return fail("seqlens_q has to be contiguous");
    if (!seqlens_k || seqlens_k->device < 0 || !is_index32(seqlens_k->dtype)) return fail("seqlens_k must be a cuda tensor");
    if (!contiguous(seqlens_k)) return fail("seqlens_k has to be contiguous");
    if (seqlens_q->rank != 1) return fail("seqlens_q must have rank 1");
    const int64_t nseqlens_q = seqlens_q->shape[0];
    const int64_t batch = nseqlens_q - 1;
    if (q->rank != 3) return fail("flash-attn-varlen expects input tensors of rank 3 (q: " + std::to_string(q->rank) +
                                  ", k: " + std::to_string(k->rank) + ", v: " + std::to_string(v->rank));
    const int64_t total_q = q->shape[0], h = q->shape[1], d = q->shape[2];
    (void)total_q;
    if (block_table) {
        if (block_table->device < 0 || !is_index32(block_table->dtype)) return fail("block_table must be a cuda tensor");
        if (block_table->stride[block_table->rank - 1] != 1) return fail("block_table must be contiguous");
    }
    if (!block_table && (k->rank != 3 || v->rank != 3))
        return fail("flash-attn-varlen expects input tensors of rank 3 (q: " + std::to_string(q->rank) + ", k: " +
                    std::to_string(k->rank) + ", v: " + std::to_string(v->rank));
    if (block_table && (k->rank != 4 || v->rank != 4))
        return fail("flash-attn-varlen expects input tensors of rank 4 (q: " + std::to_string(q->rank) + ", k: " +
                    std::to_string(k->rank) + ", v: " + std::to_string(v->rank));
    if (q->stride[2] != 1) return fail("the last dim of q must be contiguous " + strides_str(q));
    if (k->stride[k->rank - 1] != 1) return fail("the last dim of k must be contiguous " + strides_str(k));
    if (v->stride[v->rank - 1] != 1) return fail("the last dim of v must be contiguous " + strides_str(v));
    int64_t page = 0, hk;
    if (block_table) {
        if (block_table->rank != 2 || block_table->shape[0] != batch)
            return fail("shape mismatch of block_table (got " + dims(block_table) + ") expected [" + std::to_string(batch) +
                        ", " + std::to_string(block_table->rank == 2 ? block_table->shape[1] : 0) + "])");
        page = k->shape[1];
        hk = k->shape[2];
        if (page % 16 != 0) return fail("page_block_size must be a multiple of 16, got " + std::to_string(page));
    } else {
        hk = k->shape[1];
    }
    if (batch == 0) return fail("batch_size must be > 0");
    if (int e = check_head(d, h, hk)) return e;
    if (block_table) {
        const int64_t want[4] = {k->shape[0], page, hk, d};
        if (memcmp(k->shape, want, sizeof(want)))
            return fail("shape mismatch of k (got " + dims(k) + ") expected [" + std::to_string(want[0]) + ", " +
                        std::to_string(page) + ", " + std::to_string(hk) + ", " + std::to_string(d) + "])");
        if (memcmp(v->shape, want, sizeof(want)))
            return fail("shape mismatch of v (got " + dims(v) + ") expected [" + std::to_string(want[0]) + ", " +
                        std::to_string(page) + ", " + std::to_string(hk) + ", " + std::to_string(d) + "])");
    } else {
        const int64_t want[3] = {k->shape[0], hk, d};
        if (memcmp(k->shape, want, sizeof(want)))
            return fail("shape mismatch of k (got " + dims(k) + ") expected [" + std::to_string(want[0]) + ", " +
                        std::to_string(hk) + ", " + std::to_string(d) + "])");
        if (memcmp(v->shape, want, sizeof(want)))
            return fail("shape mismatch of v (got " + dims(v) + ") expected [" + std::to_string(want[0]) + ", " +
                        std::to_string(hk) + ", " + std::to_string(d) + "])");
    }
    if (seqlens_k->rank != 1 || seqlens_k->shape[0] != batch + 1)
        return fail("shape mismatch of seqlens_k (got " + dims(seqlens_k) + ") expected " + std::to_string(batch + 1) + ")");
    if (nseqlens_q < 2) return fail("seqlens_q should have a len >= 2 " + std::to_string(nseqlens_q));
    if (int e = check_alibi(alibi, h)) return e;
    if (int e = check_out(out, q)) return e;

    // window canonicalisation (lib.rs:946-995): > max_seqlen_k or None -> -1; causal == (left<0, right==0)
    int wl = (window_left >= 0 && window_left <= max_seqlen_k) ? (int)window_left : -1;
    int wr = (window_right >= 0 && window_right <= max_seqlen_k) ? (int)window_right : -1;
    MhaCall c;
    c.is_causal = (wl < 0 && wr == 0) ? 1 : 0;
    if (wl < 0 && wr >= 0) wl = (int)max_seqlen_k;
    if (wl >= 0 && wr < 0) wr = (int)max_seqlen_k;
    c.wl = wl; c.wr = wr;
    c.q = q; c.k = k; c.v = v; c.out = out; c.alibi = alibi;
    c.seqlens_q = seqlens_q; c.seqlens_k = seqlens_k; c.block_table = block_table; c.seqused_k = seqused_k;
    c.b = batch; c.h = h; c.h_k = hk; c.d = d; c.seqlen_q = max_seqlen_q; c.seqlen_k = max_seqlen_k; c.page = page;
    c.scale = scale;
    c.qs[0] = 0; c.qs[1] = (uint32_t)q->stride[0]; c.qs[2] = (uint32_t)q->stride[1];
    c.os[0] = 0; c.os[1] = (uint32_t)(h * d); c.os[2] = (uint32_t)d;
    if (block_table) {
        c.ks[0] = (uint32_t)k->stride[0]; c.ks[1] = (uint32_t)k->stride[1]; c.ks[2] = (uint32_t)k->stride[2];
        c.vs[0] = (uint32_t)v->stride[0]; c.vs[1] = (uint32_t)v->stride[1]; c.vs[2] = (uint32_t)v->stride[2];
    } else {
        c.ks[0] = 0; c.ks[1] = (uint32_t)k->stride[0]; c.ks[2] = (uint32_t)k->stride[1];
        c.vs[0] = 0; c.vs[1] = (uint32_t)v->stride[0]; c.vs[2] = (uint32_t)v->stride[1];
    }
    c.force_split = block_table != nullptr;
    return launch(c);
}

// csrc/src/lib.rs:1521-1855 (FlashAttentionKvCache::cuda_fwd_t)
static int flash_attn_kv_cache_full(const atoma_tensor *q, const atoma_tensor *kc, const atoma_tensor *vc,
                                    const atoma_tensor *alibi, float scale, const atoma_tensor *block_table,
                                    const atoma_tensor *seqlens_k, int64_t window_left, int64_t window_right, atoma_tensor *out) {
    if (int e = check_common(q, kc, vc, "flash-attn")) return e;
    if (block_table) {
        if (block_table->device < 0 || !is_index32(block_table->dtype)) return fail("block_table must be a cuda tensor");
        if (block_table->stride[block_table->rank - 1] != 1) return fail("block_table must be contiguous");
    }
    if (q->rank != 4 || kc->rank != 4 || vc->rank != 4)
        return fail("flash-attn expects input tensors of rank 4 (q: " + std::to_string(q->rank) + ", k: " +
                    std::to_string(kc->rank) + ", v: " + std::to_string(vc->rank) + ")");
    const int64_t batch = q->shape[0], sq = q->shape[1], h = q->shape[2], d = q->shape[3];
    int64_t max_blocks = 0;
    if (block_table) {
        if (block_table->rank != 2 || block_table->shape[0] != batch)
            return fail("shape mismatch of block_table (got " + dims(block_table) + ") expected [" + std::to_string(batch) +
                        ", " + std::to_string(block_table->rank == 2 ? block_table->shape[1] : 0) + "])");
        max_blocks = block_table->shape[1];
    }
    int64_t page, seqlen_k, hk;
    if (block_table) {
        page = kc->shape[1];
        seqlen_k = max_blocks * page;   // lib.rs:1596
        hk = kc->shape[2];
        if (page % 16 != 0) return fail("page_block_size must be a multiple of 16 when block_table is provided");
    } else {
        page = 0;
        seqlen_k = kc->shape[1];
        hk = kc->shape[2];
    }
    if (q->stride[3] != 1) return fail("the last dim of q must be contiguous " + strides_str(q));
    if (kc->stride[3] != 1) return fail("the last dim of k must be contiguous " + strides_str(kc));
    if (vc->stride[3] != 1) return fail("the last dim of v must be contiguous " + strides_str(vc));
    if (int e = check_alibi(alibi, h)) return e;
    if (int e = check_head(d, h, hk)) return e;   // (the reference leaves these to the kernel; it would misbehave)
    if (seqlens_k) {
        if (seqlens_k->rank != 1 || seqlens_k->shape[0] != batch)
            return fail("shape mismatch of seqlens_k (got " + dims(seqlens_k) + ") expected [" + std::to_string(batch) + "])");
        if (seqlens_k->dtype != ATOMA_U32 && seqlens_k->dtype != ATOMA_I32)
            return fail(std::string("DType mismatch seqlens_k ") + dtype_name(seqlens_k->dtype) + ", expected U32");
        if (seqlens_k->device < 0) return fail("seqlens_k must be a cuda tensor");
        if (seqlens_k->stride[0] != 1)
            return fail("the last dim of seqlens_k must be contiguous " + strides_str(seqlens_k));
    }
    if (int e = check_out(out, q)) return e;
    MhaCall c;
    // lib.rs:1606-1640: windows of seqlen_k or more -> none; causal = (None, Some(0)); one-sided windows get seqlen_k on the other side
    int wl = (window_left >= 0 && window_left < seqlen_k) ? (int)window_left : -1;
    int wr = (window_right >= 0 && window_right < seqlen_k) ? (int)window_right : -1;
    bool is_causal = wl < 0 && wr == 0;
    if (sq == 1 && !alibi) is_causal = false;      // lib.rs:1629-1631
    if (is_causal) wr = 0;
    if (wl < 0 && wr >= 0) wl = (int)seqlen_k;
    if (wr < 0 && wl >= 0) wr = (int)seqlen_k;
    c.is_causal = is_causal ? 1 : 0;
    c.wl = wl; c.wr = wr;
    c.q = q; c.k = kc; c.v = vc; c.out = out; c.alibi = alibi;
    c.seqlens_k = seqlens_k; c.is_seqlens_k_cumulative = seqlens_k == nullptr;
    c.block_table = block_table;
    c.b = batch; c.h = h; c.h_k = hk; c.d = d; c.seqlen_q = sq; c.seqlen_k = seqlen_k; c.page = page;
    c.scale = scale;
    c.unpadded_lse = false;
    c.force_split = block_table != nullptr;
    c.qs[0] = (uint32_t)q->stride[0]; c.qs[1] = (uint32_t)q->stride[1]; c.qs[2] = (uint32_t)q->stride[2];
    c.ks[0] = (uint32_t)kc->stride[0]; c.ks[1] = (uint32_t)kc->stride[1]; c.ks[2] = (uint32_t)kc->stride[2];
    c.vs[0] = (uint32_t)vc->stride[0]; c.vs[1] = (uint32_t)vc->stride[1]; c.vs[2] = (uint32_t)vc->stride[2];
    c.os[0] = (uint32_t)(sq * h * d); c.os[1] = (uint32_t)(h * d); c.os[2] = (uint32_t)d;
    return launch(c);
}

// csrc/src/cache_manager.rs:319-535
static int reshape_and_cache_flash_t(const atoma_tensor *key, const atoma_tensor *value, const atoma_tensor *key_cache,
                                     const atoma_tensor *value_cache, const atoma_tensor *slot_mapping, hipStream_t stream) {
    if (key->dtype != value->dtype || key->dtype != key_cache->dtype || key->dtype != value_cache->dtype)
        return fail("Only support f16/bf16 dtypes and key, value, key_cache and value_cache must have same dtype");
    if (key->dtype != ATOMA_F16 && key->dtype != ATOMA_BF16) return fail("Only support f16/bf16 dtypes must have same dtype");
    if (key->device < 0) return fail("device must be a cuda device");
    if (value->device < 0 || key_cache->device < 0 || value_cache->device < 0 || slot_mapping->device < 0)
        return fail("value, key_cache and value_cache must be on the same device");
    if (key->device != value->device || key->device != key_cache->device || key->device != value_cache->device ||
        key->device != slot_mapping->device)
        return fail("key, value, key_cache, value_cache and slot_mapping must be on the same device");
    if (key->rank != 3 || value->rank != 3)
        return fail("Only support key and value tensors with rank 3 (got " + std::to_string(key->rank) + " and v_rank " +
                    std::to_string(value->rank) + ")");
    if (key_cache->rank != 4) return fail("Only support key_cache tensors with rank 4 (got " + std::to_string(key_cache->rank) + ")");
    if (value_cache->rank != 4)
        return fail("Only support value_cache tensors with rank 4 (got " + std::to_string(value_cache->rank) + ")");
    const int64_t T = key->shape[0], H = key->shape[1], D = key->shape[2];
    const int64_t nb = key_cache->shape[0], bs = key_cache->shape[1];
    if (key_cache->stride[0] != value_cache->stride[0])
        return fail("Only support block_stride == value_cache.stride[0] (got block_stride " +
                    std::to_string(key_cache->stride[0]) + " and value_cache.stride[0] " +
                    std::to_string(value_cache->stride[0]) + ")");
    const int64_t want[4] = {nb, bs, H, D};
    const std::string ws = "[" + std::to_string(nb) + ", " + std::to_string(bs) + ", " + std::to_string(H) + ", " + std::to_string(D) + "]";
    if (memcmp(key_cache->shape, want, sizeof(want))) return fail("Only support key_cache with shape " + ws + " (got " + dims(key_cache) + ")");
    if (memcmp(value_cache->shape, want, sizeof(want)))
        return fail("Only support value_cache with shape " + ws + " (got " + dims(value_cache) + ")");
    const int64_t wv[3] = {T, H, D};
    if (memcmp(value->shape, wv, sizeof(wv)))
        return fail("Only support value with shape [" + std::to_string(T) + ", " + std::to_string(H) + ", " + std::to_string(D) +
                    "] (got " + dims(value) + ")");
    if (slot_mapping->rank != 1 || slot_mapping->shape[0] != T)
        return fail("Only support slot_mapping with shape [" + std::to_string(T) + "] (got " + dims(slot_mapping) + ")");
    if (slot_mapping->dtype != ATOMA_I64) return fail("slot_mapping must be an i64 tensor");
    reshape_and_cache_flash(key->data, value->data, key_cache->data, value_cache->data,
                            static_cast<int64_t *>(slot_mapping->data), key_cache->stride[0], T, H, D, bs,
                            key->stride[0], value->stride[0], (uint32_t)key->dtype, stream);
    return has_error() ? -1 : 0;
}

}  // namespace atoma

extern "C" {

int atoma_flash_attn(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, float softmax_scale, int causal,
                     atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn(q, k, v, nullptr, softmax_scale, -1, causal ? 0 : -1, out);
}
// csrc/src/lib.rs:432-450, 464-487, 506-527, 552-572: the same op with windows (negative = None) and / or ALiBi slopes.  Sliding windows
// and softcap are compiled out of the reference's kernels (csrc/kernels/static_switch.h:8-11,66-83): as there, only the causal
// combination (None, Some(0)) acts, and softcap is accepted and ignored.
int atoma_flash_attn_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, float softmax_scale, int64_t window_size_left,
                              int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn(q, k, v, nullptr, softmax_scale, window_size_left, window_size_right, out);
}
int atoma_flash_attn_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes, float softmax_scale,
                           int causal, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn(q, k, v, alibi_slopes, softmax_scale, -1, causal ? 0 : -1, out);
}
int atoma_flash_attn_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                    float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn(q, k, v, alibi_slopes, softmax_scale, window_size_left, window_size_right, out);
}
int atoma_flash_attn_alibi_windowed_with_softcap(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                                 float softmax_scale, int64_t window_size_left, int64_t window_size_right, float softcap,
                                                 atoma_tensor *out) {
    (void)softcap;
    return atoma_flash_attn_alibi_windowed(q, k, v, alibi_slopes, softmax_scale, window_size_left, window_size_right, out);
}

// csrc/src/lib.rs:1160-1188
int atoma_flash_attn_varlen(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                            const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q,
                            int64_t max_seqlen_k, float softmax_scale, int causal, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_varlen(q, k, v, nullptr, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
                                    -1, causal ? 0 : -1, nullptr, out);
}

// csrc/src/lib.rs:1392-1420
int atoma_flash_attn_varlen_with_block_table(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                             const atoma_tensor *alibi_slopes, const atoma_tensor *seqlens_q,
                                             const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                             float softmax_scale, int64_t window_size_left, int64_t window_size_right,
                                             const atoma_tensor *block_table, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_varlen(q, k, v, alibi_slopes, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
                                    window_size_left, window_size_right, block_table, out);
}
// csrc/src/lib.rs:1218-1248, 1268-1300, 1328-1360, 1464-1495: the other forms of the varlen op (windows: negative = None; softcap ignored as in
// the reference's build; seqused_k u32 [B] or NULL)
int atoma_flash_attn_varlen_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *seqlens_q,
                                     const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k, float softmax_scale,
                                     int64_t window_size_left, int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_varlen(q, k, v, nullptr, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, window_size_left,
                                    window_size_right, nullptr, out);
}
int atoma_flash_attn_varlen_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                  const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                  float softmax_scale, int causal, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn_varlen(q, k, v, alibi_slopes, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, -1, causal ? 0 : -1,
                                    nullptr, out);
}
int atoma_flash_attn_varlen_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                           const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                           float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn_varlen(q, k, v, alibi_slopes, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, window_size_left,
                                    window_size_right, nullptr, out);
}
int atoma_flash_attn_varlen_full(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                 const atoma_tensor *seqlens_q, const atoma_tensor *seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                 float softmax_scale, int64_t window_size_left, int64_t window_size_right, const atoma_tensor *block_table,
                                 const atoma_tensor *seqused_k, float softcap, atoma_tensor *out) {
    (void)softcap;
    atoma::clear_error();
    return atoma::flash_attn_varlen(q, k, v, alibi_slopes, seqlens_q, seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, window_size_left,
                                    window_size_right, block_table, out, seqused_k);
}

int atoma_flash_attn_kv_cache_full(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v,
                                   const atoma_tensor *alibi_slopes, float softmax_scale, const atoma_tensor *block_table,
                                   const atoma_tensor *seqlens_k, int causal, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_kv_cache_full(q, k, v, alibi_slopes, softmax_scale, block_table, seqlens_k, -1, causal ? 0 : -1, out);
}
// csrc/src/lib.rs:1907-1925, 1949-1966, 1989-2007, 2036-2053: the other forms of the kv-cache op (no block table)
int atoma_flash_attn_kv_cache(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, float softmax_scale, int causal, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_kv_cache_full(q, k, v, nullptr, softmax_scale, nullptr, nullptr, -1, causal ? 0 : -1, out);
}
int atoma_flash_attn_kv_cache_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *seqlens_k,
                                       float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    return atoma::flash_attn_kv_cache_full(q, k, v, nullptr, softmax_scale, nullptr, seqlens_k, window_size_left, window_size_right, out);
}
int atoma_flash_attn_kv_cache_alibi(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                    const atoma_tensor *seqlens_k, float softmax_scale, int causal, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn_kv_cache_full(q, k, v, alibi_slopes, softmax_scale, nullptr, seqlens_k, -1, causal ? 0 : -1, out);
}
int atoma_flash_attn_kv_cache_alibi_windowed(const atoma_tensor *q, const atoma_tensor *k, const atoma_tensor *v, const atoma_tensor *alibi_slopes,
                                             float softmax_scale, int64_t window_size_left, int64_t window_size_right, atoma_tensor *out) {
    atoma::clear_error();
    if (!alibi_slopes) return atoma::fail("alibi_slopes is required");
    return atoma::flash_attn_kv_cache_full(q, k, v, alibi_slopes, softmax_scale, nullptr, nullptr, window_size_left, window_size_right, out);
}

int atoma_reshape_and_cache_flash(const atoma_tensor *key, const atoma_tensor *value, const atoma_tensor *key_cache,
                                  const atoma_tensor *value_cache, const atoma_tensor *slot_mapping) {
    atoma::clear_error();
    return atoma::reshape_and_cache_flash_t(key, value, key_cache, value_cache, slot_mapping, nullptr);
}

// csrc/src/cache_manager.rs:148-307
int atoma_copy_blocks(const atoma_tensor *const *key_caches, int64_t num_key_caches,
                      const atoma_tensor *const *value_caches, int64_t num_value_caches,
                      const atoma_tensor *block_mapping) {
    atoma::clear_error();
    using atoma::fail;
    if (num_key_caches > 0 && num_value_caches > 0) {
        const int kd = key_caches[0]->dtype, vd = value_caches[0]->dtype;
        if (!((kd == ATOMA_F16 && vd == ATOMA_F16) || (kd == ATOMA_BF16 && vd == ATOMA_BF16)))
            return fail("Only support f16/bf16 dtypes and src and dst must have same dtype");
    }
    if (num_key_caches > 0 && key_caches[0]->device < 0) return fail("device must be a cuda device");
    if (num_key_caches != num_value_caches) return fail("key_caches and value_caches must have the same length");
    const int64_t L = num_key_caches;
    if (L == 0) return 0;
    if (value_caches[0]->device < 0) return fail("key_caches and value_caches must be on the same device");
    for (int64_t l = 0; l < L; ++l) {
        if (key_caches[l]->device < 0) return fail("key_caches must be a cuda tensor");
        if (value_caches[l]->device < 0) return fail("value_caches must be a cuda tensor");
    }
    if (block_mapping->rank != 2 || block_mapping->shape[1] != 2) return fail("block_mapping must have shape [num_pairs, 2]");
    if (block_mapping->device < 0) return fail("block_mapping must be a cuda tensor");
    if (block_mapping->dtype != ATOMA_I64) return fail("block_mapping must be an i64 tensor");  // SURVEY B/Q1
    const int64_t P = block_mapping->shape[0];
    int64_t numel_per_block = 1;
    for (int i = 1; i < key_caches[0]->rank; ++i) numel_per_block *= key_caches[0]->shape[i];
    // per-layer base pointers -> device (the reference builds two Candle tensors per call for this)
    std::vector<int64_t> host(2 * (size_t)L);
    for (int64_t l = 0; l < L; ++l) {
        host[(size_t)l] = reinterpret_cast<int64_t>(key_caches[l]->data);
        host[(size_t)(L + l)] = reinterpret_cast<int64_t>(value_caches[l]->data);
    }
    // the workspace serves the split partials too; pointer table lives past the first 16 MiB? keep it simple:
    static thread_local void *dev_tbl = nullptr;
    static thread_local size_t dev_cap = 0;
    if (dev_cap < host.size() * 8) {
        if (dev_tbl) (void)hipFree(dev_tbl);
        dev_cap = host.size() * 8 * 2;
        if (!atoma::check_hip(hipMalloc(&dev_tbl, dev_cap), "copy_blocks pointer table")) { dev_cap = 0; dev_tbl = nullptr; return -1; }
    }
    if (!atoma::check_hip(hipMemcpyAsync(dev_tbl, host.data(), host.size() * 8, hipMemcpyHostToDevice, nullptr),
                          "copy_blocks pointer upload"))
        return -1;
    (void)hipStreamSynchronize(nullptr);  // `host` dies at return; pageable H2D is staged, but be explicit
    auto *tbl = static_cast<int64_t *>(dev_tbl);
    if (key_caches[0]->dtype == ATOMA_F16) copy_blocks_f16(tbl, tbl + L, block_mapping->data, L, P, numel_per_block, nullptr);
    else copy_blocks_bf16(tbl, tbl + L, block_mapping->data, L, P, numel_per_block, nullptr);
    return atoma::has_error() ? -1 : 0;
}

// csrc/src/cache_manager.rs:18-128
int atoma_swap_blocks_tensor(const atoma_tensor *src, atoma_tensor *dst, const uint32_t *mapping_pairs, int64_t num_pairs) {
    atoma::clear_error();
    using atoma::fail;
    int kind;
    if (src->device >= 0 && dst->device >= 0) {
        if (src->device != dst->device) return fail("swap_blocks: Both src and dst tensors should be on the same device to swap");
        kind = ATOMA_SWAP_GPU_TO_GPU;
    } else if (src->device < 0 && dst->device >= 0) kind = ATOMA_SWAP_CPU_TO_GPU;
    else if (src->device >= 0 && dst->device < 0) kind = ATOMA_SWAP_GPU_TO_CPU;
    else
        return fail("swap_blocks: Either src and dst are on the same cuda device, or src and dst are on cpu and cuda devices, alternately");
    if ((src->dtype != ATOMA_F16 && src->dtype != ATOMA_BF16) || src->dtype != dst->dtype)
        return fail(kind == ATOMA_SWAP_GPU_TO_GPU ? "Only support f16/bf16 dtypes and src and dst must have same dtype"
                                                  : "swap_blocks: Invalid combination of src and dst tensors storage to swap");
    int64_t block_bytes = (int64_t)atoma::dtype_size(src->dtype);
    for (int i = 1; i < src->rank; ++i) block_bytes *= src->shape[i];
    std::vector<int64_t> m(2 * (size_t)num_pairs);
    for (int64_t i = 0; i < 2 * num_pairs; ++i) m[(size_t)i] = mapping_pairs[i];
    const int rc = atoma_swap_blocks(src->data, dst->data, m.data(), num_pairs, block_bytes, kind, nullptr);
    return rc;
}

// models/src/flash_attention.rs:198-230
int atoma_flash_attention_new(atoma_flash_attention *self, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim,
                              float softmax_scale, const atoma_tensor *alibi_slopes, int64_t sliding_window,
                              int32_t kv_cache_dtype, int32_t device) {
    atoma::clear_error();
    if (num_kv_heads == 0 || num_heads % num_kv_heads != 0)
        return atoma::fail("number of heads " + std::to_string(num_heads) + " must divide number of kv heads " + std::to_string(num_kv_heads));
    // supported_head_sizes() (flash_attention.rs:233-235) lists 80 and 112, which the reference's own
    // kernels cannot run (uneven-K is compiled out, SURVEY B/Q7); the list is mirrored for the check only.
    const int64_t sizes[] = {64, 80, 96, 112, 128, 192, 256};
    bool ok = false;
    for (int64_t s : sizes) ok |= (s == head_dim);
    if (!ok) return atoma::fail("head_dim " + std::to_string(head_dim) + " is not supported");
    self->num_heads = num_heads; self->num_kv_heads = num_kv_heads; self->head_dim = head_dim;
    self->softmax_scale = softmax_scale; self->alibi_slopes = alibi_slopes; self->sliding_window = sliding_window;
    self->kv_cache_dtype = kv_cache_dtype; self->device = device;
    return 0;
}

// models/src/flash_attention.rs:322-469
int atoma_flash_attention_forward(const atoma_flash_attention *self, const atoma_tensor *q, const atoma_tensor *k,
                                  const atoma_tensor *v, const atoma_tensor *kv_cache, const atoma_attn_metadata *meta,
                                  atoma_tensor *out) {
    atoma::clear_error();
    using atoma::fail;
    using std::to_string;
    if (q->rank != 3 || k->rank != 3 || v->rank != 3) return fail("query, key and value must have rank 3");
    const int64_t qT = q->shape[0], qH = q->shape[1], qD = q->shape[2];
    const int64_t kT = k->shape[0], kH = k->shape[1], kD = k->shape[2];
    const int64_t vT = v->shape[0], vH = v->shape[1], vD = v->shape[2];
    if (qT != kT || qT != vT)
        return fail("query, key, and value must have the same number of tokens (got " + to_string(qT) + ", " + to_string(kT) + ", " + to_string(vT) + ")");
    if (kH != vH || kD != vD)
        return fail("key and value must have the same shape (got [" + to_string(kT) + ", " + to_string(kH) + ", " + to_string(kD) +
                    "], [" + to_string(vT) + ", " + to_string(vH) + ", " + to_string(vD) + "])");
    if (qH != self->num_heads || qD != self->head_dim)
        return fail("query must have [num_head, hidden_dim] = [" + to_string(self->num_heads) + ", " + to_string(self->head_dim) +
                    "] (got [" + to_string(qH) + ", " + to_string(qD) + "])");
    if (kH != self->num_kv_heads || kD != self->head_dim)
        return fail("key must have k_num_heads = " + to_string(self->num_kv_heads) + " and hidden dim " + to_string(self->head_dim) +
                    " (got " + to_string(kH) + ", " + to_string(kD) + ")");
    // split_kv_cache (flash_attention.rs:247-279)
    if (kv_cache->rank != 5) return fail("KV cache must have rank 5 (got " + to_string(kv_cache->rank) + ")");
    if (kv_cache->shape[0] != 2) return fail("KV cache must have cache_size 2 (got " + to_string(kv_cache->shape[0]) + ")");
    if (kv_cache->shape[3] != self->num_kv_heads)
        return fail("KV cache must have num_heads " + to_string(self->num_kv_heads) + " (got " + to_string(kv_cache->shape[3]) + ")");
    if (kv_cache->shape[4] != self->head_dim)
        return fail("KV cache must have head_dim " + to_string(self->head_dim) + " (got " + to_string(kv_cache->shape[4]) + ")");
    atoma_tensor kc{}, vc{};
    kc.dtype = vc.dtype = kv_cache->dtype;
    kc.device = vc.device = kv_cache->device;
    kc.rank = vc.rank = 4;
    for (int i = 0; i < 4; ++i) {
        kc.shape[i] = vc.shape[i] = kv_cache->shape[i + 1];
        kc.stride[i] = vc.stride[i] = kv_cache->stride[i + 1];
    }
    const size_t esz = atoma::dtype_size(kv_cache->dtype);
    kc.data = kv_cache->data;
    vc.data = static_cast<char *>(kv_cache->data) + kv_cache->stride[0] * (int64_t)esz;

    if (atoma::reshape_and_cache_flash_t(k, v, &kc, &vc, meta->slot_mapping, nullptr)) return -1;
    const int64_t np = meta->num_prefill_tokens, nd = meta->num_decoding_tokens;
    if (kT != np + nd) return fail("query must have number of tokens " + to_string(np + nd) + " (got " + to_string(qT) + ")");
    if (!out || !out->data || out->dtype != q->dtype || atoma::numel(out) != qT * qH * qD || !atoma::contiguous(out))
        return fail("output tensor must be contiguous [num_tokens, num_heads * head_dim] with the dtype of q");
    const size_t qsz = atoma::dtype_size(q->dtype);
    // rows no kernel will write stay zero, as in the reference's Tensor::zeros + slice_set
    if ((np > 0 && !meta->has_prefill) || (nd > 0 && !meta->has_decoding)) {
        if (!atoma::check_hip(hipMemsetAsync(out->data, 0, (size_t)(qT * qH * qD) * qsz, nullptr), "zero output")) return -1;
    }
    auto rows = [&](const atoma_tensor *t, int64_t r0, int64_t n) {
        atoma_tensor s = *t;
        s.data = static_cast<char *>(t->data) + r0 * t->stride[0] * (int64_t)atoma::dtype_size(t->dtype);
        s.shape[0] = n;
        return s;
    };
    atoma_tensor out3{};
    out3.data = out->data; out3.dtype = q->dtype; out3.device = q->device; out3.rank = 3;
    out3.shape[0] = qT; out3.shape[1] = qH; out3.shape[2] = qD;
    out3.stride[0] = qH * qD; out3.stride[1] = qD; out3.stride[2] = 1;

    if (meta->has_prefill && np > 0) {
        atoma_tensor qp = rows(q, 0, np), kp = rows(k, 0, np), vp = rows(v, 0, np), op = rows(&out3, 0, np);
        const atoma_tensor *pbt = meta->prefill_block_tables;
        const bool no_prefix = !pbt || atoma::numel(pbt) == 0;
        if (no_prefix) {
            if (!meta->sequence_start_locations) return fail("Missing sequence start locations tensor for prefill inference");
            // flash_attention.rs:399-409: causal = (q_num_tokens > 1)
            if (atoma::flash_attn_varlen(&qp, &kp, &vp, nullptr, meta->sequence_start_locations, meta->sequence_start_locations,
                                         meta->max_prefill_sequence_length, meta->max_prefill_sequence_length,
                                         self->softmax_scale, -1, qT > 1 ? 0 : -1, nullptr, &op))
                return -1;
        } else {
            if (!meta->query_start_locations)
                return fail("Missing query start locations tensor for prefill inference, with prefix enabled attention");
            if (!meta->sequence_start_locations)
                return fail("Missing sequence start locations tensor for prefill inference, with prefix enabled attention");
            // flash_attention.rs:435-448: window_size_right = None, i.e. NOT causal (SURVEY B/Q3, mirrored as is)
            if (atoma::flash_attn_varlen(&qp, &kc, &vc, self->alibi_slopes, meta->query_start_locations,
                                         meta->sequence_start_locations, meta->max_prefill_sequence_length,
                                         meta->max_sequence_length_k, self->softmax_scale, self->sliding_window, -1, pbt, &op))
                return -1;
        }
    }
    if (meta->has_decoding && nd > 0) {
        atoma_tensor qd = rows(q, np, nd), od = rows(&out3, np, nd);
        atoma_tensor q4{}, o4{};
        q4 = qd; q4.rank = 4;                            // unsqueeze(1): [nd, 1, h, d]
        q4.shape[0] = nd; q4.shape[1] = 1; q4.shape[2] = qH; q4.shape[3] = qD;
        q4.stride[0] = q->stride[0]; q4.stride[1] = q->stride[0]; q4.stride[2] = q->stride[1]; q4.stride[3] = q->stride[2];
        o4 = od; o4.rank = 4;
        o4.shape[0] = nd; o4.shape[1] = 1; o4.shape[2] = qH; o4.shape[3] = qD;
        o4.stride[0] = qH * qD; o4.stride[1] = qH * qD; o4.stride[2] = qD; o4.stride[3] = 1;
        if (atoma::flash_attn_kv_cache_full(&q4, &kc, &vc, self->alibi_slopes, self->softmax_scale, meta->decoding_block_tables,
                                            meta->decoding_sequence_lengths, -1, 0, &o4))      // causal = (None, Some(0)), flash_attention.rs:459
            return -1;
    }
    return 0;
}

}  // extern "C"
