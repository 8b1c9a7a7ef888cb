"""One Llama decode step composed from the C-ABI entry points of libatoma_hip.so (test and bench plumbing).

The op sequence is the reference's `Llama::forward` for decode tokens (models/src/llama.rs:392-410, 253-314, 364-365,
456-478 and models/src/flash_attention.rs:322-469): embedding -> per layer [RMSNorm -> q/k/v projection -> RoPE(q, k) +
KV-cache write -> paged decode attention -> o projection -> residual -> RMSNorm -> gate/up projection -> SiLU.up ->
down projection -> residual] -> RMSNorm -> lm_head -> argmax.  q/k/v and gate/up weights are stored concatenated
(row blocks of one matrix), which changes nothing in the arithmetic: every output row is an independent dot product.
Projections: atoma_linear (weight streaming up to 4 rows, hipBLASLt above).  Everything is enqueued on one stream, so a step can be captured in a hipGraph.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import atoma_hip as ah  # noqa: E402

BF16 = 1


class Config:
    def __init__(self, layers, hidden, heads, kv_heads, head_dim, intermediate, vocab, page=16, eps=1e-5, theta=500000.0, max_pos=8192):
        self.layers, self.hidden, self.h, self.hk, self.d = layers, hidden, heads, kv_heads, head_dim
        self.inter, self.vocab, self.page, self.eps, self.theta, self.max_pos = intermediate, vocab, page, eps, theta, max_pos
        self.qkv = (heads + 2 * kv_heads) * head_dim


LLAMA_3_1_8B = Config(32, 4096, 32, 8, 128, 14336, 128256)


class DecodeStep:
    """Device-resident weights, caches and activations of a model + `run()` = one decode step for `batch` sequences."""

    def __init__(self, cfg, batch, num_pages, max_blocks, weights, stream, keep_intermediates=False, fused_epilogues=False, fuse_norm=True,
                 allreduce=None, kv_fp8=False, kv_scale=0.05, own_projections=False, allreduce_norm=None):
        """allreduce(ptr, count): in-place sum of a [batch, hidden] bf16 tensor over the tensor-parallel ranks, enqueued on
        `stream` -- called after the o and the down projection when `cfg` / `weights` are ONE rank's shard
        (llama_nccl.rs:139,195 via TensorParallelRowLinear, multi_gpu.rs:48-50); None = not tensor parallel."""
        c = self.cfg = cfg
        self.B, self.stream, self.keep = batch, stream, keep_intermediates
        self.allreduce = allreduce
        # allreduce_norm(in, residual, weight, x_out, norm_out, rows) -> True when it enqueued "x_out = residual + allreduce(in); norm_out =
        # RMSNorm(x_out) * weight" as ONE launch (atoma_xgmi_allreduce_add_rms_norm / atoma_allreduce_add_rms_norm); False = not available now:
        # the step falls back to allreduce + atoma_add_rms_norm (same bits)
        self.allreduce_norm = allreduce_norm
        self.after_attention = None                        # optional callable(layer), invoked right after a layer's attention call is enqueued (probes: staggering two half-batches)
        # Residual adds, SiLU.up (and at 1 row the RMSNorms) inside the library's own projection kernels.  Up to 128 rows that is the
        # faster step although the vendor GEMM wins most of the single products from 5 rows up (three launches per layer fewer:
        # batch 8 5.17 -> 4.62 ms, 32 7.58 -> 6.96, 64 11.2 -> 10.3, 96 14.9 -> 14.6, 128 16.9 -> 16.8; at 256 rows round 2's
        # linear_big_kernel lost (19.4 -> 20.1), round 6's linear_wide_kernel is level with the vendor GEMM product for product and ahead on the
        # step: 18.88 -> 18.73 ms with NO vendor kernel left in it (tools/probes/c3_own_vs_vendor.py); a TP rank must
        # all-reduce before the residual.
        own_ok = fused_epilogues and not keep_intermediates and batch <= int(os.environ.get("ATOMA_STEP_FUSED_MAX_BATCH", "256"))
        self.fused = own_ok and allreduce is None
        # A tensor-parallel rank: the library's own projection kernels as well -- q/k/v with RoPE and the cache write behind one entry,
        # gate/up with SiLU.up inside; o and down stay plain (their outputs are partial sums: the all-reduce comes before the residual)
        self.tp_own = own_ok and allreduce is not None
        # own_projections: the op-by-op path on atoma_linear_decode at every batch (tests: the fused step must equal it bit for bit)
        self.linear = ah.lib.atoma_linear_decode if (own_projections or self.tp_own or (self.fused and batch > 128)) else ah.lib.atoma_linear
        # 17..256 rows on the fused path: the q/k/v projection, RoPE and the cache write behind one entry (atoma_linear_decode_qkv_rope_cache)
        self.qkv_fused = (self.fused or self.tp_own) and 16 < batch <= 256 and os.environ.get("ATOMA_STEP_QKV_FUSED", "1") != "0"
        self.norm_in_proj = os.environ.get("ATOMA_STEP_NORM_IN_PROJ", "1") != "0"   # fused path: RMSNorm inside the q/k/v and gate/up projections (A/B switch)
        self.fuse_norm = fuse_norm and not keep_intermediates and not self.fused   # residual add + the RMSNorm that follows it in one kernel
        self.w = weights                                   # dict of DeviceBuffers, see random_weights / upload_weights
        page_elems = c.page * c.hk * c.d
        # kv_fp8: the KV cache holds e4m3fn bytes with one dequantisation scale per kv head (atoma_rope_qk_cache_fp8 /
        # atoma_paged_decode_fp8); everything else of the step is unchanged
        self.kv_fp8 = kv_fp8
        cache_dt = np.uint8 if kv_fp8 else np.uint16
        self.kc = [ah.DeviceBuffer.zeros((num_pages * page_elems,), cache_dt) for _ in range(c.layers)]
        self.vc = [ah.DeviceBuffer.zeros((num_pages * page_elems,), cache_dt) for _ in range(c.layers)]
        self.k_scale = ah.DeviceBuffer.from_numpy(np.full(c.hk, kv_scale, np.float32))
        self.v_scale = ah.DeviceBuffer.from_numpy(np.full(c.hk, kv_scale, np.float32))
        self.max_blocks = max_blocks
        B = batch
        self.ids = ah.DeviceBuffer.zeros((B,), np.int32)
        self.pos = ah.DeviceBuffer.zeros((B,), np.int64)
        self.slots = ah.DeviceBuffer.zeros((B,), np.int64)
        self.lens = ah.DeviceBuffer.zeros((B,), np.int32)
        self.bt = ah.DeviceBuffer.zeros((B, max_blocks), np.int32)
        self.logits = ah.DeviceBuffer(B * c.vocab * 2)
        self.next_ids = ah.DeviceBuffer.zeros((B,), np.int32)
        self.next_val = ah.DeviceBuffer.zeros((B,), np.float32)
        self.trace = []                                    # (op name, layer, dict of buffers) when keep_intermediates
        self._bufs = {}

    def _buf(self, name, layer, nbytes):
        key = (name, layer if self.keep else 0)
        if key not in self._bufs:
            self._bufs[key] = ah.DeviceBuffer(nbytes)
        return self._bufs[key]

    def set_inputs(self, ids, positions, slots, seqlens, block_table):
        self.ids.upload(np.asarray(ids, np.int32))
        self.pos.upload(np.asarray(positions, np.int64))
        self.slots.upload(np.asarray(slots, np.int64))
        self.lens.upload(np.asarray(seqlens, np.int32))
        bt = np.zeros((self.B, self.max_blocks), np.int32)
        block_table = np.asarray(block_table, np.int32)
        bt[:, :block_table.shape[1]] = block_table
        self.bt.upload(bt)

    def bind_metadata(self, ids, positions, slots, seqlens, block_table, max_blocks):
        """Use tensors that live elsewhere on the device (raw pointers), e.g. the packed buffer atoma_prepare_inputs uploads."""
        P = type("P", (), {})
        for name, ptr in (("ids", ids), ("pos", positions), ("slots", slots), ("lens", seqlens), ("bt", block_table)):
            o = P()
            o.ptr = int(ptr)
            setattr(self, name, o)
        self.max_blocks = max_blocks

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {ah.last_error()}")

    def run(self):
        c, B, s, L = self.cfg, self.B, self.stream.s, ah.lib
        H, qkvw, hd = c.hidden, c.qkv, c.h * c.d
        self.trace = []
        x = self._buf("x_emb", 0, B * H * 2)
        self._ok(L.atoma_embedding(self.ids.ptr, 0, self.w["emb"].ptr, x.ptr, B, H, c.vocab, H, BF16, s), "embedding")
        if self.keep:
            self.trace.append(("embedding", 0, dict(out=x)))
        xf = self._buf("xf", 0, B * H * 2)
        for l in range(c.layers):
            xn = self._buf("xn1", l, B * H * 2)
            qkv = self._buf("qkv", l, B * qkvw * 2)
            if self.qkv_fused and not self.kv_fp8:          # 17..64 rows: projection -> RoPE -> cache write behind one entry
                if not (self.fuse_norm and l > 0):         # with fuse_norm the previous layer's last add produced xn already
                    self._ok(L.atoma_rms_norm(x.ptr, self.w["norm1"][l].ptr, xn.ptr, B, H, H, H, c.eps, BF16, s), "rms_norm")
                self._ok(L.atoma_linear_decode_qkv_rope_cache(xn.ptr, self.w["wqkv"][l].ptr, qkv.ptr, self.kc[l].ptr, self.vc[l].ptr, self.slots.ptr,
                                                              self.w["cos"].ptr, self.w["sin"].ptr, self.pos.ptr, B, H, c.h, c.hk, c.d, H, H, qkvw,
                                                              c.page * c.hk * c.d, c.page, BF16, 1, s), "qkv projection + rope + cache write")
            elif self.fused and self.norm_in_proj:          # the input norm inside the projection (batch <= 4: no launch of its own)
                self._ok(L.atoma_linear_decode_rmsnorm(x.ptr, self.w["norm1"][l].ptr, c.eps, self.w["wqkv"][l].ptr, qkv.ptr, xn.ptr, B, H, qkvw, H, H, qkvw,
                                                       BF16, s), "rms_norm + qkv projection")
            else:
                if not (self.fuse_norm and l > 0):         # with fuse_norm the previous layer's last add produced xn already
                    self._ok(L.atoma_rms_norm(x.ptr, self.w["norm1"][l].ptr, xn.ptr, B, H, H, H, c.eps, BF16, s), "rms_norm")
                self._ok(self.linear(xn.ptr, self.w["wqkv"][l].ptr, qkv.ptr, B, H, qkvw, H, H, qkvw, BF16, s), "qkv projection")
            qkv_pre = None
            if self.keep:                                   # RoPE works in place: keep the projection's output for the checker
                qkv_pre = self._buf("qkv_pre", l, B * qkvw * 2)
                ah.hip_check(ah.hip.hipMemcpyAsync(qkv_pre.ptr, qkv.ptr, B * qkvw * 2, 3, s), "copy")
            kptr, vptr = qkv.ptr + hd * 2, qkv.ptr + (hd + c.hk * c.d) * 2
            att = self._buf("att", l, B * hd * 2)
            if self.kv_fp8:
                self._ok(L.atoma_rope_qk_cache_fp8(qkv.ptr, kptr, vptr, self.kc[l].ptr, self.vc[l].ptr, self.slots.ptr, self.k_scale.ptr, self.v_scale.ptr,
                                                   self.w["cos"].ptr, self.w["sin"].ptr, self.pos.ptr, B, c.h, c.hk, c.d, qkvw, qkvw, qkvw,
                                                   c.page * c.hk * c.d, c.page, BF16, 1, s), "rope + fp8 cache write")
                self._ok(L.atoma_paged_decode_fp8(qkv.ptr, self.kc[l].ptr, self.vc[l].ptr, att.ptr, self.k_scale.ptr, self.v_scale.ptr, self.bt.ptr,
                                                  self.lens.ptr, B, c.h, c.hk, c.d, self.max_blocks, c.page, qkvw, c.d, hd, c.d, c.page * c.hk * c.d,
                                                  c.hk * c.d, c.d, c.d ** -0.5, BF16, s), "paged decode over the fp8 cache")
            else:
                if not self.qkv_fused:                      # (fused: rotated and cached by the projection's own merge kernel above)
                    self._ok(L.atoma_rope_qk_cache(qkv.ptr, kptr, vptr, self.kc[l].ptr, self.vc[l].ptr, self.slots.ptr, self.w["cos"].ptr,
                                                   self.w["sin"].ptr, self.pos.ptr, B, c.h, c.hk, c.d, qkvw, qkvw, qkvw, c.page * c.hk * c.d,
                                                   c.page, BF16, 1, s), "rope + cache write")
                ah.run_mha(qkv, self.kc[l], self.vc[l], att, b=B, h=c.h, h_k=c.hk, d=c.d, seqlen_q=1, seqlen_k=self.max_blocks * c.page,
                           softmax_scale=c.d ** -0.5, is_bf16=BF16, q_strides=(qkvw, qkvw, c.d), o_strides=(hd, hd, c.d),
                           k_strides=(c.page * c.hk * c.d, c.hk * c.d, c.d), v_strides=(c.page * c.hk * c.d, c.hk * c.d, c.d),
                           cu_seqlens_k=self.lens.ptr, is_seqlens_k_cumulative=False, block_table=self.bt.ptr, block_table_batch_stride=self.max_blocks,
                           page_block_size=c.page, force_split_kernel=True, unpadded_lse=False, stream=s)
            if self.after_attention is not None:
                self.after_attention(l)
            x1 = self._buf("x1", l, B * H * 2)
            xn2 = self._buf("xn2", l, B * H * 2)
            act = self._buf("act", l, B * c.inter * 2)
            # ping-pong the residual stream so that no kernel reads and writes the same buffer
            x2 = self._buf("x2", l, B * H * 2) if self.keep else self._buf("x2a" if l % 2 == 0 else "x2b", 0, B * H * 2)
            o = gu = dn = None
            if self.fused:
                self._ok(L.atoma_linear_decode_residual(att.ptr, self.w["wo"][l].ptr, x.ptr, x1.ptr, B, hd, H, hd, hd, H, H, BF16, s), "o projection + residual")
                if self.norm_in_proj:
                    self._ok(L.atoma_linear_decode_rmsnorm_silu_mul(x1.ptr, self.w["norm2"][l].ptr, c.eps, self.w["wgu"][l].ptr, act.ptr, xn2.ptr, B, H, c.inter,
                                                                    H, H, c.inter, BF16, s), "rms_norm + gate/up projection + silu * up")
                else:
                    self._ok(L.atoma_rms_norm(x1.ptr, self.w["norm2"][l].ptr, xn2.ptr, B, H, H, H, c.eps, BF16, s), "rms_norm")
                    self._ok(L.atoma_linear_decode_silu_mul(xn2.ptr, self.w["wgu"][l].ptr, act.ptr, B, H, c.inter, H, H, c.inter, BF16, s), "gate/up projection + silu * up")
                self._ok(L.atoma_linear_decode_residual(act.ptr, self.w["wdown"][l].ptr, x1.ptr, x2.ptr, B, c.inter, H, c.inter, c.inter, H, H, BF16, s), "down projection + residual")
            else:
                o = self._buf("o", l, B * H * 2)
                self._ok(self.linear(att.ptr, self.w["wo"][l].ptr, o.ptr, B, hd, H, hd, hd, H, BF16, s), "o projection")
                # all-reduce + residual add + RMSNorm in the all-reduce's own launch where the engine offers it, else the two calls (same bits)
                one = bool(self.fuse_norm and self.allreduce_norm and self.allreduce_norm(o.ptr, x.ptr, self.w["norm2"][l].ptr, x1.ptr, xn2.ptr, B))
                if not one and self.allreduce:
                    self.allreduce(o.ptr, B * H)
                if one:
                    pass
                elif self.fuse_norm:
                    self._ok(L.atoma_add_rms_norm(x.ptr, o.ptr, self.w["norm2"][l].ptr, x1.ptr, xn2.ptr, B, H, H, H, H, H, c.eps, BF16, s), "residual add + rms_norm")
                else:
                    self._ok(L.atoma_add(x.ptr, o.ptr, x1.ptr, B * H, BF16, s), "residual add")
                    self._ok(L.atoma_rms_norm(x1.ptr, self.w["norm2"][l].ptr, xn2.ptr, B, H, H, H, c.eps, BF16, s), "rms_norm")
                if self.tp_own:
                    self._ok(L.atoma_linear_decode_silu_mul(xn2.ptr, self.w["wgu"][l].ptr, act.ptr, B, H, c.inter, H, H, c.inter, BF16, s), "gate/up projection + silu * up")
                else:
                    gu = self._buf("gu", l, B * 2 * c.inter * 2)
                    self._ok(self.linear(xn2.ptr, self.w["wgu"][l].ptr, gu.ptr, B, H, 2 * c.inter, H, H, 2 * c.inter, BF16, s), "gate/up projection")
                    self._ok(L.atoma_silu_mul(gu.ptr, gu.ptr + c.inter * 2, act.ptr, B, c.inter, 2 * c.inter, 2 * c.inter, c.inter, BF16, s), "silu * up")
                dn = self._buf("dn", l, B * H * 2)
                self._ok(self.linear(act.ptr, self.w["wdown"][l].ptr, dn.ptr, B, c.inter, H, c.inter, c.inter, H, BF16, s), "down projection")
                last = l == c.layers - 1
                nw, nout = (self.w["norm_f"], xf) if last else (self.w["norm1"][l + 1], self._buf("xn1", l + 1, B * H * 2))
                one = bool(self.fuse_norm and self.allreduce_norm and self.allreduce_norm(dn.ptr, x1.ptr, nw.ptr, x2.ptr, nout.ptr, B))
                if not one and self.allreduce:
                    self.allreduce(dn.ptr, B * H)
                if one:
                    pass
                elif self.fuse_norm:                     # ... + the next layer's input norm (or the final norm)
                    self._ok(L.atoma_add_rms_norm(x1.ptr, dn.ptr, nw.ptr, x2.ptr, nout.ptr, B, H, H, H, H, H, c.eps, BF16, s), "residual add + rms_norm")
                else:
                    self._ok(L.atoma_add(x1.ptr, dn.ptr, x2.ptr, B * H, BF16, s), "residual add")
            if self.keep:
                self.trace.append(("layer", l, dict(x=x, xn1=xn, qkv_pre=qkv_pre, qkv=qkv, att=att, o=o, x1=x1, xn2=xn2, gu=gu, act=act, dn=dn, x2=x2)))
            x = x2
        if not self.fuse_norm:
            self._ok(L.atoma_rms_norm(x.ptr, self.w["norm_f"].ptr, xf.ptr, B, H, H, H, c.eps, BF16, s), "rms_norm")
        self._ok(self.linear(xf.ptr, self.w["lm_head"].ptr, self.logits.ptr, B, H, c.vocab, H, H, c.vocab, BF16, s), "lm_head")
        self._ok(L.atoma_argmax_rows(self.logits.ptr, B, c.vocab, c.vocab, BF16, self.next_ids.ptr, self.next_val.ptr, s), "argmax")
        if self.keep:
            self.trace.append(("head", 0, dict(x=x, xf=xf, logits=self.logits)))


class PrefillStep:
    """One whole prompt of T tokens of ONE sequence through the model (the reference prefills with flash_attn_varlen over
    cu_seqlens, flash_attention.rs:369-409): embedding -> per layer [RMSNorm -> q/k/v GEMM -> RoPE + cache write -> causal
    prefill attention over the prompt's own K/V -> o GEMM -> residual -> RMSNorm -> gate/up GEMM -> SiLU.up -> down GEMM ->
    residual] -> RMSNorm of the last token -> lm_head -> argmax.  Shares the KV caches of a DecodeStep."""

    def __init__(self, cfg, T, decode_step, stream, prompts=1, allreduce=None):
        """T tokens in all = `prompts` prompts of T / prompts tokens each, back to back (varlen batch).
        allreduce(ptr, count): in-place sum of a [T, hidden] bf16 tensor over the tensor-parallel ranks, enqueued on `stream` -- called
        after the o and the down projection when `cfg` and the weights are ONE rank's shard (llama_nccl.rs:139,195: the same two
        TensorParallelRowLinear outputs as in the decode step; at T = 4096 and hidden 8192 that is a 64 MiB message)."""
        c = self.cfg = cfg
        assert T % prompts == 0
        self.allreduce = allreduce
        self.T, self.n, self.stream, self.w, self.kc, self.vc = T, prompts, stream, decode_step.w, decode_step.kc, decode_step.vc
        self.ids = ah.DeviceBuffer.zeros((T,), np.int32)
        self.pos = ah.DeviceBuffer.from_numpy(np.tile(np.arange(T // prompts, dtype=np.int64), prompts))
        self.slots = ah.DeviceBuffer.zeros((T,), np.int64)
        self.cu = ah.DeviceBuffer.from_numpy((np.arange(prompts + 1) * (T // prompts)).astype(np.int32))
        H = c.hidden
        buf = lambda n: ah.DeviceBuffer(T * n * 2)
        self.x, self.x1, self.x2, self.xn, self.qkv, self.att = buf(H), buf(H), buf(H), buf(H), buf(c.qkv), buf(c.h * c.d)
        self.o, self.gu, self.act = buf(H), buf(2 * c.inter), buf(c.inter)
        self.xf = ah.DeviceBuffer(prompts * H * 2)
        self.logits = ah.DeviceBuffer(prompts * c.vocab * 2)
        self.next_id = ah.DeviceBuffer.zeros((prompts,), np.int32)
        self.next_val = ah.DeviceBuffer.zeros((prompts,), np.float32)

    def set_inputs(self, ids, slots):
        self.ids.upload(np.asarray(ids, np.int32))
        self.slots.upload(np.asarray(slots, np.int64))

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {ah.last_error()}")

    def run(self):
        c, T, s, L = self.cfg, self.T, self.stream.s, ah.lib
        H, qkvw, hd = c.hidden, c.qkv, c.h * c.d
        x, x1, x2 = self.x, self.x1, self.x2
        self._ok(L.atoma_embedding(self.ids.ptr, 0, self.w["emb"].ptr, x.ptr, T, H, c.vocab, H, BF16, s), "embedding")
        for l in range(c.layers):
            if l == 0:                                      # later layers: the previous layer's last add produced xn already
                self._ok(L.atoma_rms_norm(x.ptr, self.w["norm1"][l].ptr, self.xn.ptr, T, H, H, H, c.eps, BF16, s), "rms_norm")
            self._ok(L.atoma_linear(self.xn.ptr, self.w["wqkv"][l].ptr, self.qkv.ptr, T, H, qkvw, H, H, qkvw, BF16, s), "qkv projection")
            kptr, vptr = self.qkv.ptr + hd * 2, self.qkv.ptr + (hd + c.hk * c.d) * 2
            self._ok(L.atoma_rope_qk_cache(self.qkv.ptr, kptr, vptr, self.kc[l].ptr, self.vc[l].ptr, self.slots.ptr, self.w["cos"].ptr,
                                           self.w["sin"].ptr, self.pos.ptr, T, c.h, c.hk, c.d, qkvw, qkvw, qkvw, c.page * c.hk * c.d,
                                           c.page, BF16, 1, s), "rope + cache write")
            ah.run_mha(self.qkv.ptr, kptr, vptr, self.att, b=self.n, h=c.h, h_k=c.hk, d=c.d, seqlen_q=T // self.n, seqlen_k=T // self.n, softmax_scale=c.d ** -0.5,
                       is_bf16=BF16, q_strides=(0, qkvw, c.d), k_strides=(0, qkvw, c.d), v_strides=(0, qkvw, c.d), o_strides=(0, hd, c.d),
                       is_causal=1, cu_seqlens_q=self.cu, cu_seqlens_k=self.cu, stream=s)
            self._ok(L.atoma_linear(self.att.ptr, self.w["wo"][l].ptr, self.o.ptr, T, hd, H, hd, hd, H, BF16, s), "o projection")
            if self.allreduce:
                self.allreduce(self.o.ptr, T * H)
            self._ok(L.atoma_add_rms_norm(x.ptr, self.o.ptr, self.w["norm2"][l].ptr, x1.ptr, self.xn.ptr, T, H, H, H, H, H, c.eps, BF16, s), "residual add + rms_norm")
            self._ok(L.atoma_linear(self.xn.ptr, self.w["wgu"][l].ptr, self.gu.ptr, T, H, 2 * c.inter, H, H, 2 * c.inter, BF16, s), "gate/up projection")
            self._ok(L.atoma_silu_mul(self.gu.ptr, self.gu.ptr + c.inter * 2, self.act.ptr, T, c.inter, 2 * c.inter, 2 * c.inter, c.inter, BF16, s), "silu * up")
            self._ok(L.atoma_linear(self.act.ptr, self.w["wdown"][l].ptr, self.o.ptr, T, c.inter, H, c.inter, c.inter, H, BF16, s), "down projection")
            if self.allreduce:
                self.allreduce(self.o.ptr, T * H)
            if l + 1 < c.layers:
                self._ok(L.atoma_add_rms_norm(x1.ptr, self.o.ptr, self.w["norm1"][l + 1].ptr, x2.ptr, self.xn.ptr, T, H, H, H, H, H, c.eps, BF16, s), "residual add + rms_norm")
            else:                                           # the final norm only touches the last token of every prompt
                self._ok(L.atoma_add(x1.ptr, self.o.ptr, x2.ptr, T * H, BF16, s), "residual add")
            x, x2 = x2, x
        Ts, n = T // self.n, self.n
        last = x.ptr + (Ts - 1) * H * 2                      # the last token of every prompt: rows Ts - 1, 2 Ts - 1, ...
        self._ok(L.atoma_rms_norm(last, self.w["norm_f"].ptr, self.xf.ptr, n, H, Ts * H, H, c.eps, BF16, s), "rms_norm")
        self._ok(L.atoma_linear(self.xf.ptr, self.w["lm_head"].ptr, self.logits.ptr, n, H, c.vocab, H, H, c.vocab, BF16, s), "lm_head")
        self._ok(L.atoma_argmax_rows(self.logits.ptr, n, c.vocab, c.vocab, BF16, self.next_id.ptr, self.next_val.ptr, s), "argmax")


def rope_tables(cfg):
    """cos / sin [max_pos, d/2] from the library's own table builder (atoma_rope_table = `Cache::new`, llama.rs:154-200)."""
    cos = np.zeros((cfg.max_pos, cfg.d // 2), np.uint16)
    sin = np.zeros_like(cos)
    rc = ah.lib.atoma_rope_table(cos.ctypes.data, sin.ctypes.data, cfg.max_pos, cfg.d, float(cfg.theta), 0.0, 0.0, 0.0, 0, BF16)
    if rc != 0:
        raise RuntimeError(ah.last_error())
    return cos, sin


def random_host_weights(rng, cfg):
    """Synthetic bf16 weights of a whole model on the host (numpy uint16), scaled so that activations stay O(1):
    the dict layout `upload_weights` and `tp.shard_weights` take."""
    from halfs import from_f32
    H, I = cfg.hidden, cfg.inter
    r = lambda shape, scale: from_f32((rng.standard_normal(shape) * scale).astype(np.float32), BF16)
    near1 = lambda: from_f32((1 + 0.1 * rng.standard_normal(H)).astype(np.float32), BF16)
    return dict(emb=r((cfg.vocab, H), 1.0), norm1=[near1() for _ in range(cfg.layers)],
                wqkv=[r((cfg.qkv, H), H ** -0.5) for _ in range(cfg.layers)],
                wo=[r((H, cfg.h * cfg.d), (cfg.h * cfg.d) ** -0.5) for _ in range(cfg.layers)],
                norm2=[near1() for _ in range(cfg.layers)], wgu=[r((2 * I, H), H ** -0.5) for _ in range(cfg.layers)],
                wdown=[r((H, I), I ** -0.5) for _ in range(cfg.layers)], norm_f=near1(), lm_head=r((cfg.vocab, H), H ** -0.5))


def upload_weights(cfg, host):
    """host: dict of numpy uint16 arrays (emb, norm1[l], wqkv[l], wo[l], norm2[l], wgu[l], wdown[l], norm_f, lm_head)."""
    up = ah.DeviceBuffer.from_numpy
    w = {k: up(v) for k, v in host.items() if not isinstance(v, list)}
    for k, v in host.items():
        if isinstance(v, list):
            w[k] = [up(a) for a in v]
    cos, sin = rope_tables(cfg)
    w["cos"], w["sin"] = up(cos), up(sin)
    return w
