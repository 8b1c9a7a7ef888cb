"""One Llama decode step on the CPU oracle, whole or as ONE rank of a tensor-parallel job (test infrastructure).

The op order is models/src/llama.rs:392-410,253-314,364-365 (llama_nccl.rs for the TP variant: the row-parallel o and down
projections are followed by a sum all-reduce, multi_gpu.rs:48-50).  `allreduce(x_bits)` returns the summed tensor; with
None the step is the unsharded model.  Shared by the 2-process gloo test and usable against the device step."""
import numpy as np

from oracle import attn_oracle as A
from oracle import cache_oracle as CO
from oracle import elementwise_oracle as EO
from oracle import linear_oracle as LO
from oracle import norm_rope_oracle as NR
from oracle.halfs import BF16


def decode_step(cfg, w, kc, vc, ids, ctx, slots, lens, bt, cos, sin, allreduce=None):
    """cfg / w: whole model or one rank's shard (tp.shard_config / tp.shard_weights); kc, vc: per-layer paged caches
    [nb, page, hk, d] (modified in place).  Returns (logits bits [B, vocab], trace dict of per-layer tensors)."""
    B, H, hd = len(ids), cfg.hidden, cfg.h * cfg.d
    x = EO.embedding(ids, w["emb"].reshape(cfg.vocab, H))
    trace = []
    for l in range(cfg.layers):
        xn = NR.rms_norm(x, w["norm1"][l], cfg.eps, BF16)
        qkv = LO.linear(xn, w["wqkv"][l].reshape(cfg.qkv, H), BF16)
        q = np.ascontiguousarray(qkv[:, :hd]).reshape(B, cfg.h, cfg.d)
        k = np.ascontiguousarray(qkv[:, hd:hd + cfg.hk * cfg.d]).reshape(B, cfg.hk, cfg.d)
        v = np.ascontiguousarray(qkv[:, hd + cfg.hk * cfg.d:]).reshape(B, cfg.hk, cfg.d)
        q, k = NR.rope(q, cos, sin, ctx, BF16), NR.rope(k, cos, sin, ctx, BF16)
        CO.reshape_and_cache_flash(k, v, kc[l], vc[l], slots)
        att = A.flash_attn_kv_cache(q[:, None], kc[l], vc[l], cfg.d ** -0.5, BF16, bt, lens)[:, 0].reshape(B, hd)
        o = LO.linear(att, w["wo"][l].reshape(H, hd), BF16)
        if allreduce:
            o = allreduce(o)
        x1 = EO.add(x, o, BF16)
        xn2 = NR.rms_norm(x1, w["norm2"][l], cfg.eps, BF16)
        gu = LO.linear(xn2, w["wgu"][l].reshape(2 * cfg.inter, H), BF16)
        act = EO.silu_mul(np.ascontiguousarray(gu[:, :cfg.inter]), np.ascontiguousarray(gu[:, cfg.inter:]), BF16)
        dn = LO.linear(act, w["wdown"][l].reshape(H, cfg.inter), BF16)
        if allreduce:
            dn = allreduce(dn)
        x = EO.add(x1, dn, BF16)
        trace.append(dict(o=o, dn=dn, x=x))
    xf = NR.rms_norm(x, w["norm_f"], cfg.eps, BF16)
    return LO.linear(xf, w["lm_head"].reshape(cfg.vocab, H), BF16), trace
