"""A whole Llama decode step on the device (tools/decode_step.py: embedding, RMSNorm, projections, RoPE + cache write,
paged decode attention, residuals, SiLU.up, lm_head, argmax), every op replayed on the CPU oracle with the inputs the
device op saw -- models/src/llama.rs:392-478 for decode tokens, at a small model size."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from oracle import attn_oracle as A
from oracle import cache_oracle as CO
from oracle import elementwise_oracle as EO
from oracle import linear_oracle as LO
from oracle import norm_rope_oracle as NR
from oracle.halfs import BF16, to_f32
from util import rand_half, ATOL_FEW_KEYS

pytestmark = pytest.mark.gpu


def ulps(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max())


def close_linear(got, ref, what):
    g, r = to_f32(got, BF16), to_f32(ref, BF16)
    assert (np.abs(g - r) <= 2.0 ** -7 * np.abs(r) + 3e-5).all(), what


@pytest.mark.parametrize("family", ["llama", "llama-1b-heads", "phi3-heads"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_llama_decode_step_op_by_op(gpu, graph, family):
    """One decode step, every intermediate against the oracle.  The layer is the one the reference's three model files share (llama.rs, mistral.rs:
    head size 128; phi3.rs: the same block with fused qkv / gate_up weights -- the layout this step uses anyway -- and 32 MHA heads of size 96)."""
    import decode_step as DS
    rng = np.random.default_rng(11)
    heads, kv_heads, head_dim = {"llama": (4, 2, 128), "llama-1b-heads": (8, 2, 64), "phi3-heads": (4, 4, 96)}[family]
    cfg = DS.Config(layers=2, hidden=512, heads=heads, kv_heads=kv_heads, head_dim=head_dim, intermediate=1024, vocab=1008, page=16, max_pos=256)
    B = 5
    ctx = np.array([0, 17, 40, 64, 100])                    # tokens already in the cache; the step adds one per sequence
    lens = (ctx + 1).astype(np.int32)
    blocks = [(int(L) + cfg.page - 1) // cfg.page for L in lens]
    num_pages = sum(blocks) + 3
    perm = rng.permutation(num_pages)
    bt = np.zeros((B, max(blocks)), np.int32)
    pos0 = 0
    for i, n in enumerate(blocks):
        bt[i, :n] = perm[pos0:pos0 + n]
        pos0 += n
    H, I = cfg.hidden, cfg.inter
    host = dict(
        emb=rand_half(rng, (cfg.vocab, H), BF16),
        norm1=[rand_half(rng, (H,), BF16, 0.1) + np.uint16(0) for _ in range(cfg.layers)],
        wqkv=[rand_half(rng, (cfg.qkv, H), BF16, H ** -0.5) for _ in range(cfg.layers)],
        wo=[rand_half(rng, (H, cfg.h * cfg.d), BF16, (cfg.h * cfg.d) ** -0.5) for _ in range(cfg.layers)],
        norm2=[rand_half(rng, (H,), BF16, 0.1) for _ in range(cfg.layers)],
        wgu=[rand_half(rng, (2 * I, H), BF16, H ** -0.5) for _ in range(cfg.layers)],
        wdown=[rand_half(rng, (H, I), BF16, I ** -0.5) for _ in range(cfg.layers)],
        norm_f=rand_half(rng, (H,), BF16, 0.1),
        lm_head=rand_half(rng, (cfg.vocab, H), BF16, H ** -0.5))
    for key in ("norm1", "norm2"):                          # norm weights around 1
        host[key] = [NR.from_f32(1 + to_f32(a, BF16), BF16) for a in host[key]]
    host["norm_f"] = NR.from_f32(1 + to_f32(host["norm_f"], BF16), BF16)
    st = gpu.Stream()
    step = DS.DecodeStep(cfg, B, num_pages, bt.shape[1], DS.upload_weights(cfg, host), st, keep_intermediates=True)
    kc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    vc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    for l in range(cfg.layers):
        step.kc[l].upload(kc0[l])
        step.vc[l].upload(vc0[l])
    ids = rng.integers(0, cfg.vocab, B)
    slots = np.array([CO.slot_mapping_for(bt[i], int(ctx[i]), int(ctx[i]) + 1, cfg.page)[0] for i in range(B)], np.int64)
    step.set_inputs(ids, ctx, slots, lens, bt)
    if graph:
        step.run()                                          # eager once: split-KV workspace of this stream
        st.synchronize()
        for l in range(cfg.layers):                         # restore the caches the first run wrote into
            step.kc[l].upload(kc0[l])
            step.vc[l].upload(vc0[l])
        with gpu.Graph.capture(st) as g:
            step.run()
        g.launch()
    else:
        step.run()
    st.synchronize()

    cos, sin = DS.rope_tables(cfg)
    dl = lambda buf, shape: buf.numpy(np.uint16, shape)
    hd, qw = cfg.h * cfg.d, cfg.qkv
    name, _, t = step.trace[0]
    x_prev = dl(t["out"], (B, H))
    assert np.array_equal(x_prev, EO.embedding(ids, host["emb"]))
    for (name, l, t) in step.trace[1:-1]:
        x = dl(t["x"], (B, H))
        assert np.array_equal(x, x_prev)
        xn1 = dl(t["xn1"], (B, H))
        assert ulps(xn1, NR.rms_norm(x, host["norm1"][l], cfg.eps, BF16)) <= 1
        qkv_pre = dl(t["qkv_pre"], (B, qw))
        close_linear(qkv_pre, LO.linear(xn1, host["wqkv"][l], BF16), f"layer {l} qkv projection")
        qkv = dl(t["qkv"], (B, qw))
        q_pre, k_pre = qkv_pre[:, :hd].reshape(B, cfg.h, cfg.d), qkv_pre[:, hd:hd + cfg.hk * cfg.d].reshape(B, cfg.hk, cfg.d)
        v = np.ascontiguousarray(qkv_pre[:, hd + cfg.hk * cfg.d:]).reshape(B, cfg.hk, cfg.d)
        q_rot, k_rot = NR.rope(q_pre, cos, sin, ctx, BF16), NR.rope(k_pre, cos, sin, ctx, BF16)
        assert np.array_equal(qkv[:, :hd].reshape(B, cfg.h, cfg.d), q_rot) and np.array_equal(qkv[:, hd:hd + cfg.hk * cfg.d].reshape(B, cfg.hk, cfg.d), k_rot)
        assert np.array_equal(qkv[:, hd + cfg.hk * cfg.d:], qkv_pre[:, hd + cfg.hk * cfg.d:])          # v untouched
        kc, vc = kc0[l].copy(), vc0[l].copy()
        CO.reshape_and_cache_flash(k_rot, v, kc, vc, slots)
        shape = (num_pages, cfg.page, cfg.hk, cfg.d)
        assert np.array_equal(dl(step.kc[l], shape), kc) and np.array_equal(dl(step.vc[l], shape), vc)
        att = dl(t["att"], (B, hd))
        ref = A.flash_attn_kv_cache(q_rot[:, None], kc, vc, cfg.d ** -0.5, BF16, bt, lens)[:, 0].reshape(B, hd)
        a32, r32 = to_f32(att, BF16), to_f32(ref, BF16)
        assert (np.abs(a32 - r32) <= ATOL_FEW_KEYS[BF16] + 2.0 ** -7 * np.abs(r32)).all(), f"layer {l} attention"
        o = dl(t["o"], (B, H))
        close_linear(o, LO.linear(att, host["wo"][l], BF16), f"layer {l} o projection")
        x1 = dl(t["x1"], (B, H))
        assert np.array_equal(x1, EO.add(x, o, BF16))
        xn2 = dl(t["xn2"], (B, H))
        assert ulps(xn2, NR.rms_norm(x1, host["norm2"][l], cfg.eps, BF16)) <= 1
        gu = dl(t["gu"], (B, 2 * I))
        close_linear(gu, LO.linear(xn2, host["wgu"][l], BF16), f"layer {l} gate/up projection")
        act = dl(t["act"], (B, I))
        assert ulps(act, EO.silu_mul(np.ascontiguousarray(gu[:, :I]), np.ascontiguousarray(gu[:, I:]), BF16)) <= 1
        dn = dl(t["dn"], (B, H))
        close_linear(dn, LO.linear(act, host["wdown"][l], BF16), f"layer {l} down projection")
        x2 = dl(t["x2"], (B, H))
        assert np.array_equal(x2, EO.add(x1, dn, BF16))
        x_prev = x2
    _, _, t = step.trace[-1]
    xf = dl(t["xf"], (B, H))
    assert ulps(xf, NR.rms_norm(x_prev, host["norm_f"], cfg.eps, BF16)) <= 1
    logits = dl(t["logits"], (B, cfg.vocab))
    close_linear(logits, LO.linear(xf, host["lm_head"], BF16), "lm_head")
    assert np.array_equal(step.next_ids.numpy(np.int32, (B,)), to_f32(logits, BF16).argmax(1))


@pytest.mark.parametrize("B", [3, 2, 24, 64, 100, 256])
def test_llama_decode_step_fused_epilogues_are_bit_identical(gpu, B):
    """Residual adds and SiLU.up folded into the projections' split merge (and, at 1-2 rows, the RMSNorms folded into the q/k/v and
    gate/up projections) keep the reference's rounding points, so the fused step must reproduce the op-by-op step bit for bit
    (logits, next tokens, caches) -- at 2-3 rows on the weight-streaming kernel, at 24 / 64 rows on the 17..64-row kernel, at 100 / 256 rows
    on round 6's linear_wide_kernel (q/k/v + RoPE + cache write behind one entry; the lm_head of 1008 rows in slices of 64 batch rows); the
    op-by-op step runs on the library's own projection kernels too (own_projections), the vendor GEMM rounds differently."""
    import decode_step as DS
    rng = np.random.default_rng(12)
    cfg = DS.Config(layers=3, hidden=512, heads=4, kv_heads=2, head_dim=128, intermediate=1024, vocab=1008, page=16, max_pos=256)
    ctx = np.array([5, 33, 90])[:B] if B <= 3 else rng.integers(1, 120, B)
    lens = (ctx + 1).astype(np.int32)
    blocks = [(int(L) + cfg.page - 1) // cfg.page for L in lens]
    num_pages = sum(blocks) + 1
    bt = np.zeros((B, max(blocks)), np.int32)
    perm, p0 = rng.permutation(num_pages), 0
    for i, n in enumerate(blocks):
        bt[i, :n] = perm[p0:p0 + n]
        p0 += n
    H, I = cfg.hidden, cfg.inter
    one = lambda: NR.from_f32(1 + 0.1 * rng.standard_normal(H).astype(np.float32), BF16)
    host = dict(emb=rand_half(rng, (cfg.vocab, H), BF16), norm1=[one() for _ in range(cfg.layers)], norm2=[one() for _ in range(cfg.layers)],
                wqkv=[rand_half(rng, (cfg.qkv, H), BF16, H ** -0.5) for _ in range(cfg.layers)],
                wo=[rand_half(rng, (H, cfg.h * cfg.d), BF16, (cfg.h * cfg.d) ** -0.5) for _ in range(cfg.layers)],
                wgu=[rand_half(rng, (2 * I, H), BF16, H ** -0.5) for _ in range(cfg.layers)],
                wdown=[rand_half(rng, (H, I), BF16, I ** -0.5) for _ in range(cfg.layers)], norm_f=one(),
                lm_head=rand_half(rng, (cfg.vocab, H), BF16, H ** -0.5))
    w = DS.upload_weights(cfg, host)
    st = gpu.Stream()
    kc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    vc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    ids = rng.integers(0, cfg.vocab, B)
    slots = np.array([CO.slot_mapping_for(bt[i], int(ctx[i]), int(ctx[i]) + 1, cfg.page)[0] for i in range(B)], np.int64)
    res = []
    for fused, fuse_norm in ((False, False), (True, False), (False, True)):   # op by op / fused projection epilogues / fused add + RMSNorm
        step = DS.DecodeStep(cfg, B, num_pages, bt.shape[1], w, st, fused_epilogues=fused, fuse_norm=fuse_norm, own_projections=True)
        assert step.fused == fused
        for l in range(cfg.layers):
            step.kc[l].upload(kc0[l])
            step.vc[l].upload(vc0[l])
        step.set_inputs(ids, ctx, slots, lens, bt)
        step.run()
        st.synchronize()
        res.append((step.logits.numpy(np.uint16, (B, cfg.vocab)), step.next_ids.numpy(np.int32, (B,)),
                    [step.kc[l].numpy(np.uint16) for l in range(cfg.layers)]))
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1])
        assert all(np.array_equal(a, b) for a, b in zip(res[0][2], other[2]))


def test_prefill_step_matches_token_by_token_decode(gpu):
    """PrefillStep (GEMM projections + causal prefill attention over the prompt) leaves the same KV cache and predicts the same
    next token as feeding the prompt one token at a time through DecodeStep (streaming projections + paged decode attention):
    two routes through different kernels, equal up to the rounding of their accumulation orders."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import decode_step as DS
    from oracle.halfs import from_f32, to_f32, BF16 as BF
    rng = np.random.default_rng(77)
    c = DS.Config(2, 256, 4, 2, 64, 512, 1024, page=16, max_pos=256)
    bf = lambda shape, scale: from_f32((rng.standard_normal(shape) * scale).astype(np.float32), BF)
    host = dict(emb=bf((c.vocab, c.hidden), 1.0), lm_head=bf((c.vocab, c.hidden), c.hidden ** -0.5), norm_f=bf((c.hidden,), 0.1),
                norm1=[from_f32(np.ones(c.hidden, np.float32), BF) for _ in range(c.layers)], norm2=[from_f32(np.ones(c.hidden, np.float32), BF) for _ in range(c.layers)],
                wqkv=[bf((c.qkv, c.hidden), c.hidden ** -0.5) for _ in range(c.layers)], wo=[bf((c.hidden, c.h * c.d), (c.h * c.d) ** -0.5) for _ in range(c.layers)],
                wgu=[bf((2 * c.inter, c.hidden), c.hidden ** -0.5) for _ in range(c.layers)], wdown=[bf((c.hidden, c.inter), c.inter ** -0.5) for _ in range(c.layers)])
    host["norm_f"] = from_f32(np.ones(c.hidden, np.float32), BF)
    T, pages = 40, 4
    ids = rng.integers(0, c.vocab, T)
    table = np.array([[2, 0, 3, 1]], np.int32)
    slots = table[0, np.arange(T) // c.page].astype(np.int64) * c.page + np.arange(T) % c.page

    def fresh():
        st = gpu.Stream()
        return st, DS.DecodeStep(c, 1, pages, pages, DS.upload_weights(c, host), st)
    st, step = fresh()
    pre = DS.PrefillStep(c, T, step, st)
    pre.set_inputs(ids, slots)
    pre.run()
    st.synchronize()
    tok_prefill = int(pre.next_id.numpy(np.int32, (1,))[0])
    logits_prefill = to_f32(pre.logits.numpy(np.uint16, (c.vocab,)), BF)
    kc_prefill = [to_f32(b.numpy(np.uint16, (pages * c.page, c.hk, c.d)), BF) for b in step.kc]
    vc_prefill = [to_f32(b.numpy(np.uint16, (pages * c.page, c.hk, c.d)), BF) for b in step.vc]
    st2, step2 = fresh()
    for t in range(T):
        step2.set_inputs([ids[t]], [t], [slots[t]], [t + 1], table)
        step2.run()
    st2.synchronize()
    logits_decode = to_f32(step2.logits.numpy(np.uint16, (c.vocab,)), BF)
    for l in range(c.layers):
        for a, b_, name in ((kc_prefill[l], step2.kc[l], "K"), (vc_prefill[l], step2.vc[l], "V")):
            b2 = to_f32(b_.numpy(np.uint16, (pages * c.page, c.hk, c.d)), BF)
            assert np.abs(a[slots] - b2[slots]).max() <= 0.06 * max(1.0, np.abs(b2[slots]).max()), f"layer {l} {name} cache"
            assert (np.abs(a[slots] - b2[slots]) <= 2.0 ** -6 * np.abs(b2[slots]) + 1e-2).mean() > 0.99, f"layer {l} {name} cache"
    assert np.abs(logits_prefill - logits_decode).max() < 0.1
    assert tok_prefill == int(step2.next_ids.numpy(np.int32, (1,))[0]) or np.sort(logits_decode)[-1] - np.sort(logits_decode)[-2] < 0.1


def test_decode_step_with_fp8_kv_cache_tracks_the_bf16_step(gpu):
    """DecodeStep(kv_fp8=True): RoPE + quantised cache write + decode over the fp8 cache inside a whole step.  With caches that
    hold the SAME values (the bf16 cache = the dequantised fp8 cache) the two steps differ only by the quantisation of the one
    new K/V row per sequence: logits agree closely and the newly written cache rows are the oracle's bytes."""
    import decode_step as DS
    from oracle import fp8_oracle as F8
    rng = np.random.default_rng(31)
    cfg = DS.Config(layers=2, hidden=512, heads=4, kv_heads=2, head_dim=128, intermediate=1024, vocab=1008, page=16, max_pos=256)
    B = 4
    ctx = np.array([3, 17, 40, 100])
    lens = (ctx + 1).astype(np.int32)
    blocks = [(int(L) + cfg.page - 1) // cfg.page for L in lens]
    num_pages = sum(blocks) + 2
    perm = rng.permutation(num_pages)
    bt = np.zeros((B, max(blocks)), np.int32)
    p0 = 0
    for i, n in enumerate(blocks):
        bt[i, :n] = perm[p0:p0 + n]
        p0 += n
    host = DS.random_host_weights(rng, cfg)
    w = DS.upload_weights(cfg, host)
    scale = 0.02
    shape = (num_pages, cfg.page, cfg.hk, cfg.d)
    k8 = [rng.integers(0, 0x70, shape, dtype=np.uint8) | (rng.integers(0, 2, shape, dtype=np.uint8) << 7) for _ in range(cfg.layers)]
    v8 = [rng.integers(0, 0x70, shape, dtype=np.uint8) | (rng.integers(0, 2, shape, dtype=np.uint8) << 7) for _ in range(cfg.layers)]
    from oracle.halfs import from_f32
    ids = rng.integers(0, cfg.vocab, B)
    slots = np.array([CO.slot_mapping_for(bt[i], int(ctx[i]), int(ctx[i]) + 1, cfg.page)[0] for i in range(B)], np.int64)
    st = gpu.Stream()
    s8 = DS.DecodeStep(cfg, B, num_pages, bt.shape[1], w, st, kv_fp8=True, kv_scale=scale, keep_intermediates=True)
    s16 = DS.DecodeStep(cfg, B, num_pages, bt.shape[1], w, st)
    for l in range(cfg.layers):
        s8.kc[l].upload(k8[l]); s8.vc[l].upload(v8[l])
        s16.kc[l].upload(from_f32(F8.decode(k8[l]) * np.float32(scale), BF16))       # exactly representable: 4 significant bits x a scale
        s16.vc[l].upload(from_f32(F8.decode(v8[l]) * np.float32(scale), BF16))
    for s in (s8, s16):
        s.set_inputs(ids, ctx, slots, lens, bt)
        s.run()
    st.synchronize()
    l8 = to_f32(s8.logits.numpy(np.uint16, (B, cfg.vocab)), BF16)
    l16 = to_f32(s16.logits.numpy(np.uint16, (B, cfg.vocab)), BF16)
    assert np.isfinite(l8).all() and np.abs(l8 - l16).max() < 0.25 and np.abs(l8 - l16).mean() < 0.02
    # the rows this step wrote: layer 0's rotated k / v (kept by the trace) quantised by the oracle
    _, _, t = s8.trace[1]
    qkv = t["qkv"].numpy(np.uint16, (B, cfg.qkv))
    hd = cfg.h * cfg.d
    k_rot = np.ascontiguousarray(qkv[:, hd:hd + cfg.hk * cfg.d]).reshape(B, cfg.hk, cfg.d)
    v_new = np.ascontiguousarray(qkv[:, hd + cfg.hk * cfg.d:]).reshape(B, cfg.hk, cfg.d)
    kc, vc = k8[0].copy(), v8[0].copy()
    F8.reshape_and_cache_flash_fp8(k_rot, v_new, kc, vc, slots, np.full(cfg.hk, scale, np.float32), np.full(cfg.hk, scale, np.float32), BF16)
    assert np.array_equal(s8.kc[0].numpy(np.uint8, shape), kc) and np.array_equal(s8.vc[0].numpy(np.uint8, shape), vc)
