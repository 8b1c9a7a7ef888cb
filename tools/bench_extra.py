"""The end-to-end numbers bench.py carries beside its headline, each timed with HIP events in the SAME run:

  c3_decode_step   one whole Llama-3.1-8B decode step on the device (BASELINE configs[2] mid-trace: batch 256, contexts
                   U[2048, 2560), block 16, hipGraph replay) -> ms, decode tokens/s/GPU, fraction of the byte roofline;
  prefill          causal varlen FlashAttention-2 prefill, 16 prompts of 2048 tokens, 32 q / 8 kv heads, d = 128
                   -> TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak;
  c4_rank_step     one rank of the 70B TP = 8 decode step of configs[3] without its all-reduces (batch 64, context 4096, 80 layers);
  swap             (+ the pinned hipMemcpyAsync ceiling of the same byte count beside it)
                   CPU<->GPU KV swap of BASELINE configs[4]: 64 tensors (32 layers x K, V), 256 pages of 32 KiB, pinned host
                   memory, both directions -> GB/s over PCIe.
  c2b_mha, c2c_ragged, k4_reshape_and_cache, k5_copy_blocks, n1_norm_rope, p1_prefill_4096
                   the remaining rows of BASELINE.md section 2, each with its algorithmic bytes / flops and fraction of the
                   peak, each checked: bit-exact ops here against their definition, attention / norm / RoPE on sampled outputs by bench.py
                   against the oracle.
Bench plumbing over the C ABI; synthetic data; nothing here imports oracle/."""
import ctypes as C
import os
import sys

import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)
import atoma_hip as ah  # noqa: E402
import decode_step as DS  # noqa: E402
import tp_step as TS  # noqa: E402

HBM_PEAK, MFMA_PEAK_BF16 = 8e12, 2.5e15


WARM_MS = float(os.environ.get("ATOMA_BENCH_WARM_MS", "60"))


def _timed(stream, fn, iters, warm=2):
    """Mean time per call at the device's SUSTAINED clocks: the warm-up lasts at least WARM_MS of device time (after the uploads that precede
    every case the device needs ~30 ms of load to leave its idle power state: tools/probes/warm_probe.py, profiles/r04_clock_ramp_probe.txt)."""
    t0, n = time.perf_counter(), 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < WARM_MS:
        fn()
        stream.synchronize()
        n += 1
    stream.synchronize()
    a, b = ah.Event(), ah.Event()
    a.record(stream.s)
    for _ in range(iters):
        fn()
    b.record(stream.s)
    b.synchronize()
    return a.elapsed_ms(b) / iters


PHI_3_MINI = DS.Config(32, 3072, 32, 32, 96, 8192, 32064)      # models/src/phi3.rs: 32 MHA heads of size 96, fused qkv / gate_up weights


def c3_decode_step(iters=10, batch=256, seed=9, kv_fp8=False, weights=None, cfg=None, name="Llama-3.1-8B decode step (BASELINE configs[2] mid-trace)"):
    rng = np.random.default_rng(seed)
    c = cfg or DS.LLAMA_3_1_8B
    w = weights or TS.random_shard_weights(rng, c)
    st = ah.Stream()
    S = 2560
    pps = S // c.page + 1
    step = DS.DecodeStep(c, batch, batch * pps + 2, pps, w, st, fused_epilogues=True, kv_fp8=kv_fp8)
    bt = rng.permutation(batch * pps).astype(np.int32).reshape(batch, pps)
    ctx = rng.integers(2048, 2560, batch)
    slots = bt[np.arange(batch), ctx // c.page].astype(np.int64) * c.page + ctx % c.page
    step.set_inputs(rng.integers(0, c.vocab, batch), ctx, slots, ctx + 1, bt)
    step.run()
    st.synchronize()
    with ah.Graph.capture(st) as g:
        step.run()
    ms = _timed(st, g.launch, iters)
    weight_bytes = 2 * (c.vocab * c.hidden + c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden)) + 2 * batch * c.hidden
    nbytes = weight_bytes + 2 * int((ctx + 1).sum()) * c.hk * c.d * (1 if kv_fp8 else 2) * c.layers
    out = {"workload": f"{name}, batch {batch}, contexts U[2048,2560), block {c.page}, bf16 weights and activations, "
                       + ("fp8 e4m3fn KV cache" if kv_fp8 else "bf16 KV cache") + ", hipGraph replay",
           "ms_per_step": round(ms, 3), "decode_tokens_per_s_per_gpu": round(batch / (ms * 1e-3), 1), "algorithmic_bytes": int(nbytes),
           "roofline_tokens_per_s": round(batch / (nbytes / HBM_PEAK), 1), "frac_of_hbm_roofline": round(nbytes / HBM_PEAK / (ms * 1e-3), 4)}
    del g, step
    if weights is None:
        del w
    return out


def c5_phi3_mini_decode_step(iters=10):
    """PROBE, not a row of the driver's line (models/src/phi3.rs is outside SURVEY 8's scope).  The same step with the reference's third model family (models/src/phi3.rs): Phi-3-mini shapes -- head size 96 runs on attn_generic.hip's streaming
    decode kernel, not on the tuned head-size-128 kernel; batch 64 (MHA: 384 KiB of KV per token)."""
    return c3_decode_step(iters=iters, batch=64, cfg=PHI_3_MINI, name="Phi-3-mini-shaped decode step (32 layers, hidden 3072, 32 MHA heads of 96)")


def _with_option(name, value, fn):
    """run fn() with a library option set (what an engine would set once at start-up), restored afterwards; the result says so"""
    assert ah.lib.atoma_set_option(name.encode(), value) == 0, ah.last_error()
    try:
        out = fn()
    finally:
        ah.lib.atoma_set_option(name.encode(), 0)
    out["library_option"] = f"{name} = {value} (default 0)"
    return out


def c3_decode_step_paired(iters=10):
    """VARIANT of c3_decode_step, not the default dispatch: `decode_pair = 1` -- what the dispatcher does BY DEFAULT for a batch atoma_prepare_inputs packed (it records that the lengths
    differ; this micro-step sets its metadata directly, so it says it through the option): two sequences per workgroup, the i-th shortest with the i-th longest
    (paged_decode_pair_kernel).  +3..5.6 % on ragged batches that give every resident wavefront one unit, -1.8 % on exactly uniform ones, hence hint-driven."""
    return _with_option("decode_pair", 1, lambda: c3_decode_step(iters=iters, name="Llama-3.1-8B decode step (BASELINE configs[2] mid-trace), decode_pair = 1"))


def c2c_ragged_paired(iters=20):
    """VARIANT of c2c_ragged with `decode_pair = 1` (see c3_decode_step_paired)."""
    return _with_option("decode_pair", 1, lambda: c2c_ragged(iters=iters))


def c3_decode_step_fp8_kv(iters=10):
    """The same step over an fp8 (e4m3fn) KV cache: the attention bytes halve; a VARIANT, not the headline --
    the reference's cache is 16-bit."""
    return c3_decode_step(iters=iters, kv_fp8=True)


def prefill(iters=10, S=2048, nseq=16, h=32, hk=8, d=128, seed=1, with_sample=False):
    rng = np.random.default_rng(seed)
    T = S * nseq
    q, k, v = (TS.rand_dev(rng, T * n * d * 2) for n in (h, hk, hk))
    o = ah.DeviceBuffer(T * h * d * 2)
    cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
    st = ah.Stream()

    def run():
        ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1,
                   q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d),
                   is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu, stream=st.s)
    ms = _timed(st, run, iters)
    flops = 4 * S * S * h * d / 2 * nseq
    # a small host-side sample of what the timed calls wrote, for bench.py's check against the oracle (rows near both ends and at tile / block seams)
    sample = []
    rows = [0, 1, 63, 64, 255, 256, 511, 512, 1023, S - 2, S - 1] if with_sample else []
    qh, kh, vh, oh = (b_.numpy(np.uint16, (T, n, d)) for b_, n in ((q, h), (k, hk), (v, hk), (o, h))) if with_sample else (None,) * 4
    for b_, hq in ((0, 0), (nseq - 1, h - 1)) if with_sample else ():
        s0 = b_ * S
        sample.append({"rows": rows, "scale": d ** -0.5, "q": qh[s0 + np.array(rows), hq].copy(), "o": oh[s0 + np.array(rows), hq].copy(),
                       "k": kh[s0:s0 + S, hq // (h // hk)].copy(), "v": vh[s0:s0 + S, hq // (h // hk)].copy()})
    return {**({"sample": sample} if with_sample else {}),
            "workload": f"FlashAttention-2 prefill, causal varlen, {nseq} x {S} tokens, {h} q / {hk} kv heads, d = {d}, bf16", "ms": round(ms, 4),
            "flops": int(flops), "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1), "frac_mfma": round(flops / (ms * 1e-3) / MFMA_PEAK_BF16, 4),
            "prompt_tokens_per_s_attention_only": round(T / (ms * 1e-3))}


def swap(iters=3, tensors=64, pages=256, nb=512, page_bytes=16 * 8 * 128 * 2, seed=4):
    rng = np.random.default_rng(seed)
    gpu = [ah.DeviceBuffer(nb * page_bytes) for _ in range(tensors)]
    host = [ah.lib.atoma_host_alloc(nb * page_bytes) for _ in range(tensors)]
    gp, hp = (C.c_void_p * tensors)(*[b.ptr for b in gpu]), (C.c_void_p * tensors)(*host)
    m = np.stack([rng.permutation(nb)[:pages], rng.permutation(nb)[:pages]], 1).astype(np.int64)
    st = ah.Stream()
    res = {"workload": f"swap_blocks (BASELINE configs[4]): {tensors} tensors x {pages} pages x {page_bytes // 1024} KiB fp16, pinned host memory, one gather/scatter launch per direction",
           "bytes_one_way": tensors * pages * page_bytes}
    for kind, s, d_, label in ((2, gp, hp, "gpu_to_cpu"), (1, hp, gp, "cpu_to_gpu")):
        def run():
            assert ah.lib.atoma_swap_blocks_multi(s, d_, tensors, m.ctypes.data, pages, page_bytes, kind, st.s) == 0, ah.last_error()
        ms = _timed(st, run, iters, warm=1)
        res[label + "_GBps"] = round(tensors * pages * page_bytes / (ms * 1e-3) / 1e9, 1)
    # The ceiling beside it: ONE contiguous pinned hipMemcpyAsync of the same byte count each way, timed the same way
    total = tensors * pages * page_bytes
    big_h, big_d = ah.lib.atoma_host_alloc(total), ah.DeviceBuffer(total)
    for kind, label in ((2, "pinned_memcpy_d2h_GBps"), (1, "pinned_memcpy_h2d_GBps")):
        def copy():
            src, dst = (big_d.ptr, big_h) if kind == 2 else (big_h, big_d.ptr)
            ah.hip_check(ah.hip.hipMemcpyAsync(dst, src, total, kind, st.s), "hipMemcpyAsync")
        ms = _timed(st, copy, iters, warm=1)
        res[label] = round(total / (ms * 1e-3) / 1e9, 1)
    res["gpu_to_cpu_frac_of_memcpy"] = round(res["gpu_to_cpu_GBps"] / res["pinned_memcpy_d2h_GBps"], 3)
    res["cpu_to_gpu_frac_of_memcpy"] = round(res["cpu_to_gpu_GBps"] / res["pinned_memcpy_h2d_GBps"], 3)
    ah.lib.atoma_host_free(big_h)
    for p in host:
        ah.lib.atoma_host_free(p)
    return res


def _decode_case(name, B, S, h, hk, ragged, iters, seed, sample_seqs, identity=False):
    """A paged-decode workload of BASELINE.md section 2 timed like the headline (one run_mha call per step) + host copies of `sample_seqs`
    sequences (q, the output the TIMED calls wrote, their K / V gathered through the block table) for bench.py's check against the oracle."""
    rng = np.random.default_rng(seed)
    d, page = 128, 16
    pps = S // page
    n_pages = int(B * pps * 1.125)
    bt = (np.arange(B * pps) if identity else rng.permutation(n_pages)[: B * pps]).astype(np.int32).reshape(B, pps)
    lens = (rng.integers(S // 2, S + 1, B) if ragged else np.full(B, S)).astype(np.int32)
    kc, vc = TS.rand_dev(rng, n_pages * page * hk * d * 2), TS.rand_dev(rng, n_pages * page * hk * d * 2)
    q = TS.rand_dev(rng, B * h * d * 2)
    o = ah.DeviceBuffer(B * h * d * 2)
    dbt, dl = ah.DeviceBuffer.from_numpy(bt), ah.DeviceBuffer.from_numpy(lens)
    st = ah.Stream()

    def run():
        ah.run_mha(q, kc, vc, o, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=1,
                   q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
                   cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt, block_table_batch_stride=pps, page_block_size=page,
                   force_split_kernel=True, unpadded_lse=False, stream=st.s)
    ms = _timed(st, run, iters)
    kernel = (ah.lib.atoma_last_decode_kernel() or b"").decode()
    tot = int(lens.astype(np.int64).sum())
    nbytes = 2 * tot * hk * d * 2 + 2 * B * h * d * 2 + 4 * int(((lens + page - 1) // page).sum()) + 4 * B
    sample = []
    if sample_seqs:
        page_bytes = page * hk * d * 2
        qh, oh = q.numpy(np.uint16, (B, h, d)), o.numpy(np.uint16, (B, h, d))
        for b_ in sample_seqs:
            L = int(lens[b_])
            ks, vs = np.empty((pps, page, hk, d), np.uint16), np.empty((pps, page, hk, d), np.uint16)
            for j in range((L + page - 1) // page):
                for dst, src in ((ks, kc), (vs, vc)):
                    ah.hip_check(ah.hip.hipMemcpy(dst[j].ctypes.data, src.ptr + int(bt[b_, j]) * page_bytes, page_bytes, ah.D2H), "sample page")
            sample.append({"kind": "decode", "q": qh[b_].copy(), "o": oh[b_].copy(), "k": ks.reshape(pps * page, hk, d)[:L].copy(),
                           "v": vs.reshape(pps * page, hk, d)[:L].copy(), "scale": d ** -0.5, "L": L})
    for b_ in (kc, vc, q, o, dbt, dl):
        b_.free()
    return {"sample": sample, "workload": name, "ms": round(ms, 4), "kernel": kernel, "algorithmic_bytes": int(nbytes),
            "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_hbm": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "decode_tokens_per_s": round(B / (ms * 1e-3))}


def c2b_mha(iters=20):
    """the headline workload with 32 kv heads (MHA stress): 17.18 GB per call."""
    return _decode_case("C2b paged decode, MHA stress: bs=256, 32 q / 32 kv heads, d=128, seq=4096, block 16, bf16, random block table", 256, 4096, 32, 32, False, iters, 2, (3, 200))


def c2c_ragged(iters=20):
    """the headline workload with ragged lengths U[2048, 4096] (bytes from the actual lengths)."""
    return _decode_case("C2c paged decode, ragged lengths U[2048,4096]: bs=256, 32 q / 8 kv heads, d=128, block 16, bf16, random block table", 256, 4096, 32, 8, True, iters, 3,
                        (0, 101, 255))


def k4_reshape_and_cache(iters=20, T=8192, hk=8, d=128, page=16, seed=5):
    """reshape_and_cache_flash, T new tokens of 8 kv heads x 128 into random slots; bytes = 4 T h_k d 2 + 8 T.  The check is the
    op's definition (cache row of slot[t] == key[t]), bit-exact, on every token."""
    rng = np.random.default_rng(seed)
    nb = 2 * T // page
    k, v = TS.rand_dev(rng, T * hk * d * 2), TS.rand_dev(rng, T * hk * d * 2)
    kc, vc = ah.DeviceBuffer.zeros((nb * page * hk * d,), np.uint16), ah.DeviceBuffer.zeros((nb * page * hk * d,), np.uint16)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    ds = ah.DeviceBuffer.from_numpy(slots)
    st = ah.Stream()

    def run():
        ah.lib.reshape_and_cache_flash(k.ptr, v.ptr, kc.ptr, vc.ptr, ds.ptr, page * hk * d, T, hk, d, page, hk * d, hk * d, 1, st.s)
    ms = _timed(st, run, iters)
    kh, vh = k.numpy(np.uint16, (T, hk * d)), v.numpy(np.uint16, (T, hk * d))
    ok = bool(np.array_equal(kc.numpy(np.uint16, (nb * page, hk * d))[slots], kh) and np.array_equal(vc.numpy(np.uint16, (nb * page, hk * d))[slots], vh))
    nbytes = 4 * T * hk * d * 2 + 8 * T
    return {"workload": f"K4 reshape_and_cache_flash: {T} tokens x {hk} kv heads x {d}, block {page}, bf16, random slots", "ms": round(ms, 4), "algorithmic_bytes": nbytes,
            "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_hbm": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "bit_exact": ok, "checked_tokens": T}


def k5_copy_blocks(iters=10, layers=32, pairs=1024, page_bytes=32768, nb=4096, seed=6):
    """copy_blocks over 32 layers x (K, V), 1024 (src, dst) page pairs of 32 KiB: bytes = 4 P L 32 KiB (read + write, K and V).
    Check: every destination page equals its source page, bit for bit, in the first and the last layer."""
    rng = np.random.default_rng(seed)
    kcs, vcs = [TS.rand_dev(rng, nb * page_bytes) for _ in range(layers)], [TS.rand_dev(rng, nb * page_bytes) for _ in range(layers)]
    perm = rng.permutation(nb)
    m = np.stack([perm[:pairs], perm[pairs:2 * pairs]], 1).astype(np.int64)      # disjoint sources and destinations
    dm = ah.DeviceBuffer.from_numpy(m)
    kp, vp = ah.DeviceBuffer.from_numpy(np.array([b.ptr for b in kcs], np.int64)), ah.DeviceBuffer.from_numpy(np.array([b.ptr for b in vcs], np.int64))
    st = ah.Stream()

    def run():
        ah.lib.copy_blocks_bf16(kp.ptr, vp.ptr, dm.ptr, layers, pairs, page_bytes // 2, st.s)
    ms = _timed(st, run, iters)
    ok = True
    for buf in (kcs[0], vcs[-1]):
        pg = buf.numpy(np.uint16, (nb, page_bytes // 2))
        ok = ok and bool(np.array_equal(pg[m[:, 1]], pg[m[:, 0]]))
    nbytes = 4 * pairs * layers * page_bytes
    return {"workload": f"K5 copy_blocks: {layers} layers x (K, V), {pairs} page pairs of {page_bytes // 1024} KiB, bf16", "ms": round(ms, 4), "algorithmic_bytes": nbytes,
            "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_hbm": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "bit_exact": ok}


def n1_norm_rope(iters=20, T=2048, hidden=4096, h=32, hk=8, d=128, seed=7):
    """RMSNorm over [T, 4096] (bytes 2 T hidden 2) and RoPE of q and k in place ([T, (32 + 8) x 128]: bytes 2 T (h + h_k) d 2).
    bench.py checks sampled rows of both against the f32 definitions (oracle/norm_rope_oracle.py)."""
    rng = np.random.default_rng(seed)
    c = DS.Config(1, hidden, h, hk, d, 14336, 128256)
    x, w = TS.rand_dev(rng, T * hidden * 2), TS.norm_dev(rng, hidden)
    y = ah.DeviceBuffer(T * hidden * 2)
    st = ah.Stream()
    ms_norm = _timed(st, lambda: ah.lib.atoma_rms_norm(x.ptr, w.ptr, y.ptr, T, hidden, hidden, hidden, c.eps, 1, st.s), iters)
    cos, sin = DS.rope_tables(c)
    dcos, dsin = ah.DeviceBuffer.from_numpy(cos), ah.DeviceBuffer.from_numpy(sin)
    qk0 = TS.rand_dev(rng, T * (h + hk) * d * 2)
    qk = ah.DeviceBuffer(T * (h + hk) * d * 2)
    pos = rng.integers(0, c.max_pos, T).astype(np.int64)
    dpos = ah.DeviceBuffer.from_numpy(pos)
    row = (h + hk) * d

    def rope():
        assert ah.lib.atoma_rope_qk(qk.ptr, qk.ptr + h * d * 2, dcos.ptr, dsin.ptr, dpos.ptr, T, h, hk, d, row, row, 1, 1, st.s) == 0, ah.last_error()
    ms_rope = _timed(st, rope, iters)
    # a fresh, single application for the sample (the timed calls rotated the buffer in place many times)
    ah.hip_check(ah.hip.hipMemcpy(qk.ptr, qk0.ptr, T * row * 2, ah.D2D), "restore")
    rope()
    st.synchronize()
    rows = [0, 1, T // 2, T - 1]
    xs, ys = x.numpy(np.uint16, (T, hidden)), y.numpy(np.uint16, (T, hidden))
    q0, q1 = qk0.numpy(np.uint16, (T, h + hk, d)), qk.numpy(np.uint16, (T, h + hk, d))
    sample = [{"kind": "rms_norm", "x": xs[rows].copy(), "w": w.numpy(np.uint16, (hidden,)), "y": ys[rows].copy(), "eps": c.eps},
              {"kind": "rope", "x": q0[rows].copy(), "y": q1[rows].copy(), "cos": cos[pos[rows]].copy(), "sin": sin[pos[rows]].copy()}]
    nb_norm, nb_rope = 2 * T * hidden * 2, 2 * T * (h + hk) * d * 2
    return {"sample": sample,
            "rms_norm": {"workload": f"N1 RMSNorm [{T}, {hidden}] bf16", "ms": round(ms_norm, 4), "algorithmic_bytes": nb_norm, "GBps": round(nb_norm / (ms_norm * 1e-3) / 1e9, 1),
                         "frac_hbm": round(nb_norm / (ms_norm * 1e-3) / HBM_PEAK, 4)},
            "rope": {"workload": f"N1 RoPE of q and k in place, [{T}, ({h} + {hk}) x {d}] bf16, random positions", "ms": round(ms_rope, 4), "algorithmic_bytes": nb_rope,
                     "GBps": round(nb_rope / (ms_rope * 1e-3) / 1e9, 1), "frac_hbm": round(nb_rope / (ms_rope * 1e-3) / HBM_PEAK, 4)}}


def p1_prefill_4096(iters=10):
    """BASELINE.md P1 at S = 4096: B such that the tokens in flight stay at 8192 -> 2 prompts of 4096."""
    return prefill(iters=iters, S=4096, nseq=2, with_sample=True)


def p2_prefill_d96(iters=10):
    """Prefill at a head size without a hand-scheduled kernel (Phi-3-mini's 96): attn_generic.hip's 64-row tiled kernel, 4 causal prompts of 2048."""
    return prefill(iters=iters, S=2048, nseq=4, d=96, with_sample=True)


def p2_prefill_d256(iters=10):
    """The same at head size 256 (Gemma-class heads)."""
    return prefill(iters=iters, S=2048, nseq=4, d=256, with_sample=True)


def c3_trace():
    """BASELINE configs[2] AS WRITTEN, end to end (tools/engine_trace.py): 256 requests, prompts of 2048 tokens prefilled
    two per graph replay, then 512 decode steps at batch 256 with continuous-batching metadata rebuilt every step (atoma_prepare_inputs) and
    the sampled tokens read back -- wall-clock tokens/s/GPU over the decode phase (host loop included) against SURVEY 8(d)'s byte bound, the
    prefill phase beside it, and a sample of the LAST step's last-layer attention for the oracle."""
    import engine_trace
    r = engine_trace.run()
    keep = ("workload", "prefill_s", "prefill_tokens_per_s", "decode_s", "decode_ms_per_step", "decode_tokens_per_s_per_gpu", "decode_roofline_tokens_per_s", "decode_attention_kernel",
            "decode_frac_of_roofline", "host_metadata_ms_per_step", "trace_s", "generated_tokens_per_s_over_trace", "data", "sample")
    return {k: r[k] for k in keep}


def c3_trace_ragged():
    """VARIANT of c3_trace (not BASELINE configs[2], which has equal prompts): the requests' prompts are 2048 - U[0, 512) tokens long, so every decode batch is
    ragged (contexts spread over 512 tokens, as in a serving loop) and the DEFAULT dispatch -- atoma_prepare_inputs has seen that the lengths differ -- takes
    paged_decode_pair_kernel (`decode_attention_kernel`).  The same checks: a sample of the last step's last-layer attention against the oracle."""
    import engine_trace
    r = engine_trace.run(ragged_spread=512)
    keep = ("workload", "decode_s", "decode_ms_per_step", "decode_tokens_per_s_per_gpu", "decode_roofline_tokens_per_s", "decode_attention_kernel", "decode_frac_of_roofline",
            "host_metadata_ms_per_step", "data", "sample")
    return {k: r[k] for k in keep}


def c5_swap_sizes(iters=3):
    """BASELINE.md C5's other map sizes: swap-out / swap-in of n = 16 and n = 4096 random distinct pages (n = 256 is extra.swap)."""
    return {"n16": swap(iters=iters, pages=16, nb=512), "n4096": swap(iters=iters, pages=4096, nb=8192, tensors=16)}


def k4_sizes(iters=20):
    """BASELINE.md K4's other sizes: reshape_and_cache_flash at T = 256 (a decode batch) and T = 2048 (one prompt)."""
    return {"T256": k4_reshape_and_cache(iters=iters, T=256), "T2048": k4_reshape_and_cache(iters=iters, T=2048)}


def k5_sizes(iters=10):
    """BASELINE.md K5's other sizes: copy_blocks of P = 1 and P = 64 pairs over 32 layers."""
    return {"P1": k5_copy_blocks(iters=iters, pairs=1, nb=64), "P64": k5_copy_blocks(iters=iters, pairs=64, nb=512)}


def n1_norm_rope_t256(iters=20):
    """BASELINE.md N1 at T = 256 (a decode batch)."""
    return n1_norm_rope(iters=iters, T=256)


def c2c_identity(iters=20):
    """the headline shape with the identity block table (physical page i = logical page i)."""
    return _decode_case("C2c decode identity table: B=256, 32/8 heads, d=128, seq 4096, identity block table", 256, 4096, 32, 8, False, iters, 8, (3,), identity=True)


def c2a_seeds(iters=20):
    """The headline workload on seeds 1 and 2 (seeds 0, 1, 2; seed 0 is the headline itself)."""
    return {"seed%d" % s_: {k: v for k, v in _decode_case("C2a decode, seed %d" % s_, 256, 4096, 32, 8, False, iters, s_, ()).items() if k != "sample"} for s_ in (1, 2)}


def c4_rank_step(iters=10):
    """One rank of the Llama-3.1-70B TP = 8 decode step of configs[3] without its all-reduces (tools/rank_step.py): what a rank
    computes between the exchanges, batch 64, context 4096, 80 layers."""
    import rank_step
    return rank_step.run(iters=iters)


ALL = ("c3_decode_step", "c3_decode_step_paired", "c3_trace", "c3_trace_ragged", "c3_decode_step_fp8_kv", "c4_rank_step", "prefill", "p1_prefill_4096", "p2_prefill_d96", "p2_prefill_d256", "c2b_mha", "c2c_ragged", "c2c_ragged_paired", "c2c_identity",
       "c2a_seeds", "k4_reshape_and_cache", "k4_sizes", "k5_copy_blocks", "k5_sizes", "n1_norm_rope", "n1_norm_rope_t256", "swap", "c5_swap_sizes")


def collect(which=ALL, prefill_sample=False):
    """Every entry that carries a "sample" (host copies of a few inputs / outputs of the TIMED calls) is checked by bench.py against the
    oracle; bench.py pops the samples before it prints the line."""
    out = {}
    for name in which:
        try:
            out[name] = globals()[name](with_sample=True) if (name == "prefill" and prefill_sample) else globals()[name]()
        except Exception as e:          # an extra must never take the headline down with it
            out[name] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    import json
    ah.set_device(0)
    res = collect(tuple(sys.argv[1:]) or ALL)
    for v in res.values():
        if isinstance(v, dict):
            v.pop("sample", None)
    print(json.dumps(res))
