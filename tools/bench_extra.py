"""The end-to-end numbers bench.py carries beside its headline, each timed with HIP events in the SAME run:

  c3_decode_step   one whole Llama-3.1-8B decode step on the device (BASELINE configs[2] mid-trace: batch 256, contexts
                   U[2048, 2560), block 16, hipGraph replay) -> ms, decode tokens/s/GPU, fraction of the byte roofline;
  prefill          causal varlen FlashAttention-2 prefill, 16 prompts of 2048 tokens, 32 q / 8 kv heads, d = 128
                   -> TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak;
  c4_rank_step     one rank of the 70B TP = 8 decode step of configs[3] without its all-reduces (batch 64, context 4096, 80 layers);
  swap             (+ the pinned hipMemcpyAsync ceiling of the same byte count beside it)
                   CPU<->GPU KV swap of BASELINE configs[4]: 64 tensors (32 layers x K, V), 256 pages of 32 KiB, pinned host
                   memory, both directions -> GB/s over PCIe.
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


def c3_decode_step(iters=10, batch=256, seed=9, kv_fp8=False, weights=None):
    rng = np.random.default_rng(seed)
    c = DS.LLAMA_3_1_8B
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
    out = {"workload": f"Llama-3.1-8B decode step (BASELINE configs[2] mid-trace), batch {batch}, contexts U[2048,2560), block {c.page}, bf16 weights and activations, "
                       + ("fp8 e4m3fn KV cache" if kv_fp8 else "bf16 KV cache") + ", hipGraph replay",
           "ms_per_step": round(ms, 3), "decode_tokens_per_s_per_gpu": round(batch / (ms * 1e-3), 1), "algorithmic_bytes": int(nbytes),
           "roofline_tokens_per_s": round(batch / (nbytes / HBM_PEAK), 1), "frac_of_hbm_roofline": round(nbytes / HBM_PEAK / (ms * 1e-3), 4)}
    del g, step
    if weights is None:
        del w
    return out


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


def c4_rank_step(iters=10):
    """One rank of the Llama-3.1-70B TP = 8 decode step of configs[3] without its all-reduces (tools/rank_step.py): what a rank
    computes between the exchanges, batch 64, context 4096, 80 layers."""
    import rank_step
    return rank_step.run(iters=iters)


def collect(which=("c3_decode_step", "c3_decode_step_fp8_kv", "c4_rank_step", "prefill", "swap"), prefill_sample=False):
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
    print(json.dumps(collect(tuple(sys.argv[1:]) or ("c3_decode_step", "c3_decode_step_fp8_kv", "c4_rank_step", "prefill", "swap"))))
