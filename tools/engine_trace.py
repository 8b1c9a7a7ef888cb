"""C3-lite: a continuous-batching trace on one GPU, composed from the C-ABI entry points (bench plumbing, Python host loop).

SURVEY.md 8d C3: Llama-3.1-8B shapes, synthetic bf16 weights, 256 requests with 2048-token prompts and 512 decode steps,
batch <= 256, block size 16.  Phase 1 prefills the prompts two at a time (a hipGraph of PrefillStep replayed per
pair); phase 2 runs the decode steps at batch 256: per step the sampled tokens come back to the host (1 KiB), the
step's metadata is rebuilt by atoma_prepare_inputs (packed, one pinned H2D copy) and the captured decode graph is replayed
on the uploaded tensors.  Prints one JSON line: prefill tokens/s, decode tokens/s, whole-trace tokens/s.

    python tools/engine_trace.py [--requests 256] [--prompt 2048] [--decode-steps 512]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk  # noqa: E402
import decode_step as DS  # noqa: E402

ah = bk.ah


def random_weights(rng, c):
    w = dict(emb=bk.rand_dev(rng, c.vocab * c.hidden * 2), lm_head=bk.rand_dev(rng, c.vocab * c.hidden * 2), norm_f=bk.rand_dev(rng, c.hidden * 2),
             norm1=[bk.rand_dev(rng, c.hidden * 2) for _ in range(c.layers)], norm2=[bk.rand_dev(rng, c.hidden * 2) for _ in range(c.layers)],
             wqkv=[], wo=[], wgu=[], wdown=[])
    cos, sin = DS.rope_tables(c)
    w["cos"], w["sin"] = ah.DeviceBuffer.from_numpy(cos), ah.DeviceBuffer.from_numpy(sin)
    for _ in range(c.layers):
        w["wqkv"].append(bk.rand_dev(rng, c.qkv * c.hidden * 2))
        w["wo"].append(bk.rand_dev(rng, c.hidden * c.h * c.d * 2))
        w["wgu"].append(bk.rand_dev(rng, 2 * c.inter * c.hidden * 2))
        w["wdown"].append(bk.rand_dev(rng, c.hidden * c.inter * 2))
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=256)
    ap.add_argument("--prompt", type=int, default=2048)
    ap.add_argument("--decode-steps", type=int, default=512)
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (quick runs)")
    ap.add_argument("--model", default="8b", choices=["8b", "70b-tp8-shard"],
                    help="70b-tp8-shard: what ONE rank of a TP=8 Llama-3.1-70B job computes (8 q heads / 1 kv head, 1/8 of the MLP and of "
                         "the vocabulary); the two all-reduces of [batch, 8192] per layer are NOT included (single-GPU box)")
    ap.add_argument("--prompts-per-launch", type=int, default=2, help="prompts prefilled together (one varlen batch per graph replay)")
    a = ap.parse_args()
    ah.set_device(0)
    out = run(a.requests, a.prompt, a.decode_steps, a.layers, a.model, a.prompts_per_launch)
    out.pop("sample", None)
    print(json.dumps(out), flush=True)


def run(requests=256, prompt=2048, decode_steps=512, layers=0, model="8b", prompts_per_launch=2, sample_seqs=(0, 131), ragged_spread=0):
    """The trace; returns the result dict.  "sample" = host copies of what the LAST decode step's LAST layer attended over for `sample_seqs`
    (its rotated q, the attention output the timed graph wrote, the sequence's K / V gathered from that layer's cache through the trace's
    block table): bench.py checks them against the oracle -- every cache write, slot mapping and block table of the whole trace is behind them.
    ragged_spread > 0 (a VARIANT, not BASELINE configs[2]): request i's prompt is its first prompt - U[0, ragged_spread) tokens -- the prefill phase still
    computes `prompt` positions per request (its graph has one shape), the decode phase starts at the shorter lengths and overwrites the rest of the cache
    rows as it goes, so the decode batches are ragged as in a serving loop and atoma_prepare_inputs' length hint picks the dispatch."""
    rng = np.random.default_rng(3)
    c = DS.LLAMA_3_1_8B if model == "8b" else DS.Config(80, 8192, 8, 1, 128, 28672 // 8, 128256 // 8)
    if layers:
        c = DS.Config(layers, c.hidden, c.h, c.hk, c.d, c.inter, c.vocab)
    B, P, N = requests, prompt, decode_steps
    pps = (P + N + c.page - 1) // c.page                                  # pages a request owns (reserved up front)
    num_pages = B * pps + 1
    w = random_weights(rng, c)
    st = ah.Stream()
    step = DS.DecodeStep(c, B, num_pages, pps, w, st, fused_epilogues=True)
    G = prompts_per_launch
    assert B % G == 0
    pre = DS.PrefillStep(c, P * G, step, st, prompts=G)
    tables = (1 + rng.permutation(B * pps)).astype(np.uint32).reshape(B, pps)   # page 0 is never used
    tokens = np.zeros((B, P + N + 1), np.uint32)
    tokens[:, :P] = rng.integers(0, c.vocab, (B, P))
    lengths = np.full(B, P, np.int64)
    rows_all = np.arange(B)

    # ---- phase 1: prefill, one prompt per graph replay ----
    slots_of = lambda r: (tables[r, np.arange(P) // c.page].astype(np.int64) * c.page + np.arange(P) % c.page)
    group = lambda r: (tokens[r:r + G, :P].reshape(-1), np.concatenate([slots_of(i) for i in range(r, r + G)]))
    pre.set_inputs(*group(0))
    pre.run(); st.synchronize()                                          # warm-up: workspaces, hipBLASLt plans (autotuned on first use)
    with ah.Graph.capture(st) as gpre:
        pre.run()
    first = np.zeros(B, np.int32)
    t0 = time.perf_counter()
    for r in range(0, B, G):
        pre.set_inputs(*group(r))
        gpre.launch()
        st.synchronize()
        first[r:r + G] = pre.next_id.numpy(np.int32, (G,))
    t_prefill = time.perf_counter() - t0
    if ragged_spread:
        lengths -= rng.integers(0, ragged_spread, B)
    tokens[rows_all, lengths] = first                                    # (ragged: a synthetic first token -- the one sampled at position prompt - 1)
    lengths += 1

    # ---- phase 2: decode at batch B, metadata through atoma_prepare_inputs ----
    descs = (ah.SeqDesc * B)()
    for r in range(B):
        d = descs[r]
        d.is_prompt, d.no_block_tables, d.length, d.num_computed_tokens, d.token_chunk_size = 0, 0, int(lengths[r]), 0, 1
        d.token_ids = tokens[r].ctypes.data_as(C.POINTER(C.c_uint32))
        d.block_table, d.block_table_len = tables[r].ctypes.data_as(C.POINTER(C.c_uint32)), pps
    desc_view = np.frombuffer(descs, dtype=np.dtype([("is_prompt", "<i4"), ("nbt", "<i4"), ("length", "<i8"), ("computed", "<i8"), ("chunk", "<i8"),
                                                     ("tok", "<u8"), ("bt", "<u8"), ("btlen", "<i8")]))
    lay = ah.BatchLayout()
    assert ah.lib.atoma_prepare_inputs(descs, B, c.page, 0, 0, None, 0, None, 0, C.byref(lay), None) == 0, ah.last_error()
    host = ah.lib.atoma_host_alloc(lay.total_bytes)
    meta = ah.DeviceBuffer(lay.total_bytes)
    step.bind_metadata(meta.ptr + lay.off_input_tokens, meta.ptr + lay.off_input_positions, meta.ptr + lay.off_slot_mapping,
                       meta.ptr + lay.off_seq_lens, meta.ptr + lay.off_block_tables, int(lay.max_block_table_len))

    def upload_metadata():
        assert ah.lib.atoma_prepare_inputs(descs, B, c.page, 0, 0, host, lay.total_bytes, meta.ptr, lay.total_bytes, C.byref(lay), st.s) == 0, ah.last_error()
    upload_metadata()
    step.run(); st.synchronize()
    with ah.Graph.capture(st) as gdec:
        step.run()
    rows = np.arange(B)
    host_s = 0.0
    t0 = time.perf_counter()
    for i in range(N):
        h0 = time.perf_counter()
        upload_metadata()
        host_s += time.perf_counter() - h0
        gdec.launch()
        st.synchronize()
        nxt = step.next_ids.numpy(np.int32, (B,))                        # sampled tokens back to the host (detokenizer, stop checks)
        h0 = time.perf_counter()
        tokens[rows, lengths] = nxt
        lengths += 1
        desc_view["length"] = lengths
        host_s += time.perf_counter() - h0
    t_decode = time.perf_counter() - t0
    weight_bytes = 2 * (c.vocab * c.hidden + c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden))   # lm_head + layers
    kv_bytes_per_token = 2 * c.layers * c.hk * c.d * 2
    # ---- sample: the last step's last-layer attention of a few sequences (buffers are shared across layers: they hold the last layer's values)
    sample = []
    hd, qkvw = c.h * c.d, c.qkv
    qkv_h = step._buf("qkv", 0, B * qkvw * 2).numpy(np.uint16, (B, qkvw))
    att_h = step._buf("att", 0, B * hd * 2).numpy(np.uint16, (B, c.h, c.d))
    page_bytes = c.page * c.hk * c.d * 2
    for b_ in [x for x in sample_seqs if x < B]:
        L = int(lengths[b_]) - 1                           # keys the last step attended over, its own token included (`lengths` already counts the token it sampled)
        npg = (L + c.page - 1) // c.page
        ks, vs = np.empty((npg, c.page, c.hk, c.d), np.uint16), np.empty((npg, c.page, c.hk, c.d), np.uint16)
        for j in range(npg):
            for dst, src in ((ks, step.kc[c.layers - 1]), (vs, step.vc[c.layers - 1])):
                ah.hip_check(ah.hip.hipMemcpy(dst[j].ctypes.data, src.ptr + int(tables[b_, j]) * page_bytes, page_bytes, ah.D2H), "sample page")
        sample.append({"kind": "decode", "q": qkv_h[b_, :hd].reshape(c.h, c.d).copy(), "o": att_h[b_].copy(), "k": ks.reshape(npg * c.page, c.hk, c.d)[:L].copy(),
                       "v": vs.reshape(npg * c.page, c.hk, c.d)[:L].copy(), "scale": c.d ** -0.5, "L": L})
    name = "C3-lite trace: Llama-3.1-8B shapes" if model == "8b" else "C4-lite trace: one rank of Llama-3.1-70B TP=8 (no all-reduce)"
    out = {"workload": f"{name} ({c.layers} layers), {B} requests, prompt {P}" + (f" - U[0,{ragged_spread}) (ragged decode batches)" if ragged_spread else "") + f", {N} decode steps, block {c.page}",
           "prefill_s": round(t_prefill, 3), "prefill_tokens_per_s": round(B * P / t_prefill), "prefill_ms_per_prompt": round(t_prefill / B * 1e3, 2),
           "decode_s": round(t_decode, 3), "decode_ms_per_step": round(t_decode / N * 1e3, 3), "decode_tokens_per_s": round(B * N / t_decode),
           "host_metadata_ms_per_step": round(host_s / N * 1e3, 4), "trace_s": round(t_prefill + t_decode, 3),
           "generated_tokens_per_s_over_trace": round(B * (N + 1) / (t_prefill + t_decode)),
           "decode_roofline_tokens_per_s": round(B / ((weight_bytes + (float(lengths.sum()) - B * (N / 2 + 1)) * kv_bytes_per_token) / 8e12)),   # mean context over the decode phase
           "data": "synthetic weights and prompts; greedy sampling on the device", "sample": sample}
    out["decode_attention_kernel"] = (ah.lib.atoma_last_decode_kernel() or b"").decode()     # (ragged batches packed by atoma_prepare_inputs: the paired kernel by default)
    ah.lib.atoma_hint_decode_lengths(0, 0, 0)                              # the trace's batches are gone: later decode calls of this process must not inherit their hint
    out["decode_tokens_per_s_per_gpu"] = out["decode_tokens_per_s"]
    out["decode_frac_of_roofline"] = round(out["decode_tokens_per_s"] / out["decode_roofline_tokens_per_s"], 4)
    for b_ in step.kc + step.vc:
        b_.free()
    for v in w.values():
        for b_ in (v if isinstance(v, list) else [v]):
            b_.free()
    return out


if __name__ == "__main__":
    main()
