#!/usr/bin/env python3
"""One rank of the tensor-parallel Llama-3.1-70B decode step (BASELINE.json configs[3]) on ONE GPU, without its all-reduces
("virtual rank": what a rank computes between the exchanges) -- bench plumbing over the C ABI.

TP = 8 shard: 8 q heads / 1 kv head, d = 128, hidden 8192, 1/8 of the MLP (3584) and of the vocabulary, 80 layers; batch 64,
context 4096.  Per layer a rank must move 214 MB of weights and 134 MB of KV cache = 43.5 us at 8 TB/s (3.56 ms per step with the
head); VERDICT r2 item 1 asks for <= 6 ms.  The step is captured in a hipGraph and replayed.

    python tools/rank_step.py [--layers 80] [--batch 64] [--context 4096] [--iters 20] [--tp 8]
    rocprofv3 --kernel-trace --output-format csv -d out -o step -- python tools/rank_step.py --layers 8 --iters 3
        then  python tools/step_breakdown.py out/.../step_kernel_trace.csv
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)
import atoma_hip as ah  # noqa: E402
import decode_step as DS  # noqa: E402
import tp_step as TS  # noqa: E402


def run(layers=80, batch=64, context=4096, iters=20, tp=8, model="70b", seed=5):
    ah.set_device(0)
    rng = np.random.default_rng(seed)
    full = TS.LLAMA_3_1_70B if model == "70b" else DS.LLAMA_3_1_8B
    c = DS.Config(layers, full.hidden, full.h // tp, max(1, full.hk // tp), full.d, full.inter // tp, full.vocab // tp)
    w = TS.random_shard_weights(rng, c)
    st = ah.Stream()
    B, S = batch, context
    pps = S // c.page + 1
    step = DS.DecodeStep(c, B, B * pps + 2, pps, w, st, fused_epilogues=True)
    bt = rng.permutation(B * pps).astype(np.int32).reshape(B, pps)
    ctx = np.full(B, S)
    slots = bt[np.arange(B), ctx // c.page].astype(np.int64) * c.page + ctx % c.page
    step.set_inputs(rng.integers(0, c.vocab, B), ctx, slots, ctx + 1, bt)
    step.run()
    st.synchronize()
    with ah.Graph.capture(st) as g:
        step.run()
    ms = TS.timed(st, g.launch, iters)
    # the head (embedding rows, final norm, lm_head shard, argmax) timed by difference against a 0-layer step is not needed:
    # per-layer time = (step - head) / layers with the head measured as its own graph
    nbytes = TS.step_bytes(c, B, S)
    layer_bytes = 2 * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden) + 2 * B * (S + 1) * c.hk * c.d * 2
    out = {"workload": f"one rank of Llama-3.1-{model.upper()} TP={tp} decode step ({layers} layers), batch {B}, context {S}, no all-reduce, hipGraph replay",
           "ms_per_step": round(ms, 4), "us_per_layer_incl_head": round(ms * 1e3 / layers, 2), "decode_tokens_per_s": round(B / (ms * 1e-3)),
           "algorithmic_bytes": int(nbytes), "frac_of_hbm_roofline": round(nbytes / (ms * 1e-3) / 8e12, 4),
           "layer_bytes": int(layer_bytes), "layer_us_at_8TBps": round(layer_bytes / 8e12 * 1e6, 2), "launches_per_layer": getattr(step, "launches_per_layer", None)}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=80)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--context", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tp", type=int, default=8)
    ap.add_argument("--model", default="70b", choices=["70b", "8b"])
    a = ap.parse_args()
    print(json.dumps(run(a.layers, a.batch, a.context, a.iters, a.tp, a.model)), flush=True)
