#!/usr/bin/env python3
"""Prefill at the head sizes without a tuned kernel: the 64-row LDS-staged kernel (attn_prefill_tile64_kernel) and the 16-row kernel (attn_prefill_tile16_kernel) against the row-per-wavefront
kernels they replace (atoma_set_option generic_prefill_tile / generic_prefill_rq), same process, same buffers.  4 causal prompts of 2048 tokens, 32 q / 8 kv heads."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_extra as BE  # noqa: E402

BE.ah.set_device(0)
res = {}
for d in (32, 96, 160, 192, 256):
    e = {}
    for name, env, rq in (("tile64", None, None), ("tile64_one_row_block_per_wavefront", None, "1"), ("tile16", "16", None), ("row_per_wavefront", "0", None)):
        if name.startswith("tile64_one") and d > 128:
            continue                    # (above head size 128 the default already is one block per wavefront)
        BE.ah.lib.atoma_set_option(b"generic_prefill_tile", 64 if env is None else int(env))
        BE.ah.lib.atoma_set_option(b"generic_prefill_rq", 0 if rq is None else int(rq))
        r = BE.prefill(iters=3 if env else 10, S=2048, nseq=4, d=d)
        e[name] = {"ms": r["ms"], "TFLOPs": r["TFLOPs"]}
    BE.ah.lib.atoma_set_option(b"generic_prefill_tile", 64)
    BE.ah.lib.atoma_set_option(b"generic_prefill_rq", 0)
    e["speedup_over_row"] = round(e["row_per_wavefront"]["ms"] / e["tile64"]["ms"], 2)
    res["d=%d" % d] = e
print(json.dumps(res, indent=1))
