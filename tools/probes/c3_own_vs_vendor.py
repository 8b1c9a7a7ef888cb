#!/usr/bin/env python3
"""configs[2] decode step (Llama-3.1-8B, batch 256, contexts U[2048,2560), hipGraph): the library's own projections with fused epilogues
(linear_wide_kernel: q/k/v + RoPE + cache write, o + residual, gate/up + SiLU.up, down + residual) against the vendor GEMM + separate
epilogue kernels, interleaved in one process on the same weights.   python tools/probes/c3_own_vs_vendor.py [rounds]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_extra as BE  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(9)
w = BE.TS.random_shard_weights(rng, BE.DS.LLAMA_3_1_8B)
res = {"own": [], "vendor": []}
for r in range(rounds):
    for mode, mb in (("vendor", "128"), ("own", "256")):      # 128: the vendor GEMM above 128 rows (the default up to round 5)
        os.environ["ATOMA_STEP_FUSED_MAX_BATCH"] = mb
        out = BE.c3_decode_step(iters=10, weights=w)
        res[mode].append(out["ms_per_step"])
print(json.dumps({k: {"ms": v, "median": float(np.median(v))} for k, v in res.items()}))
