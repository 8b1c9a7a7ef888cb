#!/usr/bin/env python3
"""attn_prefill_tile64_kernel: key-tile size (32 / 64, atoma_set_option generic_prefill_kt) per head size, 4 causal prompts of 2048 tokens, 32 q / 8 kv heads; and a longer prompt (2 x 8192) for the chosen default."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_extra as BE  # noqa: E402

BE.ah.set_device(0)
res = {}
for d in (32, 96, 160, 192, 224, 256):          # (64 / 128 without ALiBi run on the hand-scheduled kernels)
    e = {}
    for cfg in ("64", "32"):
        BE.ah.lib.atoma_set_option(b"generic_prefill_kt", int(cfg))
        r = BE.prefill(iters=5, S=2048, nseq=4, d=d)
        e[cfg] = r["ms"]
    BE.ah.lib.atoma_set_option(b"generic_prefill_kt", 0)
    e["best"] = min(e, key=e.get)
    res["d=%d" % d] = e
print(json.dumps(res, indent=1))
