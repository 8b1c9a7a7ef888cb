#!/usr/bin/env python3
"""A few launches of the generic prefill kernel at one head size (for counter passes): 4 causal prompts of 2048 tokens, 32 q / 8 kv heads."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_extra as BE  # noqa: E402

BE.ah.set_device(0)
print(BE.prefill(iters=3, S=2048, nseq=4, d=int(sys.argv[1])))
