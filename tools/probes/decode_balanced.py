"""Ragged decode batches: one wavefront per (sequence, kv head) against the balanced mode (option decode_stream),
see DESIGN.md 4.1."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk  # noqa: E402

opt = bk.ah.lib.atoma_set_option
rng = np.random.default_rng(5)
straggler = np.full(256, 1024, np.int32)
straggler[7] = 16384
longtail = np.minimum(16384, (512 * rng.pareto(1.5, 256) + 256)).astype(np.int32)


CASES = [
    ("uniform C2a", dict(B=256, S=4096, h=32, hk=8)),
    ("uniform S=3072", dict(B=256, S=3072, h=32, hk=8)),
    ("uniform B=320 S=4096 (1.25 x resident)", dict(B=320, S=4096, h=32, hk=8)),
    ("ragged U[2048,4096]", dict(B=256, S=4096, h=32, hk=8, ragged=True)),
    ("ragged U[2048,4096] MHA", dict(B=256, S=4096, h=32, hk=32, ragged=True)),
    ("ragged 70B shape", dict(B=256, S=4096, h=64, hk=8, ragged=True)),
    ("ragged d=64", dict(B=256, S=4096, h=32, hk=8, d=64, ragged=True)),
    ("straggler 255x1024 + 1x16384", dict(B=256, S=16384, h=32, hk=8, lens=straggler)),
    (f"long tail (Pareto, max 16384, mean {int(longtail.mean())})", dict(B=256, S=16384, h=32, hk=8, lens=longtail)),
]
for name, kw in CASES:          # modes interleaved per case: the clocks drift over a long run
    kw = dict(kw)
    B, S, h, hk = kw.pop("B"), kw.pop("S"), kw.pop("h"), kw.pop("hk")
    for rep in range(2):
        for mode in (0, 1):
            opt(b"decode_stream", mode)
            bk.decode_case(f"{'balanced    ' if mode else 'per-sequence'} {name}", B, S, h, hk, **kw)
opt(b"decode_stream", 1)
for wpc in (8, 12, 8, 12):
    opt(b"decode_stream_waves_per_cu", wpc)
    bk.decode_case(f"balanced, {wpc} wavefronts/CU: ragged U[2048,4096]", 256, 4096, 32, 8, ragged=True)
