#!/usr/bin/env python3
"""Within-process interleaved A/B of the 17..64-row projection kernels (linear_tile_kernel with options linear_tile_*, linear_mid_kernel, the vendor GEMM) on the projection shapes of one rank of
the 70B TP = 8 job at 64 rows.  Every variant is timed in ROUNDS interleaved rounds of ITERS back-to-back launches (HIP events);
median and min per variant.  Weights rotate through NBUF copies so that a launch never finds its matrix in the 256 MB cache.

    python tools/probes/ks_ab.py [shape ...]      shapes: qkv o gate_up down (default: all)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk  # noqa: E402

ah = bk.ah
SHAPES = {"qkv": (1280, 8192, 0), "o": (8192, 1024, 1), "gate_up": (7168, 8192, 2), "down": (8192, 3584, 1)}
VARIANTS = [("tile", {"tile": 1}), ("tile nw64", {"tile": 1, "linear_tile_nw": 64}), ("tile nw128", {"tile": 1, "linear_tile_nw": 128}), ("tile nw32 s1", {"tile": 1, "linear_tile_nw": 32, "linear_tile_splits": 1}),
            ("tile nw64 s4", {"tile": 1, "linear_tile_nw": 64, "linear_tile_splits": 4}),
            ("mid kernel", {"mid": 1}), ("vendor", {"vendor": 1})]
DEFAULTS = {"linear_tile_nw": 0, "linear_tile_splits": 0}
ROUNDS, ITERS = 7, 12


def main():
    ah.set_device(0)
    rng = np.random.default_rng(11)
    B = int(os.environ.get("KS_AB_BATCH", "64"))
    for name in (sys.argv[1:] or list(SHAPES)):
        N, K, ep = SHAPES[name]
        nbuf = max(2, int(600e6 // (N * K * 2)))
        ws = [bk.rand_dev(rng, N * K * 2) for _ in range(nbuf)]
        x, r = bk.rand_dev(rng, B * K * 2), bk.rand_dev(rng, B * N * 2)
        y = ah.DeviceBuffer(B * N * 2)
        L = ah.lib
        vendor = {0: lambda w: L.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)}
        calls = {0: lambda w: L.atoma_linear_decode(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None),
                 1: lambda w: L.atoma_linear_decode_residual(x.ptr, w.ptr, r.ptr, y.ptr, B, K, N, K, K, N, N, 1, None),
                 2: lambda w: L.atoma_linear_decode_silu_mul(x.ptr, w.ptr, y.ptr, B, K, N // 2, K, K, N // 2, 1, None)}
        times = {v[0]: [] for v in VARIANTS}
        for rnd in range(ROUNDS + 1):
            for vname, opts in VARIANTS:
                for k, v in DEFAULTS.items():
                    assert L.atoma_set_option(k.encode(), opts.get(k, v)) == 0
                assert L.atoma_set_option(b"linear_tile", 1 if opts.get("tile") else 0) == 0
                a, b = ah.Event(), ah.Event()
                fn = vendor[0] if opts.get("vendor") else calls[ep]
                assert fn(ws[0]) == 0, ah.last_error()
                ah.synchronize()
                a.record(None)
                for i in range(ITERS):
                    fn(ws[i % nbuf])
                b.record(None)
                b.synchronize()
                if rnd > 0:
                    times[vname].append(a.elapsed_ms(b) / ITERS * 1e3)
        for k, v in DEFAULTS.items():
            L.atoma_set_option(k.encode(), v)
        L.atoma_set_option(b"linear_tile", 1)
        for vname, _ in VARIANTS:
            t = times[vname]
            if t:
                print(json.dumps({"shape": f"{name} [{N} x {K}] batch {B}", "variant": vname, "median_us": round(float(np.median(t)), 2), "min_us": round(min(t), 2),
                                  "GBps_W": round(N * K * 2 / np.median(t) / 1e3)}), flush=True)
        for w in ws:
            w.free()


if __name__ == "__main__":
    main()
