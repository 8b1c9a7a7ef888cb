#!/usr/bin/env python3
"""Within-process interleaved A/B of the 65..256-row projection kernels on the Llama-3.1-8B layer shapes at 256 rows (BASELINE configs[2]):
linear_wide_kernel (options linear_wide_nw / _splits / _xcd), linear_big_kernel (linear_wide = 0), the vendor GEMM (+ the separate epilogue
launch it needs).  ROUNDS interleaved rounds of ITERS back-to-back launches (HIP events); median and min per variant.  Weights rotate
through NBUF copies so that a launch never finds its matrix in the 256 MB cache.

    python tools/probes/linear256_ab.py [shape ...]      shapes: qkv o gate_up down (default: all);  L256_BATCH=256  L256_VARIANTS=name,name
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk  # noqa: E402

ah = bk.ah
SHAPES = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 1), "gate_up": (28672, 4096, 2), "down": (4096, 14336, 1), "qkv_rope": (6144, 4096, 3), "lm_head": (128256, 4096, 0),
          # what ONE rank of the 70B TP=8 step multiplies (BASELINE configs[3]; run with L256_BATCH=64)
          "r_qkv": (1280, 8192, 0), "r_o": (8192, 1024, 1), "r_gate_up": (7168, 8192, 2), "r_down": (8192, 3584, 1)}
VARIANTS = [("wide", {}), ("wide nw64", {"linear_wide_nw": 64}), ("wide nw128", {"linear_wide_nw": 128}),
            ("wide s1", {"linear_wide_splits": 1}), ("wide s2", {"linear_wide_splits": 2}), ("wide s4", {"linear_wide_splits": 4}), ("wide s8", {"linear_wide_splits": 8}),
            ("wide nw128 s4", {"linear_wide_nw": 128, "linear_wide_splits": 4}), ("wide nw128 s8", {"linear_wide_nw": 128, "linear_wide_splits": 8}),
            ("wide nw64 s2", {"linear_wide_nw": 64, "linear_wide_splits": 2}), ("wide nw64 s4", {"linear_wide_nw": 64, "linear_wide_splits": 4}),
            ("wide bb1", {"linear_wide_bb": 1}), ("wide bb2", {"linear_wide_bb": 2}), ("wide bb2 nw64 s1", {"linear_wide_bb": 2, "linear_wide_nw": 64, "linear_wide_splits": 1}),
            ("wide bb2 nw64 s2", {"linear_wide_bb": 2, "linear_wide_nw": 64, "linear_wide_splits": 2}), ("wide bb2 nw64 s4", {"linear_wide_bb": 2, "linear_wide_nw": 64, "linear_wide_splits": 4}),
            ("wide bb2 nw128 s2", {"linear_wide_bb": 2, "linear_wide_nw": 128, "linear_wide_splits": 2}), ("wide bb2 nw128 s4", {"linear_wide_bb": 2, "linear_wide_nw": 128, "linear_wide_splits": 4}),
            ("wide bb2 nw128 s1", {"linear_wide_bb": 2, "linear_wide_nw": 128, "linear_wide_splits": 1}),
            ("wide 128-row tiles", {"linear_wide_gu112": 0}),
            ("wide xcd map", {"linear_wide_xcd": 1}), ("wide W first", {"linear_wide_var": 1}), ("wide W plain", {"linear_wide_var": 2}), ("wide x nt", {"linear_wide_var": 4}),
            ("wide W first W plain", {"linear_wide_var": 3}),
            ("big kernel", {"linear_wide": 0}), ("vendor", {"vendor": 1}), ("vendor + epilogue", {"vendor": 2})]
DEFAULTS = {"linear_wide": 1, "linear_wide_nw": 0, "linear_wide_splits": 0, "linear_wide_xcd": 0, "linear_wide_var": 0, "linear_wide_bb": 0, "linear_wide_gu112": 1}
ROUNDS, ITERS = 7, 12


def main():
    ah.set_device(0)
    rng = np.random.default_rng(11)
    B = int(os.environ.get("L256_BATCH", "256"))
    only = [v for v in os.environ.get("L256_VARIANTS", "").split(",") if v]
    variants = [v for v in VARIANTS if not only or v[0] in only]
    for name in (sys.argv[1:] or list(SHAPES)):
        N, K, ep = SHAPES[name]
        nbuf = int(os.environ.get("L256_NBUF", "0")) or max(2, int(600e6 // (N * K * 2)))       # L256_NBUF=1: every launch finds its matrix where the last one left it (256 MB cache)
        ws = [bk.rand_dev(rng, N * K * 2) for _ in range(nbuf)]
        x, r = bk.rand_dev(rng, B * K * 2), bk.rand_dev(rng, B * N * 2)
        y, y2 = ah.DeviceBuffer(B * N * 2), ah.DeviceBuffer(B * N * 2)
        L = ah.lib
        calls = {0: lambda w: L.atoma_linear_decode(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None),
                 1: lambda w: L.atoma_linear_decode_residual(x.ptr, w.ptr, r.ptr, y.ptr, B, K, N, K, K, N, N, 1, None),
                 2: lambda w: L.atoma_linear_decode_silu_mul(x.ptr, w.ptr, y.ptr, B, K, N // 2, K, K, N // 2, 1, None)}

        def vendor(w, with_epilogue):
            rc = L.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)
            if with_epilogue and ep == 1:
                rc |= L.atoma_add(y.ptr, r.ptr, y2.ptr, B * N, 1, None)
            if with_epilogue and ep == 2:
                rc |= L.atoma_silu_mul(y.ptr, y.ptr + N, y2.ptr, B, N // 2, N, N, N // 2, 1, None)
            return rc
        if ep == 3:       # q/k/v projection + RoPE + KV-cache write (32 / 8 heads of 128): one entry against vendor GEMM + atoma_rope_qk_cache
            h, hk, d, page, nb = 32, 8, 128, 16, 64
            cos, sin = bk.rand_dev(rng, 8192 * d), bk.rand_dev(rng, 8192 * d)
            pos = ah.DeviceBuffer.from_numpy((rng.integers(0, 8192, B) * (0 if os.environ.get("L256_POS0") else 1)).astype(np.int64))
            slots = ah.DeviceBuffer.from_numpy(np.full(B, -1, np.int64) if os.environ.get("L256_NOSLOTS") else rng.permutation(nb * page)[:B].astype(np.int64))
            kc, vc = ah.DeviceBuffer(nb * page * hk * d * 2), ah.DeviceBuffer(nb * page * hk * d * 2)
            calls[3] = lambda w: L.atoma_linear_decode_qkv_rope_cache(x.ptr, w.ptr, y.ptr, kc.ptr, vc.ptr, slots.ptr, cos.ptr, sin.ptr, pos.ptr, B, K, h, hk, d, K, K, N,
                                                                      page * hk * d, page, 1, 1, None)

            def vendor(w, with_epilogue):   # noqa: F811
                rc = L.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)
                if with_epilogue:
                    rc |= L.atoma_rope_qk_cache(y.ptr, y.ptr + h * d * 2, y.ptr + (h + hk) * d * 2, kc.ptr, vc.ptr, slots.ptr, cos.ptr, sin.ptr, pos.ptr, B, h, hk, d,
                                                N, N, N, page * hk * d, page, 1, 1, None)
                return rc
        times = {v[0]: [] for v in variants}
        for rnd in range(ROUNDS + 1):
            for vname, opts in variants:
                for k, v in DEFAULTS.items():
                    assert L.atoma_set_option(k.encode(), opts.get(k, v)) == 0
                a, b = ah.Event(), ah.Event()
                fn = (lambda w, m=opts["vendor"]: vendor(w, m == 2)) if opts.get("vendor") else calls[ep]
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
        for vname, _ in variants:
            t = times[vname]
            if t:
                print(json.dumps({"shape": f"{name} [{N} x {K}] batch {B}", "weight_copies": nbuf, "variant": vname, "median_us": round(float(np.median(t)), 2), "min_us": round(min(t), 2),
                                  "GBps_W": round(N * K * 2 / np.median(t) / 1e3), "TFLOPs": round(2 * B * N * K / np.median(t) / 1e6)}), flush=True)
        for w in ws:
            w.free()


if __name__ == "__main__":
    main()
