#!/usr/bin/env python3
"""Per-phase cycle accounting of the prefill kernel (needs the -DPF_TIMING build: tools/probes/libatoma_hip_timing.so).
Prints, averaged over workgroups: cycles per K/V tile spent in (wait+barrier, DMA issue, QK^T, softmax, PV)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import atoma_hip as ah
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", os.environ.get("PF_TIMING_LIB", "libatoma_hip_timing.so")))
lib.run_mha.argtypes = ah._RUN_MHA_ARGS
lib.atoma_set_option.argtypes = [C.c_char_p, C.c_int]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk
ah.set_device(0)
cfg = int(os.environ.get("ATOMA_PREFILL_CFG", "0"))
lib.atoma_set_option(b"prefill_cfg", cfg)
rng = np.random.default_rng(1)
h, hk, d = 32, 8, 128
for S, nseq in ((2048, 4), (2048, 16)):
    T = S * nseq
    q, k, v = bk.rand_dev(rng, T * h * d * 2), bk.rand_dev(rng, T * hk * d * 2), bk.rand_dev(rng, T * hk * d * 2)
    o = ah.DeviceBuffer(T * h * d * 2)
    cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
    nwg = 8 * ((nseq * h + 7) // 8) * ((S + 127) // 128) * 2
    dbg = ah.DeviceBuffer.zeros((nwg, 8), np.float32)
    rnd = lambda x, m: (x + m - 1) // m * m
    args = [q.ptr, k.ptr, v.ptr, o.ptr, dbg.ptr, None, cu.ptr, cu.ptr, True, 0, 0, 0, 0, 0, h * d, hk * d, hk * d, h * d, d, d, d, d,
            0, nseq, h, hk, d, 128, float(d ** -0.5), float(d ** -0.5 * 1.4426950408889634), None, 0, 0, None, S, S, rnd(S, 128), rnd(S, 128),
            1, 1, -1, 0, 0.0, True, False, None, None]
    for _ in range(3):
        lib.run_mha(*args)
    ah.synchronize()
    t = dbg.numpy()
    t = t[t[:, 5] > 0]
    per_tile = t[:, :5].sum(0) / t[:, 5].sum()
    names = ["wait+barrier", "dma issue", "qk", "softmax", "pv"] if cfg != 2 else ["wait+barrier", "dma issue", "phase B", "phase C", "raise max"]
    print(f"S={S} x{nseq} cfg={cfg}: workgroups {len(t)}, tiles/wg {t[:,5].mean():.1f}, cycles per tile:",
          {n: int(x) for n, x in zip(names, per_tile)}, "sum", int(per_tile.sum()), "total/tiles", int(t[:, 6].sum() / t[:, 5].sum()), "vmcnt wait", int(t[:, 7].sum() / t[:, 5].sum()))
