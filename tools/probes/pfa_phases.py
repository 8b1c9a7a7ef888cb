#!/usr/bin/env python3
"""Phase timers of the hand-scheduled prefill kernel (needs `make -C atoma-infer_amd timing`: tools/probes/libatoma_hip_timing.so).
Per wavefront the kernel accumulates s_memtime differences: phase 1, phase 2, counted waits, barrier, loop control, prologue, epilogue,
iterations.  Prints cycles per K/V tile and per 256-row block, averaged over the wavefronts of the launch.
    python tools/probes/pfa_phases.py [S nseq causal exact_keys]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import atoma_hip as ah
import tp_step as TS
BLOCK = os.environ.get("PFA_TIMING_BLOCK", "0") == "1"      # libatoma_hip_timing2.so: the regions between two blocks' loops
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", os.environ.get("PFA_TIMING_LIB", "libatoma_hip_timing2.so" if BLOCK else "libatoma_hip_timing.so")))
lib.run_mha.argtypes = ah._RUN_MHA_ARGS
lib.atoma_set_option.argtypes = [C.c_char_p, C.c_int]
ah.set_device(0)
lib.atoma_set_option(b"prefill_cfg", 4)
rng = np.random.default_rng(1)
h, hk, d = 32, 8, 128
shapes = [(2048, 16, 1, 512), (2048, 16, 1, 0), (2048, 16, 0, 0), (512, 16, 1, 512)]
if len(sys.argv) > 4:
    shapes = [tuple(int(x) for x in sys.argv[1:5])]
for S, nseq, causal, ek in shapes:
    lib.atoma_set_option(b"prefill_exact_keys", ek)
    T = S * nseq
    q, k, v = (TS.rand_dev(rng, T * n * d * 2) for n in (h, hk, hk))
    o = ah.DeviceBuffer(T * h * d * 2)
    cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
    nblk = nseq * h * ((S + 255) // 256)
    nwg = min(256, 8 * ((nseq * h + 7) // 8) * ((S + 255) // 256))       # persistent: one workgroup per CU
    dbg = ah.DeviceBuffer.zeros((nwg * 4, 8), np.uint32)
    rnd = lambda x, m: (x + m - 1) // m * m
    args = [q.ptr, k.ptr, v.ptr, o.ptr, dbg.ptr, None, cu.ptr, cu.ptr, True, 0, 0, 0, 0, 0, h * d, hk * d, hk * d, h * d, d, d, d, d,
            0, nseq, h, hk, d, 128, float(d ** -0.5), float(d ** -0.5 * 1.4426950408889634), None, 0, 0, None, S, S, rnd(S, 128), rnd(S, 128),
            1, causal, -1, 0, 0.0, True, False, None, None]
    for _ in range(3):
        lib.run_mha(*args)
    ah.synchronize()
    t = dbg.numpy().astype(np.float64)
    t = t[t[:, :6].sum(1) > 0]
    nt = [(4 * (i + 1) if causal else (S + 63) // 64) for i in range((S + 255) // 256)]      # K/V tiles of the blocks of one (sequence, head)
    nt = [min(x, (S + 63) // 64) for x in nt]
    it = nseq * h * sum(nt) / nwg                                        # iterations per wavefront
    bpw = nblk / nwg                                                     # blocks per workgroup
    if BLOCK:
        names = ["drain+barrier", "stash + next entry fetched", "epilogue slot 0 (+ woven requests)", "epilogue slot 1 (+ woven requests)",
                 "peek + state + Q wait + first d-steps + K(0) barrier", "S(0) .. loop entry"]
        print(f"S={S} x{nseq} causal={causal} exact_keys={ek}: cycles per block:", {n: int(t[:, i].mean() / bpw) for i, n in enumerate(names)}, flush=True)
        continue
    names = ["phase1", "phase2", "wait+barrier", "control", "prologue", "epilogue"]
    per_tile = {n: int(t[:, i].mean() / it) for i, n in enumerate(names[:4])}
    tot = t[:, :6].sum(1)
    print(f"S={S} x{nseq} causal={causal} exact_keys={ek}: wavefronts {len(t)}, {bpw:.1f} blocks and {it:.0f} iterations per workgroup; cycles per iteration "
          f"{per_tile} sum {sum(per_tile.values())}; per block: prologue {int(t[:, 4].mean() / bpw)} epilogue {int(t[:, 5].mean() / bpw)} "
          f"loop {int(t[:, :4].sum(1).mean() / bpw)}; per wavefront total {int(tot.mean())} (min {int(tot.min())} max {int(tot.max())})", flush=True)
