"""Diagnostic: where do two in-process TP ranks on one device stop meeting? (status after every phase)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG_S", "60")), exit=True)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tp_step as TS  # noqa: E402

ah, DS, tp = TS.ah, TS.DS, TS.tp
os.environ["ATOMA_XGMI_TIMEOUT_MS"] = "2000"
os.environ["ATOMA_XGMI_ONESHOT_MAX"] = str(1 << 20)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = DS.Config(layers, 8192, 64, 8, 128, 28672, 128256)
ah.set_device(0)
W = 2
ranks = []
for r in range(W):
    rk = TS.Rank.__new__(TS.Rank)
    TS.Rank.__init__(rk, cfg, r, 8, B, 4096, 0)
    ranks.append(rk)
xs = []
for r in range(W):
    h = C.c_void_p()
    assert ah.lib.atoma_xgmi_create(C.byref(h), r, W, 0, 1 << 20) == 0
    xs.append(h)
blobs = (C.c_uint8 * (128 * W))()
for r in range(W):
    one = (C.c_uint8 * 128)()
    ah.lib.atoma_xgmi_handle(xs[r], one)
    C.memmove(C.addressof(blobs) + 128 * r, one, 128)
for r in range(W):
    assert ah.lib.atoma_xgmi_connect(xs[r], blobs) == 0


def status(tag):
    print(tag, [ah.lib.atoma_xgmi_status(x) for x in xs], flush=True)


for rk in ranks:
    rk.engine = lambda ptr, count: None
    rk.step.run()
    rk.stream.synchronize()
status("warm (no exchange)")
calls = [0, 0]
for mode in ([int(m) for m in os.environ.get("MODES", "2,1").split(",")]):
    for r, rk in enumerate(ranks):
        def eng(ptr, count, r=r, rk=rk, mode=mode):
            calls[r] += 1
            assert ah.lib.atoma_xgmi_allreduce_sum_mode(xs[r], ptr, ptr, count, 1, mode, rk.stream.s) == 0, ah.last_error()
        rk.engine = eng
    # all-reduce alone, both ranks
    t0 = time.perf_counter()
    for rk in ranks:
        for _ in range(4):
            rk.engine(rk.ar_buf.ptr, B * 8192)
    for rk in ranks:
        rk.stream.synchronize()
    status(f"mode {mode}: 4 all-reduces alone, {time.perf_counter() - t0:.3f} s")
    t0 = time.perf_counter()
    for rk in ranks:
        rk.step.run()
    t1 = time.perf_counter()
    for rk in ranks:
        rk.stream.synchronize()
    status(f"mode {mode}: eager step, enqueue {t1 - t0:.3f} s, total {time.perf_counter() - t0:.3f} s, calls {calls}")
    if any(ah.lib.atoma_xgmi_status(x) for x in xs):
        break

# ---- graphs ----
graphs = []
for rk in ranks:
    with ah.Graph.capture(rk.stream) as g:
        rk.step.run()
    graphs.append(g)
status("captured")
for it in range(3):
    for g in graphs:
        g.launch()
    for rk in ranks:
        rk.stream.synchronize()
    status(f"replay {it} (synced)")
t0 = time.perf_counter()
for it in range(5):
    for g in graphs:
        g.launch()
for rk in ranks:
    rk.stream.synchronize()
status(f"5 replays back to back: {(time.perf_counter() - t0) * 1e3 / 5:.3f} ms per step")
