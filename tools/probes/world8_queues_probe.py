#!/usr/bin/env python3
"""Do W in-process virtual ranks (one stream each, all on device 0) of the direct all-reduce meet on the device with the default
number of hardware queues, or does the HIP runtime have to be given one queue per stream (GPU_MAX_HW_QUEUES)?
usage: [GPU_MAX_HW_QUEUES=16] python tools/probes/world8_queues_probe.py [world]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
os.environ.setdefault("ATOMA_XGMI_TIMEOUT_MS", "3000")
import atoma_hip as ah  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ah.set_device(0)
xs = []
for r in range(world):
    h = C.c_void_p()
    assert ah.lib.atoma_xgmi_create(C.byref(h), r, world, 0, 1 << 20) == 0, ah.last_error()
    xs.append(h)
blobs = (C.c_uint8 * (128 * world))()
for r in range(world):
    one = (C.c_uint8 * 128)()
    assert ah.lib.atoma_xgmi_handle(xs[r], one) == 0
    C.memmove(C.addressof(blobs) + 128 * r, one, 128)
for r in range(world):
    assert ah.lib.atoma_xgmi_connect(xs[r], blobs) == 0, ah.last_error()
streams = [ah.Stream() for _ in range(world)]
count = 64 * 8192
bufs = [ah.DeviceBuffer.from_numpy(np.full(count, 0x3F80, np.uint16)) for _ in range(world)]
t0 = time.perf_counter()
for r in range(world):
    assert ah.lib.atoma_xgmi_allreduce_sum(xs[r], bufs[r].ptr, bufs[r].ptr, count, 1, streams[r].s) == 0, ah.last_error()
for r in range(world):
    streams[r].synchronize()
dt = time.perf_counter() - t0
st = [ah.lib.atoma_xgmi_status(x) for x in xs]
ok = all((b.numpy(np.uint16, (count,)) == 0x4100).all() for b in bufs) if world == 8 else None
print(f"world {world} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} elapsed {dt * 1e3:.1f} ms status {st} sum_is_8 {ok}")
