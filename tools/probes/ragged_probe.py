"""Where do ragged batches lose against uniform ones on the balanced decode line?  B = 256, 32 / 8 heads, d = 128, page 16.
    python tools/probes/ragged_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    sys.path.insert(0, p)
import numpy as np
import atoma_hip as ah
import bench_kernels as BK

ah.set_device(0)
rng = np.random.default_rng(3)
B = 256
cases = {
    "uniform 4096": np.full(B, 4096),
    "uniform 3072": np.full(B, 3072),
    "uniform 3080 (not a multiple of 16)": np.full(B, 3080),
    "two lengths 2048 / 4096 alternating": np.where(np.arange(B) % 2 == 0, 2048, 4096),
    "3072 +- 16 (one tile of jitter)": 3072 + rng.integers(-16, 17, B),
    "3072 +- 256": 3072 + rng.integers(-256, 257, B),
    "U[2048, 4096]": rng.integers(2048, 4097, B),
    "U[2048, 4096] sorted": np.sort(rng.integers(2048, 4097, B)),
    "U[2048, 4096] rounded to 16": rng.integers(128, 257, B) * 16,
}
for opt, val in ((b"decode_stream", 1), (b"decode_stream", 0)):
    ah.lib.atoma_set_option(opt, val)
    for name, lens in cases.items():
        BK.decode_case(f"[decode_stream={val}] {name}", B, 4096, 32, 8, lens=lens.astype(np.int32))
