"""What does a cut sequence cost on the balanced decode line?  (needs `make -C atoma-infer_amd cutprobe`; results of levels 2 / 3 are wrong)
    ATOMA_HIP_LIB=tools/probes/libatoma_hip_cutprobe.so python tools/probes/cut_probe.py"""
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
jit = (3072 + rng.integers(-16, 17, B)).astype(np.int32)
for rep in range(2):
    for lm, what in ((1, "merged in the launch"), (0, "combine kernel"), (2, "PROBE: nobody merges"), (3, "PROBE: cut pieces not even stored"), (4, "PROBE: cut pieces stored write-back, nobody merges"), (5, "PROBE: only the first piece of a wavefront stored"), (6, "PROBE: only the last piece stored")):
        ah.lib.atoma_set_option(b"decode_line_merge", lm)
        BK.decode_case(f"3072 +- 16, {what}", B, 4096, 32, 8, lens=jit)
    ah.lib.atoma_set_option(b"decode_line_merge", 1)
    BK.decode_case("uniform 3072 (no cut)", B, 4096, 32, 8, lens=np.full(B, 3072, np.int32))
