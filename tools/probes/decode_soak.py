"""More seeds of tests/test_decode_gpu.py::test_decode_random_shapes_through_the_default_dispatch (a soak run for the last-arriver merges
and the dispatcher): python tools/probes/decode_soak.py [first_seed] [count]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import atoma_hip as ah
import test_decode_gpu as T

ah.set_device(0)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
kernels = collections.Counter()
for seed in range(first, first + count):
    T.test_decode_random_shapes_through_the_default_dispatch(ah, seed)
    kernels[ah.lib.atoma_last_decode_kernel().decode().split("<")[0] + " / " + ah.lib.atoma_last_decode_kernel().decode().split(",")[-1]] += 1
print(f"{count} seeds from {first}: all within tolerance, twice to the bit")
for k, n in kernels.most_common():
    print(f"  {n:3d}  {k}")
