"""Decode bandwidth against the number of active wavefronts: uniform S=4096 batches, KV splitting switched off
(one wavefront per (sequence, kv head)), B = 16..512.  Input to the ragged-batch model in DESIGN.md 4.1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk  # noqa: E402

ah = bk.ah
ah.lib.atoma_set_option(b"decode_min_tiles", 1 << 20)   # never split
for B in (16, 32, 64, 96, 128, 160, 192, 256, 384, 512):
    bk.decode_case(f"curve unsplit B={B} S=4096 waves={8 * B}", B, 4096, 32, 8)
for S in (2048, 3072):
    bk.decode_case(f"curve unsplit B=256 S={S}", 256, S, 32, 8)
bk.decode_case("curve ragged U[2048,4096] B=256", 256, 4096, 32, 8, ragged=True)
bk.decode_case("curve ragged U[2048,4096] B=128", 128, 4096, 32, 8, ragged=True)
ah.lib.atoma_set_option(b"decode_min_tiles", 8)
bk.decode_case("curve split-allowed B=128 S=4096", 128, 4096, 32, 8)
bk.decode_case("curve split-allowed ragged B=128", 128, 4096, 32, 8, ragged=True)
