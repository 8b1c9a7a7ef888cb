import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk
bk.ah.set_device(0)
for ct in (0, 4, 8, 16, 32, 64):
    bk.ah.lib.atoma_set_option(b"decode_chunk_tiles", ct)
    print("chunk_tiles", ct, flush=True)
    bk.decode_case("  ragged U[2048,4096] GQA", 256, 4096, 32, 8, ragged=True)
    bk.decode_case("  ragged U[2048,4096] MHA", 256, 4096, 32, 32, ragged=True)
    bk.decode_case("  uniform GQA", 256, 4096, 32, 8)
