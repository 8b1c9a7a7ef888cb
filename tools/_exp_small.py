import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_kernels as bk
bk.ah.set_device(0)
for wpc in (0, 2, 4, 8, 16):
    for mt in (4, 8, 16, 32):
        bk.ah.lib.atoma_set_option(b"decode_waves_per_cu", wpc); bk.ah.lib.atoma_set_option(b"decode_min_tiles", mt)
        bk.decode_case(f"wpc={wpc} min_tiles={mt} 70B-TP8 shard B=64 h=8 hk=1", 64, 4096, 8, 1)
        bk.decode_case(f"wpc={wpc} min_tiles={mt} 8B-TP8 shard B=256 h=4 hk=1", 256, 4096, 4, 1)
        bk.decode_case(f"wpc={wpc} min_tiles={mt} B=1", 1, 4096, 32, 8)
