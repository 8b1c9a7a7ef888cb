import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk
bk.ah.set_device(0)
for mq in (0, 3, 0, 3):
    bk.ah.lib.atoma_set_option(b"decode_mqk", mq)
    bk.decode_case(f"mqk={mq} 70B TP1 B=256 h=64 hk=8 S=4096", 256, 4096, 64, 8)
    bk.decode_case(f"mqk={mq} 70B TP8 shard B=64 h=8 hk=1 S=4096", 64, 4096, 8, 1)
    bk.decode_case(f"mqk={mq} 70B TP8 shard B=256 h=8 hk=1 S=4096", 256, 4096, 8, 1)
    bk.decode_case(f"mqk={mq} 8B C2a B=256 h=32 hk=8 S=4096", 256, 4096, 32, 8)
    bk.decode_case(f"mqk={mq} 8B ragged", 256, 4096, 32, 8, ragged=True)
    bk.decode_case(f"mqk={mq} MHA hk=32", 256, 4096, 32, 32)
    bk.decode_case(f"mqk={mq} B=1 S=4096 8B", 1, 4096, 32, 8)
