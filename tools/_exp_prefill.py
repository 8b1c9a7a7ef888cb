import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_kernels as bk
bk.ah.set_device(0)
bk.bench_prefill()
