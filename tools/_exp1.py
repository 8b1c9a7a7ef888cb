import sys; sys.path.insert(0,"tools"); import bench_kernels as bk; bk.ah.set_device(0)
bk.decode_case("uniform", 256, 4096, 32, 8)
