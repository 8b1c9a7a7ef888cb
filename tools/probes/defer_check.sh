(timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_kv_fp8_gpu.py tests/test_graph_capture_gpu.py tests/test_decode_step_gpu.py -q --tb=short 2>&1 | tail -6) > gpurun_out/defer_tests.log
cat gpurun_out/defer_tests.log
ATOMA_HIP_LIB=tools/probes/libatoma_hip_cutprobe.so timeout 600 python tools/probes/cut_probe.py 2>&1 | grep workload | cut -c14-80,80-96 > gpurun_out/cut_probe2.txt; cat gpurun_out/cut_probe2.txt
timeout 300 python tools/probes/ragged_probe.py 2>&1 | grep "stream=1" | cut -c14-75,76-100 
ATOMA_BENCH_DECODE_SHAPE=ragged timeout 300 python tools/bench_kernels.py decode decode_fp8 2>&1 | grep ms | cut -c14-120
