(timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_decode_dispatch_gpu.py tests/test_host_ops_gpu.py tests/test_kv_fp8_gpu.py tests/test_decode_step_gpu.py -q --tb=short 2>&1 | tail -25) > gpurun_out/pair64_tests.log
cat gpurun_out/pair64_tests.log
