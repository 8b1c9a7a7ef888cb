mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_full_size_gpu.py tests/test_prefill_gpu.py tests/test_bench_multirank_gpu.py tests/test_bench_contract.py tests/test_tp_step_gpu.py -m gpu -q --tb=short 2>&1 | tail -40) > gpurun_out/r03_t10.log
(timeout 900 python bench.py 2>gpurun_out/r03_bench_try1.err | tail -1) > gpurun_out/r03_bench_try1.json
tail -30 gpurun_out/r03_t10.log; cut -c1-1500 gpurun_out/r03_bench_try1.json
