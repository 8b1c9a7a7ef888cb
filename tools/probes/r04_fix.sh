(timeout 600 python tools/bench_kernels.py prefill prefill_paged 2>&1) > gpurun_out/kernels_prefill_r04.jsonl
(timeout 900 python -m pytest tests/test_host_ops_gpu.py tests/test_decode_gpu.py -q --tb=short -k "other_forms or random_shapes or wrappers" 2>&1 | tail -25) > gpurun_out/new_tests.log
cat gpurun_out/kernels_prefill_r04.jsonl | cut -c1-200; cat gpurun_out/new_tests.log
