mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -15) > gpurun_out/r03_t2.log
(timeout 600 python tools/bench_kernels.py linear_mid 2>&1 | grep -v "8B\|batch=32") > gpurun_out/r03_ks_on.jsonl
(ATOMA_LINEAR_KS_DW=8 timeout 600 python tools/bench_kernels.py linear_mid 2>&1 | grep -v "8B\|batch=32\|vendor") > gpurun_out/r03_ks_dw8.jsonl
(ATOMA_LINEAR_KS_RT=1 timeout 600 python tools/bench_kernels.py linear_mid 2>&1 | grep -v "8B\|batch=32\|vendor") > gpurun_out/r03_ks_rt1.jsonl
(ATOMA_LINEAR_KS_RT=2 timeout 600 python tools/bench_kernels.py linear_mid 2>&1 | grep -v "8B\|batch=32\|vendor") > gpurun_out/r03_ks_rt2.jsonl
(timeout 600 python tools/rank_step.py --layers 80 --iters 10 2>&1 | tail -1) > gpurun_out/r03_rank_ks.json
cat gpurun_out/r03_t2.log; cat gpurun_out/r03_rank_ks.json
