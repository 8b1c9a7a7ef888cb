mkdir -p gpurun_out
for rot in 0 16 8; do
(ATOMA_LINEAR_KS_ROT=$rot timeout 600 python tools/bench_kernels.py linear_mid 2>&1 | grep -v "8B\|batch=32\|vendor") > gpurun_out/r03_ks_rot$rot.jsonl
done
(timeout 300 python -m pytest tests/test_linear_gpu.py -m gpu -q -x --tb=short -k "mid_batch or any_batch" 2>&1 | tail -5) > gpurun_out/r03_t3.log
cat gpurun_out/r03_t3.log
