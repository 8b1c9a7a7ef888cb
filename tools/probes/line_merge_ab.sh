(timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_kv_fp8_gpu.py tests/test_linear_gpu.py tests/test_decode_step_gpu.py -q -x --tb=short 2>&1 | tail -15) > gpurun_out/line_merge_tests.log
for m in 1 0 1 0; do echo "== line_merge $m"; ATOMA_DECODE_LINE_MERGE=$m timeout 600 python tools/bench_kernels.py decode 2>&1 | grep "C2a\|ragged\|S=1024\|d=64"; done > gpurun_out/line_merge_ab.txt 2>&1
(timeout 400 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1) >> gpurun_out/line_merge_ab.txt
(ATOMA_LINEAR_TILE_SPLITS=4 timeout 400 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1) >> gpurun_out/line_merge_ab.txt
cat gpurun_out/line_merge_tests.log; cat gpurun_out/line_merge_ab.txt | cut -c1-260
