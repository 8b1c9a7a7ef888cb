(timeout 900 python -m pytest tests/test_row_stats_gpu.py tests/test_linear_gpu.py tests/test_decode_step_gpu.py tests/test_tp_step_gpu.py -q -x --tb=short 2>&1 | tail -25) > gpurun_out/row_stats_tests.log
for rep in 1 2; do
for cfg in "ATOMA_STEP_NORM_STATS=1 ATOMA_LINEAR_TILE_MAX_SPLITS=8" "ATOMA_STEP_NORM_STATS=0 ATOMA_LINEAR_TILE_MAX_SPLITS=8" "ATOMA_STEP_NORM_STATS=0 ATOMA_LINEAR_TILE_MAX_SPLITS=4"; do
  echo "== $cfg"; env $cfg timeout 400 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1 | cut -c1-260
done; done > gpurun_out/rank_ab.txt 2>&1
cat gpurun_out/row_stats_tests.log; cat gpurun_out/rank_ab.txt
