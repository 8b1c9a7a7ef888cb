mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py tests/test_graph_capture_gpu.py tests/test_norm_rope_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -15) > gpurun_out/r03_t5.log
(timeout 600 python tools/rank_step.py --layers 80 --iters 10 2>&1 | tail -1) > gpurun_out/r03_rank_tile2.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank_tile2 -o step -- python $GRAFT_REPO_ROOT/tools/rank_step.py --layers 8 --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_rank_tile2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $(find gpurun_out/prof_rank_tile2 -name "*kernel_trace.csv" | head -1) > gpurun_out/r03_rank_tile2_breakdown.json
cat gpurun_out/r03_t5.log; cat gpurun_out/r03_rank_tile2.json
