mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -15) > gpurun_out/r03_t6.log
(timeout 600 python tools/probes/linear64_ab.py 2>&1) > gpurun_out/r03_l64_ab3.jsonl
(timeout 600 python tools/rank_step.py --layers 80 --iters 10 2>&1 | tail -1) > gpurun_out/r03_rank_tile3.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank_tile3 -o step -- python $GRAFT_REPO_ROOT/tools/rank_step.py --layers 8 --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_rank_tile3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $(find gpurun_out/prof_rank_tile3 -name "*kernel_trace.csv" | head -1) > gpurun_out/r03_rank_tile3_breakdown.json
cat gpurun_out/r03_t6.log; cat gpurun_out/r03_rank_tile3.json
