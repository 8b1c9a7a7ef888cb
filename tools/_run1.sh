mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_sampling_gpu.py tests/test_kv_format_gpu.py tests/test_allreduce_xgmi_gpu.py tests/test_cache_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -15) > gpurun_out/r03_t1.log
(timeout 600 python tools/rank_step.py --layers 80 --iters 10 2>&1 | tail -1) > gpurun_out/r03_rank_base.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank_base -o step -- python $GRAFT_REPO_ROOT/tools/rank_step.py --layers 8 --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_rank_base.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_rank_base -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $f > gpurun_out/r03_rank_base_breakdown.json
cat gpurun_out/r03_t1.log; cat gpurun_out/r03_rank_base.json
