mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -15) > gpurun_out/r03_t4.log
(timeout 600 python tools/probes/ks_ab.py 2>&1) > gpurun_out/r03_ks_ab2.jsonl
(timeout 600 python tools/rank_step.py --layers 80 --iters 10 2>&1 | tail -1) > gpurun_out/r03_rank_tile.json
cat gpurun_out/r03_t4.log; cat gpurun_out/r03_rank_tile.json
