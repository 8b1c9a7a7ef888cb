#!/bin/bash
# round 5, call p: the tiled generic prefill kernels -- parity, then the tile-size x LDS-buffer probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05p
timeout 600 python -m pytest tests/test_attention_golden_gpu.py tests/test_host_ops_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05p/pytest.txt
timeout 400 python tools/probes/generic_prefill_ab.py > gpurun_out/r05p/generic_prefill_ab.json 2> gpurun_out/r05p/ab.err
cat gpurun_out/r05p/pytest.txt; cat gpurun_out/r05p/generic_prefill_ab.json; tail -5 gpurun_out/r05p/ab.err
