#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_decode_step_gpu.py -x -q -m gpu -k "op_by_op" 2>&1 | tail -25 | tee gpurun_out/r05r/pytest.txt
