#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05r
timeout 600 python tools/bench_extra.py c5_phi3_mini_decode_step 2>&1 | tail -3 | tee gpurun_out/r05r/phi3_step.json
