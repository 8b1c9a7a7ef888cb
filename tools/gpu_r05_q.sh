#!/bin/bash
# round 5, call q: the second generic decode kernel -- parity, then first vs second version on the d = 96 / d = 256 shapes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_decode_gpu.py -x -q -m gpu -k "other_head_sizes" 2>&1 | tail -8 | tee gpurun_out/r05q/pytest.txt
for v in new 1 0 new 1; do
  for shape in "d=96" "d=256"; do
    if [ $v = new ]; then unset ATOMA_GENERIC_DECODE_STREAM; else export ATOMA_GENERIC_DECODE_STREAM=$v; fi
    ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 150 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-260 | sed "s/^/[$v] /"
  done
done | tee gpurun_out/r05q/generic_decode_ab.txt
