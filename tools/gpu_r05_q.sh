#!/bin/bash
# round 5, call q: the generic decode kernel -- parity, then 1 / 2 / 4 wavefronts per unit on the small-batch d = 96 shapes, and the large shapes once more
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_decode_gpu.py -x -q -m gpu -k "other_head_sizes" 2>&1 | tail -8 | tee gpurun_out/r05q/pytest.txt
for v in auto 4 8 2 auto 8; do
  for shape in "d=96 B=64" "d=96 B=8" "d=96 (Phi" "d=256 B=256"; do
    unset ATOMA_GENERIC_DECODE_WAVES
    [ $v != auto ] && export ATOMA_GENERIC_DECODE_WAVES=$v
    ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 150 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-260 | sed "s/^/[waves=$v] /"
  done
done | tee gpurun_out/r05q/generic_decode_waves_ab.txt
