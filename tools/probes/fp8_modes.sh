#!/bin/bash
# fp8 decode: uniform / ragged x balanced line forced or not x K fetch format  ->  gpurun_out/fp8modes.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for st in 3 2; do for kl in 0 1 2; do
  echo "== stream=$st klines=$kl"
  ATOMA_DECODE_STREAM=$st ATOMA_DECODE_FP8_KLINES=$kl python tools/bench_kernels.py decode_fp8 2>&1 | cut -c1-140
done; done
done
