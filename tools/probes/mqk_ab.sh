#!/bin/bash
# decode_mqk = 5 (matrix-core scores for groups above 4 and tiny batches) against 7 (every group size), all decode shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in 5 7; do
  echo "== mqk=$v"
  ATOMA_DECODE_MQK=$v python tools/bench_kernels.py decode 2>&1 | cut -c1-140
done
done
