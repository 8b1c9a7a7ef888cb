#!/bin/bash
# decode_mqk variants (bit 0: groups above 4, bit 1: all smaller groups, bit 2: tiny batches, bit 3: groups of 2..4 on the line), all decode shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in ${MQK_VALUES:-29 13}; do
  echo "== mqk=$v"
  ATOMA_DECODE_MQK=$v python tools/bench_kernels.py decode 2>&1 | cut -c1-140
done
done
