#!/bin/bash
# uniform batches through the balanced line (option decode_stream = 2) against the one-wavefront-per-(sequence, head) order
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for st in 3 2; do
  echo "== stream=$st"
  ATOMA_DECODE_STREAM=$st python tools/bench_kernels.py decode 2>&1 | cut -c1-140
done
done
