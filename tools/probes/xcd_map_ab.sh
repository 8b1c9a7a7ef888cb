#!/bin/bash
# workgroup -> (sequence, kv head) order: all kv heads of a sequence on one XCD (decode_xcd_map = 1) against kv head fastest (0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for xm in 1 0; do
  echo "== xcd_map=$xm"
  ATOMA_DECODE_XCD_MAP=$xm python tools/bench_kernels.py decode_fp8 2>&1 | cut -c1-140
  ATOMA_DECODE_XCD_MAP=$xm python tools/bench_kernels.py decode 2>&1 | cut -c1-140
done
done
