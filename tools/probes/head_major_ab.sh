#!/bin/bash
# workgroup order of the non-balanced launches (split-KV, small batches): kv head slowest (decode_head_major = 1) against fastest (0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for hm in 1 0; do
  echo "== head_major=$hm"
  ATOMA_DECODE_HEAD_MAJOR=$hm python tools/bench_kernels.py decode 2>&1 | cut -c1-140
  ATOMA_DECODE_HEAD_MAJOR=$hm python tools/bench_kernels.py decode_fp8 2>&1 | cut -c1-140
  ATOMA_DECODE_FP8_WG=0 ATOMA_DECODE_HEAD_MAJOR=$hm python tools/bench_kernels.py decode_fp8 2>&1 | sed 's/"workload": "/"workload": "[one wavefront per workgroup] /' | cut -c1-160
done
done
