#!/bin/bash
# fp8 split-KV launches: wavefronts per workgroup x wavefronts per CU (the matrix-core kernel keeps 3 per SIMD resident)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in "B=16" "shard"; do
export ATOMA_FP8_SHAPE="$shape"
for v in "ATOMA_DECODE_FP8_WG=1 ATOMA_DECODE_WAVES_PER_CU=8" "ATOMA_DECODE_FP8_WG=0 ATOMA_DECODE_WAVES_PER_CU=8" "ATOMA_DECODE_FP8_WG=1 ATOMA_DECODE_WAVES_PER_CU=12" "ATOMA_DECODE_FP8_WG=0 ATOMA_DECODE_WAVES_PER_CU=12" "ATOMA_DECODE_FP8_WG=0 ATOMA_DECODE_WAVES_PER_CU=16" "ATOMA_DECODE_FP8_WG=0 ATOMA_DECODE_WAVES_PER_CU=12 ATOMA_DECODE_FP8_KLINES=0"; do
  echo "== $v"; env $v python tools/bench_kernels.py decode_fp8 | cut -c1-130; env $v python tools/bench_kernels.py decode_fp8 | cut -c1-130
done
done
