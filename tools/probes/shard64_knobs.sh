#!/bin/bash
# the rank-step attention (70B TP = 8 shard, batch 64, context 4096) under the split / merge knobs
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ATOMA_BENCH_DECODE_SHAPE="B=64"
for rep in 1 2; do
for v in "X=0" "ATOMA_DECODE_WAVES_PER_CU=4" "ATOMA_DECODE_WAVES_PER_CU=8" "ATOMA_DECODE_WAVES_PER_CU=12" "ATOMA_DECODE_WAVES_PER_CU=8 ATOMA_DECODE_MIN_TILES=4" "ATOMA_DECODE_MQK_P8=2" "ATOMA_DECODE_MQK_P8=2 ATOMA_DECODE_WAVES_PER_CU=12" "ATOMA_DECODE_WG_MERGE=0" "ATOMA_DECODE_WG_MERGE=2" "ATOMA_DECODE_WG_MERGE=2 ATOMA_DECODE_WAVES_PER_CU=8" "ATOMA_DECODE_NT=0"; do
  echo -n "$v   "; env $v python tools/bench_kernels.py decode | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms'], d['frac_hbm'])"
done
done
