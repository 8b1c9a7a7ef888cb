#!/bin/bash
# the headline workload (bench.py) under the decode knobs, interleaved, 2 repetitions
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in ${KNOBS:-"X=0" "ATOMA_DECODE_P=2" "ATOMA_DECODE_NT=0" "ATOMA_DECODE_STREAM_WAVES_PER_CU=6" "ATOMA_DECODE_STREAM_WAVES_PER_CU=10" "ATOMA_DECODE_STREAM_WAVES_PER_CU=12" "ATOMA_DECODE_STREAM_WAVES_PER_CU=16" "ATOMA_DECODE_MQK=5" "ATOMA_DECODE_STREAM=3"}; do
  echo -n "$v  "; env $v python bench.py --no-cpu-baseline --no-extra --no-traffic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))"
done
done
