#!/bin/bash
# The short GPU-box pass: gpu tests, the bench line, kernel-trace stats and the two HBM traffic counter passes of the headline.
#   gpurun --timeout 1800 -- 'bash tools/gpu_round_lite.sh r04'
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $OUT/pytest_gpu_$TAG.log
(timeout 600 python bench.py 2>&1 | tail -2) > $OUT/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o decode -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $OUT/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_write_$TAG.log 2>&1
cd $REPO
tail -3 $OUT/pytest_gpu_$TAG.log; cat $OUT/bench_$TAG.log | cut -c1-800
