#!/bin/bash
# One GPU-box pass: gpu tests, bench, rocprofv3 kernel-trace stats, PMC passes.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01'
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150) > $OUT/pytest_gpu.log
(timeout 300 python bench.py --steps 50 --warmup 5 2>&1 | tail -3) > $OUT/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o decode -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_write_$TAG.log 2>&1
cd $REPO
find $OUT/prof_$TAG -name "*stats*" | head; ls -R $OUT | head -50
