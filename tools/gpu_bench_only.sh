TAG=r03
REPO=$(pwd)
OUT=$REPO/gpurun_out
(timeout 600 python bench.py 2>&1 | tail -2) > $OUT/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o decode -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $OUT/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_write_$TAG.log 2>&1
cat $OUT/bench_$TAG.log | cut -c1-300
