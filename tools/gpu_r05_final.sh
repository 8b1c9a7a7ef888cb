#!/bin/bash
# round 5, final confirmation of the binary at HEAD: the whole gpu suite, the bench line, smoke, kernel-trace stats of the bench command, the generic-kernel probes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
(timeout 600 python bench.py 2>&1 | tail -2) > $O/bench.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o decode -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $O/prof.log 2>&1)
(timeout 200 python tools/probes/generic_prefill_ab.py 2>&1) > $O/generic_prefill_ab.json
for shape in "d=96 B=64" "d=96 B=8" "d=96 (Phi" "d=256 B=256"; do ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 150 python tools/bench_kernels.py decode 2>&1 | grep workload; done > $O/generic_decode.jsonl
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
tail -3 $O/pytest_gpu.log; cat $O/smoke.log; cut -c1-400 $O/bench.log; head -2 $O/kernel_stats.csv | cut -c1-300; cat $O/generic_decode.jsonl | cut -c1-200
