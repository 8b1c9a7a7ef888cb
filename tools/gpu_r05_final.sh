#!/bin/bash
# round 5, final confirmation of the binary at HEAD: the whole gpu suite, the bench line, smoke, kernel-trace stats of the bench command, the per-kernel lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
(timeout 600 python bench.py 2>&1 | tail -2) > $O/bench.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o decode -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $O/prof.log 2>&1)
(timeout 900 python tools/bench_kernels.py decode prefill prefill_paged 2>&1) > $O/kernels.jsonl
(timeout 300 python tools/bench_extra.py c5_phi3_mini_decode_step 2>&1 | tail -1) > $O/phi3_step.json
(timeout 200 python tools/probes/generic_prefill_ab.py 2>&1) > $O/generic_prefill_ab.json
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
tail -3 $O/pytest_gpu.log; cat $O/smoke.log; cut -c1-700 $O/bench.log; head -3 $O/kernel_stats.csv | cut -c1-300; cat $O/phi3_step.json | cut -c1-400
