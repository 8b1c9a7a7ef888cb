#!/bin/bash
# One GPU-box pass: gpu tests, bench (with its in-run extras and PMC traffic), rocprofv3 kernel-trace stats, PMC passes,
# per-kernel measurements, the C3-lite trace and the 70B step.  Run through gpurun:
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02'
# then: python tools/summarize_profiles.py r02   (copies the summaries into profiles/)
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $OUT/pytest_gpu_$TAG.log
(timeout 600 python bench.py 2>&1 | tail -2) > $OUT/bench_$TAG.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $OUT/smoke_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o decode -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $OUT/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o decode -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/pmc_write_$TAG.log 2>&1
ATOMA_BENCH_STEP_CASES=256r timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_step_$TAG -o step -- python $REPO/tools/bench_kernels.py step > $OUT/prof_step_$TAG.log 2>&1
ATOMA_BENCH_STEP_CASES=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_step1_$TAG -o step -- python $REPO/tools/bench_kernels.py step > $OUT/prof_step1_$TAG.log 2>&1
# the 70B TP = 8 rank step (configs[3] without its all-reduces): whole step, and a kernel trace of 8 layers for the per-kernel breakdown
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_rank_$TAG -o step -- python $REPO/tools/rank_step.py --layers 8 --iters 3 > $OUT/prof_rank_$TAG.log 2>&1
cd $REPO
(timeout 400 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1) > $OUT/rank_step_$TAG.json
(timeout 1500 python tools/bench_kernels.py decode decode_fp8 prefill prefill_paged cache norm sampling linear linear_mid linear_big graph step swap prep 2>&1) > $OUT/kernels_$TAG.jsonl
(timeout 400 python tools/engine_trace.py 2>&1 | tail -1) > $OUT/trace_$TAG.json
(timeout 400 python tools/engine_trace.py --model 70b-tp8-shard --requests 64 --prompt 4096 --decode-steps 256 2>&1 | tail -1) > $OUT/trace_70b_tp8_rank_$TAG.json
(timeout 400 python tools/tp_step.py --steps 10 2>&1 | tail -1) > $OUT/tp_step_n1_$TAG.json
# configs[3] executed as 8 ranks on this one device (round 5): the 80-layer decode step and the 4096-token prefill chunk, direct all-reduce among 8
(timeout 600 python tools/tp_step.py --virtual-ranks 8 --steps 5 2>&1 | tail -1) > $OUT/tp_step_8_virtual_ranks_$TAG.json
# (the prefill chunk: 8 ranks x 2 layers and 4 ranks x 16 layers -- eight ranks' vendor GEMMs beside spinning all-reduces stall on ONE device from 4 layers on, profiles/r05_tp_prefill_virtual_ranks_limits.txt)
(ATOMA_XGMI_TIMEOUT_MS=10000 timeout 300 python tools/tp_step.py --virtual-ranks 8 --prefill 4096 --layers 2 --steps 2 --check-unsharded 2>&1 | tail -1) > $OUT/tp_prefill_8_virtual_ranks_$TAG.json
(timeout 300 python tools/tp_step.py --virtual-ranks 8 --layers 8 --steps 3 --check-unsharded 2>&1 | tail -1) > $OUT/tp_step_8_ranks_vs_unsharded_$TAG.json
find $OUT/prof_$TAG -name "*stats*" | head; tail -3 $OUT/pytest_gpu_$TAG.log; cat $OUT/smoke_$TAG.log; cat $OUT/bench_$TAG.log | cut -c1-600
