#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-call tools/gpu_r05_[a-s].sh of round 5).  Run through the pool's client, e.g.
#   gpurun --timeout 1800 -- 'bash tools/gpu_call.sh r06 tests bench stats pmc'
# usage: tools/gpu_call.sh TAG STEP [STEP ...]        outputs under gpurun_out/TAG/ (copy what should be judged into profiles/)
#   tests[:K_EXPR]     pytest -m gpu (optionally -k K_EXPR)
#   bench              the driver's line: python bench.py                      -> bench.json
#   stats              rocprofv3 --kernel-trace --stats of the headline        -> kernel_stats/
#   pmc                FETCH_SIZE / WRITE_SIZE passes of the headline          -> pmc_fetch/ pmc_write/
#   step[:own|vendor]  per-kernel breakdown of the configs[2] decode step      -> step_breakdown_b256_<which>.json
#   rank               per-kernel breakdown of the configs[3] rank step        -> rank_step_breakdown.json
#   lin256[:SHAPES]    tools/probes/linear256_ab.py (own 256-row projections vs the vendor GEMM)
#   c3ab               tools/probes/c3_own_vs_vendor.py
#   prefill_pmc        tools/prefill_pmc.sh (matrix-pipe counters of the prefill kernel)
TAG=${1:?tag}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$REPO" || exit 1
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "== $step"
  case $name in
    tests) (timeout 1700 python -m pytest tests -m gpu -q --tb=short ${arg:+-k "$arg"} 2>&1 | tail -40) > "$O/pytest_gpu.log"; tail -3 "$O/pytest_gpu.log" ;;
    bench) timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; tail -c 600 "$O/bench.json" ;;
    stats) (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kernel_stats" -o decode -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > "$O/stats.log" 2>&1) ;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmc_$c" -o decode -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > "$O/pmc_$c.log" 2>&1); done ;;
    step) which=${arg:-own}; mb=256; [[ $which == vendor ]] && mb=128
          (cd /tmp && ATOMA_STEP_FUSED_MAX_BATCH=$mb ATOMA_BENCH_STEP_CASES=256r timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$O/prof_step" -o step -- python "$REPO/tools/bench_kernels.py" step > "$O/prof_step.log" 2>&1)
          f=$(find "$O/prof_step" -name "*kernel_trace.csv" | head -1)
          python tools/step_breakdown.py "$f" "Llama-3.1-8B decode step, batch 256, contexts U[2048,2560), $TAG, $which projections" > "$O/step_breakdown_b256_$which.json"; rm -rf "$O/prof_step"; head -40 "$O/step_breakdown_b256_$which.json" ;;
    rank) (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$O/prof_rank" -o rank -- python "$REPO/tools/rank_step.py" --layers 8 > "$O/prof_rank.log" 2>&1)
          f=$(find "$O/prof_rank" -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py "$f" "one rank of the 70B TP=8 step, 8 layers, $TAG" > "$O/rank_step_breakdown.json"; rm -rf "$O/prof_rank" ;;
    lin256) timeout 600 python tools/probes/linear256_ab.py ${arg//,/ } > "$O/linear256_ab.jsonl" 2>&1; cut -c1-160 "$O/linear256_ab.jsonl" ;;
    c3ab) timeout 600 python tools/probes/c3_own_vs_vendor.py 3 2>&1 | tail -1 | tee "$O/c3_own_vs_vendor.json" ;;
    prefill_pmc) bash tools/prefill_pmc.sh "$TAG" ;;
    *) echo "unknown step $step" ;;
  esac
done
