REPO=$(pwd); OUT=$REPO/gpurun_out; cd /tmp && export TMPDIR=/tmp
for ns in 1 0; do
ATOMA_STEP_NORM_STATS=$ns timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_rank_ns$ns -o step -- python $REPO/tools/rank_step.py --layers 8 --iters 3 > $OUT/prof_rank_ns$ns.log 2>&1
f=$(find $OUT/prof_rank_ns$ns -name "*kernel_trace.csv" | head -1)
python $REPO/tools/step_breakdown.py $f > $OUT/rank_breakdown_ns$ns.json 2>&1
done
cd $REPO; python - <<'PY'
import json
for ns in (1,0):
    d=json.load(open(f"gpurun_out/rank_breakdown_ns{ns}.json"))
    print("== norm_stats",ns, d.get("kernel_busy_ms"))
    for k in d["kernels"][:12]: print(f'  {k["avg_us"]:7.1f} us x{k["launches"]:3d}  {k["kernel"][:110]}')
PY
