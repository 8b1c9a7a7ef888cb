for pw in 0 100 0 100 300; do echo "== prewarm $pw"; python bench.py --steps 50 --warmup 5 --prewarm-ms $pw --no-cpu-baseline --no-extra --no-traffic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"; done
python - <<'PY'
import sys, json
sys.path.insert(0,'tools')
import bench_extra as BE
print(json.dumps({k:v for k,v in BE.prefill().items() if k!='sample'}))
PY
timeout 300 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1 | cut -c100-260
