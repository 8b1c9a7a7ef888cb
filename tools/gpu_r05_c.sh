#!/bin/bash
# round 5, GPU call C: the whole GPU suite on the epoch-ticket build, the 2-layer TP prefill probe, bench.py with the new extras (timed)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05c; mkdir -p $O
echo "== prefill debug, 2 layers"; timeout 600 python tools/probes/tp_prefill_debug.py 4096 2 2>&1 | tail -25 | tee $O/tp_prefill_debug_2layers.txt
echo "== all gpu tests"; ( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 ) 2>&1 | tail -40 | tee $O/tests.txt
echo "== bench.py"; ( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05c/bench.json') if x.startswith('{')]
d=json.loads(l[-1])
print('value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline'])
print('verified',d.get('verified'))
for k,v in d['extra'].items(): print(k, json.dumps(v)[:400])
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
PY
