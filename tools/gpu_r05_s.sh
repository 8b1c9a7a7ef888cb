#!/bin/bash
# round 5, call s: the option plumbing of attn_generic.hip (tests switch kernels with atoma_set_option) + the probes that use it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05s
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_attention_golden_gpu.py tests/test_host_ops_gpu.py tests/test_decode_step_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05s/pytest.txt
timeout 300 python tools/probes/generic_prefill_ab.py > gpurun_out/r05s/generic_prefill_ab.json 2> gpurun_out/r05s/ab.err; tail -3 gpurun_out/r05s/ab.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05s/generic_prefill_ab.json"))
for k, v in d.items():
    print(k, {n: e["ms"] for n, e in v.items() if isinstance(e, dict)})
PY
