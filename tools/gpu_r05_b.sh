#!/bin/bash
# round 5, GPU call B: epoch tickets (tests + headline against the kernarg-fix binary), the TP prefill discrepancy, rank-step / ragged timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05b; mkdir -p $O
echo "== ticket + tp tests"; timeout 900 python -m pytest tests/test_sync_ticket_gpu.py tests/test_tp_step_gpu.py tests/test_graph_capture_gpu.py tests/test_decode_gpu.py tests/test_linear_gpu.py tests/test_prefill_gpu.py tests/test_cache_gpu.py -q -m gpu -x 2>&1 | tail -30 | tee $O/tests.txt
echo "== prefill debug"; timeout 600 python tools/probes/tp_prefill_debug.py 4096 1 2>&1 | tail -25 | tee $O/tp_prefill_debug.txt
echo "== binary A/B: epoch tickets vs kernarg fix"; timeout 900 python tools/ab_binary.py --rounds 3 --libs kernargfix=tools/probes/bisect/libatoma_hip_kernargfix.so epoch=atoma-infer_amd/lib/libatoma_hip.so > $O/epoch_ab_headline.json 2>$O/epoch_ab.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/epoch_ab_headline.json'))
for k,v in d['libs'].items(): print(k, v.get('kernel_ms'), v.get('median_kernel_ms'), v.get('error','')[:300])
PY
echo "== rank step, both binaries"; for lib in tools/probes/bisect/libatoma_hip_kernargfix.so atoma-infer_amd/lib/libatoma_hip.so; do for i in 1 2; do ATOMA_HIP_LIB=$lib timeout 600 python tools/rank_step.py 2>&1 | tail -1 | cut -c1-400; done; done | tee $O/rank_step_ab.txt
echo "== kernels decode"; for lib in tools/probes/bisect/libatoma_hip_kernargfix.so atoma-infer_amd/lib/libatoma_hip.so; do ATOMA_HIP_LIB=$lib timeout 900 python tools/bench_kernels.py decode 2>&1 | tail -40 > $O/kernels_decode_$(basename $lib .so).txt; done; tail -40 $O/kernels_decode_libatoma_hip.txt | cut -c1-260
