#!/bin/bash
# round 5, GPU call H: the fused all-reduce + add + norm (tests, TP steps), the 80-layer prefill chunk as 8 ranks, more hipBLASLt candidates for the configs[2] step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_allreduce_xgmi_gpu.py tests/test_tp_step_gpu.py tests/test_tp_world8_gpu.py tests/test_bench_multirank_gpu.py tests/test_bench_contract.py -q -m gpu --durations=8 2>&1 | tail -30 | tee $O/tests.txt
echo "== 8 virtual ranks, decode, 80 layers, fused engines"; (ATOMA_TP_STEP_WATCHDOG_S=500 timeout 600 python tools/tp_step.py --virtual-ranks 8 --steps 5 2>&1 | tail -1) | tee $O/tp_step_8_virtual_ranks.json | cut -c1-1200
echo "== 8 virtual ranks, prefill chunk 4096, 80 layers"; (ATOMA_TP_STEP_WATCHDOG_S=500 timeout 600 python tools/tp_step.py --virtual-ranks 8 --prefill 4096 --steps 2 2>&1 | tail -1) | tee $O/tp_prefill_8_virtual_ranks.json | cut -c1-1200
echo "== C3 step with 16 / 64 / 128 timed candidates"; for n in 16 64 128; do ATOMA_LINEAR_CANDIDATES=$n timeout 600 python tools/bench_extra.py c3_decode_step 2>&1 | tail -1 | cut -c1-330 | sed "s/^/candidates=$n /"; done | tee $O/c3_candidates.txt
