#!/bin/bash
# round 5, GPU call I: the 80-layer prefill chunk as 8 ranks after the GEMM lock no longer covers the launch (16 layers first, short timeouts)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 ATOMA_XGMI_TIMEOUT_MS=10000
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O
for L in 16 80; do echo "== prefill chunk, $L layers"; (ATOMA_TP_STEP_WATCHDOG_S=240 timeout 300 python tools/tp_step.py --virtual-ranks 8 --prefill 4096 --steps 2 --layers $L 2>&1 | tail -3) | tee $O/tp_prefill_8_virtual_ranks_L$L.json | cut -c1-1000; done
echo "== linear / step tests on the new lock scope"; timeout 600 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py tests/test_tp_step_gpu.py -q -m gpu 2>&1 | tail -4
