#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 ATOMA_XGMI_TIMEOUT_MS=4000 ATOMA_TP_STEP_VERBOSE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
for L in 4 8 16; do echo "== prefill chunk, $L layers"; (ATOMA_TP_STEP_WATCHDOG_S=100 timeout 130 python tools/tp_step.py --virtual-ranks 8 --prefill 4096 --steps 2 --layers $L 2>&1 | tail -4) | tee $O/L$L.txt | cut -c1-1500; done
echo "== fewer ranks, 16 layers"; for W in 2 4; do (ATOMA_TP_STEP_WATCHDOG_S=100 timeout 130 python tools/tp_step.py --virtual-ranks $W --prefill 4096 --steps 2 --layers 16 2>&1 | tail -3) | cut -c1-700; done | tee $O/fewer_ranks.txt
echo "== 8 ranks 16 layers, shorter chunk T=1024"; (ATOMA_TP_STEP_WATCHDOG_S=100 timeout 130 python tools/tp_step.py --virtual-ranks 8 --prefill 1024 --steps 2 --layers 16 2>&1 | tail -3) | cut -c1-700 | tee $O/T1024.txt
