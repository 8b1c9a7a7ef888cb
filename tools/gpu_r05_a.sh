#!/bin/bash
# round 5, GPU call A: (1) do 8 in-process virtual ranks need their own hardware queues, (2) the headline A/B of BINARIES (previous
# builds vs the kernarg fix) with the write counters, (3) the new world-8 tests, (4) the 80-layer 8-virtual-rank decode step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05a; mkdir -p $O
echo "== queues probe"; for q in "" 16; do ( [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; timeout 120 python tools/probes/world8_queues_probe.py 8 ) 2>&1 | tail -2; done | tee $O/world8_queues_probe.txt
echo "== binary A/B"; timeout 1500 python tools/ab_binary.py --rounds 3 --pmc --libs e1c0f19=tools/probes/bisect/libatoma_hip_e1c0f19.so 08eb9f0=tools/probes/bisect/libatoma_hip_08eb9f0.so a67dc09=tools/probes/bisect/libatoma_hip_a67dc09.so kernargfix=tools/probes/bisect/libatoma_hip_kernargfix.so > $O/headline_binary_ab.json 2> $O/headline_binary_ab.err; tail -60 $O/headline_binary_ab.json
echo "== new tests"; timeout 1500 python -m pytest tests/test_allreduce_xgmi_gpu.py tests/test_tp_step_gpu.py tests/test_tp_world8_gpu.py tests/test_bench_multirank_gpu.py -q -m gpu --durations=15 2>&1 | tail -60 | tee $O/new_tests.txt
echo "== 80 layers, 8 virtual ranks"; timeout 900 python tools/tp_step.py --virtual-ranks 8 --steps 5 > $O/tp_step_8_virtual_ranks.json 2> $O/tp_step_8_virtual_ranks.err; tail -3 $O/tp_step_8_virtual_ranks.json; tail -5 $O/tp_step_8_virtual_ranks.err
