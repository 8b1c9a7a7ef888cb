#!/bin/bash
# round 5, GPU call O: the 16-byte partial store with its hazard nop -- parity first (everything that stores or merges pieces), then previous binary vs new
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $O
echo "== parity"; timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_kv_fp8_gpu.py tests/test_sync_ticket_gpu.py tests/test_graph_capture_gpu.py tests/test_decode_step_gpu.py tests/test_attention_golden_gpu.py tests/test_decode_dispatch_gpu.py tests/test_full_size_gpu.py tests/test_tp_step_gpu.py -q -m gpu 2>&1 | tail -12 | tee $O/parity.txt
PREV=tools/probes/bisect/libatoma_hip_prev.so; NEW=atoma-infer_amd/lib/libatoma_hip.so
echo "== previous vs new"; for lib in $PREV $NEW $PREV $NEW; do n=$([ $lib = $PREV ] && echo prev || echo new); for shape in "C2c decode ragged" "B=64 h=8" "B=16 S=8192" "C2a decode" "d=96"; do ATOMA_HIP_LIB=$lib ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 150 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-150 | sed "s/^/[$n] /"; done; ATOMA_HIP_LIB=$lib timeout 300 python tools/rank_step.py 2>&1 | tail -1 | cut -c1-190 | sed "s/^/[$n] /"; done | tee $O/partial_store16_ab.txt
