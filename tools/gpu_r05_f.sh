#!/bin/bash
# round 5, GPU call F: the 15 % cut threshold against the previous binary (narrow / wide spreads, headline), decode parity, the C3 step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O
echo "== decode shapes: previous binary (kernargfix) vs the cut threshold"; for lib in tools/probes/bisect/libatoma_hip_kernargfix.so tools/probes/bisect/libatoma_hip_cut15.so tools/probes/bisect/libatoma_hip_kernargfix.so tools/probes/bisect/libatoma_hip_cut15.so; do for shape in "narrow spread U[2048" "narrow spread U[3800" "C2c decode ragged" "C2a decode" "d=64 ragged" "ragged U[2048,4096] MHA"; do ATOMA_HIP_LIB=$lib ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 120 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-200 | sed "s/^/$(basename $lib .so | sed s/libatoma_hip_//) /"; done; done | tee $O/cut_threshold_ab.txt
echo "== decode parity"; timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_decode_dispatch_gpu.py tests/test_graph_capture_gpu.py tests/test_sync_ticket_gpu.py tests/test_kv_fp8_gpu.py tests/test_decode_step_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee $O/parity.txt
echo "== C3 step"; for lib in tools/probes/bisect/libatoma_hip_kernargfix.so tools/probes/bisect/libatoma_hip_cut15.so; do ATOMA_HIP_LIB=$lib timeout 300 python tools/bench_extra.py c3_decode_step 2>&1 | tail -1 | cut -c1-500; done | tee $O/c3_ab.txt
