#!/bin/bash
# round 5, GPU call G: the arrival tickets' memory order -- relaxed atomics around write-through stores (shipped) vs release ticket + acquire fence (syncrel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O
for lib in atoma-infer_amd/lib/libatoma_hip.so tools/probes/libatoma_hip_syncrel.so atoma-infer_amd/lib/libatoma_hip.so tools/probes/libatoma_hip_syncrel.so; do n=$(basename $lib .so | sed s/libatoma_hip_*//); for shape in "C2c decode ragged" "ragged U[2048,4096] MHA" "B=64 h=8" "B=16 S=8192"; do ATOMA_HIP_LIB=$lib ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 120 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-170 | sed "s/^/[$n] /"; done; ATOMA_HIP_LIB=$lib timeout 300 python tools/rank_step.py 2>&1 | tail -1 | cut -c1-230 | sed "s/^/[$n] /"; done | tee $O/sync_release_ab.txt
echo "== parity on the release build"; ATOMA_HIP_LIB=tools/probes/libatoma_hip_syncrel.so timeout 600 python -m pytest tests/test_sync_ticket_gpu.py tests/test_decode_gpu.py tests/test_linear_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/syncrel_parity.txt
