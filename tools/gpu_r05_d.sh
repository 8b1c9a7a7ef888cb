#!/bin/bash
# round 5, GPU call D: TP prefill at 2 layers after the per-stream GEMM handles, any-order launch probe, small batches on the line (A/B + parity)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05d; mkdir -p $O
echo "== prefill debug, 2 layers"; timeout 200 python tools/probes/tp_prefill_debug.py 4096 2 2>&1 | tail -25 | tee $O/tp_prefill_debug_2layers.txt
echo "== world8 tests"; timeout 500 python -m pytest tests/test_tp_world8_gpu.py -q -m gpu 2>&1 | tail -15 | tee $O/world8_tests.txt
echo "== any-order launch probe"; timeout 60 tools/probes/anyorder_probe 2>&1 | tee $O/anyorder_probe.txt
echo "== small batches on the line"; for v in 0 1; do for shape in "B=64 h=8" "B=1 S" "B=16 S=8192" "d=64 B=16" "TP=8 shard: B=256 h=4"; do ATOMA_DECODE_LINE_SMALL=$v ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 120 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-200 | sed "s/^/line_small=$v /"; done; done | tee $O/line_small_ab.txt
echo "== parity with line_small=1"; ATOMA_DECODE_LINE_SMALL=1 timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_decode_dispatch_gpu.py tests/test_graph_capture_gpu.py tests/test_sync_ticket_gpu.py -q -m gpu 2>&1 | tail -15 | tee $O/line_small_parity.txt
