#!/bin/bash
# round 5, GPU call E: copy_blocks A/B against the previous binary, narrow-spread ragged batches on / off the cut line, the C3 step's
# per-kernel breakdown on the new build, the configs[2]-contexts full-size test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
echo "== k5 copy_blocks: previous binary vs new"; for lib in tools/probes/bisect/libatoma_hip_kernargfix.so atoma-infer_amd/lib/libatoma_hip.so; do for i in 1 2; do ATOMA_HIP_LIB=$lib timeout 120 python tools/bench_extra.py k5_copy_blocks k4_reshape_and_cache 2>&1 | tail -1 | cut -c1-700; done; done | tee $O/k5_ab.txt
echo "== narrow spreads: line (cuts) vs one wavefront per (sequence, kv head) in kv-head-major order"; for v in 1 0 1 0; do for shape in "narrow spread U[2048" "narrow spread U[3800" "C2c decode ragged"; do ATOMA_DECODE_STREAM=$v ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 120 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-220 | sed "s/^/decode_stream=$v /"; done; done | tee $O/narrow_spread_ab.txt
echo "== full-size test"; timeout 600 python -m pytest tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -8 | tee $O/full_size_test.txt
echo "== C3 step breakdown"; cd /tmp; ATOMA_BENCH_STEP_CASES=256r timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_step -o step -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py step > $O/prof_step.log 2>&1; cd $GRAFT_REPO_ROOT; f=$(find $O/prof_step -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $f "Llama-3.1-8B decode step, batch 256, contexts U[2048,2560), round 5" > $O/step_breakdown_b256.json; head -50 $O/step_breakdown_b256.json; rm -rf $O/prof_step
