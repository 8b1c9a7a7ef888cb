#!/bin/bash
# Page layout A/B through the ABI's strides: [page][h_k][d] (the reference's) against [h_k][page][d]
# ("head-major": the 16 rows of one kv head contiguous), fp8 and bf16 caches, 3 / 4 / 5 tiles in flight (make fp8p).
# Run on the GPU box:  bash tools/probes/fp8_layout.sh   (writes gpurun_out/fp8layout/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fp8layout
mkdir -p $OUT
cd $REPO
for lay in "" head_major; do
  ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_p3.jsonl 2>&1
  ATOMA_DECODE_FP8_WG=2 ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_p3_wg2.jsonl 2>&1
  ATOMA_DECODE_FP8_WG=0 ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_p3_wg0.jsonl 2>&1
  for P in 4 5; do
    ATOMA_HIP_LIB=$REPO/tools/probes/libatoma_hip_fp8p$P.so ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_p$P.jsonl 2>&1
  done
  ATOMA_KV_LAYOUT=$lay python tools/bench_kernels.py decode >> $OUT/bf16.jsonl 2>&1
done
tail -n +1 $OUT/*.jsonl
# timing-only probe (wrong results): K fetched in full 128-byte lines
for lay in "" head_major; do
  ATOMA_HIP_LIB=$REPO/tools/probes/libatoma_hip_fp8kfull.so ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_kfull_TIMING_ONLY.jsonl 2>&1
  ATOMA_DECODE_FP8_WG=0 ATOMA_HIP_LIB=$REPO/tools/probes/libatoma_hip_fp8kfull.so ATOMA_FP8_LAYOUT=$lay python tools/bench_kernels.py decode_fp8 >> $OUT/fp8_kfull_wg0_TIMING_ONLY.jsonl 2>&1
done
tail -n +1 $OUT/*kfull*.jsonl
