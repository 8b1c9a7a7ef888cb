cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for rot in 1 0; do
  echo "== head_rot=$rot"
  ATOMA_DECODE_HEAD_ROT=$rot python tools/bench_kernels.py decode_fp8 2>&1 | cut -c1-140
  ATOMA_DECODE_HEAD_ROT=$rot python tools/bench_kernels.py decode 2>&1 | cut -c1-140
done
done
echo "== stream waves 12"
ATOMA_DECODE_STREAM_WAVES_PER_CU=12 python tools/bench_kernels.py decode_fp8 2>&1 | cut -c1-140
