(timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_attention_golden_gpu.py -q --tb=short 2>&1 | tail -4) > gpurun_out/split_tests.log
cat gpurun_out/split_tests.log
for rep in 1 2 3; do
for lib in atoma-infer_amd/lib/libatoma_hip.so tools/probes/libatoma_hip_prev.so; do
  echo "== $lib"; ATOMA_PREFILL_CFG=4 ATOMA_HIP_LIB=$lib timeout 300 python tools/bench_kernels.py prefill 2>&1 | grep "d=128" | cut -c31-50,82-100
done; done
