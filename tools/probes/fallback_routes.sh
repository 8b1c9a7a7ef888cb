# the routes behind the A/B options still pass the parity tests (dispatch-table assertions excluded: they pin the defaults)
(ATOMA_DECODE_LINE_MERGE=0 ATOMA_DECODE_PAIR64=0 timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_kv_fp8_gpu.py tests/test_host_ops_gpu.py tests/test_decode_step_gpu.py -q --tb=line -x 2>&1 | tail -6) > gpurun_out/fallback_decode.log
(ATOMA_PREFILL_CFG=0 timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_attention_golden_gpu.py tests/test_host_ops_gpu.py -q --tb=line 2>&1 | tail -6) > gpurun_out/fallback_prefill.log
(ATOMA_LINEAR_TILE=0 timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_decode_step_gpu.py -q --tb=line 2>&1 | tail -6) > gpurun_out/fallback_linear.log
tail -3 gpurun_out/fallback_decode.log gpurun_out/fallback_prefill.log gpurun_out/fallback_linear.log
