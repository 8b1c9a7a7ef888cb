bash tools/prefill_pmc.sh r04s 4 > /dev/null 2>&1
(timeout 600 python tools/bench_kernels.py prefill prefill_paged 2>&1) > gpurun_out/kernels_prefill_r04.jsonl
(timeout 300 python tools/probes/pfa_try.py bench 2>&1 | grep bench) > gpurun_out/pfa_try_bench_r04.txt
cat gpurun_out/kernels_prefill_r04.jsonl | cut -c1-200; cat gpurun_out/pfa_try_bench_r04.txt
