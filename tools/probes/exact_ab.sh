# the two arithmetic variants of the hand-scheduled prefill stream at sustained clocks (60 ms warm-up in bench_kernels.timeit)
for rep in 1 2; do for ek in 0 512 2147483647; do
  echo "== prefill_exact_keys=$ek"; ATOMA_PREFILL_CFG=4 ATOMA_PREFILL_EXACT_KEYS=$ek timeout 300 python tools/bench_kernels.py prefill 2>&1 | grep "d=128" | cut -c14-80,95-170
done; done
