# Do repeated projections over the SAME weights (117 MB gate/up, 21 MB q/k/v: both fit the 256 MB memory-side cache) run faster when the
# weight pieces are fetched without the non-temporal hint?  If yes, prefetching the next op's weights during the current op would pay.
for nt in 1 0; do echo "== ATOMA_LINEAR_TILE_W_NT=$nt"; ATOMA_LINEAR_TILE_W_NT=$nt timeout 300 python tools/bench_kernels.py linear_mid 2>&1 | grep "70B/8" | grep "own kernel (tile" | cut -c14-150; done
for nt in 1 0; do echo "== rank step ATOMA_LINEAR_TILE_W_NT=$nt"; ATOMA_LINEAR_TILE_W_NT=$nt timeout 300 python tools/rank_step.py --layers 80 --iters 20 2>&1 | tail -1 | cut -c100-230; done
