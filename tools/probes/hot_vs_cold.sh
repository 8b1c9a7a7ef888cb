cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06i
for nb in 0 1; do for b in 64 256; do
  if [ $b = 64 ]; then sh="r_qkv r_o r_gate_up r_down"; else sh="qkv o gate_up down"; fi
  L256_NBUF=$nb L256_BATCH=$b L256_VARIANTS=wide timeout 300 python tools/probes/linear256_ab.py $sh
done; done > gpurun_out/r06i/hot_vs_cold.jsonl 2>&1
cut -c1-200 gpurun_out/r06i/hot_vs_cold.jsonl
