# the 70B TP = 8 shard's decode call (B = 64, one kv head, 8 q heads, context 4096): tiles in flight x wavefronts per CU x in-launch merge
export ATOMA_BENCH_DECODE_SHAPE="B=64 h=8"
for p8 in 3 4 5; do for wpc in 0 2 8; do for wg in 1 2; do
  echo "== P8=$p8 waves_per_cu=$wpc wg_merge=$wg"; ATOMA_DECODE_MQK_P8=$p8 ATOMA_DECODE_WAVES_PER_CU=$wpc ATOMA_DECODE_WG_MERGE=$wg timeout 200 python tools/bench_kernels.py decode 2>&1 | grep ms | cut -c60-200
done; done; done
