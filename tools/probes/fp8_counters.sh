#!/bin/bash
# What bounds the fp8 decode kernel: counter passes on the C2a shape over the fp8 cache and, for comparison,
# on the bf16 headline kernel; and the same kernel on head-major pages (a stride change through the ABI).
# Run on the GPU box:  bash tools/probes/fp8_counters.sh   (writes gpurun_out/fp8ctr/)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fp8ctr
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
export ATOMA_FP8_SHAPE="C2a"
python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/plain.jsonl 2>&1
ATOMA_FP8_LAYOUT=head_major python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/head_major.jsonl 2>&1
ATOMA_DECODE_FP8_WG=0 python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/plain_wg0.jsonl 2>&1
ATOMA_DECODE_FP8_WG=0 ATOMA_FP8_LAYOUT=head_major python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/head_major_wg0.jsonl 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TA_BUSY_avr TD_BUSY_avr" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/fp8_$i -o c -- python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/fp8_$i.log 2>&1
  ATOMA_BENCH_DECODE_SHAPE=C2a timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/bf16_$i -o c -- python $REPO/tools/bench_kernels.py decode > $OUT/bf16_$i.log 2>&1
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/fp8ctr"
res = {}
for tag in ("fp8", "bf16"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "paged_decode" not in k:
                continue
            agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[tag] = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
cat $OUT/plain.jsonl $OUT/head_major.jsonl $OUT/plain_wg0.jsonl $OUT/head_major_wg0.jsonl
