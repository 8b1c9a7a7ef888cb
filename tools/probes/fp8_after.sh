#!/bin/bash
# After the full-line K fetch: request counters of the fp8 kernel on C2a, and kernel-only times of the ragged (balanced) launch.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fp8after
mkdir -p $OUT
export ATOMA_FP8_SHAPE="C2a"
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/fp8_$i -o c -- python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/fp8_$i.log 2>&1
done
export ATOMA_FP8_SHAPE="C2c"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ragged -o c -- python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/ragged.log 2>&1
export ATOMA_FP8_SHAPE="C2a"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/uniform -o c -- python $REPO/tools/bench_kernels.py decode_fp8 > $OUT/uniform.log 2>&1
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/fp8after"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/fp8_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "paged_decode" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
for tag in ("ragged", "uniform"):
    for f in glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True):
        print(tag)
        for r in list(csv.DictReader(open(f)))[:4]:
            print("  ", r["Name"][:80], r["Calls"], r["AverageNs"])
PY
