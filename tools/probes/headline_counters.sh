#!/bin/bash
# issue / wait counters of the headline decode kernel (bench.py workload) and of the rank-step attention (70B TP = 8 shard, batch 64)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/headctr
mkdir -p $OUT
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/head_$i -o c -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-traffic > $OUT/head_$i.log 2>&1
  ATOMA_BENCH_DECODE_SHAPE="B=64" timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/shard_$i -o c -- python $REPO/tools/bench_kernels.py decode > $OUT/shard_$i.log 2>&1
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/headctr"
res = {}
for tag in ("head", "shard"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "paged_decode" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        v = {c: sum(x) / len(x) for c, x in cs.items()}
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["derived"] = {"valu_pipe_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3), "wave_cycles_waiting_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
                        "wave_cycles_issuing_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1),
                        "read_requests_per_cu_per_cycle": round(v["TCP_TCC_READ_REQ_sum"] / 256 / cyc, 4), "wave_life_frac_of_kernel": round(v["SQ_WAVE_CYCLES"] * 4 / v["SQ_WAVES"] / cyc, 3)}
        res[tag + ": " + k] = v
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))
PY
