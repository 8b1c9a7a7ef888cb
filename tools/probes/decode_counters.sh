#!/bin/bash
# issue / wait / request counters of the decode kernels on the shapes VERDICT r4 item 6 names (separate --pmc passes with kernel-trace off, as the guide prescribes):
# the headline, the ragged line, MHA, the 70B TP = 8 shard at B = 64, B = 1.   -> gpurun_out/decode_counters/summary.json
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/decode_counters
mkdir -p $OUT
shapes=("C2a decode" "C2c decode ragged" "C2b decode MHA" "B=64 h=8" "B=1 S" "narrow spread U[2048")
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); j=0
  for shape in "${shapes[@]}"; do
    j=$((j+1))
    ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s${j}_$i -o c -- python $REPO/tools/bench_kernels.py decode > $OUT/s${j}_$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/decode_counters"
names = ["C2a headline", "C2c ragged U[2048,4096]", "C2b MHA", "70B TP=8 shard B=64 (split + combine)", "B=1 (split + combine)", "narrow spread U[2048,2560)"]
res = {}
for j, nm in enumerate(names, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/s{j}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "paged_decode" in r["Kernel_Name"] or "decode_combine" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        v = {c: round(sum(x) / len(x), 1) for c, x in cs.items()}
        try:
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            v["derived"] = {"valu_pipe_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3), "wave_cycles_waiting_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
                            "wave_cycles_issuing_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1),
                            "read_requests_per_cu_per_cycle": round(v["TCP_TCC_READ_REQ_sum"] / 256 / cyc, 4), "l2_hit_frac": round(v["TCC_HIT_sum"] / max(1.0, v["TCC_REQ_sum"]), 3),
                            "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "write_KiB": v["WRITE_SIZE"]}
        except KeyError as e:
            v["derived"] = {"missing": str(e)}
        res[nm + ": " + k.replace("void atoma::", "")] = v
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))
PY
rm -rf $OUT/s*_[0-9]
