#!/bin/bash
# issue / wait / request / traffic counters of attn_decode_anyd2_kernel on the d = 96 (Phi-3-mini heads) and d = 256 shapes, separate --pmc passes
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/generic_decode_counters
mkdir -p $OUT
shapes=("d=96 (Phi" "d=256 B=256" "d=96 B=8")
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); j=0
  for shape in "${shapes[@]}"; do
    j=$((j+1))
    ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s${j}_$i -o c -- python $REPO/tools/bench_kernels.py decode > $OUT/s${j}_$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/generic_decode_counters"
names = ["d=96, 32 MHA heads, B=256 x 2048 (4 wavefronts per unit)", "d=256, 16 / 4 heads, B=256 x 2048 (4 wavefronts per unit)", "d=96, 32 MHA heads, B=8 x 4096 (8 wavefronts per unit)"]
res = {}
for j, nm in enumerate(names, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/s{j}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_decode_anyd2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = {c: round(sum(x) / len(x), 1) for c, x in agg.items()}
    try:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["derived"] = {"valu_pipe_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3), "wave_cycles_waiting_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
                        "wave_cycles_issuing_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1),
                        "read_requests_per_cu_per_cycle": round(v["TCP_TCC_READ_REQ_sum"] / 256 / cyc, 4), "l2_hit_frac": round(v["TCC_HIT_sum"] / max(1.0, v["TCC_REQ_sum"]), 3),
                        "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
    except KeyError as e:
        v["derived"] = {"missing": str(e)}
    res[nm] = v
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))
PY
rm -rf $OUT/s*_[0-9]
