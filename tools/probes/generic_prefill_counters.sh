#!/bin/bash
# issue / wait / LDS counters of attn_prefill_tile64_kernel at d = 96 and d = 256 (4 causal prompts of 2048 tokens), one --pmc pass per group
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/generic_prefill_counters
mkdir -p $OUT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  for d in 96 256; do
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/d${d}_$i -o c -- python $REPO/tools/probes/generic_prefill_one.py $d > $OUT/d${d}_$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/generic_prefill_counters"
res = {}
for d in (96, 256):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/d{d}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_prefill_tile64" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = {c: round(sum(x) / len(x), 1) for c, x in agg.items()}
    try:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["derived"] = {"cycles_per_launch": round(cyc), "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 8 / 256 / 4 / cyc * 4, 3),
                        "valu_per_mfma": round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 2), "salu_per_mfma": round(v["SQ_INSTS_SALU"] / v["SQ_INSTS_MFMA"], 2),
                        "lds_per_mfma": round(v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"], 2), "lds_bank_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"]), 3),
                        "wave_cycles_waiting_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "wave_cycles_wait_lds_frac": round(v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], 3),
                        "wave_cycles_issuing_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "waves": v["SQ_WAVES"]}
    except KeyError as e:
        v["derived"] = {"missing": str(e)}
    res[f"d={d}"] = v
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/d*_[0-9]
