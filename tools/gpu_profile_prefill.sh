#!/bin/bash
# rocprofv3 evidence for the prefill kernel (P1: 8B head shape, causal, S = 2048 x 16): kernel-trace stats,
# then MFMA-busy / clock counters in their own pass.   gpurun -- 'bash tools/gpu_profile_prefill.sh r01'
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_prefill_$TAG -o prefill -- python $REPO/tools/_exp_prefill.py > $OUT/prof_prefill_$TAG.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_prefill_$TAG -o prefill -- python $REPO/tools/_exp_prefill.py > $OUT/pmc_prefill_$TAG.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/pmc_prefill_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg, dur = collections.defaultdict(list), []
for r in csv.DictReader(open(f)):
    if "prefill" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
ns = sum(dur) / len(dur)
clk = m["GRBM_GUI_ACTIVE"] / 8 / (ns * 1e-9) / 1e9            # the counter sums the 8 XCDs
simd_cycles = 1024 * m["GRBM_GUI_ACTIVE"] / 8
out = {"kernel": "prefill_mfma_kernel<bf16,128,causal,4 waves,2 buffers>", "workload": "S=2048 x16, h=32, h_k=8, d=128, causal",
       "counters_mean_per_launch": m, "duration_ns": ns, "effective_clock_GHz": round(clk, 3),
       "mfma_pipe_busy_frac_at_actual_clock": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
       "flops": 4 * 2048 * 2048 * 32 * 128 / 2 * 16, "TFLOPs": round(4 * 2048 * 2048 * 32 * 128 / 2 * 16 / (ns * 1e-9) / 1e12, 1),
       "valu_insts_per_mfma": round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2)}
json.dump(out, open("$OUT/prefill_pmc_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
