#!/usr/bin/env python3
"""gpurun_out/pfprof_<tag> + pfpmc_<tag>_* (tools/prefill_pmc.sh) -> profiles/<tag>_prefill_pmc.json and
profiles/<tag>_prefill_kernel_stats.csv: duration, counters per launch, effective clock, MFMA pipe busy fraction at that clock
(SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs: / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)), VALU instructions per MFMA."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "prefill_asm_persistent"      # or prefill_mfma_kernel
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
dur = None
for f in glob.glob(os.path.join(root, "gpurun_out", f"pfprof_{tag}", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, f"{tag}_prefill_kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            dur, name = float(r["AverageNs"]), r["Name"]
# the sustained-clock duration: the last 10 launches of the trace (the timed ones; the 60 ms of warm-up launches before them include the clock ramp)
for f in glob.glob(os.path.join(root, "gpurun_out", f"pfprof_{tag}", "*kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d10 = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-10:]]
    if d10:
        dur_all, dur = dur, sum(d10) / len(d10)
ctr = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "gpurun_out", f"pfpmc_{tag}_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in ctr.items()}
S, nseq, h, d = 2048, 16, 32, 128
flops = 4.0 * S * S * h * d / 2 * nseq
res = {"kernel": name.split("(")[0], "workload": "S=2048 x16, h=32, h_k=8, d=128, causal, bf16", "counters_mean_per_launch": c, "duration_ns": dur,
       "flops": flops, "TFLOPs": round(flops / dur / 1e3, 1), "frac_of_2.5PF": round(flops / dur / 1e3 / 2500, 4)}
if "GRBM_GUI_ACTIVE" in c:
    gui = c["GRBM_GUI_ACTIVE"] / 8          # the counter is summed over the 8 XCDs
    res["effective_clock_GHz"] = round(gui / dur, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        res["mfma_pipe_busy_frac_at_actual_clock"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * gui), 4)
if c.get("SQ_INSTS_MFMA"):
    res["valu_insts_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"] - 1, 2)    # SQ_INSTS_VALU counts the MFMAs too
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
        if k in c:
            res[k.lower()[3:] + "_per_mfma"] = round(c[k] / c["SQ_INSTS_MFMA"], 3)
if c.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_VMEM"):
        if k in c:
            res[k.lower()[3:] + "_frac_of_wave_cycles"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
res["note"] = ("duration_ns = mean of the last 10 launches of the --kernel-trace pass (sustained clocks: 60 ms of warm-up launches precede them); the counters are "
               "means over all launches of their own passes (cycle and instruction counts do not depend on the clock)")
json.dump(res, open(os.path.join(out, f"{tag}_prefill_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
