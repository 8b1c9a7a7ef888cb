#!/usr/bin/env python3
"""Counters of the 256-row projection kernels (DESIGN.md 4.8d's claim: bound by the CU's request path, not by HBM or the matrix pipe): one
rocprofv3 --pmc pass per counter group over `linear256_ab.py <shape>` with one variant, summarised per kernel -> one JSON object.

    python tools/probes/linear256_counters.py [shape=gate_up] [variant=wide] > profiles/r06_linear_wide_counters.json
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shape = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
variant = sys.argv[2] if len(sys.argv) > 2 else "wide"
GROUPS = ["GRBM_GUI_ACTIVE SQ_BUSY_CYCLES", "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES",
          "SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS", "FETCH_SIZE", "WRITE_SIZE",
          "TCC_HIT_sum TCC_MISS_sum"]
env = dict(os.environ, TMPDIR="/tmp", L256_VARIANTS=variant)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
d = tempfile.mkdtemp(prefix="l256_", dir="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.join(ROOT, "tools", "probes", "linear256_ab.py"), shape],
               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
shutil.rmtree(d, ignore_errors=True)
for g in GROUPS:
    d = tempfile.mkdtemp(prefix="l256_", dir="/tmp")
    subprocess.run(["rocprofv3", "--pmc"] + g.split() + ["--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.join(ROOT, "tools", "probes", "linear256_ab.py"), shape],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
out = {}
for k, cs in ctr.items():
    if "linear_wide" not in k and "Cijk" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    e = {"launches_traced": len(dur.get(k, [])), "duration_us_last_half": None, "counters_mean_per_launch": {n: round(v, 1) for n, v in c.items()}}
    ds = sorted(dur.get(k, []))
    if ds:
        e["duration_us_median"] = round(ds[len(ds) // 2] / 1e3, 2)
    if "GRBM_GUI_ACTIVE" in c and ds:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (ds[len(ds) // 2] * 1e-9)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["effective_clock_GHz"] = round(clk / 1e9, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:                            # summed over SIMDs: / (4 SIMDs x 256 CUs x active cycles)
            e["mfma_pipe_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * c["GRBM_GUI_ACTIVE"] / 8), 4)
    if "SQ_WAVE_CYCLES" in c:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n in c:
                e[n.lower() + "_frac_of_wave_cycles"] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac_of_lds_cycles"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_side_traffic_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        e["traffic_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md's gfx950 correction; counts what leaves the L2s towards the fabric (Infinity-Cache hits included)"
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        e["l2_hit_frac"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2)
        e["lds_per_mfma"] = round(c.get("SQ_INSTS_LDS", 0) / c["SQ_INSTS_MFMA"], 2)
    out[k] = e
print(json.dumps({"shape": shape, "variant": variant, "kernels": out}, indent=1))
