#!/usr/bin/env python3
"""Per-kernel breakdown of ONE decode step from a rocprofv3 kernel trace of `tools/bench_kernels.py step`.

usage: python tools/step_breakdown.py gpurun_out/prof_step_<tag>/step_kernel_trace.csv > profiles/<tag>_step_breakdown.json
The trace holds warm-up steps, the vendor library's tuning runs and several timed steps; the LAST step (from its embedding
kernel to the end of the trace) is the one summarised: launches, average and total time per kernel, busy time and wall time."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "embedding_kernel" in r["Kernel_Name"]]
last = rows[first[-1]:]
agg = collections.OrderedDict()
busy = 0
for r in last:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    a = agg.setdefault(r["Kernel_Name"].split("(")[0][:120], [0, 0])
    a[0] += 1
    a[1] += d
wall = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
out = {"step_wall_ms": round(wall / 1e6, 3), "kernel_busy_ms": round(busy / 1e6, 3), "launches": len(last), "kernels": [
    {"kernel": k, "launches": n, "avg_us": round(d / n / 1e3, 1), "total_ms": round(d / 1e6, 3), "share": round(d / busy, 3)}
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])]}
if len(sys.argv) > 2:
    out["workload"] = sys.argv[2]
print(json.dumps(out, indent=1))
