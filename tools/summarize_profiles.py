#!/usr/bin/env python3
"""Turns the rocprofv3 outputs under gpurun_out/ into the small, committed summaries under profiles/.

usage: python tools/summarize_profiles.py r01
  gpurun_out/prof_<tag>/*_kernel_stats.csv           -> profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
  gpurun_out/pmc_fetch_<tag>, pmc_write_<tag> (CSV)  -> profiles/<tag>_pmc.json            (separate --pmc passes)
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream, so
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
for src in glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}*", "*_kernel_stats.csv")):
    name = os.path.basename(os.path.dirname(src)).replace("prof_", "")
    shutil.copy(src, os.path.join(out, f"{name}_kernel_stats.csv"))
    print("copied", src)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{kind}_{tag}*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for kern, ctrs in pmc.items():
    if "rocclr" in kern:
        continue
    e = {c: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for c, v in ctrs.items()}
    f = e.get("FETCH_SIZE", {}).get("mean_KiB")
    w = e.get("WRITE_SIZE", {}).get("mean_KiB")
    if f is not None and w is not None:
        e["hbm_traffic_bytes_per_launch"] = (2 * f + w) * 1024
        e["correction"] = "2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, KiB -> bytes"
    summary[kern] = e
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# per-kernel breakdown of one decode step (tools/step_breakdown.py over the kernel trace of `bench_kernels.py step`)
import subprocess
for sub, label, name in ((f"prof_step_{tag}", "Llama-3.1-8B decode step, batch 256, context U[2048,2560), bf16 KV", "step_breakdown_b256"),
                         (f"prof_step1_{tag}", "Llama-3.1-8B decode step, batch 1, context 4096, bf16 KV", "step_breakdown_b1"),
                         (f"prof_rank_{tag}", "one rank of the Llama-3.1-70B TP=8 decode step, batch 64, context 4096, 8 of 80 layers", "rank_step_breakdown")):
    for f in glob.glob(os.path.join(root, "gpurun_out", sub, "*kernel_trace.csv")):
        txt = subprocess.run([sys.executable, os.path.join(root, "tools", "step_breakdown.py"), f, label], capture_output=True, text=True).stdout
        open(os.path.join(out, f"{tag}_{name}.json"), "w").write(txt)
        print("wrote", f"{tag}_{name}.json")
# plain copies of the builder-run measurement files
for src, dst in ((f"bench_{tag}.log", f"{tag}_bench.json"), (f"kernels_{tag}.jsonl", f"{tag}_kernels.jsonl"), (f"trace_{tag}.json", f"{tag}_trace.json"),
                 (f"trace_70b_tp8_rank_{tag}.json", f"{tag}_trace_70b_tp8_rank.json"), (f"tp_step_n1_{tag}.json", f"{tag}_tp_step_n1.json"),
                 (f"rank_step_{tag}.json", f"{tag}_rank_step.json"), (f"tp_step_8_virtual_ranks_{tag}.json", f"{tag}_tp_step_8_virtual_ranks_80_layers.json"),
                 (f"tp_prefill_8_virtual_ranks_{tag}.json", f"{tag}_tp_prefill_chunk_8_virtual_ranks_2_layers.json"),
                 (f"tp_step_8_ranks_vs_unsharded_{tag}.json", f"{tag}_tp_step_8_ranks_vs_unsharded_8_layers.json"), (f"rank_step_r02route_{tag}.json", f"{tag}_rank_step_r02_route.json"),
                 (f"linear64_ab_{tag}.jsonl", f"{tag}_linear64_ab.jsonl"), (f"gemm64_probe_{tag}.txt", f"{tag}_gemm64_probe.txt"),
                 (f"fp8_modes_{tag}.txt", f"{tag}_fp8_modes.txt"), (f"stream_force_ab_{tag}.txt", f"{tag}_stream_force_ab.txt"),
                 (f"mqk_ab_{tag}.txt", f"{tag}_mqk_ab.txt"), (f"headline_knobs_{tag}.txt", f"{tag}_headline_knobs.txt")):
    p = os.path.join(root, "gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(out, dst))
        print("copied", src, "->", dst)
