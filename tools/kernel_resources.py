#!/usr/bin/env python3
"""usage: tools/kernel_resources.py <csrc/file.hip> -- per-kernel VGPR / spill / scratch / occupancy (gfx950)."""
import re, subprocess, sys, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "atoma-infer_amd")
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast-honor-pragmas", "-fno-honor-nans", "-fno-signed-zeros",
       "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr
rows, cur = [], None
pats = {"vgpr": r" VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "sgpr": r" SGPRs: (\d+)", "spill": r"VGPR Spill: (\d+)",
        "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)",
        "lds": r"LDS Size \[bytes/block\]: (\d+)"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if "error" in line:
        print(line)
    for k, p in pats.items():
        m = re.search(p, line)
        if m and cur is not None:
            cur[k] = m.group(1)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"atoma::", "", name)
    name = re.sub(r"\(.*", "", name)
    print("%-72s vgpr=%s agpr=%s sgpr=%s spill=%s scratch=%s occ=%s lds=%s" % (
        name[:72], r.get("vgpr"), r.get("agpr"), r.get("sgpr"), r.get("spill"), r.get("scratch"), r.get("occ"), r.get("lds")))
