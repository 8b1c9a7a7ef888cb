#!/bin/bash
# fp8 decode kernel variants (tiles in flight, K fetch format): bash tools/probes/fp8_variants.sh  -> gpurun_out/fp8var/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fp8var
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
for v in "" p3 p6 kl0; do
  lib=$REPO/atoma-infer_amd/lib/libatoma_hip.so
  [ -n "$v" ] && lib=$REPO/tools/probes/libatoma_hip_fp8$v.so
  ATOMA_HIP_LIB=$lib python tools/bench_kernels.py decode_fp8 2>&1 | sed "s/\"workload\": \"/\"workload\": \"[${v:-default}] /" >> $OUT/variants.jsonl
done
done
python - <<'PY'
import json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/fp8var/variants.jsonl"
t = collections.defaultdict(list)
for l in open(out):
    try: d = json.loads(l)
    except Exception: continue
    t[d["workload"]].append(d["frac_hbm"])
for k, v in t.items(): print("%-100s %s" % (k[:100], " ".join("%.3f" % x for x in v)))
PY
