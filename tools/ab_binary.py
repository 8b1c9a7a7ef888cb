#!/usr/bin/env python3
"""A/B of whole BINARIES, interleaved on one box -- the rule for every "-x %" claim about the headline.

Round 4 claimed "headline -1 %" for a change by comparing two SETTINGS of the new binary; the new binary itself was 4 % slower than
the one before it (every wavefront copied its 288-byte kernel arguments to scratch: 36 MiB of writes per launch) and nothing compared
the two.  So: a claim about a change is measured against the library built from the commit BEFORE it, never against an option of the
library that contains it.

    tools/ab_binary.py --build e1c0f19 HEAD~1        # builds those commits under .bisect/<sha>/ (git worktree + make), then runs
    tools/ab_binary.py --libs old=path/libatoma_hip.so new=atoma-infer_amd/lib/libatoma_hip.so [--rounds 3] [--pmc] [-- bench args]

Each round runs `bench.py --no-extra --no-cpu-baseline --no-traffic <bench args>` once per library (ATOMA_HIP_LIB), in order
A B C / A B C / ...; --pmc adds one pass per library with the traffic counters on (FETCH_SIZE / WRITE_SIZE per launch).  Prints one
JSON object: per library the kernel_ms of every round, their median, the roofline fraction, and the counters.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(sha):
    """git worktree of `sha` under .bisect/<sha> + make; returns the path of its libatoma_hip.so."""
    d = os.path.join(ROOT, ".bisect", sha)
    if not os.path.isdir(d):
        subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "-f", d, sha])
    subprocess.check_call(["make", "-C", os.path.join(d, "atoma-infer_amd"), "-j", str(min(8, os.cpu_count() or 4))])
    return os.path.join(d, "atoma-infer_amd", "lib", "libatoma_hip.so")


def run_bench(lib, bench_args, traffic):
    env = dict(os.environ, ATOMA_HIP_LIB=os.path.abspath(lib))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline"] + ([] if traffic else ["--no-traffic"]) + bench_args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-500:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[], help="name=path ...")
    ap.add_argument("--build", nargs="*", default=[], help="commits to build under .bisect/ and add to the comparison")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("bench_args", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    bench_args = [x for x in a.bench_args if x != "--"]
    libs = []
    for sha in a.build:
        libs.append((sha, build(sha)))
    for item in a.libs:
        name, path = item.split("=", 1)
        libs.append((name, path))
    if len(libs) < 2:
        sys.exit("need at least two libraries to compare (the previous binary and the new one)")
    res = {name: {"lib": os.path.relpath(os.path.abspath(path), ROOT), "kernel_ms": [], "ms_per_step": []} for name, path in libs}
    for _ in range(a.rounds):
        for name, path in libs:
            o = run_bench(path, bench_args, False)
            if "error" in o:
                res[name]["error"] = o["error"]
                continue
            res[name]["kernel_ms"].append(o["roofline"]["kernel_ms"])
            res[name]["ms_per_step"].append(o["ms_per_step"])
            res[name]["kernel"] = o["roofline"]["kernel"]
            res[name]["algorithmic_bytes_per_launch"] = o["roofline"]["algorithmic_bytes_per_launch"]
    for name, path in libs:
        e = res[name]
        if e["kernel_ms"]:
            e["median_kernel_ms"] = round(statistics.median(e["kernel_ms"]), 4)
            e["frac_of_8TBps"] = round(e["algorithmic_bytes_per_launch"] / (e["median_kernel_ms"] * 1e-3) / 8e12, 4)
        if a.pmc:
            o = run_bench(path, bench_args + ["--steps", "5"], True)
            e["traffic_bytes_per_launch"] = (o.get("roofline") or {}).get("traffic")
            e["traffic_counters_KiB_per_launch"] = (o.get("roofline") or {}).get("traffic_counters_KiB_per_launch")
    print(json.dumps({"workload": "bench.py " + " ".join(bench_args), "order": "interleaved, %d rounds" % a.rounds, "libs": res}, indent=1))


if __name__ == "__main__":
    main()
