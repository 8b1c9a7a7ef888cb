"""The bench line is a contract with the driver: ONE JSON line on stdout with the fields below.  The CPU test checks the line
committed under profiles/ (what the last GPU round printed); the GPU test runs bench.py itself on a short schedule."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = json.load(open(os.path.join(ROOT, "BASELINE.json")))

TOP = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
ROOF = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


def check_line(d, full):
    for k in TOP:
        assert k in d, k
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None       # no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "configs[1]" in d["config"]["workload"]                                               # the configuration the metric is quoted on
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for k in ROOF:
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1.0
    if full:
        assert r["traffic"] is not None and 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10   # no wasted re-reads
        c = d["cpu_baseline"]
        for k in CPU:
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_committed_bench_line_has_the_contract_fields():
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r03_bench.json")) if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, full=True)
    assert d["n_gpus"] == 1 and "north_star" in BASE
    assert d["roofline"]["frac"] >= 0.70          # north_star: >= 70 % of HBM peak on the paged-attention decode micro-bench
    assert "paged_decode" in d["roofline"]["kernel"] and "G=4" in d["roofline"]["kernel"]      # named by the dispatcher (atoma_last_decode_kernel)
    sw = d["extra"]["swap"]
    assert 0.9 < sw["gpu_to_cpu_frac_of_memcpy"] <= 1.05 and 0.9 < sw["cpu_to_gpu_frac_of_memcpy"] <= 1.05   # the swap against its pinned-memcpy ceiling
    assert d["extra"]["c4_rank_step"]["ms_per_step"] < 9.5                                      # the 70B TP = 8 rank step (round 2: 10.3-10.5 ms)
    assert set(d["cpu_baseline"]["placement"]) >= {"OMP_PROC_BIND", "numa_nodes", "cgroup_cpu_max", "cpus_allowed"}


@pytest.mark.gpu
def test_bench_py_prints_one_json_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra", "--no-traffic", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    check_line(d, full=False)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1
