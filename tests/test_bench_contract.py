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
    """schema of the line the last GPU run printed (profiles/r0N_bench.json); the numbers are asserted where bench.py actually
    runs (the GPU test below): a committed artifact says nothing about the code"""
    import glob
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))[-1]
    lines = [l for l in open(path) if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, full=True)
    assert d["n_gpus"] == 1 and "north_star" in BASE
    assert "paged_decode" in d["roofline"]["kernel"]                                            # named by the dispatcher (atoma_last_decode_kernel)
    assert set(d["cpu_baseline"]["placement"]) >= {"OMP_PROC_BIND", "numa_nodes", "cgroup_cpu_max", "cpus_allowed"}


# Performance floors: by default only WIDE margins that a functional regression breaks (a fallback kernel, a serialised launch:
# factors, not per cent) -- a correctness suite must not flake on a noisy or shared box.  ATOMA_TEST_PERF_FLOORS=1 turns on the tight floors
# (within ~5 % of what the pool's boxes measure), for the builder's own regression runs.
STRICT = os.environ.get("ATOMA_TEST_PERF_FLOORS", "0") == "1"


def floor(strict, loose):
    return strict if STRICT else loose


@pytest.mark.gpu
def test_bench_py_prints_one_json_line(gpu):
    """the headline on a short schedule, with the CPU leg: the line's fields, the north_star fraction, and the check of what the timed
    region wrote against the oracle"""
    env = dict(os.environ, ATOMA_BENCH_CPU_THREADS="8")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-extra", "--no-traffic", "--cpu-sample-seqs", "4"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    check_line(d, full=False)
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1
    assert d["roofline"]["frac"] >= floor(0.80, 0.70)   # north_star asks >= 0.70; 0.83-0.855 measured over the boxes of the pool in rounds 5-6 (round 4's regressed binary: 0.798-0.808)
    assert "G=4" in d["roofline"]["kernel"]
    v = d["verified"]
    assert v["ok"] is True and v["sequences"] == 4 and v["max_err"] < 0.05


@pytest.mark.gpu
def test_bench_extras_hold_their_floors(gpu):
    """the rows of BASELINE.md that ride in the driver's line (tools/bench_extra.py), where they actually run: bit-exact ops bit-exact, and
    each number above a floor a regression would break (the swap / rank-step thresholds belong in a GPU test, not nowhere)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_extra.py"), "swap", "c4_rank_step", "k5_copy_blocks", "k4_reshape_and_cache", "c2b_mha", "c2c_ragged", "p2_prefill_d96", "p2_prefill_d256"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert all("error" not in v for v in d.values()), d
    assert d["swap"]["gpu_to_cpu_frac_of_memcpy"] >= floor(0.85, 0.5) and d["swap"]["cpu_to_gpu_frac_of_memcpy"] >= floor(0.85, 0.5), d["swap"]
    assert d["c4_rank_step"]["ms_per_step"] <= floor(9.5, 14.0) and d["c4_rank_step"]["frac_of_hbm_roofline"] >= floor(0.37, 0.25), d["c4_rank_step"]
    assert d["k5_copy_blocks"]["bit_exact"] and d["k5_copy_blocks"]["frac_hbm"] >= floor(0.65, 0.4), d["k5_copy_blocks"]
    assert d["k4_reshape_and_cache"]["bit_exact"] and d["k4_reshape_and_cache"]["frac_hbm"] >= floor(0.45, 0.25), d["k4_reshape_and_cache"]
    assert d["c2b_mha"]["frac_hbm"] >= floor(0.75, 0.5) and "balanced" in d["c2b_mha"]["kernel"], d["c2b_mha"]
    assert d["c2c_ragged"]["frac_hbm"] >= floor(0.72, 0.5), d["c2c_ragged"]
    # the other head sizes run on the matrix pipe (the row-per-wavefront kernel reads 7-8 TFLOP/s here)
    assert d["p2_prefill_d96"]["TFLOPs"] >= floor(150, 60) and d["p2_prefill_d256"]["TFLOPs"] >= floor(150, 60), (d["p2_prefill_d96"], d["p2_prefill_d256"])
