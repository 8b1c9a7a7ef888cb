"""bench.py --gpus 2 / --gpus 8 end to end on ONE device: the ranks under torch.distributed.run, kv heads sharded over them,
the communicator (direct all-reduce over HIP IPC: RCCL refuses two ranks on one device) counted through itself, and the
tensor-parallel step of configs[3] in its reduced plumbing form -- what the 8-GPU SCALE run executes, minus the physical links."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_on_one_device(gpu, world):
    """world 8: the command line of the driver's 8-GPU run (`-m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`), all eight
    ranks on device 0 -- each holds 4 q heads / 1 kv head of the headline workload, the communicator has 8 members."""
    env = dict(os.environ, ATOMA_BENCH_ONE_DEVICE="1", ATOMA_BENCH_TP_SMALL="1", ATOMA_XGMI_TIMEOUT_MS="30000", ATOMA_BENCH_EXTRA_TIMEOUT="300",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29533 + world),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "32", "--seq", "1024", "--no-traffic"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1])
    assert out["n_gpus"] == world and out["ranks_expected"] == world
    assert out["ranks_seen"] == world, out                     # counted through the communicator itself
    assert sorted(d["rank"] for d in out["rank_devices"]) == list(range(world)) and all(d["pci_bus_id"] for d in out["rank_devices"])
    assert f"tp{world}" in out["config"]["parallelism"] and out["scaling"] == "strong"
    pr = out["per_rank"]                                        # every rank's own shard and kernel time
    assert [r_["rank"] for r_ in pr] == list(range(world))
    assert all(r_["q_heads"] == 32 // world and r_["kv_heads"] == 8 // world and r_["kernel_ms"] > 0 and "paged_decode" in r_["kernel"] for r_ in pr)
    assert "paged_decode" in out["roofline"]["kernel"]
    tps = out["tp_step"]
    assert tps["engines_available"].get("xgmi") is True and tps["xgmi_step_ms"] and tps["xgmi_allreduce_us"], tps
    assert tps["rccl_step_ms"] is None and "reduced" in tps
    assert "cpu_baseline" not in out                           # an N = 1 leg
    pf = out["preflight"]["direct"]                            # first contact before the timed region: known pattern, every size the staging region takes
    assert pf["16B"]["ok"] and pf["16B"]["us"] > 0 and pf["1MiB"]["ok"] and pf["1MiB"]["us"] > 0, pf
    assert pf["64MiB"]["ok"] and "skipped" in pf["64MiB"].get("note", ""), pf      # bench.py's direct-only handle stages 2 MiB: reported, not silently dropped


def test_bench_one_rank_through_the_rccl_communicator(gpu):
    """ATOMA_BENCH_FORCE_COMM=1: one rank takes the multi-rank code path of bench.py -- gloo rendezvous, an RCCL communicator (ncclCommInitRank
    with one member), the first-contact preflight through BOTH engines of that communicator (ncclAllReduce, and the direct kernels the
    communicator builds on request), ranks counted through it -- the part of the 8-GPU run's plumbing that RCCL lets a 1-GPU box execute."""
    env = dict(os.environ, ATOMA_BENCH_FORCE_COMM="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "32", "--seq", "1024", "--no-traffic", "--no-extra",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1])
    assert out["ranks_seen"] == 1 and "rccl" in out["communicator"]["kind"]
    pf = out["preflight"]
    for size in ("16B", "1MiB", "64MiB"):
        assert pf["rccl"][size]["ok"] and pf["rccl"][size]["us"] > 0, pf
        assert pf["direct"][size]["ok"], pf
    assert "[preflight] rccl" in r.stderr and "PASS" in r.stderr
