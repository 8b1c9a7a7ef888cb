"""configs[3] (Llama-3.1-70B, TP = 8, batch 64, prefill 4096 + decode) EXECUTED as 8 ranks -- on one device, which is the hardware
there is: tools/tp_step.py --virtual-ranks 8 places the eight shards of the TP = 8 job (own weights, KV caches, streams, staging
regions) on device 0 and runs the step with the direct all-reduce among all eight.  Every layer has the full configs[3] shapes
(hidden 8192, 8 q / 1 kv heads and 3584 MLP columns per rank, batch 64 at context 4096; prefill: one 4096-token chunk, 64 MiB
all-reduces); the LAYER COUNT is cut to keep the test short (the 80-layer runs are committed under profiles/), and the output says so.
Checked: every all-reduce completed (no timed-out wait), all 8 ranks end with bit-identical logits, and -- shards cut on the device from
one full model -- the logits equal the unsharded step's up to the moved rounding points.
Reference: models/src/llama_nccl.rs:118-200 (the layer), 139,195 (the two all-reduces), multi_gpu.rs:141-179, model_executor.rs:413,436-439."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tp_step(*args, timeout=420):
    env = dict(os.environ, ATOMA_XGMI_TIMEOUT_MS="30000", ATOMA_TP_STEP_WATCHDOG_S=str(timeout - 30))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tp_step.py"), "--virtual-ranks", "8"] + list(args), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[-1])


def common_checks(out, layers, message_bytes):
    assert out["world"] == 8 and "8 ranks of a TP=8 job on ONE device" in out["workload"]
    assert out["allreduces_per_step"] == 2 * layers and out["allreduce_message_bytes"] == message_bytes
    assert out["xgmi_status"] == [0] * 8, out["xgmi_status"]              # no rank ever gave up waiting for a peer
    assert out["ranks_bit_identical"] is True and out["ranks_compared"] == 8
    assert "reduced" in out and f"{layers} of the model's 80 layers" in out["reduced"]
    assert all(v and v["step_ms"] > 0 for v in out["engines"].values())


def test_decode_step_as_eight_ranks_full_shapes():
    out = tp_step("--layers", "4", "--steps", "3", "--check-unsharded")
    common_checks(out, 4, 64 * 8192 * 2)
    # the 1 MiB message through both kernels, plain and with the residual add + RMSNorm inside the all-reduce's launch (the last engine measured
    # leaves the logits that are compared: a fused one)
    assert set(out["engines"]) == {"xgmi_one_shot", "xgmi_two_shot", "xgmi_two_shot_fused_add_norm", "xgmi_one_shot_fused_add_norm"}
    v = out["vs_unsharded"]
    assert v["rows"] == 64 and v["max_abs_diff"] < 0.12 and v["argmax_agree_where_clear"], v


def test_prefill_chunk_as_eight_ranks_full_shapes():
    out = tp_step("--layers", "2", "--steps", "1", "--prefill", "4096", "--check-unsharded")
    common_checks(out, 2, 4096 * 8192 * 2)                                 # [4096, 8192] bf16 = 64 MiB
    assert "prefill chunk of 4096 tokens" in out["workload"]
    v = out["vs_unsharded"]
    assert v["rows"] == 1 and v["max_abs_diff"] < 0.12, v
