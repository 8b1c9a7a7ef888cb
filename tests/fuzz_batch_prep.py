"""Differential fuzzing of atoma_prepare_inputs (host logic: runs WITHOUT a GPU) against the restatement of ModelWorker::prepare_input_tensors
(backends/vllm/src/worker.rs:224-460; oracle/batch_prep_oracle.py): integer work, bit-exact.  Random batches of 1 .. 256 sequences, prompts / decode tokens in
any mix, chunked prefill, sliding windows, pages of 8 .. 64 tokens, lengths up to 2000.  Test infrastructure.

    python tests/fuzz_batch_prep.py --seconds 240        (round 6: 54 955 cases, no finding)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)

import atoma_hip as ah  # noqa: E402
from oracle import batch_prep_oracle as BO  # noqa: E402
import test_batch_prep as T  # noqa: E402


def one(seed):
    rng = np.random.default_rng(10_000 + seed)
    chunked = bool(rng.integers(2))
    sliding = [None, None, 16, 40, 64, 300][int(rng.integers(6))]
    block = int(rng.choice([8, 16, 32, 64]))
    nseq = int(rng.choice([1, 2, 7, 40, 100, 256]))
    seqs = T.random_batch(rng, nseq, block, chunked, p_prompt=float(rng.choice([0, 0.1, 0.5, 1.0])), max_len=int(rng.choice([3, 40, 300, 2000])))
    got, _ = ah.prepare_inputs_host(seqs, block, sliding, chunked)
    T.same(got, BO.prepare_inputs(seqs, block, sliding, chunked))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    t0, n, fails, seed = time.time(), 0, [], a.seed
    while time.time() - t0 < a.seconds and len(fails) < 20:
        try:
            one(seed)
        except Exception as e:        # noqa: BLE001  (a finding, whatever it is)
            fails.append(dict(seed=seed, finding=repr(e)[:300]))
        n, seed = n + 1, seed + 1
    print(json.dumps(dict(cases=n, seeds=[a.seed, seed - 1], failures=len(fails), findings=fails)))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
