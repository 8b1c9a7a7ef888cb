"""A fixed slice of tests/fuzz_ops.py's case space in the GPU suite (projection entries with every epilogue, q/k/v + RoPE + cache, RMSNorm / RoPE,
swap_blocks_multi); campaigns: `python tests/fuzz_ops.py --seconds N`."""
import pytest

import fuzz_ops as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed0", list(range(0, 48, 12)) + [F.SWAP_BASE, F.FP8_BASE])
def test_random_cases_match_the_oracle(gpu, seed0):
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 12) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings
