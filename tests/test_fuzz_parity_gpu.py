"""A fixed slice of tests/fuzz_parity.py's case space in the GPU suite: random shapes of the three run_mha entry families against the f32 definition
(the long campaigns run as `python tests/fuzz_parity.py --seconds N`; their findings become seeds here)."""
import pytest

import fuzz_parity as F

pytestmark = pytest.mark.gpu


def test_case_space_is_stable():
    """a seed names a case for good (failure reports quote seeds)"""
    assert F.draw(7) == F.draw(7) and {F.draw(s)["kind"] for s in range(40)} == set(F.KINDS)


@pytest.mark.parametrize("seed0", range(0, 48, 8))
def test_random_cases_match_the_oracle(gpu, seed0):
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 8) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings


@pytest.mark.parametrize("seed0", range(F.FORWARD_BASE, F.FORWARD_BASE + 24, 8))
def test_random_operator_calls_match_the_oracle(gpu, seed0):
    """FlashAttention::forward (atoma_flash_attention_forward) on mixed batches of prompts and decode tokens: cache write bit-exact, rows against the definition"""
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 8) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings


@pytest.mark.parametrize("seed0", range(F.LONG_BASE, F.LONG_BASE + 8, 4))
def test_random_long_prompts_match_the_oracle(gpu, seed0):
    """prompts of 500 .. 5000 tokens (a persistent prefill workgroup walks many blocks): nothing unwritten, ~40 sampled rows per sequence against the definition"""
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 4) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings


@pytest.mark.parametrize("seed0", range(F.STRIDE_BASE, F.STRIDE_BASE + 24, 8))
def test_random_cases_with_padded_strides_match_the_oracle(gpu, seed0):
    """q / k / v / o rows as slices of wider buffers, seqlen_q / seqlen_k arguments larger than any sequence; the padding between output rows stays untouched"""
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 8) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings


@pytest.mark.parametrize("seed0", [100, F.DECODE_BASE + 100, F.FORWARD_BASE + 100])
def test_bursts_of_calls_without_a_synchronisation_in_between(gpu, seed0):
    """12 cases launched back to back: the library's scratch block (grown by a later call while an earlier kernel may still run), plan tables and
    arrival counters pass from launch to launch in stream order only"""
    assert not F.run_burst(gpu, [F.draw(s) for s in range(seed0, seed0 + 12)])


@pytest.mark.parametrize("seed", [F.DECODE_BASE + 300, F.DECODE_BASE + 301, F.DECODE_BASE + 302])
def test_captured_decode_call_replayed_with_other_lengths(gpu, seed):
    """graph_case: eager call, capture, then four replays with other lengths in the same block tables (the captured host decisions must hold for any of them)"""
    c = F.draw(seed)
    assert F.graph_case(gpu, c) is None


@pytest.mark.parametrize("seed0", [200, F.STRIDE_BASE + 200])
def test_three_host_threads_with_a_stream_each(gpu, seed0):
    """24 cases dealt to 3 threads that call the library concurrently, each on its own stream (the reference runs one thread per GPU; a server with several
    engines in one process does this): per-stream scratch and counters, shared options / registries"""
    assert not F.run_threads(gpu, [F.draw(s) for s in range(seed0, seed0 + 24)], threads=3)


@pytest.mark.parametrize("seed0", range(F.DECODE_BASE, F.DECODE_BASE + 16, 8))
def test_random_large_decode_batches_match_the_oracle(gpu, seed0):
    """64 .. 512 sequences through whatever the dispatcher picks (the balanced line, the paired kernel when a length hint says ragged, kv-head pairs at d = 64)"""
    findings = [(F.draw(s), msg) for s in range(seed0, seed0 + 8) for msg in [F.try_case(gpu, F.draw(s))] if msg]
    assert not findings, findings


@pytest.mark.parametrize("seed", [6614, 1010, 1031, 1836, 1904, 3798, 4006, 7029, 7052, 7072, 7089])
def test_seeds_that_once_were_findings(gpu, seed):
    """6614: a kv_cache call with 3 query rows and a sequence without keys in the middle of the batch -- the hand-scheduled prefill kernel's block after a
    tile-less block read its K fragments from the wrong ring slot (fixed in tools/pfasm/kernel.py BLOCK_DRAIN).  The others: rows whose softmax mass sits
    on a few keys although they see >= 512 (a sharp scale, ALiBi) -- the checker's simple line was wrong for them, their own bound holds."""
    assert F.try_case(gpu, F.draw(seed)) is None
