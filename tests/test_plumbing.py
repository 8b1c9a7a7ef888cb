"""C1 plumbing on the CPU oracle: the continuous-batching trace of tests/plumbing.py must be
self-consistent -- every paged decode equals dense attention over the sequence's own K/V history,
bit for bit, across page boundaries, an early finish and page reuse by a late arrival."""
from plumbing import Dims, OracleOps, Weights, run_trace


def test_paged_trace_equals_dense_history():
    hist = run_trace(OracleOps(Dims), Weights(Dims), Dims, dense_check=True, steps=32)
    assert sorted(hist) == [0, 1, 2, 3]
    assert len(hist[0]) == 32 and len(hist[1]) == 10 and len(hist[2]) == 32 and len(hist[3]) >= 19
    assert all(0 <= t < Dims.vocab for toks in hist.values() for t in toks)


def test_trace_is_deterministic():
    a = run_trace(OracleOps(Dims), Weights(Dims), Dims, steps=14)
    b = run_trace(OracleOps(Dims), Weights(Dims), Dims, steps=14)
    assert a == b
