"""The RMSNorm / RoPE restatement (oracle/norm_rope_oracle.py) against a third, independent implementation: torch, run in
the build container by tests/golden/gen_norm_rope_torch.py.  torch is not the reference (Candle's source is absent, so the
parity of these two ops stays "unpinned"); this catches transcription errors in the restatement.  CPU only."""
import os

import numpy as np
import pytest

from oracle import norm_rope_oracle as NR
from oracle.halfs import to_f32

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "norm_rope_torch.npz"))
CODES = {"f16": 0, "bf16": 1}


def ulps(a_bits, b_bits, dtype):
    """distance in units of the last place of the storage dtype (monotone integer mapping of the sign-magnitude bits)"""
    def key(b):
        b = b.astype(np.int32)
        return np.where(b & 0x8000, 0x8000 - b, b)
    return np.abs(key(a_bits) - key(b_bits))


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_rms_norm_oracle_vs_torch(tag):
    dt = CODES[tag]
    got = NR.rms_norm(FX[f"rms_{tag}_x"], FX[f"rms_{tag}_w"], float(FX[f"rms_{tag}_eps"]), dt)
    want = FX[f"rms_{tag}_y"]
    d = ulps(got, want, dt)
    # torch multiplies (x * rsqrt) * w in f32 with an f32 mean; the restatement uses an f64 sum: results may sit on
    # opposite sides of a rounding boundary, never further
    assert d.max() <= 1, f"max {d.max()} ulp"
    assert (d == 0).mean() > 0.995


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_rope_table_oracle_vs_torch(tag):
    dt = CODES[tag]
    cos, sin = NR.rope_table(640, 128, 500000.0, dt)
    # numpy's and torch's powf differ by an f32 ulp in a few inv_freq entries, i.e. the angle pos * inv_freq (up to
    # ~640 rad) differs by up to ~640 * 2^-24 = 4e-5 absolute: compare cos / sin absolutely at that scale plus one
    # storage ulp, and require the overwhelming majority of entries to be identical
    eps = 2.0 ** -7 if dt == 1 else 2.0 ** -10
    for got, want in ((cos, FX[f"tab_{tag}_cos"]), (sin, FX[f"tab_{tag}_sin"])):
        g, w = to_f32(got, dt), to_f32(want, dt)
        assert (np.abs(g - w) <= 1e-4 + eps * np.abs(w)).all()
        assert (got == want).mean() > 0.99


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_rope_oracle_vs_torch(tag):
    dt = CODES[tag]
    x, pos = FX[f"rope_{tag}_x"], FX[f"rope_{tag}_pos"]
    cos, sin = FX[f"tab_{tag}_cos"], FX[f"tab_{tag}_sin"]          # torch's own table: isolates the rotation arithmetic
    per_op = NR.rope(x, cos, sin, pos, dt, mode="per_op")
    assert np.array_equal(per_op, FX[f"rope_{tag}_per_op"]), "per-op (Candle-style, tensor-dtype arithmetic) must be bit-exact"
    fused = NR.rope(x, cos, sin, pos, dt, mode="fused")
    assert np.array_equal(fused, FX[f"rope_{tag}_fused"]), "f32 round-once variant must be bit-exact"
    assert np.isfinite(to_f32(per_op, dt)).all()
