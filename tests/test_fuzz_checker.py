"""The fuzzers' checker, checked (CPU): tests/fuzz_parity.py must pass what the reference's arithmetic produces and must flag what a broken kernel
produces -- a fuzzer whose checker accepts everything proves nothing.  The "kernel" here is the oracle's kernel-faithful mode (P rounded to the
storage type before P.V, as the reference's MMA does), then deliberately damaged."""
import numpy as np

import fuzz_parity as F
import fuzz_ops as FO
from oracle import attn_oracle as A
from oracle.halfs import BF16, F16, to_f32, from_f32
from util import rand_half


def _case(dtype, Lq=40, Lk=700, h=4, hk=2, d=64, scale=None, causal=True, seed=0):
    rng = np.random.default_rng(seed)
    q, k, v = rand_half(rng, (Lq, h, d), dtype), rand_half(rng, (Lk, hk, d), dtype), rand_half(rng, (Lk, hk, d), dtype)
    qf, kf, vf = to_f32(q, dtype), to_f32(k, dtype), to_f32(v, dtype)
    scale = scale or d ** -0.5
    ref, lse = A.attend_rows(qf, kf, vf, scale, causal=causal)
    ker, lse_k = A.attend_rows(qf, kf, vf, scale, causal=causal, mode="kernel", dtype=dtype)
    visible = np.minimum(Lk, np.arange(Lq) + Lk - Lq + 1) if causal else np.full(Lq, Lk)
    bounds = lambda: F.p_bounds(qf, kf, vf, scale, causal, None)
    return from_f32(ker, dtype), lse_k, from_f32(ref, dtype), lse, visible, bounds


def test_the_reference_arithmetic_passes_also_with_a_sharp_scale():
    for dtype in (BF16, F16):
        for scale in (None, 1.7 * 64 ** -0.5, 0.5 * 64 ** -0.5):
            for causal in (True, False):
                out, lse_k, ref, lse, vis, bounds = _case(dtype, scale=scale, causal=causal, seed=int(dtype) + causal)
                msg, _ = F._check(out, lse_k, ref, lse, vis, dtype, "reference arithmetic", bounds)
                assert msg is None, msg


def test_damaged_outputs_are_findings():
    out, lse_k, ref, lse, vis, bounds = _case(BF16)
    bad = out.copy()
    bad[17, 2, 5] = from_f32(to_f32(out[17, 2, 5:6], BF16) + np.float32(6e-3), BF16)[0]      # ONE element off by 6e-3 (outputs are ~0.1 here: the row sees 678 keys)
    assert "beyond the bound" in F._check(bad, lse_k, ref, lse, vis, BF16, "x", bounds)[0]
    bad = out.copy()
    bad[3] = 0xFFFF                                            # a row nobody wrote (the harness poisons the output with 0xFF bytes)
    assert "non-finite" in F._check(bad, lse_k, ref, lse, vis, BF16, "x", bounds)[0]
    bad = out.copy()
    bad[20] = out[21]                                          # a row computed from its neighbour's query
    assert F._check(bad, lse_k, ref, lse, vis, BF16, "x", bounds)[0]
    bad_lse = lse_k.copy()
    bad_lse[1, 7] += 0.01
    assert "LSE" in F._check(out, bad_lse, ref, lse, vis, BF16, "x", bounds)[0]


def test_rows_without_keys_must_be_zeros_with_infinite_lse():
    out, lse_k, ref, lse, vis, bounds = _case(BF16, Lq=40, Lk=25)          # causal, more queries than keys: rows 0..14 see nothing
    assert (vis[:15] <= 0).all() and F._check(out, lse_k, ref, lse, vis, BF16, "x", bounds)[0] is None
    bad = out.copy()
    bad[4, 0, 0] = 0x3F80
    assert "zeros" in F._check(bad, lse_k, ref, lse, vis, BF16, "x", bounds)[0]
    bad_lse = lse_k.copy()
    bad_lse[0, 4] = 0.0
    assert "zeros" in F._check(out, bad_lse, ref, lse, vis, BF16, "x", bounds)[0]


def test_seeds_keep_naming_their_cases():
    """failure reports quote seeds; new case spaces are added ABOVE the old ones (FORWARD / DECODE / STRIDE / LONG bases)"""
    c = F.draw(6614)
    assert (c["kind"], c["sq"], c["B"], c["page"], c["h"], c["hk"], c["dtype"]) == ("kv_cache", 3, 16, 0, 40, 8, F16) and c["lens_k"][6] == 0
    assert F.draw(F.FORWARD_BASE)["kind"] == "forward" and F.draw(F.DECODE_BASE)["B"] == 200 and "qpad" in F.draw(F.STRIDE_BASE) and F.draw(F.LONG_BASE)["sample"]
    assert FO.draw(1)["K"] == 128 and FO.draw(FO.SWAP_BASE)["kind"] == "swap" and FO.draw(FO.FP8_BASE)["kind"] == "fp8_decode"
    a, b = np.zeros((4, 8), np.uint16), np.zeros((4, 8), np.uint16)
    b[1, 1] = 0x3C00
    assert FO._ulp_check(a, a, F16, "x") is None and "max err" in FO._ulp_check(b, a, F16, "x")
