"""oracle/fp8_oracle.py against torch.float8_e4m3fn values produced in the build container (tests/golden/gen_fp8_torch.py):
all 256 decodings, and ~45 000 encodings including every exact tie between neighbouring codes and out-of-range inputs
(clamped first: the cache-write rule saturates).  CPU only."""
import os

import numpy as np

from oracle import fp8_oracle as F

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "fp8_torch.npz"))


def test_decode_table_matches_torch():
    want = FX["decode"]
    assert np.array_equal(np.isnan(want), np.isnan(F.DECODE)) and np.isnan(F.DECODE).sum() == 2
    ok = ~np.isnan(want)
    assert np.array_equal(want[ok], F.DECODE[ok])
    assert F.DECODE[0x7E] == 448 and F.DECODE[0x01] == 2.0 ** -9 and F.DECODE[0x08] == 2.0 ** -6


def test_encode_matches_torch_including_ties_and_saturation():
    assert np.array_equal(F.encode(FX["x"]), FX["encode_clamped"])
    assert F.encode(np.float32([1e9, -1e9, np.nan])).tolist() == [0x7E, 0xFE, 0x7F]
    codes = np.arange(256, dtype=np.uint8)
    finite = ~np.isnan(F.DECODE)
    rt = F.encode(F.DECODE[finite])
    assert np.array_equal(rt[1:], codes[finite][1:]) or np.array_equal(F.decode(rt), F.DECODE[finite])     # -0.0 / +0.0 keep their sign bit


def test_cache_write_rule():
    from oracle.halfs import BF16
    from util import rand_half
    rng = np.random.default_rng(0)
    k, v = rand_half(rng, (5, 2, 16), BF16, 3.0), rand_half(rng, (5, 2, 16), BF16, 3.0)
    kc, vc = np.zeros((3, 4, 2, 16), np.uint8), np.zeros((3, 4, 2, 16), np.uint8)
    ks, vs = np.float32([0.02, 0.5]), np.float32([1.0, 0.01])
    F.reshape_and_cache_flash_fp8(k, v, kc, vc, np.array([5, -1, 0, 11, 6]), ks, vs, BF16)
    assert not kc[0, 1:].any() and not kc[2, :3].any()                       # untouched slots, the padding token skipped
    assert np.array_equal(kc[1, 1], F.quantize(k[:1], BF16, ks)[0]) and np.array_equal(vc[2, 3], F.quantize(v[3:4], BF16, vs)[0])
    assert (F.decode(vc[..., 1, :]) <= 448).all() and (np.abs(F.decode(vc[2, 3, 1])) == 448).any()     # 3 sigma / 0.01 saturates
    deq = F.dequantize(kc, ks)
    from oracle.halfs import to_f32
    err = np.abs(deq[1, 1] - to_f32(k[0], BF16))
    assert (err <= np.maximum(2.0 ** -4 * np.abs(to_f32(k[0], BF16)), ks[:, None] * 2.0 ** -10) + 1e-6).all()   # half an e4m3 ulp (3 mantissa bits)
