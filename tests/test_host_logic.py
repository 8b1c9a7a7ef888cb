"""Host-side logic of the library that needs no GPU: rope table builder, split heuristic."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import attn_oracle as A, norm_rope_oracle as NR
from oracle.halfs import F16, BF16, to_f32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "atoma-infer_amd", "lib", "libatoma_hip.so"))
lib.atoma_rope_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int64, C.c_int]
lib.atoma_num_splits_heuristic.argtypes = [C.c_int64] * 4
lib.atoma_compute_num_splits.argtypes = [C.c_int64] * 5 + [C.c_int]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("scaled", [False, True])
def test_rope_table_matches_oracle(dtype, scaled):
    """models/src/llama.rs:146-200.  libm powf/cosf/sinf vs numpy's float32 routines can differ in
    the last f32 bit: a one-ulp difference in inv_freq moves the angle pos*inv_freq by up to
    pos * 2^-23 (relative to inv_freq = 1), i.e. the table entry by ~2.5e-4 at pos = 2047 -- the
    same uncertainty separates either implementation from the reference's own (Rust powf + CUDA
    cosf).  Entries therefore agree to that absolute bound plus one storage rounding unit."""
    max_pos, d = 2048, 128
    cos, sin = np.zeros((max_pos, d // 2), np.uint16), np.zeros((max_pos, d // 2), np.uint16)
    sc = dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    rc = lib.atoma_rope_table(cos.ctypes.data, sin.ctypes.data, max_pos, d, 500000.0, 8.0 if scaled else 0.0, 1.0, 4.0,
                              8192, dtype)
    assert rc == 0
    rc_, rs_ = NR.rope_table(max_pos, d, 500000.0, dtype, sc if scaled else None)
    for got, ref in ((cos, rc_), (sin, rs_)):
        g, r = to_f32(got, dtype), to_f32(ref, dtype)
        eps = 2.0 ** -7 if dtype == BF16 else 2.0 ** -10   # one unit in the last place
        assert (np.abs(g - r) <= max_pos * 2.0 ** -22 + eps * np.abs(r)).all()
        assert (got != ref).mean() < 0.02


def test_split_heuristic_matches_oracle_exhaustively():
    """csrc/src/lib.rs:2122-2199 restated twice (C++ in the library, numpy in the oracle)."""
    rng = np.random.default_rng(0)
    for _ in range(400):
        bnm, sms, nb, mx = int(rng.integers(1, 3000)), int(rng.integers(1, 1024)), int(rng.integers(1, 300)), 128
        assert lib.atoma_num_splits_heuristic(bnm, sms, nb, mx) == A.num_splits_heuristic(bnm, sms, nb, mx)
    for b, h, d, sk, sq in [(1, 32, 128, 4096, 1), (64, 8, 128, 4096, 1), (256, 32, 128, 4096, 1), (4, 32, 64, 8192, 1),
                            (2, 16, 256, 1000, 7)]:
        assert lib.atoma_compute_num_splits(b, h, d, sk, sq, 256) == A.compute_num_splits(b, h, d, sk, sq, 256)


def test_linear_oracle_small_known_case_and_rounding():
    """oracle/linear_oracle.py: exactly accumulated product, one rounding (the small-batch projection's checker)."""
    import numpy as np
    from oracle import linear_oracle as LO
    from oracle.halfs import BF16, F16, from_f32, to_f32
    x = from_f32(np.array([[1.0, 2.0, -3.0, 0.5]], np.float32), BF16)
    w = from_f32(np.array([[1.0, 1.0, 1.0, 1.0], [0.5, -0.25, 2.0, 4.0], [0, 0, 0, 0]], np.float32), BF16)
    assert to_f32(LO.linear(x, w, BF16), BF16).tolist() == [[0.5, -4.0, 0.0]]
    # one rounding at the end: 256 + 1 is not representable in bf16 (8 significant bits) but the sum 257 + 255 = 512 is
    x2 = from_f32(np.array([[256.0, 1.0, 255.0]], np.float32), BF16)
    w2 = from_f32(np.ones((1, 3), np.float32), BF16)
    assert to_f32(LO.linear(x2, w2, BF16), BF16).tolist() == [[512.0]]
    rng = np.random.default_rng(0)
    xf, wf = rng.standard_normal((3, 64)).astype(np.float32), rng.standard_normal((5, 64)).astype(np.float32)
    got = to_f32(LO.linear(from_f32(xf, F16), from_f32(wf, F16), F16), F16)
    ref = to_f32(from_f32(xf, F16), F16).astype(np.float64) @ to_f32(from_f32(wf, F16), F16).astype(np.float64).T
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max()


def test_elementwise_oracle_known_values():
    import numpy as np
    from oracle import elementwise_oracle as EO
    from oracle.halfs import BF16, from_f32, to_f32
    table = from_f32(np.arange(12, dtype=np.float32).reshape(4, 3), BF16)
    assert to_f32(EO.embedding([2, 0, 2], table), BF16).tolist() == [[6, 7, 8], [0, 1, 2], [6, 7, 8]]
    a, b = from_f32(np.array([1.0, 256.0], np.float32), BF16), from_f32(np.array([2.0, 1.0], np.float32), BF16)
    assert to_f32(EO.add(a, b, BF16), BF16).tolist() == [3.0, 256.0]          # 257 rounds to even: 256
    g, u = from_f32(np.array([0.0, 20.0, -20.0], np.float32), BF16), from_f32(np.array([3.0, 2.0, 5.0], np.float32), BF16)
    out = to_f32(EO.silu_mul(g, u, BF16), BF16)
    assert out[0] == 0.0 and out[1] == 40.0 and abs(out[2]) < 1e-6            # silu(0)=0, silu(20)~20, silu(-20)~-4e-8
