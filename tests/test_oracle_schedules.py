"""Why the attention tolerance has two regimes, shown on the CPU with the oracle alone.

The reference kernel rounds p = exp2(s - m) to bf16 before P.V (softmax.h:65-91 + the bf16 MMA).  WHICH running max m a
key is rounded against depends on the kernel's online-softmax schedule (tile size, which accumulator owns which key), so
two faithful kernels differ from each other -- and from fa_acausal -- by up to 2^-9 |v| per output on rows with few keys,
and by much less once the mass is spread over hundreds of keys.  Consequences used by the GPU tests:
  * against the f32 oracle and against any *other* schedule: atol 4e-3 (bf16) below 512 keys, 1e-3 from 512 keys on;
  * against the kernel's *own* schedule (attend_decode_online with DECODE_SCHEDULES): 1e-3 + 1 ulp at every length."""
import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle.halfs import BF16, F16, to_f32, from_f32
from util import rand_half

SCHEDULES = {"global max (mode=kernel)": (1 << 30, 1, lambda j: 0), "reference split-KV tile 128": (128, 1, lambda j: 0),
             "hip dot2 d=128": (16, 4, lambda j: j % 4), "hip mqk": (16, 4, lambda j: j // 4)}


def run(qf, kf, vf, sc, dtype, name):
    tile, ng, gof = SCHEDULES[name]
    return A.attend_decode_online(qf, kf, vf, sc, dtype, tile, ng, gof)


@pytest.mark.parametrize("dtype", [BF16, F16])
def test_global_max_schedule_is_mode_kernel(dtype):
    rng = np.random.default_rng(1)
    q, k, v = (to_f32(rand_half(rng, s, dtype), dtype) for s in ((8, 128), (300, 2, 128), (300, 2, 128)))
    sc = np.float32(128 ** -0.5)
    ref, _ = A.attend_rows(q[None], k, v, sc, mode="kernel", dtype=dtype)
    got = run(q, k, v, sc, dtype, "global max (mode=kernel)")
    assert np.abs(got - ref[0]).max() < 2e-5        # same arithmetic; f32 summation order and the odd p that rounds the other way


def test_schedules_differ_by_the_p_rounding_bound_on_short_rows_only():
    rng = np.random.default_rng(2)
    sc = np.float32(128 ** -0.5)
    worst_short, worst_long = 0.0, 0.0
    for trial in range(20):
        for L, short in ((3, True), (20, True), (2048, False)):
            q, k, v = (to_f32(rand_half(rng, s, BF16), BF16) for s in ((8, 128), (L, 2, 128), (L, 2, 128)))
            outs = [run(q, k, v, sc, BF16, n) for n in SCHEDULES]
            f32, _ = A.attend_rows(q[None], k, v, sc, mode="f32")
            spread = max(np.abs(a - b).max() for a in outs for b in outs)
            vs_f32 = max(np.abs(a - f32[0]).max() for a in outs)
            bound = 2.0 ** -9 * np.abs(v).max() * 1.01 + 1e-6
            assert spread <= 2 * bound and vs_f32 <= bound, (L, spread, vs_f32, bound)
            if short:
                worst_short = max(worst_short, spread)
            else:
                worst_long = max(worst_long, spread, vs_f32)
    assert worst_short > 1e-3, "few keys: two faithful schedules are expected to disagree by more than 1e-3 before the output rounding"
    assert worst_long < 5e-4, "thousands of keys: every schedule is well inside 1e-3"


def test_online_oracle_entry_point_matches_f32_mode_within_policy():
    from util import make_paged_cache, assert_close, attn_atol
    rng = np.random.default_rng(3)
    lens = np.array([0, 1, 2, 15, 16, 17, 100, 600], np.int32)
    for d, variants in ((128, ("dot2", "mqk")), (64, ("dot2",))):
        kc, vc, bt = make_paged_cache(rng, 80, 16, 2, d, BF16, lens)
        q = rand_half(rng, (len(lens), 1, 4, d), BF16)
        sc = np.float32(d ** -0.5)
        ref = A.flash_attn_kv_cache(q, kc, vc, sc, BF16, bt, lens)
        for var in variants:
            got = A.flash_attn_kv_cache_online(q, kc, vc, sc, BF16, bt, lens, var)
            assert not got[0].any()
            for i, L in enumerate(lens):
                assert_close(got[i], ref[i], BF16, atol=attn_atol(BF16, int(L)), what=f"{var} d={d} L={L}")


def test_prefill_online_schedule_restates_the_definition():
    """attend_prefill_online (64-key tiles, deferred raise of the running max) against the f32 definition: inside the P-rounding
    bound on every row, identical to the one-tile kernel mode when the whole row fits one tile, and independent of the deferral
    threshold up to that bound (threshold 0 = the textbook online softmax)."""
    rng = np.random.default_rng(5)
    L, h, hk, d = 300, 4, 2, 64
    q, k, v = (to_f32(from_f32(rng.standard_normal(s).astype(np.float32), BF16), BF16) for s in ((L, h, d), (L, hk, d), (L, hk, d)))
    scale = np.float32(d ** -0.5)
    ref, _ = A.attend_rows(q, k, v, scale, causal=True)
    bound = 2.0 ** -9 * np.abs(v).max() + 1e-5
    for defer in (0.0, 8.0, 1e9):
        o = A.attend_prefill_online(q, k, v, scale, True, BF16, defer=defer)
        assert np.abs(o - ref).max() <= bound, defer
    one_tile, _ = A.attend_rows(q[:40], k[:40], v[:40], scale, causal=True, mode="kernel", dtype=BF16)
    o = A.attend_prefill_online(q[:40], k[:40], v[:40], scale, True, BF16, tile=64, defer=1e9)
    # one tile, max taken once (the first tile always raises): the kernel-mode restatement up to the f32 order of the sums
    assert np.abs(o - one_tile).max() < 2e-6
