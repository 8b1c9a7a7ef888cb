"""Pins the oracle: reference golden tables, reference equivalence properties, and agreement
of the two independent restatements (numpy and C).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import attn_oracle as A, cache_oracle as CO, norm_rope_oracle as NR
from oracle.halfs import F16, BF16, to_f32, from_f32
from util import rand_half, make_paged_cache, c_attention, oracle_c, assert_close

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = np.load(os.path.join(GOLD, "reference_tables.npz"))
CASES = np.load(os.path.join(GOLD, "oracle_cases.npz"))


def round4(x):  # to_vec3_round(t, 4): csrc/tests/flash_attn_tests.rs:5-17
    return np.round(x.astype(np.float32) * np.float32(1e4)) / np.float32(1e4)


def bsd(x):  # [heads, seq, d] -> [1, seq, heads, d] (the tests' transpose(1, 2))
    return np.ascontiguousarray(x.transpose(1, 0, 2))[None]


@pytest.mark.parametrize("mode", ["f32", "kernel"])
def test_g1_non_causal_table(mode):
    """csrc/tests/flash_attn_tests.rs:31-93 (flash_attn_acausal)."""
    out = A.flash_attn(bsd(REF["q"]), bsd(REF["k"]), bsd(REF["v"]), 0.5, False, F16, mode=mode)
    got = round4(to_f32(out, F16)[0].transpose(1, 0, 2))
    assert np.array_equal(got, REF["G1"])


@pytest.mark.parametrize("mode", ["f32", "kernel"])
def test_g2_causal_table(mode):
    """models/src/flash_attention.rs:632-705 (test_forward_with_varlen): causal since tokens > 1."""
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2)) for n in "qkv")  # [T=2, h=3, d]
    cu = np.array([0, 2], np.int32)
    out = A.flash_attn_varlen(q, k, v, cu, cu, 0.5, True, F16, mode=mode)
    got = round4(to_f32(out, F16).transpose(1, 0, 2))
    assert np.array_equal(got, REF["G2"])


def test_g1_through_varlen_and_kv_cache_entry_points():
    """flash_attn_tests.rs:95-138 (varlen) and :194-236 (kv_cache with seqlens_k=[2])."""
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2)) for n in "qkv")
    cu = np.array([0, 2], np.int32)
    out = A.flash_attn_varlen(q, k, v, cu, cu, 0.5, False, F16)
    assert np.array_equal(round4(to_f32(out, F16).transpose(1, 0, 2)), REF["G1"])
    out = A.flash_attn_kv_cache(q[None], k[None], v[None], 0.5, F16, seqlens_k=np.array([2]))
    assert np.array_equal(round4(to_f32(out, F16)[0].transpose(1, 0, 2)), REF["G1"])


def test_g1_g2_c_oracle():
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2)) for n in "qkv")  # [2,3,8]
    for causal, table in ((0, "G1"), (1, "G2")):
        o = c_attention(q, k, v, b=1, h=3, h_k=3, d=8, seqlen_q=2, seqlen_k=2, scale=0.5, is_bf16=0,
                        q_strides=(48, 24, 8), k_strides=(48, 24, 8), v_strides=(48, 24, 8),
                        o_shape=(2, 3, 8), o_strides=(48, 24, 8), causal=causal)
        assert np.array_equal(round4(to_f32(o, F16).transpose(1, 0, 2)), REF[table])


def test_p1_paged_equals_contiguous():
    """flash_attn_tests.rs:140-192: block-table K/V == the same rows laid out contiguously
    (batch-0 semantics; the test's batch 1 reads out of bounds, SURVEY B/Q10)."""
    base = np.arange(512, dtype=np.float32).astype(np.float16).reshape(32, 2, 8)
    mk = lambda c: (base * np.float16(1.0 / c)).astype(np.float16).view(np.uint16)
    q, k, v = mk(30), mk(40), mk(50)
    cu = np.array([0, 32], np.int32)
    bt = np.arange(2, dtype=np.int32).reshape(1, 2)
    paged = A.flash_attn_varlen(q, k.reshape(2, 16, 2, 8), v.reshape(2, 16, 2, 8), cu, cu, 0.5, False, F16,
                                block_table=bt)
    dense = A.flash_attn_varlen(q, k, v, cu, cu, 0.5, False, F16)
    assert np.array_equal(paged, dense)


def test_p2_kv_cache_equals_varlen_with_block_table():
    """flash_attn_tests.rs:238-303: 32 single-token decodes (seqlens_k = 1) == varlen with
    cu_seqlens 0..32, both through a [32, 2] block table.  The reference test indexes pages
    that do not exist (arange(64) over 2 pages); the in-bounds restatement uses 64 pages."""
    rng = np.random.default_rng(3)
    kc = rand_half(rng, (64, 16, 2, 8), F16)
    vc = rand_half(rng, (64, 16, 2, 8), F16)
    q = rand_half(rng, (32, 1, 2, 8), F16)
    bt = np.arange(64, dtype=np.int32).reshape(32, 2)
    a = A.flash_attn_kv_cache(q, kc, vc, 0.5, F16, bt, np.ones(32, np.int32))
    cu = np.arange(33, dtype=np.int32)
    b = A.flash_attn_varlen(q[:, 0], kc, vc, cu, cu, 0.5, False, F16, block_table=bt)
    assert np.array_equal(a[:, 0], b)


def test_committed_oracle_vectors_reproduce():
    """tests/golden/oracle_cases.npz was produced by this oracle: regenerate and compare."""
    c = CASES
    for tag, dt in (("d1", BF16), ("d2", F16)):
        for mode in ("f32", "kernel"):
            out = A.flash_attn_kv_cache(c[f"{tag}_q"], c[f"{tag}_kc"], c[f"{tag}_vc"], c[f"{tag}_scale"], dt,
                                        c[f"{tag}_bt"], c[f"{tag}_lens"], mode=mode)
            assert np.array_equal(out, c[f"{tag}_out_{mode}"])
    out = A.flash_attn_varlen(c["p1_q"], c["p1_k"], c["p1_v"], c["p1_cu"], c["p1_cu"], c["p1_scale"], True, BF16)
    assert np.array_equal(out, c["p1_out_f32"])
    assert np.array_equal(NR.rms_norm(c["n1_x"], c["n1_w"], 1e-5, BF16), c["n1_y"])
    assert np.array_equal(NR.rope(c["o1_x"], c["o1_cos"], c["o1_sin"], c["o1_pos"], BF16), c["o1_y"])


def test_zero_length_sequence_gives_zero_output():
    """flash_fwd_kernel.h:97-133,543-582: no visible key -> O = 0."""
    c = CASES
    assert int(c["d1_lens"][0]) == 0
    assert not c["d1_out_f32"][0].any() and not c["d1_out_kernel"][0].any()


@pytest.mark.parametrize("dtype", [F16, BF16])
def test_numpy_and_c_oracles_agree_paged_decode(dtype):
    rng = np.random.default_rng(11 + dtype)
    lens = np.array([1, 31, 64, 100], np.int32)
    h, hk, d, page = 8, 2, 64, 16
    kc, vc, bt = make_paged_cache(rng, 20, page, hk, d, dtype, lens)
    q = rand_half(rng, (len(lens), 1, h, d), dtype)
    sc = np.float32(d ** -0.5)
    ref = A.flash_attn_kv_cache(q, kc, vc, sc, dtype, bt, lens)
    got = c_attention(q, kc, vc, b=len(lens), h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page,
                      scale=float(sc), is_bf16=dtype, q_strides=(h * d, h * d, d),
                      k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
                      o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens, k_cumulative=False,
                      block_table=bt, page=page)
    # two f32 summation orders, then one rounding: at most one unit in the last place apart
    assert_close(got, ref, dtype, atol=1e-6, what="numpy vs C oracle")


def test_numpy_and_c_oracles_agree_causal_varlen():
    c = CASES
    got = c_attention(c["p1_q"], c["p1_k"], c["p1_v"], b=3, h=4, h_k=2, d=64, seqlen_q=52, seqlen_k=52,
                      scale=float(c["p1_scale"]), is_bf16=1, q_strides=(0, 256, 64), k_strides=(0, 128, 64),
                      v_strides=(0, 128, 64), o_shape=c["p1_q"].shape, o_strides=(0, 256, 64), causal=1,
                      cu_q=c["p1_cu"], cu_k=c["p1_cu"])
    assert_close(got, c["p1_out_f32"], BF16, atol=1e-6, what="numpy vs C oracle (causal varlen)")


def test_split_kv_merge_is_exact_reassociation():
    """Split partials + LSE merge (flash_fwd_kernel.h:1204-1236) == the unsplit result."""
    rng = np.random.default_rng(5)
    lens = np.array([700, 129, 0], np.int32)
    kc, vc, bt = make_paged_cache(rng, 60, 16, 2, 64, BF16, lens)
    q = rand_half(rng, (3, 1, 4, 64), BF16)
    # f32 mode: a pure f32 reassociation -> one rounding unit; kernel mode (P rounded to bf16
    # against a per-split max) moves each term by 2^-9 relative -> absolute tolerance 1e-3
    for mode, atol in (("f32", 1e-6), ("kernel", 1e-3)):
        one = A.flash_attn_kv_cache(q, kc, vc, 0.125, BF16, bt, lens, mode=mode)
        for splits in (2, 3, 7):
            many = A.flash_attn_kv_cache(q, kc, vc, 0.125, BF16, bt, lens, mode=mode, num_splits=splits)
            assert_close(many, one, BF16, atol=atol, what=f"{mode} {splits} splits")


def test_p3_cache_ops_bit_exact_properties():
    """cache_manager_tests.rs:24-62 (swap), :230-239,242-347 (copy), :553-616 (reshape)."""
    rng = np.random.default_rng(9)
    # swap: dst[d] == src[s]; untouched dst blocks unchanged
    src = rand_half(rng, (3, 16, 2, 8), F16)
    dst = rand_half(rng, (3, 16, 2, 8), F16)
    before = dst.copy()
    CO.swap_blocks(src, dst, {0: 2, 1: 0})
    assert np.array_equal(dst[2], src[0]) and np.array_equal(dst[0], src[1]) and np.array_equal(dst[1], before[1])
    # copy: mapping [[0,2],[1,3]] on 2 layers
    ks = [rand_half(rng, (4, 64, 2, 8), BF16) for _ in range(2)]
    vs = [rand_half(rng, (4, 64, 2, 8), BF16) for _ in range(2)]
    k0 = [k.copy() for k in ks]
    CO.copy_blocks(ks, vs, [[0, 2], [1, 3]])
    for l in range(2):
        assert np.array_equal(ks[l][2], k0[l][0]) and np.array_equal(ks[l][3], k0[l][1])
        assert np.array_equal(ks[l][:2], k0[l][:2])
    # reshape_and_cache_flash fixture
    c = CASES
    for i in range(10):
        assert np.array_equal(c["r1_kcache"][i // 8, i % 8], c["r1_key"][i])
        assert np.array_equal(c["r1_vcache"][i // 8, i % 8], c["r1_val"][i])


def test_c_cache_oracles_match_numpy():
    lib = oracle_c()
    rng = np.random.default_rng(13)
    T, hk, d, page, nb = 37, 2, 32, 16, 6
    big = rand_half(rng, (T, hk * d + 16), BF16)               # row stride > hk*d
    key, val = big[:, : hk * d].reshape(T, hk, d), rand_half(rng, (T, hk, d), BF16)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    slots[[3, 20]] = -1
    kc1, vc1 = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    kc2, vc2 = kc1.copy(), vc1.copy()
    CO.reshape_and_cache_flash(key, val, kc1, vc1, slots)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.oracle_reshape_and_cache_flash.argtypes = [C.c_void_p] * 5 + [C.c_int64] * 7
    lib.oracle_reshape_and_cache_flash(vp(big), vp(val), vp(kc2), vp(vc2), vp(slots), page * hk * d, T, hk, d, page,
                                       big.shape[1], hk * d)
    assert np.array_equal(kc1, kc2) and np.array_equal(vc1, vc2)


def test_num_splits_heuristic_examples():
    """csrc/src/lib.rs:2122-2199 (doc comment: 48 batch*heads on 108 SMs -> 2 splits)."""
    assert A.num_splits_heuristic(48, 108, 64, 128) == 2
    assert A.num_splits_heuristic(256 * 32, 512, 32, 128) == 1       # C2a: enough CTAs, no split
    assert A.compute_num_splits(1, 32, 128, 4096, 1, 256) >= 8        # bs=1 decode splits
    assert A.compute_num_splits(256, 32, 128, 4096, 1, 256) == 1


def test_rope_table_llama3_scaling_monotone_and_bounded():
    cos, sin = NR.rope_table(32, 128, 500000.0, BF16, dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                                            original_max_position_embeddings=8192))
    c, s = to_f32(cos, BF16), to_f32(sin, BF16)
    assert c.shape == (32, 64) and np.all(c[0] == 1) and np.all(s[0] == 0)
    assert np.abs(c * c + s * s - 1).max() < 2 ** -6
    f = NR.inv_freq(128, 500000.0)
    fs = NR.inv_freq(128, 500000.0, dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                         original_max_position_embeddings=8192))
    assert np.all(fs <= f) and np.allclose(fs[:16], f[:16]) and np.allclose(fs[-1], f[-1] / 8)


def test_c_grouped_decode_equals_the_plain_restatement():
    """oracle_decode_grouped (what bench.py times on the host cores: KV rows converted once per query group, vectorised
    loops) computes the same fa_acausal result as oracle_attention; only the order of the sums inside a dot product differs,
    so after the one rounding the two agree bit for bit except for boundary cases, never by more than one unit."""
    import ctypes as C
    from util import oracle_c, rand_half, make_paged_cache, c_attention
    lib = oracle_c()
    rng = np.random.default_rng(6)
    B, h, hk, d, page = 5, 8, 2, 128, 16
    lens = np.array([0, 1, 33, 200, 517], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 2
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    q = rand_half(rng, (B, 1, h, d), BF16)
    sc = float(d ** -0.5)
    ref = c_attention(q, kc, vc, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, scale=sc, is_bf16=1,
                      q_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
                      o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens, k_cumulative=False, block_table=bt, page=page)
    out = np.zeros_like(q)
    i64 = C.c_int64
    lib.oracle_decode_grouped.argtypes = [C.c_void_p] * 6 + [i64, C.c_int, i64, i64, i64] + [C.c_int] * 4 + [C.c_float, C.c_int]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    btc, lc = np.ascontiguousarray(bt, np.int32), np.ascontiguousarray(lens, np.int32)
    lib.oracle_decode_grouped(vp(q), vp(kc), vp(vc), vp(out), vp(lc), vp(btc), btc.shape[1], page, page * hk * d, hk * d, d, B, h, hk, d, sc, 2)
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01
    assert not out[0].any()
