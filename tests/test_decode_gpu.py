"""Paged decode attention (run_mha with seqlen_q = 1) against the oracle, through the C ABI."""
import os

import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle.halfs import F16, BF16, to_f32
from util import rand_half, make_paged_cache, assert_close, c_attention, ATOL_VS_F32, attn_atol, poison_unwritten_slots

pytestmark = pytest.mark.gpu
CASES = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_cases.npz"))


def gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype, causal=1, alibi=None, seqlens_cumulative=False):
    """q [B,1,h,d]; paged caches [nb,page,hk,d] (bt given) or contiguous [B,S,hk,d] (bt None)."""
    B, _, h, d = q.shape
    hk = kc.shape[2]
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc))
    do = gpu.DeviceBuffer(q.nbytes)
    do.fill_bytes(0xFF)                                          # poison: every element must be written
    dlse = gpu.DeviceBuffer.zeros((B, h), np.float32)
    dbt = gpu.DeviceBuffer.from_numpy(np.ascontiguousarray(bt, np.int32)) if bt is not None else None
    dl = gpu.DeviceBuffer.from_numpy(np.ascontiguousarray(lens, np.int32)) if lens is not None else None
    da = gpu.DeviceBuffer.from_numpy(np.asarray(alibi, np.float32)) if alibi is not None else None
    page = kc.shape[1] if bt is not None else 0
    seqlen_k = bt.shape[1] * page if bt is not None else kc.shape[1]
    gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=seqlen_k, softmax_scale=float(scale),
                is_bf16=dtype, q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d),
                k_strides=(kc.shape[1] * hk * d, hk * d, d), v_strides=(vc.shape[1] * hk * d, hk * d, d),
                is_causal=0 if alibi is None else causal,      # lib.rs:1629-1631: causal off for seqlen_q == 1
                cu_seqlens_k=dl, is_seqlens_k_cumulative=seqlens_cumulative, block_table=dbt,
                block_table_batch_stride=0 if bt is None else bt.shape[1], page_block_size=page,
                alibi_slopes=da, softmax_lse=dlse, force_split_kernel=bt is not None, unpadded_lse=False)
    gpu.synchronize()
    return do.numpy(np.uint16, q.shape), dlse.numpy()


@pytest.mark.parametrize("tag,dtype", [("d1", BF16), ("d2", F16)])
def test_decode_golden_fixtures(gpu, tag, dtype):
    c = CASES
    out, _ = gpu_decode(gpu, c[f"{tag}_q"], c[f"{tag}_kc"], c[f"{tag}_vc"], c[f"{tag}_bt"], c[f"{tag}_lens"],
                        c[f"{tag}_scale"], dtype)
    for i, L in enumerate(c[f"{tag}_lens"]):
        for mode in ("f32", "kernel"):
            assert_close(out[i], c[f"{tag}_out_{mode}"][i], dtype, atol=attn_atol(dtype, L),
                         what=f"{tag} seq {i} (L={L}) vs {mode} oracle")


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("d,h,hk,page", [(128, 32, 8, 16), (128, 8, 8, 16), (128, 16, 2, 16), (128, 6, 2, 32),
                                         (64, 32, 8, 16), (64, 4, 4, 64), (128, 32, 2, 16), (64, 16, 8, 16)])
def test_decode_matches_oracle_ragged(gpu, dtype, d, h, hk, page):
    """GQA group sizes 1,2,3,4,8,16, both head sizes, pages of 16/32/64 tokens, ragged lengths
    around every tile/page boundary, including the empty sequence."""
    rng = np.random.default_rng(d + h * 7 + hk + page)
    lens = np.array([0, 1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 200, 333], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 3
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (len(lens), 1, h, d), dtype)
    scale = np.float32(d ** -0.5)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype)
    for mode in ("f32", "kernel"):
        ref = A.flash_attn_kv_cache(q, kc, vc, scale, dtype, bt, lens, mode=mode)
        for i, L in enumerate(lens):
            assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"decode L={L} vs {mode} oracle")
    assert not out[0].any(), "empty sequence must produce exact zeros (flash_fwd_kernel.h:97-133)"
    assert np.isposinf(lse[0]).all() and np.isfinite(lse[1:]).all()


def test_decode_lse_values(gpu):
    rng = np.random.default_rng(77)
    lens = np.array([40, 129], np.int32)
    kc, vc, bt = make_paged_cache(rng, 16, 16, 2, 128, BF16, lens)
    q = rand_half(rng, (2, 1, 8, 128), BF16)
    sc = np.float32(128 ** -0.5)
    _, lse = gpu_decode(gpu, q, kc, vc, bt, lens, sc, BF16)
    for b in range(2):
        kb = A.gather_paged(to_f32(kc, BF16), bt[b], lens[b], 16)
        _, want = A.attend_rows(to_f32(q, BF16)[b], kb, kb, sc)
        assert np.allclose(lse[b], want[:, 0], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,L", [(1, 4096), (2, 1500), (3, 257)])
def test_decode_split_kv_small_batch(gpu, B, L):
    """Few sequences, long context: the library splits KV across wavefronts and merges with the
    combine kernel ('paged_attention v2'); result must still match the unsplit oracle."""
    rng = np.random.default_rng(B * 1000 + L)
    lens = np.array([L - 7 * i for i in range(B)], np.int32)
    h, hk, d, page = 32, 8, 128, 16
    nb = int(sum((x + page - 1) // page for x in lens)) + 2
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    q = rand_half(rng, (B, 1, h, d), BF16)
    scale = np.float32(d ** -0.5)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, scale, BF16)
    ref = c_attention(q, kc, vc, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page,
                      scale=float(scale), is_bf16=1, q_strides=(h * d, h * d, d),
                      k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
                      o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens, k_cumulative=False,
                      block_table=bt, page=page)
    assert_close(out, ref, BF16, atol=1e-3, what="split-KV decode vs C oracle (f32)")
    assert np.isfinite(lse).all()


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,L,h,hk,d", [(3, 2048, 8, 1, 128), (2, 4096, 32, 8, 128), (64, 1024, 8, 1, 128), (5, 3000, 16, 8, 64), (9, 700, 5, 1, 128),
                                         (40, 600, 8, 8, 128)])
def test_decode_split_kv_merged_inside_the_launch(gpu, dtype, B, L, h, hk, d):
    """paged_decode_wg_kernel (option decode_wg_merge): workgroups of 2 / 4 / 8 wavefronts merge their KV pieces in LDS; with 3 = forced
    also the last-arriver merge across the workgroups of one sequence (agent-scope partials + arrival counter).  Ragged lengths incl. an
    empty sequence, matrix-core (groups of 5, 8) and dot2 (groups of 1, 2, 4) score kernels, d = 64 / 128; every mode against the
    oracle, the LSE output against the two-kernel route, and twice in a row (the counters must come back to zero)."""
    rng = np.random.default_rng(B * 131 + L + h)
    lens = np.array([max(0, L - 37 * i) for i in range(B)], np.int32)
    lens[B // 2] = 0
    page = 16
    nb = int(sum((x + page - 1) // page for x in lens)) + 2
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (B, 1, h, d), dtype)
    scale = np.float32(d ** -0.5)
    ref = _oracle_decode(q, kc, vc, bt, lens, dtype)
    outs = {}
    try:
        for mode in (0, 1, 3, 3):
            assert gpu.lib.atoma_set_option(b"decode_wg_merge", mode) == 0
            out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype)
            for i, Li in enumerate(lens):
                assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, int(Li)), what=f"decode_wg_merge={mode} seq {i} (L={Li})")
            assert not out[B // 2].any() and np.isinf(lse[B // 2]).all()
            outs.setdefault(mode, []).append((out, lse))
    finally:
        gpu.lib.atoma_set_option(b"decode_wg_merge", 1)
    assert np.array_equal(outs[3][0][0], outs[3][1][0]), "the second forced launch differs: arrival counters not back at zero?"
    live = lens > 0
    assert np.allclose(outs[3][0][1][live], outs[0][0][1][live], rtol=1e-5, atol=1e-5)


def _oracle_decode(q, kc, vc, bt, lens, dtype):
    B, _, h, d = q.shape
    hk, page = kc.shape[2], kc.shape[1]
    return c_attention(q, kc, vc, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, scale=d ** -0.5,
                       is_bf16=dtype, q_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                       v_strides=(page * hk * d, hk * d, d), o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens,
                       k_cumulative=False, block_table=bt, page=page)


@pytest.mark.parametrize("kind", ["ragged", "straggler", "more_than_resident", "mostly_empty"])
def test_decode_large_batch_balanced_mode(gpu, kind):
    """A batch big enough to fill the chip without KV splitting (B * h_k > 1024): the kernel lays all tiles of the
    batch on one line and gives every wavefront the same share; sequences cut between wavefronts are merged inside the
    launch by the last wavefront to arrive (option decode_line_merge = 0: by the combine kernel), whole ones are written
    directly, empty ones by the wavefronts of the line (or the combine kernel).  Same answer as the oracle,
    and as the one-wavefront-per-sequence route (option decode_stream = 0)."""
    rng = np.random.default_rng(99)
    h, hk, d, page = 4, 4, 64, 16
    if kind == "ragged":
        B = 272
        lens = rng.integers(16, 3000, B).astype(np.int32)
        lens[:6] = [0, 1, 1024, 1025, 16, 17]            # empty / single token / tile boundaries
    elif kind == "straggler":
        B = 300
        lens = rng.integers(1, 200, B).astype(np.int32)
        lens[123] = 20000                                   # one sequence as long as the rest of the batch together
    elif kind == "more_than_resident":
        B = 600                                             # 2400 (sequence, kv head) pairs > 2048 resident wavefronts, uniform
        lens = np.full(B, 700, np.int32)
    else:
        B = 272
        lens = np.zeros(B, np.int32)
        lens[[3, 100, 271]] = [5000, 33, 1]
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    q = rand_half(rng, (B, 1, h, d), BF16)
    ref = _oracle_decode(q, kc, vc, bt, lens, BF16)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    out_again, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    assert np.array_equal(out, out_again), "balanced mode is deterministic"
    for _ in range(3):      # cut sequences are merged by the LAST wavefront to arrive: whoever that is, the bits are the same
        again, lse_again = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
        assert np.array_equal(out, again) and np.array_equal(lse, lse_again, equal_nan=True)
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], BF16, atol=attn_atol(BF16, L), what=f"{kind} batch seq {i} (L={L})")
    empty = lens == 0
    assert gpu.lib.atoma_set_option(b"decode_line_merge", 0) == 0               # the same pieces merged by decode_combine_kernel
    try:
        out_ck, lse_ck = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    finally:
        assert gpu.lib.atoma_set_option(b"decode_line_merge", 1) == 0
    assert np.array_equal(out[empty], out_ck[empty])
    for i, L in enumerate(lens):
        assert_close(out_ck[i], ref[i], BF16, atol=attn_atol(BF16, L), what=f"{kind} batch (combine kernel) seq {i} (L={L})")
    assert (out != out_ck).mean() < 2e-3, "the two merges differ only in the summation order of three or more pieces"
    np.testing.assert_allclose(lse[~empty], lse_ck[~empty], rtol=0, atol=1e-5)
    assert not out[empty].any() and np.isposinf(lse[empty]).all() and np.isfinite(lse[~empty]).all()
    assert gpu.lib.atoma_set_option(b"decode_stream", 0) == 0
    try:
        out2, lse2 = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)        # one wavefront per (sequence, kv head)
    finally:
        assert gpu.lib.atoma_set_option(b"decode_stream", 1) == 0
    for i, L in enumerate(lens):
        assert_close(out2[i], ref[i], BF16, atol=attn_atol(BF16, L), what=f"{kind} batch (unbalanced route) seq {i} (L={L})")
    np.testing.assert_allclose(lse[~empty], lse2[~empty], rtol=0, atol=2e-3)


@pytest.mark.parametrize("uniform", [False, True], ids=["ragged", "uniform"])
@pytest.mark.parametrize("mqk", [29, 5], ids=["mfma-on-the-line", "dot2"])
@pytest.mark.parametrize("dtype,h,hk", [(BF16, 32, 8), (F16, 6, 2), (BF16, 4, 2)])
def test_decode_small_groups_on_the_balanced_line(gpu, dtype, h, hk, mqk, uniform):
    """Groups of 2..4 q heads at d = 128 with several kv heads: every resident launch takes the kv-head-major line (uniform batches
    too), by default with the matrix-core kernel (bit 3 of decode_mqk), with the dot2 kernel when that bit is off.  Both against the
    oracle; the dispatcher's choice is read back."""
    rng = np.random.default_rng(17 + h)
    d, page = 128, 16
    B = 1100 // hk + 3
    lens = np.full(B, 300, np.int32) if uniform else rng.integers(1, 400, B).astype(np.int32)
    if not uniform:
        lens[5], lens[B // 2] = 0, 3000
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (B, 1, h, d), dtype)
    ref = _oracle_decode(q, kc, vc, bt, lens, dtype)
    with _options(gpu, decode_mqk=mqk):
        out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
        name = gpu.lib.atoma_last_decode_kernel().decode()
    assert ("paged_decode_mqk_kernel" in name) == (mqk == 29) and "balanced" in name, name
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"seq {i} (L={L})")
    if not uniform:
        assert not out[5].any() and np.isposinf(lse[5]).all()
    with _options(gpu, decode_mqk=mqk, decode_stream=3):      # the per-sequence order for uniform batches (dot2 kernel): same numbers up to the tolerance
        out3, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    for i, L in enumerate(lens):
        assert_close(out3[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"decode_stream=3 seq {i} (L={L})")


@pytest.mark.parametrize("dtype,B,h,hk", [(BF16, 141, 32, 8), (F16, 400, 16, 4), (BF16, 257, 32, 8), (BF16, 250, 16, 8)])
def test_decode_two_sequences_per_workgroup_option(gpu, dtype, B, h, hk):
    """decode_pair (round 6, opt-in): a workgroup of 8 wavefronts takes the i-th shortest and the i-th longest sequence (ranked on the device) and
    4 kv heads, every (sequence, kv head) in two halves merged in LDS -- against the oracle on a ragged batch with empty sequences, equal lengths
    (the rank's tie-break), an odd batch (the middle sequence pairs with itself) and the lengths' extremes."""
    rng = np.random.default_rng(B + h)
    d, page = 128, 16
    lens = rng.integers(1, 700, B).astype(np.int32)
    lens[0] = 0
    lens[B // 2] = 2100
    if B > 10:
        lens[3:7] = 333
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (B, 1, h, d), dtype)
    ref = _oracle_decode(q, kc, vc, bt, lens, dtype)
    with _options(gpu, decode_pair=2):
        out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
        name = gpu.lib.atoma_last_decode_kernel().decode()
    assert "paged_decode_pair_kernel" in name, name
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"seq {i} (L={L})")
    assert not out[0].any() and np.isposinf(lse[0]).all()
    # a batch of EQUAL lengths through the same option: one unit per wavefront, written directly (no halves, no merge)
    lens_u = np.full(B, 333, np.int32)
    kcu, vcu, btu = make_paged_cache(rng, B * 21 + 1, page, hk, d, dtype, lens_u)
    ref_u = _oracle_decode(q, kcu, vcu, btu, lens_u, dtype)
    with _options(gpu, decode_pair=2):
        out_u, _ = gpu_decode(gpu, q, kcu, vcu, btu, lens_u, d ** -0.5, dtype)
        assert "paged_decode_pair_kernel" in gpu.lib.atoma_last_decode_kernel().decode()
    for i in range(B):
        assert_close(out_u[i], ref_u[i], dtype, atol=attn_atol(dtype, 333), what=f"uniform batch, seq {i}")
    out_d, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)          # default: only on the hint of a batch packed by atoma_prepare_inputs
    assert "pair" not in gpu.lib.atoma_last_decode_kernel().decode()
    if B * hk >= 256 * 6 and B * hk <= 256 * 8:                                # one unit per resident wavefront: the hint decides
        assert gpu.lib.atoma_hint_decode_lengths(int(lens.min()), int(lens.max()), B) == 0
        out_h, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
        assert "pair" in gpu.lib.atoma_last_decode_kernel().decode() and np.array_equal(out_h, out)
        assert gpu.lib.atoma_hint_decode_lengths(333, 333, B) == 0             # "all equal": the line kernel
        gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
        assert "pair" not in gpu.lib.atoma_last_decode_kernel().decode()
        assert gpu.lib.atoma_hint_decode_lengths(5, 3, B) == -1 and "min_len" in gpu.last_error()
    for i, L in enumerate(lens):
        assert_close(out[i], out_d[i], dtype, atol=attn_atol(dtype, L), what=f"pair vs line, seq {i}")


@pytest.mark.parametrize("B,L,h,hk,d", [(16, 3000, 32, 8, 128), (3, 5000, 16, 2, 128), (40, 900, 8, 8, 64), (7, 2000, 64, 8, 128)])
def test_decode_workgroup_order_does_not_change_the_bits(gpu, B, L, h, hk, d):
    """Split-KV and small resident launches: kv head slowest (default: the wavefronts that share a CU are the kv heads of one piece)
    against kv head fastest -- the same pieces, computed somewhere else."""
    rng = np.random.default_rng(B + L)
    page = 16
    lens = rng.integers(L // 2, L + 1, B).astype(np.int32)
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    q = rand_half(rng, (B, 1, h, d), BF16)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    with _options(gpu, decode_head_major=0):
        out0, lse0 = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    assert np.array_equal(out, out0) and np.array_equal(lse, lse0)
    ref = _oracle_decode(q, kc, vc, bt, lens, BF16)
    for i, Li in enumerate(lens):
        assert_close(out[i], ref[i], BF16, atol=attn_atol(BF16, Li), what=f"seq {i} (L={Li})")


@pytest.mark.parametrize("dtype,h,hk", [(BF16, 16, 2), (F16, 8, 8), (BF16, 12, 2)])
def test_decode_balanced_mode_head_layouts(gpu, dtype, h, hk):
    """Balanced mode with the matrix-core kernel (groups of 8 and 6), MHA, f16, d = 128, page 32, ALiBi off."""
    rng = np.random.default_rng(7)
    d, page = 128, 32
    B = 1100 // hk + 3
    lens = rng.integers(1, 400, B).astype(np.int32)
    lens[5], lens[B // 2] = 0, 4000
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (B, 1, h, d), dtype)
    ref = _oracle_decode(q, kc, vc, bt, lens, dtype)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"seq {i} (L={L})")
    assert not out[5].any() and np.isposinf(lse[5]).all()


def test_decode_balanced_mode_cumulative_varlen(gpu):
    """Balanced mode over a contiguous varlen K/V addressed by cumulative cu_seqlens_k (block_info.h:16-23)."""
    rng = np.random.default_rng(8)
    B, h, hk, d = 300, 4, 4, 64
    lens = rng.integers(0, 300, B).astype(np.int32)
    lens[17] = 3000
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    k, v = rand_half(rng, (int(cu[-1]), hk, d), BF16), rand_half(rng, (int(cu[-1]), hk, d), BF16)
    q = rand_half(rng, (B, 1, h, d), BF16)
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v))
    do = gpu.DeviceBuffer(q.nbytes)
    do.fill_bytes(0xFF)
    dcu = gpu.DeviceBuffer.from_numpy(cu)
    gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=3000, softmax_scale=0.125, is_bf16=1,
                q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(0, hk * d, d),
                v_strides=(0, hk * d, d), cu_seqlens_k=dcu, is_seqlens_k_cumulative=True, unpadded_lse=False)
    gpu.synchronize()
    out = do.numpy(np.uint16, q.shape)
    for b in range(B):
        if lens[b] == 0:
            assert not out[b].any()
            continue
        ref = A.flash_attn_kv_cache(q[b:b + 1], k[cu[b]: cu[b + 1]][None], v[cu[b]: cu[b + 1]][None], 0.125, BF16)
        assert_close(out[b:b + 1], ref, BF16, atol=attn_atol(BF16, lens[b]), what=f"cumulative lens seq {b}")


def test_decode_contiguous_cache_without_block_table(gpu):
    """flash_attn_kv_cache_full with block_table = None: caches [B, S, hk, d], per-sequence
    lengths; the ragged last tile must not read past the cache (rows are clamped)."""
    rng = np.random.default_rng(8)
    B, S, h, hk, d = 5, 100, 8, 2, 128
    kc, vc = rand_half(rng, (B, S, hk, d), BF16), rand_half(rng, (B, S, hk, d), BF16)
    q = rand_half(rng, (B, 1, h, d), BF16)
    lens = np.array([100, 1, 50, 99, 16], np.int32)
    out, _ = gpu_decode(gpu, q, kc, vc, None, lens, 0.09, BF16)
    ref = A.flash_attn_kv_cache(q, kc, vc, 0.09, BF16, None, lens)
    assert_close(out, ref, BF16, atol=ATOL_VS_F32[BF16], what="contiguous kv cache")
    out, _ = gpu_decode(gpu, q, kc, vc, None, None, 0.09, BF16)      # seqlens_k = None -> full S
    assert_close(out, A.flash_attn_kv_cache(q, kc, vc, 0.09, BF16), BF16, atol=ATOL_VS_F32[BF16], what="contiguous kv cache, full length")


def test_decode_alibi(gpu):
    rng = np.random.default_rng(9)
    lens = np.array([70, 33], np.int32)
    kc, vc, bt = make_paged_cache(rng, 10, 16, 2, 128, BF16, lens)
    q = rand_half(rng, (2, 1, 8, 128), BF16)
    slopes = (2.0 ** -np.arange(1, 9)).astype(np.float32)
    out, _ = gpu_decode(gpu, q, kc, vc, bt, lens, 0.088, BF16, alibi=slopes)
    ref = A.flash_attn_kv_cache(q, kc, vc, 0.088, BF16, bt, lens, causal=True, alibi_slopes=slopes)
    assert_close(out, ref, BF16, atol=ATOL_VS_F32[BF16], what="decode + ALiBi")


def test_decode_block_table_permutation_invariance_bit_exact(gpu):
    """Size-independent property: where the pages physically live must not change a single
    bit of the result (same rows, same order of arithmetic)."""
    rng = np.random.default_rng(10)
    lens = np.array([513, 1000, 64], np.int32)
    h, hk, d, page = 16, 4, 128, 16
    need = [(int(L) + page - 1) // page for L in lens]
    nb = sum(need)
    kc0, vc0, bt0 = make_paged_cache(rng, nb, page, hk, d, BF16, lens, shuffle=False)
    q = rand_half(rng, (3, 1, h, d), BF16)
    perm = rng.permutation(nb)                               # new physical home of each page
    kc1, vc1 = np.empty_like(kc0), np.empty_like(vc0)
    kc1[perm], vc1[perm] = kc0, vc0
    bt1 = np.where(np.arange(bt0.shape[1])[None] < np.array(need)[:, None], perm[bt0], 0).astype(np.int32)
    a, _ = gpu_decode(gpu, q, kc0, vc0, bt0, lens, 0.088, BF16)
    b, _ = gpu_decode(gpu, q, kc1, vc1, bt1, lens, 0.088, BF16)
    assert np.array_equal(a, b)


def test_decode_full_size_properties_c2a(gpu):
    """BASELINE.json configs[1] at full size (B=256, h=32, h_k=8, d=128, seq=4096, page 16, bf16):
    too big for the oracle, so check properties: (1) V = per-(page,row) constant makes the
    output a convex combination -> bounded by V's range and equal across d; (2) a sequence whose
    keys are all identical gets the exact mean of its V rows; (3) a sampled subset of sequences
    matches the C oracle."""
    rng = np.random.default_rng(12)
    B, h, hk, d, S, page = 256, 32, 8, 128, 4096, 16
    nb = B * (S // page)
    bt = rng.permutation(nb).astype(np.int32).reshape(B, S // page)
    kc = rand_half(rng, (nb, page, hk, d), BF16)
    vrow = rand_half(rng, (nb, page, hk, 1), BF16)
    vc = np.broadcast_to(vrow, (nb, page, hk, d)).copy()
    kc[bt[0]] = kc[bt[0][0], 0, 0, 0]                          # sequence 0: all keys identical
    q = rand_half(rng, (B, 1, h, d), BF16)
    lens = np.full(B, S, np.int32)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    o = to_f32(out, BF16)
    assert np.isfinite(o).all() and np.isfinite(lse).all()
    assert np.ptp(o, axis=-1).max() == 0, "V constant along d -> output constant along d"
    v = to_f32(vrow, BF16)
    assert o.max() <= v.max() + 1e-6 and o.min() >= v.min() - 1e-6
    mean_v = to_f32(vc, BF16)[bt[0]].reshape(S, hk, d).mean(0)               # [hk, d]
    got0 = o[0, 0].reshape(hk, h // hk, d)
    assert np.abs(got0 - mean_v[:, None]).max() < 2e-3
    pick = [1, 100, 255]
    ref = c_attention(q[pick], kc, vc, b=len(pick), h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=S, scale=d ** -0.5,
                      is_bf16=1, q_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                      v_strides=(page * hk * d, hk * d, d), o_shape=(len(pick), 1, h, d), o_strides=(h * d, h * d, d),
                      cu_k=lens[pick], k_cumulative=False, block_table=bt[pick], page=page)
    assert_close(out[pick], ref, BF16, atol=1e-3, what="C2a sampled sequences vs C oracle (f32)")


def test_decode_seqused_k_and_cumulative_lengths(gpu):
    """block_info.h:16-23: seqused_k overrides the length; cumulative cu_seqlens_k addresses a varlen K/V."""
    rng = np.random.default_rng(55)
    B, h, hk, d = 4, 8, 2, 128
    lens = np.array([30, 64, 1, 100], np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    k, v = rand_half(rng, (int(cu[-1]), hk, d), BF16), rand_half(rng, (int(cu[-1]), hk, d), BF16)
    q = rand_half(rng, (B, 1, h, d), BF16)
    used = np.array([10, 64, 1, 77], np.int32)
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v))
    do = gpu.DeviceBuffer(q.nbytes)
    dcu, dused = gpu.DeviceBuffer.from_numpy(cu), gpu.DeviceBuffer.from_numpy(used)
    for su, want_lens in ((None, lens), (dused, used)):
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=100, softmax_scale=0.088, is_bf16=1,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(0, hk * d, d),
                    v_strides=(0, hk * d, d), cu_seqlens_k=dcu, is_seqlens_k_cumulative=True, seqused_k=su,
                    unpadded_lse=False)
        gpu.synchronize()
        out = do.numpy(np.uint16, q.shape)
        for b in range(B):
            kb, vb = k[cu[b]: cu[b] + want_lens[b]][None], v[cu[b]: cu[b] + want_lens[b]][None]
            ref = A.flash_attn_kv_cache(q[b:b + 1], kb, vb, 0.088, BF16)
            assert_close(out[b:b + 1], ref, BF16, atol=ATOL_VS_F32[BF16], what=f"cumulative lens seq {b}")


@pytest.mark.parametrize("dtype", [F16])
def test_decode_f16_long_sequences(gpu, dtype):
    rng = np.random.default_rng(66)
    lens = np.array([4096, 2500, 4095], np.int32)
    h, hk, d, page = 32, 8, 128, 16
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (3, 1, h, d), dtype)
    out, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    ref = c_attention(q, kc, vc, b=3, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, scale=d ** -0.5,
                      is_bf16=0, q_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                      v_strides=(page * hk * d, hk * d, d), o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens,
                      k_cumulative=False, block_table=bt, page=page)
    assert_close(out, ref, dtype, atol=1e-3, what="f16 long decode vs C oracle")


# ---- groups of more than 4 q heads at d = 128: q.K^T on the matrix cores (paged_decode_mqk_kernel) ----

@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("h,hk,page", [(64, 8, 16), (8, 1, 16), (5, 1, 16), (12, 2, 32), (14, 2, 16), (16, 1, 64)])
def test_decode_large_groups_matrix_core_scores(gpu, dtype, h, hk, page):
    """Llama-70B head geometry (64 q / 8 kv heads) and groups of 5, 6, 7, 8 and 16 q heads per kv head (padded
    columns of the score tile; 16 = two passes of 8), ragged lengths around every tile boundary, empty sequence."""
    d = 128
    rng = np.random.default_rng(h * 13 + hk + page)
    lens = np.array([0, 1, 3, 4, 5, 15, 16, 17, 47, 48, 49, 64, 130, 333], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 3
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (len(lens), 1, h, d), dtype)
    scale = np.float32(d ** -0.5)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype)
    for mode in ("f32", "kernel"):
        ref = A.flash_attn_kv_cache(q, kc, vc, scale, dtype, bt, lens, mode=mode)
        for i, L in enumerate(lens):
            assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"decode g={h // hk} L={L} vs {mode} oracle")
    assert not out[0].any() and np.isposinf(lse[0]).all() and np.isfinite(lse[1:]).all()


def test_decode_large_groups_both_kernels_agree(gpu):
    """The matrix-core and the dot2 kernel compute the same thing in different orders: outputs agree to the P-rounding
    bound, LSE to f32 rounding; also with ALiBi, a contiguous cache, and the split-KV path (one long sequence)."""
    rng = np.random.default_rng(77)
    h, hk, d, page = 16, 2, 128, 16
    slopes = (2.0 ** -np.linspace(0.5, 8, h)).astype(np.float32)
    cases = []
    lens = np.array([70, 33, 500, 16], np.int32)
    kc, vc, bt = make_paged_cache(rng, 45, page, hk, d, BF16, lens)
    cases.append(("paged", dict(kc=kc, vc=vc, bt=bt, lens=lens, alibi=None)))
    cases.append(("paged+alibi", dict(kc=kc, vc=vc, bt=bt, lens=lens, alibi=slopes)))
    kcc, vcc = rand_half(rng, (4, 100, hk, d), BF16), rand_half(rng, (4, 100, hk, d), BF16)
    cases.append(("contiguous", dict(kc=kcc, vc=vcc, bt=None, lens=np.array([100, 1, 50, 99], np.int32), alibi=None)))
    lens1 = np.array([4096], np.int32)
    kc1, vc1, bt1 = make_paged_cache(rng, 260, page, hk, d, BF16, lens1)
    cases.append(("split-KV", dict(kc=kc1, vc=vc1, bt=bt1, lens=lens1, alibi=None)))
    for name, c in cases:
        q = rand_half(rng, (len(c["lens"]), 1, h, d), BF16)
        res = {}
        for mqk in (1, 0):
            assert gpu.lib.atoma_set_option(b"decode_mqk", mqk) == 0
            res[mqk] = gpu_decode(gpu, q, c["kc"], c["vc"], c["bt"], c["lens"], 0.088, BF16, alibi=c["alibi"])
        gpu.lib.atoma_set_option(b"decode_mqk", 29)
        assert_close(res[1][0], res[0][0], BF16, atol=ATOL_VS_F32[BF16], what=f"{name}: matrix-core vs dot2 kernel")
        assert np.allclose(res[1][1], res[0][1], rtol=1e-5, atol=1e-5), name
        ref = A.flash_attn_kv_cache(q, c["kc"], c["vc"], 0.088, BF16, c["bt"], c["lens"], causal=True, alibi_slopes=c["alibi"])
        assert_close(res[1][0], ref, BF16, atol=ATOL_VS_F32[BF16], what=f"{name}: matrix-core kernel vs oracle")


def test_decode_full_size_70b_shape_properties(gpu):
    """Llama-70B decode shape at full size (B=256, 64 q / 8 kv heads, S=4096): size-independent properties --
    V constant along d => output constant along d, equal to that constant where V is constant over tokens; block-table
    permutation invariance, bit for bit."""
    rng = np.random.default_rng(70)
    B, S, h, hk, d, page = 256, 4096, 64, 8, 128, 16
    pps = S // page
    nb = B * pps
    kc = rand_half(rng, (nb, page, hk, d), BF16)
    vrow = rand_half(rng, (nb, page, hk, 1), BF16)
    vc = np.ascontiguousarray(np.broadcast_to(vrow, (nb, page, hk, d)))
    bt = rng.permutation(nb).astype(np.int32).reshape(B, pps)
    lens = np.full(B, S, np.int32)
    q = rand_half(rng, (B, 1, h, d), BF16)
    out, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, BF16)
    assert (out == out[..., :1]).all(), "V constant along d must give outputs constant along d"
    o32, v32 = to_f32(out, BF16), to_f32(vrow, BF16)
    assert o32.min() >= v32.min() - 1e-2 and o32.max() <= v32.max() + 1e-2
    # the same pages listed in another order for every sequence with K and V moved along: identical bits
    perm = rng.permutation(nb)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(nb)
    out2, _ = gpu_decode(gpu, q, kc[perm], vc[perm], inv[bt].astype(np.int32), lens, d ** -0.5, BF16)
    assert np.array_equal(out, out2)


class _options:
    """atoma_set_option for the duration of a test (defaults restored afterwards)."""
    DEFAULTS = {"decode_mqk": 29, "decode_min_tiles": 8, "decode_stream": 1, "decode_head_major": 1, "decode_pair64": 1, "decode_pair": 0}

    def __init__(self, gpu, **kw):
        self.gpu, self.kw = gpu, kw

    def __enter__(self):
        for k, v in self.kw.items():
            assert self.gpu.lib.atoma_set_option(k.encode(), v) == 0

    def __exit__(self, *a):
        for k in self.kw:
            self.gpu.lib.atoma_set_option(k.encode(), self.DEFAULTS[k])


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("d,h,hk,variant", [(128, 32, 8, "dot2"), (128, 8, 8, "dot2"), (128, 16, 2, "dot2"), (64, 32, 8, "dot2"),
                                            (128, 32, 8, "mqk"), (128, 16, 2, "mqk"), (128, 6, 2, "mqk"),
                                            # head_dim 64 as pairs of kv heads on the matrix-core kernel (round 4): groups of 4, 2 and 1 q heads
                                            (64, 32, 8, "mqk"), (64, 8, 4, "mqk"), (64, 4, 4, "mqk")])
def test_decode_matches_own_schedule_tightly(gpu, dtype, d, h, hk, variant):
    """1e-3 + 1 ulp at EVERY row length (no few-keys allowance) against the oracle evaluated under this kernel's own
    online-softmax schedule (oracle/attn_oracle.py attend_decode_online; tests/test_oracle_schedules.py shows on the CPU
    why another schedule cannot be compared that tightly).  KV splitting and the balanced mode are switched off so that one
    wavefront owns a whole sequence -- their merges are fp32 LSE arithmetic and are covered by the other tests."""
    rng = np.random.default_rng(d + h + hk + (7 if variant == "mqk" else 0))
    lens = np.array([0, 1, 2, 3, 5, 15, 16, 17, 31, 33, 64, 65, 127, 200, 333, 700], np.int32)
    nb = int(sum((L + 15) // 16 for L in lens)) + 3
    kc, vc, bt = make_paged_cache(rng, nb, 16, hk, d, dtype, lens)
    q = rand_half(rng, (len(lens), 1, h, d), dtype)
    scale = np.float32(d ** -0.5)
    with _options(gpu, decode_mqk=7 if variant == "mqk" else 0, decode_min_tiles=1 << 20, decode_stream=0, decode_pair64=1 if variant == "mqk" else 0):
        out, _ = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype)
        if d == 64:
            assert ("kv-head pairs" in gpu.lib.atoma_last_decode_kernel().decode()) == (variant == "mqk")
    ref = A.flash_attn_kv_cache_online(q, kc, vc, scale, dtype, bt, lens, variant)
    assert_close(out, ref, dtype, atol=1e-3, what=f"decode vs own-schedule oracle ({variant}, d={d})")
    # and it is far tighter than that almost everywhere: at most a handful of outputs differ at all (a p on a rounding
    # boundary where v_exp_f32 and numpy's exp2 differ in the last f32 bit)
    assert (out != ref).mean() < 0.01


@pytest.mark.parametrize("dtype,nan", [(BF16, 0x7FC0), (F16, 0x7E00)])
@pytest.mark.parametrize("d,h,hk,page", [(128, 32, 8, 16), (128, 8, 8, 16), (128, 64, 8, 32), (64, 32, 8, 16), (128, 8, 1, 64)])
def test_decode_never_written_slots_do_not_reach_the_output(gpu, dtype, nan, d, h, hk, page):
    """rows >= L of the last page get p = 0 but still enter P.V (matrix-core and dot2 kernels alike): NaN patterns in
    those never-written slots -- and in pages no sequence owns -- must leave every output bit where it was."""
    rng = np.random.default_rng(d + h + hk + page)
    lens = np.array([1, 2, 15, 17, 31, 33, 100, 333, 1000, 2049], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 5
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (len(lens), 1, h, d), dtype)
    clean, _ = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    kp, vp = poison_unwritten_slots(kc, vc, bt, lens, nan)
    got, lse = gpu_decode(gpu, q, kp, vp, bt, lens, d ** -0.5, dtype)
    assert np.isfinite(to_f32(got, dtype)).all() and np.isfinite(lse).all()
    assert np.array_equal(got, clean)


@pytest.mark.parametrize("seed", range(16))
def test_decode_random_shapes_through_the_default_dispatch(gpu, seed):
    """Whatever the dispatcher picks by default (dot2 / matrix-core kernel, kv-head pairs at head_dim 64, the balanced line with its
    in-launch merge, split-KV with the combine kernel or the workgroup merge) for a random batch -- random head counts and group sizes,
    both head sizes, pages of 16 / 32 / 64 tokens, batch 1..320, lengths from empty to a few thousand with a few long stragglers,
    a shuffled block table -- the answer is the C oracle's, twice in a row to the bit."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([64, 128]))
    hk = int(rng.choice([1, 2, 3, 4, 8]))
    g = int(rng.choice([1, 2, 3, 4, 6, 8]))
    h = hk * g
    page = int(rng.choice([16, 32, 64]))
    dtype = BF16 if seed % 3 else F16
    B = int(rng.choice([1, 3, 17, 64, 130, 257, 320]))
    top = int(rng.choice([40, 300, 1500]))
    top = max(20, min(top, (48 << 20) // (B * hk * d)))                                # (keeps the caches of the largest batches at ~100 MB)
    lens = rng.integers(0, top + 1, B).astype(np.int32)
    lens[rng.integers(0, B, max(1, B // 50))] = rng.integers(top, 4 * top + 50)       # stragglers
    if seed % 4 == 0:
        lens[:] = int(lens.max())                                                       # a uniform batch now and then
    nb = int(sum((int(x) + page - 1) // page for x in lens)) + 2
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
    q = rand_half(rng, (B, 1, h, d), dtype)
    ref = _oracle_decode(q, kc, vc, bt, lens, dtype)
    out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    name = gpu.lib.atoma_last_decode_kernel().decode()
    again, lse2 = gpu_decode(gpu, q, kc, vc, bt, lens, d ** -0.5, dtype)
    assert np.array_equal(out, again) and np.array_equal(lse, lse2), name
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, L), what=f"seed {seed} {name}: B={B} h={h}/{hk} d={d} page={page} seq {i} (L={L})")
    empty = lens == 0
    assert not out[empty].any() and np.isposinf(lse[empty]).all() and np.isfinite(lse[~empty]).all(), name


@pytest.mark.parametrize("d", [8, 32, 96, 160, 192, 224, 256])
@pytest.mark.parametrize("dtype", [BF16, F16])
def test_decode_other_head_sizes_streaming_kernel(gpu, d, dtype):
    """Decode for the head sizes the reference instantiates besides 64 / 128 (csrc/build.rs:7-74) -- `attn_decode_anyd_kernel`: one wavefront per
    (sequence, kv head, up to 4 q heads) streaming K / V once, instead of the row-per-lane coverage kernel.  Ragged lengths incl. 0, 1 and a
    non-multiple of 16, groups of 1 / 3 / 5 q heads (one and two chunks), paged and contiguous caches, ALiBi, the LSE output; against the
    oracle, and against the coverage kernel on the same call (atoma_set_option generic_decode_stream switches back to it)."""
    rng = np.random.default_rng(1000 + d + dtype)
    for h, hk, page in ((4, 4, 16), (6, 2, 16), (5, 1, 16), (4, 2, 32), (4, 4, 48)):      # groups of 1 / 3 / 5 / 2 q heads; pages of 16, 32 and 48 tokens
        lens = np.array([0, 1, 15, 16, 17, 100, 333, 77], np.int32)
        nb = int(sum((x + page - 1) // page for x in lens)) + 3
        kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
        q = rand_half(rng, (len(lens), 1, h, d), dtype)
        scale = np.float32(d ** -0.5)
        for alibi in (None, (0.5 ** np.arange(1, h + 1)).astype(np.float32)):
            ref = A.flash_attn_kv_cache(q, kc, vc, scale, dtype, bt, lens, alibi_slopes=alibi)
            out, lse = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype, alibi=alibi)
            for i, L in enumerate(lens):
                assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, int(L)), what=f"d={d} h={h}/{hk} alibi={alibi is not None} seq {i} (L={L})")
            assert not out[0].any() and np.isinf(lse[0]).all()
            for waves in (1, 4, 8):      # 1 / 4 / 8 wavefronts (the default here: 2) share a unit's tiles and merge in LDS (what small batches take)
                assert gpu.lib.atoma_set_option(b"generic_decode_waves", waves) == 0
                try:
                    out_w, lse_w = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype, alibi=alibi)
                finally:
                    gpu.lib.atoma_set_option(b"generic_decode_waves", 0)
                for i, L in enumerate(lens):
                    assert_close(out_w[i], ref[i], dtype, atol=attn_atol(dtype, int(L)), what=f"d={d} h={h}/{hk} {waves} wavefronts per unit, seq {i} (L={L})")
                assert not out_w[0].any() and np.isinf(lse_w[0]).all()
                assert np.allclose(lse_w[lens > 0], lse[lens > 0], rtol=1e-4, atol=1e-4)
            for other in (1, 0):      # the first streaming kernel; the row-per-lane coverage kernel -- on the same call
                assert gpu.lib.atoma_set_option(b"generic_decode_stream", other) == 0
                try:
                    old, lse_old = gpu_decode(gpu, q, kc, vc, bt, lens, scale, dtype, alibi=alibi)
                finally:
                    gpu.lib.atoma_set_option(b"generic_decode_stream", 2)
                for i, L in enumerate(lens):
                    assert_close(out[i], old[i], dtype, atol=attn_atol(dtype, int(L)), what=f"d={d} vs kernel {other}, seq {i}")
                live = lens > 0
                assert np.allclose(lse[live], lse_old[live], rtol=1e-4, atol=1e-4)
        # contiguous cache [B, S, hk, d] with per-sequence lengths
        S = 64
        kd, vd = rand_half(rng, (4, S, hk, d), dtype), rand_half(rng, (4, S, hk, d), dtype)
        ld = np.array([64, 1, 33, 50], np.int32)
        qd = rand_half(rng, (4, 1, h, d), dtype)
        out, _ = gpu_decode(gpu, qd, kd, vd, None, ld, scale, dtype)
        ref = A.flash_attn_kv_cache(qd, kd, vd, scale, dtype, None, ld)
        for i, L in enumerate(ld):
            assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, int(L)), what=f"d={d} contiguous seq {i}")

