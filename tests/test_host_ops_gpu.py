"""The host mirror of the reference's operator layer, end to end on the GPU: the reference's
models/src/flash_attention.rs tests (test_forward, test_forward_with_varlen) and the csrc wrapper
paths, checked against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import attn_oracle as A, cache_oracle as CO
from oracle.halfs import F16, BF16, to_f32
from util import rand_half, make_paged_cache, assert_close, ATOL_VS_F32

pytestmark = pytest.mark.gpu
REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_tables.npz"))


def round4(x):
    return np.round(x.astype(np.float32) * np.float32(1e4)) / np.float32(1e4)


def dev(gpu, a):
    return gpu.DeviceBuffer.from_numpy(a)


def test_forward_with_varlen_golden_g2(gpu):
    """models/src/flash_attention.rs:632-705: prefill of one 2-token sequence, 3 heads, d = 8,
    kv_cache [2,128,16,3,8], slot_mapping arange(2); expected = the causal golden table."""
    ah = gpu
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2)) for n in "qkv")  # [2, 3, 8]
    dq, dk, dv = dev(ah, q), dev(ah, k), dev(ah, v)
    dkv = ah.DeviceBuffer.zeros((2, 128, 16, 3, 8), np.uint16)
    dslots = dev(ah, np.arange(2, dtype=np.int64))
    dcu = dev(ah, np.array([0, 2], np.uint32))
    dout = ah.DeviceBuffer(q.nbytes)
    fa = ah.FlashAttention()
    # head_dim 8 is below supported_head_sizes(); the reference test builds the struct literally
    fa.num_heads, fa.num_kv_heads, fa.head_dim, fa.softmax_scale, fa.sliding_window = 3, 3, 8, 0.5, -1
    fa.kv_cache_dtype, fa.device = F16, 0
    tq, tk, tv = (ah.tensor(b, (2, 3, 8), F16) for b in (dq, dk, dv))
    tkv = ah.tensor(dkv, (2, 128, 16, 3, 8), F16)
    tslots, tcu = ah.tensor(dslots, (2,), ah.I64), ah.tensor(dcu, (2,), ah.U32)
    tout = ah.tensor(dout, (2, 24), F16)
    meta = ah.AttnMetadata()
    meta.slot_mapping = C.pointer(tslots)
    meta.num_prefill_tokens, meta.num_decoding_tokens = 2, 0
    meta.has_prefill, meta.max_prefill_sequence_length = 1, 32
    meta.sequence_start_locations = C.pointer(tcu)
    rc = ah.lib.atoma_flash_attention_forward(C.byref(fa), ah.ref(tq), ah.ref(tk), ah.ref(tv), ah.ref(tkv), C.byref(meta), ah.ref(tout))
    assert rc == 0, ah.last_error()
    ah.synchronize()
    got = to_f32(dout.numpy(np.uint16, (2, 3, 8)), F16).transpose(1, 0, 2)
    assert np.array_equal(round4(got), REF["G2"])
    kv = dkv.numpy(np.uint16, (2, 128, 16, 3, 8))
    assert np.array_equal(kv[0, 0, :2], k) and np.array_equal(kv[1, 0, :2], v) and not kv[:, 1:].any()


@pytest.mark.parametrize("prefix", [False, True])
def test_forward_mixed_prefill_and_decode(gpu, prefix):
    """models/src/flash_attention.rs:548-630 (test_forward), with values checked: 2 prefill sequences of
    5 tokens + 5 decode tokens, 16 q / 8 kv heads.  `prefix=True` routes prefill through the paged cache
    (block tables present), which the reference launches NON-causal (SURVEY B/Q3) -- mirrored as is."""
    ah = gpu
    rng = np.random.default_rng(17 + prefix)
    h, hk, d, page, nb = 16, 8, 64, 32, 10
    T, n_pre, n_dec = 15, 10, 5
    q, k, v = rand_half(rng, (T, h, d), BF16), rand_half(rng, (T, hk, d), BF16), rand_half(rng, (T, hk, d), BF16)
    kv = rand_half(rng, (2, nb, page, hk, d), BF16)
    pre_bt = np.array([[0], [1]], np.uint32)
    dec_bt = np.arange(2, 7, dtype=np.uint32).reshape(5, 1)
    dec_lens = np.full(5, 3, np.uint32)
    # prefill tokens land in pages 0,1 (slots 0..4 / 32..36); decode tokens are the 3rd token of pages 2..6
    slots = np.concatenate([np.arange(5), page + np.arange(5), dec_bt[:, 0] * page + 2]).astype(np.int64)
    cu = np.array([0, 5, 10], np.uint32)
    bufs = {n: dev(ah, a) for n, a in dict(q=q, k=k, v=v, kv=kv, slots=slots, cu=cu, pre_bt=pre_bt, dec_bt=dec_bt, dec_lens=dec_lens).items()}
    dout = ah.DeviceBuffer(q.nbytes)
    dout.fill_bytes(0xFF)
    fa = ah.FlashAttention()
    assert ah.lib.atoma_flash_attention_new(C.byref(fa), h, hk, d, float(d ** -0.5), None, -1, BF16, 0) == 0
    t = dict(q=ah.tensor(bufs["q"], (T, h, d), BF16), k=ah.tensor(bufs["k"], (T, hk, d), BF16),
             v=ah.tensor(bufs["v"], (T, hk, d), BF16), kv=ah.tensor(bufs["kv"], (2, nb, page, hk, d), BF16),
             slots=ah.tensor(bufs["slots"], (T,), ah.I64), cu=ah.tensor(bufs["cu"], (3,), ah.U32),
             pre_bt=ah.tensor(bufs["pre_bt"], (2, 1), ah.U32), dec_bt=ah.tensor(bufs["dec_bt"], (5, 1), ah.U32),
             dec_lens=ah.tensor(bufs["dec_lens"], (5,), ah.U32), out=ah.tensor(dout, (T, h * d), BF16))
    meta = ah.AttnMetadata()
    meta.slot_mapping = C.pointer(t["slots"])
    meta.num_prefill_tokens, meta.num_decoding_tokens = n_pre, n_dec
    meta.has_prefill, meta.max_prefill_sequence_length, meta.max_sequence_length_k = 1, 5, 5
    meta.sequence_start_locations = C.pointer(t["cu"])
    meta.query_start_locations = C.pointer(t["cu"])
    if prefix:
        meta.prefill_block_tables = C.pointer(t["pre_bt"])
    meta.has_decoding = 1
    meta.decoding_block_tables = C.pointer(t["dec_bt"])
    meta.decoding_sequence_lengths = C.pointer(t["dec_lens"])
    rc = ah.lib.atoma_flash_attention_forward(C.byref(fa), ah.ref(t["q"]), ah.ref(t["k"]), ah.ref(t["v"]), ah.ref(t["kv"]),
                                              C.byref(meta), ah.ref(t["out"]))
    assert rc == 0, ah.last_error()
    ah.synchronize()
    out = dout.numpy(np.uint16, (T, h, d))
    # oracle: cache write first (flash_attention.rs:360-361), then the two attention calls
    kc, vc = kv[0].copy(), kv[1].copy()
    CO.reshape_and_cache_flash(k, v, kc, vc, slots)
    cu_i = cu.astype(np.int32)
    if prefix:
        ref_pre = A.flash_attn_varlen(q[:n_pre], kc, vc, cu_i, cu_i, d ** -0.5, False, BF16, block_table=pre_bt.astype(np.int32))
    else:
        ref_pre = A.flash_attn_varlen(q[:n_pre], k[:n_pre], v[:n_pre], cu_i, cu_i, d ** -0.5, True, BF16)
    ref_dec = A.flash_attn_kv_cache(q[n_pre:, None], kc, vc, d ** -0.5, BF16, dec_bt.astype(np.int32), dec_lens.astype(np.int32))
    assert_close(out[:n_pre], ref_pre, BF16, atol=ATOL_VS_F32[BF16], what="forward: prefill rows")
    assert_close(out[n_pre:], ref_dec[:, 0], BF16, atol=ATOL_VS_F32[BF16], what="forward: decode rows")
    assert np.array_equal(bufs["kv"].numpy(np.uint16, kv.shape)[0], kc), "cache write is bit-exact"
    assert (to_f32(out, BF16) != 0).all()                     # the reference's own assertion ("no zeros")


def test_wrappers_match_oracle(gpu):
    """csrc::flash_attn / flash_attn_varlen / flash_attn_kv_cache_full through tensor descriptors,
    including a transposed (strided) q like the reference's tests build (flash_attn_tests.rs:44-49)."""
    ah = gpu
    rng = np.random.default_rng(23)
    b, s, h, hk, d = 2, 40, 4, 2, 64
    q_hsd = rand_half(rng, (b, h, s, d), F16)                   # stored [b, h, s, d]; viewed as [b, s, h, d]
    k, v = rand_half(rng, (b, s, hk, d), F16), rand_half(rng, (b, s, hk, d), F16)
    dq, dk, dv = dev(ah, q_hsd), dev(ah, k), dev(ah, v)
    dout = ah.DeviceBuffer(q_hsd.nbytes)
    tq = ah.tensor(dq, (b, s, h, d), F16, strides=(h * s * d, d, s * d, 1))          # .transpose(1, 2)
    tk, tv = ah.tensor(dk, (b, s, hk, d), F16), ah.tensor(dv, (b, s, hk, d), F16)
    to = ah.tensor(dout, (b, s, h, d), F16)
    for causal in (0, 1):
        assert ah.lib.atoma_flash_attn(ah.ref(tq), ah.ref(tk), ah.ref(tv), 0.125, causal, ah.ref(to)) == 0, ah.last_error()
        ah.synchronize()
        ref = A.flash_attn(np.ascontiguousarray(q_hsd.transpose(0, 2, 1, 3)), k, v, 0.125, bool(causal), F16)
        assert_close(dout.numpy(np.uint16, (b, s, h, d)), ref, F16, atol=ATOL_VS_F32[F16], what=f"flash_attn causal={causal}")
    # kv-cache entry point without block table, seqlens_k given
    lens = np.array([17, 40], np.uint32)
    q1 = rand_half(rng, (b, 1, h, d), F16)
    dq1, dl = dev(ah, q1), dev(ah, lens)
    do1 = ah.DeviceBuffer(q1.nbytes)
    rc = ah.lib.atoma_flash_attn_kv_cache_full(ah.ref(ah.tensor(dq1, (b, 1, h, d), F16)), ah.ref(tk), ah.ref(tv), None, 0.125,
                                               None, ah.ref(ah.tensor(dl, (b,), ah.U32)), 1, ah.ref(ah.tensor(do1, (b, 1, h, d), F16)))
    assert rc == 0, ah.last_error()
    ah.synchronize()
    ref = A.flash_attn_kv_cache(q1, k, v, 0.125, F16, None, lens.astype(np.int32))
    assert_close(do1.numpy(np.uint16, q1.shape), ref, F16, atol=ATOL_VS_F32[F16], what="kv_cache_full")


def test_the_other_forms_of_the_three_ops(gpu):
    """The twelve convenience forms of csrc/src/lib.rs (:432-572, :1218-1495, :1907-2053) through tensor descriptors: windows as
    Option (negative = None; (None, Some(0)) is the causal mask -- sliding windows are compiled out of the reference's kernels, so any
    other window is a no-op there and here), ALiBi slopes against the oracle, seqused_k of the varlen form, and every form bit for bit
    the form it abbreviates."""
    ah = gpu
    rng = np.random.default_rng(31)
    b, s, h, hk, d = 2, 48, 4, 2, 64
    q, k, v = rand_half(rng, (b, s, h, d), F16), rand_half(rng, (b, s, hk, d), F16), rand_half(rng, (b, s, hk, d), F16)
    slopes = (2.0 ** -np.arange(1, h + 1)).astype(np.float32)
    dq, dk, dv, ds = dev(ah, q), dev(ah, k), dev(ah, v), dev(ah, slopes)
    tq, tk, tv = ah.tensor(dq, q.shape, F16), ah.tensor(dk, k.shape, F16), ah.tensor(dv, v.shape, F16)
    ts = ah.tensor(ds, (h,), ah.F32)
    L = ah.lib

    def run(fn, *args):
        out = ah.DeviceBuffer(q.nbytes)
        out.fill_bytes(0xEE)
        assert fn(*args, ah.ref(ah.tensor(out, q.shape, F16))) == 0, ah.last_error()
        ah.synchronize()
        return out.numpy(np.uint16, q.shape)
    R = ah.ref
    plain = {c: run(L.atoma_flash_attn, R(tq), R(tk), R(tv), 0.125, c) for c in (0, 1)}
    assert np.array_equal(run(L.atoma_flash_attn_windowed, R(tq), R(tk), R(tv), 0.125, -1, 0), plain[1])        # (None, Some(0)) = causal
    assert np.array_equal(run(L.atoma_flash_attn_windowed, R(tq), R(tk), R(tv), 0.125, -1, -1), plain[0])
    assert np.array_equal(run(L.atoma_flash_attn_windowed, R(tq), R(tk), R(tv), 0.125, 7, 3), plain[0])         # a real window: compiled out
    assert np.array_equal(run(L.atoma_flash_attn_windowed, R(tq), R(tk), R(tv), 0.125, -1, s + 5), plain[0])    # beyond seqlen_k -> None
    for causal in (0, 1):
        got = run(L.atoma_flash_attn_alibi, R(tq), R(tk), R(tv), R(ts), 0.125, causal)
        ref = A.flash_attn(q, k, v, 0.125, bool(causal), F16, alibi_slopes=slopes)
        assert_close(got, ref, F16, atol=ATOL_VS_F32[F16], what=f"flash_attn_alibi causal={causal}")
        assert np.array_equal(run(L.atoma_flash_attn_alibi_windowed, R(tq), R(tk), R(tv), R(ts), 0.125, -1, 0 if causal else -1), got)
        assert np.array_equal(run(L.atoma_flash_attn_alibi_windowed_with_softcap, R(tq), R(tk), R(tv), R(ts), 0.125, -1, 0 if causal else -1, 30.0), got)
    assert L.atoma_flash_attn_alibi(R(tq), R(tk), R(tv), None, 0.125, 0, R(tq)) == -1 and "alibi_slopes" in ah.last_error()
    # varlen forms on the same data packed as two sequences
    T = b * s
    cu = np.array([0, s, 2 * s], np.uint32)
    dcu = dev(ah, cu)
    tcu = ah.tensor(dcu, (3,), ah.U32)
    q3, k3, v3 = (ah.tensor(x, shp, F16) for x, shp in ((dq, (T, h, d)), (dk, (T, hk, d)), (dv, (T, hk, d))))

    def run3(fn, *args):
        out = ah.DeviceBuffer(q.nbytes)
        out.fill_bytes(0xEE)
        assert fn(*args, ah.ref(ah.tensor(out, (T, h, d), F16))) == 0, ah.last_error()
        ah.synchronize()
        return out.numpy(np.uint16, q.shape)
    for causal in (0, 1):
        base = run3(L.atoma_flash_attn_varlen, R(q3), R(k3), R(v3), R(tcu), R(tcu), s, s, 0.125, causal)
        assert np.array_equal(base, plain[causal])                                   # the padded and the packed op agree to the bit here
        assert np.array_equal(run3(L.atoma_flash_attn_varlen_windowed, R(q3), R(k3), R(v3), R(tcu), R(tcu), s, s, 0.125, -1, 0 if causal else -1), base)
        al = run3(L.atoma_flash_attn_varlen_alibi, R(q3), R(k3), R(v3), R(ts), R(tcu), R(tcu), s, s, 0.125, causal)
        ref = A.flash_attn_varlen(q.reshape(T, h, d), k.reshape(T, hk, d), v.reshape(T, hk, d), cu.astype(np.int32), cu.astype(np.int32), 0.125, bool(causal), F16,
                                  alibi_slopes=slopes)
        assert_close(al.reshape(T, h, d), ref, F16, atol=ATOL_VS_F32[F16], what=f"flash_attn_varlen_alibi causal={causal}")
        assert np.array_equal(run3(L.atoma_flash_attn_varlen_alibi_windowed, R(q3), R(k3), R(v3), R(ts), R(tcu), R(tcu), s, s, 0.125, -1, 0 if causal else -1), al)
        assert np.array_equal(run3(L.atoma_flash_attn_varlen_full, R(q3), R(k3), R(v3), R(ts), R(tcu), R(tcu), s, s, 0.125, -1, 0 if causal else -1, None, None, 0.0), al)
    # seqused_k: only the first `used` keys of every sequence take part (block_info.h:16-23)
    used = np.array([20, 33], np.uint32)
    du = dev(ah, used)
    got = run3(L.atoma_flash_attn_varlen_full, R(q3), R(k3), R(v3), None, R(tcu), R(tcu), s, s, 0.125, -1, -1, None, R(ah.tensor(du, (b,), ah.U32)), 0.0)
    for i in range(b):
        ref = A.flash_attn(q[i:i + 1], k[i:i + 1, :used[i]], v[i:i + 1, :used[i]], 0.125, False, F16)
        assert_close(got[i:i + 1], ref, F16, atol=ATOL_VS_F32[F16], what=f"varlen_full seqused_k seq {i}")
    # kv-cache forms (one query row per sequence, contiguous caches)
    q1 = rand_half(rng, (b, 1, h, d), F16)
    dq1 = dev(ah, q1)
    tq1 = ah.tensor(dq1, q1.shape, F16)
    lens = np.array([17, 48], np.uint32)
    dl = dev(ah, lens)
    tl = ah.tensor(dl, (b,), ah.U32)

    def run1(fn, *args):
        out = ah.DeviceBuffer(q1.nbytes)
        assert fn(*args, ah.ref(ah.tensor(out, q1.shape, F16))) == 0, ah.last_error()
        ah.synchronize()
        return out.numpy(np.uint16, q1.shape)
    full = run1(L.atoma_flash_attn_kv_cache_full, R(tq1), R(tk), R(tv), None, 0.125, None, None, 1)
    assert np.array_equal(run1(L.atoma_flash_attn_kv_cache, R(tq1), R(tk), R(tv), 0.125, 1), full)
    with_lens = run1(L.atoma_flash_attn_kv_cache_full, R(tq1), R(tk), R(tv), None, 0.125, None, R(tl), 0)
    assert np.array_equal(run1(L.atoma_flash_attn_kv_cache_windowed, R(tq1), R(tk), R(tv), R(tl), 0.125, -1, -1), with_lens)
    al = run1(L.atoma_flash_attn_kv_cache_alibi, R(tq1), R(tk), R(tv), R(ts), R(tl), 0.125, 1)
    ref = A.flash_attn_kv_cache(q1, k, v, 0.125, F16, None, lens.astype(np.int32), causal=True, alibi_slopes=slopes)
    assert_close(al, ref, F16, atol=ATOL_VS_F32[F16], what="flash_attn_kv_cache_alibi")
    al_all = run1(L.atoma_flash_attn_kv_cache_alibi_windowed, R(tq1), R(tk), R(tv), R(ts), 0.125, -1, 0)
    ref = A.flash_attn_kv_cache(q1, k, v, 0.125, F16, None, None, causal=True, alibi_slopes=slopes)
    assert_close(al_all, ref, F16, atol=ATOL_VS_F32[F16], what="flash_attn_kv_cache_alibi_windowed")


def test_copy_and_swap_through_tensor_api(gpu):
    ah = gpu
    rng = np.random.default_rng(29)
    shape = (6, 16, 2, 64)
    ks = [rand_half(rng, shape, BF16) for _ in range(3)]
    vs = [rand_half(rng, shape, BF16) for _ in range(3)]
    dks, dvs = [dev(ah, a) for a in ks], [dev(ah, a) for a in vs]
    tks, tvs = [ah.tensor(b, shape, BF16) for b in dks], [ah.tensor(b, shape, BF16) for b in dvs]
    mapping = np.array([[0, 3], [1, 5]], np.int64)
    dm = dev(ah, mapping)
    rc = ah.lib.atoma_copy_blocks(ah.tensor_array(tks), 3, ah.tensor_array(tvs), 3, ah.ref(ah.tensor(dm, (2, 2), ah.I64)))
    assert rc == 0, ah.last_error()
    ah.synchronize()
    CO.copy_blocks(ks, vs, mapping)
    for l in range(3):
        assert np.array_equal(dks[l].numpy(np.uint16, shape), ks[l]) and np.array_equal(dvs[l].numpy(np.uint16, shape), vs[l])
    # swap gpu -> cpu (pageable numpy memory) and back, HashMap<u32,u32> as flat pairs
    host = np.zeros(shape, np.uint16)
    pairs = (C.c_uint32 * 4)(0, 2, 3, 1)
    th = ah.tensor(host, shape, BF16)
    assert ah.lib.atoma_swap_blocks_tensor(ah.ref(tks[0]), ah.ref(th), pairs, 2) == 0, ah.last_error()
    ah.synchronize()
    assert np.array_equal(host[2], ks[0][0]) and np.array_equal(host[1], ks[0][3]) and not host[[0, 3, 4, 5]].any()


def test_allreduce_single_rank_communicator(gpu):
    """multi_gpu.rs:141-179 with a world of one (the reference's own TP test does the same,
    models/src/llama_nccl.rs:363-407): out-of-place sum == copy.  Multi-rank behaviour is RCCL's."""
    ah = gpu
    raw = (C.c_uint8 * 128)()
    assert ah.lib.atoma_comm_unique_id(raw) == 0, ah.last_error()
    comm = C.c_void_p()
    assert ah.lib.atoma_comm_init(C.byref(comm), 0, 1, raw, 0) == 0, ah.last_error()
    x = rand_half(np.random.default_rng(1), (256, 4096), BF16)
    dx, dy = dev(ah, x), ah.DeviceBuffer(x.nbytes)
    assert ah.lib.atoma_allreduce_sum(comm, dx.ptr, dy.ptr, x.size, BF16, None) == 0, ah.last_error()
    ah.synchronize()
    assert np.array_equal(dy.numpy(np.uint16, x.shape), x)
    assert ah.lib.atoma_comm_destroy(comm) == 0
