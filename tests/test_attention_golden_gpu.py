"""The reference's own attention tests, run against libatoma_hip through run_mha
(/root/reference/csrc/tests/flash_attn_tests.rs, models/src/flash_attention.rs:632-705)."""
import os

import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle.halfs import F16, BF16, to_f32
from util import rand_half, make_paged_cache, assert_close, ATOL_VS_F32

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = np.load(os.path.join(GOLD, "reference_tables.npz"))
CASES = np.load(os.path.join(GOLD, "oracle_cases.npz"))


def round4(x):
    return np.round(x.astype(np.float32) * np.float32(1e4)) / np.float32(1e4)


def gpu_varlen(gpu, q, k, v, cu_q, cu_k, scale, causal, dtype, bt=None, max_q=None, max_k=None, alibi=None):
    """q [Tq,h,d]; k,v [Tk,hk,d] or paged [nb,page,hk,d]."""
    Tq, h, d = q.shape
    hk = k.shape[-2]
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v))
    do = gpu.DeviceBuffer(q.nbytes)
    do.fill_bytes(0xFF)
    dcq = gpu.DeviceBuffer.from_numpy(np.asarray(cu_q, np.int32))
    dck = gpu.DeviceBuffer.from_numpy(np.asarray(cu_k, np.int32))
    dbt = gpu.DeviceBuffer.from_numpy(np.ascontiguousarray(bt, np.int32)) if bt is not None else None
    da = gpu.DeviceBuffer.from_numpy(np.asarray(alibi, np.float32)) if alibi is not None else None
    dlse = gpu.DeviceBuffer.zeros((h, Tq), np.float32)
    B = len(cu_q) - 1
    lq = np.diff(cu_q).max() if max_q is None else max_q
    lk = np.diff(cu_k).max() if max_k is None else max_k
    page = k.shape[1] if bt is not None else 0
    kstr = (page * hk * d, hk * d, d) if bt is not None else (0, hk * d, d)
    gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=int(lq), seqlen_k=int(lk),
                softmax_scale=float(scale), is_bf16=dtype, q_strides=(0, h * d, d), o_strides=(0, h * d, d),
                k_strides=kstr, v_strides=kstr, is_causal=int(causal), cu_seqlens_q=dcq, cu_seqlens_k=dck,
                block_table=dbt, block_table_batch_stride=0 if bt is None else bt.shape[1], page_block_size=page,
                alibi_slopes=da, softmax_lse=dlse, force_split_kernel=bt is not None)
    gpu.synchronize()
    return do.numpy(np.uint16, q.shape), dlse.numpy()


def test_g1_flash_attn_acausal(gpu):
    """flash_attn_tests.rs:31-93 through the dense entry (no cu_seqlens): q [1,2,3,8]."""
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2))[None] for n in "qkv")
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v))
    do = gpu.DeviceBuffer(q.nbytes)
    gpu.run_mha(dq, dk, dv, do, b=1, h=3, h_k=3, d=8, seqlen_q=2, seqlen_k=2, softmax_scale=0.5, is_bf16=0,
                q_strides=(48, 24, 8), k_strides=(48, 24, 8), v_strides=(48, 24, 8), o_strides=(48, 24, 8))
    gpu.synchronize()
    got = round4(to_f32(do.numpy(np.uint16, q.shape), F16)[0].transpose(1, 0, 2))
    assert np.array_equal(got, REF["G1"])


def test_g1_flash_attn_varlen_and_g2_causal(gpu):
    """flash_attn_tests.rs:95-138 and models/src/flash_attention.rs:632-705."""
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2)) for n in "qkv")
    cu = np.array([0, 2], np.int32)
    out, _ = gpu_varlen(gpu, q, k, v, cu, cu, 0.5, False, F16, max_q=32, max_k=32)
    assert np.array_equal(round4(to_f32(out, F16).transpose(1, 0, 2)), REF["G1"])
    out, _ = gpu_varlen(gpu, q, k, v, cu, cu, 0.5, True, F16, max_q=32, max_k=32)
    assert np.array_equal(round4(to_f32(out, F16).transpose(1, 0, 2)), REF["G2"])


def test_g1_flash_attn_kv_cache(gpu):
    """flash_attn_tests.rs:194-236: contiguous cache [1,2,3,8], seqlens_k = [2], d = 8."""
    from test_decode_gpu import gpu_decode  # noqa: F401  (shape d=8 goes to the generic kernel)
    q, k, v = (np.ascontiguousarray(REF[n].transpose(1, 0, 2))[None] for n in "qkv")
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v))
    do = gpu.DeviceBuffer(q.nbytes)
    dl = gpu.DeviceBuffer.from_numpy(np.array([2], np.int32))
    gpu.run_mha(dq, dk, dv, do, b=1, h=3, h_k=3, d=8, seqlen_q=2, seqlen_k=2, softmax_scale=0.5, is_bf16=0,
                q_strides=(48, 24, 8), k_strides=(48, 24, 8), v_strides=(48, 24, 8), o_strides=(48, 24, 8),
                cu_seqlens_k=dl, is_seqlens_k_cumulative=False, unpadded_lse=False)
    gpu.synchronize()
    got = round4(to_f32(do.numpy(np.uint16, q.shape), F16)[0].transpose(1, 0, 2))
    assert np.array_equal(got, REF["G1"])


def test_p1_paged_equals_contiguous_bitwise(gpu):
    """flash_attn_tests.rs:140-192 (batch-0 semantics), compared at full precision."""
    base = np.arange(512, dtype=np.float32).astype(np.float16).reshape(32, 2, 8)
    mk = lambda c: (base * np.float16(1.0 / c)).astype(np.float16).view(np.uint16)
    q, k, v = mk(30), mk(40), mk(50)
    cu = np.array([0, 32], np.int32)
    bt = np.arange(2, dtype=np.int32).reshape(1, 2)
    paged, _ = gpu_varlen(gpu, q, k.reshape(2, 16, 2, 8), v.reshape(2, 16, 2, 8), cu, cu, 0.5, False, F16, bt=bt)
    dense, _ = gpu_varlen(gpu, q, k, v, cu, cu, 0.5, False, F16)
    assert np.array_equal(paged, dense)
    assert_close(dense, A.flash_attn_varlen(q, k, v, cu, cu, 0.5, False, F16), F16, what="P1 vs oracle")


def test_p2_kv_cache_equals_varlen_with_block_table(gpu):
    """flash_attn_tests.rs:238-303 (in-bounds restatement with 64 pages)."""
    from test_decode_gpu import gpu_decode
    rng = np.random.default_rng(3)
    kc, vc = rand_half(rng, (64, 16, 2, 8), F16), rand_half(rng, (64, 16, 2, 8), F16)
    q = rand_half(rng, (32, 1, 2, 8), F16)
    bt = np.arange(64, dtype=np.int32).reshape(32, 2)
    cu = np.arange(33, dtype=np.int32)
    b, _ = gpu_varlen(gpu, q[:, 0], kc, vc, cu, cu, 0.5, False, F16, bt=bt)
    # kv-cache entry point, d = 8 -> generic kernel with per-sequence lengths
    dq, dk, dv = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc))
    do = gpu.DeviceBuffer(q.nbytes)
    dl = gpu.DeviceBuffer.from_numpy(np.ones(32, np.int32))
    dbt = gpu.DeviceBuffer.from_numpy(bt)
    gpu.run_mha(dq, dk, dv, do, b=32, h=2, h_k=2, d=8, seqlen_q=1, seqlen_k=32, softmax_scale=0.5, is_bf16=0,
                q_strides=(16, 16, 8), o_strides=(16, 16, 8), k_strides=(256, 16, 8), v_strides=(256, 16, 8),
                cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt, block_table_batch_stride=2,
                page_block_size=16, force_split_kernel=True, unpadded_lse=False)
    gpu.synchronize()
    a = do.numpy(np.uint16, q.shape)
    assert np.array_equal(a[:, 0], b)


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_prefill_fixture(gpu, causal):
    c = CASES
    out, _ = gpu_varlen(gpu, c["p1_q"], c["p1_k"], c["p1_v"], c["p1_cu"], c["p1_cu"], c["p1_scale"], causal, BF16)
    if causal:
        assert_close(out, c["p1_out_kernel"], BF16, atol=ATOL_VS_F32[BF16], what="causal varlen fixture (kernel oracle)")
        assert_close(out, c["p1_out_f32"], BF16, atol=ATOL_VS_F32[BF16], what="causal varlen fixture (f32 oracle)")
    else:
        args = (c["p1_q"], c["p1_k"], c["p1_v"], c["p1_cu"], c["p1_cu"], c["p1_scale"], False, BF16)
        assert_close(out, A.flash_attn_varlen(*args), BF16, atol=ATOL_VS_F32[BF16], what="non-causal varlen (f32 oracle)")


@pytest.mark.parametrize("d", [32, 96, 160, 256])
def test_other_head_sizes_causal_paged_prefix(gpu, d):
    """Head sizes the reference instantiates besides 64/128 (csrc/build.rs:7-74), chunked/prefix
    prefill against the paged cache with Lq < Lk (causal offset Lk - Lq, mask.h:170)."""
    rng = np.random.default_rng(d)
    lens_k = np.array([40, 17, 64], np.int32)
    lens_q = np.array([8, 17, 1], np.int32)
    kc, vc, bt = make_paged_cache(rng, 10, 16, 2, d, BF16, lens_k)
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lens_k)]).astype(np.int32)
    q = rand_half(rng, (int(cu_q[-1]), 4, d), BF16)
    for causal in (True, False):
        out, _ = gpu_varlen(gpu, q, kc, vc, cu_q, cu_k, d ** -0.5, causal, BF16, bt=bt)
        ref = A.flash_attn_varlen(q, kc, vc, cu_q, cu_k, d ** -0.5, causal, BF16, block_table=bt)
        assert_close(out, ref, BF16, atol=ATOL_VS_F32[BF16], what=f"d={d} causal={causal} (f32 oracle)")


@pytest.mark.parametrize("d,dtype", [(32, BF16), (96, F16), (192, BF16), (224, F16), (256, BF16)])
def test_other_head_sizes_tiled_prefill(gpu, d, dtype):
    """The 16-query-row MFMA kernel for the other head sizes (attn_generic.hip: attn_prefill_tile16_kernel): several row blocks per sequence,
    lengths that are not multiples of the 16-key tile, Lq < Lk, Lq > Lk (rows that see no key: O = 0, LSE = +inf), an empty sequence, GQA,
    ALiBi, paged and contiguous K / V -- against the f32 oracle, and against the row-per-wavefront kernel it replaces (LSE included)."""
    rng = np.random.default_rng(1000 + d)
    lens_k = np.array([150, 33, 16, 5, 0, 97], np.int32)
    lens_q = np.array([150, 20, 16, 9, 3, 1], np.int32)
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lens_k)]).astype(np.int32)
    h, hk = 6, 2
    q = rand_half(rng, (int(cu_q[-1]), h, d), dtype)
    kc, vc, bt = make_paged_cache(rng, 24, 16, hk, d, dtype, lens_k)
    kd, vd = rand_half(rng, (int(cu_k[-1]), hk, d), dtype), rand_half(rng, (int(cu_k[-1]), hk, d), dtype)
    slopes = (2.0 ** -np.arange(1, h + 1)).astype(np.float32)
    for causal in (True, False):
        for alibi in (None, slopes):
            for paged in (True, False):
                k, v, kw = (kc, vc, dict(bt=bt)) if paged else (kd, vd, {})
                out, lse = gpu_varlen(gpu, q, k, v, cu_q, cu_k, d ** -0.5, causal, dtype, alibi=alibi, **kw)
                okw = dict(block_table=bt) if paged else {}
                ref = A.flash_attn_varlen(q, k, v, cu_q, cu_k, d ** -0.5, causal, dtype, alibi_slopes=alibi, **okw)
                what = f"d={d} causal={causal} alibi={alibi is not None} paged={paged}"
                assert_close(out, ref, dtype, atol=ATOL_VS_F32[dtype], what=what + " (f32 oracle)")
                for other in ("rq1", "16", "0"):      # 64-row workgroups (one row block per wavefront); the 16-row kernel alone; the row-per-wavefront kernel
                    name, val, dflt = (b"generic_prefill_rq", 1, 0) if other == "rq1" else (b"generic_prefill_tile", int(other), 64)
                    assert gpu.lib.atoma_set_option(name, val) == 0
                    try:
                        out_o, lse_o = gpu_varlen(gpu, q, k, v, cu_q, cu_k, d ** -0.5, causal, dtype, alibi=alibi, **kw)
                    finally:
                        gpu.lib.atoma_set_option(name, dflt)
                    if other != "0":
                        assert_close(out_o, ref, dtype, atol=ATOL_VS_F32[dtype], what=what + f" (kernel {other}, f32 oracle)")
                    assert_close(out, out_o, dtype, atol=ATOL_VS_F32[dtype], what=what + f" (kernel {other})")
                    assert np.array_equal(np.isinf(lse), np.isinf(lse_o)), what
                    fin = np.isfinite(lse_o)
                    assert np.allclose(lse[fin], lse_o[fin], atol=2e-3, rtol=1e-3), what


@pytest.mark.parametrize("scale", [-0.125, 0.0])
def test_other_head_sizes_prefill_with_a_non_positive_scale(gpu, scale):
    """A zero or negative softmax_scale is legal for the reference (any finite scale); the tiled prefill kernels assume scale > 0 (they keep
    scores raw and scale the row maximum), so such calls must take the row-per-wavefront kernel and still equal the oracle."""
    rng = np.random.default_rng(77)
    d, h, hk = 96, 4, 2
    lens = np.array([70, 33], np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    q, k, v = rand_half(rng, (int(cu[-1]), h, d), BF16), rand_half(rng, (int(cu[-1]), hk, d), BF16), rand_half(rng, (int(cu[-1]), hk, d), BF16)
    # non-causal only: with a mask the reference's own arithmetic (-inf * scale) yields NaN rows for these scales -- nothing to compare
    out, _ = gpu_varlen(gpu, q, k, v, cu, cu, scale, False, BF16)
    ref = A.flash_attn_varlen(q, k, v, cu, cu, scale, False, BF16)
    assert np.isfinite(to_f32(ref, BF16)).all() and np.isfinite(to_f32(out, BF16)).all()
    assert_close(out, ref, BF16, atol=ATOL_VS_F32[BF16], what=f"scale {scale}")


def test_alibi_causal_and_non_causal(gpu):
    rng = np.random.default_rng(21)
    cu = np.array([0, 20, 50], np.int32)
    q, k, v = rand_half(rng, (50, 4, 64), F16), rand_half(rng, (50, 2, 64), F16), rand_half(rng, (50, 2, 64), F16)
    slopes = np.array([0.5, 0.25, 0.125, 0.0625], np.float32)
    for causal in (True, False):
        out, _ = gpu_varlen(gpu, q, k, v, cu, cu, 0.125, causal, F16, alibi=slopes)
        ref = A.flash_attn_varlen(q, k, v, cu, cu, 0.125, causal, F16, alibi_slopes=slopes)
        assert_close(out, ref, F16, atol=ATOL_VS_F32[F16], what=f"alibi causal={causal}")


def test_run_mha_argument_errors(gpu):
    z = gpu.DeviceBuffer(4096)
    common = dict(b=1, h=3, h_k=2, d=8, seqlen_q=1, seqlen_k=1, softmax_scale=1.0, is_bf16=1,
                  q_strides=(8, 8, 8), k_strides=(8, 8, 8), v_strides=(8, 8, 8), o_strides=(8, 8, 8))
    with pytest.raises(RuntimeError, match="must divide"):
        gpu.run_mha(z, z, z, z, **common)
    common.update(h=2, d=12)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        gpu.run_mha(z, z, z, z, **common)
    common.update(d=264)
    with pytest.raises(RuntimeError, match="at most 256"):
        gpu.run_mha(z, z, z, z, **common)
    common.update(d=8)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        gpu.run_mha(z, z, z, z, block_table=z, block_table_batch_stride=1, page_block_size=8, **common)
    gpu.lib.atoma_clear_error()
