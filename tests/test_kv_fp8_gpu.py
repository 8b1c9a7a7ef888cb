"""fp8 (e4m3fn) KV cache on the device: the cache write is bit-exact against oracle/fp8_oracle.py (itself pinned to torch's
float8_e4m3fn), the decode kernel is the 16-bit path's arithmetic on the dequantised values: same tolerance policy against
the f32 oracle evaluated on e4m3 * scale, and exact invariants (permutation of the block table, empty sequences)."""
import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle import fp8_oracle as F8
from oracle import norm_rope_oracle as NR
from oracle.halfs import F16, BF16, to_f32, from_f32
from util import rand_half, assert_close, attn_atol

pytestmark = pytest.mark.gpu


def dev_scales(gpu, ks, vs):
    return gpu.DeviceBuffer.from_numpy(np.asarray(ks, np.float32)), gpu.DeviceBuffer.from_numpy(np.asarray(vs, np.float32))


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("T,hk,d,page", [(37, 8, 128, 16), (5, 2, 64, 32), (300, 1, 128, 16)])
def test_cache_write_bit_exact(gpu, dtype, T, hk, d, page):
    rng = np.random.default_rng(T + hk)
    nb = (T + page - 1) // page + 3
    k, v = rand_half(rng, (T, hk, d), dtype, 2.0), rand_half(rng, (T, hk, d), dtype, 2.0)
    k[0, 0, :4] = from_f32(np.float32([1e4, -1e4, 0.0, -0.0]), dtype)          # saturation and signed zeros
    ks, vs = rng.uniform(0.005, 0.05, hk).astype(np.float32), rng.uniform(0.005, 0.05, hk).astype(np.float32)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    slots[rng.integers(0, T, 3)] = -1                                           # padding tokens
    kc = rng.integers(0, 256, (nb, page, hk, d)).astype(np.uint8)               # untouched bytes must survive
    vc = rng.integers(0, 256, (nb, page, hk, d)).astype(np.uint8)
    dk, dv, dkc, dvc, dsl = (gpu.DeviceBuffer.from_numpy(a) for a in (k, v, kc, vc, slots))
    dks, dvs = dev_scales(gpu, ks, vs)
    rc = gpu.lib.atoma_reshape_and_cache_flash_fp8(dk.ptr, dv.ptr, dkc.ptr, dvc.ptr, dsl.ptr, dks.ptr, dvs.ptr, page * hk * d, T, hk, d, page,
                                                   hk * d, hk * d, dtype, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    F8.reshape_and_cache_flash_fp8(k, v, kc, vc, slots, ks, vs, dtype)
    assert np.array_equal(dkc.numpy(np.uint8, kc.shape), kc) and np.array_equal(dvc.numpy(np.uint8, vc.shape), vc)


def test_rope_cache_fp8_equals_two_ops(gpu):
    rng = np.random.default_rng(4)
    T, h, hk, d, page, nb = 9, 8, 2, 128, 16, 4
    qkv = rand_half(rng, (T, (h + 2 * hk) * d), BF16)
    cos, sin = NR.rope_table(64, d, 500000.0, BF16)
    pos = rng.integers(0, 64, T).astype(np.int64)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    ks, vs = np.float32([0.03, 0.04]), np.float32([0.02, 0.05])
    dqkv, dc, ds, dp, dsl = (gpu.DeviceBuffer.from_numpy(a) for a in (qkv, cos, sin, pos, slots))
    dkc, dvc = gpu.DeviceBuffer.zeros((nb, page, hk, d), np.uint8), gpu.DeviceBuffer.zeros((nb, page, hk, d), np.uint8)
    dks, dvs = dev_scales(gpu, ks, vs)
    W = (h + 2 * hk) * d
    rc = gpu.lib.atoma_rope_qk_cache_fp8(dqkv.ptr, dqkv.ptr + h * d * 2, dqkv.ptr + (h + hk) * d * 2, dkc.ptr, dvc.ptr, dsl.ptr, dks.ptr, dvs.ptr,
                                         dc.ptr, ds.ptr, dp.ptr, T, h, hk, d, W, W, W, page * hk * d, page, BF16, 1, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    q = np.ascontiguousarray(qkv[:, :h * d]).reshape(T, h, d)
    k = np.ascontiguousarray(qkv[:, h * d:(h + hk) * d]).reshape(T, hk, d)
    v = np.ascontiguousarray(qkv[:, (h + hk) * d:]).reshape(T, hk, d)
    qr, kr = NR.rope(q, cos, sin, pos, BF16), NR.rope(k, cos, sin, pos, BF16)
    kc, vc = np.zeros((nb, page, hk, d), np.uint8), np.zeros((nb, page, hk, d), np.uint8)
    F8.reshape_and_cache_flash_fp8(kr, v, kc, vc, slots, ks, vs, BF16)
    out = dqkv.numpy(np.uint16, qkv.shape)
    assert np.array_equal(out[:, :h * d].reshape(T, h, d), qr) and np.array_equal(out[:, h * d:(h + hk) * d].reshape(T, hk, d), kr)
    assert np.array_equal(dkc.numpy(np.uint8, kc.shape), kc) and np.array_equal(dvc.numpy(np.uint8, vc.shape), vc)


def gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, scale, dtype, cache_strides=None):
    """cache_strides = (block, row, head) in bytes for another physical layout of the same [nb, page, hk, d] pages (kc8 / vc8 are then
    the flat byte images)"""
    B, h, d = q.shape
    if cache_strides is None:
        nb, page, hk, _ = kc8.shape
        cache_strides = (page * hk * d, hk * d, d)
    else:
        page, hk = cache_strides[3], cache_strides[4]
    dq, dk, dv, dbt, dl = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc8, vc8, np.ascontiguousarray(bt, np.int32), np.ascontiguousarray(lens, np.int32)))
    dks, dvs = dev_scales(gpu, ks, vs)
    do = gpu.DeviceBuffer(q.nbytes)
    do.fill_bytes(0xFF)
    rc = gpu.lib.atoma_paged_decode_fp8(dq.ptr, dk.ptr, dv.ptr, do.ptr, dks.ptr, dvs.ptr, dbt.ptr, dl.ptr, B, h, hk, d, bt.shape[1], page,
                                        h * d, d, h * d, d, cache_strides[0], cache_strides[1], cache_strides[2], float(scale), dtype, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return do.numpy(np.uint16, q.shape)


def make_fp8_cache(rng, nb, page, hk, d, lens):
    from util import make_paged_cache
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    ks, vs = rng.uniform(0.01, 0.03, hk).astype(np.float32), rng.uniform(0.01, 0.03, hk).astype(np.float32)
    kc8 = F8.quantize(kc.reshape(-1, hk, d), BF16, ks).reshape(nb, page, hk, d)
    vc8 = F8.quantize(vc.reshape(-1, hk, d), BF16, vs).reshape(nb, page, hk, d)
    return kc8, vc8, ks, vs, bt


def oracle_decode(q, kc8, vc8, ks, vs, bt, lens, scale, dtype):
    """fa_acausal (f32) over the dequantised cache, one rounding to the output dtype."""
    kf, vf = F8.dequantize(kc8, ks), F8.dequantize(vc8, vs)
    qf = to_f32(q, dtype)
    out = np.zeros(qf.shape, np.float32)
    page = kc8.shape[1]
    for b in range(q.shape[0]):
        L = int(lens[b])
        if L:
            kb, vb = A.gather_paged(kf, bt[b], L, page), A.gather_paged(vf, bt[b], L, page)
            out[b], _ = A.attend_rows(qf[b][None], kb, vb, scale)
    return from_f32(out, dtype)


@pytest.fixture(params=[1, 0], ids=["mfma-qk", "dot2-qk"])
def qk_variant(gpu, request):
    """both fp8 decode kernels: q.K^T on the matrix cores (default) and on v_dot2c"""
    assert gpu.lib.atoma_set_option(b"decode_fp8_mqk", request.param) == 0
    yield request.param
    gpu.lib.atoma_set_option(b"decode_fp8_mqk", 1)


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("h,hk,page", [(32, 8, 16), (8, 8, 16), (16, 8, 32), (6, 2, 16), (16, 2, 64), (16, 1, 16), (40, 2, 16), (5, 1, 32), (13, 1, 16)])
def test_decode_fp8_matches_oracle_ragged(gpu, qk_variant, dtype, h, hk, page):
    """group sizes 1, 2, 3, 4, 5, 8, 13, 16 and 20 (= 16 + 4: two passes of the matrix-core kernel) (one pass on the matrix cores, two chunks of 4 in the dot2 kernel), pages of 16 / 32 / 64, ragged lengths incl. the empty sequence"""
    rng = np.random.default_rng(h * 3 + hk + page)
    d = 128
    lens = np.array([0, 1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 200, 333, 600], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 3
    kc8, vc8, ks, vs, bt = make_fp8_cache(rng, nb, page, hk, d, lens)
    q = rand_half(rng, (len(lens), h, d), dtype)
    scale = np.float32(d ** -0.5)
    out = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, scale, dtype)
    ref = oracle_decode(q, kc8, vc8, ks, vs, bt, lens, scale, dtype)
    for i, L in enumerate(lens):       # tolerance policy of the 16-bit path (tests/util.py): the arithmetic after dequantisation is the same
        assert_close(out[i], ref[i], dtype, atol=attn_atol(dtype, int(L)), what=f"fp8 decode L={L}")
    assert not out[0].any(), "empty sequence must produce exact zeros"


def test_decode_fp8_split_kv_balanced_and_invariants(gpu, qk_variant):
    rng = np.random.default_rng(9)
    h, hk, d, page = 32, 8, 128, 16
    # (a) one long sequence: split over many wavefronts + combine
    lens = np.array([5000], np.int32)
    kc8, vc8, ks, vs, bt = make_fp8_cache(rng, 320, page, hk, d, lens)
    q = rand_half(rng, (1, h, d), BF16)
    sc = np.float32(d ** -0.5)
    out = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
    assert_close(out, oracle_decode(q, kc8, vc8, ks, vs, bt, lens, sc, BF16), BF16, atol=1e-3, what="fp8 split-KV")
    # (b) a ragged batch large enough for the balanced mode (b * h_k >= half the resident wavefronts)
    B = 160
    lens = rng.integers(1, 700, B).astype(np.int32)
    lens[7] = 0
    nb = int(sum((L + page - 1) // page for L in lens)) + 2
    kc8, vc8, ks, vs, bt = make_fp8_cache(rng, nb, page, hk, d, lens)
    q = rand_half(rng, (B, h, d), BF16)
    out = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
    ref = oracle_decode(q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
    for i, L in enumerate(lens):
        assert_close(out[i], ref[i], BF16, atol=attn_atol(BF16, int(L)), what=f"fp8 balanced L={L}")
    # (c) the result does not depend on where pages live: permute the physical pages
    perm = rng.permutation(nb)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(nb)
    out2 = gpu_decode_fp8(gpu, q, kc8[inv], vc8[inv], ks, vs, perm[bt].astype(np.int32), lens, sc, BF16)
    assert np.array_equal(out, out2)
    # (d) scaling the V scale by 2 doubles the output exactly (power of two), scaling K's scale equals scaling softmax_scale
    base8 = gpu_decode_fp8(gpu, q[:8], kc8, vc8, ks, vs, bt[:8], lens[:8], sc, BF16)          # same launch shape as the scaled runs below
    out3 = gpu_decode_fp8(gpu, q[:8], kc8, vc8, ks, vs * 2, bt[:8], lens[:8], sc, BF16)
    assert np.array_equal(to_f32(out3, BF16), 2 * to_f32(base8, BF16))
    out4 = gpu_decode_fp8(gpu, q[:8], kc8, vc8, ks * 4, vs, bt[:8], lens[:8], sc / 4, BF16)
    assert np.array_equal(out4, base8)


@pytest.mark.parametrize("h,hk", [(32, 8), (8, 2), (4, 4)])
def test_decode_fp8_other_page_layouts_bit_identical(gpu, qk_variant, h, hk):
    """The ABI takes the cache strides: head-major pages [nb][hk][page][d] and pages with padded rows must give the SAME bits as the
    reference layout (the kernel's K / V fetch addresses are all derived from the strides; the arithmetic does not change) --
    resident batch (full-line K fetch), a split sequence, and a ragged batch in the balanced mode."""
    rng = np.random.default_rng(41 + h)
    d, page = 128, 16
    sc = np.float32(d ** -0.5)
    for lens in (np.array([0, 1, 17, 64, 333, 600], np.int32), np.array([5000], np.int32), rng.integers(1, 700, 160).astype(np.int32)):
        nb = int(sum((L + page - 1) // page for L in lens)) + 2
        kc8, vc8, ks, vs, bt = make_fp8_cache(rng, nb, page, hk, d, lens)
        q = rand_half(rng, (len(lens), h, d), BF16)
        base = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
        # head-major pages
        khm, vhm = np.ascontiguousarray(kc8.transpose(0, 2, 1, 3)), np.ascontiguousarray(vc8.transpose(0, 2, 1, 3))
        out = gpu_decode_fp8(gpu, q, khm, vhm, ks, vs, bt, lens, sc, BF16, cache_strides=(page * hk * d, d, page * d, page, hk))
        assert np.array_equal(out, base), "head-major pages"
        # rows padded by 64 bytes, pages by one row
        row = hk * d + 64
        kp, vp = np.zeros((nb, page + 1, row), np.uint8), np.zeros((nb, page + 1, row), np.uint8)
        kp[:, :page, :hk * d] = kc8.reshape(nb, page, hk * d)
        vp[:, :page, :hk * d] = vc8.reshape(nb, page, hk * d)
        kp[:, :, hk * d:], vp[:, :, hk * d:] = 0x7E, 0x7E        # large finite codes in the padding: a wrong address shows
        out = gpu_decode_fp8(gpu, q, kp, vp, ks, vs, bt, lens, sc, BF16, cache_strides=((page + 1) * row, row, d, page, hk))
        assert np.array_equal(out, base), "padded rows"


def test_decode_fp8_launch_orders_bit_identical(gpu):
    """The workgroup order (kv head slowest, the default; kv head fastest; 8 wavefronts = the 8 kv heads of a sequence per workgroup,
    option decode_fp8_wg) changes where a piece runs, never what it computes: same bits for a split sequence, a small resident batch
    and a ragged batch on the balanced line."""
    rng = np.random.default_rng(77)
    h, hk, d, page = 32, 8, 128, 16
    sc = np.float32(d ** -0.5)
    for lens in (np.array([5000, 4000], np.int32), np.array([0, 1, 17, 64, 333, 600] * 3, np.int32), rng.integers(1, 700, 160).astype(np.int32)):
        nb = int(sum((L + page - 1) // page for L in lens)) + 2
        kc8, vc8, ks, vs, bt = make_fp8_cache(rng, nb, page, hk, d, lens)
        q = rand_half(rng, (len(lens), h, d), BF16)
        base = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
        for opt, val, restore in ((b"decode_head_major", 0, 1), (b"decode_fp8_wg", 2, 0)):
            assert gpu.lib.atoma_set_option(opt, val) == 0
            try:
                out = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, sc, BF16)
            finally:
                gpu.lib.atoma_set_option(opt, restore)
            assert np.array_equal(out, base), opt


def test_decode_fp8_rejects_bad_arguments(gpu):
    d = gpu.DeviceBuffer(4096)
    args = lambda **kw: [d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, kw.get("b", 1), kw.get("h", 8), kw.get("hk", 2), kw.get("d", 128), 4,
                         kw.get("page", 16), 1024, 128, 1024, 128, 16 * 2 * 128, 2 * 128, 128, 0.088, kw.get("dtype", 1), None]
    assert gpu.lib.atoma_paged_decode_fp8(*args(d=64)) == -1 and "head_dim" in gpu.last_error()
    assert gpu.lib.atoma_paged_decode_fp8(*args(h=7)) == -1 and "head counts" in gpu.last_error()
    assert gpu.lib.atoma_paged_decode_fp8(*args(page=8)) == -1 and "page_size" in gpu.last_error()
    assert gpu.lib.atoma_paged_decode_fp8(*args(dtype=2)) == -1 and "dtype" in gpu.last_error()
    assert gpu.lib.atoma_paged_decode_fp8(*args(b=0)) == 0


@pytest.mark.parametrize("code", [0x7F, 0xFF], ids=["+nan", "-nan"])
@pytest.mark.parametrize("h,hk,page", [(32, 8, 16), (8, 8, 32), (16, 1, 16)])
def test_decode_fp8_never_written_slots_do_not_reach_the_output(gpu, qk_variant, h, hk, page, code):
    """e4m3fn NaN codes in the slots of the last page behind the sequence (p = 0 there, but the products still run) and in pages
    nobody owns: every output bit stays where it was"""
    from util import poison_unwritten_slots
    rng = np.random.default_rng(h + hk + page + code)
    d = 128
    lens = np.array([1, 2, 15, 17, 31, 33, 100, 333, 1000], np.int32)
    nb = int(sum((L + page - 1) // page for L in lens)) + 4
    kc8, vc8, ks, vs, bt = make_fp8_cache(rng, nb, page, hk, d, lens)
    q = rand_half(rng, (len(lens), h, d), BF16)
    scale = np.float32(d ** -0.5)
    clean = gpu_decode_fp8(gpu, q, kc8, vc8, ks, vs, bt, lens, scale, BF16)
    kp, vp = poison_unwritten_slots(kc8, vc8, bt, lens, code)
    got = gpu_decode_fp8(gpu, q, kp, vp, ks, vs, bt, lens, scale, BF16)
    assert np.isfinite(to_f32(got, BF16)).all()
    assert np.array_equal(got, clean)
