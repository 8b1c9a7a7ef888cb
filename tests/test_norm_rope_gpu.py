"""RMSNorm / RoPE kernels against the oracle's restatement of Candle's semantics (parity is
unpinned by the reference's own tests: oracle/norm_rope_oracle.py says why)."""
import os

import numpy as np
import pytest

from oracle import norm_rope_oracle as NR
from oracle.halfs import F16, BF16, to_f32
from util import rand_half, assert_close

pytestmark = pytest.mark.gpu
CASES = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_cases.npz"))


def ulps_apart(a_bits, b_bits, dtype):
    """distance in units of the last place between two storage-form arrays (same sign assumed)."""
    a, b = a_bits.astype(np.int32), b_bits.astype(np.int32)
    return np.abs(a - b)


def gpu_rms_norm(gpu, x, w, eps, dtype, x_row_stride=None):
    rows, hidden = x.shape
    dx, dw = gpu.DeviceBuffer.from_numpy(x), gpu.DeviceBuffer.from_numpy(w)
    dy = gpu.DeviceBuffer(rows * hidden * 2)
    rc = gpu.lib.atoma_rms_norm(dx.ptr, dw.ptr, dy.ptr, rows, hidden, x_row_stride or hidden, hidden, eps, dtype, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return dy.numpy(np.uint16, (rows, hidden))


def test_rms_norm_fixture(gpu):
    c = CASES
    got = gpu_rms_norm(gpu, c["n1_x"], c["n1_w"], 1e-5, BF16)
    assert ulps_apart(got, c["n1_y"], BF16).max() <= 1     # f32 sum order + rsqrt vs 1/sqrt: one rounding unit


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("rows,hidden", [(256, 4096), (2048, 4096), (3, 2048), (7, 8192), (5, 16384), (4, 100)])
def test_rms_norm_llama_shapes(gpu, dtype, rows, hidden):
    rng = np.random.default_rng(rows + hidden)
    x, w = rand_half(rng, (rows, hidden), dtype), rand_half(rng, (hidden,), dtype, 0.5)
    got = gpu_rms_norm(gpu, x, w, 1e-5, dtype)
    ref = NR.rms_norm(x, w, 1e-5, dtype)
    d = ulps_apart(got, ref, dtype)
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())


def gpu_rope(gpu, x, cos, sin, pos, dtype, per_op=1, inplace=False):
    T, H, d = x.shape
    dx = gpu.DeviceBuffer.from_numpy(x)
    dy = dx if inplace else gpu.DeviceBuffer(x.nbytes)
    dc, ds = gpu.DeviceBuffer.from_numpy(cos), gpu.DeviceBuffer.from_numpy(sin)
    dp = gpu.DeviceBuffer.from_numpy(np.asarray(pos, np.int64))
    rc = gpu.lib.atoma_rope(dx.ptr, dy.ptr, dc.ptr, ds.ptr, dp.ptr, T, H, d, H * d, d, H * d, d, dtype, per_op, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return dy.numpy(np.uint16, x.shape)


def test_rope_fixture_bit_exact(gpu):
    """Candle evaluates rope in the tensor dtype (every product and the sum rounded): with the
    same table the kernel's per-op mode is bit-exact against the oracle."""
    c = CASES
    got = gpu_rope(gpu, c["o1_x"], c["o1_cos"], c["o1_sin"], c["o1_pos"], BF16)
    assert np.array_equal(got, c["o1_y"])
    assert np.array_equal(gpu_rope(gpu, c["o1_x"], c["o1_cos"], c["o1_sin"], c["o1_pos"], BF16, inplace=True), c["o1_y"])


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("T,H,d", [(256, 32, 128), (2048, 8, 128), (33, 32, 64), (5, 3, 32)])
def test_rope_llama_shapes_bit_exact(gpu, dtype, T, H, d):
    rng = np.random.default_rng(T + H + d)
    cos, sin = NR.rope_table(4096, d, 500000.0, dtype)
    x = rand_half(rng, (T, H, d), dtype)
    pos = rng.integers(0, 4096, T)
    assert np.array_equal(gpu_rope(gpu, x, cos, sin, pos, dtype), NR.rope(x, cos, sin, pos, dtype))
    fused = gpu_rope(gpu, x, cos, sin, pos, dtype, per_op=0)
    ref = NR.rope(x, cos, sin, pos, dtype, mode="fused")
    assert ulps_apart(fused, ref, dtype).max() <= 1            # fma contraction vs two roundings in f32


def test_rope_qk_in_place_one_launch(gpu):
    rng = np.random.default_rng(31)
    T, hq, hk, d = 100, 32, 8, 128
    cos, sin = NR.rope_table(512, d, 500000.0, BF16)
    q, k = rand_half(rng, (T, hq, d), BF16), rand_half(rng, (T, hk, d), BF16)
    pos = rng.integers(0, 512, T)
    dq, dk = gpu.DeviceBuffer.from_numpy(q), gpu.DeviceBuffer.from_numpy(k)
    dc, ds = gpu.DeviceBuffer.from_numpy(cos), gpu.DeviceBuffer.from_numpy(sin)
    dp = gpu.DeviceBuffer.from_numpy(np.asarray(pos, np.int64))
    rc = gpu.lib.atoma_rope_qk(dq.ptr, dk.ptr, dc.ptr, ds.ptr, dp.ptr, T, hq, hk, d, hq * d, hk * d, BF16, 1, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(dq.numpy(np.uint16, q.shape), NR.rope(q, cos, sin, pos, BF16))
    assert np.array_equal(dk.numpy(np.uint16, k.shape), NR.rope(k, cos, sin, pos, BF16))


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("T,h,hk,d,page", [(256, 32, 8, 128, 16), (37, 8, 2, 64, 16), (5, 4, 4, 32, 8), (1, 32, 8, 128, 16)])
def test_fused_rope_qk_cache_bit_exact(gpu, dtype, T, h, hk, d, page):
    """atoma_rope_qk_cache == rope(q), rope(k) (Candle per-op rounding) followed by reshape_and_cache_flash(rope(k), v),
    bit for bit; q and k slices of a fused qkv projection (token stride (h+2hk)*d); padding tokens (slot -1) leave the
    caches untouched."""
    from oracle import cache_oracle as CO
    rng = np.random.default_rng(T + h + d)
    nb = (T + page - 1) // page + 2
    qkv = rand_half(rng, (T, (h + 2 * hk) * d), dtype)
    q = np.ascontiguousarray(qkv[:, :h * d]).reshape(T, h, d)
    k = np.ascontiguousarray(qkv[:, h * d:(h + hk) * d]).reshape(T, hk, d)
    v = np.ascontiguousarray(qkv[:, (h + hk) * d:]).reshape(T, hk, d)
    kc, vc = rand_half(rng, (nb, page, hk, d), dtype), rand_half(rng, (nb, page, hk, d), dtype)
    cos, sin = NR.rope_table(2048, d, 500000.0, dtype)
    pos = rng.integers(0, 2048, T).astype(np.int64)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    if T > 4:
        slots[rng.integers(0, T, 2)] = -1
    dqkv, dkc, dvc = (gpu.DeviceBuffer.from_numpy(a) for a in (qkv, kc, vc))
    dc, ds = gpu.DeviceBuffer.from_numpy(cos), gpu.DeviceBuffer.from_numpy(sin)
    dp, dsl = gpu.DeviceBuffer.from_numpy(pos), gpu.DeviceBuffer.from_numpy(slots)
    row = (h + 2 * hk) * d
    rc = gpu.lib.atoma_rope_qk_cache(dqkv.ptr, dqkv.ptr + h * d * 2, dqkv.ptr + (h + hk) * d * 2, dkc.ptr, dvc.ptr, dsl.ptr,
                                     dc.ptr, ds.ptr, dp.ptr, T, h, hk, d, row, row, row, page * hk * d, page, dtype, 1, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    got = dqkv.numpy(np.uint16, qkv.shape)
    qr, kr = NR.rope(q, cos, sin, pos, dtype), NR.rope(k, cos, sin, pos, dtype)
    assert np.array_equal(got[:, :h * d].reshape(T, h, d), qr)
    assert np.array_equal(got[:, h * d:(h + hk) * d].reshape(T, hk, d), kr)
    assert np.array_equal(got[:, (h + hk) * d:].reshape(T, hk, d), v)          # v is only read
    CO.reshape_and_cache_flash(kr, v, kc, vc, slots)
    assert np.array_equal(dkc.numpy(np.uint16, kc.shape), kc) and np.array_equal(dvc.numpy(np.uint16, vc.shape), vc)


def test_fused_rope_qk_cache_rejects_bad_arguments(gpu):
    d = gpu.DeviceBuffer(4096)
    args = lambda **kw: [d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, 1, 1, 1, kw.get("hd", 64), 64, 64, 64,
                         kw.get("bs", 1024), kw.get("page", 16), kw.get("dtype", BF16), 1, None]
    assert gpu.lib.atoma_rope_qk_cache(*args(hd=24)) == -1 and "head_dim" in gpu.last_error()
    assert gpu.lib.atoma_rope_qk_cache(*args(page=0)) == -1 and "page_size" in gpu.last_error()
    assert gpu.lib.atoma_rope_qk_cache(*args(dtype=7)) == -1 and "dtype" in gpu.last_error()
    assert gpu.lib.atoma_rope_qk_cache(*args(bs=1021)) == -1 and "strides" in gpu.last_error()


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("rows,hidden", [(1, 4096), (7, 2048), (256, 4096), (3, 8192), (5, 520)])
def test_add_rms_norm_is_bit_identical_to_the_two_ops(gpu, dtype, rows, hidden):
    """atoma_add_rms_norm = atoma_add then atoma_rms_norm, same rounding points (the sum is rounded before it is normalised)."""
    rng = np.random.default_rng(rows + hidden)
    a, b = rand_half(rng, (rows, hidden), dtype), rand_half(rng, (rows, hidden), dtype)
    w = rand_half(rng, (hidden,), dtype)
    da, db, dw = (gpu.DeviceBuffer.from_numpy(t) for t in (a, b, w))
    s1, y1, s2, y2 = (gpu.DeviceBuffer(a.nbytes) for _ in range(4))
    assert gpu.lib.atoma_add(da.ptr, db.ptr, s1.ptr, rows * hidden, dtype, None) == 0, gpu.last_error()
    assert gpu.lib.atoma_rms_norm(s1.ptr, dw.ptr, y1.ptr, rows, hidden, hidden, hidden, 1e-5, dtype, None) == 0, gpu.last_error()
    assert gpu.lib.atoma_add_rms_norm(da.ptr, db.ptr, dw.ptr, s2.ptr, y2.ptr, rows, hidden, hidden, hidden, hidden, hidden, 1e-5, dtype, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(s1.numpy(np.uint16, a.shape), s2.numpy(np.uint16, a.shape))
    assert np.array_equal(y1.numpy(np.uint16, a.shape), y2.numpy(np.uint16, a.shape))
    assert gpu.lib.atoma_add_rms_norm(da.ptr, db.ptr, dw.ptr, s2.ptr, y2.ptr, rows, 4100, 4100, 4100, 4100, 4100, 1e-5, dtype, None) == -1


TORCH_FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "norm_rope_torch.npz"))


@pytest.mark.parametrize("tag,dtype", [("bf16", BF16), ("f16", F16)])
def test_kernels_against_torch_fixtures(gpu, tag, dtype):
    """The device kernels against values torch produced in the build container (tests/golden/gen_norm_rope_torch.py): a
    third implementation beside the kernel and the numpy restatement (Candle's source is absent: parity unpinned)."""
    fx = TORCH_FX
    got = gpu_rms_norm(gpu, fx[f"rms_{tag}_x"], fx[f"rms_{tag}_w"], float(fx[f"rms_{tag}_eps"]), dtype)
    d = ulps_apart(got, fx[f"rms_{tag}_y"], dtype)
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())
    x, pos, cos, sin = fx[f"rope_{tag}_x"], fx[f"rope_{tag}_pos"], fx[f"tab_{tag}_cos"], fx[f"tab_{tag}_sin"]
    assert np.array_equal(gpu_rope(gpu, x, cos, sin, pos, dtype, per_op=1), fx[f"rope_{tag}_per_op"])
    fused = gpu_rope(gpu, x, cos, sin, pos, dtype, per_op=0)
    d = ulps_apart(fused, fx[f"rope_{tag}_fused"], dtype)
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())     # fma contraction: a boundary flip at most


def test_rope_positions_beyond_the_table_read_its_last_row_when_the_option_is_set(gpu):
    """the FFI carries no table length.  With atoma_set_option("rope_table_rows", n) a position >= n (or < 0) uses the
    table's last (first) row instead of memory behind it; unset, the kernels behave as before."""
    rng = np.random.default_rng(77)
    T, h, d, rows = 6, 4, 64, 10
    x = rand_half(rng, (T, h, d), BF16)
    cos, sin = rand_half(rng, (rows, d // 2), BF16), rand_half(rng, (rows, d // 2), BF16)
    pos = np.array([0, 9, 10, 500, -3, 4], np.int64)
    clamped = np.clip(pos, 0, rows - 1)
    dx, dc, ds = (gpu.DeviceBuffer.from_numpy(a) for a in (x, cos, sin))
    outs = []
    try:
        assert gpu.lib.atoma_set_option(b"rope_table_rows", rows) == 0
        for p_ in (pos, clamped):
            dp, dy = gpu.DeviceBuffer.from_numpy(p_), gpu.DeviceBuffer(x.size * 2)
            assert gpu.lib.atoma_rope(dx.ptr, dy.ptr, dc.ptr, ds.ptr, dp.ptr, T, h, d, h * d, d, h * d, d, BF16, 1, None) == 0, gpu.last_error()
            gpu.synchronize()
            outs.append(dy.numpy(np.uint16, x.shape))
    finally:
        gpu.lib.atoma_set_option(b"rope_table_rows", 0)
    assert np.array_equal(outs[0], outs[1])
