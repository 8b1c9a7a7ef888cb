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
