"""Small-batch linear layer (atoma_linear_decode) against the oracle: fp32 accumulation on the matrix cores vs the exactly
accumulated product, at most one unit in the last place apart."""
import numpy as np
import pytest

from oracle import linear_oracle as LO
from oracle.halfs import F16, BF16, to_f32
from util import rand_half

pytestmark = pytest.mark.gpu


def gpu_linear(gpu, x, w, dtype, x_stride=None, y_stride=None):
    B, K = x.shape[0], w.shape[1]
    N = w.shape[0]
    dx, dw = gpu.DeviceBuffer.from_numpy(x), gpu.DeviceBuffer.from_numpy(w)
    ys = y_stride or N
    dy = gpu.DeviceBuffer(max(B, 1) * ys * 2)
    dy.fill_bytes(0xFF)
    rc = gpu.lib.atoma_linear_decode(dx.ptr, dw.ptr, dy.ptr, B, K, N, x_stride or x.shape[1], K, ys, dtype, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return dy.numpy(np.uint16, (max(B, 1), ys))[:B, :N]


def check(got, ref, dtype):
    """fp32 accumulation (error ~ 6e-8 * sqrt(K) * |partial sums| <= 3e-5 for the O(1) outputs of these tests) and
    one rounding: |got - ref| <= one unit in the last place of the storage dtype + 3e-5 (the absolute part matters for
    the few outputs that cancel to nearly zero), and nearly all outputs bit-identical."""
    g, r = to_f32(got, dtype), to_f32(ref, dtype)
    assert np.isfinite(g).all()
    ulp = 2.0 ** -7 if dtype == BF16 else 2.0 ** -10
    err = np.abs(g - r)
    assert (err <= ulp * np.abs(r) + 3e-5).all(), f"max err {err.max():.3e}"
    assert (got != ref).mean() < 0.01, f"{(got != ref).mean():.4f} of the outputs differ from the exactly accumulated product"


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,N", [(1, 4096, 4096), (16, 4096, 6144), (3, 14336, 4096), (7, 128, 16), (16, 2048, 512), (5, 4096, 28672),
                                   (2, 8192, 1024), (17, 4096, 4096), (32, 2048, 6144), (33, 4096, 1024), (48, 1024, 4112), (64, 4096, 4096),
                                   (50, 14336, 512)])
def test_linear_decode_llama_projection_shapes(gpu, dtype, B, K, N):
    """q/k/v (fused), o, gate/up, down projection shapes of Llama-3.x 1B / 8B / 70B at batch 1..16, with and without K splitting."""
    rng = np.random.default_rng(B * 7 + K + N)
    x = rand_half(rng, (B, K), dtype)
    w = rand_half(rng, (N, K), dtype, K ** -0.5)
    check(gpu_linear(gpu, x, w, dtype), LO.linear(x, w, dtype), dtype)


def test_linear_decode_strided_activations_and_padded_output(gpu):
    """x rows are slices of a wider buffer (token stride > K), y rows land in a wider buffer whose padding stays untouched."""
    rng = np.random.default_rng(3)
    B, K, N = 4, 1024, 256
    wide = rand_half(rng, (B, K + 256), BF16)
    w = rand_half(rng, (N, K), BF16, K ** -0.5)
    dx, dw = gpu.DeviceBuffer.from_numpy(wide), gpu.DeviceBuffer.from_numpy(w)
    ys = N + 64
    dy = gpu.DeviceBuffer(B * ys * 2)
    dy.fill_bytes(0xAB)
    assert gpu.lib.atoma_linear_decode(dx.ptr + 128 * 2, dw.ptr, dy.ptr, B, K, N, K + 256, K, ys, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    out = dy.numpy(np.uint16, (B, ys))
    check(out[:, :N], LO.linear(np.ascontiguousarray(wide[:, 128:128 + K]), w, BF16), BF16)
    assert (out[:, N:] == 0xABAB).all()


def test_linear_decode_exact_cases_and_linearity(gpu):
    """Integer-valued inputs: every partial sum is exact in fp32, so the result is bit-exact whatever the split; and
    W = identity reproduces x."""
    rng = np.random.default_rng(4)
    B, K, N = 16, 2048, 2048
    from oracle.halfs import from_f32
    x = from_f32(rng.integers(-4, 5, (B, K)).astype(np.float32), BF16)
    w = from_f32(rng.integers(-2, 3, (N, K)).astype(np.float32), BF16)
    assert np.array_equal(gpu_linear(gpu, x, w, BF16), LO.linear(x, w, BF16))
    eye = from_f32(np.eye(K, dtype=np.float32), BF16)
    xr = rand_half(rng, (B, K), BF16)
    assert np.array_equal(gpu_linear(gpu, xr, eye, BF16), xr)


def test_linear_decode_rejects_bad_arguments(gpu):
    d = gpu.DeviceBuffer(1 << 16)
    call = lambda B=1, K=128, N=16, xs=None, dt=BF16: gpu.lib.atoma_linear_decode(d.ptr, d.ptr, d.ptr, B, K, N, xs or K, K, N, dt, None)
    assert call(B=257) == -1 and "batch" in gpu.last_error()
    assert call(B=65) == 0, gpu.last_error()                                # 65..256 rows with a shape no tile kernel takes (N = 16): served in 64-row slices since round 6
    assert call(K=100) == -1 and "in_features" in gpu.last_error()
    assert call(N=24) == -1 and "out_features" in gpu.last_error()
    assert call(xs=64) == -1 and "strides" in gpu.last_error()
    assert call(dt=2) == -1 and "dtype" in gpu.last_error()
    assert call(B=0) == 0


def test_linear_decode_fused_epilogues_match_the_separate_ops(gpu):
    """_residual and _silu_mul keep the rounding points of the separate ops, so they must reproduce
    atoma_linear_decode followed by atoma_add / atoma_silu_mul bit for bit; and both agree with the oracle chain."""
    from oracle import elementwise_oracle as EO
    rng = np.random.default_rng(21)
    B, K, N, I = 7, 1024, 512, 768
    x = rand_half(rng, (B, K), BF16)
    w = rand_half(rng, (N, K), BF16, K ** -0.5)
    res = rand_half(rng, (B, N), BF16)
    wgu = rand_half(rng, (2 * I, K), BF16, K ** -0.5)
    dx, dw, dr, dwgu = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, res, wgu))
    y, y2, gu, act, act2 = (gpu.DeviceBuffer(n * 2) for n in (B * N, B * N, B * 2 * I, B * I, B * I))
    L = gpu.lib
    assert L.atoma_linear_decode(dx.ptr, dw.ptr, y.ptr, B, K, N, K, K, N, BF16, None) == 0
    assert L.atoma_add(y.ptr, dr.ptr, y.ptr, B * N, BF16, None) == 0
    assert L.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y2.ptr, B, K, N, K, K, N, N, BF16, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode(dx.ptr, dwgu.ptr, gu.ptr, B, K, 2 * I, K, K, 2 * I, BF16, None) == 0
    assert L.atoma_silu_mul(gu.ptr, gu.ptr + I * 2, act.ptr, B, I, 2 * I, 2 * I, I, BF16, None) == 0
    assert L.atoma_linear_decode_silu_mul(dx.ptr, dwgu.ptr, act2.ptr, B, K, I, K, K, I, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(y.numpy(np.uint16, (B, N)), y2.numpy(np.uint16, (B, N)))
    assert np.array_equal(act.numpy(np.uint16, (B, I)), act2.numpy(np.uint16, (B, I)))
    g = gu.numpy(np.uint16, (B, 2 * I))
    ref_act = EO.silu_mul(np.ascontiguousarray(g[:, :I]), np.ascontiguousarray(g[:, I:]), BF16)
    d = np.abs(act.numpy(np.uint16, (B, I)).astype(np.int32) - ref_act.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01


@pytest.mark.parametrize("B", [1, 2, 4, 8, 16])
def test_fused_gate_up_equals_projection_plus_silu_mul_at_model_shapes(gpu, B):
    """The stacked gate / up launch has half as many workgroups as the plain projection of the same matrix; its K split (wavefronts
    per workgroup) is derived from the weight rows, so the two must still agree bit for bit at the shapes where the split is not 1
    (Llama-3.1-8B MLP: [2 x 14336, 4096]) -- weight-streaming kernels at 1..4 and 5..16 rows."""
    rng = np.random.default_rng(100 + B)
    K, I = 4096, 14336
    x = rand_half(rng, (B, K), BF16)
    wgu = rand_half(rng, (2 * I, K), BF16, K ** -0.5)
    dx, dw = gpu.DeviceBuffer.from_numpy(x), gpu.DeviceBuffer.from_numpy(wgu)
    gu, act, act2 = gpu.DeviceBuffer(B * 2 * I * 2), gpu.DeviceBuffer(B * I * 2), gpu.DeviceBuffer(B * I * 2)
    L = gpu.lib
    assert L.atoma_linear_decode(dx.ptr, dw.ptr, gu.ptr, B, K, 2 * I, K, K, 2 * I, BF16, None) == 0, gpu.last_error()
    assert L.atoma_silu_mul(gu.ptr, gu.ptr + I * 2, act.ptr, B, I, 2 * I, 2 * I, I, BF16, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode_silu_mul(dx.ptr, dw.ptr, act2.ptr, B, K, I, K, K, I, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(act.numpy(np.uint16, (B, I)), act2.numpy(np.uint16, (B, I)))


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,N,I", [(1, 4096, 6144, 14336), (2, 4096, 512, 1024), (3, 8192, 1280, 3584), (4, 2048, 1024, 512), (1, 128, 16, 16),
                                     (4, 16384, 256, 128), (7, 4096, 512, 768), (40, 1024, 256, 256)])
def test_linear_decode_rmsnorm_is_the_two_ops_bit_for_bit(gpu, dtype, B, K, N, I):
    """The RMSNorm folded into the projection (f1: fused RMSNorm -> QKV / gate-up): at 1 row the projection kernel normalises
    its own input with atoma_rms_norm's arithmetic, above that the entry point runs the two kernels through xn_scratch -- either way
    the outputs must be the bits of atoma_rms_norm followed by atoma_linear_decode / atoma_linear_decode_silu_mul; x with a padded
    row stride, rows of very different magnitude (the scale is per row)."""
    rng = np.random.default_rng(B * 1000 + K + N)
    xs = K + 64
    x = rand_half(rng, (B, xs), dtype, 1.0)
    x16 = x.view(np.uint16) if x.dtype != np.uint16 else x
    if B > 1:   # one row 100x larger, one 100x smaller
        from oracle.halfs import from_f32
        f = to_f32(x, dtype)
        f[0] *= 100.0
        f[B - 1] *= 0.01
        x = from_f32(f, dtype)
    g = rand_half(rng, (K,), dtype, 1.0)
    w = rand_half(rng, (N, K), dtype, K ** -0.5)
    wgu = rand_half(rng, (2 * I, K), dtype, K ** -0.5)
    eps = 1e-5
    L = gpu.lib
    dx, dg, dw, dwgu = (gpu.DeviceBuffer.from_numpy(a) for a in (x, g, w, wgu))
    xn, scratch = gpu.DeviceBuffer(B * K * 2), gpu.DeviceBuffer(B * K * 2)
    y, y2, act, act2 = (gpu.DeviceBuffer(n * 2) for n in (B * N, B * N, B * I, B * I))
    assert L.atoma_rms_norm(dx.ptr, dg.ptr, xn.ptr, B, K, xs, K, eps, dtype, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode(xn.ptr, dw.ptr, y.ptr, B, K, N, K, K, N, dtype, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode_silu_mul(xn.ptr, dwgu.ptr, act.ptr, B, K, I, K, K, I, dtype, None) == 0, gpu.last_error()
    sc = scratch.ptr if B > 1 else None
    assert L.atoma_linear_decode_rmsnorm(dx.ptr, dg.ptr, eps, dw.ptr, y2.ptr, sc, B, K, N, xs, K, N, dtype, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode_rmsnorm_silu_mul(dx.ptr, dg.ptr, eps, dwgu.ptr, act2.ptr, sc, B, K, I, xs, K, I, dtype, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(y.numpy(np.uint16, (B, N)), y2.numpy(np.uint16, (B, N)))
    assert np.array_equal(act.numpy(np.uint16, (B, I)), act2.numpy(np.uint16, (B, I)))
    assert np.isfinite(to_f32(y2.numpy(np.uint16, (B, N)), dtype)).all()
    if B > 1:   # the larger batches need the scratch rows, and say so
        assert L.atoma_linear_decode_rmsnorm(dx.ptr, dg.ptr, eps, dw.ptr, y2.ptr, None, B, K, N, xs, K, N, dtype, None) == -1
        assert "xn_scratch" in gpu.last_error()
    assert L.atoma_linear_decode_rmsnorm(dx.ptr, None, eps, dw.ptr, y2.ptr, None, B, K, N, xs, K, N, dtype, None) == -1


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,N", [(65, 1024, 256), (100, 4096, 1024), (128, 4096, 4096), (129, 2048, 6144), (200, 14336, 512), (256, 4096, 4096), (256, 1024, 28672),
                                   (256, 8192, 1280),
                                   (130, 1024, 1008), (256, 512, 48)])      # out_features no tile kernel takes: slices of 64 rows through the small-batch kernels
def test_linear_decode_65_to_256_rows(gpu, dtype, B, K, N):
    """linear_big_kernel (32x32x16 MFMA tile of 128 weight rows x 128 / 256 batch rows, with and without K splitting)."""
    rng = np.random.default_rng(B + K + N)
    x = rand_half(rng, (B, K), dtype)
    w = rand_half(rng, (N, K), dtype, K ** -0.5)
    check(gpu_linear(gpu, x, w, dtype), LO.linear(x, w, dtype), dtype)


@pytest.mark.parametrize("opts", [dict(linear_wide_nw=64, linear_wide_splits=1), dict(linear_wide_nw=128, linear_wide_splits=1), dict(linear_wide_nw=64, linear_wide_splits=2),
                                  dict(linear_wide_nw=128, linear_wide_splits=3), dict(linear_wide_nw=64, linear_wide_splits=4), dict(linear_wide_nw=128, linear_wide_splits=8),
                                  dict(linear_wide_nw=64, linear_wide_splits=4, linear_wide_xcd=0), dict(linear_wide_nw=128, linear_wide_splits=5), dict(linear_wide=0)])
@pytest.mark.parametrize("B", [65, 128, 129, 160, 192, 200, 256])
def test_linear_wide_every_tile_shape_and_split(gpu, opts, B):
    """linear_wide_kernel (round 6: 65..256 rows): every template variant (64 / 128 weight rows x 128 / 256 batch rows), K unsplit and
    split 2..8 ways with the in-launch merge, the XCD map of the splits on and off, uneven splits -- against the oracle bound on random
    inputs, BIT-exact on integer-valued inputs (every partial sum exact in fp32: any tiling and any merge order must agree), and
    linear_big_kernel (the kernel it replaces) through the same checks."""
    from oracle.halfs import from_f32
    rng = np.random.default_rng(B + sum(opts.values()))
    K, N = 2048, 1152                                   # 1152 = 9 x 128 = 18 x 64: a tile count no XCD group divides
    L = gpu.lib
    names = ("linear_wide", "linear_wide_nw", "linear_wide_splits", "linear_wide_xcd")
    defaults = dict(linear_wide=1, linear_wide_nw=0, linear_wide_splits=0, linear_wide_xcd=1)
    try:
        for n in names:
            assert L.atoma_set_option(n.encode(), opts.get(n, defaults[n])) == 0
        for dtype in (BF16, F16):
            x = rand_half(rng, (B, K), dtype)
            w = rand_half(rng, (N, K), dtype, K ** -0.5)
            check(gpu_linear(gpu, x, w, dtype), LO.linear(x, w, dtype), dtype)
        xi = from_f32(rng.integers(-4, 5, (B, K)).astype(np.float32), BF16)
        wi = from_f32(rng.integers(-2, 3, (N, K)).astype(np.float32), BF16)
        assert np.array_equal(gpu_linear(gpu, xi, wi, BF16), LO.linear(xi, wi, BF16))
    finally:
        for n in names:
            L.atoma_set_option(n.encode(), defaults[n])


def test_linear_decode_big_exact_cases(gpu):
    """Integer-valued inputs (every partial sum exact in fp32): bit-exact whatever the tiling / split; W = I reproduces x; a
    strided x and a padded y."""
    rng = np.random.default_rng(41)
    from oracle.halfs import from_f32
    B, K, N = 200, 2048, 1024
    x = from_f32(rng.integers(-4, 5, (B, K)).astype(np.float32), BF16)
    w = from_f32(rng.integers(-2, 3, (N, K)).astype(np.float32), BF16)
    assert np.array_equal(gpu_linear(gpu, x, w, BF16), LO.linear(x, w, BF16))
    eye = from_f32(np.eye(1024, dtype=np.float32), BF16)
    xr = rand_half(rng, (256, 1024), BF16)
    assert np.array_equal(gpu_linear(gpu, xr, eye, BF16), xr)
    wide = from_f32(rng.integers(-4, 5, (B, K + 64)).astype(np.float32), BF16)
    dx, dw = gpu.DeviceBuffer.from_numpy(wide), gpu.DeviceBuffer.from_numpy(w)
    ys = N + 64
    dy = gpu.DeviceBuffer(B * ys * 2)
    dy.fill_bytes(0xAB)
    assert gpu.lib.atoma_linear_decode(dx.ptr + 32 * 2, dw.ptr, dy.ptr, B, K, N, K + 64, K, ys, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    out = dy.numpy(np.uint16, (B, ys))
    assert np.array_equal(out[:, :N], LO.linear(np.ascontiguousarray(wide[:, 32:32 + K]), w, BF16)) and (out[:, N:] == 0xABAB).all()


@pytest.mark.parametrize("B,K,N,I", [(17, 1024, 512, 768), (40, 4096, 1024, 1024), (64, 1024, 256, 128), (33, 8192, 512, 3584), (64, 2048, 4096, 2048),
                                     # linear_ks_kernel (one launch, K interleaved over the wavefronts): the shapes of one rank of the 70B TP = 8 job
                                     # (K split over workgroups for the 1280-row q/k/v shard only), 1 / 9 / 8 chunks of 256 inputs (short path, tail
                                     # path, exactly one unrolled iteration), 16- and 32-row workgroups, row counts that are not a multiple of 32
                                     (64, 8192, 1280, 3584), (64, 1024, 8192, 256), (48, 3584, 8192, 512), (17, 256, 32, 16), (31, 2304, 48, 80),
                                     (64, 2048, 6144, 1024), (50, 4352, 4112, 48), (20, 1280, 128, 64),   # K = 1280: not a multiple of 256 -> linear_mid_kernel
                                     (65, 1024, 256, 128), (130, 4096, 1024, 1024), (256, 2048, 4096, 14336), (256, 14336, 512, 256),
                                     # 193..256 rows, unsplit gate/up with 56 | intermediate: linear_wide_gu112_kernel (112-row interleaved tiles, the halves of a wavefront swap accumulators)
                                     (256, 4096, 128, 14336), (200, 512, 128, 28672), (193, 1024, 64, 14336)])
def test_linear_mid_batch_epilogues_match_the_separate_ops(gpu, B, K, N, I):
    """17..64 rows (linear_mid_kernel: one workgroup per 64 features) and 65..256 rows (linear_big_kernel), with and without K
    splitting: the fused epilogues keep the rounding points of projection + atoma_add / atoma_silu_mul -- bit for bit -- and the
    plain projection meets the oracle bound."""
    rng = np.random.default_rng(B + K + N)
    x = rand_half(rng, (B, K), BF16)
    w = rand_half(rng, (N, K), BF16, K ** -0.5)
    res = rand_half(rng, (B, N), BF16)
    wgu = rand_half(rng, (2 * I, K), BF16, K ** -0.5)
    dx, dw, dr, dwgu = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, res, wgu))
    y, y2, gu, act, act2 = (gpu.DeviceBuffer(n * 2) for n in (B * N, B * N, B * 2 * I, B * I, B * I))
    L = gpu.lib
    assert L.atoma_linear_decode(dx.ptr, dw.ptr, y.ptr, B, K, N, K, K, N, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    check(y.numpy(np.uint16, (B, N)), LO.linear(x, w, BF16), BF16)
    assert L.atoma_add(y.ptr, dr.ptr, y.ptr, B * N, BF16, None) == 0
    assert L.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y2.ptr, B, K, N, K, K, N, N, BF16, None) == 0, gpu.last_error()
    assert L.atoma_linear_decode(dx.ptr, dwgu.ptr, gu.ptr, B, K, 2 * I, K, K, 2 * I, BF16, None) == 0
    assert L.atoma_silu_mul(gu.ptr, gu.ptr + I * 2, act.ptr, B, I, 2 * I, 2 * I, I, BF16, None) == 0
    assert L.atoma_linear_decode_silu_mul(dx.ptr, dwgu.ptr, act2.ptr, B, K, I, K, K, I, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(y.numpy(np.uint16, (B, N)), y2.numpy(np.uint16, (B, N)))
    assert np.array_equal(act.numpy(np.uint16, (B, I)), act2.numpy(np.uint16, (B, I)))


def gpu_linear_any(gpu, x, w, dtype, x_stride=None, y_stride=None):
    B, K, N = x.shape[0], w.shape[1], w.shape[0]
    dx, dw = gpu.DeviceBuffer.from_numpy(x), gpu.DeviceBuffer.from_numpy(w)
    ys = y_stride or N
    dy = gpu.DeviceBuffer(max(B, 1) * ys * 2)
    dy.fill_bytes(0xAB)
    rc = gpu.lib.atoma_linear(dx.ptr, dw.ptr, dy.ptr, B, K, N, x_stride or x.shape[1], K, ys, dtype, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return dy.numpy(np.uint16, (max(B, 1), ys))


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,N", [(4, 1024, 512), (5, 4096, 1024), (16, 4096, 6144), (33, 4096, 1024), (64, 4096, 4096), (128, 2048, 6144), (256, 4096, 1024), (300, 1024, 4112),
                                   (2048, 512, 256)])
def test_linear_any_batch_gemm_route(gpu, dtype, B, K, N):
    """atoma_linear: the weight-streaming kernel up to 4 rows, the vendor GEMM (fp32 accumulation, one rounding) above:
    same bound against the exactly accumulated product -- one unit in the last place."""
    rng = np.random.default_rng(B + K + N)
    x = rand_half(rng, (B, K), dtype)
    w = rand_half(rng, (N, K), dtype, K ** -0.5)
    check(gpu_linear_any(gpu, x, w, dtype), LO.linear(x, w, dtype), dtype)


def test_linear_any_batch_strides_exactness_and_errors(gpu):
    """GEMM route: strided x rows, padded y rows left untouched; integer-valued inputs are bit-exact; W = I reproduces x."""
    rng = np.random.default_rng(12)
    from oracle.halfs import from_f32
    B, K, N = 96, 1024, 256
    wide = from_f32(rng.integers(-4, 5, (B, K + 64)).astype(np.float32), BF16)
    w = from_f32(rng.integers(-2, 3, (N, K)).astype(np.float32), BF16)
    dx, dw = gpu.DeviceBuffer.from_numpy(wide), gpu.DeviceBuffer.from_numpy(w)
    ys = N + 64
    dy = gpu.DeviceBuffer(B * ys * 2)
    dy.fill_bytes(0xAB)
    assert gpu.lib.atoma_linear(dx.ptr + 32 * 2, dw.ptr, dy.ptr, B, K, N, K + 64, K, ys, BF16, None) == 0, gpu.last_error()
    gpu.synchronize()
    out = dy.numpy(np.uint16, (B, ys))
    assert np.array_equal(out[:, :N], LO.linear(np.ascontiguousarray(wide[:, 32:32 + K]), w, BF16))
    assert (out[:, N:] == 0xABAB).all()
    eye = from_f32(np.eye(512, dtype=np.float32), F16)
    x = rand_half(rng, (100, 512), F16)
    assert np.array_equal(gpu_linear_any(gpu, x, eye, F16), x)
    assert gpu.lib.atoma_linear(dx.ptr, dw.ptr, dy.ptr, 100, 1028, N, 1028, 1028, ys, BF16, None) == -1
    assert "multiple of 8" in gpu.last_error()
    assert gpu.lib.atoma_linear(dx.ptr + 2, dw.ptr, dy.ptr, 100, K, N, K + 64, K, ys, BF16, None) == -1
    assert "aligned" in gpu.last_error()


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,h,hk,d", [(64, 8192, 8, 1, 128), (33, 2048, 4, 2, 64), (64, 4096, 32, 8, 128), (17, 1024, 2, 1, 128), (8, 1024, 2, 1, 128),
                                        (40, 8192, 2, 1, 128), (48, 2048, 16, 16, 128), (64, 1024, 8, 8, 64),
                                        # 65..256 rows (round 6: linear_wide_kernel with the same epilogue): the 8B layer, a rank's shard, d = 64, ragged batches
                                        (256, 4096, 32, 8, 128), (65, 8192, 8, 1, 128), (200, 2048, 4, 2, 64), (129, 1024, 2, 1, 128), (256, 1024, 8, 8, 64)])
def test_qkv_projection_rope_cache_entry_is_the_two_ops_bit_for_bit(gpu, dtype, B, K, h, hk, d):
    """atoma_linear_decode_qkv_rope_cache = atoma_linear_decode followed by atoma_rope_qk_cache on the same buffers, bit for bit: the
    shard of a tensor-parallel rank (1280 rows: 32-row tiles, K split 4 ways and merged inside the launch, RoPE + cache write as the
    epilogue -- the rotation's partner sits in the neighbouring wavefront), 64- and 128-row tiles (both partners in one lane), d = 64,
    a matrix with so few rows that K is split 8 ways (the RoPE / cache kernel merges the fp32 partials), MHA, and a batch outside
    17..64 (the entry runs the two ops)."""
    rng = np.random.default_rng(B + K + h)
    width, page, nb = (h + 2 * hk) * d, 16, max(12, B // 16 + 3)
    x = rand_half(rng, (B, K), dtype)
    w = rand_half(rng, (width, K), dtype, K ** -0.5)
    cos = rand_half(rng, (4096, d // 2), dtype)
    sin = rand_half(rng, (4096, d // 2), dtype)
    pos = rng.integers(0, 4096, B).astype(np.int64)
    slots = rng.permutation(nb * page)[:B].astype(np.int64)
    slots[B // 2] = -1                                      # a padding token: rotated, not cached
    dx, dw, dc, ds, dp, dsl = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, cos, sin, pos, slots))
    outs = []
    for fused in (False, True):
        qkv = gpu.DeviceBuffer.zeros((B, width), np.uint16)
        kc, vc = gpu.DeviceBuffer.zeros((nb * page * hk * d,), np.uint16), gpu.DeviceBuffer.zeros((nb * page * hk * d,), np.uint16)
        L = gpu.lib
        if fused:
            rc = L.atoma_linear_decode_qkv_rope_cache(dx.ptr, dw.ptr, qkv.ptr, kc.ptr, vc.ptr, dsl.ptr, dc.ptr, ds.ptr, dp.ptr, B, K, h, hk, d, K, K, width,
                                                      page * hk * d, page, dtype, 1, None)
            assert rc == 0, gpu.last_error()
        else:
            assert L.atoma_linear_decode(dx.ptr, dw.ptr, qkv.ptr, B, K, width, K, K, width, dtype, None) == 0, gpu.last_error()
            assert L.atoma_rope_qk_cache(qkv.ptr, qkv.ptr + h * d * 2, qkv.ptr + (h + hk) * d * 2, kc.ptr, vc.ptr, dsl.ptr, dc.ptr, ds.ptr, dp.ptr, B, h, hk, d,
                                         width, width, width, page * hk * d, page, dtype, 1, None) == 0, gpu.last_error()
        gpu.synchronize()
        outs.append((qkv.numpy(np.uint16, (B, width)), kc.numpy(np.uint16, (nb * page * hk * d,)), vc.numpy(np.uint16, (nb * page * hk * d,))))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][1].any() and outs[0][2].any()


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("B,K,N", [(37, 2048, 4096), (64, 1024, 8192), (19, 8192, 1280), (48, 512, 256)])
def test_tile_kernel_strides_padding_exactness(gpu, dtype, B, K, N):
    """linear_tile_kernel (17..64 rows) with everything a caller may hand it: x rows that are slices of a wider buffer, weight rows with
    padding between them, y (and the residual) in wider buffers whose padding must stay untouched, batches that are not a multiple
    of 16, both dtypes; K split merged in the launch (4096 x 2048: 2 splits), not split (8192 x 1024), split 6 ways, unevenly (1280 x 8192) and
    a matrix of four 64-row tiles (fp32 partials + merge kernel).  Integer-valued inputs: every partial sum is exact in fp32, so the
    result must be bit-exact whatever the split and the order of arrival -- ten runs in a row must agree to the bit."""
    rng = np.random.default_rng(B + K + N + dtype)
    from oracle.halfs import from_f32
    xs, ws, ys, rs = K + 64, K + 128, N + 32, N + 8
    xw = from_f32(rng.integers(-3, 4, (B, xs)).astype(np.float32), dtype)
    ww = from_f32(rng.integers(-2, 3, (N, ws)).astype(np.float32), dtype)
    res = from_f32(rng.integers(-8, 9, (B, rs)).astype(np.float32), dtype)
    dx, dw, dr = (gpu.DeviceBuffer.from_numpy(a) for a in (xw, ww, res))
    x, w = np.ascontiguousarray(xw[:, 32:32 + K]), np.ascontiguousarray(ww[:, 64:64 + K])
    from oracle import elementwise_oracle as EO
    exact = to_f32(x, dtype).astype(np.float64) @ to_f32(w, dtype).astype(np.float64).T          # small integers: exact in fp32 in any order
    assert np.abs(exact).max() < 2 ** 20
    want = from_f32(exact.astype(np.float32), dtype)          # the projection's one rounding
    want_res = EO.add(want, np.ascontiguousarray(res[:, :N]), dtype)                              # then the residual add rounds again
    first = None
    for rep in range(10):
        dy = gpu.DeviceBuffer(B * ys * 2)
        dy.fill_bytes(0xAB)
        assert gpu.lib.atoma_linear_decode_residual(dx.ptr + 32 * 2, dw.ptr + 64 * 2, dr.ptr, dy.ptr, B, K, N, xs, ws, rs, ys, dtype, None) == 0, gpu.last_error()
        gpu.synchronize()
        out = dy.numpy(np.uint16, (B, ys))
        assert (out[:, N:] == 0xABAB).all()
        assert np.array_equal(out[:, :N], want_res), rep
        first = out if first is None else first
        assert np.array_equal(out, first)
    dy = gpu.DeviceBuffer(B * ys * 2)
    assert gpu.lib.atoma_linear_decode(dx.ptr + 32 * 2, dw.ptr + 64 * 2, dy.ptr, B, K, N, xs, ws, ys, dtype, None) == 0, gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(dy.numpy(np.uint16, (B, ys))[:, :N], want)
