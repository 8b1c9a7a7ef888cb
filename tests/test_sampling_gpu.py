"""On-device greedy selection (atoma_argmax_rows) against numpy argmax: index work, bit-exact."""
import numpy as np
import pytest

from oracle.halfs import F16, BF16, to_f32
from util import rand_half

pytestmark = pytest.mark.gpu
F32 = 2


def gpu_argmax(gpu, logits, dtype, vocab=None, stride=None):
    rows = logits.shape[0]
    vocab = vocab or logits.shape[1]
    stride = stride or logits.shape[1]
    dl = gpu.DeviceBuffer.from_numpy(logits)
    di, dv = gpu.DeviceBuffer.zeros((rows,), np.int32), gpu.DeviceBuffer.zeros((rows,), np.float32)
    rc = gpu.lib.atoma_argmax_rows(dl.ptr, rows, vocab, stride, dtype, di.ptr, dv.ptr, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return di.numpy(np.int32, (rows,)), dv.numpy(np.float32, (rows,))


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("rows,vocab", [(256, 128256), (1, 128256), (7, 32000), (3, 50257), (5, 97), (2, 4096), (300, 1000)])
def test_argmax_rows_matches_numpy(gpu, dtype, rows, vocab):
    rng = np.random.default_rng(rows + vocab)
    if dtype == F32:
        logits = rng.standard_normal((rows, vocab)).astype(np.float32)
        ref32 = logits
    else:
        logits = rand_half(rng, (rows, vocab), dtype, 3.0)
        ref32 = to_f32(logits, dtype)
    idx, val = gpu_argmax(gpu, logits, dtype)
    assert np.array_equal(idx, ref32.argmax(1).astype(np.int32))      # 16-bit logits tie often: first index wins
    assert np.array_equal(val, ref32[np.arange(rows), idx])


def test_argmax_rows_ties_strides_and_degenerate_rows(gpu):
    rng = np.random.default_rng(0)
    rows, vocab, stride = 6, 5000, 5120
    x = rng.standard_normal((rows, stride)).astype(np.float32)
    x[:, vocab:] = 100.0                               # padding columns beyond vocab must be ignored
    x[0, [4999, 17, 3000]] = 50.0                      # three-way tie: the smallest index
    x[1, :vocab] = -np.inf                             # all -inf: index 0 (numpy)
    x[2, :vocab] = 7.0                                 # constant row
    x[3, 123] = np.nan                                 # a NaN is never selected
    x[3, 4000] = 60.0
    x[4, vocab - 1] = 90.0                             # last element (tail after the vector loop)
    idx, val = gpu_argmax(gpu, x, F32, vocab=vocab, stride=stride)
    assert idx.tolist()[:3] == [17, 0, 0] and idx[3] == 4000 and idx[4] == vocab - 1
    assert idx[5] == x[5, :vocab].argmax()
    assert val[0] == 50.0 and np.isneginf(val[1]) and val[2] == 7.0 and val[3] == 60.0
    # unaligned rows (odd stride of 16-bit elements): scalar path, same answer
    y = rand_half(rng, (4, 1001), BF16)
    idx, _ = gpu_argmax(gpu, y, BF16)
    assert np.array_equal(idx, to_f32(y, BF16).argmax(1))


def test_argmax_rows_rejects_bad_arguments(gpu):
    d = gpu.DeviceBuffer(1024)
    assert gpu.lib.atoma_argmax_rows(d.ptr, 1, 16, 16, 5, d.ptr, None, None) == -1 and "dtype" in gpu.last_error()
    assert gpu.lib.atoma_argmax_rows(d.ptr, 1, 16, 8, F32, d.ptr, None, None) == -1 and "row_stride" in gpu.last_error()
    assert gpu.lib.atoma_argmax_rows(d.ptr, 1, 16, 16, F32, None, None, None) == -1 and "out_idx" in gpu.last_error()
    assert gpu.lib.atoma_argmax_rows(d.ptr, 0, 16, 16, F32, d.ptr, None, None) == 0


def gpu_topk(gpu, logits, dtype, k, vocab=None, stride=None):
    rows = logits.shape[0]
    vocab = vocab or logits.shape[1]
    stride = stride or logits.shape[1]
    dl = gpu.DeviceBuffer.from_numpy(logits)
    dv, di = gpu.DeviceBuffer.zeros((rows, k), np.float32), gpu.DeviceBuffer.zeros((rows, k), np.int32)
    rc = gpu.lib.atoma_topk_rows(dl.ptr, rows, vocab, stride, dtype, k, dv.ptr, di.ptr, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return dv.numpy(np.float32, (rows, k)), di.numpy(np.int32, (rows, k))


def np_topk(x32, k):
    """(value descending, index ascending): numpy lexsort with the index as the secondary key."""
    idx = np.stack([np.lexsort((np.arange(r.size), -r))[:k] for r in x32]).astype(np.int32)
    return np.take_along_axis(x32, idx, 1), idx


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("rows,vocab,k", [(256, 128256, 50), (3, 128256, 1), (2, 128256, 1024), (7, 32000, 40), (5, 97, 97), (4, 50257, 64),
                                          (4, 131072, 256), (3, 4096, 100), (2, 128256, 256), (2, 131080, 50)])
def test_topk_rows_matches_numpy(gpu, dtype, rows, vocab, k):
    rng = np.random.default_rng(rows + vocab + k)
    if dtype == F32:
        logits = (rng.standard_normal((rows, vocab)) * 4).astype(np.float32)
        ref32 = logits
    else:
        logits = rand_half(rng, (rows, vocab), dtype, 4.0)      # 16-bit logits: plenty of exact ties around the cut
        ref32 = to_f32(logits, dtype)
    val, idx = gpu_topk(gpu, logits, dtype, k)
    rv, ri = np_topk(ref32, k)
    assert np.array_equal(idx, ri)
    assert np.array_equal(val, rv)


def test_topk_rows_degenerate_rows_and_strides(gpu):
    """Constant rows, all -inf, massive ties at the cut (the candidate list overflows: k-round fallback), padded rows."""
    rng = np.random.default_rng(1)
    rows, vocab, stride, k = 6, 20000, 20480, 33
    x = rng.standard_normal((rows, stride)).astype(np.float32)
    x[:, vocab:] = 1e9                                   # padding columns must be ignored
    x[0, :vocab] = 2.5                                   # constant row: indices 0..k-1
    x[1, :vocab] = -np.inf
    x[2, :vocab] = np.where(rng.random(vocab) < 0.5, 1.0, 0.0)    # two values, 10k ties each
    x[3, 5000:15000] = 7.0                               # a plateau of 10k equal maxima
    x[4, 100] = -0.0
    x[4, 50] = 0.0                                       # signed zeros are equal: index order decides among them
    val, idx = gpu_topk(gpu, x, F32, k, vocab=vocab, stride=stride)
    rv, ri = np_topk(x[:, :vocab], k)
    assert np.array_equal(idx, ri)
    assert np.array_equal(val, rv)
    assert idx[0].tolist() == list(range(k)) and idx[3].tolist() == list(range(5000, 5000 + k))


def test_topk_rows_k1_equals_argmax_and_rejects_bad_arguments(gpu):
    rng = np.random.default_rng(2)
    x = rand_half(rng, (9, 4096), BF16)
    _, idx = gpu_topk(gpu, x, BF16, 1)
    a, _ = gpu_argmax(gpu, x, BF16)
    assert np.array_equal(idx[:, 0], a)
    d = gpu.DeviceBuffer(4096)
    assert gpu.lib.atoma_topk_rows(d.ptr, 1, 16, 16, F32, 0, d.ptr, d.ptr, None) == -1 and "k must" in gpu.last_error()
    assert gpu.lib.atoma_topk_rows(d.ptr, 1, 16, 16, F32, 17, d.ptr, d.ptr, None) == -1
    assert gpu.lib.atoma_topk_rows(d.ptr, 1, 16, 16, 9, 4, d.ptr, d.ptr, None) == -1 and "dtype" in gpu.last_error()
    assert gpu.lib.atoma_topk_rows(d.ptr, 1, 16, 16, F32, 4, None, d.ptr, None) == -1


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
def test_topk_and_argmax_with_nans(gpu, dtype):
    """NaNs (both signs, quiet and signalling patterns) are never selected by argmax and sort LAST in top-k -- decided from
    the bit pattern, because the library is built with -fno-honor-nans."""
    rng = np.random.default_rng(5)
    rows, vocab, k = 5, 9000, 40
    if dtype == F32:
        x = (rng.standard_normal((rows, vocab)) * 3).astype(np.float32)
        bits = x.view(np.uint32)
        nan_patterns = [0x7fc00000, 0xffc00000, 0x7f800001, 0xff812345]
    else:
        x = rand_half(rng, (rows, vocab), dtype, 3.0)
        bits = x
        nan_patterns = [0x7fc0, 0xffc0, 0x7f81, 0xff85] if dtype == BF16 else [0x7e00, 0xfe00, 0x7c01, 0xfd55]
    for r in range(rows - 1):
        pos = rng.choice(vocab, 300, replace=False)
        bits[r, pos] = np.array(nan_patterns, bits.dtype)[np.arange(300) % 4]
    bits[0, 0] = nan_patterns[0]                        # a NaN in front of the row
    bits[rows - 1, :] = nan_patterns[1]                 # an all-NaN row ...
    bits[rows - 1, 17] = 0                              # ... but for one finite value
    x32 = x if dtype == F32 else to_f32(x, dtype)
    clean = np.where(np.isnan(x32), -np.inf, x32)       # NaN ranks below -inf; no -inf in this data
    idx, val = gpu_argmax(gpu, x, dtype)
    assert np.array_equal(idx, clean.argmax(1)) and idx[rows - 1] == 17
    assert np.array_equal(val, clean[np.arange(rows), idx])
    tv, ti = gpu_topk(gpu, x, dtype, k)
    rv, ri = np_topk(clean[: rows - 1], k)
    assert np.array_equal(ti[: rows - 1], ri) and np.array_equal(tv[: rows - 1], rv)
    assert ti[rows - 1, 0] == 17 and np.isnan(tv[rows - 1, 1:]).all()      # after the one finite value: NaNs, index order
    assert ti[rows - 1, 1:].tolist() == [i for i in range(k + 1) if i != 17][: k - 1]


# ---- stochastic selection (atoma_sample_rows) ----
from oracle import sampling_oracle as SO


def gpu_sample(gpu, logits, dtype, u, temperature, top_k=0, top_p=1.0, stride=None, vocab=None):
    rows = logits.shape[0]
    vocab = vocab or logits.shape[1]
    stride = stride or logits.shape[1]
    dl, du = gpu.DeviceBuffer.from_numpy(logits), gpu.DeviceBuffer.from_numpy(np.asarray(u, np.float32))
    di, dv = gpu.DeviceBuffer.zeros((rows,), np.int32), gpu.DeviceBuffer.zeros((rows,), np.float32)
    rc = gpu.lib.atoma_sample_rows(dl.ptr, rows, vocab, stride, dtype, temperature, top_k, top_p, du.ptr, di.ptr, dv.ptr, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    return di.numpy(np.int32, (rows,)), dv.numpy(np.float32, (rows,))


def check_draws(x32, idx, u, temperature, top_k=0, top_p=1.0, slack=2e-5):
    """Every answer must be the oracle's token for this u, up to f32 rounding of the running sums: u lies inside the token's
    interval widened by `slack` (relative to a total of 1)."""
    exact = 0
    for r in range(x32.shape[0]):
        lo, hi = SO.bracket(x32[r], int(idx[r]), temperature, top_k, top_p)
        assert lo - slack <= u[r] < hi + slack, (r, int(idx[r]), float(u[r]), lo, hi)
        exact += int(idx[r]) == SO.sample(x32[r], float(u[r]), temperature, top_k, top_p)
    assert exact >= x32.shape[0] - max(1, x32.shape[0] // 25)      # a draw within rounding distance of an interval edge may fall on either side


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("rows,vocab,temperature,top_k,top_p", [(64, 128256, 1.0, 0, 1.0), (32, 32000, 0.7, 0, 1.0), (48, 128256, 0.8, 50, 1.0),
                                                                (48, 128256, 1.0, 0, 0.9), (40, 50257, 1.3, 40, 0.95), (16, 97, 1.0, 0, 1.0), (8, 513, 0.5, 7, 0.5)])
def test_sample_rows_matches_oracle(gpu, dtype, rows, vocab, temperature, top_k, top_p):
    rng = np.random.default_rng(rows + vocab + top_k)
    if dtype == F32:
        logits = (rng.standard_normal((rows, vocab)) * 2.5).astype(np.float32)
        x32 = logits
    else:
        logits = rand_half(rng, (rows, vocab), dtype, 2.5)
        x32 = to_f32(logits, dtype)
    u = rng.random(rows).astype(np.float32)
    u[0], u[1 % rows] = 0.0, np.float32(1.0 - 2.0 ** -24)           # the two ends of [0, 1)
    idx, val = gpu_sample(gpu, logits, dtype, u, temperature, top_k, top_p)
    assert (idx >= 0).all() and (idx < vocab).all()
    assert np.array_equal(val, x32[np.arange(rows), idx])
    check_draws(x32, idx, u, temperature, top_k, top_p)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("temperature,top_p,scale", [(1.0, 0.9, 1.0), (2.0, 0.95, 2.5), (1.0, 0.5, 0.25)])
def test_top_p_nucleus_larger_than_the_sorted_list(gpu, dtype, temperature, top_p, scale):
    """Flat rows over a 128k vocabulary: the 1024 most probable tokens hold a few per cent of the mass, so the nucleus of
    Candle's sample_topp (every token, most probable first, until the cumulative probability reaches top_p) has tens of
    thousands of members -- the full-row path of atoma_sample_rows (the distribution must not be truncated to the
    sorted top-1024 list).  bf16 rows carry thousands of equal logits: the cut and the draw fall inside tie classes."""
    rows, vocab = 24, 128256
    rng = np.random.default_rng(int(top_p * 100) + dtype)
    if dtype == F32:
        logits = (rng.standard_normal((rows, vocab)) * scale).astype(np.float32)
        x32 = logits
    else:
        logits = rand_half(rng, (rows, vocab), dtype, scale)
        x32 = to_f32(logits, dtype)
    logits[3, :] = logits[3, 0]                          # a constant row: one tie class of 128 256 tokens
    x32 = x32.copy()
    x32[3, :] = x32[3, 0]
    u = rng.random(rows).astype(np.float32)
    u[0], u[1] = 0.0, np.float32(1.0 - 2.0 ** -24)
    idx, val = gpu_sample(gpu, logits, dtype, u, temperature, 0, top_p)
    assert np.array_equal(val, x32[np.arange(rows), idx])
    sizes = [len(SO.kept_weights(x32[r], temperature, 0, top_p)[0]) for r in range(rows)]
    assert min(sizes) > 1024, "the test must exercise the full-row path"
    check_draws(x32, idx, u, temperature, 0, top_p, slack=1e-4)
    assert idx[3] == min(int(u[3] * sizes[3]), sizes[3] - 1) or abs(idx[3] - u[3] * sizes[3]) < 16   # constant row: u picks the position


def test_sample_rows_distribution_and_degenerate_rows(gpu):
    """Equally spaced uniforms reproduce the distribution; -inf / NaN logits are never drawn; a one-hot row always returns its token;
    padded rows ignore the padding."""
    rng = np.random.default_rng(77)
    vocab, stride, n = 1000, 1024, 4096
    row = (rng.standard_normal(vocab) * 2).astype(np.float32)
    row[[3, 500]] = -np.inf
    row[7] = np.nan
    x = np.full((n, stride), 50.0, np.float32)                      # padding columns hold a huge logit
    x[:, :vocab] = row
    u = ((np.arange(n) + 0.5) / n).astype(np.float32)
    idx, _ = gpu_sample(gpu, x, F32, u, 1.0, vocab=vocab, stride=stride)
    assert (idx < vocab).all() and not np.isin(idx, [3, 500, 7]).any()
    order, w = SO.kept_weights(row, 1.0)
    p = w / w.sum()
    freq = np.bincount(idx, minlength=vocab) / n
    assert np.abs(freq - p).max() < 2.0 / n + 1e-6                  # stratified uniforms: every frequency within 2 strata of its probability
    onehot = np.full((5, vocab), -np.inf, np.float32)
    onehot[np.arange(5), [0, 17, 511, 512, 999]] = 1.0
    idx, _ = gpu_sample(gpu, onehot, F32, rng.random(5).astype(np.float32), 0.7)
    assert idx.tolist() == [0, 17, 511, 512, 999]
    idx, _ = gpu_sample(gpu, onehot, F32, rng.random(5).astype(np.float32), 0.7, top_k=5, top_p=0.9)
    assert idx.tolist() == [0, 17, 511, 512, 999]
    d = gpu.DeviceBuffer(4096)
    L = gpu.lib
    assert L.atoma_sample_rows(d.ptr, 1, 16, 16, F32, 0.0, 0, 1.0, d.ptr, d.ptr, None, None) == -1 and "temperature" in gpu.last_error()
    assert L.atoma_sample_rows(d.ptr, 1, 16, 16, F32, 1.0, 2000, 1.0, d.ptr, d.ptr, None, None) == -1 and "top_k" in gpu.last_error()
    assert L.atoma_sample_rows(d.ptr, 1, 16, 16, F32, 1.0, 0, 0.0, d.ptr, d.ptr, None, None) == -1 and "top_p" in gpu.last_error()
    assert L.atoma_sample_rows(d.ptr, 1, 16, 16, F32, 1.0, 0, 1.0, None, d.ptr, None, None) == -1
