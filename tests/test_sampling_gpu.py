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
