"""C1 plumbing on the GPU: the same continuous-batching trace as test_plumbing.py, every
hot-path op through the C ABI of libatoma_hip.so with the paged KV caches resident on the device across
steps; every call is replayed on the CPU oracle with the same inputs."""
import numpy as np
import pytest

from oracle.halfs import BF16
from plumbing import Compare, Dims, OracleOps, Weights, run_trace
from util import ATOL_FEW_KEYS

pytestmark = pytest.mark.gpu


class GpuOps:
    def __init__(self, gpu, dims):
        self.g, self.D = gpu, dims
        nbytes = dims.num_pages * dims.page * dims.hk * dims.d * 2
        self.kc = [gpu.DeviceBuffer.zeros((nbytes,), np.uint8) for _ in range(dims.layers)]
        self.vc = [gpu.DeviceBuffer.zeros((nbytes,), np.uint8) for _ in range(dims.layers)]
        self.shape = (dims.num_pages, dims.page, dims.hk, dims.d)

    def _ok(self, rc=0):
        assert rc == 0, self.g.last_error()
        self.g.check()

    def rms_norm(self, x, w):
        rows, hidden = x.shape
        dx, dw = self.g.DeviceBuffer.from_numpy(x), self.g.DeviceBuffer.from_numpy(w)
        dy = self.g.DeviceBuffer(x.nbytes)
        self._ok(self.g.lib.atoma_rms_norm(dx.ptr, dw.ptr, dy.ptr, rows, hidden, hidden, hidden, self.D.eps, BF16, None))
        self.g.synchronize()
        return dy.numpy(np.uint16, x.shape)

    def rope_qk(self, q, k, cos, sin, pos):
        D = self.D
        dq, dk = self.g.DeviceBuffer.from_numpy(q), self.g.DeviceBuffer.from_numpy(k)
        dc, ds = self.g.DeviceBuffer.from_numpy(cos), self.g.DeviceBuffer.from_numpy(sin)
        dp = self.g.DeviceBuffer.from_numpy(np.asarray(pos, np.int64))
        self._ok(self.g.lib.atoma_rope_qk(dq.ptr, dk.ptr, dc.ptr, ds.ptr, dp.ptr, q.shape[0], D.h, D.hk, D.d,
                                          D.h * D.d, D.hk * D.d, BF16, 1, None))
        self.g.synchronize()
        return dq.numpy(np.uint16, q.shape), dk.numpy(np.uint16, k.shape)

    def reshape_and_cache(self, layer, k, v, slots):
        D = self.D
        dk, dv = self.g.DeviceBuffer.from_numpy(k), self.g.DeviceBuffer.from_numpy(v)
        ds = self.g.DeviceBuffer.from_numpy(np.asarray(slots, np.int64))
        self.g.lib.reshape_and_cache_flash(dk.ptr, dv.ptr, self.kc[layer].ptr, self.vc[layer].ptr, ds.ptr,
                                           D.page * D.hk * D.d, k.shape[0], D.hk, D.d, D.page, D.hk * D.d, D.hk * D.d,
                                           BF16, None)
        self._ok()
        self.g.synchronize()

    def caches(self, layer):
        return self.kc[layer].numpy(np.uint16, self.shape), self.vc[layer].numpy(np.uint16, self.shape)

    def prefill(self, q, k, v, cu):
        D = self.D
        dq, dk, dv = (self.g.DeviceBuffer.from_numpy(a) for a in (q, k, v))
        do = self.g.DeviceBuffer(q.nbytes)
        dcu = self.g.DeviceBuffer.from_numpy(np.asarray(cu, np.int32))
        L = int(np.diff(cu).max())
        self.g.run_mha(dq, dk, dv, do, b=len(cu) - 1, h=D.h, h_k=D.hk, d=D.d, seqlen_q=L, seqlen_k=L,
                       softmax_scale=D.d ** -0.5, is_bf16=BF16, q_strides=(0, D.h * D.d, D.d), o_strides=(0, D.h * D.d, D.d),
                       k_strides=(0, D.hk * D.d, D.d), v_strides=(0, D.hk * D.d, D.d), is_causal=1, cu_seqlens_q=dcu,
                       cu_seqlens_k=dcu)
        self.g.synchronize()
        return do.numpy(np.uint16, q.shape)

    def decode(self, layer, q, bt, lens):
        D = self.D
        B = q.shape[0]
        dq = self.g.DeviceBuffer.from_numpy(q)
        do = self.g.DeviceBuffer(q.nbytes)
        dbt = self.g.DeviceBuffer.from_numpy(np.ascontiguousarray(bt, np.int32))
        dl = self.g.DeviceBuffer.from_numpy(np.asarray(lens, np.int32))
        kstr = (D.page * D.hk * D.d, D.hk * D.d, D.d)
        self.g.run_mha(dq, self.kc[layer], self.vc[layer], do, b=B, h=D.h, h_k=D.hk, d=D.d, seqlen_q=1,
                       seqlen_k=bt.shape[1] * D.page, softmax_scale=D.d ** -0.5, is_bf16=BF16,
                       q_strides=(D.h * D.d, D.h * D.d, D.d), o_strides=(D.h * D.d, D.h * D.d, D.d), k_strides=kstr,
                       v_strides=kstr, cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                       block_table_batch_stride=bt.shape[1], page_block_size=D.page, force_split_kernel=True,
                       unpadded_lse=False)
        self.g.synchronize()
        return do.numpy(np.uint16, q.shape)


class GpuOpsFused(GpuOps):
    """RoPE(q, k) and the KV-cache write in one launch (atoma_rope_qk_cache)."""

    def rope_qk_cache(self, layer, q, k, v, cos, sin, pos, slots):
        D = self.D
        dq, dk, dv = (self.g.DeviceBuffer.from_numpy(a) for a in (q, k, v))
        dc, ds = self.g.DeviceBuffer.from_numpy(cos), self.g.DeviceBuffer.from_numpy(sin)
        dp = self.g.DeviceBuffer.from_numpy(np.asarray(pos, np.int64))
        dsl = self.g.DeviceBuffer.from_numpy(np.asarray(slots, np.int64))
        self._ok(self.g.lib.atoma_rope_qk_cache(dq.ptr, dk.ptr, dv.ptr, self.kc[layer].ptr, self.vc[layer].ptr, dsl.ptr,
                                                dc.ptr, ds.ptr, dp.ptr, q.shape[0], D.h, D.hk, D.d, D.h * D.d, D.hk * D.d,
                                                D.hk * D.d, D.page * D.hk * D.d, D.page, BF16, 1, None))
        self.g.synchronize()
        return dq.numpy(np.uint16, q.shape), dk.numpy(np.uint16, k.shape)


@pytest.mark.parametrize("ops_cls", [GpuOps, GpuOpsFused], ids=["separate-ops", "fused-rope-cache"])
def test_continuous_batching_trace_op_by_op(gpu, ops_cls):
    cmp = Compare()
    hist = run_trace(ops_cls(gpu, Dims), Weights(Dims), Dims, check=OracleOps(Dims), cmp=cmp, steps=32)
    assert len(hist[0]) == 32 and len(hist[1]) == 10 and len(hist[3]) >= 19
    assert cmp.calls >= 2 * 33
    assert cmp.rope_exact, "RoPE (per-op rounding) must be bit-exact"
    assert cmp.cache_exact, "device KV cache != oracle KV cache after reshape_and_cache_flash"
    assert cmp.rms_ulps <= 1, cmp.rms_ulps
    assert cmp.attn_err <= ATOL_FEW_KEYS[BF16], cmp.attn_err      # 17..77 keys per row: the P-rounding bound (util.py)
