"""Litmus stress for the in-launch merges.  The arrival tickets are relaxed atomics around write-through stores
(csrc/sync_ticket.h): right by what the gfx950 instructions do, outside what the HIP / LLVM memory model promises.  The static half of the
guard reads the emitted ISA (tests/test_kernel_resources.py: every partial store / load carries sc1, every ticket sits behind a vmcnt(0)
drain); this is the dynamic half: hundreds of launches whose cut points move from launch to launch (ragged lengths drawn anew every time), the
scratch block -- every fp32 partial, LSE and slab -- filled with NaN poison before each launch, a second stream hammering HBM and the L2s
beside them so that arrival order and cache residency keep changing, and EVERY output word checked: a launch must return exactly the bits of
its twin launched under different load (a stale or poisoned partial shows up as a NaN or as a different sum), and sampled launches the
oracle's values."""
import ctypes as C

import numpy as np
import pytest

from oracle.halfs import BF16, to_f32
from util import rand_half, assert_close, c_attention, attn_atol

pytestmark = pytest.mark.gpu


def poison_scratch(gpu, st):
    p, n = C.c_void_p(), C.c_int64()
    assert gpu.lib.atoma_debug_workspace(st.s, C.byref(p), C.byref(n)) == 0, gpu.last_error()
    if p.value and n.value:
        gpu.hip_check(gpu.hip.hipMemsetAsync(p, 0xFF, n.value, st.s), "poison the scratch block")      # 0xFFFFFFFF: a NaN in every fp32 partial


class Noise:
    """a second stream that keeps the memory system busy: device-to-device copies of a buffer larger than the L2s, started before a launch"""

    def __init__(self, gpu, mbytes=96):
        self.gpu, self.st = gpu, gpu.Stream()
        self.a, self.b = gpu.DeviceBuffer(mbytes << 20), gpu.DeviceBuffer(mbytes << 20)
        self.a.fill_bytes(1)

    def kick(self, copies):
        for _ in range(copies):
            self.gpu.hip_check(self.gpu.hip.hipMemcpyAsync(self.b.ptr, self.a.ptr, self.a.nbytes, 3, self.st.s), "noise copy")

    def drain(self):
        self.st.synchronize()


def test_balanced_line_cut_points_under_poison_and_load(gpu):
    rng = np.random.default_rng(61)
    B, h, hk, d, page, cap = 272, 8, 4, 128, 16, 2400
    pps = (cap + page - 1) // page
    nb = B * pps
    kc, vc = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    bt = rng.permutation(nb).astype(np.int32).reshape(B, pps)
    q = rand_half(rng, (B, 1, h, d), BF16)
    dq, dk, dv, dbt = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt))
    dl = gpu.DeviceBuffer.zeros((B,), np.int32)
    do = gpu.DeviceBuffer(q.nbytes)
    st, noise = gpu.Stream(), Noise(gpu)

    def call():
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=BF16,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                    v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                    block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
    launches = 0
    for it in range(150):
        # adversarial lengths: a few long stragglers among short sequences, uniform, narrow spreads, zeros -- every launch cuts elsewhere
        kind = it % 5
        if kind == 0:
            lens = rng.integers(1, cap, B)
        elif kind == 1:
            lens = rng.integers(1, 64, B); lens[rng.integers(0, B, 3)] = cap
        elif kind == 2:
            lens = np.full(B, int(rng.integers(16, cap)))
        elif kind == 3:
            lo = int(rng.integers(1, cap - 200)); lens = rng.integers(lo, lo + 200, B)
        else:
            lens = rng.integers(0, cap, B); lens[rng.integers(0, B, B // 4)] = 0
        lens = lens.astype(np.int32)
        dl.upload(lens)
        outs = []
        for load in (0, 3):                             # idle chip, then beside the noise stream
            poison_scratch(gpu, st)
            do.fill_bytes(0xEE)
            noise.kick(load)
            call()
            st.synchronize()
            noise.drain()
            outs.append(do.numpy(np.uint16, q.shape).copy())
            launches += 1
        f = to_f32(outs[0], BF16)
        assert np.isfinite(f).all(), f"launch {it}: a poisoned partial reached the output"
        assert np.array_equal(outs[0], outs[1]), f"launch {it} (kind {kind}): the bits depend on the load beside the launch"
        assert not outs[0][lens == 0].any()
        if it % 25 == 0:
            assert "balanced" in gpu.lib.atoma_last_decode_kernel().decode()
            ref = c_attention(q, kc, vc, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, scale=d ** -0.5, is_bf16=BF16, q_strides=(h * d, h * d, d),
                              k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d), o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens,
                              k_cumulative=False, block_table=bt, page=page)
            for j in range(0, B, 7):
                assert_close(outs[0][j], ref[j], BF16, atol=attn_atol(BF16, int(lens[j])), what=f"launch {it} seq {j} (L={lens[j]})")
    assert launches == 300


@pytest.mark.parametrize("B,K,N", [(48, 2048, 4096), (256, 4096, 4096), (200, 14336, 1024)])
def test_k_split_projections_under_poison_and_load(gpu, B, K, N):
    """the tile kernels' K-split merge (linear_tile_kernel at 48 rows, linear_wide_kernel at 200 / 256): 120 launches, slabs poisoned, load beside
    every second one -- always the bits of the first launch, which meet the oracle bound"""
    from oracle import linear_oracle as LO
    rng = np.random.default_rng(B + K)
    x, w, r = rand_half(rng, (B, K), BF16), rand_half(rng, (N, K), BF16, K ** -0.5), rand_half(rng, (B, N), BF16)
    dx, dw, dr = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, r))
    y = gpu.DeviceBuffer.zeros((B, N), np.uint16)
    st, noise = gpu.Stream(), Noise(gpu)

    def call():
        assert gpu.lib.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y.ptr, B, K, N, K, K, N, N, BF16, st.s) == 0, gpu.last_error()
    call()
    st.synchronize()
    want = y.numpy(np.uint16, (B, N)).copy()
    ref = to_f32(LO.linear(x, w, BF16), BF16) + to_f32(r, BF16)
    assert np.abs(to_f32(want, BF16) - ref).max() <= 2.0 ** -7 * np.abs(ref).max() + 1e-2
    for it in range(120):
        poison_scratch(gpu, st)
        y.fill_bytes(0xEE)
        noise.kick(2 if it % 2 else 0)
        call()
        st.synchronize()
        noise.drain()
        assert np.array_equal(y.numpy(np.uint16, (B, N)), want), f"launch {it}: a K-split merge returned other bits"
