"""The arrival tickets of the kernels that merge their own pieces inside the launch (csrc/sync_ticket.h): a launch must not depend on
what EARLIER launches left in the arrival words -- the drop-in boundary is a stateless callee (csrc/src/ffi.rs:3-102).
Shown two ways: (1) the epoch of a launch (its AQL dispatch id) grows from launch to launch on a stream, eagerly and from a replayed
graph -- the property the scheme rests on; (2) the words are filled with what an aborted or concurrent launch could have left (counts
without a reset, words of older epochs) and the next call returns the same bits as before, with no atoma_reset_sync_counters."""
import ctypes as C

import numpy as np
import pytest

from oracle.halfs import BF16
from util import rand_half

pytestmark = pytest.mark.gpu


def sync_words(gpu, st):
    p, n = C.c_void_p(), C.c_int64()
    assert gpu.lib.atoma_debug_sync_words(st.s, C.byref(p), C.byref(n)) == 0, gpu.last_error()
    return p.value, n.value


def epoch_of_a_launch(gpu, st, buf):
    assert gpu.lib.atoma_debug_launch_epoch(st.s, buf.ptr) == 0, gpu.last_error()
    st.synchronize()
    return int(buf.numpy(np.uint64, (1,))[0])


def poison(gpu, st, rng, kind):
    """What the words may hold when a launch starts: `counts` = arrivals that nobody reset (round 4's failure mode: a word left at 1..7),
    `old_epochs` = complete garbage from launches before this one (any epoch below the current, any count)."""
    ptr, n = sync_words(gpu, st)
    eb = gpu.DeviceBuffer(8)
    now = epoch_of_a_launch(gpu, st, eb) >> 16
    if kind == "counts":
        w = rng.integers(1, 8, n).astype(np.uint64)
    else:
        w = (rng.integers(0, now, n).astype(np.uint64) << np.uint64(16)) | rng.integers(0, 65536, n).astype(np.uint64)
    gpu.hip_check(gpu.hip.hipMemcpy(C.c_void_p(ptr), w.ctypes.data, n * 8, gpu.H2D), "poison the arrival words")


def test_epochs_grow_from_launch_to_launch_also_under_graph_replay(gpu):
    st = gpu.Stream()
    buf = gpu.DeviceBuffer(8)
    seen = [epoch_of_a_launch(gpu, st, buf) for _ in range(4)]
    with gpu.Graph.capture(st) as g:
        assert gpu.lib.atoma_debug_launch_epoch(st.s, buf.ptr) == 0
    for _ in range(4):
        g.launch()
        st.synchronize()
        seen.append(int(buf.numpy(np.uint64, (1,))[0]))
    seen.append(epoch_of_a_launch(gpu, st, buf))
    assert all(e & 0xFFFF == 0 and e > 0 for e in seen), seen
    assert all(b > a for a, b in zip(seen, seen[1:])), f"epochs must strictly grow along a stream: {[e >> 16 for e in seen]}"


@pytest.mark.parametrize("kind", ["counts", "old_epochs"])
def test_balanced_line_ignores_stale_arrival_words(gpu, kind):
    """A ragged batch on the balanced line: nearly every (sequence, kv head) is cut and merged by its last wavefront."""
    rng = np.random.default_rng(5)
    B, h, hk, d, page, cap = 272, 8, 4, 128, 16, 3000
    pps = (cap + page - 1) // page
    nb = B * pps
    kc, vc = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    bt = rng.permutation(nb).astype(np.int32).reshape(B, pps)
    q = rand_half(rng, (B, 1, h, d), BF16)
    lens = rng.integers(1, cap, B).astype(np.int32)
    dq, dk, dv, dbt, dl = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt, lens))
    do = gpu.DeviceBuffer(q.nbytes)
    st = gpu.Stream()

    def call():
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=BF16,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                    v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                    block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
    call()
    st.synchronize()
    assert "balanced" in gpu.lib.atoma_last_decode_kernel().decode()
    want = do.numpy(np.uint16, q.shape).copy()
    with gpu.Graph.capture(st) as g:
        call()
    for replay in (False, True, False):
        poison(gpu, st, rng, kind)
        do.fill_bytes(0xEE)
        g.launch() if replay else call()
        st.synchronize()
        assert np.array_equal(do.numpy(np.uint16, q.shape), want), f"{kind}: stale arrival words changed the result ({'replay' if replay else 'eager'})"


@pytest.mark.parametrize("kind", ["counts", "old_epochs"])
def test_split_kv_across_workgroups_ignores_stale_arrival_words(gpu, kind):
    """Split-KV merged inside the launch with the last-arriver level across workgroups (decode_wg_merge = 3: forced)."""
    rng = np.random.default_rng(6)
    B, h, hk, d, page, L = 16, 32, 8, 128, 16, 8192
    pps = L // page
    nb = B * pps
    kc, vc = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    bt = rng.permutation(nb).astype(np.int32).reshape(B, pps)
    q = rand_half(rng, (B, 1, h, d), BF16)
    lens = np.full(B, L, np.int32)
    dq, dk, dv, dbt, dl = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt, lens))
    do = gpu.DeviceBuffer(q.nbytes)
    st = gpu.Stream()
    assert gpu.lib.atoma_set_option(b"decode_wg_merge", 3) == 0
    try:
        def call():
            gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=L, softmax_scale=d ** -0.5, is_bf16=BF16,
                        q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                        v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                        block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
        call()
        st.synchronize()
        assert "workgroups per sequence, merged in the launch" in gpu.lib.atoma_last_decode_kernel().decode(), gpu.lib.atoma_last_decode_kernel()
        want = do.numpy(np.uint16, q.shape).copy()
        for _ in range(3):
            poison(gpu, st, rng, kind)
            do.fill_bytes(0xEE)
            call()
            st.synchronize()
            assert np.array_equal(do.numpy(np.uint16, q.shape), want), f"{kind}: stale arrival words changed the result"
    finally:
        gpu.lib.atoma_set_option(b"decode_wg_merge", 1)


@pytest.mark.parametrize("kind", ["counts", "old_epochs"])
def test_tile_projection_ignores_stale_arrival_words(gpu, kind):
    """The 17..64-row projection kernel: K split merged by the last workgroup to arrive at a tile."""
    rng = np.random.default_rng(7)
    B, K, N = 48, 2048, 4096
    st = gpu.Stream()
    x, w, r = rand_half(rng, (B, K), BF16), rand_half(rng, (N, K), BF16, K ** -0.5), rand_half(rng, (B, N), BF16)
    dx, dw, dr = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, r))
    y = gpu.DeviceBuffer.zeros((B, N), np.uint16)

    def call():
        assert gpu.lib.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y.ptr, B, K, N, K, K, N, N, BF16, st.s) == 0, gpu.last_error()
    call()
    st.synchronize()
    want = y.numpy(np.uint16, (B, N)).copy()
    with gpu.Graph.capture(st) as g:
        call()
    for replay in (False, True, True, False):
        poison(gpu, st, rng, kind)
        y.fill_bytes(0)
        g.launch() if replay else call()
        st.synchronize()
        assert np.array_equal(y.numpy(np.uint16, (B, N)), want), f"{kind}: stale arrival words changed the result"
