"""The launch-bound end of the path (a decode step at small batch: a dozen 5-60 us kernels per layer) is meant to be
captured in a hipGraph by the caller.  Everything the library does on a stream must therefore be capturable once its
split-KV workspace exists (one eager call on that stream first): no allocation, synchronisation or legacy-stream
work inside the entry points.  Replays must read the buffers' current contents."""
import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle.halfs import BF16, to_f32
from util import rand_half, make_paged_cache, assert_close, ATOL_VS_F32, c_attention, attn_atol

pytestmark = pytest.mark.gpu


def test_decode_step_ops_capture_and_replay(gpu):
    rng = np.random.default_rng(5)
    layers, B, h, hk, d, page, hidden, vocab = 3, 2, 8, 2, 128, 16, 512, 1000
    lens = np.array([1500, 700], np.int32)
    nb = int(sum((x + page - 1) // page for x in lens)) + 2
    st = gpu.Stream()
    caches, bts = [], None
    for _ in range(layers):
        kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
        caches.append((kc, vc, gpu.DeviceBuffer.from_numpy(kc), gpu.DeviceBuffer.from_numpy(vc)))
        bts = bt
    dbt, dl = gpu.DeviceBuffer.from_numpy(bts), gpu.DeviceBuffer.from_numpy(lens)
    x = rand_half(rng, (B, hidden), BF16)
    w = rand_half(rng, (hidden,), BF16, 0.5)
    q = rand_half(rng, (B, 1, h, d), BF16)
    logits = rng.standard_normal((B, vocab)).astype(np.float32)
    dx, dw, dy = gpu.DeviceBuffer.from_numpy(x), gpu.DeviceBuffer.from_numpy(w), gpu.DeviceBuffer(x.nbytes)
    dq, dlog = gpu.DeviceBuffer.from_numpy(q), gpu.DeviceBuffer.from_numpy(logits)
    outs = [gpu.DeviceBuffer(q.nbytes) for _ in range(layers)]
    didx, dval = gpu.DeviceBuffer.zeros((B,), np.int32), gpu.DeviceBuffer.zeros((B,), np.float32)

    def step():
        for l in range(layers):
            assert gpu.lib.atoma_rms_norm(dx.ptr, dw.ptr, dy.ptr, B, hidden, hidden, hidden, 1e-5, BF16, st.s) == 0
            gpu.run_mha(dq, caches[l][2], caches[l][3], outs[l], b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bts.shape[1] * page,
                        softmax_scale=d ** -0.5, is_bf16=BF16, q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d),
                        k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl,
                        is_seqlens_k_cumulative=False, block_table=dbt, block_table_batch_stride=bts.shape[1], page_block_size=page,
                        force_split_kernel=True, unpadded_lse=False, stream=st.s)
        assert gpu.lib.atoma_argmax_rows(dlog.ptr, B, vocab, vocab, 2, didx.ptr, dval.ptr, st.s) == 0

    step()                                   # eager: creates the stream's split-KV workspace
    st.synchronize()
    eager = [o.numpy(np.uint16, q.shape) for o in outs]
    for o in outs:
        o.fill_bytes(0)
    with gpu.Graph.capture(st) as g:
        step()
    g.launch()
    st.synchronize()
    for l in range(layers):
        assert np.array_equal(outs[l].numpy(np.uint16, q.shape), eager[l]), f"layer {l}: graph replay differs from the eager run"
    assert np.array_equal(didx.numpy(np.int32, (B,)), logits.argmax(1))
    # new inputs in the same buffers: the replay must see them
    q2 = rand_half(rng, (B, 1, h, d), BF16)
    dq.upload(q2)
    logits2 = rng.standard_normal((B, vocab)).astype(np.float32)
    dlog.upload(logits2)
    g.launch()
    st.synchronize()
    for l in range(layers):
        ref = A.flash_attn_kv_cache(q2, caches[l][0], caches[l][1], d ** -0.5, BF16, bts, lens)
        assert_close(outs[l].numpy(np.uint16, q.shape), ref, BF16, atol=ATOL_VS_F32[BF16], what=f"graph replay, layer {l}")
    assert np.array_equal(didx.numpy(np.int32, (B,)), logits2.argmax(1))


def test_capture_without_warm_up_fails_loudly(gpu):
    """The split scratch of a stream cannot grow during capture: the call reports it instead of launching on freed memory."""
    rng = np.random.default_rng(5)
    B, h, hk, d, page, L = 2, 8, 2, 128, 16, 700
    from util import rand_half, make_paged_cache
    lens = np.full(B, L, np.int32)
    nb = B * ((L + page - 1) // page) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, 1, lens)
    q = rand_half(rng, (B, 1, h, d), 1)
    dq, dk, dv, dbt, dl = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt, lens))
    do = gpu.DeviceBuffer(q.nbytes)
    st = gpu.Stream()                                   # a fresh stream: no scratch yet

    def call():
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, softmax_scale=d ** -0.5, is_bf16=1,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                    v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                    block_table_batch_stride=bt.shape[1], page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
    # Stream handles are recycled, so this stream may inherit a scratch that is already large enough; if it is not, the
    # call must say so (RuntimeError from atoma_last_error) instead of launching on memory it could not grow.
    try:
        with gpu.Graph.capture(st):
            call()
    except RuntimeError as e:
        assert "hipGraph capture" in str(e)
    call()                                              # eagerly: allocates the scratch
    st.synchronize()
    want = do.numpy(np.uint16, q.shape).copy()
    do.fill_bytes(0)
    with gpu.Graph.capture(st) as g:
        call()
    g.launch()
    st.synchronize()
    assert np.array_equal(do.numpy(np.uint16, q.shape), want)


def _decode_case(gpu, rng, B, L, h=8, hk=2, d=128, page=16):
    lens = np.full(B, L, np.int32)
    nb = B * ((L + page - 1) // page) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    q = rand_half(rng, (B, 1, h, d), BF16)
    dq, dk, dv, dbt, dl = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt, lens))
    do = gpu.DeviceBuffer(q.nbytes)

    def call(st):
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, softmax_scale=d ** -0.5, is_bf16=1,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                    v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                    block_table_batch_stride=bt.shape[1], page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
    ref = A.flash_attn_kv_cache(q, kc, vc, d ** -0.5, BF16, bt, lens)
    keep = (dq, dk, dv, dbt, dl)
    return call, do, q.shape, ref, keep


def test_warmup_makes_first_call_capturable(gpu):
    """atoma_warmup sizes the stream's scratch for every decode call up to the given shape, so a capture needs no eager call first
    (the Rust caller believes the callee is stateless: SURVEY 8b)."""
    assert gpu.lib.atoma_release_workspaces() == 0        # forget whatever earlier tests left behind
    rng = np.random.default_rng(11)
    st = gpu.Stream()
    assert gpu.lib.atoma_warmup(st.s, 4, 8, 2, 128, 4096, 0) == 0, gpu.last_error()
    for B, L in ((1, 4096), (3, 1000), (4, 70)):          # split-KV, split-KV, tiny
        call, do, shape, ref, keep = _decode_case(gpu, rng, B, L)
        with gpu.Graph.capture(st) as g:
            call(st)
        g.launch()
        st.synchronize()
        assert_close(do.numpy(np.uint16, shape), ref, BF16, atol=ATOL_VS_F32[BF16], what=f"captured without an eager call, B={B} L={L}")
    assert gpu.lib.atoma_warmup(st.s, 0, 8, 2, 128, 4096, 0) == -1 and "invalid" in gpu.last_error()


def test_warmup_covers_the_balanced_line_of_both_caches(gpu):
    """A resident batch (uniform, d = 128, 4 kv heads) takes the balanced line -- partial slots for every wavefront, 16 q heads wide over
    an fp8 cache: atoma_warmup must have reserved them, for the 16-bit and the fp8 cache alike, so that the FIRST call can be a capture."""
    from oracle import fp8_oracle as F8
    assert gpu.lib.atoma_release_workspaces() == 0
    rng = np.random.default_rng(12)
    B, L, h, hk, d, page = 300, 200, 16, 4, 128, 16
    st = gpu.Stream()
    assert gpu.lib.atoma_warmup(st.s, B, h, hk, d, 4096, 0) == 0, gpu.last_error()
    call, do, shape, ref, keep = _decode_case(gpu, rng, B, L, h=h, hk=hk)
    with gpu.Graph.capture(st) as g:
        call(st)
    g.launch()
    st.synchronize()
    assert "balanced" in gpu.lib.atoma_last_decode_kernel().decode()
    assert_close(do.numpy(np.uint16, shape), ref, BF16, atol=ATOL_VS_F32[BF16], what="16-bit cache, captured without an eager call")
    # the same heads over an fp8 cache
    lens = np.full(B, L, np.int32)
    nb = B * ((L + page - 1) // page) + 1
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, BF16, lens)
    ks, vs = np.full(hk, 0.02, np.float32), np.full(hk, 0.02, np.float32)
    kc8 = F8.quantize(kc.reshape(-1, hk, d), BF16, ks).reshape(nb, page, hk, d)
    vc8 = F8.quantize(vc.reshape(-1, hk, d), BF16, vs).reshape(nb, page, hk, d)
    q = rand_half(rng, (B, h, d), BF16)
    dq, dk, dv, dbt, dl, dks, dvs = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc8, vc8, np.ascontiguousarray(bt, np.int32), lens, ks, vs))
    do8 = gpu.DeviceBuffer(q.nbytes)

    def call8(stream):
        rc = gpu.lib.atoma_paged_decode_fp8(dq.ptr, dk.ptr, dv.ptr, do8.ptr, dks.ptr, dvs.ptr, dbt.ptr, dl.ptr, B, h, hk, d, bt.shape[1], page,
                                            h * d, d, h * d, d, page * hk * d, hk * d, d, float(d ** -0.5), BF16, stream)
        assert rc == 0, gpu.last_error()
    with gpu.Graph.capture(st) as g8:
        call8(st.s)
    g8.launch()
    st.synchronize()
    got = do8.numpy(np.uint16, q.shape).copy()
    do8.fill_bytes(0)
    call8(st.s)
    st.synchronize()
    assert np.array_equal(got, do8.numpy(np.uint16, q.shape)), "fp8 cache: replay of a first-call capture equals the eager call"


def test_warmup_makes_projection_with_in_launch_merge_capturable(gpu):
    """The 17..64-row projection kernel merges its K splits inside the launch: arrival counters, fp32 tiles in the stream's scratch and a
    raised LDS limit.  After atoma_warmup (counters + the limit for every variant; extra_bytes covers the tiles) a CAPTURE may be the
    first call, and replays -- the counters come back to zero each time -- reproduce the eager result bit for bit, also with new inputs."""
    from oracle import linear_oracle as LO
    from util import rand_half
    assert gpu.lib.atoma_release_workspaces() == 0
    rng = np.random.default_rng(13)
    B, K, N = 48, 2048, 4096                              # 64-row tiles x 2 K splits x 128 workgroups... merged by the last arriver
    st = gpu.Stream()
    assert gpu.lib.atoma_warmup(st.s, 4, 8, 2, 128, 1024, 8 * N * 64 * 4) == 0, gpu.last_error()
    x, w, r = rand_half(rng, (B, K), BF16), rand_half(rng, (N, K), BF16, K ** -0.5), rand_half(rng, (B, N), BF16)
    dx, dw, dr = (gpu.DeviceBuffer.from_numpy(a) for a in (x, w, r))
    y = gpu.DeviceBuffer.zeros((B, N), np.uint16)
    with gpu.Graph.capture(st) as g:
        assert gpu.lib.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, y.ptr, B, K, N, K, K, N, N, BF16, st.s) == 0, gpu.last_error()
    for rep in range(3):
        if rep == 2:
            x = rand_half(rng, (B, K), BF16)
            dx.upload(x)
        y.fill_bytes(0)
        g.launch()
        st.synchronize()
        got = to_f32(y.numpy(np.uint16, (B, N)), BF16)
        ref = to_f32(LO.linear(x, w, BF16), BF16) + to_f32(r, BF16)
        assert (np.abs(got - ref) <= 2.0 ** -6 * np.abs(ref) + 1e-2).all(), rep
        if rep == 0:
            first = y.numpy(np.uint16, (B, N)).copy()
        if rep == 1:
            assert np.array_equal(y.numpy(np.uint16, (B, N)), first)
    yb = gpu.DeviceBuffer.zeros((B, N), np.uint16)
    assert gpu.lib.atoma_linear_decode_residual(dx.ptr, dw.ptr, dr.ptr, yb.ptr, B, K, N, K, K, N, N, BF16, st.s) == 0
    st.synchronize()
    assert np.array_equal(yb.numpy(np.uint16, (B, N)), y.numpy(np.uint16, (B, N)))


def test_scratch_growth_keeps_captured_graphs_valid(gpu):
    """A graph captured against a small scratch block must stay valid after a later, larger eager call grew the scratch:
    the old block is retired, not freed (replaying used freed memory)."""
    assert gpu.lib.atoma_release_workspaces() == 0
    rng = np.random.default_rng(12)
    st = gpu.Stream()
    small_call, small_o, small_shape, small_ref, k1 = _decode_case(gpu, rng, 1, 2048)
    small_call(st)                                        # eager: allocates a small block
    st.synchronize()
    with gpu.Graph.capture(st) as g:
        small_call(st)
    big_call, big_o, big_shape, big_ref, k2 = _decode_case(gpu, rng, 16, 3000, h=32, hk=8)
    big_call(st)                                          # needs (much) more scratch: grows
    st.synchronize()
    assert_close(big_o.numpy(np.uint16, big_shape), big_ref, BF16, atol=ATOL_VS_F32[BF16], what="call that grew the scratch")
    # churn the allocator so that a freed block would be reused and overwritten
    junk = [gpu.DeviceBuffer(1 << 20) for _ in range(8)]
    for j in junk:
        j.fill_bytes(0xff)
    small_o.fill_bytes(0)
    g.launch()
    big_call(st)
    g.launch()
    st.synchronize()
    assert_close(small_o.numpy(np.uint16, small_shape), small_ref, BF16, atol=ATOL_VS_F32[BF16], what="replay after the scratch grew")
    del g
    assert gpu.lib.atoma_release_workspaces() == 0


def test_balanced_line_replays_with_other_lengths(gpu):
    """The balanced line plans on the device and merges cut sequences inside the launch (arrival counters that return to zero): ONE captured
    graph must serve batches whose lengths change between replays -- ragged, uniform, with empty sequences, with a straggler -- and a
    counter reset between two replays (atoma_reset_sync_counters, for hosts that saw a failed launch) must not change a bit."""
    rng = np.random.default_rng(77)
    B, h, hk, d, page, cap = 272, 8, 4, 128, 16, 3000
    pps = (cap + page - 1) // page
    nb = B * pps
    kc, vc = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    bt = rng.permutation(nb).astype(np.int32).reshape(B, pps)
    q = rand_half(rng, (B, 1, h, d), BF16)
    dq, dk, dv, dbt = (gpu.DeviceBuffer.from_numpy(a) for a in (q, kc, vc, bt))
    dl = gpu.DeviceBuffer.zeros((B,), np.int32)
    do = gpu.DeviceBuffer(q.nbytes)
    st = gpu.Stream()

    def call():
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=BF16,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                    v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                    block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)
    first = rng.integers(1, cap, B).astype(np.int32)
    dl.upload(first)
    call()                                   # eager once: scratch and counters exist
    st.synchronize()
    assert "balanced" in gpu.lib.atoma_last_decode_kernel().decode()
    with gpu.Graph.capture(st) as g:
        call()
    batches = [first, np.full(B, 1777, np.int32), rng.integers(0, 40, B).astype(np.int32)]
    strag = rng.integers(1, 100, B).astype(np.int32)
    strag[200] = cap
    batches.append(strag)
    batches.append(rng.integers(cap // 2, cap, B).astype(np.int32))
    for i, lens in enumerate(batches):
        dl.upload(lens)
        do.fill_bytes(0xEE)
        g.launch()
        st.synchronize()
        got = do.numpy(np.uint16, q.shape).copy()
        ref = c_attention(q, kc, vc, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, scale=d ** -0.5, is_bf16=BF16, q_strides=(h * d, h * d, d),
                          k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d), o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens,
                          k_cumulative=False, block_table=bt, page=page)
        for j, L in enumerate(lens):
            assert_close(got[j], ref[j], BF16, atol=attn_atol(BF16, L), what=f"replay {i} seq {j} (L={L})")
        assert not got[lens == 0].any()
        if i == 2:
            assert gpu.lib.atoma_reset_sync_counters(st.s) == 0, gpu.last_error()
        g.launch()
        st.synchronize()
        assert np.array_equal(got, do.numpy(np.uint16, q.shape)), f"replay {i} twice"


def test_warmup_prefill_covers_a_ragged_varlen_batch(gpu):
    """atoma_warmup_prefill(stream, longest sequence, sequences, q heads) sizes the persistent prefill kernel's plan table exactly as the
    launch does -- padded to sequences x the LONGEST sequence's 256-row blocks -- so that the FIRST prefill call of a stream can be a capture
    even for a ragged batch (the old estimate, tokens / 256 + sequences blocks, was ~6 x too small for [2048, 128, 128, 128]);
    a table that is too small still fails loudly."""
    from oracle import attn_oracle as A
    from util import rand_half
    assert gpu.lib.atoma_release_workspaces() == 0
    rng = np.random.default_rng(23)
    h, hk, d = 8, 2, 128
    lens = np.array([2048, 128, 100, 128], np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T = int(cu[-1])
    q, k, v = rand_half(rng, (T, h, d), BF16), rand_half(rng, (T, hk, d), BF16), rand_half(rng, (T, hk, d), BF16)
    dq, dk, dv, dcu = (gpu.DeviceBuffer.from_numpy(a) for a in (q, k, v, cu))
    do = gpu.DeviceBuffer(q.nbytes)

    def call(st):
        gpu.run_mha(dq, dk, dv, do, b=len(lens), h=h, h_k=hk, d=d, seqlen_q=int(lens.max()), seqlen_k=int(lens.max()), softmax_scale=d ** -0.5, is_bf16=BF16,
                    q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d), is_causal=1,
                    cu_seqlens_q=dcu, cu_seqlens_k=dcu, stream=st.s)
    st = gpu.Stream()
    assert gpu.lib.atoma_warmup_prefill(st.s, 256, len(lens), h) == 0, gpu.last_error()     # too small: one block per sequence
    try:
        with gpu.Graph.capture(st):
            call(st)
        raise AssertionError("a prefill whose plan table must grow inside a capture should fail")
    except RuntimeError as e:
        assert "hipGraph capture" in str(e)
    st2 = gpu.Stream()
    assert gpu.lib.atoma_warmup_prefill(st2.s, int(lens.max()), len(lens), h) == 0, gpu.last_error()
    do.fill_bytes(0xEE)
    with gpu.Graph.capture(st2) as g:
        call(st2)
    g.launch()
    st2.synchronize()
    ref = A.flash_attn_varlen(q, k, v, cu, cu, d ** -0.5, True, BF16)
    assert_close(do.numpy(np.uint16, q.shape), ref, BF16, atol=ATOL_VS_F32[BF16], what="ragged varlen prefill captured as the stream's first call")
    assert gpu.lib.atoma_warmup_prefill(st2.s, 0, 4, 8) == -1 and "invalid" in gpu.last_error()
