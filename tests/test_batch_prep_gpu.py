"""atoma_prepare_inputs on the device: one pinned H2D copy, and the uploaded block table / lengths / slot mapping driving
reshape_and_cache_flash and the paged decode directly (the same bytes the reference's tensors would hold)."""
import ctypes as C

import numpy as np
import pytest

from oracle import batch_prep_oracle as BO
from oracle.halfs import BF16
from util import rand_half, assert_close, c_attention, attn_atol

pytestmark = pytest.mark.gpu


def test_prepare_inputs_upload_and_use(gpu):
    rng = np.random.default_rng(21)
    B, h, hk, d, page = 24, 8, 2, 128, 16
    seqs, nxt = [], 1
    for _ in range(B):
        L = int(rng.integers(1, 700))
        n = (L + page - 1) // page
        seqs.append(dict(is_prompt=False, tokens=rng.integers(0, 32000, L), chunk=1, block_table=rng.permutation(np.arange(nxt, nxt + n))))
        nxt += n
    ref = BO.prepare_inputs(seqs, page)
    arr, keep = gpu.make_seq_descs(seqs)
    lay = gpu.BatchLayout()
    assert gpu.lib.atoma_prepare_inputs(arr, B, page, 0, 0, None, 0, None, 0, C.byref(lay), None) == 0
    host = gpu.lib.atoma_host_alloc(lay.total_bytes)
    dev = gpu.DeviceBuffer(lay.total_bytes)
    dev.fill_bytes(0xEE)
    st = gpu.Stream()
    assert gpu.lib.atoma_prepare_inputs(arr, B, page, 0, 0, host, lay.total_bytes, dev.ptr, lay.total_bytes, C.byref(lay), st.s) == 0, gpu.last_error()
    st.synchronize()
    got = gpu.unpack_batch(dev.numpy(np.uint8, (lay.total_bytes,)), lay)
    for k in ("input_tokens", "input_positions", "slot_mapping", "seq_lens", "context_lens", "query_start_loc", "seq_start_loc", "block_tables"):
        assert np.array_equal(got[k], ref[k]), k
    # the uploaded metadata as kernel arguments: write this step's K/V rows through slot_mapping, then decode over block_tables / seq_lens
    lens = ref["seq_lens"].astype(np.int32)
    kc, vc = rand_half(rng, (nxt, page, hk, d), BF16), rand_half(rng, (nxt, page, hk, d), BF16)
    knew, vnew = rand_half(rng, (B, hk, d), BF16), rand_half(rng, (B, hk, d), BF16)
    q = rand_half(rng, (B, 1, h, d), BF16)
    dkc, dvc, dk, dv, dq = (gpu.DeviceBuffer.from_numpy(a) for a in (kc, vc, knew, vnew, q))
    do = gpu.DeviceBuffer(q.nbytes)
    gpu.lib.reshape_and_cache_flash(dk.ptr, dv.ptr, dkc.ptr, dvc.ptr, dev.ptr + lay.off_slot_mapping, page * hk * d, B, hk, d, page,
                                    hk * d, hk * d, 1, st.s)
    mb = int(lay.max_block_table_len)
    gpu.run_mha(dq, dkc, dvc, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=mb * page, softmax_scale=d ** -0.5, is_bf16=1,
                q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=dev.ptr + lay.off_seq_lens, is_seqlens_k_cumulative=False,
                block_table=dev.ptr + lay.off_block_tables, block_table_batch_stride=mb, page_block_size=page,
                force_split_kernel=True, unpadded_lse=False, stream=st.s)
    st.synchronize()
    out = do.numpy(np.uint16, q.shape)
    slots = ref["slot_mapping"]
    kc2, vc2 = kc.copy(), vc.copy()
    kc2.reshape(-1, hk, d)[slots] = knew
    vc2.reshape(-1, hk, d)[slots] = vnew
    assert np.array_equal(dkc.numpy(np.uint16, kc.shape), kc2) and np.array_equal(dvc.numpy(np.uint16, vc.shape), vc2)
    bt = ref["block_tables"].astype(np.int32)
    want = c_attention(q, kc2, vc2, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=mb * page, scale=d ** -0.5, is_bf16=1,
                       q_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d),
                       o_shape=q.shape, o_strides=(h * d, h * d, d), cu_k=lens, k_cumulative=False, block_table=bt, page=page)
    for i, L in enumerate(lens):
        assert_close(out[i], want[i], BF16, atol=attn_atol(BF16, L), what=f"decode over the uploaded metadata, seq {i} (L={L})")
    gpu.lib.atoma_host_free(host)
