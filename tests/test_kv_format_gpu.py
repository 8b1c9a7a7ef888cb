"""KV block container (atoma_kv_pack_blocks / atoma_kv_unpack_blocks): the image is byte-identical to oracle/kv_format_oracle.py,
survives a round trip into different pages of another cache bit-exactly, and corrupted / mismatched images are rejected."""
import ctypes as C

import numpy as np
import pytest

from oracle import kv_format_oracle as KF
from oracle.halfs import BF16, F16
from util import rand_half

pytestmark = pytest.mark.gpu
U8 = 5


class Header(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_uint32), ("dtype", C.c_uint32), ("num_layers", C.c_uint32), ("num_kv_heads", C.c_uint32),
                ("head_dim", C.c_uint32), ("block_size", C.c_uint32), ("num_blocks", C.c_uint64), ("page_bytes", C.c_uint64), ("payload_offset", C.c_uint64),
                ("total_bytes", C.c_uint64), ("checksum", C.c_uint64), ("reserved", C.c_uint8 * 56)]


@pytest.mark.parametrize("dtype,pinned", [(BF16, True), (F16, False), (U8, True)])
def test_pack_matches_oracle_and_round_trips(gpu, dtype, pinned):
    rng = np.random.default_rng(dtype + 2 * pinned)
    L, nb, page, hk, d, n = 3, 20, 16, 2, 128, 7
    mk = (lambda: rng.integers(0, 256, (nb, page, hk, d), dtype=np.uint8)) if dtype == U8 else (lambda: rand_half(rng, (nb, page, hk, d), dtype))
    kc, vc = [mk() for _ in range(L)], [mk() for _ in range(L)]
    dk, dv = [gpu.DeviceBuffer.from_numpy(a) for a in kc], [gpu.DeviceBuffer.from_numpy(a) for a in vc]
    ids = rng.permutation(nb)[:n].astype(np.int64)
    ks = rng.uniform(0.01, 0.1, (L, hk)).astype(np.float32) if dtype == U8 else None
    vs = rng.uniform(0.01, 0.1, (L, hk)).astype(np.float32) if dtype == U8 else None
    size = gpu.lib.atoma_kv_blocks_packed_size(L, hk, d, page, n, dtype)
    want = KF.pack(kc, vc, ids, dtype, ks, vs)
    assert size == len(want) and C.sizeof(Header) == 128
    if pinned:
        hptr = gpu.lib.atoma_host_alloc(size)
        img = np.ctypeslib.as_array(C.cast(hptr, C.POINTER(C.c_uint8)), shape=(size,))
    else:
        img = np.zeros(size, np.uint8)
        hptr = img.ctypes.data
    kp, vp = (C.c_void_p * L)(*[b.ptr for b in dk]), (C.c_void_p * L)(*[b.ptr for b in dv])
    sp = lambda a: None if a is None else a.ctypes.data
    st = gpu.Stream()
    rc = gpu.lib.atoma_kv_pack_blocks(kp, vp, L, hk, d, page, ids.ctypes.data, n, dtype, sp(ks), sp(vs), hptr, size, st.s)
    assert rc == 0, gpu.last_error()
    assert bytes(img) == want, "image differs from the oracle's byte layout"
    hdr = Header()
    assert gpu.lib.atoma_kv_read_header(hptr, size, C.byref(hdr)) == 0, gpu.last_error()
    assert (hdr.magic, hdr.num_layers, hdr.num_blocks, hdr.page_bytes, hdr.total_bytes) == (b"ATOMAKV1", L, n, page * hk * d * KF.ELT[dtype], size)
    # receiver: another cache, other pages
    kc2, vc2 = [np.zeros_like(a) for a in kc], [np.zeros_like(a) for a in vc]
    dk2, dv2 = [gpu.DeviceBuffer.from_numpy(a) for a in kc2], [gpu.DeviceBuffer.from_numpy(a) for a in vc2]
    dst = rng.permutation(nb)[:n].astype(np.int64)
    kp2, vp2 = (C.c_void_p * L)(*[b.ptr for b in dk2]), (C.c_void_p * L)(*[b.ptr for b in dv2])
    ks2 = np.zeros((L, hk), np.float32) if dtype == U8 else None
    vs2 = np.zeros((L, hk), np.float32) if dtype == U8 else None
    rc = gpu.lib.atoma_kv_unpack_blocks(hptr, size, kp2, vp2, L, hk, d, page, dtype, dst.ctypes.data, n, sp(ks2), sp(vs2), st.s)
    assert rc == 0, gpu.last_error()
    st.synchronize()
    KF.unpack(want, kc2, vc2, dst)
    for l in range(L):
        assert np.array_equal(dk2[l].numpy(), kc2[l]) and np.array_equal(dv2[l].numpy(), vc2[l])
        assert np.array_equal(kc2[l][dst], kc[l][ids])
    if dtype == U8:
        assert np.array_equal(ks2, ks) and np.array_equal(vs2, vs)
    # corruption and mismatch are refused before anything is written
    bad = img.copy()
    bad[size - 5] ^= 1
    assert gpu.lib.atoma_kv_unpack_blocks(bad.ctypes.data, size, kp2, vp2, L, hk, d, page, dtype, dst.ctypes.data, n, None, None, st.s) == -1
    assert "checksum" in gpu.last_error()
    assert gpu.lib.atoma_kv_unpack_blocks(hptr, size, kp2, vp2, L + 1, hk, d, page, dtype, dst.ctypes.data, n, None, None, st.s) == -1
    assert "geometry" in gpu.last_error()
    assert gpu.lib.atoma_kv_read_header(hptr, 64, C.byref(hdr)) == -1
    assert gpu.lib.atoma_kv_pack_blocks(kp, vp, L, hk, d, page, ids.ctypes.data, n, dtype, sp(ks), sp(vs), hptr, size - 1, st.s) == -1
    if pinned:
        gpu.lib.atoma_host_free(hptr)
