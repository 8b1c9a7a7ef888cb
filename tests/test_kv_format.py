"""CPU side of the KV block container: the header is the documented 128 bytes, the library's checksum and header parser agree
with oracle/kv_format_oracle.py on an image built by the oracle alone (no GPU)."""
import ctypes as C
import os

import numpy as np

from oracle import kv_format_oracle as KF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "atoma-infer_amd", "lib", "libatoma_hip.so")


def test_oracle_image_parses_in_the_library():
    lib = C.CDLL(LIB)
    lib.atoma_last_error.restype = C.c_char_p
    lib.atoma_kv_read_header.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.atoma_kv_blocks_packed_size.argtypes = [C.c_int64] * 5 + [C.c_int]
    lib.atoma_kv_blocks_packed_size.restype = C.c_int64
    rng = np.random.default_rng(0)
    L, nb, page, hk, d = 2, 6, 16, 2, 64
    kc = [rng.integers(0, 65536, (nb, page, hk, d)).astype(np.uint16) for _ in range(L)]
    vc = [rng.integers(0, 65536, (nb, page, hk, d)).astype(np.uint16) for _ in range(L)]
    img = KF.pack(kc, vc, [4, 1, 3], 1)
    assert KF.HEADER.size == 128 and len(img) == lib.atoma_kv_blocks_packed_size(L, hk, d, page, 3, 1)
    buf = np.frombuffer(img, np.uint8).copy()
    hdr = (C.c_uint8 * 128)()
    assert lib.atoma_kv_read_header(buf.ctypes.data, len(img), hdr) == 0, lib.atoma_last_error()
    assert bytes(hdr) == img[:128]
    buf[200] ^= 0x10
    assert lib.atoma_kv_read_header(buf.ctypes.data, len(img), hdr) == -1 and b"checksum" in lib.atoma_last_error()
    buf[200] ^= 0x10
    # a crafted header: every field that a reader would dereference is either covered by the checksum or re-derived from the
    # geometry -- flipping page_bytes / payload_offset / a geometry field (with or without a recomputed checksum) is refused
    fields = list(KF.HEADER.unpack(img[:128]))
    for idx, val in ((8, fields[8] * 2), (9, fields[9] + 256), (9, 128 + 8 * 3), (3, fields[3] + 1), (7, (1 << 62))):
        f = list(fields)
        f[idx] = val
        for fix_checksum in (False, True):
            f[11] = 0
            if fix_checksum:
                f[11] = KF.checksum(KF.HEADER.pack(*f), img[128:])
            else:
                f[11] = fields[11]
            forged = np.frombuffer(KF.HEADER.pack(*f) + img[128:], np.uint8).copy()
            assert lib.atoma_kv_read_header(forged.ctypes.data, len(forged), hdr) == -1, (idx, val, fix_checksum)
    # geometry whose byte count leaves int64 is refused, not wrapped
    assert lib.atoma_kv_blocks_packed_size(1 << 31, 1 << 31, 1 << 31, 16, 4, 1) == -1
    assert fields[9] % 256 == 0                      # payload_offset: every page 16-byte aligned for the vector copy path
    kc2, vc2 = [np.zeros_like(a) for a in kc], [np.zeros_like(a) for a in vc]
    ids, _, _ = KF.unpack(img, kc2, vc2, [0, 5, 2])
    assert ids.tolist() == [4, 1, 3] and np.array_equal(kc2[1][5], kc[1][1]) and np.array_equal(vc2[0][2], vc[0][3])
