"""The KV block container of libatoma_hip (csrc/kv_format.hip) restated in numpy (TEST INFRASTRUCTURE).

Not a reference format -- the reference swaps raw pages inside one process (csrc/src/cache_manager.rs:18-128) -- so this
oracle pins OUR format: byte layout, checksum, and what pack / unpack must move."""
import struct

import numpy as np

MAGIC = b"ATOMAKV1"
HEADER = struct.Struct("<8s6I5Q56s")          # magic, version dtype layers h_k d block_size, num_blocks page_bytes payload_offset total checksum, reserved
U8 = 5
ELT = {0: 2, 1: 2, U8: 1}
M64 = (1 << 64) - 1
K = 0x9E3779B97F4A7C15


def checksum(header_zeroed, body):
    """64-bit multiply-mix over the header (checksum field zeroed) followed by the body."""
    b = bytes(header_zeroed) + bytes(body)
    h = (K ^ len(b)) & M64
    n8 = len(b) // 8
    for w in np.frombuffer(b[:n8 * 8], "<u8"):
        h = ((h ^ int(w)) * K) & M64
        h ^= h >> 32
    tail = int.from_bytes(b[n8 * 8:], "little")
    h = ((h ^ tail) * K) & M64
    h ^= h >> 29
    return h


def pack(k_caches, v_caches, block_ids, dtype, k_scales=None, v_scales=None):
    """caches: per-layer numpy arrays [nb, page, hk, d] (uint16 bits or uint8); returns the image as bytes."""
    L = len(k_caches)
    nb, page, hk, d = k_caches[0].shape
    ids = np.asarray(block_ids, np.int64)
    body = ids.tobytes()
    if dtype == U8:
        body += np.asarray(k_scales, np.float32).tobytes() + np.asarray(v_scales, np.float32).tobytes()
    body += b"\0" * (-(HEADER.size + len(body)) % 256)          # the payload starts on a multiple of 256 bytes
    payload_offset = HEADER.size + len(body)
    for l in range(L):
        body += k_caches[l][ids].tobytes() + v_caches[l][ids].tobytes()
    page_bytes = page * hk * d * ELT[dtype]
    total = HEADER.size + len(body)
    fields = [MAGIC, 2, dtype, L, hk, d, page, len(ids), page_bytes, payload_offset, total, 0, b"\0" * 56]
    fields[11] = checksum(HEADER.pack(*fields), body)
    return HEADER.pack(*fields) + body


def unpack(image, k_caches, v_caches, dst_ids):
    """Scatter an image into caches (in place); returns (block ids, k_scales, v_scales)."""
    magic, ver, dtype, L, hk, d, page, n, page_bytes, off, total, cs, _ = HEADER.unpack(image[:HEADER.size])
    zeroed = HEADER.pack(magic, ver, dtype, L, hk, d, page, n, page_bytes, off, total, 0, _)
    assert magic == MAGIC and ver == 2 and total == len(image) and checksum(zeroed, image[HEADER.size:]) == cs
    assert page_bytes == page * hk * d * ELT[dtype] and off == -(-(HEADER.size + 8 * n + (8 * L * hk if dtype == U8 else 0)) // 256) * 256
    ids = np.frombuffer(image, np.int64, n, HEADER.size)
    ks = vs = None
    if dtype == U8:
        ks = np.frombuffer(image, np.float32, L * hk, HEADER.size + 8 * n).reshape(L, hk)
        vs = np.frombuffer(image, np.float32, L * hk, HEADER.size + 8 * n + 4 * L * hk).reshape(L, hk)
    dt = np.uint8 if dtype == U8 else np.uint16
    per = n * page * hk * d
    pay = np.frombuffer(image, dt, 2 * L * per, off).reshape(L, 2, n, page, hk, d)
    for l in range(L):
        k_caches[l][np.asarray(dst_ids)] = pay[l, 0]
        v_caches[l][np.asarray(dst_ids)] = pay[l, 1]
    return ids, ks, vs
