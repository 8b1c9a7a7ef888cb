"""OCP fp8 e4m3fn and the fp8 KV cache, restated in numpy (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference has no quantised KV cache (README.md:35 lists "quantization" on its roadmap); what is restated is (a) the
published OCP 8-bit floating point format E4M3 in its finite-only "fn" variant -- 1 sign, 4 exponent (bias 7), 3 mantissa
bits, no infinities, S.1111.111 = NaN, largest finite 448, subnormals m/8 * 2^-6 -- with round-to-nearest-even and
SATURATION to +-448, and (b) the cache-write rule of libatoma_hip (csrc/kv_fp8.hip), which keeps the reference's slot
arithmetic (cache_manager.cu:139-170) and stores byte = e4m3fn(clamp(f32(x) * (f32(1) / scale[head]), -448, 448)).
Pinned against torch.float8_e4m3fn in the build container (tests/golden/gen_fp8_torch.py -> tests/golden/fp8_torch.npz).
"""
import numpy as np

from .halfs import to_f32


def _decode_table():
    t = np.zeros(256, np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m / 8.0 * 2.0 ** -6
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = -v if s else v
    return t


DECODE = _decode_table()
_POS = DECODE[:127].astype(np.float64)                       # codes 0x00..0x7E: 0 .. 448, ascending
_MID = (_POS[1:] + _POS[:-1]) / 2                            # decision points between neighbouring codes


def decode(bits):
    return DECODE[np.asarray(bits, np.uint8)]


def encode(x):
    """float32 array -> e4m3fn bytes: round to nearest, ties to the even code, saturate to +-448, NaN -> 0x7F."""
    x = np.asarray(x, np.float32)
    a = np.minimum(np.abs(x).astype(np.float64), 448.0)
    code = np.searchsorted(_MID, a, side="left")              # a == midpoint -> lower code; fixed below for ties
    tie = (code < 126) & (a == _MID[np.minimum(code, 125)])
    code = np.where(tie & (code % 2 == 1), code + 1, code)    # ties go to the code with an even mantissa bit
    out = code.astype(np.uint8) | (np.signbit(x).astype(np.uint8) << 7)
    return np.where(np.isnan(x), np.uint8(0x7F), out).astype(np.uint8)


def quantize(x_bits, dtype, scale_per_head):
    """x ``[T, heads, d]`` storage-form (f16 / bf16 bits) -> bytes, with the kernel's arithmetic: f32(x) * (f32(1) / scale)."""
    inv = (np.float32(1) / np.asarray(scale_per_head, np.float32)).astype(np.float32)
    return encode((to_f32(x_bits, dtype) * inv[None, :, None]).astype(np.float32))


def reshape_and_cache_flash_fp8(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale, dtype):
    """key/value ``[T, hk, d]`` storage-form; caches ``[nb, page, hk, d]`` uint8, modified in place; slot < 0 = padding."""
    page = key_cache.shape[1]
    kq, vq = quantize(key, dtype, k_scale), quantize(value, dtype, v_scale)
    for t, s in enumerate(np.asarray(slot_mapping)):
        if s < 0:
            continue
        key_cache[s // page, s % page] = kq[t]
        value_cache[s // page, s % page] = vq[t]


def dequantize(cache_u8, scale_per_head):
    """``[nb, page, hk, d]`` bytes -> float32 values e4m3 * scale[head]."""
    return (decode(cache_u8) * np.asarray(scale_per_head, np.float32)[None, None, :, None]).astype(np.float32)
