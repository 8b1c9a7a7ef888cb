"""f32 <-> bf16 / f16 storage bits for the bench and tool drivers (numpy has no bfloat16).

Product-side plumbing: bench.py and tools/ build their synthetic tensors with this, so that nothing outside tests/ and
the cpu_baseline leg touches oracle/.  Round-to-nearest-even, NaN -> quiet NaN, like the device's v_cvt_pk_bf16_f32."""
import numpy as np

F16, BF16 = 0, 1


def from_f32(x, dtype):
    """float32 array -> uint16 storage bits."""
    x = np.ascontiguousarray(x, np.float32)
    if dtype == F16:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32)
    bits = ((u + (np.uint32(0x7FFF) + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    return np.where(np.isnan(x), np.uint16(0x7FC0), bits)


def to_f32(bits, dtype):
    """uint16 storage bits -> float32."""
    b = np.ascontiguousarray(bits, np.uint16)
    if dtype == F16:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)
