"""bf16 / f16 <-> f32 helpers on numpy arrays (test infrastructure, see oracle/__init__.py).

bf16 values travel as ``uint16`` bit patterns (numpy has no bfloat16); f16 as
``numpy.float16``.  Rounding is round-to-nearest-even, which is what the
reference's CUDA ``__float2bfloat16_rn`` / cutlass ``NumericArrayConverter``
do (csrc/kernels/utils.h:240-262).
"""
import numpy as np

F16 = 0   # dtype codes of the C ABI (csrc/src/cache_manager.rs:384-390: 0 => f16, 1 => bf16)
BF16 = 1


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    nan = np.isnan(x)
    rounding = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + rounding) >> 16).astype(np.uint16)
    if nan.any():
        out = np.where(nan, np.uint16(0x7FC0), out)
    return out


def bf16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def to_f32(a, dtype):
    """Array in storage form (f16 ndarray or bf16 bit ndarray) -> float32."""
    if dtype == F16:
        return np.asarray(a).view(np.float16).astype(np.float32) if np.asarray(a).dtype == np.uint16 \
            else np.asarray(a, dtype=np.float16).astype(np.float32)
    return bf16_bits_to_f32(a)


def from_f32(x, dtype):
    """float32 -> storage form as uint16 bit patterns (both dtypes)."""
    if dtype == F16:
        return np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)
    return f32_to_bf16_bits(x)


def round_through(x, dtype):
    """Round a float32 array to the storage dtype and come back to float32."""
    return to_f32(from_f32(x, dtype), dtype)
