"""CPU restatement of the element-wise ops between the kernels of a Llama decode step (test infrastructure only).

embedding = index_select of the table rows (models/src/llama.rs:456-458); residual add (llama.rs:404,409) and
SiLU(gate) * up (llama.rs:364-365) follow Candle's op-by-op evaluation: each op computes in f32 and rounds its result to
the tensor dtype (silu and the product are two ops: two roundings).  Candle is not in /root/reference and no reference
test holds a value for these ops: PARITY UNPINNED.
"""
import numpy as np

from .halfs import from_f32, round_through, to_f32


def embedding(ids, table):
    return table[np.asarray(ids)]


def add(a, b, dtype):
    return from_f32(to_f32(a, dtype) + to_f32(b, dtype), dtype)


def silu_mul(gate, up, dtype):
    g = to_f32(gate, dtype).astype(np.float64)
    s = round_through((g / (1.0 + np.exp(-g))).astype(np.float32), dtype)
    return from_f32(s * to_f32(up, dtype), dtype)
