"""CPU restatement of the linear layer of a decode step (test infrastructure only).

Reference: candle_nn::Linear::forward on [B, hidden] activations (models/src/llama.rs:269-271,311,364-365: q/k/v/o
and MLP projections) = x . W^T through cuBLAS with fp32 accumulation and one rounding to the tensor dtype.  Candle /
cuBLAS are third-party (candle 0.9.2-alpha.1, Cargo.lock) and absent from /root/reference, and no reference test holds a
value for it: PARITY UNPINNED.  The oracle is the exactly accumulated product (f64) rounded once to the storage dtype;
an fp32-accumulating kernel may differ from it by one unit in the last place.
"""
import numpy as np

from .halfs import from_f32, to_f32


def linear(x, w, dtype):
    """x ``[B, K]``, w ``[N, K]`` storage-form (uint16) -> y ``[B, N]`` storage-form."""
    y = to_f32(x, dtype).astype(np.float64) @ to_f32(w, dtype).astype(np.float64).T
    return from_f32(y.astype(np.float32), dtype)
