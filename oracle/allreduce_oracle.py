"""numpy restatement of the tensor-parallel sum all-reduce (TEST INFRASTRUCTURE, see oracle/__init__.py).

Reference: models/src/multi_gpu.rs:141-179 (`AllReduce::cuda_fwd` -> cudarc `ncclAllReduce(Sum)`, out-of-place, bf16 / f16),
call sites models/src/llama_nccl.rs:139,195 (after o_proj and down_proj).  PARITY UNPINNED in the last bit: NCCL (cudarc
0.17.3, Cargo.lock:833-835) is not in the tree, the reference has no numeric TP test, and NCCL's own summation order
depends on its algorithm (ring / tree) and rank count.  What every implementation guarantees, and what is restated here:

* ``exact``      the f64 sum of the ranks' values rounded once to the storage dtype -- any fp32-accumulating all-reduce is
                 within one unit in the last place of it;
* ``rank_order`` fp32 accumulation in rank order 0..W-1 then one rounding: the contract of libatoma_hip's direct xGMI
                 kernels (bit-exact target), identical on every rank.
"""
import numpy as np

from .halfs import to_f32, from_f32

F32 = 2


def _as_f32(a, dtype):
    return np.asarray(a, np.float32) if dtype == F32 else to_f32(a, dtype)


def _store(x, dtype):
    return np.asarray(x, np.float32) if dtype == F32 else from_f32(np.asarray(x, np.float32), dtype)


def allreduce_sum(parts, dtype, mode="rank_order"):
    """parts: one storage-form array per rank (uint16 bits for f16 / bf16, float32 for dtype 2), same shape."""
    if mode == "rank_order":
        acc = _as_f32(parts[0], dtype).copy()
        for p in parts[1:]:
            acc = (acc + _as_f32(p, dtype)).astype(np.float32)
        return _store(acc, dtype)
    if mode == "exact":
        acc = sum(_as_f32(p, dtype).astype(np.float64) for p in parts)
        return _store(acc.astype(np.float32), dtype)
    raise ValueError(mode)


def column_shard(w, rank, world):
    """TensorParallelColumnLinear (multi_gpu.rs:20-24, `shard(0, rank, size)`): rows [rank*N/W, (rank+1)*N/W) of w [N, K]."""
    n = w.shape[0]
    assert n % world == 0
    return w[rank * n // world: (rank + 1) * n // world]


def row_shard(w, rank, world):
    """TensorParallelRowLinear (multi_gpu.rs:52-57, `shard(1, rank, size)`): columns [rank*K/W, (rank+1)*K/W) of w [N, K]."""
    k = w.shape[1]
    assert k % world == 0
    return w[:, rank * k // world: (rank + 1) * k // world]
