"""Tensor-parallel plumbing shared by bench.py and the multi-process tests.

The reference shards attention Megatron-style (models/src/llama_nccl.rs:153-171, worker.rs:584-591):
rank r owns q heads [r*h/N, (r+1)*h/N) and kv heads [r*h_k/N, ...), keeps its own KV cache
[2, nb, page, h_k/N, d], sees the SAME block tables / slot mappings as every other rank, and the only
exchange is the sum all-reduce of the row-parallel o_proj output (models/src/multi_gpu.rs:48-50,141-179).
The communicator is bootstrapped like model_executor.rs:413,436-439: one unique id made on rank 0 and
handed to every rank out of band (here: a torch.distributed broadcast over gloo/TCP).
"""
import ctypes as C


def head_shard(h, h_k, rank, world):
    """(q-head slice, kv-head slice) of `rank`; h_k % world == 0 (llama_nccl.rs:153-171)."""
    if h_k % world or h % world:
        raise ValueError(f"kv heads {h_k} / q heads {h} must divide over {world} ranks")
    return slice(rank * h // world, (rank + 1) * h // world), slice(rank * h_k // world, (rank + 1) * h_k // world)


def broadcast_unique_id(dist, make_id, rank):
    """rank 0 calls make_id() -> 128 bytes; everybody returns the same 128 bytes."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.tensor(list(make_id()), dtype=torch.uint8)
    dist.broadcast(buf, 0)
    return bytes(buf.tolist())


def rccl_comm(ah, dist, rank, world, device):
    """An atoma_comm over RCCL for this rank (one process per GPU)."""
    def make_id():
        raw = (C.c_uint8 * 128)()
        if ah.lib.atoma_comm_unique_id(raw) != 0:
            raise RuntimeError(ah.last_error())
        return bytes(raw)
    uid = broadcast_unique_id(dist, make_id, rank)
    raw = (C.c_uint8 * 128)(*uid)
    comm = C.c_void_p()
    if ah.lib.atoma_comm_init(C.byref(comm), rank, world, raw, device) != 0:
        raise RuntimeError(ah.last_error())
    return comm
