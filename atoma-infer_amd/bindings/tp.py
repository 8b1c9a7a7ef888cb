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


def xgmi_comm(ah, dist, rank, world, device, capacity_bytes):
    """A direct-only communicator (atoma_xgmi_*: DESIGN.md 4.5) without RCCL: every rank creates its staging region, the 128-byte
    handles travel over the rendezvous (gloo all-gather), every rank maps its peers.  What a host that does not want RCCL does
    (INTEGRATION.md), and what lets two ranks share ONE device in the plumbing tests (RCCL refuses that)."""
    import torch
    h = C.c_void_p()
    if ah.lib.atoma_xgmi_create(C.byref(h), rank, world, device, capacity_bytes) != 0:
        raise RuntimeError(ah.last_error())
    one = (C.c_uint8 * 128)()
    if ah.lib.atoma_xgmi_handle(h, one) != 0:
        raise RuntimeError(ah.last_error())
    mine = torch.tensor(list(bytes(one)), dtype=torch.uint8)
    allh = [torch.zeros(128, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(allh, mine)
    blobs = (C.c_uint8 * (128 * world))(*[int(v) for t in allh for v in t.tolist()])
    if ah.lib.atoma_xgmi_connect(h, blobs) != 0:
        raise RuntimeError(ah.last_error())
    dist.barrier()
    return h


def shard_config(cfg, world):
    """The model dimensions ONE rank of a `world`-way tensor-parallel job works with (llama_nccl.rs:165-166: heads and kv
    heads divided by the world size; the MLP width is split by TensorParallelColumnLinear / RowLinear).  hidden, vocab,
    layers stay whole: embedding, norms and lm_head are replicated (llama_nccl.rs:270,321)."""
    if cfg.h % world or cfg.hk % world or cfg.inter % world:
        raise ValueError(f"heads {cfg.h} / kv heads {cfg.hk} / intermediate {cfg.inter} must divide over {world} ranks")
    return type(cfg)(cfg.layers, cfg.hidden, cfg.h // world, cfg.hk // world, cfg.d, cfg.inter // world, cfg.vocab, cfg.page, cfg.eps,
                     cfg.theta, cfg.max_pos)


def shard_weights(host, cfg, rank, world):
    """This rank's slice of a full set of host weights (numpy uint16, the layout tools/decode_step.py uses:
    wqkv [(h + 2 h_k) d, H] = q rows, k rows, v rows; wo [H, h d]; wgu [2 inter, H] = gate rows, up rows; wdown [H, inter]).
    Column-parallel layers (q, k, v, gate, up: multi_gpu.rs:20-35 `shard(0, rank, size)`) keep a block of OUTPUT rows,
    row-parallel layers (o, down: multi_gpu.rs:52-57 `shard(1, ...)`) a block of INPUT columns; their outputs are partial
    sums that the all-reduce completes."""
    import numpy as np
    c, d = cfg, cfg.d
    hq, hk, it = c.h // world, c.hk // world, c.inter // world
    out = {k: v for k, v in host.items() if not isinstance(v, list)}       # emb, norm_f, lm_head: replicated
    out["norm1"], out["norm2"] = list(host["norm1"]), list(host["norm2"])
    out["wqkv"], out["wo"], out["wgu"], out["wdown"] = [], [], [], []
    for l in range(c.layers):
        wqkv = host["wqkv"][l].reshape((c.h + 2 * c.hk) * d, c.hidden)
        q, k, v = wqkv[:c.h * d], wqkv[c.h * d:(c.h + c.hk) * d], wqkv[(c.h + c.hk) * d:]
        out["wqkv"].append(np.ascontiguousarray(np.concatenate([q[rank * hq * d:(rank + 1) * hq * d], k[rank * hk * d:(rank + 1) * hk * d],
                                                                v[rank * hk * d:(rank + 1) * hk * d]])))
        out["wo"].append(np.ascontiguousarray(host["wo"][l].reshape(c.hidden, c.h * d)[:, rank * hq * d:(rank + 1) * hq * d]))
        wgu = host["wgu"][l].reshape(2 * c.inter, c.hidden)
        out["wgu"].append(np.ascontiguousarray(np.concatenate([wgu[rank * it:(rank + 1) * it], wgu[c.inter + rank * it:c.inter + (rank + 1) * it]])))
        out["wdown"].append(np.ascontiguousarray(host["wdown"][l].reshape(c.hidden, c.inter)[:, rank * it:(rank + 1) * it]))
    return out
