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


def preflight(ah, dist, rank, world, comm=None, xgmi=None, sizes=(16, 1 << 20, 64 << 20), log=None, timeout_s=60.0):
    """First contact: before ANYTHING is timed, every all-reduce engine this job can use sums a known pattern at
    16 B / 1 MiB / 64 MiB over all ranks; each (engine, size) gets pass / fail and its time in us, agreed over the rendezvous (min over
    ranks of ok, max of time) -- a wrong sum, an error or a timeout on ANY rank fails the pair for everybody.  Engines: "rccl"
    (ncclAllReduce through the communicator) and "direct" (the xGMI kernels: through the communicator's direct mode, or a direct-only
    handle), tested separately.  A pair that would not fit the direct engine's staging region is reported as skipped.  A watchdog turns a
    hang (RCCL has no timeout of its own) into a failed pair and abandons the remaining ones: the caller's headline must not depend on them.
    rank value pattern: element i of rank r = (r + 1) * (i % 5 + 1) in bf16 -- small integers, every partial sum exact in any order."""
    import threading
    import time

    import numpy as np
    import torch
    BF16_ONE = 0x3F80

    def bf16_of(v):          # small non-negative integers as bf16 bit patterns
        return (np.asarray(v, np.float32).view(np.uint32) >> 16).astype(np.uint16)
    engines = []
    direct_state = {"built": None, "note": None}

    def direct_via_comm(i, o, n):
        # collective: every rank asks for the direct path alike (it is built on first use, over RCCL's own bootstrap) -- inside the watched
        # attempt, so that a set-up that hangs on its first real peer mapping is a failed pair, not a lost run
        if direct_state["built"] is None:
            direct_state["built"] = ah.lib.atoma_comm_set_mode(comm, 1) == 0
            direct_state["note"] = None if direct_state["built"] else ah.last_error()
            ah.lib.atoma_comm_set_mode(comm, 0)
        if not direct_state["built"]:
            raise RuntimeError("direct engine unavailable: %s" % direct_state["note"])
        ah.lib.atoma_comm_set_mode(comm, 1)
        rc = ah.lib.atoma_allreduce_sum(comm, i, o, n, 1, None)
        ah.lib.atoma_comm_set_mode(comm, 0)
        return rc
    if comm is not None:
        engines.append(("rccl", lambda i, o, n: (ah.lib.atoma_comm_set_mode(comm, 0), ah.lib.atoma_allreduce_sum(comm, i, o, n, 1, None))[1], None))
        engines.append(("direct", direct_via_comm, None))
    if xgmi is not None:
        cap = int(ah.lib.atoma_xgmi_capacity(xgmi))
        engines.append(("direct", lambda i, o, n: ah.lib.atoma_xgmi_allreduce_sum(xgmi, i, o, n, 1, None) if n * 2 <= cap else -2, None))
    res = {}
    state = {"hung": None}
    for name, fn, note in engines:
        res[name] = {}
        for nbytes in sizes:
            key = "%dB" % nbytes if nbytes < 1024 else ("%dMiB" % (nbytes >> 20))
            if state["hung"]:
                res[name][key] = {"ok": False, "us": None, "note": "not run: an earlier pair hung (%s)" % state["hung"]}
                continue
            if fn is None:
                res[name][key] = {"ok": False, "us": None, "note": "engine unavailable: %s" % note}
                continue
            n = nbytes // 2
            mine = bf16_of((rank + 1) * (np.arange(n) % 5 + 1))
            want = bf16_of(world * (world + 1) // 2 * (np.arange(n) % 5 + 1))
            src, dst = ah.DeviceBuffer.from_numpy(mine), ah.DeviceBuffer.zeros((n,), np.uint16)
            out = {"ok": 0.0, "us": 0.0, "note": None}

            def attempt():
                try:
                    rc = fn(src.ptr, dst.ptr, n)                        # first touch (connection set-up, code load), then three timed ones
                    ah.synchronize()
                    if rc == -2:
                        out["note"] = "skipped: larger than the direct engine's staging region"
                        out["ok"] = 1.0
                        return
                    if rc != 0:
                        out["note"] = "error: " + ah.last_error()
                        return
                    t0 = time.perf_counter()
                    for _ in range(3):
                        rc |= fn(src.ptr, dst.ptr, n)
                    ah.synchronize()
                    out["us"] = (time.perf_counter() - t0) / 3 * 1e6
                    good = rc == 0 and bool(np.array_equal(dst.numpy(np.uint16, (n,)), want))
                    out["ok"] = 1.0 if good else 0.0
                    if not good:
                        out["note"] = "wrong sum" if rc == 0 else "error: " + ah.last_error()
                except Exception as e:                                  # a timed-out wait of the direct kernels surfaces as an error of the next call
                    out["note"] = "exception: %r" % (e,)
            th = threading.Thread(target=attempt, daemon=True)
            th.start()
            th.join(timeout_s)
            hung_here = th.is_alive()
            if hung_here:
                out["note"] = "hung: no return within %.0f s" % timeout_s
                out["ok"] = 0.0
            # the agreement is ALWAYS taken, by every rank, once per pair (CPU tensors over the rendezvous: a hung device call does not block it):
            # a rank that skipped it would leave the others waiting in it and desynchronise every later collective of the run
            t = torch.tensor([out["ok"], -out["us"], 0.0 if hung_here else 1.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)                    # ok on every rank; the slowest rank's time; nobody hung
            if t[2].item() < 1.0:
                state["hung"] = "%s %s" % (name, key)                   # on EVERY rank alike: the remaining pairs are skipped consistently
                if not hung_here:
                    out["note"] = ((out["note"] + "; ") if out["note"] else "") + "another rank hung in this pair"
            res[name][key] = {"ok": bool(t[0].item() >= 1.0), "us": round(-t[1].item(), 1) if (out["us"] and t[0].item() >= 1.0) else None}
            if out["note"]:
                res[name][key]["note"] = out["note"]
            if log is not None and rank == 0:
                log("[preflight] %-6s all-reduce of %-6s over %d ranks: %s%s" % (name, key, world, "PASS" if res[name][key]["ok"] else "FAIL",
                                                                                ("  %.1f us" % res[name][key]["us"]) if res[name][key]["us"] else "") + (("  (" + out["note"] + ")") if out["note"] else ""))
            if not state["hung"]:
                src.free(); dst.free()
    return res
