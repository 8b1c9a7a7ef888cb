#!/usr/bin/env python3
"""C4: the decode step of a tensor-parallel Llama-3.1-70B-shaped job, one rank per GPU (bench plumbing over the C ABI).

SURVEY.md 8d C4 / BASELINE.json configs[3]: hidden 8192, 64 q / 8 kv heads, d = 128, intermediate 28672, 80 layers,
vocabulary 128256, synthetic bf16 weights; batch 64, context 4096 (+ the decode position).  Every rank holds 1/N of the
heads, of the KV cache and of the MLP (tp.shard_config: llama_nccl.rs:153-171), runs the same DecodeStep on identical
metadata and completes the two row-parallel projections of every layer with a sum all-reduce of [batch, 8192] bf16 = 1 MiB
(llama_nccl.rs:139,195; multi_gpu.rs:141-179): 160 all-reduces per step.  The step is captured in a hipGraph and replayed.

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/tp_step.py [--steps 20]
    python tools/tp_step.py                      # N = 1: the whole model on one GPU (140 GB of weights), no exchange
    python tools/tp_step.py --virtual-ranks 2 --batch 16   # EXPERIMENTAL: 2 ranks of a TP=8 job on ONE device (direct all-reduce only)
        Two ranks sharing one GPU is a functional check, not a benchmark: a rank's all-reduce spins on the device until its peer
        arrives, and the peer's kernels must become resident beside it.  Small shapes do (tests/test_tp_step_gpu.py, batch <= 16
        here); at batch 64 the peer's vendor GEMMs (whole-chip grids) and the spinning blocks starve each other and the run
        stalls (observed; each rank on its own GPU cannot get there: a stream is in order, nothing of the rank runs beside its
        own all-reduce).

Measured per engine (RCCL's ncclAllReduce, the direct xGMI kernels): ms per step (max over ranks), and the all-reduce alone
(graph of 160 back-to-back calls).  Prints one JSON line on rank 0; `run()` returns the same dict for bench.py.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)
import atoma_hip as ah  # noqa: E402
import decode_step as DS  # noqa: E402
import tp  # noqa: E402
from halfs import BF16, from_f32  # noqa: E402

LLAMA_3_1_70B = DS.Config(80, 8192, 64, 8, 128, 28672, 128256)


def rand_dev(rng, nbytes, slab_bytes=32 << 20):
    """nbytes of bf16 N(0,1) on the device: one random slab uploaded once, then doubled device-to-device."""
    buf = ah.DeviceBuffer(nbytes)
    n0 = min(nbytes, slab_bytes)
    slab = from_f32(rng.standard_normal(n0 // 2, dtype=np.float32), BF16)
    ah.hip_check(ah.hip.hipMemcpy(buf.ptr, slab.ctypes.data, n0, ah.H2D), "upload slab")
    done = n0
    while done < nbytes:
        n = min(done, nbytes - done)
        ah.hip_check(ah.hip.hipMemcpy(buf.ptr + done, buf.ptr, n, ah.D2D), "tile slab")
        done += n
    return buf


def random_shard_weights(rng, c):
    """Device-resident synthetic weights with the shapes of `c` (a whole model or one rank's shard)."""
    cos, sin = DS.rope_tables(c)
    w = dict(emb=rand_dev(rng, c.vocab * c.hidden * 2), lm_head=rand_dev(rng, c.vocab * c.hidden * 2), norm_f=rand_dev(rng, c.hidden * 2),
             cos=ah.DeviceBuffer.from_numpy(cos), sin=ah.DeviceBuffer.from_numpy(sin),
             norm1=[rand_dev(rng, c.hidden * 2) for _ in range(c.layers)], norm2=[rand_dev(rng, c.hidden * 2) for _ in range(c.layers)],
             wqkv=[], wo=[], wgu=[], wdown=[])
    for _ in range(c.layers):
        w["wqkv"].append(rand_dev(rng, c.qkv * c.hidden * 2))
        w["wo"].append(rand_dev(rng, c.hidden * c.h * c.d * 2))
        w["wgu"].append(rand_dev(rng, 2 * c.inter * c.hidden * 2))
        w["wdown"].append(rand_dev(rng, c.hidden * c.inter * 2))
    return w


def step_bytes(c, B, ctx):
    """HBM bytes one rank's step must move: its weights once (embedding: B rows) + its KV cache of the batch."""
    w = 2 * (c.vocab * c.hidden + c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden)) + 2 * B * c.hidden
    return w + 2 * B * (ctx + 1) * c.hk * c.d * 2 * c.layers


def timed(stream, fn, iters, warm_ms=60.0):
    import time
    t0, n = time.perf_counter(), 0
    while n < 1 or (time.perf_counter() - t0) * 1e3 < warm_ms:     # at least warm_ms of device time: sustained clocks (tools/probes/warm_probe.py)
        fn()
        stream.synchronize()
        n += 1
    a, b = ah.Event(), ah.Event()
    a.record(stream.s)
    for _ in range(iters):
        fn()
    b.record(stream.s)
    b.synchronize()
    return a.elapsed_ms(b) / iters


class Rank:
    """One tensor-parallel rank: its shard, its DecodeStep, its stream and a switchable all-reduce engine."""

    def __init__(self, full_cfg, rank, world, B, ctx, device, seed=3):
        ah.set_device(device)
        self.rank, self.world, self.B = rank, world, B
        self.c = tp.shard_config(full_cfg, world)
        self.stream = ah.Stream()
        self.engine = None                                   # callable(ptr, count) or None
        rng = np.random.default_rng(seed + rank)
        self.w = random_shard_weights(rng, self.c)
        pps = (ctx + 1 + self.c.page - 1) // self.c.page
        meta = np.random.default_rng(seed)                   # identical metadata on every rank
        self.step = DS.DecodeStep(self.c, B, B * pps + 2, pps, self.w, self.stream, fused_epilogues=True,
                                  allreduce=(lambda ptr, count: self.engine(ptr, count)) if world > 1 else None)
        bt = meta.permutation(B * pps).astype(np.int32).reshape(B, pps)
        pos = np.full(B, ctx)
        slots = bt[np.arange(B), pos // self.c.page].astype(np.int64) * self.c.page + pos % self.c.page
        self.step.set_inputs(meta.integers(0, self.c.vocab, B), pos, slots, pos + 1, bt)
        self.ar_buf = ah.DeviceBuffer.zeros((B, self.c.hidden), np.uint16)

    def graph_of(self, fn):
        fn()                                                 # eager once: scratch, vendor-GEMM plans
        self.stream.synchronize()
        with ah.Graph.capture(self.stream) as g:
            fn()
        return g


def measure(ranks, engines, steps, barrier, reduce_max, progress=None):
    """ranks: the Rank objects THIS process drives (1 under torch.distributed.run, W with --virtual-ranks).
    engines: name -> list (one per local rank) of callables(ptr, count, stream) or None when unavailable."""
    out = {}
    n_ar = 2 * ranks[0].c.layers
    # One step with the exchange switched off: creates every stream's scratch and lets the vendor GEMM behind the
    # batch-64 projections time its algorithms.  That tuning synchronises the host with the stream -- with several ranks driven
    # from ONE host thread (virtual ranks) it must not happen while a rank's all-reduce waits on the device for a peer
    # whose kernels this thread has not enqueued yet.
    for rk in ranks:
        rk.engine = lambda ptr, count: None
        rk.step.run()
        rk.stream.synchronize()
    for name, fns in engines.items():
        if fns is None:
            out[name] = None
            continue
        for rk, fn in zip(ranks, fns):
            rk.engine = (lambda ptr, count, fn=fn, rk=rk: fn(ptr, count, rk.stream.s))
        res = {}
        # ---- the whole step ----
        graphs = []
        if len(ranks) == 1:
            graphs = [ranks[0].graph_of(ranks[0].step.run)]
        else:                                                # virtual ranks: every eager / capture phase for all ranks before syncing
            for rk in ranks:
                rk.step.run()
            for rk in ranks:
                rk.stream.synchronize()
            for rk in ranks:
                with ah.Graph.capture(rk.stream) as g:
                    rk.step.run()
                graphs.append(g)
        barrier()

        def replay_all(gs):
            for g in gs:
                g.launch()
        for _ in range(2):
            replay_all(graphs)
        for rk in ranks:
            rk.stream.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            replay_all(graphs)
        for rk in ranks:
            rk.stream.synchronize()
        barrier()
        res["step_ms"] = reduce_max((time.perf_counter() - t0) * 1e3 / steps)
        # ---- the all-reduce alone: one graph of 2 x layers calls on the [B, hidden] message ----
        if ranks[0].world > 1:
            def only_ar(rk):
                for _ in range(n_ar):
                    rk.engine(rk.ar_buf.ptr, rk.B * rk.c.hidden)
            gs = []
            for rk in ranks:
                only_ar(rk)
            for rk in ranks:
                rk.stream.synchronize()
            for rk in ranks:
                with ah.Graph.capture(rk.stream) as g:
                    only_ar(rk)
                gs.append(g)
            barrier()
            replay_all(gs)
            for rk in ranks:
                rk.stream.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                replay_all(gs)
            for rk in ranks:
                rk.stream.synchronize()
            barrier()
            res["allreduce_us"] = reduce_max((time.perf_counter() - t0) * 1e6 / (5 * n_ar))
            res["allreduce_share_of_step"] = round(res["allreduce_us"] * n_ar / 1e3 / res["step_ms"], 3)
        out[name] = {k: round(v, 4) if isinstance(v, float) else v for k, v in res.items()}
        if progress is not None:
            progress.setdefault("engines", {})[name] = out[name]
    return out


def run(full_cfg=LLAMA_3_1_70B, B=64, ctx=4096, steps=20, dist=None, rank=0, world=1, local_rank=0, virtual_ranks=0, comm=None, xgmi=None, progress=None):
    """dist: an initialised torch.distributed (gloo) module when world > 1, used only for barriers and max-over-ranks.  comm: an
    atoma_comm (RCCL + the direct path behind it); xgmi: a direct-only communicator instead (tp.xgmi_comm); progress: a dict that
    receives every engine's numbers as soon as they exist (bench.py's watchdog prints it if the run does not come back)."""
    barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)

    def reduce_max(x):
        if dist is None:
            return float(x)
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    engines, xs, own_comm = {}, [], None
    if virtual_ranks:
        tp_world = 8                                          # shapes of a TP=8 rank; only `virtual_ranks` of them exist
        scfg = tp.shard_config(full_cfg, tp_world)
        msg = B * full_cfg.hidden * 2
        ranks = [Rank.__new__(Rank) for _ in range(virtual_ranks)]
        for r, rk in enumerate(ranks):                        # a virtual rank is a TP=8 shard whose communicator has `virtual_ranks` members
            Rank.__init__(rk, full_cfg, r, tp_world, B, ctx, 0)
            rk.world = virtual_ranks
        os.environ["ATOMA_XGMI_ONESHOT_MAX"] = str(max(msg, 1 << 20))   # so that both kernels can be forced on the 1 MiB message
        for r in range(virtual_ranks):
            h = C.c_void_p()
            assert ah.lib.atoma_xgmi_create(C.byref(h), r, virtual_ranks, 0, max(msg, 1 << 20)) == 0, ah.last_error()
            xs.append(h)
        blobs = (C.c_uint8 * (128 * virtual_ranks))()
        for r in range(virtual_ranks):
            one = (C.c_uint8 * 128)()
            assert ah.lib.atoma_xgmi_handle(xs[r], one) == 0, ah.last_error()
            C.memmove(C.addressof(blobs) + 128 * r, one, 128)
        for r in range(virtual_ranks):
            assert ah.lib.atoma_xgmi_connect(xs[r], blobs) == 0, ah.last_error()

        def mk(mode):
            def make(r):
                def f(ptr, count, s):
                    assert ah.lib.atoma_xgmi_allreduce_sum_mode(xs[r], ptr, ptr, count, BF16, mode, s) == 0, ah.last_error()
                return f
            return [make(r) for r in range(virtual_ranks)]
        engines = {"xgmi_one_shot": mk(1), "xgmi_two_shot": mk(2)}
        c = scfg
    else:
        ranks = [Rank(full_cfg, rank, world, B, ctx, local_rank)]
        c = ranks[0].c
        if world > 1 and xgmi is not None:                   # direct kernels only (no RCCL communicator)
            def via_xgmi(ptr, count, s):
                assert ah.lib.atoma_xgmi_allreduce_sum(xgmi, ptr, ptr, count, BF16, s) == 0, ah.last_error()
            engines = {"rccl": None, "xgmi": [via_xgmi]}
            info = "xgmi: ready (direct-only communicator, no RCCL)"
        elif world > 1:
            if comm is None:
                comm = own_comm = tp.rccl_comm(ah, dist, rank, world, local_rank)
            ah.lib.atoma_comm_set_mode(comm, 2)              # builds the direct path behind the communicator (collective: every rank alike)
            info = ah.lib.atoma_comm_info(comm).decode()
            ah.lib.atoma_comm_set_mode(comm, 0)

            def via_comm(mode):
                def f(ptr, count, s):
                    assert ah.lib.atoma_comm_set_mode(comm, mode) == 0, ah.last_error()
                    assert ah.lib.atoma_allreduce_sum(comm, ptr, ptr, count, BF16, s) == 0, ah.last_error()
                return [f]
            engines = {"rccl": via_comm(0), "xgmi": via_comm(1) if "ready" in info else None}
            if engines["xgmi"]:
                # preflight: one direct all-reduce of a known pattern; every rank must agree that it worked before it is timed
                probe = ah.DeviceBuffer.from_numpy(np.full(4096, from_f32(np.float32([rank + 1]), BF16)[0], np.uint16))
                ok = 1.0
                try:
                    engines["xgmi"][0](probe.ptr, 4096, None)
                    ah.synchronize()
                    want = from_f32(np.float32([world * (world + 1) / 2]), BF16)[0]
                    ok = 1.0 if (probe.numpy(np.uint16, (4096,)) == want).all() else 0.0
                except Exception as e:       # a timed-out wait surfaces as an error of the next call
                    ok, info = 0.0, info + f"; preflight failed: {e}"
                import torch
                t = torch.tensor([ok], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                if t.item() < 1.0:
                    engines["xgmi"] = None
                    info += "; preflight: wrong sum or timeout on some rank -- not measured"
        else:
            engines = {"none": [lambda ptr, count, s: None]}
            info = "single rank"
    if progress is not None:
        progress.update(workload="Llama-3.1-70B-shaped decode step, TP=%d" % world, world=world, xgmi_setup=info if world > 1 else None)
    res = measure(ranks, engines, steps, barrier, reduce_max, progress)
    nbytes = step_bytes(c, B, ctx)
    out = {"workload": f"Llama-3.1-70B-shaped decode step (SURVEY C4), {full_cfg.layers} layers, batch {B}, context {ctx}, bf16, "
                       + (f"{virtual_ranks} ranks of a TP=8 job on ONE device (direct all-reduce only)" if virtual_ranks else f"TP={world}, one rank per GPU"),
           "world": virtual_ranks or world, "allreduce_message_bytes": B * full_cfg.hidden * 2, "allreduces_per_step": 2 * c.layers if (virtual_ranks or world) > 1 else 0,
           "rank_step_bytes": int(nbytes), "rank_roofline_ms_at_8TBps": round(nbytes / 8e12 * 1e3, 3), "engines": res}
    if not virtual_ranks and world > 1:
        out["xgmi_setup"] = info
    best = min((v["step_ms"] for v in res.values() if v), default=None)
    if best:
        out["tokens_per_s"] = round(B / (best * 1e-3), 1)
        out["step_frac_of_roofline"] = round(out["rank_roofline_ms_at_8TBps"] / best, 4)
    for x in xs:
        ah.lib.atoma_xgmi_destroy(x)
    if own_comm is not None:
        barrier()
        ah.lib.atoma_comm_destroy(own_comm)
    return out


def main():
    import faulthandler
    faulthandler.enable()
    if os.environ.get("ATOMA_TP_STEP_WATCHDOG_S"):       # debugging aid: dump every thread's Python stack and exit if the run takes longer
        faulthandler.dump_traceback_later(int(os.environ["ATOMA_TP_STEP_WATCHDOG_S"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (quick runs; say so when quoting)")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="2 or 3: that many TP=8-shaped ranks on device 0, direct all-reduce (no RCCL)")
    a = ap.parse_args()
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    cfg = LLAMA_3_1_70B if not a.layers else DS.Config(a.layers, 8192, 64, 8, 128, 28672, 128256)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ah.set_device(local_rank)
    out = run(cfg, a.batch, a.ctx, a.steps, dist, rank, world, local_rank, a.virtual_ranks)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
