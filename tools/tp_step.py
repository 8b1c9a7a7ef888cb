#!/usr/bin/env python3
"""C4: the decode step of a tensor-parallel Llama-3.1-70B-shaped job, one rank per GPU (bench plumbing over the C ABI).

SURVEY.md 8d C4 / BASELINE.json configs[3]: hidden 8192, 64 q / 8 kv heads, d = 128, intermediate 28672, 80 layers,
vocabulary 128256, synthetic bf16 weights; batch 64, context 4096 (+ the decode position).  Every rank holds 1/N of the
heads, of the KV cache and of the MLP (tp.shard_config: llama_nccl.rs:153-171), runs the same DecodeStep on identical
metadata and completes the two row-parallel projections of every layer with a sum all-reduce of [batch, 8192] bf16 = 1 MiB
(llama_nccl.rs:139,195; multi_gpu.rs:141-179): 160 all-reduces per step.  The step is captured in a hipGraph and replayed.

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/tp_step.py [--steps 20]
    python tools/tp_step.py                      # N = 1: the whole model on one GPU (140 GB of weights), no exchange
    python tools/tp_step.py --virtual-ranks 8               # ALL 8 ranks of the TP=8 job on ONE device (direct all-reduce only): configs[3]
        executed as 8 ranks where the hardware is one GPU -- 8 shards (8 x 17 GB of weights, 8 x 11 GB of KV cache), 8 streams, 8 staging
        regions, 160 all-reduces of 1 MiB among 8 members per step.  A functional statement (every rank ends with bit-identical
        logits), not a scaling number: a rank's all-reduce spins on the device until its seven peers arrive, their kernels share the CUs.
    python tools/tp_step.py --virtual-ranks 8 --prefill 4096   # the prefill chunk of configs[3]: 4096-token projections, causal
        attention over 8 q / 1 kv heads, two all-reduces of [4096, 8192] bf16 = 64 MiB per layer
    python tools/tp_step.py --virtual-ranks 8 --layers 8 --check-unsharded   # shards cut ON THE DEVICE from one full model; the unsharded
        step runs beside them and the logits are compared (the full 80 layers + 8 shards would need 280 GB)

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
if "--virtual-ranks" in sys.argv:
    # W ranks in ONE process = W streams whose kernels WAIT FOR EACH OTHER on the device: every stream needs a hardware queue of its own
    # (the HIP runtime multiplexes streams over 4 by default and a queue runs its packets in order: rank 0's all-reduce would wait for a
    # rank-4 kernel queued behind it -- tools/probes/world8_queues_probe.py).  Read when the runtime starts, hence before the import below.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)
import atoma_hip as ah  # noqa: E402
import decode_step as DS  # noqa: E402
import tp  # noqa: E402
from halfs import BF16, from_f32  # noqa: E402

LLAMA_3_1_70B = DS.Config(80, 8192, 64, 8, 128, 28672, 128256)


def rand_dev(rng, nbytes, scale=1.0, slab_bytes=32 << 20):
    """nbytes of bf16 N(0, scale^2) on the device: one random slab uploaded once, then doubled device-to-device."""
    buf = ah.DeviceBuffer(nbytes)
    n0 = min(nbytes, slab_bytes)
    slab = from_f32(rng.standard_normal(n0 // 2, dtype=np.float32) * np.float32(scale), BF16)
    ah.hip_check(ah.hip.hipMemcpy(buf.ptr, slab.ctypes.data, n0, ah.H2D), "upload slab")
    done = n0
    while done < nbytes:
        n = min(done, nbytes - done)
        ah.hip_check(ah.hip.hipMemcpy(buf.ptr + done, buf.ptr, n, ah.D2D), "tile slab")
        done += n
    return buf


def norm_dev(rng, n):
    """An RMSNorm weight: 1 + 0.1 N(0,1)."""
    return ah.DeviceBuffer.from_numpy(from_f32((1 + 0.1 * rng.standard_normal(n)).astype(np.float32), BF16))


def random_shard_weights(rng, c, replicated=None):
    """Device-resident synthetic weights with the shapes of `c` (a whole model or one rank's shard), scaled like an initialised model
    (projection rows ~ N(0, 1 / fan_in)) so that activations stay O(1) through 80 layers.  `replicated`: another rank's dict ON THE SAME
    DEVICE whose embedding / lm_head / final norm / RoPE tables are reused -- those tensors are replicated over the ranks of a
    tensor-parallel job (llama_nccl.rs:270,321), and eight virtual ranks on one device need not hold eight copies of 2 x 2.1 GB."""
    if replicated is not None:
        w = {k: replicated[k] for k in ("emb", "lm_head", "norm_f", "cos", "sin")}
    else:
        cos, sin = DS.rope_tables(c)
        w = dict(emb=rand_dev(rng, c.vocab * c.hidden * 2), lm_head=rand_dev(rng, c.vocab * c.hidden * 2, c.hidden ** -0.5), norm_f=norm_dev(rng, c.hidden),
                 cos=ah.DeviceBuffer.from_numpy(cos), sin=ah.DeviceBuffer.from_numpy(sin))
    w.update(norm1=[norm_dev(rng, c.hidden) for _ in range(c.layers)], norm2=[norm_dev(rng, c.hidden) for _ in range(c.layers)],
             wqkv=[], wo=[], wgu=[], wdown=[])
    for _ in range(c.layers):
        w["wqkv"].append(rand_dev(rng, c.qkv * c.hidden * 2, c.hidden ** -0.5))
        w["wo"].append(rand_dev(rng, c.hidden * c.h * c.d * 2, (c.h * c.d) ** -0.5))
        w["wgu"].append(rand_dev(rng, 2 * c.inter * c.hidden * 2, c.hidden ** -0.5))
        w["wdown"].append(rand_dev(rng, c.hidden * c.inter * 2, c.inter ** -0.5))
    return w


ah.hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]


def shard_device_weights(full, cfg, rank, world):
    """This rank's slice of a full set of DEVICE weights, cut on the device (the layout and the rule of tp.shard_weights, which does it
    on the host): column-parallel q / k / v / gate / up keep a block of output ROWS (contiguous copies), row-parallel o / down a block of
    input COLUMNS (2-D copies); norms, embedding, lm_head, RoPE tables are the full model's own buffers (replicated)."""
    c, d, H = cfg, cfg.d, cfg.hidden
    hq, hk, it = c.h // world, c.hk // world, c.inter // world
    out = {k: full[k] for k in ("emb", "lm_head", "norm_f", "cos", "sin")}
    out["norm1"], out["norm2"] = list(full["norm1"]), list(full["norm2"])
    out["wqkv"], out["wo"], out["wgu"], out["wdown"] = [], [], [], []

    def rows(dst, dst_row, src, src_row, n, width):
        ah.hip_check(ah.hip.hipMemcpy(dst.ptr + dst_row * width * 2, src.ptr + src_row * width * 2, n * width * 2, ah.D2D), "shard rows")

    def cols(src, n_rows, src_width, col0, width):
        dst = ah.DeviceBuffer(n_rows * width * 2)
        ah.hip_check(ah.hip.hipMemcpy2D(dst.ptr, width * 2, src.ptr + col0 * 2, src_width * 2, width * 2, n_rows, ah.D2D), "shard columns")
        return dst
    for l in range(c.layers):
        qkv = ah.DeviceBuffer((hq + 2 * hk) * d * H * 2)
        rows(qkv, 0, full["wqkv"][l], rank * hq * d, hq * d, H)
        rows(qkv, hq * d, full["wqkv"][l], c.h * d + rank * hk * d, hk * d, H)
        rows(qkv, (hq + hk) * d, full["wqkv"][l], (c.h + c.hk) * d + rank * hk * d, hk * d, H)
        out["wqkv"].append(qkv)
        out["wo"].append(cols(full["wo"][l], H, c.h * d, rank * hq * d, hq * d))
        gu = ah.DeviceBuffer(2 * it * H * 2)
        rows(gu, 0, full["wgu"][l], rank * it, it, H)
        rows(gu, it, full["wgu"][l], c.inter + rank * it, it, H)
        out["wgu"].append(gu)
        out["wdown"].append(cols(full["wdown"][l], H, c.inter, rank * it, it))
    return out


def step_bytes(c, B, ctx):
    """HBM bytes one rank's step must move: its weights once (embedding: B rows) + its KV cache of the batch."""
    w = 2 * (c.vocab * c.hidden + c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden)) + 2 * B * c.hidden
    return w + 2 * B * (ctx + 1) * c.hk * c.d * 2 * c.layers


def prefill_flops(c, T):
    """Floating-point operations of one rank's prefill chunk of T tokens (one prompt): the four projections of every layer, causal
    attention (half of the T x T score and P.V products), lm_head for the last token."""
    proj = 2 * T * c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden)
    attn = c.layers * 4 * c.h * c.d * T * T // 2
    return proj + attn + 2 * c.vocab * c.hidden


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
    """One tensor-parallel rank: its shard, its DecodeStep (or, with prefill = T, the PrefillStep of one T-token prompt), its stream and a
    switchable all-reduce engine.  `rows` = rows of the [rows, hidden] message its all-reduces carry."""

    def __init__(self, full_cfg, rank, world, B, ctx, device, seed=3, weights=None, replicated=None, prefill=0):
        ah.set_device(device)
        self.rank, self.world, self.B = rank, world, B
        self.c = tp.shard_config(full_cfg, world)
        self.stream = ah.Stream()
        self.engine = None                                   # callable(ptr, count) or None
        self.engine_norm = None                              # callable(in, residual, weight, x_out, norm_out, rows) -> all-reduce + add + RMSNorm in one launch, or None
        rng = np.random.default_rng(seed + rank)
        self.w = weights if weights is not None else random_shard_weights(rng, self.c, replicated)
        hook = (lambda ptr, count: self.engine(ptr, count)) if world > 1 else None
        hook_norm = (lambda *a: (self.engine_norm(*a) or True) if self.engine_norm is not None else False) if world > 1 else None
        meta = np.random.default_rng(seed)                   # identical metadata on every rank
        if prefill:
            T = prefill
            pages = (T + self.c.page - 1) // self.c.page
            self.dstep = DS.DecodeStep(self.c, 1, pages + 2, pages, self.w, self.stream, fused_epilogues=True, allreduce=hook)   # owns the KV cache the prompt is written to
            self.step = DS.PrefillStep(self.c, T, self.dstep, self.stream, prompts=1, allreduce=hook)
            bt = meta.permutation(pages).astype(np.int64)
            tok = np.arange(T)
            self.step.set_inputs(meta.integers(0, self.c.vocab, T), bt[tok // self.c.page] * self.c.page + tok % self.c.page)
            self.rows, self.logits_rows = T, 1
        else:
            pps = (ctx + 1 + self.c.page - 1) // self.c.page
            self.step = DS.DecodeStep(self.c, B, B * pps + 2, pps, self.w, self.stream, fused_epilogues=True, allreduce=hook, allreduce_norm=hook_norm)
            bt = meta.permutation(B * pps).astype(np.int32).reshape(B, pps)
            pos = np.full(B, ctx)
            slots = bt[np.arange(B), pos // self.c.page].astype(np.int64) * self.c.page + pos % self.c.page
            self.step.set_inputs(meta.integers(0, self.c.vocab, B), pos, slots, pos + 1, bt)
            self.rows, self.logits_rows = B, B
        self.ar_buf = ah.DeviceBuffer.zeros((self.rows, self.c.hidden), np.uint16)

    def logits(self):
        return self.step.logits.numpy(np.uint16, (self.logits_rows, self.c.vocab))

    def graph_of(self, fn):
        fn()                                                 # eager once: scratch, vendor-GEMM plans
        self.stream.synchronize()
        with ah.Graph.capture(self.stream) as g:
            fn()
        return g


def measure(ranks, engines, steps, barrier, reduce_max, progress=None, use_graph=True):
    """ranks: the Rank objects THIS process drives (1 under torch.distributed.run, W with --virtual-ranks).
    engines: name -> list (one per local rank) of callables(ptr, count, stream) or None when unavailable.
    use_graph False: every step is enqueued eagerly (the prefill chunk: a few hundred long kernels, nothing launch-bound)."""
    out = {}
    n_ar = 2 * ranks[0].c.layers
    # One step with the exchange switched off: creates every stream's scratch and lets the vendor GEMM behind the
    # batch-64 projections time its algorithms.  That tuning synchronises the host with the stream -- with several ranks driven
    # from ONE host thread (virtual ranks) it must not happen while a rank's all-reduce waits on the device for a peer
    # whose kernels this thread has not enqueued yet.
    for rk in ranks:
        rk.engine = lambda ptr, count: None
        rk.engine_norm = None
        rk.step.run()
        rk.stream.synchronize()
    for name, fns in engines.items():
        if fns is None:
            out[name] = None
            continue
        fused = None
        if isinstance(fns, dict):                            # an engine that also offers all-reduce + residual add + RMSNorm as one launch
            fns, fused = fns["allreduce"], fns["fused"]
        for i, (rk, fn) in enumerate(zip(ranks, fns)):
            rk.engine = (lambda ptr, count, fn=fn, rk=rk: fn(ptr, count, rk.stream.s))
            rk.engine_norm = (lambda *a, g=fused[i], rk=rk: g(*a, rk.stream.s)) if fused else None
        res = {}
        # ---- the whole step ----
        graphs = []
        if not use_graph:
            graphs = [type("Eager", (), {"launch": (lambda self, rk=rk: rk.step.run())})() for rk in ranks]
        elif len(ranks) == 1:
            graphs = [ranks[0].graph_of(ranks[0].step.run)]
        else:
            # virtual ranks: captured straight away -- the pass with the exchange switched off has created every plan and scratch block, the direct
            # all-reduce needs no warm-up, and an EAGER step of several ranks in one process is what must be avoided: past a few hundred
            # outstanding launches the runtime makes the enqueuing thread wait for its oldest command, and that command is an all-reduce waiting
            # for a peer whose launches the same process has yet to issue (round 5: the 16- and 80-layer prefill chunks timed out so, from one
            # thread and from one thread per rank alike; a graph launch is ONE call per rank and step)
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
            if os.environ.get("ATOMA_TP_STEP_VERBOSE"):      # (debugging aid: one replay at a time, with the communicators' state)
                for rk in ranks:
                    rk.stream.synchronize()
                print(f"[tp_step] {name}: warm replay done at {time.perf_counter():.3f} s, xgmi status {[int(ah.lib.atoma_xgmi_status(x)) for x in getattr(measure, 'xs', [])]}", file=sys.stderr, flush=True)
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
        # ---- the all-reduce alone: one graph of 2 x layers calls on the [rows, hidden] message ----
        if ranks[0].world > 1:
            def only_ar(rk):
                for _ in range(n_ar):
                    rk.engine(rk.ar_buf.ptr, rk.rows * rk.c.hidden)
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


def logits_agreement(ranks, dist, world):
    """After the last timed step: do all ranks hold the SAME logits, bit for bit?  (The all-reduce sums in rank order on every rank and the
    replicated part of the model runs on identical inputs, so they must.)  Virtual ranks are compared here; one process per GPU
    compares digests over the rendezvous."""
    import hashlib
    digests = [hashlib.sha1(rk.logits().tobytes()).hexdigest() for rk in ranks]
    if dist is not None and world > 1 and len(ranks) == 1:
        allv = [None] * world
        dist.all_gather_object(allv, digests[0])
        digests = allv
    return {"ranks_bit_identical": len(set(digests)) == 1, "ranks_compared": len(digests), "logits_sha1": digests[0]}


def run(full_cfg=LLAMA_3_1_70B, B=64, ctx=4096, steps=20, dist=None, rank=0, world=1, local_rank=0, virtual_ranks=0, comm=None, xgmi=None, progress=None,
        prefill=0, check_unsharded=False):
    """dist: an initialised torch.distributed (gloo) module when world > 1, used only for barriers and max-over-ranks.  comm: an
    atoma_comm (RCCL + the direct path behind it); xgmi: a direct-only communicator instead (tp.xgmi_comm); progress: a dict that
    receives every engine's numbers as soon as they exist (bench.py's watchdog prints it if the run does not come back).
    prefill = T > 0: the prefill chunk of T tokens instead of the decode step; check_unsharded (virtual ranks): the shards are cut on the
    device from ONE full model whose own step runs beside them, and the logits are compared."""
    barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)

    def reduce_max(x):
        if dist is None:
            return float(x)
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    engines, xs, own_comm, unsharded = {}, [], None, None
    rows = prefill if prefill else B
    msg = rows * full_cfg.hidden * 2
    if virtual_ranks:
        tp_world = 8                                          # shapes of a TP=8 rank; `virtual_ranks` of them exist (8 = the whole job)
        scfg = tp.shard_config(full_cfg, tp_world)
        full_w = None
        if check_unsharded:
            if virtual_ranks != tp_world:
                raise ValueError("--check-unsharded needs all 8 ranks of the TP=8 job (the partial sums of absent ranks would be missing)")
            full_w = random_shard_weights(np.random.default_rng(11), full_cfg)
        ranks = []
        for r in range(virtual_ranks):                        # a virtual rank is a TP=8 shard whose communicator has `virtual_ranks` members
            w = shard_device_weights(full_w, full_cfg, r, tp_world) if full_w is not None else None
            rk = Rank(full_cfg, r, tp_world, B, ctx, 0, weights=w, replicated=ranks[0].w if ranks else None, prefill=prefill)
            rk.world = virtual_ranks
            ranks.append(rk)
        if check_unsharded:
            unsharded = Rank(full_cfg, 0, 1, B, ctx, 0, weights=full_w, prefill=prefill)
            if not prefill:
                # a random KV history (the steps allocate zeroed caches, which is enough for timing and says nothing about attention): the
                # full model's caches [pages, page, 8 kv heads, d], every rank's cache = its kv head's columns of them (worker.rs:584-591)
                crng = np.random.default_rng(12)
                st, d, page = unsharded.step, full_cfg.d, full_cfg.page
                n_rows = st.kc[0].nbytes // (full_cfg.hk * d * 2)
                for l in range(full_cfg.layers):
                    for name in ("kc", "vc"):
                        getattr(st, name)[l] = src = rand_dev(crng, n_rows * full_cfg.hk * d * 2)
                        for r, rk in enumerate(ranks):
                            dst = ah.DeviceBuffer(n_rows * d * 2)
                            ah.hip_check(ah.hip.hipMemcpy2D(dst.ptr, d * 2, src.ptr + r * d * 2, full_cfg.hk * d * 2, d * 2, n_rows, ah.D2D), "shard the KV cache")
                            getattr(rk.step, name)[l] = dst
        cap = max(msg, 1 << 20)
        os.environ["ATOMA_XGMI_ONESHOT_MAX"] = str(1 << 20)   # so that both kernels can be forced on the 1 MiB decode message
        for r in range(virtual_ranks):
            h = C.c_void_p()
            assert ah.lib.atoma_xgmi_create(C.byref(h), r, virtual_ranks, 0, cap) == 0, ah.last_error()
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
                    assert ah.lib.atoma_xgmi_allreduce_sum_mode(xs[r], ptr, ptr, count, BF16, mode, s) == 0, \
                        ah.last_error() + " -- status of every rank (n: the wait for rank n - 1 timed out): " + str([int(ah.lib.atoma_xgmi_status(x)) for x in xs])
                return f
            return [make(r) for r in range(virtual_ranks)]
        def mk_fused(mode):
            def make(r):
                def f(inp, res, w, xo, no, rows, s):
                    assert ah.lib.atoma_xgmi_allreduce_add_rms_norm(xs[r], inp, res, w, xo, no, rows, full_cfg.hidden, full_cfg.hidden, full_cfg.hidden, full_cfg.hidden,
                                                                    full_cfg.eps, BF16, mode, s) == 0, ah.last_error()
                return f
            return [make(r) for r in range(virtual_ranks)]
        # the 1 MiB decode message through both kernels, and with the residual add + RMSNorm that follows it inside the all-reduce's launch;
        # the 64 MiB prefill message through the size rule (two-shot)
        engines = ({"xgmi_by_size": mk(0)} if msg > (1 << 20) else
                   {"xgmi_one_shot": mk(1), "xgmi_two_shot": mk(2), "xgmi_two_shot_fused_add_norm": {"allreduce": mk(2), "fused": mk_fused(2)},
                    "xgmi_one_shot_fused_add_norm": {"allreduce": mk(1), "fused": mk_fused(1)}})
        c = scfg
        info = None
        measure.xs = xs                                       # (for the verbose trace)
    else:
        ranks = [Rank(full_cfg, rank, world, B, ctx, local_rank, prefill=prefill)]
        c = ranks[0].c
        if world > 1 and xgmi is not None:                   # direct kernels only (no RCCL communicator)
            def via_xgmi(ptr, count, s):
                assert ah.lib.atoma_xgmi_allreduce_sum(xgmi, ptr, ptr, count, BF16, s) == 0, ah.last_error()
            def fused_xgmi(inp, res, w, xo, no, rows, s):
                assert ah.lib.atoma_xgmi_allreduce_add_rms_norm(xgmi, inp, res, w, xo, no, rows, full_cfg.hidden, full_cfg.hidden, full_cfg.hidden, full_cfg.hidden,
                                                                full_cfg.eps, BF16, 0, s) == 0, ah.last_error()
            engines = {"rccl": None, "xgmi": [via_xgmi], "xgmi_fused_add_norm": None if prefill else {"allreduce": [via_xgmi], "fused": [fused_xgmi]}}
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
            def fused_comm(inp, res, w, xo, no, rows, s):
                assert ah.lib.atoma_comm_set_mode(comm, 1) == 0, ah.last_error()
                assert ah.lib.atoma_allreduce_add_rms_norm(comm, inp, res, w, xo, no, rows, full_cfg.hidden, full_cfg.eps, BF16, s) == 0, ah.last_error()
            engines = {"rccl": via_comm(0), "xgmi": via_comm(1) if "ready" in info else None,
                       "xgmi_fused_add_norm": {"allreduce": via_comm(1), "fused": [fused_comm]} if ("ready" in info and not prefill) else None}
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
    what = f"prefill chunk of {prefill} tokens" if prefill else "decode step"
    if progress is not None:
        progress.update(workload="Llama-3.1-70B-shaped %s, TP=%d" % (what, world), world=world, xgmi_setup=info if world > 1 else None)
    res = measure(ranks, engines, steps, barrier, reduce_max, progress, use_graph=(not prefill) or len(ranks) > 1)
    nworld = virtual_ranks or world
    out = {"workload": f"Llama-3.1-70B-shaped {what} (SURVEY C4), {full_cfg.layers} layers, " + (f"one prompt of {prefill} tokens" if prefill else f"batch {B}, context {ctx}")
                       + ", bf16, " + (f"{virtual_ranks} ranks of a TP=8 job on ONE device (direct all-reduce only)" if virtual_ranks else f"TP={world}, one rank per GPU"),
           "world": nworld, "allreduce_message_bytes": msg, "allreduces_per_step": 2 * c.layers if nworld > 1 else 0, "engines": res}
    if full_cfg.layers != LLAMA_3_1_70B.layers:
        out["reduced"] = f"{full_cfg.layers} of the model's 80 layers (--layers): every layer has the full configs[3] shapes; step time is not the 80-layer step's"
    best = min((v["step_ms"] for v in res.values() if v), default=None)
    if prefill:
        fl = prefill_flops(c, prefill)
        out.update(rank_step_flops=int(fl), rank_roofline_ms_at_2500TFps=round(fl / 2.5e15 * 1e3, 3))
        if best:
            out["prefill_tokens_per_s"] = round(prefill / (best * 1e-3), 1)
            out["rank_TFLOPs"] = round(fl / (best * 1e-3) / 1e12 / (virtual_ranks or 1), 1)     # virtual ranks take turns on one device
    else:
        nbytes = step_bytes(c, B, ctx)
        out.update(rank_step_bytes=int(nbytes), rank_roofline_ms_at_8TBps=round(nbytes / 8e12 * 1e3, 3))
        if best:
            out["tokens_per_s"] = round(B / (best * 1e-3), 1)
            out["step_frac_of_roofline"] = round(out["rank_roofline_ms_at_8TBps"] / best, 4)
    if not virtual_ranks and world > 1:
        out["xgmi_setup"] = info
    if nworld > 1:
        out.update(logits_agreement(ranks, dist, world))
        out["xgmi_status"] = [int(ah.lib.atoma_xgmi_status(x)) for x in xs] if xs else None
    if unsharded is not None:
        # the unsharded step of the SAME model on the same inputs: sharding moves rounding points (a rank rounds its partial projection to
        # bf16 before the all-reduce sums it), so agreement is a few bf16 ulps on O(1) logits and the same argmax wherever it is not a near tie
        from halfs import to_f32
        unsharded.step.run()
        unsharded.stream.synchronize()
        a, b = to_f32(ranks[0].logits(), BF16), to_f32(unsharded.logits(), BF16)
        top2 = np.sort(b, 1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 0.25
        out["vs_unsharded"] = {"max_abs_diff": float(np.abs(a - b).max()), "logit_rms": float(np.sqrt((b * b).mean())),
                               "argmax_agree_where_clear": bool((a.argmax(1)[clear] == b.argmax(1)[clear]).all()), "clear_rows": int(clear.sum()), "rows": int(b.shape[0])}
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
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (quick runs; the output says so)")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="2..8: that many TP=8-shaped ranks on device 0, direct all-reduce (no RCCL); 8 = the whole job")
    ap.add_argument("--prefill", type=int, default=0, help="T > 0: the prefill chunk of T tokens (one prompt) instead of the decode step")
    ap.add_argument("--check-unsharded", action="store_true", help="virtual ranks: cut the shards from one full model on the device and compare with its own step")
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
    out = run(cfg, a.batch, a.ctx, a.steps, dist, rank, world, local_rank, a.virtual_ranks, prefill=a.prefill, check_unsharded=a.check_unsharded)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
