#!/usr/bin/env python3
"""bench.py -- the reference's headline micro-benchmark on MI355X.

Workload (BASELINE.json configs[1], SURVEY 8d "C2a"): paged-attention decode, B=256 sequences,
32 q heads / 8 kv heads (Llama-3.1-8B, TP=1), d=128, seq=4096 for every sequence, block_size=16,
bf16, q/K/V ~ N(0,1), block table = seeded random permutation over 73 728 physical pages,
scale 1/sqrt(128).  One "step" = one pass of the hot path over the batch = one
`run_mha` decode call (csrc/src/ffi.rs:4-64) producing 256 new-token attention outputs.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

N > 1 shards kv heads over ranks (tensor parallel, worker.rs:584-591) -- total work fixed
("strong" scaling) -- and each step ends with the path's one exchange step, the sum all-reduce
of the [B, h*d] activations over RCCL/xGMI (multi_gpu.rs:141-179).

Prints ONE JSON line on rank 0.  `value` = algorithmic HBM GB/s of the whole job with inputs
resident in HBM; `roofline` prices the dominant kernel against 8 TB/s; `cpu_baseline` is the
oracle's C restatement (oracle/c/oracle.c, OpenMP) timed on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (hipIpcGetMemHandle / RCCL across processes on this pool): before any HIP library loads

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(B, S, h, hk, d, page, e=2):
    """K and V once + q in + o out + block table + seqlens."""
    return 2 * B * S * hk * d * e + 2 * B * h * d * e + 4 * B * ((S + page - 1) // page) + 4 * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-ms", type=float, default=100.0, help="untimed load before the warm-up steps: brings the device from its idle power state to its sustained clocks (0 = off)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--identity-table", action="store_true", help="physical page i = logical page i")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure HBM traffic with rocprofv3 PMC passes inside this run")
    ap.add_argument("--no-extra", action="store_true", help="only the headline (profiling runs): skip the end-to-end extras")
    ap.add_argument("--tp-step", action="store_true", help="N = 1: also time the whole 70B-shaped decode step on this GPU (226 GB)")
    ap.add_argument("--allreduce-in-step", action="store_true",
                    help="N > 1: also all-reduce a [B, hidden] bf16 tensor inside every timed step (round 1's step).  Off by default: "
                         "paged attention shards by kv head with NO exchange step; the tensor-parallel all-reduces belong to the layer "
                         "and are timed, with both engines, in extra.tp_step")
    ap.add_argument("--cpu-sample-seqs", type=int, default=64)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # ATOMA_BENCH_FORCE_COMM=1: take the multi-rank code path (gloo rendezvous, RCCL communicator, per-step
    # all-reduce) even with one rank -- lets a single-GPU box exercise what the 8-GPU run will execute.
    force_comm = os.environ.get("ATOMA_BENCH_FORCE_COMM") == "1"
    dist = None
    os.environ.setdefault("ATOMA_XGMI_TIMEOUT_MS", "5000")   # read when the communicator builds its direct path: a lost peer costs 5 s, not a hang
    if world > 1 or force_comm:
        import torch  # only for the rendezvous / barrier / max-over-ranks (gloo, CPU tensors)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import atoma_hip as ah
    from halfs import BF16, from_f32          # bindings/halfs.py (product-side plumbing; oracle/ is only used by cpu_baseline)

    # ATOMA_BENCH_ONE_DEVICE=1: every rank on device 0 (the plumbing test of tests/test_bench_multirank_gpu.py on a 1-GPU box; ranks
    # then meet through the direct all-reduce over HIP IPC -- RCCL refuses two ranks on one device)
    one_device = os.environ.get("ATOMA_BENCH_ONE_DEVICE") == "1"
    device = 0 if one_device else local_rank
    ah.set_device(device)
    B, S, h, hk, d, page = args.batch, args.seq, args.heads, args.kv_heads, args.head_dim, args.block_size
    import tp
    qs_, ks_ = tp.head_shard(h, hk, rank, world)               # this rank's head shard (kv-head TP)
    h_l, hk_l = qs_.stop - qs_.start, ks_.stop - ks_.start
    pages_per_seq = (S + page - 1) // page
    n_pages = int(B * pages_per_seq * 1.125)                   # 73 728 for C2a
    rng = np.random.default_rng(0)
    bt = (np.arange(B * pages_per_seq) if args.identity_table else rng.permutation(n_pages)[: B * pages_per_seq])
    bt = bt.astype(np.int32).reshape(B, pages_per_seq)
    lens = np.full(B, S, np.int32)

    # K/V caches [n_pages, page, hk_l, d] bf16 ~ N(0,1): a 64 MiB random slab tiled over the cache
    page_elems = page * hk_l * d
    slab_pages = max(1, min(n_pages, (64 << 20) // (page_elems * 2)))
    slab_k = from_f32(rng.standard_normal((slab_pages, page_elems), dtype=np.float32), BF16)
    slab_v = from_f32(rng.standard_normal((slab_pages, page_elems), dtype=np.float32), BF16)
    cache_bytes = n_pages * page_elems * 2
    dkc, dvc = ah.DeviceBuffer(cache_bytes), ah.DeviceBuffer(cache_bytes)
    for dst, slab in ((dkc, slab_k), (dvc, slab_v)):
        off = 0
        while off < cache_bytes:
            n = min(slab.nbytes, cache_bytes - off)
            ah.hip_check(ah.hip.hipMemcpy(dst.ptr + off, slab.ctypes.data, n, ah.H2D), "upload cache")
            off += n
    q = from_f32(rng.standard_normal((B, 1, h_l, d), dtype=np.float32), BF16)
    dq, dbt, dl = ah.DeviceBuffer.from_numpy(q), ah.DeviceBuffer.from_numpy(bt), ah.DeviceBuffer.from_numpy(lens)
    do = ah.DeviceBuffer(q.nbytes)
    scale = float(d ** -0.5)

    comm = None                      # RCCL-bootstrapped atoma_comm (one per GPU), or None
    xgmi = None                      # direct-only communicator (atoma_xgmi_*, handles over gloo): ATOMA_BENCH_COMM=xgmi / one-device runs
    comm_kind = os.environ.get("ATOMA_BENCH_COMM", "xgmi" if one_device else "rccl")
    ranks_seen, rank_devices, comm_note = None, None, None
    if world > 1 or force_comm:
        import tp
        import torch
        if comm_kind == "rccl":
            comm = tp.rccl_comm(ah, dist, rank, world, device)     # unique id from rank 0, one comm per GPU
        else:
            xgmi = tp.xgmi_comm(ah, dist, rank, world, device, 2 << 20)
        act = ah.DeviceBuffer.zeros((B, h * d), np.uint16)    # [B, hidden] activations to all-reduce
        act_out = ah.DeviceBuffer.zeros((B, h * d), np.uint16)
        # ---- who is here: counted THROUGH the communicator (a sum all-reduce of ones over RCCL / the direct kernels), and the
        # (rank, host, device, PCI bus id) of every rank gathered over the rendezvous -- so that the line proves its own rank count
        ones = ah.DeviceBuffer.from_numpy(np.full(64, 0x3F80, np.uint16))          # bf16 1.0
        rc = (ah.lib.atoma_allreduce_sum(comm, ones.ptr, ones.ptr, 64, BF16, None) if comm is not None
              else ah.lib.atoma_xgmi_allreduce_sum(xgmi, ones.ptr, ones.ptr, 64, BF16, None))
        ah.synchronize()
        got = ones.numpy(np.uint16, (64,))
        ranks_seen = int(round(float(np.frombuffer((got.astype(np.uint32) << 16).tobytes(), np.float32)[0]))) if rc == 0 else 0
        import socket
        bus = (C.c_char * 64)()
        ah.hip.hipDeviceGetPCIBusId(bus, 64, device)
        mine = {"rank": rank, "host": socket.gethostname(), "device": device, "pci_bus_id": bus.value.decode(errors="replace")}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_devices = gathered
        # (the direct path behind an RCCL communicator is built later, inside the watchdog-protected extras: tp_step reports whether it
        # came up -- it has never run across real links from this tree, and the headline must not depend on it)
        comm_note = "RCCL communicator; direct path: see tp_step.xgmi_setup" if comm is not None else "direct kernels only (no RCCL communicator)"
        # ---- first contact: every engine sums a known pattern at 16 B / 1 MiB / 64 MiB BEFORE anything is timed (pass / fail + us per pair, on
        # stderr as it goes and in the line); a hang becomes a failed pair, never a lost headline.  ATOMA_BENCH_PREFLIGHT=0 skips it.
        preflight = None
        if os.environ.get("ATOMA_BENCH_PREFLIGHT", "1") != "0":
            try:
                preflight = tp.preflight(ah, dist, rank, world, comm=comm, xgmi=xgmi, log=lambda m: (sys.stderr.write(m + "\n"), sys.stderr.flush()),
                                         timeout_s=float(os.environ.get("ATOMA_BENCH_PREFLIGHT_TIMEOUT", "60")))
            except Exception as e:
                preflight = {"error": repr(e)}

    def step():
        ah.run_mha(dq, dkc, dvc, do, b=B, h=h_l, h_k=hk_l, d=d, seqlen_q=1, seqlen_k=pages_per_seq * page,
                   softmax_scale=scale, is_bf16=1, q_strides=(h_l * d, h_l * d, d), o_strides=(h_l * d, h_l * d, d),
                   k_strides=(page * hk_l * d, hk_l * d, d), v_strides=(page * hk_l * d, hk_l * d, d),
                   cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                   block_table_batch_stride=pages_per_seq, page_block_size=page, force_split_kernel=True,
                   unpadded_lse=False)
        if args.allreduce_in_step and comm is not None:
            assert ah.lib.atoma_allreduce_sum(comm, act.ptr, act_out.ptr, B * h * d, BF16, None) == 0, ah.last_error()
        elif args.allreduce_in_step and xgmi is not None:
            assert ah.lib.atoma_xgmi_allreduce_sum(xgmi, act.ptr, act_out.ptr, B * h * d, BF16, None) == 0, ah.last_error()

    def barrier():
        ah.synchronize()
        if dist is not None:
            dist.barrier()
        ah.synchronize()

    # Power state first: after the uploads above the device has idled for seconds and needs ~30 ms of load to return to its sustained
    # clocks (tools/probes/warm_probe.py, profiles/r04_clock_ramp_probe.txt: the same kernel reads 18 % slower in the first milliseconds).
    # Untimed, like the W warm-up steps that follow; --prewarm-ms 0 switches it off.
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        step()
        ah.synchronize()
    # ... and until it has SETTLED there: blocks of 10 untimed steps until two consecutive blocks agree within 1.5 % (at most 1.5 s more) -- a guard against
    # a slow ramp, normally one or two blocks.  (It does NOT remove the spread between runs: the same shape reads 0.795 .. 0.873 of the peak within ONE process
    # depending on where the allocator placed the 4.3 GB of K/V -- `extra.c2a_seeds` / `c2c_identity` next to the headline in profiles/r06_bench*.json --, and a
    # process started right after another one freed tens of GB read 0.80 twice in a row, settled: DESIGN.md 5.)  Untimed; never with a collective inside the
    # step (every rank must run the same steps).
    prev = None
    while args.prewarm_ms > 0 and not args.allreduce_in_step and (time.perf_counter() - t_pw) < 1.5 + args.prewarm_ms * 1e-3:
        tb = time.perf_counter()
        for _ in range(10):
            step()
        ah.synchronize()
        cur = time.perf_counter() - tb
        if prev is not None and abs(cur - prev) <= 0.015 * prev:
            break
        prev = cur
    prewarm_used_ms = (time.perf_counter() - t_pw) * 1e3
    for _ in range(args.warmup):
        step()
    barrier()
    # per-launch duration of the attention kernel: HIP events on the stream it is launched on (NULL)
    ev = [(ah.Event(), ah.Event()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(None)
        step()
        ev[i][1].record(None)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    launch_ms = [a.elapsed_ms(b_) for a, b_ in ev]
    kern_ms = float(np.mean(launch_ms))
    gpu_out = do.numpy(np.uint16, q.shape)                  # what the timed region wrote (checked below against the CPU leg's output)

    ms_per_step = wall * 1e3 / args.steps
    total_bytes = algorithmic_bytes(B, S, h, hk, d, page)           # whole job (all ranks)
    rank_bytes = algorithmic_bytes(B, S, h_l, hk_l, d, page)
    value = total_bytes / (ms_per_step * 1e-3) / 1e9
    achieved = rank_bytes / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "paged-attn decode HBM GB/s (decode tokens/s/GPU alongside), Llama-3.1-8B shape, TP=%d" % world,
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms, "prewarm_ms_used": round(prewarm_used_ms, 1),
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        # ATTENTION ONLY: new-token attention outputs per second of this micro-benchmark (one run_mha call = B tokens of ONE layer's attention).
        # The metric's model-level "decode tokens/s/GPU" (a whole Llama-3.1-8B step, BASELINE configs[2]) is `decode_tokens_per_s_per_gpu`
        # below, taken from extra.c3_trace / extra.c3_decode_step of this same run (one line, no misquoting).
        "attention_decode_tokens_per_s": round(B / (ms_per_step * 1e-3), 1),
        "attention_decode_tokens_per_s_per_gpu": round(B / (ms_per_step * 1e-3) / world, 1),
        "config": {"workload": "paged_attention_v2 micro-bench (BASELINE.json configs[1]): bs=%d, %d heads (%d kv), "
                               "d=%d, seq=%d, block_size=%d, bf16, %s block table over %d pages"
                               % (B, h, hk, d, S, page, "identity" if args.identity_table else "random-permutation",
                                  n_pages),
                   "parallelism": "tp%d (kv-head shards)" % world, "step": "one run_mha decode call over the batch"
                   + (" + all-reduce of [B, h*d] bf16" if ((comm is not None or xgmi is not None) and args.allreduce_in_step) else ""),
                   "collective": ("all-reduce of [B, hidden] bf16 inside the timed step (--allreduce-in-step)" if ((comm is not None or xgmi is not None) and args.allreduce_in_step)
                                  else "none in the timed step: the path shards by kv head without an exchange; the layer's tensor-parallel "
                                       "all-reduces are timed in extra.tp_step" if world > 1 else "n/a (one GPU)")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": (ah.lib.atoma_last_decode_kernel() or b"").decode() or "unknown (library without atoma_last_decode_kernel)",
                     "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": rank_bytes},
    }

    if world > 1 or force_comm:
        # every rank's own shard and kernel time (HIP events on its stream): a slow or mis-sharded rank shows here, not only in the max
        mine = {"rank": rank, "q_heads": h_l, "kv_heads": hk_l, "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": rank_bytes,
                "frac_of_hbm": round(rank_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": (ah.lib.atoma_last_decode_kernel() or b"").decode()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        out["per_rank"] = per_rank
        out["ranks_seen"] = ranks_seen                      # sum of ones over the communicator: must equal n_gpus
        out["ranks_expected"] = world
        out["rank_devices"] = rank_devices                  # one entry per rank: host, device ordinal, PCI bus id
        out["preflight"] = preflight                        # per engine and message size: the known-pattern all-reduce passed on every rank, slowest rank's us
        out["communicator"] = {"kind": "rccl (atoma_comm over ncclCommInitRank)" if comm is not None else "direct xGMI kernels over HIP IPC handles (no RCCL)",
                               "info": comm_note}
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs of this same command, corrected as the MI355X guide prescribes:
    # tools/summarize_profiles.py); null when no profile of the default workload is present.
    measured = None
    if world == 1 and not args.no_traffic and not os.environ.get("ATOMA_BENCH_CHILD"):
        measured = measure_traffic(sys.argv[1:])
    if measured is not None:
        out["roofline"]["traffic"] = int(measured["bytes"])
        # the two raw counters: the output of this launch is 2 MiB, so WRITE_SIZE far above 2048 KiB means scratch or partials are being
        # written (round 4: 38 913 KiB -- a by-value kernel argument whose address escaped was copied to scratch by every wavefront)
        out["roofline"]["traffic_counters_KiB_per_launch"] = {"FETCH_SIZE": measured["FETCH_SIZE_KiB"], "WRITE_SIZE": measured["WRITE_SIZE_KiB"]}
        out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes over 3 steps of this "
                                             "same command), (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch -- the gfx950 correction of MI355X_MICROARCH.md")
    elif world == 1 and (B, S, h, hk, d, page) == (256, 4096, 32, 8, 128, 16):
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
            for kern, e in json.load(open(f)).items():
                if "atoma::paged_decode" in kern and "fp8" not in kern and "hbm_traffic_bytes_per_launch" in e:
                    out["roofline"]["traffic"] = int(e["hbm_traffic_bytes_per_launch"])
                    out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT)

    # ---- end-to-end numbers beside the headline, each timed with HIP events in this same run (tools/bench_extra.py) ----
    # N = 1: the 8B decode step of configs[2], the prefill kernel, the configs[4] swap.  N > 1: the tensor-parallel decode step
    # of configs[3] (70B-shaped shard per rank, 2 all-reduces of [64, 8192] per layer) with RCCL and with the direct xGMI kernels.
    prefill_smp, extra_smp = None, {}
    if not args.no_extra:
        for b_ in (dkc, dvc, dq, do):
            b_.free()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        extra = {}
        tp_progress = {}             # filled by tp_step.run as it goes: what the watchdog prints if the run does not come back
        watchdog = None
        if world > 1:
            # The N-rank step has never run on a multi-GPU box from this tree (the pool this was built on has 1-GPU boxes): if it
            # does not come back, the headline measured above must still go out.  Every rank arms the same timer; rank 0 prints.
            import threading
            limit = float(os.environ.get("ATOMA_BENCH_EXTRA_TIMEOUT", "240"))

            def bail():
                if rank == 0:
                    out["tp_step"] = tp_summary(tp_progress, "did not finish within %.0f s: partial results" % limit)
                    out["extra"] = {"error": "the multi-GPU extras did not finish within %.0f s" % limit, "tp_step_progress": tp_progress}
                    sys.stdout.write(json.dumps(out) + "\n")
                    sys.stdout.flush()
                os._exit(0)
            watchdog = threading.Timer(limit, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            if world == 1 and not force_comm:
                import bench_extra
                extra.update(bench_extra.collect(prefill_sample=not args.no_cpu_baseline))
                if args.tp_step:
                    import tp_step
                    extra["tp_step"] = tp_step.run(steps=10)
            elif world > 1:
                import tp_step
                small = os.environ.get("ATOMA_BENCH_TP_SMALL") == "1"     # plumbing tests: 2 layers, batch 8, context 256 (flagged in the output)
                cfg = tp_step.DS.Config(2, 8192, 64, 8, 128, 28672, 128256) if small else tp_step.LLAMA_3_1_70B
                extra["tp_step"] = tp_step.run(cfg, B=8 if small else 64, ctx=256 if small else 4096, steps=3 if small else 10, dist=dist, rank=rank, world=world,
                                               local_rank=device, comm=comm, xgmi=xgmi, progress=tp_progress)
                if small:
                    extra["tp_step"]["reduced"] = "ATOMA_BENCH_TP_SMALL=1: 2 layers, batch 8, context 256 -- a plumbing run, not configs[3]"
        except Exception as e:              # never lose the headline to an extra
            extra["error"] = repr(e)
        if watchdog is not None:
            watchdog.cancel()
        if isinstance(extra.get("prefill"), dict):
            prefill_smp = extra["prefill"].pop("sample", None)      # host copies of sampled rows: checked below, not part of the line
        for name, e in extra.items():                                # the other extras' samples (decode sequences, norm / RoPE rows, prefill rows)
            if isinstance(e, dict) and "sample" in e:
                extra_smp[name] = e.pop("sample")
        out["extra"] = extra
        # the metric's model-level number at the top level: the configs[2] trace end to end when it ran, else the mid-trace step
        tr, stp = extra.get("c3_trace"), extra.get("c3_decode_step")
        if isinstance(tr, dict) and "decode_tokens_per_s_per_gpu" in tr:
            out["decode_tokens_per_s_per_gpu"] = tr["decode_tokens_per_s_per_gpu"]
            out["decode_tokens_per_s_per_gpu_source"] = ("extra.c3_trace: Llama-3.1-8B bf16 TP=1, 256 requests, prefill 2048 + 512 decode steps at batch 256, wall clock of the decode "
                                                         "phase incl. the host loop; %.3f of the byte bound" % tr.get("decode_frac_of_roofline", float("nan")))
        elif isinstance(stp, dict) and "decode_tokens_per_s_per_gpu" in stp:
            out["decode_tokens_per_s_per_gpu"] = stp["decode_tokens_per_s_per_gpu"]
            out["decode_tokens_per_s_per_gpu_source"] = "extra.c3_decode_step: one mid-trace Llama-3.1-8B step at batch 256 (hipGraph replay)"
        if world > 1:
            out["tp_step"] = tp_summary(extra.get("tp_step") or tp_progress, extra.get("error"))

    failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is an N = 1 leg
        out["cpu_baseline"], o_cpu = cpu_baseline(args, bt, lens, slab_k, slab_v, q, n_pages, page_elems, hk_l, h_l)
        # ---- the timed region's output against the CPU leg's (the oracle's C restatement, here as the checker): the sequences the
        # baseline computed anyway, BASELINE.json's tolerance (1e-3 + one unit in the last place of bf16)
        from halfs import to_f32
        got, want = to_f32(gpu_out[:o_cpu.shape[0]], BF16), to_f32(o_cpu, BF16)
        err = np.abs(got - want)
        ok = bool(np.isfinite(got).all() and (err <= 1e-3 + 2.0 ** -7 * np.abs(want)).all())
        out["verified"] = {"ok": ok, "sequences": int(o_cpu.shape[0]), "max_err": float(err.max()), "tolerance": "1e-3 + 2^-7 |ref| (bf16)",
                           "against": "oracle/c/oracle.c oracle_decode_grouped (f32) on the same tensors"}
        failed = failed or not ok
        if prefill_smp is not None:
            out["verified"]["prefill"] = verify_prefill_sample(prefill_smp)
            failed = failed or not out["verified"]["prefill"]["ok"]
        for name, smp in extra_smp.items():                   # every other driver-timed row: sampled outputs of its TIMED calls against the oracle
            out["verified"][name] = verify_extra_sample(smp)
            failed = failed or not out["verified"][name]["ok"]
        def any_inexact(e):                                   # bit-exact ops checked against their definition where they ran (also in the nested size sweeps)
            if isinstance(e, dict):
                return e.get("bit_exact") is False or any(any_inexact(v) for v in e.values())
            return False
        failed = failed or any_inexact(out.get("extra", {}))
    if comm is not None or xgmi is not None:
        if dist is not None:
            dist.barrier()
        if comm is not None:
            ah.lib.atoma_comm_destroy(comm)
        else:
            ah.lib.atoma_xgmi_destroy(xgmi)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the one JSON line goes out LAST: RCCL's version banner sits in the C stdio buffer until it is flushed
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stderr.flush()
        print(json.dumps(out), flush=True)
    if failed:
        sys.exit(3)                                          # a line whose numbers come from wrong results must not pass for a measurement


def verify_prefill_sample(smp):
    """extra.prefill: sampled query rows of two (sequence, head) pairs against the f32 definition (oracle/attn_oracle.py attend_rows, the
    checker); rows that see >= 512 keys at 1e-3 + 1 ulp, the first rows of a sequence at the P-rounding bound (tests/util.py)."""
    from oracle import attn_oracle as A
    from halfs import to_f32, BF16
    worst, ok, n = 0.0, True, 0
    for it in smp:
        kf, vf = to_f32(it["k"], BF16)[:, None, :], to_f32(it["v"], BF16)[:, None, :]
        for r, qrow, orow in zip(it["rows"], it["q"], it["o"]):
            ref, _ = A.attend_rows(to_f32(qrow, BF16)[None, None, :], kf[:r + 1], vf[:r + 1], np.float32(it["scale"]), causal=False)
            got = to_f32(orow, BF16)
            err = np.abs(got - ref[0, 0])
            tol = (1e-3 if r + 1 >= 512 else 4e-3) + 2.0 ** -7 * np.abs(ref[0, 0])
            ok = ok and bool(np.isfinite(got).all() and (err <= tol).all())
            worst = max(worst, float(err.max()))
            n += 1
    return {"ok": ok, "rows": n, "max_err": worst, "against": "oracle/attn_oracle.py attend_rows (f32 definition)"}


def verify_extra_sample(smp):
    """Samples of tools/bench_extra.py rows against the oracle (the checker): decode sequences and prefill rows against the f32 definition
    (attend_rows; 1e-3 + 1 ulp from 512 visible keys on, the P-rounding bound below), RMSNorm rows within one ulp of the f32 definition,
    RoPE rows bit-exact in Candle's per-op arithmetic."""
    from oracle import attn_oracle as A, norm_rope_oracle as NR
    from halfs import to_f32, BF16
    if smp and smp[0].get("kind") is None:                       # prefill rows (tools/bench_extra.py prefill)
        return verify_prefill_sample(smp)
    ok, worst, n, what = True, 0.0, 0, []
    for it in smp:
        if it["kind"] == "decode":
            ref, _ = A.attend_rows(to_f32(it["q"], BF16)[None], to_f32(it["k"], BF16), to_f32(it["v"], BF16), np.float32(it["scale"]), causal=False)
            got = to_f32(it["o"], BF16)
            err = np.abs(got - ref[0])
            tol = (1e-3 if it["L"] >= 512 else 4e-3) + 2.0 ** -7 * np.abs(ref[0])
            ok = ok and bool(np.isfinite(got).all() and (err <= tol).all())
            worst, n = max(worst, float(err.max())), n + got.shape[0]
            what.append("decode rows vs attend_rows (f32)")
        elif it["kind"] == "rms_norm":
            ref = to_f32(NR.rms_norm(it["x"], it["w"], it["eps"], BF16), BF16)
            got = to_f32(it["y"], BF16)
            err = np.abs(got - ref)
            ok = ok and bool((err <= 2.0 ** -7 * np.abs(ref) + 1e-6).all())
            worst, n = max(worst, float(err.max())), n + got.shape[0]
            what.append("RMSNorm rows within 1 ulp of norm_rope_oracle.rms_norm")
        elif it["kind"] == "rope":
            T = it["x"].shape[0]
            ref = NR.rope(it["x"], it["cos"], it["sin"], np.arange(T), BF16, mode="per_op")
            ok = ok and bool(np.array_equal(ref, it["y"]))
            n += T
            what.append("RoPE rows bit-exact vs norm_rope_oracle.rope (per-op rounding)")
    return {"ok": ok, "rows": n, "max_err": worst, "against": "; ".join(sorted(set(what)))}


def tp_summary(res, note=None):
    """The tensor-parallel step of configs[3] at the TOP level of the line: step time with each all-reduce engine,
    the all-reduce alone, which engines were available -- from tp_step.run's result, or from its progress record when it did not finish."""
    eng = (res or {}).get("engines") or {}
    pick = lambda name, key: (eng.get(name) or {}).get(key)
    out = {"workload": (res or {}).get("workload"), "world": (res or {}).get("world"),
           "rccl_step_ms": pick("rccl", "step_ms"), "xgmi_step_ms": pick("xgmi", "step_ms"),
           "rccl_allreduce_us": pick("rccl", "allreduce_us"), "xgmi_allreduce_us": pick("xgmi", "allreduce_us"),
           "xgmi_fused_add_norm_step_ms": pick("xgmi_fused_add_norm", "step_ms"),   # residual add + RMSNorm inside the all-reduce's launch (2 launches per layer fewer)
           "engines_available": {k: v is not None for k, v in eng.items()}, "xgmi_setup": (res or {}).get("xgmi_setup"),
           "tokens_per_s": (res or {}).get("tokens_per_s"), "step_frac_of_roofline": (res or {}).get("step_frac_of_roofline"),
           "allreduces_per_step": (res or {}).get("allreduces_per_step"), "allreduce_message_bytes": (res or {}).get("allreduce_message_bytes")}
    if (res or {}).get("reduced"):
        out["reduced"] = res["reduced"]
    if note:
        out["note"] = note
    return out


def measure_traffic(argv):
    """HBM bytes per launch of the decode kernel from the PMC counters, collected as the MI355X guide prescribes: one
    rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE; KiB; on gfx950 FETCH_SIZE tallies a wide coalesced stream at
    half its bytes), each over a short child run of this same command.  None when rocprofv3 is missing or anything fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    keep = [a for a in argv if a not in ("--no-extra", "--no-cpu-baseline", "--no-traffic")]
    drop_next = False
    child_args = []
    for a in keep:                       # strip --steps / --warmup (and their values): the child runs 3 + 1
        if drop_next:
            drop_next = False
            continue
        if a in ("--steps", "--warmup"):
            drop_next = True
            continue
        if a.startswith("--steps=") or a.startswith("--warmup="):
            continue
        child_args.append(a)
    means = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="atoma_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--no-traffic"] + child_args
            env = dict(os.environ, TMPDIR="/tmp", ATOMA_BENCH_CHILD="1")
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    # the decode kernel the dispatcher took (paged_decode_kernel / paged_decode_mqk_kernel / ...), not the combine kernel
                    if "atoma::paged_decode" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        vals.append(float(r["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if not vals:
                return None
            means[counter] = sum(vals) / len(vals)
        return {"bytes": (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024, "FETCH_SIZE_KiB": round(means["FETCH_SIZE"], 1),
                "WRITE_SIZE_KiB": round(means["WRITE_SIZE"], 1)}
    except Exception:
        return None


def cpu_baseline(args, bt, lens, slab_k, slab_v, q, n_pages, page_elems, hk, h):
    """The oracle's C restatement (fa_acausal over the gathered pages, f32, OpenMP) on the first
    `--cpu-sample-seqs` sequences of the same workload; throughput is per byte, so the sample
    scales linearly to the batch."""
    # thread placement before libgomp starts: one thread per core, spread over the sockets (all cores, pinned)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    from util import oracle_c
    lib = oracle_c()
    avail = len(os.sched_getaffinity(0)) or 1
    few = avail < 8                                             # a cpuset of a few hardware threads (the driver's box): a short leg
    Bs = min(args.cpu_sample_seqs, 16 if few else args.cpu_sample_seqs, args.batch)
    d, page, S = args.head_dim, args.block_size, args.seq
    # host copy of the cache: the same slab tiling as on the device
    reps = -(-n_pages // slab_k.shape[0])
    kc = np.tile(slab_k, (reps, 1))[:n_pages].reshape(n_pages, page, hk, d)
    vc = np.tile(slab_v, (reps, 1))[:n_pages].reshape(n_pages, page, hk, d)
    qs = np.ascontiguousarray(q[:Bs])
    o = np.zeros_like(qs)
    bts = np.ascontiguousarray(bt[:Bs])
    ls = np.ascontiguousarray(lens[:Bs])
    i64 = C.c_int64
    lib.oracle_decode_grouped.argtypes = [C.c_void_p] * 6 + [i64, C.c_int, i64, i64, i64] + [C.c_int] * 4 + [C.c_float, C.c_int]
    lib.oracle_max_threads.restype = C.c_int
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(cores):       # one task per (sequence, kv head): K / V rows converted once for the whole query group, vectorised inner loops
        lib.oracle_decode_grouped(vp(qs), vp(kc), vp(vc), vp(o), vp(ls), vp(bts), bts.shape[1], page, page * hk * d, hk * d, d,
                                  Bs, h, hk, d, float(d ** -0.5), cores)
    # OpenMP scaling of this memory-streaming loop saturates well below the box's hardware-thread
    # count (256 on the MI355X host): pick the fastest thread count, then time it for ~10 s.
    forced = int(os.environ.get("ATOMA_BENCH_CPU_THREADS", 0))
    cands = [forced] if forced else sorted({min(avail, c) for c in (4, 8, 16, 32, 64, 128, avail)})
    sweep = {}
    best = None
    for c in cands:
        run(c)                                                      # warm the page cache / thread pool
        t0 = time.perf_counter()
        run(c)
        el = time.perf_counter() - t0
        sweep[str(c)] = round(algorithmic_bytes(Bs, S, h, hk, d, page) / el / 1e9, 2)
        if best is None or el < best[0]:
            best = (el, c)
    cores = best[1]
    t0 = time.perf_counter()
    reps_done = 0
    budget = 3.0 if few else 10.0
    while reps_done < 3 or (time.perf_counter() - t0 < budget and reps_done < 50):
        run(cores)
        reps_done += 1
    dt = (time.perf_counter() - t0) / reps_done
    nbytes = algorithmic_bytes(Bs, S, h, hk, d, page)
    def read(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    import glob
    numa_nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*")) or None
    return {"value": round(nbytes / dt / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "host_hw_threads": avail, "thread_sweep_GBps": sweep,
            "placement": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"), "numa_nodes": numa_nodes,
                          "cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"), "cpus_allowed": len(os.sched_getaffinity(0))},
            "sample": "first %d of %d sequences of the same workload (same tensors), %d repetitions, %.3f s each; "
                      "fa_acausal f32 restatement, one task per (sequence, kv head), K/V rows converted once per query group, "
                      "vectorised inner loops (oracle/c/oracle.c oracle_decode_grouped, gcc -O3 -march=x86-64-v3 -fopenmp); thread count = the "
                      "fastest of 4/8/16/32/64/128/all (thread_sweep_GBps), threads bound to cores and spread (placement)" % (Bs, args.batch, reps_done, dt),
            "decode_tokens_per_s": round(Bs / dt, 1)}, o


if __name__ == "__main__":
    main()
