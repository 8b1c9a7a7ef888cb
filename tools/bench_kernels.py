#!/usr/bin/env python3
"""Per-kernel measurements of the hot path on one MI355X (BASELINE.md section 2 workloads).

Prints one JSON line per workload: achieved GB/s (HBM-bound kernels) or TFLOP/s (prefill), the
algorithmic bytes / flops it is computed from, and the fraction of the chip peak.  Timing = HIP events
on the NULL stream around `iters` back-to-back launches after warm-up.

  python tools/bench_kernels.py [decode prefill prefill_paged cache norm sampling linear graph step swap]   (default: all)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import atoma_hip as ah  # noqa: E402
from halfs import BF16, from_f32  # noqa: E402  (atoma-infer_amd/bindings/halfs.py)

HBM, MFMA_BF16 = 8000.0, 2500.0  # GB/s, TFLOP/s dense (MI355X_MICROARCH.md)


WARM_MS = float(os.environ.get("ATOMA_BENCH_WARM_MS", "60"))


def timeit(fn, iters=20, warmup=3):
    """Mean time of `iters` back-to-back calls AT THE DEVICE'S SUSTAINED CLOCKS: the warm-up runs for at least WARM_MS of device time.
    After half a second of idling (an upload, a host-side oracle) an MI355X needs ~30 ms of load to return to its sustained clocks; ten
    calls of a 0.6 ms kernel timed after two warm-up calls read 833 TF/s where the steady state is 1020 (tools/probes/warm_probe.py,
    profiles/r04_clock_ramp_probe.txt) -- a serving engine never idles that long between steps."""
    t0, n = time.perf_counter(), 0
    while n < warmup or (time.perf_counter() - t0) * 1e3 < WARM_MS:
        fn()
        ah.synchronize()
        n += 1
    a, b = ah.Event(), ah.Event()
    a.record(None)
    for _ in range(iters):
        fn()
    b.record(None)
    b.synchronize()
    return a.elapsed_ms(b) / iters


def rand_dev(rng, nbytes):
    """nbytes of bf16 N(0,1) on the device: a 64 MiB random slab tiled."""
    slab = from_f32(rng.standard_normal(min(nbytes // 2, 32 << 20), dtype=np.float32), BF16)
    buf = ah.DeviceBuffer(nbytes)
    off = 0
    while off < nbytes:
        n = min(slab.nbytes, nbytes - off)
        ah.hip_check(ah.hip.hipMemcpy(buf.ptr + off, slab.ctypes.data, n, ah.H2D), "upload")
        off += n
    return buf


def emit(name, ms, nbytes=None, flops=None, **extra):
    rec = {"workload": name, "ms": round(ms, 4)}
    if nbytes is not None:
        gbs = nbytes / (ms * 1e-3) / 1e9
        rec.update(algorithmic_bytes=int(nbytes), GBps=round(gbs, 1), frac_hbm=round(gbs / HBM, 4))
    if flops is not None:
        tf = flops / (ms * 1e-3) / 1e12
        rec.update(flops=int(flops), TFLOPs=round(tf, 1), frac_mfma=round(tf / MFMA_BF16, 4))
    rec.update(extra)
    print(json.dumps(rec), flush=True)


def decode_case(name, B, S, h, hk, d=128, page=16, identity=False, ragged=False, seed=0, lens=None):
    if os.environ.get("ATOMA_BENCH_DECODE_SHAPE") and os.environ["ATOMA_BENCH_DECODE_SHAPE"] not in name:   # one shape, for counter passes
        return
    rng = np.random.default_rng(seed)
    pps = (S + page - 1) // page
    if lens is not None:       # explicit lengths (max S): only the pages the sequences own exist
        lens = np.asarray(lens, np.int32)
        need = (lens.astype(np.int64) + page - 1) // page
        n_pages = int(need.sum() * 1.125) + 1
        perm = rng.permutation(n_pages).astype(np.int32)
        bt = np.zeros((B, pps), np.int32)
        o = 0
        for i, n in enumerate(need):
            bt[i, :n] = perm[o:o + n]
            o += int(n)
    else:
        n_pages = int(B * pps * 1.125)
        bt = (np.arange(B * pps) if identity else rng.permutation(n_pages)[: B * pps]).astype(np.int32).reshape(B, pps)
        lens = (rng.integers(S // 2, S + 1, B) if ragged else np.full(B, S)).astype(np.int32)
    kc, vc = rand_dev(rng, n_pages * page * hk * d * 2), rand_dev(rng, n_pages * page * hk * d * 2)
    q = rand_dev(rng, B * h * d * 2)
    o = ah.DeviceBuffer(B * h * d * 2)
    dbt, dl = ah.DeviceBuffer.from_numpy(bt), ah.DeviceBuffer.from_numpy(lens)

    # the ABI takes strides: ATOMA_KV_LAYOUT=head_major lays a page out as [h_k][page][d] instead of the reference's [page][h_k][d]
    kv_strides = (page * hk * d, hk * d, d)
    if os.environ.get("ATOMA_KV_LAYOUT", "") == "head_major":
        kv_strides = (page * hk * d, d, page * d)
        name += " [head-major pages]"

    def run():
        ah.run_mha(q, kc, vc, o, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5,
                   is_bf16=1, q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=kv_strides,
                   v_strides=kv_strides, cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt,
                   block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False)
    ms = timeit(run)
    tot = int(lens.astype(np.int64).sum())
    nbytes = 2 * tot * hk * d * 2 + 2 * B * h * d * 2 + 4 * int(((lens + page - 1) // page).sum()) + 4 * B
    emit(name, ms, nbytes=nbytes, tokens_per_s=round(B / (ms * 1e-3)))
    for b_ in (kc, vc, q, o, dbt, dl):
        b_.free()


def bench_decode():
    decode_case("C2a decode B=256 h=32 hk=8 S=4096 random table", 256, 4096, 32, 8)
    decode_case("C2c decode identity table", 256, 4096, 32, 8, identity=True)
    decode_case("C2c decode ragged U[2048,4096]", 256, 4096, 32, 8, ragged=True)
    decode_case("decode ragged U[2048,4096] MHA hk=32", 256, 4096, 32, 32, ragged=True)
    decode_case("decode narrow spread U[2048,2560) (the contexts of the C3 step)", 256, 2560, 32, 8, lens=np.random.default_rng(9).integers(2048, 2560, 256))
    decode_case("decode narrow spread U[3800,4096]", 256, 4096, 32, 8, lens=np.random.default_rng(10).integers(3800, 4097, 256))
    decode_case("C2b decode MHA hk=32", 256, 4096, 32, 32)
    decode_case("C4 decode Llama-70B shape TP=1: B=256 h=64 hk=8 S=4096", 256, 4096, 64, 8)
    decode_case("C4 decode 70B TP=8 shard: B=256 h=8 hk=1 S=4096", 256, 4096, 8, 1)
    decode_case("decode 70B TP=8 shard: B=64 h=8 hk=1 S=4096 (split-KV)", 64, 4096, 8, 1)
    decode_case("decode 8B TP=2 shard: B=256 h=16 hk=4 S=4096 (one rank of bench.py --gpus 2)", 256, 4096, 16, 4)
    decode_case("decode 8B TP=4 shard: B=256 h=8 hk=2 S=4096 (one rank of bench.py --gpus 4)", 256, 4096, 8, 2)
    decode_case("decode 8B TP=8 shard: B=256 h=4 hk=1 S=4096 (split-KV; one rank of bench.py --gpus 8)", 256, 4096, 4, 1)
    decode_case("decode B=1 S=4096 (split-KV)", 1, 4096, 32, 8)
    decode_case("decode B=16 S=8192", 16, 8192, 32, 8)
    decode_case("decode B=256 S=1024", 256, 1024, 32, 8)
    decode_case("decode Llama-3.2-1B shape d=64 B=256 S=4096", 256, 4096, 32, 8, d=64)
    decode_case("decode d=64 ragged U[2048,4096] B=256 h=32 hk=8", 256, 4096, 32, 8, d=64, ragged=True)
    decode_case("decode d=64 B=16 S=8192 h=32 hk=8 (split-KV)", 16, 8192, 32, 8, d=64)
    decode_case("decode d=64 MHA B=256 S=2048 h=16 hk=16", 256, 2048, 16, 16, d=64)
    # the other head sizes the reference instantiates (attn_decode_anyd2_kernel; ATOMA_GENERIC_DECODE_STREAM=0 at load time or atoma_set_option generic_decode_stream 0: the row-per-lane coverage kernel)
    decode_case("decode d=96 (Phi-3-mini heads) B=256 S=2048 h=32 hk=32", 256, 2048, 32, 32, d=96)
    decode_case("decode d=256 B=256 S=2048 h=16 hk=4", 256, 2048, 16, 4, d=256)
    decode_case("decode d=96 B=64 S=2048 h=32 hk=32 (the Phi-3-mini-shaped step's attention)", 64, 2048, 32, 32, d=96)
    decode_case("decode d=96 B=8 S=4096 h=32 hk=32 (8 wavefronts per unit)", 8, 4096, 32, 32, d=96)


def bench_decode_fp8():
    """f4: the C2a workload over an fp8 (e4m3fn) KV cache -- same B, heads, contexts, random block table; half the K/V bytes.
    Bytes = the fp8 cache read once + q in + o out + table + lengths (+ the 2 x h_k scales)."""
    rng = np.random.default_rng(0)
    for name, B, S, h, hk, ragged in (("C2a over fp8 KV: B=256 h=32 hk=8 S=4096", 256, 4096, 32, 8, False),
                                      ("C2c over fp8 KV: ragged U[2048,4096]", 256, 4096, 32, 8, True),
                                      ("fp8 KV, 70B TP=8 shard: B=64 h=8 hk=1 S=4096 (8 q heads in one pass)", 64, 4096, 8, 1, False),
                                      ("fp8 KV, B=16 S=8192", 16, 8192, 32, 8, False)):
        if os.environ.get("ATOMA_FP8_SHAPE") and os.environ["ATOMA_FP8_SHAPE"] not in name:   # one shape, for counter passes
            continue
        d, page = 128, 16
        pps = S // page
        n_pages = int(B * pps * 1.125)
        lens = rng.integers(2048, S + 1, B).astype(np.int32) if ragged else np.full(B, S, np.int32)
        bt = rng.permutation(n_pages)[:B * pps].astype(np.int32).reshape(B, pps)
        kc = ah.DeviceBuffer(n_pages * page * hk * d)
        vc = ah.DeviceBuffer(n_pages * page * hk * d)
        slab = rng.integers(0, 0x78, 32 << 20, dtype=np.uint8) | (rng.integers(0, 2, 32 << 20, dtype=np.uint8) << 7)   # finite e4m3 codes, both signs
        for buf in (kc, vc):
            off = 0
            while off < buf.nbytes:
                n = min(slab.nbytes, buf.nbytes - off)
                ah.hip_check(ah.hip.hipMemcpy(buf.ptr + off, slab.ctypes.data, n, ah.H2D), "upload")
                off += n
        q = rand_dev(rng, B * h * d * 2)
        o = ah.DeviceBuffer(B * h * d * 2)
        ks = ah.DeviceBuffer.from_numpy(np.full(hk, 0.02, np.float32))
        vs = ah.DeviceBuffer.from_numpy(np.full(hk, 0.02, np.float32))
        dbt, dl = ah.DeviceBuffer.from_numpy(bt), ah.DeviceBuffer.from_numpy(lens)

        # the ABI takes strides: ATOMA_FP8_LAYOUT=head_major lays a page out as [h_k][page][d] (a head's 16 rows = 2 KiB contiguous)
        head_major = os.environ.get("ATOMA_FP8_LAYOUT", "") == "head_major"
        row_stride, head_stride = (d, page * d) if head_major else (hk * d, d)

        def run():
            rc = ah.lib.atoma_paged_decode_fp8(q.ptr, kc.ptr, vc.ptr, o.ptr, ks.ptr, vs.ptr, dbt.ptr, dl.ptr, B, h, hk, d, pps, page, h * d, d, h * d, d,
                                               page * hk * d, row_stride, head_stride, d ** -0.5, 1, None)
            assert rc == 0, ah.last_error()
        if head_major:
            name += " [head-major pages]"
        ms = timeit(run, iters=20)
        tok = int(lens.sum())
        nbytes = 2 * tok * hk * d + 2 * B * h * d * 2 + 4 * int(((lens + page - 1) // page).sum()) + 4 * B + 8 * hk
        emit(name, ms, nbytes=nbytes, decode_tokens_per_s=round(B / (ms * 1e-3)), bf16_cache_bytes=2 * tok * hk * d * 2)
        for b_ in (kc, vc, q, o):
            b_.free()


def bench_prefill():
    """Every shape on the default kernel (prefill_cfg 4: the hand-scheduled stream for head_dim 128) and on round 2's kernel (cfg 0) beside it;
    ATOMA_PREFILL_CFG = one configuration only."""
    cfgs = (int(os.environ["ATOMA_PREFILL_CFG"]),) if os.environ.get("ATOMA_PREFILL_CFG") else (4, 0)
    rng = np.random.default_rng(1)
    shapes = ((2048, 4, 128), (4096, 2, 128), (4096, 4, 128), (512, 16, 128), (2048, 16, 128), (2048, 16, 64))
    if os.environ.get("ATOMA_BENCH_PREFILL_SHAPE"):    # e.g. "2048x16x128": one shape, for counter passes
        shapes = (tuple(int(t) for t in os.environ["ATOMA_BENCH_PREFILL_SHAPE"].split("x")),)
    for S, nseq, d in shapes:
        h, hk = 32, 8
        T = S * nseq
        q, k, v = rand_dev(rng, T * h * d * 2), rand_dev(rng, T * hk * d * 2), rand_dev(rng, T * hk * d * 2)
        o = ah.DeviceBuffer(T * h * d * 2)
        cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))

        def run():
            ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1,
                       q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d),
                       is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu)
        for cfg in cfgs:
            if d != 128 and cfg == 4 and 0 in cfgs:
                continue                                  # (head_dim 64 runs on cfg 0 either way)
            ah.lib.atoma_set_option(b"prefill_cfg", cfg)
            ms = timeit(run, iters=10)
            flops = 4 * S * S * h * d / 2 * nseq
            emit(f"P1 prefill causal varlen S={S} x{nseq} d={d} (32 q / 8 kv heads) cfg={cfg}", ms, flops=flops, tokens_per_s=round(T / (ms * 1e-3)))
        ah.lib.atoma_set_option(b"prefill_cfg", 4)
        for b_ in (q, k, v, o, cu):
            b_.free()


def bench_prefill_paged():
    """a7: prefill over the paged cache (prefix / chunked prefill, flash_attn_varlen_with_block_table)."""
    rng = np.random.default_rng(6)
    h, hk, d, S, nseq = 32, 8, 128, 2048, 16
    T = S * nseq
    q, o = rand_dev(rng, T * h * d * 2), ah.DeviceBuffer(T * h * d * 2)
    cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
    for page in (16, 64):
        pps = S // page
        nb = nseq * pps
        k, v = rand_dev(rng, nb * page * hk * d * 2), rand_dev(rng, nb * page * hk * d * 2)
        bt = ah.DeviceBuffer.from_numpy(rng.permutation(nb).astype(np.int32).reshape(nseq, pps))

        def run():
            ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1,
                       q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(page * hk * d, hk * d, d),
                       v_strides=(page * hk * d, hk * d, d), is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu, block_table=bt,
                       block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True)
        ms = timeit(run, iters=10)
        emit(f"P2 prefill over the paged cache, causal S={S} x{nseq} d={d} page={page}", ms, flops=4 * S * S * h * d / 2 * nseq)
        for b_ in (k, v, bt):
            b_.free()


def bench_cache():
    rng = np.random.default_rng(2)
    hk, d, page, nb = 8, 128, 16, 4096
    kc, vc = rand_dev(rng, nb * page * hk * d * 2), rand_dev(rng, nb * page * hk * d * 2)
    for T in (256, 2048, 8192):
        k, v = rand_dev(rng, T * hk * d * 2), rand_dev(rng, T * hk * d * 2)
        slots = ah.DeviceBuffer.from_numpy(rng.permutation(nb * page)[:T].astype(np.int64))
        ms = timeit(lambda: ah.lib.reshape_and_cache_flash(k.ptr, v.ptr, kc.ptr, vc.ptr, slots.ptr, page * hk * d, T, hk, d,
                                                           page, hk * d, hk * d, 1, None))
        emit(f"K4 reshape_and_cache_flash T={T}", ms, nbytes=4 * T * hk * d * 2 + 8 * T)
    L = 32
    caches = [(rand_dev(rng, 1024 * page * hk * d * 2), rand_dev(rng, 1024 * page * hk * d * 2)) for _ in range(L)]
    kp = ah.DeviceBuffer.from_numpy(np.array([c[0].ptr for c in caches], np.int64))
    vp = ah.DeviceBuffer.from_numpy(np.array([c[1].ptr for c in caches], np.int64))
    for P in (1, 64, 500):
        perm = rng.permutation(1024)
        mp = ah.DeviceBuffer.from_numpy(np.stack([perm[:P], perm[P:2 * P]], 1).astype(np.int64))
        ms = timeit(lambda: ah.lib.copy_blocks_bf16(kp.ptr, vp.ptr, mp.ptr, L, P, page * hk * d, None))
        emit(f"K5 copy_blocks L=32 pairs={P}", ms, nbytes=4 * P * L * page * hk * d * 2)


def bench_norm():
    rng = np.random.default_rng(3)
    for T in (256, 2048):
        hidden = 4096
        x, w, y = rand_dev(rng, T * hidden * 2), rand_dev(rng, hidden * 2), ah.DeviceBuffer(T * hidden * 2)
        ms = timeit(lambda: ah.lib.atoma_rms_norm(x.ptr, w.ptr, y.ptr, T, hidden, hidden, hidden, 1e-5, 1, None))
        emit(f"N1 rms_norm T={T} hidden=4096", ms, nbytes=2 * T * hidden * 2 + hidden * 2)
        h, hk, d = 32, 8, 128
        q, k = rand_dev(rng, T * h * d * 2), rand_dev(rng, T * hk * d * 2)
        cos, sin = rand_dev(rng, 8192 * 64 * 2), rand_dev(rng, 8192 * 64 * 2)
        pos = ah.DeviceBuffer.from_numpy(rng.integers(0, 8192, T).astype(np.int64))
        ms = timeit(lambda: ah.lib.atoma_rope_qk(q.ptr, k.ptr, cos.ptr, sin.ptr, pos.ptr, T, h, hk, d, h * d, hk * d, 1, 1, None))
        emit(f"N1 rope q+k T={T}", ms, nbytes=2 * T * (h + hk) * d * 2 + 2 * T * (d // 2) * 2)
        # fused RoPE(q, k) + KV-cache write vs the two launches it replaces (rope_qk, reshape_and_cache_flash)
        page, nb = 16, 4096
        v = rand_dev(rng, T * hk * d * 2)
        kc, vc = rand_dev(rng, nb * page * hk * d * 2), rand_dev(rng, nb * page * hk * d * 2)
        slots = ah.DeviceBuffer.from_numpy(rng.permutation(nb * page)[:T].astype(np.int64))
        ms_f = timeit(lambda: ah.lib.atoma_rope_qk_cache(q.ptr, k.ptr, v.ptr, kc.ptr, vc.ptr, slots.ptr, cos.ptr, sin.ptr, pos.ptr, T, h,
                                                         hk, d, h * d, hk * d, hk * d, page * hk * d, page, 1, 1, None))
        def two():
            ah.lib.atoma_rope_qk(q.ptr, k.ptr, cos.ptr, sin.ptr, pos.ptr, T, h, hk, d, h * d, hk * d, 1, 1, None)
            ah.lib.reshape_and_cache_flash(k.ptr, v.ptr, kc.ptr, vc.ptr, slots.ptr, page * hk * d, T, hk, d, page, hk * d, hk * d, 1, None)
        ms_2 = timeit(two)
        emit(f"N1+K4 fused rope q+k + cache write T={T}", ms_f, nbytes=2 * T * (h + hk) * d * 2 + 3 * T * hk * d * 2 + 2 * T * (d // 2) * 2 + 16 * T,
             two_launches_ms=round(ms_2, 4))


def bench_sampling():
    rng = np.random.default_rng(5)
    vocab = 128256
    for B, dt, elt, name in ((256, 2, 4, "f32"), (256, 1, 2, "bf16"), (1, 2, 4, "f32")):
        logits = rand_dev(rng, B * vocab * elt)
        idx, val = ah.DeviceBuffer(B * 4), ah.DeviceBuffer(B * 4)
        ms = timeit(lambda: ah.lib.atoma_argmax_rows(logits.ptr, B, vocab, vocab, dt, idx.ptr, val.ptr, None))
        emit(f"S1 argmax_rows B={B} vocab={vocab} {name}", ms, nbytes=B * vocab * elt + 8 * B)
        k = 50
        tv, ti = ah.DeviceBuffer(B * k * 4), ah.DeviceBuffer(B * k * 4)
        ms = timeit(lambda: ah.lib.atoma_topk_rows(logits.ptr, B, vocab, vocab, dt, k, tv.ptr, ti.ptr, None))
        emit(f"S2 topk_rows k={k} B={B} vocab={vocab} {name}", ms, nbytes=B * vocab * elt + 8 * B * k,
             note="algorithmic bytes = one read of the logits; the kernel reads them twice (second pass from MALL / L2)")


def bench_graph():
    """A decode step's attention-path kernels for 32 layers (RMSNorm, fused RoPE + cache write, paged decode) at small
    batch, launched eagerly and replayed from a hipGraph: the launch-bound end of the path."""
    rng = np.random.default_rng(7)
    layers, hidden, page = 32, 4096, 16
    st = ah.Stream()
    for name, B, S, h, hk in (("8B B=1 S=4096", 1, 4096, 32, 8), ("70B TP=8 shard B=64 S=4096", 64, 4096, 8, 1)):
        d = 128
        pps = S // page
        nb = B * pps + 4
        kc, vc = rand_dev(rng, nb * page * hk * d * 2), rand_dev(rng, nb * page * hk * d * 2)     # one layer's cache, reused
        bt = ah.DeviceBuffer.from_numpy(rng.permutation(nb)[:B * pps].astype(np.int32).reshape(B, pps))
        lens = ah.DeviceBuffer.from_numpy(np.full(B, S, np.int32))
        x, w, y = rand_dev(rng, B * hidden * 2), rand_dev(rng, hidden * 2), ah.DeviceBuffer(B * hidden * 2)
        q, k, v = rand_dev(rng, B * h * d * 2), rand_dev(rng, B * hk * d * 2), rand_dev(rng, B * hk * d * 2)
        o = ah.DeviceBuffer(B * h * d * 2)
        cos, sin = rand_dev(rng, 8192 * 64 * 2), rand_dev(rng, 8192 * 64 * 2)
        pos = ah.DeviceBuffer.from_numpy(np.full(B, S - 1, np.int64))
        slots = ah.DeviceBuffer.from_numpy((bt.numpy(np.int32, (B, pps))[:, -1].astype(np.int64) * page + page - 1))

        def step():
            for _ in range(layers):
                ah.lib.atoma_rms_norm(x.ptr, w.ptr, y.ptr, B, hidden, hidden, hidden, 1e-5, 1, st.s)
                ah.lib.atoma_rope_qk_cache(q.ptr, k.ptr, v.ptr, kc.ptr, vc.ptr, slots.ptr, cos.ptr, sin.ptr, pos.ptr, B, h, hk, d,
                                           h * d, hk * d, hk * d, page * hk * d, page, 1, 1, st.s)
                ah.run_mha(q, kc, vc, o, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=pps * page, softmax_scale=d ** -0.5, is_bf16=1,
                           q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d),
                           v_strides=(page * hk * d, hk * d, d), cu_seqlens_k=lens, is_seqlens_k_cumulative=False, block_table=bt,
                           block_table_batch_stride=pps, page_block_size=page, force_split_kernel=True, unpadded_lse=False, stream=st.s)

        def timed(fn, iters=10):
            fn(); st.synchronize()
            a, b = ah.Event(), ah.Event()
            a.record(st.s)
            for _ in range(iters):
                fn()
            b.record(st.s)
            b.synchronize()
            return a.elapsed_ms(b) / iters
        ms_eager = timed(step)
        with ah.Graph.capture(st) as g:
            step()
        ms_graph = timed(g.launch)
        rec = {"workload": f"G1 decode attention path x{layers} layers, {name}", "ms_eager": round(ms_eager, 4), "ms_graph": round(ms_graph, 4),
               "us_per_layer_eager": round(ms_eager / layers * 1e3, 1), "us_per_layer_graph": round(ms_graph / layers * 1e3, 1),
               "launches_per_layer": "rms_norm + rope_qk_cache + paged decode + combine"}
        print(json.dumps(rec), flush=True)


def bench_linear():
    """f.1: the projections of a decode step at small batch (Llama-3.1-8B shapes): a stream over the weights."""
    rng = np.random.default_rng(8)
    for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)):
        w = rand_dev(rng, N * K * 2)
        for B in (1, 4, 8, 16, 64):
            x, y = rand_dev(rng, B * K * 2), ah.DeviceBuffer(B * N * 2)
            ms = timeit(lambda: ah.lib.atoma_linear_decode(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L1 linear_decode {name} [{N} x {K}] batch={B}", ms, nbytes=N * K * 2 + B * K * 2 + B * N * 2)
        smax = int(os.environ.get("ATOMA_LINEAR_STREAM_MAX_BATCH", "4"))
        for B in (1, 4, 8, 16, 32, 64, 128, 256, 2048):        # atoma_linear: streaming kernel up to smax rows, vendor GEMM above; 2048 = a prefill chunk
            if B <= min(smax, 64):
                continue                                  # measured above
            x, y = rand_dev(rng, B * K * 2), ah.DeviceBuffer(B * N * 2)
            route = "streaming route" if B <= min(smax, 64) else "hipBLASLt route"
            ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)   # first use: the library's candidates are timed here
            ah.synchronize()
            time.sleep(0.2)                                                        # let the clocks settle after that burst
            ms = timeit(lambda: ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L2 linear ({route}) {name} [{N} x {K}] batch={B}", ms, nbytes=N * K * 2 + B * K * 2 + B * N * 2, flops=2 * B * N * K)
        w.free()


def bench_linear_mid():
    """f.1 at 17..64 rows: atoma_linear_decode (linear_mid_kernel: x staged through LDS once per workgroup) against the vendor
    GEMM (atoma_linear above ATOMA_LINEAR_STREAM_MAX_BATCH) on the Llama-3.1-8B layer shapes and on the shapes of one rank of a
    Llama-3.1-70B TP = 8 job; also the fused epilogues (residual / SiLU.up) against projection + separate op."""
    rng = np.random.default_rng(8)
    shapes = (("8B qkv", 6144, 4096), ("8B o", 4096, 4096), ("8B gate_up", 28672, 4096), ("8B down", 4096, 14336),
              ("70B/8 qkv", 1280, 8192), ("70B/8 o", 8192, 1024), ("70B/8 gate_up", 7168, 8192), ("70B/8 down", 8192, 3584))
    for name, N, K in shapes:
        w = rand_dev(rng, N * K * 2)
        for B in (32, 64):
            x, y, r = rand_dev(rng, B * K * 2), ah.DeviceBuffer(B * N * 2), rand_dev(rng, B * N * 2)
            nbytes = N * K * 2 + B * K * 2 + B * N * 2
            ms = timeit(lambda: ah.lib.atoma_linear_decode(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L3 own kernel (tile / mid) {name} [{N} x {K}] batch={B}", ms, nbytes=nbytes)
            ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)
            ah.synchronize()
            time.sleep(0.1)
            ms_lt = timeit(lambda: ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L3 vendor GEMM {name} [{N} x {K}] batch={B}", ms_lt, nbytes=nbytes)
            if "gate_up" in name:
                ms_f = timeit(lambda: ah.lib.atoma_linear_decode_silu_mul(x.ptr, w.ptr, y.ptr, B, K, N // 2, K, K, N // 2, 1, None))
                emit(f"L3 own kernel + SiLU.up epilogue {name} batch={B}", ms_f, nbytes=N * K * 2 + B * K * 2 + B * N)
            elif "qkv" not in name:
                ms_f = timeit(lambda: ah.lib.atoma_linear_decode_residual(x.ptr, w.ptr, r.ptr, y.ptr, B, K, N, K, K, N, N, 1, None))
                emit(f"L3 own kernel + residual epilogue {name} batch={B}", ms_f, nbytes=nbytes + B * N * 2)
        w.free()


def bench_linear_big():
    """f.1 at 65..256 rows: atoma_linear_decode (linear_big_kernel) against the vendor GEMM on the Llama-3.1-8B layer shapes."""
    rng = np.random.default_rng(8)
    for name, N, K in (("8B qkv", 6144, 4096), ("8B o", 4096, 4096), ("8B gate_up", 28672, 4096), ("8B down", 4096, 14336), ("8B lm_head", 128256, 4096)):
        w = rand_dev(rng, N * K * 2)
        for B in (128, 256):
            x, y, r = rand_dev(rng, B * K * 2), ah.DeviceBuffer(B * N * 2), rand_dev(rng, B * N * 2)
            nbytes, flops = N * K * 2 + B * K * 2 + B * N * 2, 2 * B * N * K
            ms = timeit(lambda: ah.lib.atoma_linear_decode(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L4 linear_big {name} [{N} x {K}] batch={B}", ms, nbytes=nbytes, flops=flops)
            ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None)
            ah.synchronize()
            time.sleep(0.1)
            ms_lt = timeit(lambda: ah.lib.atoma_linear(x.ptr, w.ptr, y.ptr, B, K, N, K, K, N, 1, None))
            emit(f"L4 vendor GEMM {name} [{N} x {K}] batch={B}", ms_lt, nbytes=nbytes, flops=flops)
            if "gate_up" in name:
                ms_f = timeit(lambda: ah.lib.atoma_linear_decode_silu_mul(x.ptr, w.ptr, y.ptr, B, K, N // 2, K, K, N // 2, 1, None))
                emit(f"L4 linear_big + SiLU.up epilogue {name} batch={B}", ms_f, nbytes=N * K * 2 + B * K * 2 + B * N, flops=flops)
        w.free()


def bench_step():
    """C3-lite: one whole Llama-3.1-8B decode step on the device (tools/decode_step.py) at batch 1 and 16, context 4096,
    synthetic bf16 weights, eager and replayed from a hipGraph.  Bytes = weights read once + the KV cache of the batch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import decode_step as DS
    rng = np.random.default_rng(9)
    c = DS.LLAMA_3_1_8B
    cos, sin = DS.rope_tables(c)
    w = dict(emb=rand_dev(rng, c.vocab * c.hidden * 2), lm_head=rand_dev(rng, c.vocab * c.hidden * 2),
             norm_f=rand_dev(rng, c.hidden * 2), cos=ah.DeviceBuffer.from_numpy(cos), sin=ah.DeviceBuffer.from_numpy(sin),
             norm1=[rand_dev(rng, c.hidden * 2) for _ in range(c.layers)], norm2=[rand_dev(rng, c.hidden * 2) for _ in range(c.layers)],
             wqkv=[], wo=[], wgu=[], wdown=[])
    scaled = lambda n, k: rand_dev(rng, n * k * 2)         # N(0,1) weights: magnitudes do not matter for timing
    for _ in range(c.layers):
        w["wqkv"].append(scaled(c.qkv, c.hidden)); w["wo"].append(scaled(c.hidden, c.h * c.d))
        w["wgu"].append(scaled(2 * c.inter, c.hidden)); w["wdown"].append(scaled(c.hidden, c.inter))
    weight_bytes = 2 * (2 * c.vocab * c.hidden + c.layers * (c.qkv * c.hidden + c.hidden * c.h * c.d + 3 * c.inter * c.hidden))
    st = ah.Stream()
    S = 4096
    cases = ((1, False), (16, False), (64, False), (256, False), (256, True))
    if os.environ.get("ATOMA_BENCH_STEP_CASES"):      # e.g. "256r,64": a subset, for profiling one step shape
        cases = tuple((int(t.rstrip("r")), t.endswith("r")) for t in os.environ["ATOMA_BENCH_STEP_CASES"].split(","))
    for B, ragged in cases:
        pps = S // c.page + 1
        step = DS.DecodeStep(c, B, B * pps + 2, pps, w, st, fused_epilogues=True)
        bt = rng.permutation(B * pps).astype(np.int32).reshape(B, pps)
        # ragged: the C3 trace mid-flight -- prompts of 2048 tokens, sequences 0..511 tokens into their generation
        ctx = rng.integers(2048, 2560, B) if ragged else np.full(B, S)
        slots = bt[np.arange(B), ctx // c.page].astype(np.int64) * c.page + ctx % c.page
        step.set_inputs(rng.integers(0, c.vocab, B), ctx, slots, ctx + 1, bt)

        def timed(fn, iters=5):
            fn(); st.synchronize()
            a, b = ah.Event(), ah.Event()
            a.record(st.s)
            for _ in range(iters):
                fn()
            b.record(st.s)
            b.synchronize()
            return a.elapsed_ms(b) / iters
        ms_eager = timed(step.run)
        with ah.Graph.capture(st) as g:
            step.run()
        ms_graph = timed(g.launch)
        nbytes = weight_bytes - 2 * c.vocab * c.hidden + 2 * int((ctx + 1).sum()) * c.hk * c.d * 2 * c.layers   # embedding table: B rows only
        emit(f"E1 Llama-3.1-8B decode step, batch={B}, context {'U[2048,2560) (C3 mid-trace)' if ragged else S} (hipGraph replay)", ms_graph, nbytes=nbytes, ms_eager=round(ms_eager, 4),
             tokens_per_s=round(B / (ms_graph * 1e-3)), projections="weight-streaming kernel with fused residual / SiLU.up epilogues" if step.fused else "hipBLASLt + separate residual / SiLU.up kernels")
        del step


def bench_prep():
    """f.2: the metadata of one engine step (worker.rs:224-460) for 256 decode sequences at context ~2304: packed on the
    host and sent with one copy from pinned memory (atoma_prepare_inputs), against the reference's pattern -- one
    synchronous H2D copy per tensor plus one per sequence for the padded block table, from pageable memory."""
    import time
    import ctypes as C
    rng = np.random.default_rng(12)
    B, page = 256, 16
    seqs = []
    nxt = 0
    for _ in range(B):
        L = int(rng.integers(512, 4097))
        n = (L + page - 1) // page
        seqs.append(dict(is_prompt=False, tokens=rng.integers(0, 128256, L), chunk=1, block_table=np.arange(nxt, nxt + n)))
        nxt += n
    arr, keep = ah.make_seq_descs(seqs)
    lay = ah.BatchLayout()
    assert ah.lib.atoma_prepare_inputs(arr, B, page, 0, 0, None, 0, None, 0, C.byref(lay), None) == 0
    host = ah.lib.atoma_host_alloc(lay.total_bytes)
    dev = ah.DeviceBuffer(lay.total_bytes)
    st = ah.Stream()

    def packed():
        assert ah.lib.atoma_prepare_inputs(arr, B, page, 0, 0, host, lay.total_bytes, dev.ptr, lay.total_bytes, C.byref(lay), st.s) == 0
        st.synchronize()
    for _ in range(3):
        packed()
    t0 = time.perf_counter()
    for _ in range(50):
        packed()
    us_packed = (time.perf_counter() - t0) / 50 * 1e6
    # the reference's pattern with the same contents: tokens, positions, slots, seq_lens, context_lens, query_lens (+ 2 zeros
    # tensors and 2 cumsums on the device, not counted), and B block-table rows of max_len entries each
    ref = BO_prepare(seqs, page)
    small = [ref[k] for k in ("input_tokens", "input_positions", "slot_mapping", "seq_lens", "context_lens", "query_start_loc", "seq_start_loc")]
    rows = [np.ascontiguousarray(r) for r in ref["block_tables"]]
    dsmall = [ah.DeviceBuffer(a.nbytes) for a in small]
    drows = ah.DeviceBuffer(ref["block_tables"].nbytes)
    row_bytes = rows[0].nbytes

    def per_tensor():
        for a, d in zip(small, dsmall):
            ah.hip.hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, ah.H2D)
        for i, r in enumerate(rows):
            ah.hip.hipMemcpy(drows.ptr + i * row_bytes, r.ctypes.data, row_bytes, ah.H2D)
    for _ in range(3):
        per_tensor()
    t0 = time.perf_counter()
    for _ in range(20):
        per_tensor()
    us_ref = (time.perf_counter() - t0) / 20 * 1e6
    print(json.dumps({"workload": f"H1 batch prep, {B} decode sequences, context 512..4096: pack + ONE pinned H2D copy", "us_per_step": round(us_packed, 1),
                      "bytes": int(lay.total_bytes), "reference_pattern_us": round(us_ref, 1), "reference_pattern": f"{len(small)} + {B} synchronous pageable H2D copies (host-side packing not counted)",
                      "speedup": round(us_ref / us_packed, 1)}), flush=True)
    ah.lib.atoma_host_free(host)


def BO_prepare(seqs, page):
    """The contents of the emulated reference copies: the library's own host-side packing (nothing under tools/ imports oracle/)"""
    return ah.prepare_inputs_host(seqs, page)[0]


def bench_swap():
    rng = np.random.default_rng(4)
    L, page_bytes, nb = 32, 16 * 8 * 128 * 2, 2048
    gpu = [rand_dev(rng, nb * page_bytes) for _ in range(2 * L)]
    for pinned in (True, False):
        if pinned:
            host_ptrs = [ah.lib.atoma_host_alloc(nb * page_bytes) for _ in range(2 * L)]
        else:
            keep = [np.zeros(nb * page_bytes, np.uint8) for _ in range(2 * L)]
            host_ptrs = [a.ctypes.data for a in keep]
        gp = (C.c_void_p * (2 * L))(*[b.ptr for b in gpu])
        hp = (C.c_void_p * (2 * L))(*host_ptrs)
        for n in (16, 256, 2048):
            m = np.stack([rng.permutation(nb)[:n], rng.permutation(nb)[:n]], 1).astype(np.int64)
            for kind, s, d_, label in ((2, gp, hp, "gpu->cpu"), (1, hp, gp, "cpu->gpu")):
                def run():
                    rc = ah.lib.atoma_swap_blocks_multi(s, d_, 2 * L, m.ctypes.data, n, page_bytes, kind, None)
                    assert rc == 0, ah.last_error()
                ms = timeit(run, iters=3, warmup=1)
                emit(f"C5 swap {label} {'pinned' if pinned else 'pageable'} pages={n} (x{2 * L} tensors)", ms,
                     nbytes=2 * L * page_bytes * n, note="bytes moved one way over PCIe; frac_hbm is not meaningful here")
        if pinned:
            for p in host_ptrs:
                ah.lib.atoma_host_free(p)


if __name__ == "__main__":
    ah.set_device(0)
    which = sys.argv[1:] or ["decode", "prefill", "prefill_paged", "cache", "norm", "sampling", "linear", "graph", "step", "prep", "swap"]
    for w in which:
        globals()["bench_" + w]()
