"""Differential fuzzing of run_mha against the oracle (test infrastructure: the oracle is the checker, the C-ABI library the thing checked).

One case = one call of the reference's `run_mha` boundary (csrc/src/ffi.rs:3-102 -> include/atoma_hip.h) with a randomly drawn shape from the
space the reference's three entry families span (csrc/src/lib.rs:392-2105):

  prefill        flash_attn_varlen: ragged q / k lengths (empty sequences, Lq > Lk, a cached prefix in front of the queries), causal or not, ALiBi
  paged_prefill  flash_attn_varlen_with_block_table: the same over a paged cache (pages of 16 .. 256 tokens, a shuffled block table)
  kv_cache       flash_attn_kv_cache: 1 .. 6 query rows per sequence over a paged or contiguous cache with per-sequence lengths, ALiBi
  forward        the operator itself, FlashAttention::forward (models/src/flash_attention.rs:281-474 -> atoma_flash_attention_forward): a mixed batch of
                 prompts (with or without a cached prefix -- the prefix route is NON-causal in the reference, SURVEY B/Q3, mirrored) and decode tokens;
                 cache write bit-exact, every output row against the definition

head sizes 8 .. 256 (weighted towards 64 / 128), 1 .. 8 kv heads x groups of 1 .. 8 q heads, bf16 / f16.  Every row is held to the f32 definition
(oracle/attn_oracle.py attend_rows) with the tolerance of tests/util.py (1e-3 + 1 ulp from 512 visible keys on, the P-rounding bound below), rows
without a visible key must be exact zeros with LSE = +inf, all other LSE values must match to 1e-4, and nothing may be left unwritten.

    python tests/fuzz_parity.py --seconds 600 [--seed 0] [--kinds prefill,paged_prefill,kv_cache]      (prints one JSON line; failures carry their seed)
    tests/test_fuzz_parity_gpu.py runs a fixed set of seeds in the GPU suite.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import attn_oracle as A  # noqa: E402
from oracle.halfs import F16, BF16, to_f32, from_f32  # noqa: E402
from util import rand_half, make_paged_cache, ulp_tol, attn_atol  # noqa: E402

KINDS = ("prefill", "paged_prefill", "kv_cache")
LONG_BASE = 4 * 10 ** 6     # ... and from here on prompts of 500 .. 5000 tokens (a persistent prefill workgroup walks many blocks; sampled rows against the definition)
STRIDE_BASE = 3 * 10 ** 6   # ... and from here on the ordinary kinds with padded row strides of q / k / v / o (slices of a fused q/k/v projection, llama.rs:269-303) and
                            # seqlen_q / seqlen_k arguments larger than any sequence (an engine passes its configured maximum)
DECODE_BASE = 2 * 10 ** 6   # ... and from here on decode batches of 64 .. 512 sequences (the balanced line, the paired kernel on a length hint, kv-head pairs at d = 64)
FORWARD_BASE = 10 ** 6      # seeds from here on are "forward" cases (added after the first campaigns: a seed below keeps naming the case it always named)
HEAD_SIZES = [8, 32, 64, 64, 64, 96, 128, 128, 128, 128, 160, 192, 224, 256]


def draw(seed, kinds=KINDS):
    """the case of `seed`: a dict of plain ints / lists (what a failure report prints)"""
    if seed >= LONG_BASE:
        rng = np.random.default_rng(seed)
        hk, g = int(rng.choice([1, 2, 8])), int(rng.choice([1, 4]))
        d = int(rng.choice([128, 128, 128, 64, 96]))
        B = int(rng.choice([1, 2, 3, 5]))
        top = int(rng.choice([1800, 3000, 5000]))
        top = min(top, (96 << 20) // (B * hk * g * d * 2))
        lq = rng.integers(top // 8, top + 1, B)
        lq[rng.integers(0, B)] = top
        mode = int(rng.integers(4))
        lk = lq.copy() if mode <= 1 else (lq + rng.integers(0, top // 2, B) * rng.integers(0, 2, B) if mode == 2 else rng.integers(0, top + 1, B))
        c = dict(seed=int(seed), kind=("prefill", "paged_prefill")[int(rng.integers(2))], d=d, hk=hk, h=hk * g, dtype=int(rng.choice([BF16, BF16, F16])),
                 causal=bool(rng.integers(4) != 0), alibi=False, scale=float(d ** -0.5), B=B, lens_q=[int(x) for x in lq], lens_k=[int(x) for x in lk], sample=True)
        if c["kind"] == "paged_prefill":
            c["page"] = int(rng.choice([16, 16, 32, 256]))
        return c
    if seed >= STRIDE_BASE:
        c = draw(seed - STRIDE_BASE + 500_000, KINDS)
        r2 = np.random.default_rng(seed)
        c.update(seed=int(seed), qpad=int(r2.choice([0, 8, 64, 2 * c["hk"] * c["d"]])), opad=int(r2.choice([0, 8, 64])), kpad=int(r2.choice([0, 8, 64, c["h"] * c["d"]])),
                 max_extra=int(r2.choice([0, 1, 37, 300])))
        return c
    rng = np.random.default_rng(seed)
    kind = kinds[int(rng.integers(len(kinds)))]
    if seed >= FORWARD_BASE:
        kind = "forward"
    if seed >= DECODE_BASE:
        hk, g = (int(x) for x in rng.choice([(8, 4), (8, 4), (4, 4), (1, 8), (8, 8), (8, 1), (2, 2), (4, 8)]))
        B = int(rng.choice([64, 100, 130, 200, 250, 256, 257, 320, 400, 512]))
        d = int(rng.choice([64, 128, 128, 128]))
        top = max(20, min(int(rng.choice([60, 300, 1000, 3000])), (1 << 23) // (B * hk * g)))
        lens = rng.integers(max(0, top - int(rng.choice([top, top // 2, top // 8 + 1]))), top + 1, B)
        if rng.integers(5) == 0:
            lens[:] = top
        return dict(seed=int(seed), kind="kv_cache", d=d, hk=hk, h=hk * g, dtype=int(rng.choice([BF16, BF16, F16])), causal=False, alibi=False, scale=float(d ** -0.5),
                    B=B, sq=1, lens_k=[int(x) for x in lens], page=int(rng.choice([16, 16, 32])), hint=bool(rng.integers(2)))
    d = int(rng.choice(HEAD_SIZES))
    hk = int(rng.choice([1, 1, 2, 3, 4, 8]))
    g = int(rng.choice([1, 1, 2, 3, 4, 5, 8]))
    c = dict(seed=int(seed), kind=kind, d=d, hk=hk, h=hk * g, dtype=int(rng.choice([BF16, BF16, F16])), causal=bool(rng.integers(2)),
             alibi=bool(rng.integers(5) == 0), scale=float(d ** -0.5 * rng.choice([1.0, 1.0, 0.5, 1.7])))
    budget = 1 << 22                                    # ~ score elements per head the oracle computes per case (keeps a case under a second or two)
    if kind == "forward":
        c["d"] = int(rng.choice([64, 128, 128, 96, 256]))
        c["scale"] = float(c["d"] ** -0.5)
        n_pre = int(rng.choice([0, 1, 2, 4]))
        top = int(rng.choice([5, 40, 200, 600]))
        top = max(2, min(top, int((budget / (max(n_pre, 1) * c["h"])) ** 0.5)))
        n_dec = int(rng.choice([0, 1, 3, 17, 64])) if n_pre else int(rng.choice([1, 3, 17, 64]))
        dtop = int(rng.choice([5, 60, 700, 2000]))
        dtop = max(2, min(dtop, budget // (max(n_dec, 1) * c["h"])))
        prefix = bool(n_pre and rng.integers(2))
        c.update(page=int(rng.choice([16, 32])), lens_q=[int(x) for x in rng.integers(1, top + 1, n_pre)],
                 prefix=[int(x) for x in (rng.integers(0, 100, n_pre) * rng.integers(0, 2, n_pre) if prefix else np.zeros(n_pre, np.int64))],
                 use_prefix_route=prefix, dec_lens=[int(x) for x in rng.integers(1, dtop + 1, n_dec)], alibi=bool(prefix and rng.integers(4) == 0) or (not n_pre and rng.integers(5) == 0))
        c["alibi"] = bool(c["alibi"])
        return c
    if kind == "kv_cache":
        B = int(rng.choice([1, 2, 5, 16, 33, 70]))
        top = int(rng.choice([30, 200, 900, 2500]))
        top = max(16, min(top, budget // (B * c["h"])))
        lens = rng.integers(0, top + 1, B)
        lens[rng.integers(0, B)] = top
        if rng.integers(4) == 0:
            lens[:] = top
        c.update(B=B, sq=int(rng.choice([1, 1, 1, 2, 3, 6])), lens_k=[int(x) for x in lens], page=int(rng.choice([0, 16, 16, 32, 64, 128, 256])))
        return c
    B = int(rng.choice([1, 2, 3, 6, 11]))
    top = int(rng.choice([20, 130, 300, 700, 1500]))
    top = max(8, min(top, int((budget / (B * c["h"])) ** 0.5)))
    lq = rng.integers(0, top + 1, B)
    lq[rng.integers(0, B)] = top
    mode = int(rng.integers(4))                          # 0/1: keys = queries; 2: a cached prefix in front; 3: unrelated key lengths (Lq > Lk, no keys)
    if mode <= 1:
        lk = lq.copy()
    elif mode == 2:
        lk = lq + rng.integers(0, top + 1, B) * rng.integers(0, 2, B)
    else:
        lk = rng.integers(0, top + 1, B)
    c.update(B=B, lens_q=[int(x) for x in lq], lens_k=[int(x) for x in lk])
    if kind == "paged_prefill":
        c["page"] = int(rng.choice([16, 16, 32, 64, 128, 256]))
    return c


# Rounding the kernels do and the f32 definition does not (the reference's CUDA kernel does the first too): P goes to the storage type before P.V
# (softmax.h + the bf16 MMA: up to 2^-8 relative per term, bf16), and the hand-scheduled prefill kernel's "fast" blocks round Q.scale.log2(e) once
# (another ~2^-8 on a probability).  With the softmax mass spread over hundreds of keys both vanish in the sum and BASELINE.json's
# 1e-3 holds -- tests/util.py draws that line at 512 visible keys, which is right for N(0,1) scores at scale d^-1/2 but not for a sharper scale or
# ALiBi (a few recent keys carry the mass whatever the length).  So a row that misses the simple line gets the bound its OWN probabilities give:
#   |got - ref| <= 1e-3 + 1 ulp + min(E_MAX . sum_i p_i |v_i|,  6 . E_SIG . sqrt(sum_i p_i^2 v_i^2))       (worst case, 6 sigma of independent roundings)
E_MAX = {BF16: 2.0 ** -7, F16: 2.0 ** -10}
E_SIG = {BF16: 2.5e-3, F16: 3.2e-4}
LSE_ATOL_FAST = 2.5e-3      # LSE of the fast blocks (d = 128 prefill kernels, rows with >= 512 keys): off by ~1e-3 at scale d^-1/2, documented in DESIGN.md 4.2 (nobody
                            # reads it: SURVEY Q4); a scale f times sharper moves it f^2 times further (larger scores, fewer keys to average over): 3.5e-3 .. 6e-3 at f = 1.7


def p_bounds(qf, kf, vf, scale, causal, alibi):
    """per (row, head, dim): sum_i p_i |v_i| and sqrt(sum_i p_i^2 v_i^2) of the f32 definition (attend_rows' masks and ALiBi)"""
    Lq, h, d = qf.shape
    Lk, hk, _ = kf.shape
    g = h // hk
    rows, cols, shift = np.arange(Lq)[:, None], np.arange(Lk)[None, :], Lk - Lq
    ab, sq = np.zeros((Lq, h, d), np.float32), np.zeros((Lq, h, d), np.float32)
    for head in range(h):
        s = (qf[:, head] @ kf[:, head // g].T) * np.float32(scale)
        if alibi is not None:
            s = s - np.float32(alibi[head]) * np.abs(rows + shift - cols).astype(np.float32)
        if causal:
            s = np.where(cols <= rows + shift, s, -np.inf)
        m = s.max(axis=1, keepdims=True)
        p = np.exp(s - np.where(np.isfinite(m), m, 0)).astype(np.float32)
        p /= np.maximum(p.sum(axis=1, keepdims=True), np.float32(1e-30))
        v = vf[:, head // g]
        ab[:, head], sq[:, head] = p @ np.abs(v), np.sqrt((p * p) @ (v * v))
    return ab, sq


def _check(out, lse_rows, ref, ref_lse, visible, dtype, what, bounds=None, fast_lse=0.0):
    """out / ref: uint16 [rows, h, d]; lse_rows / ref_lse: f32 [h, rows]; visible: keys each row sees, [rows]; bounds() -> p_bounds of the sequence
    (called only when a row misses the simple line).  Returns (finding or None, rows that needed their own bound)."""
    got, want = to_f32(out, dtype), to_f32(ref, dtype)
    if not np.isfinite(got).all():
        return f"{what}: non-finite output ({(~np.isfinite(got)).sum()} elements; unwritten rows read as NaN)", 0
    dead = visible <= 0
    if dead.any() and (out[dead].any() or not np.isposinf(lse_rows[:, dead]).all()):
        return f"{what}: rows without a visible key must be zeros with LSE = +inf", 0
    live = ~dead
    if not live.any():
        return None, 0
    err = np.abs(got - want)
    simple = ulp_tol(want, dtype, 1e-3)
    simple[visible < 512] = ulp_tol(want[visible < 512], dtype, attn_atol(dtype, 1))
    bad = (err > simple) & live[:, None, None]
    soft = 0
    if bad.any():
        ab, sq = bounds()
        own = ulp_tol(want, dtype, 1e-3) + np.minimum(E_MAX[dtype] * ab, 6 * E_SIG[dtype] * sq)
        soft = int(bad.any(axis=(1, 2)).sum())
        bad &= err > own
        if bad.any():
            i = np.unravel_index(np.where(bad, err - own, -1).argmax(), err.shape)
            return (f"{what}: {bad.sum()} of {bad.size} elements beyond the bound of their own probabilities; worst at (row, head, dim) = {tuple(int(x) for x in i)}: "
                    f"got {got[i]:.6f}, ref {want[i]:.6f}, bound {own[i]:.2e}, visible keys {int(visible[i[0]])}"), soft
    lse_atol = np.where(visible[live] >= 512, LSE_ATOL_FAST * max(1.0, fast_lse) ** 2, 1e-4) if fast_lse else 1e-4    # fast_lse: scale / d^-1/2 where the fast blocks run, else 0
    dl = np.abs(lse_rows[:, live] - ref_lse[:, live])
    if (dl > lse_atol + 1e-4 * np.abs(ref_lse[:, live])).any():
        return f"{what}: LSE differs by up to {dl.max():.3e}", soft
    return None, soft


def _widen(rng, a, pad, dtype):
    """a [..., heads, d] -> ([..., heads * d + pad] with random padding, row width): rows as slices of a wider buffer"""
    flat = a.reshape(a.shape[:-2] + (a.shape[-2] * a.shape[-1],))
    if not pad:
        return np.ascontiguousarray(flat), flat.shape[-1]
    w = rand_half(rng, flat.shape[:-1] + (flat.shape[-1] + pad,), dtype)
    w[..., :flat.shape[-1]] = flat
    return w, flat.shape[-1] + pad


def _read_out(do, rows_shape, h, d, orow):
    """the output buffer [rows..., orow] -> [rows..., h, d]; the padding between rows must still hold the poison"""
    a = do.numpy(np.uint16, rows_shape + (orow,))
    if orow > h * d and not (a[..., h * d:] == 0xFFFF).all():
        raise RuntimeError("padding between output rows was written")
    return np.ascontiguousarray(a[..., :h * d]).reshape(rows_shape + (h, d))


def run_case(gpu, c):
    """(finding or None, rows held to the bound of their own probabilities)"""
    g = case_steps(gpu, c)
    try:
        next(g)
        next(g)
    except StopIteration as e:
        return e.value
    raise AssertionError("case_steps yields once")


def run_burst(gpu, cases):
    """All cases LAUNCHED back to back on the stream -- no synchronisation, no host read in between: the library's grow-only scratch block, its plan tables
    and arrival counters are handed from one launch to the next while the earlier ones still run -- then every result checked.  Returns [(case, finding)]."""
    gens = []
    for c in cases:
        g = case_steps(gpu, c)
        try:
            next(g)
            gens.append((c, g, None))
        except StopIteration as e:                      # a case that launches nothing (no query rows)
            gens.append((c, None, e.value))
        except (RuntimeError, AssertionError) as e:
            gens.append((c, None, (f"raised {type(e).__name__}: {str(e)[:300]}", 0)))
    out = []
    for c, g, res in gens:
        if g is not None:
            try:
                next(g)
                res = ("case_steps yields once", 0)
            except StopIteration as e:
                res = e.value
            except (RuntimeError, AssertionError) as e:
                res = (f"raised {type(e).__name__}: {str(e)[:300]}", 0)
        if res[0]:
            out.append((c, res[0]))
    return out


def run_threads(gpu, cases, threads=3):
    """The cases dealt to `threads` host threads, each with its own stream, running concurrently (ctypes releases the GIL inside the library): scratch blocks,
    plan tables and arrival counters are per (device, stream); the option table, the length hint and the kernel registry are shared.  Returns [(case, finding)]."""
    import threading
    out, lock = [], threading.Lock()

    def worker(mine):
        gpu.set_device(0)
        stream = gpu.Stream()
        for c in mine:
            c = dict(c, _stream=stream.s)
            try:
                msg = run_case(gpu, c)[0]
            except (RuntimeError, AssertionError) as e:
                msg = f"raised {type(e).__name__}: {str(e)[:300]}"
            if msg:
                with lock:
                    out.append(({k: v for k, v in c.items() if k != "_stream"}, msg))
        stream.synchronize()
    ts = [threading.Thread(target=worker, args=([c for i, c in enumerate(cases) if i % threads == t and c["kind"] != "forward"],)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out


def graph_case(gpu, c):
    """A decode call (one query row, paged cache) CAPTURED into a hipGraph after one eager call, then replayed three times with OTHER lengths in the same block
    tables (what a serving loop does every step: model_executor / worker rebuild the metadata, the graph stays): shorter, longer, ragged where the capture was
    uniform, empty sequences.  Every replay against the definition.  What the host decided at capture time (kernel, KV splits, scratch layout) must be right for
    any lengths the tables can hold.  Returns a finding or None."""
    rng = np.random.default_rng(c["seed"] + (1 << 42))
    d, h, hk, dtype, scale, B, page = c["d"], c["h"], c["hk"], c["dtype"], c["scale"], c["B"], c["page"] or 16
    lens0 = np.asarray(c["lens_k"], np.int32)
    cap = np.maximum(lens0, 1) + rng.integers(0, 3 * page, B)                      # what each sequence's block-table row can hold
    pages = (cap + page - 1) // page
    cap = (pages * page).astype(np.int32)
    nb = int(pages.sum()) + 2
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, cap)
    q = rand_half(rng, (B, 1, h, d), dtype)
    D = gpu.DeviceBuffer
    st = gpu.Stream()
    dq, dk, dv, do = D.from_numpy(q), D.from_numpy(kc), D.from_numpy(vc), D(q.nbytes)
    dlse, dbt, dl = D.zeros((B, h, 1), np.float32), D.from_numpy(np.ascontiguousarray(bt, np.int32)), D.from_numpy(lens0)

    def call():
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=1, seqlen_k=bt.shape[1] * page, softmax_scale=scale, is_bf16=dtype,
                    q_strides=(h * d, h * d, d), o_strides=(h * d, h * d, d), k_strides=(page * hk * d, hk * d, d), v_strides=(page * hk * d, hk * d, d), is_causal=0,
                    cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt, block_table_batch_stride=bt.shape[1], page_block_size=page, softmax_lse=dlse,
                    force_split_kernel=True, unpadded_lse=False, stream=st.s)
    if c.get("hint"):
        gpu.lib.atoma_hint_decode_lengths(int(lens0.min()), int(lens0.max()), B)
    call()                                                                          # eager: sizes the stream's scratch (or atoma_warmup would)
    st.synchronize()
    with gpu.Graph.capture(st) as g:
        call()
    gpu.lib.atoma_hint_decode_lengths(0, 0, 0)
    kernel = (gpu.lib.atoma_last_decode_kernel() or b"").decode()
    qf, kf, vf = to_f32(q, dtype), to_f32(kc, dtype), to_f32(vc, dtype)
    variants = [lens0, (cap * rng.random(B)).astype(np.int32), cap.copy(), np.where(rng.random(B) < 0.3, 0, np.minimum(cap, 1 + rng.integers(0, 40, B))).astype(np.int32)]
    for r, lens in enumerate(variants):
        gpu.hip_check(gpu.hip.hipMemcpy(dl.ptr, np.ascontiguousarray(lens, np.int32).ctypes.data, 4 * B, gpu.H2D), "lengths")
        do.fill_bytes(0xFF)
        g.launch()
        st.synchronize()
        out, lse = do.numpy(np.uint16, (B, 1, h, d)), dlse.numpy()
        for b in range(B):
            L = int(lens[b])
            kb, vb = A.gather_paged(kf, bt[b], L, page), A.gather_paged(vf, bt[b], L, page)
            o, l = A.attend_rows(qf[b], kb, vb, scale, dtype=dtype)
            msg, _ = _check(out[b], lse[b], from_f32(o, dtype), l, np.full(1, L), dtype, f"replay {r} ({kernel}) seq {b} (L={L}, captured at {int(lens0[b])})",
                            lambda: p_bounds(qf[b], kb, vb, scale, False, None))
            if msg:
                return msg
    return None


def case_steps(gpu, c):
    """generator: builds the inputs and launches, yields, then synchronises, reads and checks (its return value = run_case's).
    c["_stream"]: a hipStream_t to launch on (run_threads: one stream per host thread), default the null stream."""
    st = c.get("_stream")
    """(None when the library's answer is the oracle's, else a one-line description; the number of rows that were held to the bound of their own probabilities)"""
    rng = np.random.default_rng(c["seed"] + (1 << 40))
    soft = 0
    d, h, hk, dtype, scale = c["d"], c["h"], c["hk"], c["dtype"], c["scale"]
    alibi = (rng.uniform(0.02, 0.5, h).astype(np.float32) if c["alibi"] else None)
    D = gpu.DeviceBuffer
    da = D.from_numpy(alibi) if alibi is not None else None
    if c["kind"] == "forward":
        return (yield from _run_forward(gpu, c, rng, alibi))
    if c["kind"] == "kv_cache":
        B, sq, page, lens = c["B"], c["sq"], c["page"], np.asarray(c["lens_k"], np.int32)
        causal = c["causal"] or sq == 1
        q = rand_half(rng, (B, sq, h, d), dtype)
        if page:
            nb = int(sum((int(x) + page - 1) // page for x in lens)) + 2
            kc, vc, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lens)
        else:
            S = max(1, int(lens.max()))
            kc, vc, bt = rand_half(rng, (B, S, hk, d), dtype), rand_half(rng, (B, S, hk, d), dtype), None
        qpad, opad, kpad, extra = c.get("qpad", 0), c.get("opad", 0), (c.get("kpad", 0) if not page else 0), c.get("max_extra", 0)
        qw, qrow = _widen(rng, q, qpad, dtype)
        kw, krow = _widen(rng, kc, kpad, dtype)
        vw, _ = _widen(rng, vc, kpad, dtype)
        orow = h * d + opad
        dq, dk, dv, do = D.from_numpy(qw), D.from_numpy(kw), D.from_numpy(vw), D(B * sq * orow * 2)
        do.fill_bytes(0xFF)
        dlse = D.zeros((B, h, sq), np.float32)
        dbt = D.from_numpy(np.ascontiguousarray(bt, np.int32)) if bt is not None else None
        dl = D.from_numpy(lens)
        seqlen_k = (bt.shape[1] * page if bt is not None else kc.shape[1]) + (extra if bt is not None else 0)
        if c.get("hint"):                                   # what atoma_prepare_inputs would have recorded for this batch (a dispatch hint, never a result)
            gpu.lib.atoma_hint_decode_lengths(int(lens.min()), int(lens.max()), B)
        gpu.run_mha(dq, dk, dv, do, b=B, h=h, h_k=hk, d=d, seqlen_q=sq, seqlen_k=seqlen_k, softmax_scale=scale, is_bf16=dtype,
                    q_strides=(sq * qrow, qrow, d), o_strides=(sq * orow, orow, d), k_strides=(kc.shape[1] * krow, krow, d),
                    v_strides=(vc.shape[1] * krow, krow, d), is_causal=int(causal if (sq > 1 or alibi is not None) else 0),      # lib.rs:1629-1631
                    cu_seqlens_k=dl, is_seqlens_k_cumulative=False, block_table=dbt, block_table_batch_stride=0 if bt is None else bt.shape[1],
                    page_block_size=page, alibi_slopes=da, softmax_lse=dlse, force_split_kernel=bt is not None, unpadded_lse=False, stream=st)
        gpu.lib.atoma_hint_decode_lengths(0, 0, 0)
        yield
        gpu.synchronize() if st is None else gpu.hip_check(gpu.hip.hipStreamSynchronize(st), "hipStreamSynchronize")
        out, lse = _read_out(do, (B, sq), h, d, orow), dlse.numpy()
        qf, kf, vf = to_f32(q, dtype), to_f32(kc, dtype), to_f32(vc, dtype)
        for b in range(B):
            L = int(lens[b])
            kb, vb = (A.gather_paged(kf, bt[b], L, page), A.gather_paged(vf, bt[b], L, page)) if page else (kf[b, :L], vf[b, :L])
            o, l = A.attend_rows(qf[b], kb, vb, scale, causal=bool(causal and (sq > 1 or alibi is not None)), alibi_slopes=alibi, dtype=dtype)
            visible = np.minimum(L, np.arange(sq) + L - sq + 1) if (causal and (sq > 1 or alibi is not None)) else np.full(sq, L)
            cz = bool(causal and (sq > 1 or alibi is not None))
            msg, n = _check(out[b], lse[b], from_f32(o, dtype), l, visible, dtype, f"seq {b} (L={L})", lambda: p_bounds(qf[b], kb, vb, scale, cz, alibi), fast_lse=scale * d ** 0.5 if (d == 128 and sq > 1) else 0.0)
            soft += n
            if msg:
                return msg, soft
        return None, soft
    lq, lk = np.asarray(c["lens_q"], np.int64), np.asarray(c["lens_k"], np.int64)
    cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    Tq = int(cu_q[-1])
    q = rand_half(rng, (max(Tq, 1), h, d), dtype)[:Tq]
    bt, page = None, 0
    if c["kind"] == "paged_prefill":
        page = c["page"]
        nb = int(sum((int(x) + page - 1) // page for x in lk)) + 2
        k, v, bt = make_paged_cache(rng, nb, page, hk, d, dtype, lk)
    else:
        Tk = max(1, int(cu_k[-1]))
        k, v = rand_half(rng, (Tk, hk, d), dtype), rand_half(rng, (Tk, hk, d), dtype)
    if Tq == 0:
        return None, 0
    qpad, opad, kpad, extra = c.get("qpad", 0), c.get("opad", 0), (c.get("kpad", 0) if bt is None else 0), c.get("max_extra", 0)
    qw, qrow = _widen(rng, q, qpad, dtype)
    kw, krow = _widen(rng, k, kpad, dtype)
    vw, _ = _widen(rng, v, kpad, dtype)
    orow = h * d + opad
    dq, dk, dv, do = D.from_numpy(qw), D.from_numpy(kw), D.from_numpy(vw), D(Tq * orow * 2)
    do.fill_bytes(0xFF)
    dcq, dck = D.from_numpy(cu_q), D.from_numpy(cu_k)
    dbt = D.from_numpy(np.ascontiguousarray(bt, np.int32)) if bt is not None else None
    dlse = D.zeros((h, Tq), np.float32)
    kstr = (page * hk * d, hk * d, d) if bt is not None else (0, krow, d)
    gpu.run_mha(dq, dk, dv, do, b=c["B"], h=h, h_k=hk, d=d, seqlen_q=int(lq.max()) + extra, seqlen_k=int(max(1, lk.max())) + extra, softmax_scale=scale, is_bf16=dtype,
                q_strides=(0, qrow, d), o_strides=(0, orow, d), k_strides=kstr, v_strides=kstr, is_causal=int(c["causal"]), cu_seqlens_q=dcq,
                cu_seqlens_k=dck, block_table=dbt, block_table_batch_stride=0 if bt is None else bt.shape[1], page_block_size=page, alibi_slopes=da,
                softmax_lse=dlse, force_split_kernel=bt is not None, stream=st)
    yield
    gpu.synchronize() if st is None else gpu.hip_check(gpu.hip.hipStreamSynchronize(st), "hipStreamSynchronize")
    out, lse = _read_out(do, (Tq,), h, d, orow), dlse.numpy()
    if c.get("sample"):                                   # long prompts: nothing unwritten anywhere, and ~40 rows per sequence against the definition
        if not np.isfinite(to_f32(out, dtype)).all():
            return "non-finite output (unwritten rows read as NaN)", soft
        kf_all, vf_all = to_f32(k, dtype), to_f32(v, dtype)
        for b in range(c["B"]):
            q0, Lq, Lk, k0 = int(cu_q[b]), int(lq[b]), int(lk[b]), int(cu_k[b])
            if Lq == 0:
                continue
            kb, vb = (A.gather_paged(kf_all, bt[b], Lk, page), A.gather_paged(vf_all, bt[b], Lk, page)) if bt is not None else (kf_all[k0:k0 + Lk], vf_all[k0:k0 + Lk])
            rows = sorted({0, 1, Lq - 1, Lq // 2} | {r for m in range(256, Lq, 256) for r in (m - 1, m) if rng.integers(3) == 0} | {int(x) for x in rng.integers(0, Lq, 24)})
            for r in rows:
                vis = min(Lk, r + Lk - Lq + 1) if c["causal"] else Lk
                qr = to_f32(q[q0 + r:q0 + r + 1], dtype)
                o, l = A.attend_rows(qr, kb[:max(vis, 0)], vb[:max(vis, 0)], scale, dtype=dtype)
                msg, n = _check(out[q0 + r:q0 + r + 1], lse[:, q0 + r:q0 + r + 1], from_f32(o, dtype), l, np.full(1, vis), dtype, f"seq {b} (Lq={Lq}, Lk={Lk}) row {r}",
                                lambda: p_bounds(qr, kb[:vis], vb[:vis], scale, False, None), fast_lse=scale * d ** 0.5 if d == 128 else 0.0)
                soft += n
                if msg:
                    return msg, soft
        return None, soft
    ref, ref_lse = A.flash_attn_varlen(q, k, v, cu_q, cu_k, scale, c["causal"], dtype, block_table=bt, alibi_slopes=alibi, return_lse=True)
    for b in range(c["B"]):
        q0, q1, Lq, Lk = int(cu_q[b]), int(cu_q[b + 1]), int(lq[b]), int(lk[b])
        if Lq == 0:
            continue
        visible = np.minimum(Lk, np.arange(Lq) + Lk - Lq + 1) if c["causal"] else np.full(Lq, Lk)
        k0, k1 = int(cu_k[b]), int(cu_k[b + 1])

        def bounds():
            kb, vb = ((A.gather_paged(to_f32(k, dtype), bt[b], Lk, page), A.gather_paged(to_f32(v, dtype), bt[b], Lk, page)) if bt is not None
                      else (to_f32(k[k0:k1], dtype), to_f32(v[k0:k1], dtype)))
            return p_bounds(to_f32(q[q0:q1], dtype), kb, vb, scale, c["causal"], alibi)
        msg, n = _check(out[q0:q1], lse[:, q0:q1], ref[q0:q1], ref_lse[b], visible, dtype, f"seq {b} (Lq={Lq}, Lk={Lk})", bounds, fast_lse=scale * d ** 0.5 if d == 128 else 0.0)
        soft += n
        if msg:
            return msg, soft
    return None, soft


def _run_forward(gpu, c, rng, alibi):
    import ctypes as C
    from oracle import cache_oracle as CO
    d, h, hk, dtype, scale, page = c["d"], c["h"], c["hk"], c["dtype"], c["scale"], c["page"]
    lq, pre, dl = np.asarray(c["lens_q"], np.int64), np.asarray(c["prefix"], np.int64), np.asarray(c["dec_lens"], np.int64)
    n_pre_seqs, n_dec = len(lq), len(dl)
    tot = pre + lq
    need = [int((x + page - 1) // page) for x in list(tot) + list(dl)]
    nb = sum(need) + 2
    perm = rng.permutation(nb)
    tables, pos = [], 0
    for n in need:
        tables.append(perm[pos:pos + n])
        pos += n
    pre_bt = np.zeros((max(n_pre_seqs, 1), max([1] + need[:n_pre_seqs])), np.uint32)
    dec_bt = np.zeros((max(n_dec, 1), max([1] + need[n_pre_seqs:])), np.uint32)
    for i in range(n_pre_seqs):
        pre_bt[i, :need[i]] = tables[i]
    for j in range(n_dec):
        dec_bt[j, :need[n_pre_seqs + j]] = tables[n_pre_seqs + j]
    np_tok, T = int(lq.sum()), int(lq.sum()) + n_dec
    slots = np.empty(T, np.int64)
    t = 0
    for i in range(n_pre_seqs):
        p = pre[i] + np.arange(lq[i])
        slots[t:t + lq[i]] = pre_bt[i, p // page].astype(np.int64) * page + p % page
        t += int(lq[i])
    for j in range(n_dec):
        p = int(dl[j]) - 1
        slots[t] = int(dec_bt[j, p // page]) * page + p % page
        t += 1
    q, k, v = rand_half(rng, (T, h, d), dtype), rand_half(rng, (T, hk, d), dtype), rand_half(rng, (T, hk, d), dtype)
    kv = rand_half(rng, (2, nb, page, hk, d), dtype)
    cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.uint32)
    cu_k = np.concatenate([[0], np.cumsum(tot)]).astype(np.uint32)
    D = gpu.DeviceBuffer
    bufs = {n: D.from_numpy(a) for n, a in dict(q=q, k=k, v=v, kv=kv, slots=slots, cu_q=cu_q, cu_k=cu_k, pre_bt=pre_bt, dec_bt=dec_bt,
                                                 dec_lens=dl.astype(np.uint32) if n_dec else np.zeros(1, np.uint32)).items()}
    dout = D(q.nbytes)
    dout.fill_bytes(0xFF)
    fa = gpu.FlashAttention()
    ta = None
    if alibi is not None:
        bufs["alibi"] = D.from_numpy(alibi)
        ta = gpu.tensor(bufs["alibi"], (h,), gpu.F32) if hasattr(gpu, "F32") else None
        if ta is None:
            return None, 0
    if gpu.lib.atoma_flash_attention_new(C.byref(fa), h, hk, d, float(scale), gpu.ref(ta) if ta is not None else None, -1, dtype, 0) != 0:
        raise RuntimeError("flash_attention_new: " + gpu.last_error())
    tn = dict(q=gpu.tensor(bufs["q"], (T, h, d), dtype), k=gpu.tensor(bufs["k"], (T, hk, d), dtype), v=gpu.tensor(bufs["v"], (T, hk, d), dtype),
              kv=gpu.tensor(bufs["kv"], (2, nb, page, hk, d), dtype), slots=gpu.tensor(bufs["slots"], (T,), gpu.I64),
              cu_q=gpu.tensor(bufs["cu_q"], (n_pre_seqs + 1,), gpu.U32), cu_k=gpu.tensor(bufs["cu_k"], (n_pre_seqs + 1,), gpu.U32),
              pre_bt=gpu.tensor(bufs["pre_bt"], pre_bt.shape, gpu.U32), dec_bt=gpu.tensor(bufs["dec_bt"], dec_bt.shape, gpu.U32),
              dec_lens=gpu.tensor(bufs["dec_lens"], (max(n_dec, 1),), gpu.U32), out=gpu.tensor(dout, (T, h * d), dtype))
    meta = gpu.AttnMetadata()
    meta.slot_mapping = C.pointer(tn["slots"])
    meta.num_prefill_tokens, meta.num_decoding_tokens = np_tok, n_dec
    use_prefix = c["use_prefix_route"]
    if n_pre_seqs:
        meta.has_prefill, meta.max_prefill_sequence_length, meta.max_sequence_length_k = 1, int(lq.max()), int(tot.max())
        meta.query_start_locations = C.pointer(tn["cu_q"])
        meta.sequence_start_locations = C.pointer(tn["cu_k"] if use_prefix else tn["cu_q"])
        if use_prefix:
            meta.prefill_block_tables = C.pointer(tn["pre_bt"])
    if n_dec:
        meta.has_decoding = 1
        meta.decoding_block_tables = C.pointer(tn["dec_bt"])
        meta.decoding_sequence_lengths = C.pointer(tn["dec_lens"])
    rc = gpu.lib.atoma_flash_attention_forward(C.byref(fa), gpu.ref(tn["q"]), gpu.ref(tn["k"]), gpu.ref(tn["v"]), gpu.ref(tn["kv"]), C.byref(meta), gpu.ref(tn["out"]))
    if rc != 0:
        raise RuntimeError("flash_attention_forward: " + gpu.last_error())
    yield
    gpu.synchronize()
    out = dout.numpy(np.uint16, (T, h, d))
    kc, vc = kv[0].copy(), kv[1].copy()
    CO.reshape_and_cache_flash(k, v, kc, vc, slots)
    got_kv = bufs["kv"].numpy(np.uint16, kv.shape)
    if not (np.array_equal(got_kv[0], kc) and np.array_equal(got_kv[1], vc)):
        return "the cache write is not bit-exact", 0
    soft = 0
    kcf, vcf, qf = to_f32(kc, dtype), to_f32(vc, dtype), to_f32(q, dtype)
    # Prefill rows.  Without a prefix: causal over the new tokens only, no ALiBi (flash_attention.rs:399-409: alibi_slopes is not passed on this route);
    # with block tables: over the cache, NON-causal (flash_attention.rs:435-448, SURVEY B/Q3), ALiBi if the layer has slopes.
    for i in range(n_pre_seqs):
        q0, q1 = int(cu_q[i]), int(cu_q[i + 1])
        if use_prefix:
            Lk = int(tot[i])
            kb, vb = A.gather_paged(kcf, pre_bt[i].astype(np.int64), Lk, page), A.gather_paged(vcf, pre_bt[i].astype(np.int64), Lk, page)
            causal, al = False, alibi
        else:
            Lk = q1 - q0
            kb, vb = to_f32(k[q0:q1], dtype), to_f32(v[q0:q1], dtype)
            causal, al = T > 1, None
        o, l = A.attend_rows(qf[q0:q1], kb, vb, scale, causal=causal, alibi_slopes=al, dtype=dtype)
        visible = np.minimum(Lk, np.arange(q1 - q0) + Lk - (q1 - q0) + 1) if causal else np.full(q1 - q0, Lk)
        msg, n = _check(out[q0:q1], l, from_f32(o, dtype), l, visible, dtype, f"prompt {i} (Lq={q1 - q0}, Lk={Lk}, prefix route={use_prefix})",
                        lambda: p_bounds(qf[q0:q1], kb, vb, scale, causal, al))
        soft += n
        if msg:
            return msg, soft
    for j in range(n_dec):
        L, r = int(dl[j]), np_tok + j
        kb, vb = A.gather_paged(kcf, dec_bt[j].astype(np.int64), L, page), A.gather_paged(vcf, dec_bt[j].astype(np.int64), L, page)
        o, l = A.attend_rows(qf[r:r + 1], kb, vb, scale, causal=alibi is not None, alibi_slopes=alibi, dtype=dtype)
        msg, n = _check(out[r:r + 1], l, from_f32(o, dtype), l, np.full(1, L), dtype, f"decode token {j} (L={L})", lambda: p_bounds(qf[r:r + 1], kb, vb, scale, False, alibi))
        soft += n
        if msg:
            return msg, soft
    return None, soft


def try_case(gpu, c):
    """run_case with library errors turned into findings (a shape the reference serves must not be refused)"""
    try:
        return run_case(gpu, c)[0]
    except (RuntimeError, AssertionError) as e:
        return f"raised {type(e).__name__}: {str(e)[:300]}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--count", type=int, default=0, help="stop after this many cases (0 = by time)")
    ap.add_argument("--kinds", default=",".join(KINDS))
    ap.add_argument("--forward-every", type=int, default=4, help="every n-th case is a `forward` case (seed + 10^6) and every n-th a large decode batch (seed + 2.10^6); 0 = none")
    ap.add_argument("--long", type=int, default=0, help="1: only prompts of 500 .. 5000 tokens (seed + 4.10^6), sampled rows")
    ap.add_argument("--graphs", type=int, default=0, help="1: only captured-and-replayed decode calls (graph_case) on the large-batch and the ordinary kv_cache cases")
    ap.add_argument("--threads", type=int, default=0, help="with --burst N: each group of N cases is dealt to this many host threads with a stream each (run_threads)")
    ap.add_argument("--burst", type=int, default=0, help="launch this many cases back to back before the first synchronisation (run_burst)")
    ap.add_argument("--strides", type=int, default=1, help="1: every other ordinary case with padded strides / oversized seqlen arguments (seed + 3.10^6)")
    ap.add_argument("--seeds", default="", help="comma-separated seeds to run instead of a range (re-running findings)")
    a = ap.parse_args()
    import atoma_hip as gpu
    gpu.set_device(0)
    kinds = tuple(a.kinds.split(","))
    t0, n, fails, per_kind, soft_rows, soft_cases, pending = time.time(), 0, [], {}, 0, 0, []
    seed = a.seed
    todo = [int(x) for x in a.seeds.split(",") if x]
    while (todo or not a.seeds) and (time.time() - t0 < a.seconds) and (not a.count or n < a.count):
        if a.seeds:
            seed = todo.pop(0)
        fe = a.forward_every
        extra = 0 if a.seeds else (FORWARD_BASE if fe and n % fe == fe - 1 else DECODE_BASE if fe and n % fe == 0 else STRIDE_BASE if a.strides and n % 2 else 0)
        c = draw(seed + (LONG_BASE if a.long else extra), kinds)
        if a.graphs:
            c = draw((DECODE_BASE if n % 2 else 0) + seed, ("kv_cache",))
            if c["d"] not in (64, 128) and n % 4 != 2:            # mostly the head sizes the streaming kernels serve; the others now and then
                c["d"], c["scale"] = 128, 128 ** -0.5
            c.update(sq=1, alibi=False)
            try:
                msg = graph_case(gpu, c)
            except (RuntimeError, AssertionError) as e:
                msg = f"raised {type(e).__name__}: {str(e)[:300]}"
            per_kind["graph replay"] = per_kind.get("graph replay", 0) + 1
            if msg:
                fails.append(dict(case=c, finding=msg))
                print(json.dumps(fails[-1]), file=sys.stderr, flush=True)
            n += 1
            seed += 1
            continue
        kname = c["kind"] + (" (long prompts)" if c["seed"] >= LONG_BASE else " (padded strides)" if c["seed"] >= STRIDE_BASE else " (large decode batches)" if c["seed"] >= DECODE_BASE else "")
        per_kind[kname] = per_kind.get(kname, 0) + 1
        if a.burst:
            pending.append(c)
            if len(pending) == a.burst:
                for cc, msg in (run_threads(gpu, pending, a.threads) if a.threads else run_burst(gpu, pending)):
                    fails.append(dict(case=cc, finding=msg, burst=[x["seed"] for x in pending]))
                    print(json.dumps(fails[-1]), file=sys.stderr, flush=True)
                pending = []
        else:
            try:
                msg, soft = run_case(gpu, c)
            except (RuntimeError, AssertionError) as e:
                msg, soft = f"raised {type(e).__name__}: {str(e)[:300]}", 0
            soft_rows, soft_cases = soft_rows + soft, soft_cases + (soft > 0)
            if msg:
                fails.append(dict(case=c, finding=msg))
                print(json.dumps(fails[-1]), file=sys.stderr, flush=True)
        n += 1
        seed += 1
    print(json.dumps(dict(cases=n - len(pending), burst=a.burst, seeds=[a.seed, seed - 1], per_kind=per_kind, seconds=round(time.time() - t0, 1), failures=len(fails),
                          cases_with_rows_held_to_their_own_bound=soft_cases, rows_held_to_their_own_bound=soft_rows, findings=fails[:40])), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
