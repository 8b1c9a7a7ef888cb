"""C1 "plumbing" harness: a small Llama-shaped attention stack driven through a
continuous-batching trace -- prompts are prefilled, sequences decode token by token over a paged KV
cache, one finishes early, its pages are reused by a late arrival.

What is on the hot path (RMSNorm, RoPE, reshape_and_cache_flash, varlen prefill attention, paged decode
attention) goes through an `ops` backend; the projections around it are plain f32 numpy matmuls on the
host, identical for every backend.  `OracleOps` is the CPU restatement; the GPU backend lives in
test_plumbing_gpu.py.  With a `check` backend every op call is replayed on the same inputs and compared
(the primary backend's outputs drive the trajectory), so a slot-mapping / block-table / position bug
shows up at the step and op where it happens (reference flow: backends/vllm/src/worker.rs:224-460,
models/src/llama.rs:253-314,392-410).
"""
import numpy as np

from oracle import attn_oracle as A
from oracle import cache_oracle as CO
from oracle import norm_rope_oracle as NR
from oracle.halfs import BF16, from_f32, to_f32

DT = BF16


class Dims:
    layers, hidden, h, hk, d = 2, 512, 8, 2, 64     # Llama-3.2-1B head geometry (d = 64, 4 q heads per kv head)
    page, num_pages, vocab, max_pos = 16, 40, 97, 256
    eps, theta = 1e-5, 500000.0


class OracleOps:
    """The hot-path ops on numpy arrays (storage-form uint16), with the paged caches as state."""

    def __init__(self, dims):
        self.D = dims
        shape = (dims.num_pages, dims.page, dims.hk, dims.d)
        self.kc = [np.zeros(shape, np.uint16) for _ in range(dims.layers)]
        self.vc = [np.zeros(shape, np.uint16) for _ in range(dims.layers)]

    def rms_norm(self, x, w):
        return NR.rms_norm(x, w, self.D.eps, DT)

    def rope_qk(self, q, k, cos, sin, pos):
        return NR.rope(q, cos, sin, pos, DT), NR.rope(k, cos, sin, pos, DT)

    def reshape_and_cache(self, layer, k, v, slots):
        CO.reshape_and_cache_flash(k, v, self.kc[layer], self.vc[layer], slots)

    def caches(self, layer):
        return self.kc[layer], self.vc[layer]

    def prefill(self, q, k, v, cu):
        return A.flash_attn_varlen(q, k, v, cu, cu, self.D.d ** -0.5, True, DT)

    def decode(self, layer, q, bt, lens):
        return A.flash_attn_kv_cache(q[:, None], self.kc[layer], self.vc[layer], self.D.d ** -0.5, DT,
                                     block_table=bt, seqlens_k=lens)[:, 0]


class Weights:
    def __init__(self, dims, seed=0):
        rng = np.random.default_rng(seed)
        D = dims
        r = lambda *s: rng.standard_normal(s).astype(np.float32)
        self.emb = from_f32(r(D.vocab, D.hidden), DT)
        self.norm1 = [from_f32(1 + 0.1 * r(D.hidden), DT) for _ in range(D.layers)]
        self.wqkv = [r(D.hidden, (D.h + 2 * D.hk) * D.d) * D.hidden ** -0.5 for _ in range(D.layers)]
        self.wo = [r(D.h * D.d, D.hidden) * (D.h * D.d) ** -0.5 for _ in range(D.layers)]
        self.norm_f = from_f32(1 + 0.1 * r(D.hidden), DT)
        self.cos, self.sin = NR.rope_table(D.max_pos, D.d, D.theta, DT)


class Seq:
    def __init__(self, sid, prompt, max_new):
        self.sid, self.tokens, self.max_new, self.pages, self.cached = sid, list(prompt), max_new, [], 0


class Compare:
    """Replays every op on a second backend and records the worst deviation per op."""

    def __init__(self):
        self.rms_ulps, self.rope_exact, self.cache_exact, self.attn_err, self.calls = 0, True, True, 0.0, 0

    def ulps(self, a, b):
        return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()) if a.size else 0


def run_trace(ops, weights, dims, check=None, cmp=None, dense_check=False, steps=32, seed=1):
    """Drive the stack.  Returns the per-step greedy tokens of every sequence."""
    D, W = dims, weights
    rng = np.random.default_rng(seed)
    free = list(rng.permutation(D.num_pages))            # physical pages in random order
    waiting = [Seq(0, rng.integers(0, D.vocab, 32), steps), Seq(1, rng.integers(0, D.vocab, 17), 10),
               Seq(2, rng.integers(0, D.vocab, 45), steps)]
    late = Seq(3, rng.integers(0, D.vocab, 20), steps)   # arrives at step 12, into the pages sequence 1 gave back
    running, history = [], {}
    dense = {}                                           # sid -> per-layer roped K / V history (dense_check)

    def grow(seq, upto):
        while len(seq.pages) * D.page < upto:
            seq.pages.append(int(free.pop()))

    def block_table(seqs):
        width = max(len(s.pages) for s in seqs)
        bt = np.zeros((len(seqs), width), np.int32)
        for i, s in enumerate(seqs):
            bt[i, :len(s.pages)] = s.pages
        return bt

    def forward(seqs, prefill):
        # tokens of this step, their positions and cache slots (worker.rs:280-400)
        toks, pos, slots, cu = [], [], [], [0]
        for s in seqs:
            new = s.tokens[s.cached:] if prefill else s.tokens[-1:]
            start = len(s.tokens) - len(new)
            grow(s, len(s.tokens))
            toks += new
            pos += range(start, len(s.tokens))
            slots += list(CO.slot_mapping_for(np.asarray(s.pages), start, len(s.tokens), D.page))
            cu.append(cu[-1] + len(new))
        pos, slots, cu = np.asarray(pos, np.int64), np.asarray(slots, np.int64), np.asarray(cu, np.int32)
        x = W.emb[np.asarray(toks)]
        lens = np.asarray([len(s.tokens) for s in seqs], np.int32)
        bt = block_table(seqs)
        for l in range(D.layers):
            xn = ops.rms_norm(x, W.norm1[l])
            if check:
                cmp.rms_ulps = max(cmp.rms_ulps, cmp.ulps(xn, check.rms_norm(x, W.norm1[l])))
            qkv = from_f32(to_f32(xn, DT) @ W.wqkv[l], DT)
            T = len(toks)
            q = np.ascontiguousarray(qkv[:, :D.h * D.d]).reshape(T, D.h, D.d)
            k = np.ascontiguousarray(qkv[:, D.h * D.d:(D.h + D.hk) * D.d]).reshape(T, D.hk, D.d)
            v = np.ascontiguousarray(qkv[:, (D.h + D.hk) * D.d:]).reshape(T, D.hk, D.d)
            fused = hasattr(ops, "rope_qk_cache")        # one launch for RoPE(q, k) + the cache write
            qr, kr = ops.rope_qk_cache(l, q, k, v, W.cos, W.sin, pos, slots) if fused else ops.rope_qk(q, k, W.cos, W.sin, pos)
            if check:
                cq, ck = check.rope_qk(q, k, W.cos, W.sin, pos)
                cmp.rope_exact &= bool(np.array_equal(qr, cq) and np.array_equal(kr, ck))
            if not fused:
                ops.reshape_and_cache(l, kr, v, slots)
            if check:
                check.reshape_and_cache(l, kr, v, slots)
                gk, gv = ops.caches(l)
                ok, ov = check.caches(l)
                cmp.cache_exact &= bool(np.array_equal(gk, ok) and np.array_equal(gv, ov))
            if dense_check:
                for i, s in enumerate(seqs):
                    hk_, hv_ = dense.setdefault((s.sid, l), ([], []))
                    hk_.append(kr[cu[i]:cu[i + 1]])
                    hv_.append(v[cu[i]:cu[i + 1]])
            if prefill:
                att = ops.prefill(qr, kr, v, cu)
                ref = check.prefill(qr, kr, v, cu) if check else None
            else:
                att = ops.decode(l, qr, bt, lens)
                ref = check.decode(l, qr, bt, lens) if check else None
            if check:
                r32 = to_f32(ref, DT)      # excess over one unit in the last place of the storage dtype (util.ulp_tol)
                cmp.attn_err = max(cmp.attn_err, float((np.abs(to_f32(att, DT) - r32) - 2.0 ** -7 * np.abs(r32)).max()))
                cmp.calls += 1
            if dense_check and not prefill:
                # the paged decode must equal dense attention over the sequence's whole K/V history
                for i, s in enumerate(seqs):
                    kh = np.concatenate(dense[(s.sid, l)][0])
                    vh = np.concatenate(dense[(s.sid, l)][1])
                    want = A.flash_attn(qr[i][None, None], kh[None], vh[None], D.d ** -0.5, False, DT)[0, 0]
                    assert np.array_equal(att[i], want), f"paged != dense history: seq {s.sid} layer {l} len {len(s.tokens)}"
            x = from_f32(to_f32(x, DT) + to_f32(att, DT).reshape(T, D.h * D.d) @ W.wo[l], DT)
        last = x[cu[1:] - 1]
        logits = to_f32(ops.rms_norm(last, W.norm_f), DT) @ to_f32(W.emb, DT).T
        for i, s in enumerate(seqs):
            s.cached = len(s.tokens)
            s.tokens.append(int(logits[i].argmax()))
            history.setdefault(s.sid, []).append(s.tokens[-1])

    for step in range(steps + 1):
        if step == 12:
            waiting.append(late)
        if waiting:                                      # a prefill step for the new arrivals (scheduler: prefill first)
            forward(waiting, prefill=True)
            running += waiting
            waiting = []
            continue
        forward(running, prefill=False)
        for s in list(running):
            if len(history[s.sid]) >= s.max_new:
                running.remove(s)
                free.extend(s.pages)                     # pages go back to the pool (and are handed out again)
                s.pages = []
    return history
