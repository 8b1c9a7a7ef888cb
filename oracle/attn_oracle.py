"""numpy restatement of the reference's attention semantics (TEST INFRASTRUCTURE, see oracle/__init__.py).

Every public function takes/returns *storage-form* arrays: ``uint16`` bit patterns of
f16 (dtype code 0) or bf16 (dtype code 1), exactly what sits in device memory, plus
int32 index arrays.  Two arithmetic modes:

``mode="f32"``      the reference's own CPU statement ``fa_acausal``
                    (csrc/tests/flash_attn_tests.rs:19-29): f32 matmul -> *scale ->
                    softmax -> f32 matmul -> one rounding to the storage dtype.
``mode="kernel"``   what the CUDA kernel computes (csrc/kernels/softmax.h:65-91,135-185,
                    flash_fwd_kernel.h:886-1014): raw scores, max in the raw domain,
                    ``p = exp2(s*scale*log2e - max*scale*log2e)``, row sum in f32 of the
                    *unrounded* p, p rounded to the storage dtype before the P.V product,
                    f32 accumulation, one division at the end, one rounding.

Sequence-length rules follow csrc/kernels/block_info.h:11-39, masking follows
csrc/kernels/mask.h:110-209, paged addressing follows csrc/kernels/utils.h:296-314.
"""
import numpy as np

from .halfs import to_f32, from_f32, round_through

LOG2E = np.float32(1.4426950408889634)


# --------------------------------------------------------------------------------------
# one sequence, all heads
# --------------------------------------------------------------------------------------
def attend_rows(qf, kf, vf, scale, causal=False, alibi_slopes=None, mode="f32", dtype=1,
                key_lo=0, key_hi=None, return_unnormalised=False):
    """Attention of ``qf [Lq,h,d]`` over ``kf,vf [Lk,hk,d]`` (all float32).

    ``key_lo:key_hi`` restricts the *visible* key range (used to emulate one KV split,
    flash_fwd_kernel.h:534-538) while causal offsets still refer to the full ``Lk``.
    Returns ``(out f32 [Lq,h,d] -- not yet rounded, lse f32 [h,Lq])``.
    Empty visible range: out = 0 and lse = +inf (flash_fwd_kernel.h:97-133); the split
    path writes -inf instead (flash_fwd_kernel.h:543-582) -- see ``attend_rows_split``.
    """
    Lq, h, d = qf.shape
    Lk, hk, _ = kf.shape
    g = h // hk
    key_hi = Lk if key_hi is None else min(key_hi, Lk)
    out = np.zeros((Lq, h, d), np.float32)
    lse = np.full((h, Lq), np.inf, np.float32)
    if Lq == 0 or key_hi <= key_lo:
        return out, lse
    scale = np.float32(scale)
    rows = np.arange(Lq)[:, None]
    cols = np.arange(key_lo, key_hi)[None, :]
    shift = Lk - Lq                                   # mask.h:170: col <= row + seqlen_k - seqlen_q
    visible = np.ones((Lq, key_hi - key_lo), bool)
    if causal:
        visible = cols <= rows + shift
    for head in range(h):
        kh = kf[key_lo:key_hi, head // g]
        vh = vf[key_lo:key_hi, head // g]
        s = qf[:, head].astype(np.float32) @ kh.T.astype(np.float32)      # raw scores
        if alibi_slopes is not None:
            # flash_fwd_kernel.h:871 + mask.h:179-186: slope/scale added in the raw domain
            slope = np.float32(alibi_slopes[head]) / scale
            s = s - slope * np.abs(rows + shift - cols).astype(np.float32)
        s = np.where(visible, s, -np.inf).astype(np.float32)
        m = s.max(axis=1, keepdims=True)
        dead = ~np.isfinite(m[:, 0])                   # fully masked rows (softmax.h:149-151)
        m_safe = np.where(np.isfinite(m), m, np.float32(0))
        if mode == "f32":
            p = np.exp((s - m_safe) * scale).astype(np.float32)
            l = p.sum(axis=1, keepdims=True, dtype=np.float32)
            pv = p @ vh
        elif mode == "kernel":
            sl2 = np.float32(scale * LOG2E)
            p = np.exp2(s * sl2 - m_safe * sl2).astype(np.float32)
            l = p.sum(axis=1, keepdims=True, dtype=np.float32)
            pv = round_through(p, dtype) @ vh          # P cast to bf16/f16 before P.V
        else:
            raise ValueError(mode)
        l_safe = np.where((l == 0) | np.isnan(l), np.float32(1), l)   # softmax.h:176
        o = pv if return_unnormalised else pv / l_safe
        o[dead] = 0
        out[:, head] = o
        lse_h = m_safe[:, 0] * scale + np.log(l_safe[:, 0])
        lse[head] = np.where(dead, np.inf, lse_h)
    return out, lse


def attend_decode_online(qf, kf, vf, scale, dtype, tile=16, ngroups=4, group_of=None, alibi_slopes=None):
    """One decode row per head (``qf [h,d]``) over ``kf,vf [L,hk,d]`` with the reference kernel's arithmetic -- exp2 domain,
    f32 row sum of the unrounded p, p rounded to the storage dtype before P.V, f32 accumulation, one division at the end
    (softmax.h:65-91,135-185) -- evaluated under an explicit ONLINE-softmax schedule: keys arrive in tiles of ``tile``;
    key j of a tile belongs to accumulator group ``group_of(j)`` (``ngroups`` independent (max, sum, O) states, each
    rescaled when its running max rises: softmax.h:135-160), and the groups are merged at the end with the LSE rule.

    Why it exists: rounding p = exp2(s - m) to bf16 depends on WHICH running max m it is taken against, so two correct
    kernels with different schedules differ by up to 2^-9 * |v| on rows with few keys (tests/test_oracle_golden.py shows
    it on the CPU).  ``mode="kernel"`` of attend_rows is the one-tile / one-group schedule (global max); the reference's
    split-KV kernel is (tile = 128 or 64, one group); libatoma_hip's decode kernels are (16, 4 groups, j % 4) for d = 128,
    (16, 8 groups, j % 8) for d = 64 and (16, one group) for the matrix-core variant.  Given the kernel's own
    schedule the comparison is tight (1e-3 + 1 ulp at every row length).  Returns out f32 [h,d] (not yet rounded)."""
    h, d = qf.shape
    L, hk, _ = kf.shape
    g = h // hk
    group_of = group_of or (lambda j: j % ngroups)
    sl2 = np.float32(np.float32(scale) * LOG2E)
    m = np.full((ngroups, h), -np.inf, np.float32)
    l = np.zeros((ngroups, h), np.float32)
    o = np.zeros((ngroups, h, d), np.float32)
    head_kv = np.arange(h) // g
    for t0 in range(0, L, tile):
        rows = np.arange(t0, min(t0 + tile, L))
        kt = kf[rows][:, head_kv]                                   # [n, h, d]
        vt = vf[rows][:, head_kv]
        s = np.einsum("hd,nhd->nh", qf, kt, dtype=np.float32).astype(np.float32) * sl2
        if alibi_slopes is not None:                                # mask.h:183 with one query row at position L - 1
            s = s - (np.asarray(alibi_slopes, np.float32) * LOG2E)[None, :] * (L - 1 - rows)[:, None].astype(np.float32)
        grp = np.array([group_of(int(j - t0)) for j in rows])
        for gi in range(ngroups):
            sel = grp == gi
            if not sel.any():
                continue
            sg = s[sel]
            mnew = np.maximum(m[gi], sg.max(0))
            with np.errstate(invalid="ignore"):
                alpha = np.where(np.isfinite(m[gi]), np.exp2(m[gi] - mnew), np.float32(0)).astype(np.float32)
            l[gi] *= alpha
            o[gi] *= alpha[:, None]
            m[gi] = mnew
            p = np.exp2(sg - mnew[None, :]).astype(np.float32)
            l[gi] += p.sum(0, dtype=np.float32)
            o[gi] += np.einsum("nh,nhd->hd", round_through(p, dtype), vt[sel], dtype=np.float32)
    mt = m.max(0)
    with np.errstate(invalid="ignore"):
        w = np.where(np.isfinite(m), np.exp2(m - np.where(np.isfinite(mt), mt, np.float32(0))[None, :]), np.float32(0)).astype(np.float32)
    lt = (l * w).sum(0, dtype=np.float32)
    ot = (o * w[:, :, None]).sum(0, dtype=np.float32)
    return np.where(lt[:, None] > 0, ot / np.where(lt > 0, lt, np.float32(1))[:, None], np.float32(0)).astype(np.float32)


def attend_prefill_online(qf, kf, vf, scale, causal, dtype, tile=64, defer=8.0, prescale=False):
    """Prefill rows under the schedule of libatoma_hip's MFMA prefill kernel (csrc/prefill_mfma.hip, prefill_cfg = 0), with the
    reference's arithmetic (softmax.h:65-91,135-185): keys arrive in tiles of ``tile`` (64); per row, the running max (exp2
    domain) is raised to max(m, tile max) only when that exceeds m + ``defer`` (the deferred raise: p <= 2^defer), O and the row
    sum are rescaled by exp2(m_old - m_new) when it is; p = exp2(fma(s, scale.log2e, -m)) in f32, the row sum takes the
    UNROUNDED p, P.V takes p rounded to the storage dtype, f32 accumulation, one division at the end.  Why: rounding p to bf16
    depends on which running max it is taken against, so against the f32 definition rows with few keys can only be held to
    2^-9.|v| (tests/test_oracle_schedules.py); against the kernel's OWN schedule every row is held to 1e-3 + 1 ulp.
    ``prescale``: the schedule of the hand-scheduled kernel's fast variant (csrc/prefill_asm.hip, tools/pfasm/kernel.py): q is
    multiplied by scale.log2(e) and rounded to the storage dtype ONCE, the scores leave the matrix pipe in the exp2 domain and
    p = exp2(s~ - m) needs no multiply.
    ``qf [Lq,h,d]``, ``kf, vf [Lk,hk,d]`` float32; returns out f32 [Lq,h,d] (not yet rounded)."""
    Lq, h, d = qf.shape
    Lk, hk, _ = kf.shape
    g = h // hk
    out = np.zeros((Lq, h, d), np.float32)
    if Lq == 0 or Lk == 0:
        return out
    sl2 = np.float32(np.float32(scale) * LOG2E)
    rows = np.arange(Lq)[:, None]
    shift = Lk - Lq
    for head in range(h):
        kh, vh = kf[:, head // g].astype(np.float32), vf[:, head // g].astype(np.float32)
        m = np.full(Lq, -np.inf, np.float32)
        l = np.zeros(Lq, np.float32)
        o = np.zeros((Lq, d), np.float32)
        for kv0 in range(0, Lk, tile):
            cols = np.arange(kv0, min(kv0 + tile, Lk))[None, :]
            if prescale:
                qs = round_through((qf[:, head].astype(np.float32) * sl2).astype(np.float32), dtype)
                s = (qs @ kh[kv0:kv0 + tile].T).astype(np.float32)
            else:
                s = (qf[:, head].astype(np.float32) @ kh[kv0:kv0 + tile].T).astype(np.float32)
            if causal:
                s = np.where(cols <= rows + shift, s, -np.inf).astype(np.float32)
            mx = s.max(1).astype(np.float32) if prescale else (s.max(1) * sl2).astype(np.float32)   # raw max, then the scale (as the kernel)
            m_cand = np.maximum(m, mx)
            with np.errstate(invalid="ignore"):
                raise_it = m_cand > m + np.float32(defer)            # m = -inf: any finite candidate raises
            m_new = np.where(raise_it, m_cand, m).astype(np.float32)
            ms = np.where(np.isfinite(m_new), m_new, np.float32(0)).astype(np.float32)
            with np.errstate(invalid="ignore"):
                alpha = np.where(np.isfinite(m), np.exp2(m - ms), np.float32(0)).astype(np.float32)
                alpha = np.where(np.isfinite(m) | np.isfinite(m_new), alpha, np.float32(1))
            pexp = (s.astype(np.float64) * np.float64(1.0 if prescale else sl2) - ms[:, None].astype(np.float64)).astype(np.float32)   # one rounding: v_pk_fma_f32
            p = np.exp2(pexp).astype(np.float32)
            l = (l * alpha + p.sum(1, dtype=np.float32)).astype(np.float32)
            o = (o * alpha[:, None] + round_through(p, dtype) @ vh[kv0:kv0 + tile]).astype(np.float32)
            m = m_new
        out[:, head] = np.where(l[:, None] > 0, o / np.where(l > 0, l, np.float32(1))[:, None], np.float32(0))
    return out


DECODE_SCHEDULES = {            # (head_dim, variant) -> (ngroups, group_of): row ownership of libatoma_hip's decode kernels
    (128, "dot2"): (4, lambda j: j % 4),      # paged_decode_item: 16 lanes per row, lane group `sub` loads rows sub + 4r
    (64, "dot2"): (8, lambda j: j % 8),       # 8 lanes per row, 8 rows per load instruction
    (128, "mqk"): (1, lambda j: 0),           # paged_decode_mqk_item since P.V runs on the matrix cores: all 16 tokens of a tile enter one
                                              # accumulator, so the running max is common to the tile (round 2: 4 groups, j // 4)
    (64, "mqk"): (1, lambda j: 0),            # the same item on pairs of 64-dim kv heads (PAIR64): the other head's dims enter q.K^T as exact zeros
}


def flash_attn_kv_cache_online(q, kc, vc, scale, dtype, block_table, seqlens_k, variant="dot2", alibi_slopes=None):
    """flash_attn_kv_cache for seqlen_q = 1 under the decode kernel's own schedule (see attend_decode_online)."""
    qf, kf, vf = to_f32(q, dtype), to_f32(kc, dtype), to_f32(vc, dtype)
    B, _, h, d = qf.shape
    ngroups, group_of = DECODE_SCHEDULES[(d, variant)]
    out = np.zeros(qf.shape, np.float32)
    for b in range(B):
        L = int(seqlens_k[b])
        if L == 0:
            continue
        ps = kf.shape[1]
        kb, vb = gather_paged(kf, block_table[b], L, ps), gather_paged(vf, block_table[b], L, ps)
        out[b, 0] = attend_decode_online(qf[b, 0], kb, vb, scale, dtype, 16, ngroups, group_of, alibi_slopes)
    return from_f32(out, dtype)


def combine_splits(o_parts, lse_parts):
    """LSE-weighted merge of split-KV partials (flash_fwd_kernel.h:1204-1236).

    ``o_parts [S, ..., d]`` f32 (each normalised within its split), ``lse_parts [S, ...]``
    f32 with -inf for an empty split.  All-empty -> lse = +inf, O = 0.
    """
    o_parts = np.asarray(o_parts, np.float32)
    lse_parts = np.asarray(lse_parts, np.float32)
    m = lse_parts.max(axis=0)
    m_safe = np.where(np.isfinite(m), m, np.float32(0))
    w = np.exp(lse_parts - m_safe)
    tot = w.sum(axis=0, dtype=np.float32)
    with np.errstate(divide="ignore"):
        lse = np.log(tot) + m_safe
    empty = (tot == 0) | np.isnan(tot)
    lse = np.where(empty, np.inf, lse).astype(np.float32)
    wn = np.exp(lse_parts - np.where(empty, np.float32(0), lse))
    wn = np.where(empty, np.float32(0), wn)
    out = (wn[..., None] * o_parts).sum(axis=0, dtype=np.float32)
    return out, lse


def attend_rows_split(qf, kf, vf, scale, num_splits, block_n, **kw):
    """Emulate Split=true (flash_fwd_kernel.h:534-538): ``n_blocks_per_split`` blocks of
    ``block_n`` keys per split, partial O normalised per split, partial LSE (-inf when a
    split is empty), then ``combine_splits``."""
    Lk = kf.shape[0]
    n_blocks = (Lk + block_n - 1) // block_n
    per = (n_blocks + num_splits - 1) // num_splits if num_splits else n_blocks
    os_, ls_ = [], []
    for s in range(num_splits):
        o, l = attend_rows(qf, kf, vf, scale, key_lo=s * per * block_n,
                           key_hi=(s + 1) * per * block_n, **kw)
        l = np.where(np.isposinf(l), -np.inf, l)
        os_.append(o.transpose(1, 0, 2))          # [h, Lq, d]
        ls_.append(l)
    o, lse = combine_splits(np.stack(os_), np.stack(ls_))
    return o.transpose(1, 0, 2), lse


# --------------------------------------------------------------------------------------
# layouts of the reference's three entry points
# --------------------------------------------------------------------------------------
def gather_paged(cache_f32, block_table_row, length, page_size):
    """Rows ``[0, length)`` of a sequence out of ``cache [nb, page, hk, d]`` through its
    block-table row: token j lives at ``cache[bt[j // page], j % page]`` (utils.h:296-314)."""
    j = np.arange(length)
    return cache_f32[np.asarray(block_table_row)[j // page_size], j % page_size]


def flash_attn(q, k, v, scale, causal, dtype, alibi_slopes=None, mode="f32"):
    """csrc::flash_attn (csrc/src/lib.rs:392-411): q [b,sq,h,d], k,v [b,sk,hk,d] -> [b,sq,h,d]."""
    qf, kf, vf = to_f32(q, dtype), to_f32(k, dtype), to_f32(v, dtype)
    if qf.shape[1] == 1 and alibi_slopes is None:      # lib.rs:214-216
        causal = False
    out = np.empty(qf.shape, np.float32)
    for b in range(qf.shape[0]):
        out[b], _ = attend_rows(qf[b], kf[b], vf[b], scale, causal, alibi_slopes, mode, dtype)
    return from_f32(out, dtype)


def flash_attn_varlen(q, k, v, cu_q, cu_k, scale, causal, dtype, block_table=None,
                      alibi_slopes=None, mode="f32", return_lse=False):
    """csrc::flash_attn_varlen / _with_block_table (csrc/src/lib.rs:1160-1188,1392-1420).

    q ``[total_q,h,d]``; k,v ``[total_k,hk,d]`` or, with ``block_table [B,max_blocks]``,
    the paged cache ``[nb,page,hk,d]``.  ``cu_q``/``cu_k`` are cumulative ``[B+1]``.
    """
    qf, kf, vf = to_f32(q, dtype), to_f32(k, dtype), to_f32(v, dtype)
    out = np.zeros(qf.shape, np.float32)
    lses = []
    B = len(cu_q) - 1
    for b in range(B):
        q0, q1 = int(cu_q[b]), int(cu_q[b + 1])
        k0, k1 = int(cu_k[b]), int(cu_k[b + 1])
        if block_table is None:
            kb, vb = kf[k0:k1], vf[k0:k1]
        else:
            ps = kf.shape[1]
            kb = gather_paged(kf, block_table[b], k1 - k0, ps)
            vb = gather_paged(vf, block_table[b], k1 - k0, ps)
        o, lse = attend_rows(qf[q0:q1], kb, vb, scale, causal, alibi_slopes, mode, dtype)
        out[q0:q1] = o
        lses.append(lse)
    res = from_f32(out, dtype)
    return (res, lses) if return_lse else res


def flash_attn_kv_cache(q, kc, vc, scale, dtype, block_table=None, seqlens_k=None,
                        causal=False, alibi_slopes=None, mode="f32", num_splits=0,
                        block_n=128):
    """csrc::flash_attn_kv_cache_full (csrc/src/lib.rs:1521-1855,2083-2105).

    q ``[B,sq,h,d]``; caches ``[B_c,sk,hk,d]`` or paged ``[nb,page,hk,d]`` with
    ``block_table [B,max_blocks]``; ``seqlens_k [B]`` are *per-sequence* lengths
    (``is_seqlens_k_cumulative=false``, block_info.h:21-22); None -> full ``sk``.
    """
    qf, kf, vf = to_f32(q, dtype), to_f32(kc, dtype), to_f32(vc, dtype)
    B, sq = qf.shape[:2]
    if sq == 1 and alibi_slopes is None:               # lib.rs:1629-1631
        causal = False
    out = np.zeros(qf.shape, np.float32)
    for b in range(B):
        if block_table is not None:
            ps = kf.shape[1]
            L = int(seqlens_k[b]) if seqlens_k is not None else len(block_table[b]) * ps
            kb = gather_paged(kf, block_table[b], L, ps)
            vb = gather_paged(vf, block_table[b], L, ps)
        else:
            L = int(seqlens_k[b]) if seqlens_k is not None else kf.shape[1]
            kb, vb = kf[b, :L], vf[b, :L]
        kw = dict(causal=causal, alibi_slopes=alibi_slopes, mode=mode, dtype=dtype)
        if num_splits and num_splits > 1:
            out[b], _ = attend_rows_split(qf[b], kb, vb, scale, num_splits, block_n, **kw)
        else:
            out[b], _ = attend_rows(qf[b], kb, vb, scale, **kw)
    return from_f32(out, dtype)


# --------------------------------------------------------------------------------------
# host-side integer logic restated (csrc/src/lib.rs:2122-2199)
# --------------------------------------------------------------------------------------
def num_splits_heuristic(batch_nheads_mblocks, num_sms, num_n_blocks, max_splits):
    if np.float32(batch_nheads_mblocks) >= np.float32(0.8) * np.float32(num_sms):
        return 1
    max_splits = min(max_splits, num_sms, num_n_blocks)
    cdiv = lambda a, b: (a + b - 1) // b
    eligible = lambda s: s == 1 or cdiv(num_n_blocks, s) != cdiv(num_n_blocks, s - 1)
    eff, best = [], np.float32(0)
    for s in range(1, max_splits + 1):
        if not eligible(s):
            eff.append(np.float32(0))
            continue
        n_waves = np.float32(batch_nheads_mblocks * s) / np.float32(num_sms)
        e = np.float32(n_waves / np.ceil(n_waves))
        best = max(best, e)
        eff.append(e)
    for s in range(1, max_splits + 1):
        if eligible(s) and eff[s - 1] >= np.float32(0.85) * best:
            return s
    return 1


def compute_num_splits(batch, heads, head_size, max_seqlen_k, max_seqlen_q, num_sms):
    block_n = 256 if head_size <= 64 else (128 if head_size <= 128 else 64)
    n_blocks = (max_seqlen_k + block_n - 1) // block_n
    m_blocks = (max_seqlen_q + 63) // 64
    return num_splits_heuristic(batch * heads * m_blocks, num_sms * 2, n_blocks, 128)
