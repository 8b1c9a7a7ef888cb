"""Host side of the generated prefill kernel, in Python: the 64-dword entry per (query block, wavefront) -- the same arithmetic as
csrc/prefill_asm.hip's pfa_fill_entry -- and drivers that run the generated program on the simulator: a persistent workgroup
walking a plan table (contiguous K/V), or one workgroup per block with its entries in LDS (paged).
Test infrastructure (tests/test_prefill_asm_sim.py)."""
import numpy as np

from .isa import S
from .kernel import PIDX, PARAM_DWORDS, LDS_PARAMS, LDS_TOTAL, FLAG_EXACT, FLAG_VALID, build
from .sim import Memory, Workgroup

BIG = 0x3FFFFFFF
EXACT_KEYS = 512


def wave_entry(*, wave, m0, len_q, len_k, causal, q_addr, o_addr, lse_addr, q_stride, o_stride, k_addr, v_addr, k_stride,
               v_stride, scale_log2, paged=False, bt_addr=0, page_size=0, k_page_bytes=0, v_page_bytes=0, simple=False, exact=None):
    """q_addr / o_addr: address of row 0 of the SEQUENCE for this head (bytes); lse_addr: address of row 0's LSE or 0.
    k_addr / v_addr: contiguous: row 0 of the sequence for the kv head; paged: cache base + head offset.
    exact: None = by the rule of the kernel (the block's first row sees fewer than EXACT_KEYS keys)."""
    shift = len_k - len_q

    def block(j):
        r0 = m0 + 32 * j
        rows = int(np.clip(len_q - r0, 0, 32))
        if rows == 0:
            return dict(r0=r0, rows=0, n=0)
        last = r0 + rows - 1
        vis = min(len_k, last + shift + 1) if causal else len_k
        n = (vis + 63) // 64 if vis > 0 else 0
        return dict(r0=r0, rows=rows, n=n)
    blocks = [block(j) for j in range(8)]
    a, b = blocks[wave], blocks[7 - wave]
    s0, s1 = (a, b) if (a["n"], a["r0"]) <= (b["n"], b["r0"]) else (b, a)
    n_tiles = max(bl["n"] for bl in blocks)
    n0, n1 = s0["n"], s1["n"]
    if simple and n0 > 0:
        n0 = n1
    if exact is None:
        exact = (min(len_k, m0 + shift + 1) if causal else len_k) < EXACT_KEYS
    p = np.zeros(PARAM_DWORDS, np.uint32)
    if m0 >= len_q:             # an empty entry: only the launch-wide strides
        for name, val in (("q_stride", q_stride), ("o_stride", o_stride), ("k_stride", k_stride), ("v_stride", v_stride),
                          ("k_tile", 64 * k_stride), ("v_tile", 64 * v_stride)):
            p[PIDX[name]] = val
        return p

    def put(name, val):
        p[PIDX[name]] = np.uint32(int(val) & 0xFFFFFFFF)

    def put64(name, val):
        put(name + "_lo", val & 0xFFFFFFFF)
        put(name + "_hi", val >> 32)
    tms = []
    for s, sl in enumerate((s0, s1)):
        put64(f"q{s}", q_addr + sl["r0"] * q_stride)
        put64(f"o{s}", o_addr + sl["r0"] * o_stride)
        put(f"o{s}_bytes", (sl["rows"] - 1) * o_stride + 256 if sl["rows"] else 0)
        put(f"o{s}_flags", 0x00020000)
        put64(f"lse{s}", lse_addr + sl["r0"] * 4 if lse_addr else 0)
        put(f"rows{s}", sl["rows"])
        lim = sl["r0"] + shift if causal else len_k - 1
        minlim = min(lim, len_k - 1)
        tms.append(max(0, (minlim + 1) >> 6) if minlim >= 0 else 0)
        put(f"lim{s}", lim)
    put("q_stride", q_stride)
    put("o_stride", o_stride)
    put64("k", k_addr)
    put64("v", v_addr)
    put("k_stride", k_stride)
    put("v_stride", v_stride)
    put("k_tile", 64 * k_stride)
    put("v_tile", 64 * v_stride)
    put("k_bytes", (len_k - 1) * k_stride + 256 if len_k > 0 else 0)
    put("v_bytes", (len_k - 1) * v_stride + 256 if len_k > 0 else 0)
    put("k_flags", 0x00020000)
    put("v_flags", 0x00020000)
    put("len_k", len_k)
    put("n_tiles", n_tiles)
    put("n0", n0)
    put("n1", n1)
    tm0 = tms[0] if n0 > 0 else BIG
    tm1 = tms[1] if n1 > 0 else BIG
    tmm = min(tm0, tm1)
    put("tm0", tm0)
    put("tm1", tm1)
    put("tmm", tmm)
    put("n_steady", max(0, min(n0, tmm) - 1))
    put("lim_step", 1 if causal else 0)
    put("scale_log2", np.float32(scale_log2).view(np.uint32))
    put("thr", np.float32(np.float32(8.0) / np.float32(scale_log2) if exact else 8.0).view(np.uint32))
    put("mscale", np.float32(scale_log2 if exact else 1.0).view(np.uint32))
    put("flags", (FLAG_EXACT if exact else 0) | FLAG_VALID)
    if paged:
        put64("bt", bt_addr)
        put("page_shift", int(page_size).bit_length() - 1)
        put("k_page", k_page_bytes)
        put("v_page", v_page_bytes)
    put("wave", wave)
    return p


_progs = {}


def program(dtype, paged, timing=False):
    key = (dtype, paged, timing)
    if key not in _progs:
        _progs[key] = build(dtype, paged, timing=timing)
    return _progs[key]


def run_persistent(mem, entries, dtype="bf16", late_dma=True, reverse=False, g=1, timing=False, hook=None):
    """entries: [n_blocks][4] parameter entries; `g` persistent workgroups (run one after the other) share the table round-robin"""
    prog, _ = program(dtype, False, timing)
    n = len(entries)
    tab = mem.alloc(np.concatenate([np.concatenate(e) for e in entries]) if n else np.zeros(64, np.uint32))
    wgs = []
    for first in range(min(g, max(n, 1))):
        wg = Workgroup(prog, mem, n_waves=4, lds_bytes=LDS_TOTAL, late_dma=late_dma, reverse=reverse, max_steps=20_000_000)
        for w in range(4):
            s = wg.waves[w].s
            s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12] = tab & 0xFFFFFFFF, tab >> 32, first, n, g, w, 0, 0, g
        if hook:
            hook(wg)
        wg.run()
        wgs.append(wg)
    return wgs


def run_block_paged(mem, entries4, dtype="bf16", late_dma=True, reverse=False):
    prog, _ = program(dtype, True)
    wg = Workgroup(prog, mem, n_waves=4, lds_bytes=LDS_TOTAL, late_dma=late_dma, reverse=reverse)
    for w in range(4):
        wg.lds[LDS_PARAMS + 256 * w:LDS_PARAMS + 256 * (w + 1)] = entries4[w].view(np.uint8)
        wg.waves[w].s[4] = LDS_PARAMS + 256 * w
    wg.run()
    return [wg]


def prefill_varlen(q, k, v, cu_q, cu_k, scale, causal, dtype="bf16", want_lse=False, block_table=None, page_size=0, late_dma=True,
                   reverse=False, simple=False, exact=None, g=1, hook=None):
    """q [Tq, h, d] / k, v [Tk, hk, d] (or [pages, page, hk, d] when paged) uint16 bit patterns; returns (out uint16, lse f32, stats).
    Contiguous: every (sequence, head, 256-row block) is an entry of ONE plan table walked by `g` persistent workgroups;
    paged: one simulated workgroup per block."""
    Tq, h, d = q.shape
    assert d == 128
    paged = block_table is not None
    hk = k.shape[-2]
    mem = Memory()
    qa, ka, va = mem.alloc(q), mem.alloc(k), mem.alloc(v)
    out = np.zeros_like(q)
    oa = mem.alloc(out)
    lse = np.full((h, Tq), np.nan, np.float32)
    la = mem.alloc(lse) if want_lse else 0
    bta = mem.alloc(np.ascontiguousarray(block_table, np.int32)) if paged else 0
    sl2 = np.float32(np.float32(scale) * np.float32(1.4426950408889634))
    stats = {}
    entries, wgs = [], []
    for b in range(len(cu_q) - 1):
        q0, len_q = int(cu_q[b]), int(cu_q[b + 1] - cu_q[b])
        k0, len_k = int(cu_k[b]), int(cu_k[b + 1] - cu_k[b])
        for hq in range(h):
            hkv = hq // (h // hk)
            max_blocks = (int(np.diff(cu_q).max()) + 255) // 256      # shorter sequences leave empty entries, as in the plan table
            for mblk in reversed(range(max_blocks if not paged else (len_q + 255) // 256)):   # longest block first
                common = dict(m0=mblk * 256, len_q=len_q, len_k=len_k, causal=causal, q_addr=qa + (q0 * h + hq) * d * 2,
                              o_addr=oa + (q0 * h + hq) * d * 2, lse_addr=(la + (hq * Tq + q0) * 4) if want_lse else 0,
                              q_stride=h * d * 2, o_stride=h * d * 2, k_stride=hk * d * 2, v_stride=hk * d * 2, scale_log2=sl2, simple=simple, exact=exact)
                if paged:
                    common.update(paged=True, k_addr=ka + hkv * d * 2, v_addr=va + hkv * d * 2, page_size=page_size,
                                  bt_addr=bta + b * block_table.shape[1] * 4, k_page_bytes=page_size * hk * d * 2, v_page_bytes=page_size * hk * d * 2)
                    wgs += run_block_paged(mem, [wave_entry(wave=w, **common) for w in range(4)], dtype, late_dma, reverse)
                else:
                    common.update(k_addr=ka + (k0 * hk + hkv) * d * 2, v_addr=va + (k0 * hk + hkv) * d * 2)
                    entries.append([wave_entry(wave=w, **common) for w in range(4)])
    if not paged:
        # one invalid entry at the end of the list (the padded tail of an XCD's slice): the workgroup stops there
        if entries:
            pad = entries[0][0] * 0
            for name in ("q_stride", "o_stride", "k_stride", "v_stride", "k_tile", "v_tile"):
                pad[PIDX[name]] = entries[0][0][PIDX[name]]
            entries.insert(0, [pad.copy() for _ in range(4)])        # an empty entry first and last: skipped
            entries.append([pad.copy() for _ in range(4)])
        wgs = run_persistent(mem, entries, dtype, late_dma, reverse, g, hook=hook)
    for wg in wgs:
        for w in wg.waves:
            for op, n in w.stats.items():
                stats[op] = stats.get(op, 0) + n
    out = mem.get(oa, np.uint16).reshape(q.shape).copy()
    if want_lse:
        lse = mem.get(la, np.float32).reshape(h, Tq).copy()
    return out, lse, stats
