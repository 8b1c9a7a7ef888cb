"""Host side of the generated prefill kernel, in Python: the per-wavefront parameter block (the same arithmetic as
csrc/prefill_asm.hip's pfa_wave_params) and a driver that runs one workgroup of the generated program on the simulator.
Test infrastructure (tests/test_prefill_asm_sim.py)."""
import numpy as np

from .isa import S
from .kernel import PARAMS, PIDX, PARAM_DWORDS, LDS_PARAMS, LDS_TOTAL, build
from .sim import Memory, Workgroup

BIG = 0x3FFFFFFF


def wave_params(*, wave, m0, len_q, len_k, causal, q_addr, o_addr, lse_addr, q_stride, o_stride, k_addr, v_addr, k_stride,
                v_stride, scale_log2, paged=False, bt_addr=0, page_size=0, k_page_bytes=0, v_page_bytes=0, simple=False, exact=False):
    """q_addr / o_addr: address of row 0 of the SEQUENCE for this head (bytes); lse_addr: address of row 0's LSE or 0.
    k_addr / v_addr: contiguous: row 0 of the sequence for the kv head; paged: cache base + head offset."""
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
    p = np.zeros(PARAM_DWORDS, np.uint32)

    def put(name, val):
        p[PIDX[name]] = np.uint32(int(val) & 0xFFFFFFFF)

    def put64(name, val):
        put(name + "_lo", val & 0xFFFFFFFF)
        put(name + "_hi", val >> 32)
    lims, tms = [], []
    for s, sl in enumerate((s0, s1)):
        put64(f"q{s}", q_addr + sl["r0"] * q_stride)
        put64(f"o{s}", o_addr + sl["r0"] * o_stride)
        put64(f"lse{s}", lse_addr + sl["r0"] * 4 if lse_addr else 0)
        put(f"rows{s}", sl["rows"])
        lim = sl["r0"] + shift if causal else len_k - 1
        lims.append(lim)
        minlim = min(lim, len_k - 1)
        tms.append(max(0, (minlim + 1) >> 6) if minlim >= 0 else 0)
        put(f"lim{s}", lim)
    put("q_stride", q_stride)
    put("o_stride", o_stride)
    put64("k", k_addr)
    put64("v", v_addr)
    put("k_stride", k_stride)
    put("v_stride", v_stride)
    put("k_bytes", (len_k - 1) * k_stride + 256 if len_k > 0 else 0)
    put("v_bytes", (len_k - 1) * v_stride + 256 if len_k > 0 else 0)
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
    if paged:
        put64("bt", bt_addr)
        put("page_shift", int(page_size).bit_length() - 1)
        put("k_page_bytes", k_page_bytes)
        put("v_page_bytes", v_page_bytes)
    put("wave", wave)
    put("thr", np.float32(np.float32(8.0) / np.float32(scale_log2) if exact else 8.0).view(np.uint32))
    return p


_progs = {}


def program(dtype, paged, exact=False):
    key = (dtype, paged, exact)
    if key not in _progs:
        _progs[key] = build(dtype, paged, param_sgpr=S(4), exact=exact)
    return _progs[key]


def run_block(mem, params4, dtype="bf16", paged=False, late_dma=True, reverse=False, exact=False):
    """one workgroup: params4 = the four wavefronts' parameter blocks"""
    prog, _ = program(dtype, paged, exact)
    wg = Workgroup(prog, mem, n_waves=4, lds_bytes=LDS_TOTAL, late_dma=late_dma, reverse=reverse)
    for w in range(4):
        wg.lds[LDS_PARAMS + 256 * w:LDS_PARAMS + 256 * (w + 1)] = params4[w].view(np.uint8)
        wg.waves[w].s[4] = LDS_PARAMS + 256 * w
    wg.run()
    return wg


def prefill_varlen(q, k, v, cu_q, cu_k, scale, causal, dtype="bf16", want_lse=False, block_table=None, page_size=0, late_dma=True,
                   reverse=False, simple=False, only=None, exact=False):
    """q [Tq, h, d] / k, v [Tk, hk, d] (or [pages, page, hk, d] when paged) uint16 bit patterns; returns (out uint16, lse f32)
    -- every (sequence, head, 256-row block) is one simulated workgroup.  `only` = optional set of (b, hq, mblk) to run."""
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
    for b in range(len(cu_q) - 1):
        q0, len_q = int(cu_q[b]), int(cu_q[b + 1] - cu_q[b])
        k0, len_k = int(cu_k[b]), int(cu_k[b + 1] - cu_k[b])
        for hq in range(h):
            hkv = hq // (h // hk)
            for mblk in range((len_q + 255) // 256):
                if only is not None and (b, hq, mblk) not in only:
                    continue
                common = dict(m0=mblk * 256, len_q=len_q, len_k=len_k, causal=causal, q_addr=qa + (q0 * h + hq) * d * 2,
                              o_addr=oa + (q0 * h + hq) * d * 2, lse_addr=(la + (hq * Tq + q0) * 4) if want_lse else 0,
                              q_stride=h * d * 2, o_stride=h * d * 2, k_stride=hk * d * 2, v_stride=hk * d * 2, scale_log2=sl2, simple=simple, exact=exact)
                if paged:
                    common.update(paged=True, k_addr=ka + hkv * d * 2, v_addr=va + hkv * d * 2, page_size=page_size,
                                  bt_addr=bta + b * block_table.shape[1] * 4, k_page_bytes=page_size * hk * d * 2, v_page_bytes=page_size * hk * d * 2)
                else:
                    common.update(k_addr=ka + (k0 * hk + hkv) * d * 2, v_addr=va + (k0 * hk + hkv) * d * 2)
                wg = run_block(mem, [wave_params(wave=w, **common) for w in range(4)], dtype, paged, late_dma, reverse, exact)
                for w in wg.waves:
                    for op, n in w.stats.items():
                        stats[op] = stats.get(op, 0) + n
    out = mem.get(oa, np.uint16).reshape(q.shape).copy()
    if want_lse:
        lse = mem.get(la, np.float32).reshape(h, Tq).copy()
    return out, lse, stats
