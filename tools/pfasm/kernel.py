"""Generator of the hand-scheduled gfx950 FlashAttention-2 prefill kernel (d = 128, bf16 / f16, contiguous or paged K/V).

Replaces the inner loop of /root/reference/csrc/kernels/flash_fwd_kernel.h:56-500 (compute_attn_1rowblock) for the common
shapes; csrc/prefill_asm.hip wraps the text produced here in ONE asm statement and keeps everything else (work mapping,
per-wave parameter block) in C++.

Structure: workgroup = 4 wavefronts = 256 query rows of one (sequence, q head); ONE wavefront per SIMD with
the whole 512-register file; wavefront w owns two 32-row blocks ("slots": the block with fewer K/V tiles first -- under a
causal mask rows 32w.. and 32(7-w).., so all four wavefronts do the same work).  Swapped products S^T = K.Q^T and
O^T = V^T.P^T (v_mfma_f32_32x32x16): a lane's accumulator registers all belong to one query row.

    AGPR   O^T[slot][db]  a[0:127]    Q~^T[slot][j]  a[128:191]   K fragments of one tile [half][j]  a[192:255]
    VGPR   S banks v[0:63] / v[64:127] (tile t in bank t & 1; P is packed in place), C-init tuples (-m) v[128:159],
           V^T operand window v[160:191], addresses / softmax state / temporaries v[192:255]

Q is pre-multiplied by scale.log2(e) and rounded to the storage type once per block, and the first MFMA of every S chain
takes C = -m (the running reference of the row), so S' = s~ - m leaves the matrix pipe ready for v_exp_f32: 3 VALU
instructions per score element (max3/2, exp, row-sum add, cvt_pk/2) instead of 4-5.  The reference is raised only when a
tile's maximum exceeds it by more than 8 (exp2 domain): a rare, separate block (RESC).

Per K/V tile (64 keys) two phases of 32 MFMAs, software-pipelined over tiles, instruction streams merged by an EDF list
scheduler (`schedule`) with at most CAP fillers per MFMA gap:
    phase 1:  S(t+1) = K(t+1).Q~^T       ||  P(t) = exp2(S'(t)), row sums, pack      ||  V(t) transposing LDS reads
    phase 2:  O += V(t)^T.P(t)^T         ||  [mask] row max of S'(t+1), decision     ||  K(t+2) LDS reads -> AGPRs, LDS-DMA of
                                                                                          K(t+4), V(t+2), ring bookkeeping
K/V tiles travel by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per instruction, bounds-checked by the descriptor: rows
past the sequence read as zeros) into two 3-slot rings (96 KiB); one s_waitcnt vmcnt(8) + s_barrier per tile.
"""
from .isa import Program, Ins, Reg, V, A, S, VCC, M0, f32_bits

NEG_INF = float("-inf")

# ---- parameter block: 64 dwords per wavefront, written by csrc/prefill_asm.hip (PfaWaveParams) -------------------------------
PARAMS = [
    "q0_lo", "q0_hi", "q1_lo", "q1_hi",             # address of row 0 of the slot's q block (this head)
    "o0_lo", "o0_hi", "o1_lo", "o1_hi",             # same for the output
    "lse0_lo", "lse0_hi", "lse1_lo", "lse1_hi",     # row 0 of the slot in the LSE array (4-byte row stride), 0 = none
    "q_stride", "o_stride",                         # row strides, bytes
    "rows0", "rows1",                               # valid rows of the slot (0 .. 32)
    "k_lo", "k_hi", "v_lo", "v_hi",                 # contiguous: row 0 of this sequence and kv head; paged: cache base + head offset
    "k_stride", "v_stride",                         # K/V row strides, bytes
    "k_bytes", "v_bytes",                           # contiguous: bytes from row 0 to the end of the sequence's last row; paged: unused
    "len_k", "n_tiles", "n0", "n1",                 # keys; K/V tiles of the workgroup; tiles slot 0 / slot 1 take part in
    "n_steady", "tm0", "tm1", "tmm",                # leading iterations of the plain variant; first tile that needs a mask (slot 0, 1, min)
    "lim0", "lim1", "lim_step",                     # last visible key of the slot's row 0; +lim_step per row (1 causal, 0 not)
    "scale_log2",                                   # f32 bits
    "bt_lo", "bt_hi", "page_shift",                 # paged: this sequence's block table, log2(page size)
    "k_page_bytes", "v_page_bytes",                 # paged: page strides, bytes
    "wave",                                         # 0 .. 3
    "thr",                                          # f32 bits: the deferred-raise threshold in the domain S' lives in (8, or 8 / scale_log2 when exact)
    "dbg_lo", "dbg_hi",                             # timing builds: 32 bytes per wavefront for the phase timers
]
PIDX = {n: i for i, n in enumerate(PARAMS)}
PARAM_DWORDS = 64
LDS_RING = 96 * 1024            # K slots at 0 / 16K / 32K, V slots at 48K / 64K / 80K
LDS_PARAMS = LDS_RING           # 4 x 256 bytes
LDS_TOTAL = LDS_RING + 4 * PARAM_DWORDS * 4
SLOT = 16384
CAP = 5                         # fillers per MFMA gap the scheduler aims for


# ---- register map ----------------------------------------------------------------------------------------------------------------
def OA(s, db):
    return A((s * 4 + db) * 16, 16)


def QA(s, j):
    return A(128 + (s * 8 + j) * 4, 4)


def KA(h, j):
    return A(192 + (h * 8 + j) * 4, 4)


def T(bank, s, h):
    return V(bank * 64 + (s * 2 + h) * 16, 16)


def CI(s):
    return V(128 + 16 * s, 16)


def VV(i):
    return V(160 + 4 * i, 4)


V_HI4 = V(192)          # 4 * (lane >> 5)
V_LQ = V(193)           # lane & 31
KAD = [V(194 + j) for j in range(8)]
VAD = [V(202 + db) for db in range(4)]
VOFK = [V(206 + u) for u in range(4)]
VOFV = [V(210 + u) for u in range(4)]


def LS(s, c):
    return V(214 + 2 * s + c)


MX = [V(218), V(219)]
MXB = [V(220), V(221)]
THR = [V(222), V(223)]
MREF = [V(224), V(225)]
LIM = [V(226), V(227)]
V_NEGINF = V(228)
LIMREL = [V(229), V(230)]
TMP = [V(232 + i) for i in range(24)]

_snext = [36]


def _salloc(n=1, align=1):
    i = (_snext[0] + align - 1) // align * align
    _snext[0] = i + n
    assert _snext[0] <= 100, "out of SGPRs"
    return S(i, n) if n > 1 else S(i)


S_KDESC = _salloc(4, 4)
S_VDESC = _salloc(4, 4)
S_ODESC = _salloc(4, 4)
S_PAIR = _salloc(2, 2)          # 64-bit scratch (compare results, addresses)
S_VSTRIDE = _salloc()
S_KSTRIDE = _salloc()
S_KTILE = _salloc()             # 64 * k_stride
S_VTILE = _salloc()
S_KREC = _salloc()              # signed bytes left from the descriptor base to the end of the sequence
S_VREC = _salloc()
S_NT, S_N0, S_N1, S_NST, S_TM0, S_TM1, S_TMM = (_salloc() for _ in range(7))
S_T, S_T1 = _salloc(), _salloc()
S_RBASE, S_DBASE, S_DELTA, S_KDMA = (_salloc() for _ in range(4))
S_W1024 = _salloc()
S_LENK = _salloc()
S_RET = _salloc()
S_ROWS = [_salloc(), _salloc()]
S_TMP = [_salloc() for _ in range(8)]
S_SCALE = _salloc()
S_THRV = _salloc()
S_QST, S_OST = _salloc(), _salloc()
S_PTR = _salloc(2, 2)           # parameter reload scratch
S_BT = _salloc(2, 2)
S_PSHIFT, S_KPAGE, S_VPAGE = _salloc(), _salloc(), _salloc()
S_KBASE = _salloc(2, 2)
S_VBASE = _salloc(2, 2)
S_KVT = _salloc()               # paged: first key of the next K tile to request
S_VVT = _salloc()
SGPR_FIRST, SGPR_LAST = 36, _snext[0] - 1
# timing builds (contiguous K/V only) reuse the paged-addressing registers
S_TACC = [S_BT[0], S_BT[1], S_PSHIFT, S_KPAGE, S_VPAGE, S_KBASE[0], S_KBASE[1], S_KVT]
S_TNOW = S_VBASE
S_TLAST, S_TT = S_VVT, S_PTR[0]


class Item:
    """One or more instructions the scheduler keeps together; release / deadline are MFMA-gap indices."""
    __slots__ = ("ins", "release", "deadline")

    def __init__(self, ins, release=-1, deadline=None):
        self.ins = ins if isinstance(ins, list) else [ins]
        self.release, self.deadline = release, deadline


def schedule(mfmas, streams, cap=CAP, log=None):
    """Merge FIFO filler streams into the gaps of an MFMA sequence: gap g follows MFMA g (gap -1 precedes the first).
    Earliest-deadline-first among the released stream heads, about total/gaps (<= cap unless deadlines force more) per gap."""
    n = len(mfmas)
    out = []
    heads = [0] * len(streams)
    total = sum(len(it.ins) for st in streams for it in st)
    for st in streams:
        for it in st:
            if it.deadline is None:
                it.deadline = n - 1
    placed = 0
    if n == 0:
        for st in streams:
            for it in st:
                out.extend(it.ins)
        return out
    for g in range(-1, n):
        if g >= 0:
            out.append(mfmas[g])
        gaps_left = n - g          # including this one
        quota = -(-(total - placed) // gaps_left) if g >= 0 else 0
        quota = min(quota, cap) if g >= 0 else 0
        used = 0
        while True:
            best, bi = None, -1
            for si, st in enumerate(streams):
                if heads[si] < len(st):
                    it = st[heads[si]]
                    if it.release <= g and (best is None or it.deadline < best.deadline):
                        best, bi = it, si
            if best is None:
                break
            must = best.deadline <= g
            if not must and used + len(best.ins) > quota:
                break
            out.extend(best.ins)
            used += len(best.ins)
            placed += len(best.ins)
            heads[bi] += 1
        if log is not None:
            log.append(used)
    # anything whose release lies beyond the last gap
    for si, st in enumerate(streams):
        while heads[si] < len(st):
            out.extend(st[heads[si]].ins)
            heads[si] += 1
    return out


LDS_OPS = ("ds_read_b128", "ds_read_b64_tr_b16", "ds_read_b32")


def insert_lds_waits(ins_list):
    """Counted s_waitcnt lgkmcnt(N) in front of the first instruction that touches a register an LDS read is still loading
    (LDS operations return in order).  Straight-line blocks only; no scalar loads may be outstanding inside one."""
    out, queue = [], []          # queue: sets of (kind, idx) per outstanding LDS op, oldest first
    for i in ins_list:
        if i.op in ("label",) or i.op.startswith(("s_branch", "s_cbranch")):
            if queue:
                out.append(Ins("s_waitcnt", lgkmcnt=0))
                queue = []
            out.append(i)
            continue
        if i.op == "s_waitcnt" and i.mods.get("lgkmcnt") is not None:
            queue = queue[len(queue) - i.mods["lgkmcnt"]:] if i.mods["lgkmcnt"] else []
            out.append(i)
            continue
        regs = set()
        for o in i.ops:
            if isinstance(o, Reg) and o.kind in "va":
                regs.update(o.regs())
        hit = -1
        for qi, qs in enumerate(queue):
            if qs & regs:
                hit = qi
        if hit >= 0:
            keep = min(len(queue) - hit - 1, 15)                  # lgkmcnt is a 4-bit counter
            out.append(Ins("s_waitcnt", lgkmcnt=keep))
            queue = queue[len(queue) - keep:] if keep else []
        out.append(i)
        if i.op in LDS_OPS:
            queue.append(set(i.ops[0].regs()))
        elif i.op.startswith("ds_write"):
            queue.append(set())
    return out, len(queue)


class Builder:
    def __init__(self, dtype="bf16", paged=False, param_sgpr=S(4), exact=False, timing=False):
        """exact = False: Q pre-multiplied by scale.log2(e) and rounded once (S' in the exp2 domain, 3 VALU per score);
        exact = True: Q as it is, S' = s - m in the raw domain and one v_mul_f32 by scale.log2(e) in front of every v_exp_f32 --
        the reference's arithmetic to the last rounding, for rows whose softmax mass sits on a few keys."""
        assert dtype in ("bf16", "f16")
        assert not (timing and paged)
        self.dtype, self.paged, self.exact, self.timing = dtype, paged, exact, timing
        self.mfma = "v_mfma_f32_32x32x16_" + dtype
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.param_sgpr = param_sgpr
        self.p = Program()
        self.sched_log = {}
        self.ret_sites = []       # (id, label) of RESC call sites

    # ------------------------------------------------------------------------------------------------------------------------
    def e(self, op, *ops, **mods):
        return self.p.emit(op, *ops, **mods)

    def stamp(self, k):
        """timing builds: cycles since the previous stamp -> accumulator k (s_memtime: the wait drains the LDS queue too)"""
        if not self.timing:
            return []
        return [Ins("s_memtime", S_TNOW), Ins("s_waitcnt", lgkmcnt=0), Ins("s_sub_u32", S_TT, S_TNOW[0], S_TLAST),
                Ins("s_add_u32", S_TACC[k], S_TACC[k], S_TT), Ins("s_mov_b32", S_TLAST, S_TNOW[0])]

    def nop(self, states):
        """at least `states` wait states"""
        while states > 0:
            n = min(states, 16)
            self.e("s_nop", n - 1)
            states -= n

    # ---- streams ---------------------------------------------------------------------------------------------------------------
    def qk_mfmas(self, bank_n, slots):
        out = []
        for j in range(8):
            for h in range(2):
                for s in slots:
                    d = T(bank_n, s, h)
                    out.append(Ins(self.mfma, d, KA(h, j), QA(s, j), CI(s) if j == 0 else d))
        return out

    def pv_mfmas(self, bank_c, slots):
        out = []
        for o in range(16):
            h, kk, db = o >> 3, (o >> 2) & 1, o & 3
            for s in slots:
                out.append(Ins(self.mfma, OA(s, db), VV(o % 8), T(bank_c, s, h)[8 * kk:8 * kk + 4], OA(s, db)))
        return out

    def exp_items(self, bank_c, s, h, release=-1, deadline=None):
        t = T(bank_c, s, h)
        seq = []
        for kk in range(2):
            e = [t[8 * kk + i] for i in range(8)]
            pos = [t[8 * kk + c] for c in range(4)]
            l0, l1 = LS(s, 0), LS(s, 1)

            def ad(i):
                return Ins("v_add_f32", (l0, l1)[i & 1], (l0, l1)[i & 1], e[i])

            def cv(c):
                return Ins(self.cvt, pos[c], e[2 * c], e[2 * c + 1])
            if self.exact:     # the multiplies run one pair ahead of their v_exp_f32
                m = lambda x: Ins("v_mul_f32", x, S_SCALE, x)
                x2 = lambda x: Ins("v_exp_f32", x, x)
                seq += [m(e[0]), m(e[1]), m(e[2]), m(e[3]), x2(e[0]), x2(e[1]), x2(e[2]), x2(e[3]), m(e[4]), m(e[5]), ad(0), ad(1), cv(0),
                        x2(e[4]), x2(e[5]), m(e[6]), m(e[7]), ad(2), ad(3), cv(1), x2(e[6]), x2(e[7]), ad(4), ad(5), cv(2), ad(6), ad(7), cv(3)]
            else:
                x2 = lambda x: Ins("v_exp_f32", x, x)
                seq += [x2(e[0]), x2(e[1]), x2(e[2]), x2(e[3]), ad(0), ad(1), cv(0), x2(e[4]), x2(e[5]), ad(2), ad(3), cv(1),
                        x2(e[6]), x2(e[7]), ad(4), ad(5), cv(2), ad(6), ad(7), cv(3)]
        return [Item(i, release, deadline) for i in seq]

    def v_read(self, o):
        h, kk, db = o >> 3, (o >> 2) & 1, o & 3
        off = (32 * h + 16 * kk) * 256
        dst = VV(o % 8)
        return [Ins("ds_read_b64_tr_b16", dst[0:2], VAD[db], offset=off), Ins("ds_read_b64_tr_b16", dst[2:4], VAD[db], offset=off + 2048)]

    def k_reads(self):
        return [Ins("ds_read_b128", KA(h, j), KAD[j], offset=h * 8192) for h in range(2) for j in range(8)]

    def max_items(self, bank_n, s, release):
        out = []
        for h, acc in ((0, MX[s]), (1, MXB[s])):
            t = T(bank_n, s, h)
            out.append(Ins("v_max3_f32", acc, t[0], t[1], t[2]))
            for k in range(6):
                out.append(Ins("v_max3_f32", acc, acc, t[3 + 2 * k], t[4 + 2 * k]))
            out.append(Ins("v_max_f32", acc, acc, t[15]))
        # interleave the two chains, then combine
        a, b = out[:8], out[8:]
        seq = [x for pair in zip(a, b) for x in pair] + [Ins("v_max_f32", MX[s], MX[s], MXB[s])]
        return [Item(i, release) for i in seq]

    def mask_items(self, bank_n, s, release):
        """keys beyond the lane's limit -> -inf: key(r, h) = kv1 + 32h + (r&3) + 8(r>>2) + 4hi > LIM  <=>  const > LIMREL"""
        seq = [Ins("v_subrev_u32", LIMREL[s], S_T1, LIM[s]),          # LIM - 64 (t+1) ... S_T1 holds kv1 here (see iteration)
               Ins("v_sub_u32", LIMREL[s], LIMREL[s], V_HI4)]
        for h in range(2):
            t = T(bank_n, s, h)
            for r in range(16):
                c = 32 * h + (r & 3) + 8 * (r >> 2)
                seq.append(Ins("v_cmp_gt_i32", VCC, c, LIMREL[s]))
                seq.append(Ins("v_cndmask_b32", t[r], t[r], V_NEGINF, VCC))
        items = [Item(seq[0], release), Item(seq[1], release)]
        for k in range(2, len(seq), 2):
            items.append(Item([seq[k], seq[k + 1]], release))       # the compare and its select stay together (VCC)
        return items

    def dma_tile(self, desc, voffs, lds_extra):
        """four 1 KiB pieces of this wavefront: rows 16u + 4w .. +3 of the tile -> ring slot S_KDMA (+ lds_extra for V)"""
        items = []
        for u in range(4):
            items.append(Item([Ins("s_add_u32", M0, S_KDMA, lds_extra + u * 4096), Ins("s_nop", 0),
                               Ins("buffer_load_dwordx4", voffs[u], desc, 0, offen=True, lds=True)]))
        return items

    def desc_advance(self, desc, tile, rec):
        return [Item(Ins("s_add_u32", desc[0], desc[0], tile)), Item(Ins("s_addc_u32", desc[1], desc[1], 0)),
                Item(Ins("s_sub_i32", rec, rec, tile)), Item(Ins("s_max_i32", desc[2], rec, 0))]

    # ---- paged K/V: the descriptor of every piece comes from the block table -------------------------------------------------------
    def paged_piece_setup(self, u, key0, base, page_bytes, stride, desc, pg):
        """SALU: descriptor `desc` <- page of key (key0 + 16u) of this wavefront's rows; pg = SGPR holding the page id.
        voffset already carries (4w + lane/16) * stride + chunk; the descriptor base carries the page and the row of the
        16-row group inside the page; records cut at the sequence end (rows past it read as zeros)."""
        t0, t1, t2, t3 = S_TMP[0], S_TMP[1], S_TMP[2], S_TMP[3]
        return [
            Ins("s_add_u32", t0, key0, 16 * u),                          # first key of the 16-row group
            Ins("s_lshl_b32", t1, 1, S_PSHIFT), Ins("s_sub_u32", t1, t1, 1), Ins("s_and_b32", t1, t0, t1),   # row inside the page
            Ins("s_mul_i32", t1, t1, stride),
            Ins("s_mul_i32", t2, pg, page_bytes), Ins("s_mul_hi_u32", t3, pg, page_bytes),
            Ins("s_add_u32", t2, t2, t1), Ins("s_addc_u32", t3, t3, 0),
            Ins("s_add_u32", desc[0], base[0], t2), Ins("s_addc_u32", desc[1], base[1], t3),
            Ins("s_sub_i32", t0, S_LENK, t0),                            # keys left from the group's first row
            Ins("s_max_i32", t0, t0, 0), Ins("s_min_i32", t0, t0, 16),
            Ins("s_mul_i32", t0, t0, stride),                            # 16 rows or fewer: [0, rows * stride) covers them (the last row's tail
            Ins("s_mov_b32", desc[2], t0),                               #  beyond 256 bytes belongs to other heads and is never addressed)
        ]

    def paged_tile(self, key0, is_v):
        """page lookups + 4 pieces of one tile (paged): scalar loads first, one wait, then per piece setup + DMA"""
        base, page_bytes, stride, desc, voffs = ((S_VBASE, S_VPAGE, S_VSTRIDE, S_VDESC, VOFV) if is_v
                                                 else (S_KBASE, S_KPAGE, S_KSTRIDE, S_KDESC, VOFK))
        seq = []
        pgs = [S_TMP[4], S_TMP[5], S_TMP[6], S_TMP[7]]
        # clamp the looked-up key to the last key of the sequence: never index the table past the sequence's pages
        for u in range(4):
            seq += [Ins("s_add_u32", S_TMP[0], key0, 16 * u), Ins("s_sub_u32", S_TMP[1], S_LENK, 1), Ins("s_max_i32", S_TMP[1], S_TMP[1], 0),
                    Ins("s_min_u32", S_TMP[0], S_TMP[0], S_TMP[1]), Ins("s_lshr_b32", S_TMP[0], S_TMP[0], S_PSHIFT),
                    Ins("s_lshl_b32", S_TMP[0], S_TMP[0], 2), Ins("s_load_dword", pgs[u], S_BT, S_TMP[0])]
        seq.append(Ins("s_waitcnt", lgkmcnt=0))
        for u in range(4):
            seq += self.paged_piece_setup(u, key0, base, page_bytes, stride, desc, pgs[u])
            seq += [Ins("s_add_u32", M0, S_KDMA, (3 * SLOT if is_v else 0) + u * 4096), Ins("s_nop", 0),
                    Ins("buffer_load_dwordx4", voffs[u], desc, 0, offen=True, lds=True)]
        seq.append(Ins("s_add_u32", key0, key0, 64))
        return seq

    # ---- one iteration -----------------------------------------------------------------------------------------------------------
    def iteration(self, p, c, x, mk):
        """tile t in bank p: cur slots (c = 2: both, 1: slot 1, 0: none) exponentiate + P.V; next slots (x) get S(t+1)."""
        name = f"IT_{p}_{c}{x}{'m' if mk else '0'}"
        cur = {2: [0, 1], 1: [1], 0: []}[c]
        nxt = {2: [0, 1], 1: [1], 0: []}[x]
        bank_c, bank_n = p, p ^ 1
        P = self.p
        P.label(name)
        # ---------------- phase 1 ----------------
        mf1 = self.qk_mfmas(bank_n, nxt)
        n1 = len(mf1)
        exp_h0 = [it for s in cur for it in self.exp_items(bank_c, s, 0)]
        # interleave the slots' streams so that both finish early (P of half 0 is needed first by P.V)
        if len(cur) == 2:
            a, b = self.exp_items(bank_c, 0, 0), self.exp_items(bank_c, 1, 0)
            exp_h0 = [y for pair in zip(a, b) for y in pair]
            a, b = self.exp_items(bank_c, 0, 1), self.exp_items(bank_c, 1, 1)
            exp_h1 = [y for pair in zip(a, b) for y in pair]
        else:
            exp_h1 = [it for s in cur for it in self.exp_items(bank_c, s, 1)]
        vreads = []
        if cur:
            for o in range(8):
                r = self.v_read(o)
                vreads += [Item(r[0]), Item(r[1])]
        # ---------------- phase 2 ----------------
        mf2 = self.pv_mfmas(bank_c, cur)
        n2 = len(mf2)
        per = len(cur)
        v2 = []
        if cur:
            for o in range(8, 16):
                r = self.v_read(o)
                rel = (o - 8 + 1) * per - 1               # after the MFMAs of operand o - 8 (same window slot)
                dl = max(o * per - 3, rel)                # land a few gaps ahead of operand o's first MFMA
                v2 += [Item(r[0], rel, dl), Item(r[1], rel, dl)]
        mm = []
        rel_m = 4 if n2 else -1
        for s in nxt:
            if mk:
                mm += self.mask_items(bank_n, s, rel_m)
            mm += self.max_items(bank_n, s, rel_m)
        kr = []
        if nxt:
            kr = [Item(i, 1 if n2 else -1) for i in self.k_reads()]
        ring = [Item(Ins("s_mov_b32", S_TMP[0], SLOT), -1, 1), Item(Ins("s_cmp_eq_u32", S_RBASE, 2 * SLOT), -1, 1),
                Item(Ins("s_cselect_b32", S_DELTA, -2 * SLOT, S_TMP[0]), -1, 1)]
        adv = [Item(Ins("v_add_u32", KAD[j], S_DELTA, KAD[j])) for j in range(8)]
        vadv = [Item(Ins("v_add_u32", VAD[db], S_DELTA, VAD[db]), release=max(15 * per - 1, -1)) for db in range(4)]
        if self.paged:
            dma = []           # issued as a scalar block in front of phase 2 (see below)
        else:
            dma = self.dma_tile(S_KDESC, VOFK, 0) + self.desc_advance(S_KDESC, S_KTILE, S_KREC) + \
                self.dma_tile(S_VDESC, VOFV, 3 * SLOT) + self.desc_advance(S_VDESC, S_VTILE, S_VREC)
        for it in dma:
            it.release = max(it.release, 2 if n2 else -1)
        # ---- how much of the second half's softmax rides in phase 1: the same filler density in both phases ----
        cnt = lambda items: sum(len(it.ins) for it in items)
        f1 = cnt(exp_h0) + cnt(vreads)
        f2 = cnt(v2) + cnt(mm) + cnt(ring) + cnt(kr) + cnt(adv) + cnt(vadv) + cnt(dma)
        n_e1 = len(exp_h1)
        dens = -(-(f1 + f2 + n_e1) // max(n1 + n2, 1))                # fillers per MFMA gap over the whole iteration
        move = max(0, min(n_e1, dens * n1 - f1)) if (n1 and cur) else 0
        e1_in_p1, e1_in_p2 = exp_h1[:move], exp_h1[move:]
        for it in exp_h0 + vreads + e1_in_p1:
            it.deadline = max(n1 - 1, -1)
        for it in e1_in_p2:
            it.deadline = max(8 * per - 2, -1)            # P of half 1 feeds P.V from operand 8 on
        cap1 = max(dens, -(-(f1 + move) // max(n1, 1)))
        cap2 = max(dens, -(-(f2 + n_e1 - move) // max(n2, 1)))
        log1, log2 = [], []
        body1 = schedule(mf1, [exp_h0, vreads, e1_in_p1], cap=cap1, log=log1)
        body2 = schedule(mf2, [e1_in_p2, v2, mm, ring + kr + adv, dma, vadv], cap=cap2, log=log2)
        self.sched_log[name] = (log1, log2)
        blk = self.stamp(4)
        if mk:
            blk.append(Ins("s_lshl_b32", S_T1, S_T, 6))
            blk.append(Ins("s_add_u32", S_T1, S_T1, 64))       # kv1 = 64 (t + 1): LIMREL = LIM - kv1 - 4hi
        blk += body1
        blk += self.stamp(0)
        if self.paged:
            blk.append(Ins("s_waitcnt", lgkmcnt=0))
            blk += self.paged_tile(S_KVT, False) + self.paged_tile(S_VVT, True)
        blk += body2
        blk += self.stamp(1)
        blk, _ = insert_lds_waits(blk)
        P.extend(blk)
        # ---------------- decision, ring rotation, barrier ----------------
        if nxt:
            site = len(self.ret_sites)
            ret = f"RET_{site}"
            stub = f"STUB_{site}"
            if len(nxt) == 2:
                self.e("v_cmp_gt_f32", S_PAIR, MX[0], THR[0])
                self.e("v_cmp_gt_f32", VCC, MX[1], THR[1])
                self.e("s_or_b64", S_PAIR, S_PAIR, VCC)
            else:
                self.e("v_cmp_gt_f32", S_PAIR, MX[1], THR[1])
                self.e("s_or_b64", S_PAIR, S_PAIR, S_PAIR)
            self.e("s_cbranch_scc1", stub)
            P.label(ret)
            self.ret_sites.append((site, ret, stub, bank_n, x))
        self.e("s_mov_b32", S_DBASE, S_RBASE)
        self.e("s_add_u32", S_RBASE, S_RBASE, S_DELTA)
        self.e("s_add_u32", S_KDMA, S_DBASE, S_W1024)
        self.e("s_waitcnt", vmcnt=8, lgkmcnt=0)
        P.extend(self.stamp(2))
        self.e("s_barrier")
        P.extend(self.stamp(3))
        if self.timing:
            self.e("s_add_u32", S_TACC[7], S_TACC[7], 1)
        self.e("s_add_u32", S_T, S_T, 1)
        if (c, x, mk) == (2, 2, False):
            self.e("s_cmp_lt_u32", S_T, S_NST)
            self.e("s_cbranch_scc1", f"IT_{p ^ 1}_220")
        self.e("s_branch", f"DISP_{p ^ 1}")

    def idle_iteration(self):
        """a wavefront with no active slot: its share of the LDS-DMA, the barrier"""
        P = self.p
        P.label("IT_00")
        if self.paged:
            P.extend(self.paged_tile(S_KVT, False) + self.paged_tile(S_VVT, True))
        else:
            for it in self.dma_tile(S_KDESC, VOFK, 0) + self.desc_advance(S_KDESC, S_KTILE, S_KREC) + \
                    self.dma_tile(S_VDESC, VOFV, 3 * SLOT) + self.desc_advance(S_VDESC, S_VTILE, S_VREC):
                P.extend(it.ins)
        self.e("s_mov_b32", S_TMP[0], SLOT)
        self.e("s_cmp_eq_u32", S_RBASE, 2 * SLOT)
        self.e("s_cselect_b32", S_DELTA, -2 * SLOT, S_TMP[0])
        self.e("s_mov_b32", S_DBASE, S_RBASE)
        self.e("s_add_u32", S_RBASE, S_RBASE, S_DELTA)
        self.e("s_add_u32", S_KDMA, S_DBASE, S_W1024)
        self.e("s_waitcnt", vmcnt=8, lgkmcnt=0)
        self.e("s_barrier")
        self.e("s_add_u32", S_T, S_T, 1)
        self.e("s_cmp_lt_u32", S_T, S_NT)
        self.e("s_cbranch_scc1", "IT_00")
        self.e("s_branch", "EPILOGUE")

    def dispatcher(self, p):
        P = self.p
        e = self.e
        P.label(f"DISP_{p}")
        e("s_cmp_ge_u32", S_T, S_NT)
        e("s_cbranch_scc1", "EPILOGUE")
        e("s_cmp_lt_u32", S_T, S_NST)
        e("s_cbranch_scc1", f"IT_{p}_220")
        e("s_add_u32", S_T1, S_T, 1)
        e("s_cmp_lt_u32", S_T, S_N0)
        e("s_cbranch_scc0", f"D{p}_CNOT2")
        e("s_cmp_lt_u32", S_T1, S_N0)
        e("s_cbranch_scc0", f"D{p}_C2XN2")
        e("s_cmp_ge_u32", S_T1, S_TMM)
        e("s_cbranch_scc1", f"IT_{p}_22m")
        e("s_branch", f"IT_{p}_220")
        P.label(f"D{p}_C2XN2")
        e("s_cmp_lt_u32", S_T1, S_N1)
        e("s_cbranch_scc0", f"IT_{p}_200")
        e("s_cmp_ge_u32", S_T1, S_TM1)
        e("s_cbranch_scc1", f"IT_{p}_21m")
        e("s_branch", f"IT_{p}_210")
        P.label(f"D{p}_CNOT2")
        e("s_cmp_lt_u32", S_T, S_N1)
        e("s_cbranch_scc0", "IT_00")
        e("s_cmp_lt_u32", S_T1, S_N1)
        e("s_cbranch_scc0", f"IT_{p}_100")
        e("s_cmp_ge_u32", S_T1, S_TM1)
        e("s_cbranch_scc1", f"IT_{p}_11m")
        e("s_branch", f"IT_{p}_110")

    # ---- the rare block: raise the rows' reference --------------------------------------------------------------------------------
    def rescale(self, bank_n, x):
        """S'(t+1) of the slots in x sits in bank_n relative to the OLD references.  Per row (both lanes of a pair agree):
        r = row max';  raise = r > THR (THR = -inf while the row has no reference yet, 8 afterwards);  delta = raise ? r : 0;
        m += delta;  C-init = -m;  S' -= delta;  alpha = had a reference ? 2^-delta : 1;  O *= alpha;  l *= alpha."""
        e = self.e
        slots = {2: [0, 1], 1: [1]}[x]
        self.p.label(f"RESC_{bank_n}_{x}")
        self.nop(MFMA_SAFE)
        for s in slots:
            r, r2, dl, al, one = TMP[0], TMP[1], TMP[2], TMP[3], TMP[4]
            e("v_mov_b32", r, MX[s])
            e("v_mov_b32", r2, MX[s])
            self.nop(2)
            e("v_permlane32_swap_b32", r, r2)
            e("v_max_f32", r, r, r2)                               # the row's max' in both lanes of the pair
            e("v_cmp_gt_f32", VCC, r, THR[s])                      # raise?
            e("v_mov_b32", dl, 0)
            e("v_cndmask_b32", dl, dl, r, VCC)                     # delta
            if self.exact:
                e("v_mul_f32", al, S_SCALE, dl)
                e("v_sub_f32", al, 0, al)
            else:
                e("v_sub_f32", al, 0, dl)
            e("v_exp_f32", al, al)                                 # 2^-delta (exp2 domain)
            e("v_mov_b32", one, 1.0)
            e("v_cmp_gt_f32", S_PAIR, THR[s], 0)                   # the row already had a reference
            e("v_cndmask_b32", al, one, al, S_PAIR)                # alpha
            e("v_mov_b32", TMP[5], S_THRV)
            e("v_cndmask_b32", THR[s], THR[s], TMP[5], VCC)        # raised rows have a reference from now on
            e("v_add_f32", MREF[s], MREF[s], dl)
            for k in range(16):
                e("v_sub_f32", CI(s)[k], 0, MREF[s])
            for h in range(2):
                t = T(bank_n, s, h)
                for k in range(16):
                    e("v_sub_f32", t[k], t[k], dl)
            e("v_mul_f32", LS(s, 0), LS(s, 0), al)
            e("v_mul_f32", LS(s, 1), LS(s, 1), al)
            # O *= alpha unless alpha == 1 everywhere (always so at the first tile)
            skip = self.p.uniq("RESC_SKIP")
            e("v_cmp_neq_f32", VCC, al, one)
            e("s_cbranch_vccz", skip)
            for db in range(4):
                o = OA(s, db)
                for k0 in range(0, 16, 8):
                    for k in range(k0, k0 + 8):
                        e("v_accvgpr_read_b32", TMP[8 + k - k0], o[k])
                    for k in range(k0, k0 + 8):
                        e("v_mul_f32", TMP[8 + k - k0], TMP[8 + k - k0], al)
                    for k in range(k0, k0 + 8):
                        e("v_accvgpr_write_b32", o[k], TMP[8 + k - k0])
            self.p.label(skip)
        self.nop(MFMA_SRCC_SAFE)
        # return to the call site
        sites = [st for st in self.ret_sites if st[3] == bank_n and st[4] == x]
        for site, ret, stub, _, _ in sites:
            e("s_cmp_eq_u32", S_RET, site)
            e("s_cbranch_scc1", ret)
        e("s_branch", sites[0][1] if sites else "EPILOGUE")

    # ---- prologue ---------------------------------------------------------------------------------------------------------------------
    def load_params(self):
        e = self.e
        # lane ids
        e("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
        e("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])               # lane
        e("v_and_b32", V_LQ, 31, TMP[0])
        e("v_lshrrev_b32", TMP[1], 5, TMP[0])                     # hi
        e("v_lshlrev_b32", V_HI4, 2, TMP[1])
        # the parameter block: every lane reads the same 16 dwords x 4
        e("v_mov_b32", TMP[2], self.param_sgpr)
        for k in range(len(PARAMS) // 4 + (1 if len(PARAMS) % 4 else 0)):
            e("ds_read_b128", V(4 * k, 4), TMP[2], offset=16 * k)
        e("s_waitcnt", lgkmcnt=0)

    def P(self, name):
        """VGPR that holds parameter `name` right after load_params (all lanes equal)"""
        return V(PIDX[name])

    def rfl(self, dst, name):
        self.e("v_readfirstlane_b32", dst, self.P(name))

    def prologue(self):
        e = self.e
        P = self.p
        if self.timing:
            for a in S_TACC:
                e("s_mov_b32", a, 0)
            e("s_memtime", S_TNOW)
            e("s_waitcnt", lgkmcnt=0)
            e("s_mov_b32", S_TLAST, S_TNOW[0])
        self.load_params()
        lane, hi = TMP[0], TMP[1]
        # ---- scalars ----
        for dst, name in ((S_NT, "n_tiles"), (S_N0, "n0"), (S_N1, "n1"), (S_NST, "n_steady"), (S_TM0, "tm0"), (S_TM1, "tm1"),
                          (S_TMM, "tmm"), (S_LENK, "len_k"), (S_SCALE, "scale_log2"), (S_THRV, "thr"), (S_ROWS[0], "rows0"), (S_ROWS[1], "rows1"),
                          (S_QST, "q_stride"), (S_OST, "o_stride"), (S_KSTRIDE, "k_stride"), (S_VSTRIDE, "v_stride"),
                          (S_TMP[7], "wave")):
            self.rfl(dst, name)
        e("s_lshl_b32", S_W1024, S_TMP[7], 10)
        e("s_lshl_b32", S_KTILE, S_KSTRIDE, 6)
        e("s_lshl_b32", S_VTILE, S_VSTRIDE, 6)
        # ---- descriptors ----
        if self.paged:
            for dst, name in ((S_KBASE[0], "k_lo"), (S_KBASE[1], "k_hi"), (S_VBASE[0], "v_lo"), (S_VBASE[1], "v_hi"), (S_BT[0], "bt_lo"),
                              (S_BT[1], "bt_hi"), (S_PSHIFT, "page_shift"), (S_KPAGE, "k_page_bytes"), (S_VPAGE, "v_page_bytes")):
                self.rfl(dst, name)
            e("s_mov_b32", S_KVT, 0)
            e("s_mov_b32", S_VVT, 0)
        else:
            for desc, rec, lo, hi_, nb in ((S_KDESC, S_KREC, "k_lo", "k_hi", "k_bytes"), (S_VDESC, S_VREC, "v_lo", "v_hi", "v_bytes")):
                self.rfl(desc[0], lo)
                self.rfl(desc[1], hi_)
                self.rfl(rec, nb)
                e("s_max_i32", desc[2], rec, 0)
        e("s_mov_b32", S_KDESC[3], 0x00020000)
        e("s_mov_b32", S_VDESC[3], 0x00020000)
        e("s_mov_b32", S_ODESC[3], 0x00020000)
        # ---- per-lane constants ----
        # K reads: row lq, 16-byte chunk (2j + hi) ^ (lq & 15)
        e("v_and_b32", TMP[3], 15, V_LQ)
        e("v_lshlrev_b32", TMP[4], 8, V_LQ)                       # lq * 256
        for j in range(8):
            e("v_xor_b32", TMP[5], 2 * j, TMP[3])                 # (2j) ^ (lq & 15) ... + hi below: hi only flips bit 0 and 2j is even
            e("v_xor_b32", TMP[5], TMP[5], hi)
            e("v_lshl_add_u32", KAD[j], TMP[5], 4, TMP[4])
        # V^T reads: vrow = 4hi + ((lane & 15) >> 2); dcol = 32db + 16((lane >> 4) & 1) + 4(lane & 3)
        e("v_and_b32", TMP[3], 15, lane)
        e("v_lshrrev_b32", TMP[3], 2, TMP[3])                     # jrow
        e("v_add_u32", TMP[4], V_HI4, TMP[3])                     # vrow
        e("v_lshlrev_b32", TMP[5], 2, TMP[3])                     # (vrow & 3) << 2 = jrow << 2
        e("v_lshrrev_b32", TMP[6], 4, lane)
        e("v_and_b32", TMP[6], 1, TMP[6])                         # (lane >> 4) & 1
        e("v_and_b32", TMP[7], 3, lane)                           # cc
        for db in range(4):
            # chunk = (dcol >> 3) = 4db + 2 g + (cc >> 1);  (dcol & 7) * 2 = (cc & 1) * 8
            e("v_lshrrev_b32", TMP[8], 1, TMP[7])
            e("v_lshl_add_u32", TMP[8], TMP[6], 1, TMP[8])
            e("v_add_u32", TMP[8], 4 * db, TMP[8])
            e("v_xor_b32", TMP[8], TMP[8], TMP[5])
            e("v_lshlrev_b32", TMP[8], 4, TMP[8])
            e("v_and_b32", TMP[9], 1, TMP[7])
            e("v_lshl_add_u32", TMP[8], TMP[9], 3, TMP[8])
            e("v_lshl_add_u32", VAD[db], TMP[4], 8, TMP[8])
            e("v_add_u32", VAD[db], 3 * SLOT + 2 * SLOT, VAD[db])   # V(0) lives in V slot (0 + 2) % 3
        # the K read addresses start at slot 1 (K(1) is fetched in the prologue with an explicit -SLOT: see below)
        # LDS-DMA: lane -> row 4w + lane/16 of a 16-row group, 16-byte slot lane%16; source chunk = slot ^ swizzle(row)
        e("v_lshrrev_b32", TMP[3], 4, lane)                       # lane / 16
        e("v_lshl_add_u32", TMP[3], S_TMP[7], 2, TMP[3])          # row16 = 4w + lane/16   (wave in S_TMP[7])
        e("v_and_b32", TMP[4], 15, lane)                          # slot
        e("v_xor_b32", TMP[5], TMP[4], TMP[3])                    # K: slot ^ (row & 15)   (row16 < 16)
        e("v_and_b32", TMP[6], 3, TMP[3])
        e("v_lshlrev_b32", TMP[6], 2, TMP[6])
        e("v_xor_b32", TMP[6], TMP[4], TMP[6])                    # V: slot ^ ((row & 3) << 2)
        e("v_lshlrev_b32", TMP[5], 4, TMP[5])
        e("v_lshlrev_b32", TMP[6], 4, TMP[6])
        e("v_mov_b32", TMP[7], S_KSTRIDE)
        e("v_mul_lo_u32", TMP[8], TMP[3], TMP[7])                 # row16 * k_stride
        e("v_add_u32", VOFK[0], TMP[8], TMP[5])
        e("v_mov_b32", TMP[7], S_VSTRIDE)
        e("v_mul_lo_u32", TMP[8], TMP[3], TMP[7])
        e("v_add_u32", VOFV[0], TMP[8], TMP[6])
        e("s_lshl_b32", S_TMP[0], S_KSTRIDE, 4)                  # 16 rows
        e("s_lshl_b32", S_TMP[1], S_VSTRIDE, 4)
        for u in range(1, 4):
            if self.paged:                                        # every piece has its own descriptor base: same offset
                e("v_mov_b32", VOFK[u], VOFK[0])
                e("v_mov_b32", VOFV[u], VOFV[0])
            else:
                e("v_add_u32", VOFK[u], S_TMP[0], VOFK[u - 1])
                e("v_add_u32", VOFV[u], S_TMP[1], VOFV[u - 1])
        # masks: last visible key of this lane's row, per slot
        e("v_mov_b32", V_NEGINF, NEG_INF)
        self.rfl(S_TMP[2], "lim_step")
        self.rfl(S_TMP[3], "lim0")
        self.rfl(S_TMP[4], "lim1")
        e("s_sub_u32", S_TMP[5], S_LENK, 1)
        e("v_mul_lo_u32", TMP[3], V_LQ, V(PIDX["lim_step"]))
        for s in range(2):
            e("v_add_u32", LIM[s], S_TMP[3 + s], TMP[3])
            e("v_min_i32", LIM[s], S_TMP[5], LIM[s])
        # ---- Q loads (in flight while the rings are set up) ----
        for s in range(2):
            skip = f"PRO_NOQ_{s}"
            e("s_cmp_eq_u32", S_ROWS[s], 0)
            e("s_cbranch_scc1", skip)
            e("s_sub_u32", S_TMP[0], S_ROWS[s], 1)
            e("v_min_u32", TMP[10], S_TMP[0], V_LQ)               # clamp to the slot's last valid row
            e("v_mov_b32", TMP[11], S_QST)
            e("v_mul_lo_u32", TMP[10], TMP[10], TMP[11])
            e("v_lshl_add_u32", TMP[10 + 2 + s], hi, 4, TMP[10])  # + hi * 16 bytes
            P.label(skip)
        # the loads themselves come after every use of the parameter VGPRs v0..v63 (fast: they land in the S bank v64..v127 and
        # are pre-multiplied below; exact: straight into the accumulator file)
        for s in range(2):
            self.rfl(S_PAIR[0], f"q{s}_lo")
            self.rfl(S_PAIR[1], f"q{s}_hi")
            skip = f"PRO_NOQL_{s}"
            e("s_cmp_eq_u32", S_ROWS[s], 0)
            e("s_cbranch_scc1", skip)
            for j in range(8):
                e("global_load_dwordx4", QA(s, j) if self.exact else V(64 + 32 * s + 4 * j, 4), TMP[12 + s], S_PAIR, offset=32 * j)
            P.label(skip)
        # ---- zero the V ring (rows the DMA never writes must not hold NaN patterns: P = 0 times NaN) ----
        e("v_mov_b32", TMP[16], 0)
        e("v_mov_b32", TMP[17], 0)
        e("v_mov_b32", TMP[18], 0)
        e("v_mov_b32", TMP[19], 0)
        e("v_lshlrev_b32", TMP[3], 4, lane)
        e("v_add_u32", TMP[3], S_W1024, TMP[3])
        e("v_add_u32", TMP[3], 3 * SLOT, TMP[3])
        for k in range(12):
            e("ds_write_b128", TMP[3], V(TMP[16].idx, 4), offset=k * 4096)
        e("s_waitcnt", lgkmcnt=0)
        e("s_barrier")
        # ---- ring state and the first six tiles: K0 K1 V0 K2 V1 K3 ----
        e("s_mov_b32", S_T, 0)
        # K tile i lives in K slot i % 3, V tile i in V slot (i + 2) % 3.  K(3) takes K(0)'s slot: it is requested below, once
        # every wavefront has fetched its K(0) fragments.
        for kind, slot in (("k", 0), ("k", 1), ("v", 2), ("k", 2), ("v", 0)):
            self.request_tile(kind, slot)
        # ---- softmax state, O = 0 ----
        for s in range(2):
            e("v_mov_b32", LS(s, 0), 0)
            e("v_mov_b32", LS(s, 1), 0)
            e("v_mov_b32", MREF[s], 0)
            e("v_mov_b32", THR[s], NEG_INF)
            e("v_mov_b32", MX[s], NEG_INF)
            for k in range(16):
                e("v_mov_b32", CI(s)[k], 0)
        for k in range(128):
            e("v_accvgpr_write_b32", A(k), 0)
        e("s_cmp_eq_u32", S_NT, 0)
        e("s_cbranch_scc1", "DRAIN_EPILOGUE")
        # ---- Q arrives: pre-multiply by scale.log2(e), round once, park in the accumulator file ----
        e("s_waitcnt", vmcnt=20)
        e("v_mov_b32", TMP[3], S_SCALE)
        for s in range(0 if self.exact else 2):
            skip = f"PRO_NOQS_{s}"
            e("s_cmp_eq_u32", S_ROWS[s], 0)
            e("s_cbranch_scc1", skip)
            for j in range(8):
                for k in range(4):
                    x = V(64 + 32 * s + 4 * j + k)
                    lo, hi2 = TMP[4 + 2 * (k & 1)], TMP[5 + 2 * (k & 1)]
                    if self.dtype == "bf16":
                        e("v_lshlrev_b32", lo, 16, x)
                        e("v_and_b32", hi2, 0xFFFF0000, x)
                    else:
                        e("v_cvt_f32_f16", lo, x)
                        e("v_lshrrev_b32", hi2, 16, x)
                        e("v_cvt_f32_f16", hi2, hi2)
                    e("v_mul_f32", lo, lo, TMP[3])
                    e("v_mul_f32", hi2, hi2, TMP[3])
                    e(self.cvt, lo, lo, hi2)
                    e("v_accvgpr_write_b32", QA(s, j)[k], lo)
            P.label(skip)
        # ---- K(0): fragments, S(0) ----
        e("s_waitcnt", vmcnt=16)
        e("s_barrier")
        for i in self.k_reads():                                   # KAD points at slot 0 here
            P.ins.append(i)
        e("s_waitcnt", lgkmcnt=0)
        self.nop(2)
        e("s_cmp_eq_u32", S_N1, 0)
        e("s_cbranch_scc1", "PRO_K1")                              # no row of this wavefront sees a key
        e("s_cmp_eq_u32", S_N0, 0)
        e("s_cbranch_scc1", "PRO_QK1")
        P.extend(self.qk_mfmas(0, [0, 1]))
        e("s_branch", "PRO_K1")
        P.label("PRO_QK1")
        P.extend(self.qk_mfmas(0, [1]))
        # ---- K(1) fragments (K(1) sits in slot 1); K(3) may now replace K(0) ----
        P.label("PRO_K1")
        e("s_waitcnt", vmcnt=12)
        e("s_barrier")
        for i in self.k_reads():
            i.mods["offset"] = i.mods.get("offset", 0) + SLOT
            P.ins.append(i)
        for j in range(8):
            e("v_add_u32", KAD[j], 2 * SLOT, KAD[j])               # next K read: K(2) in slot 2
        self.request_tile("k", 0)
        # state at the entry of iteration 0: read slot r = 2 (K(2) and V(0) live in slot 2), DMA slot d = 1
        e("s_mov_b32", S_RBASE, 2 * SLOT)
        e("s_mov_b32", S_DBASE, 1 * SLOT)
        e("s_add_u32", S_KDMA, S_DBASE, S_W1024)
        e("s_mov_b32", S_DELTA, 0)
        # ---- mask + row max of tile 0 (always through the masking code), then the rows' first reference ----
        e("s_mov_b32", S_T1, 0)                                    # kv1 = 0
        e("s_cmp_eq_u32", S_N1, 0)
        e("s_cbranch_scc1", "PRO_ENTRY")
        self.nop(MFMA_SAFE)
        e("s_cmp_eq_u32", S_N0, 0)
        e("s_cbranch_scc1", "PRO_M1")
        for x0, lab in ((2, None), (1, "PRO_M1")):
            if lab:
                P.label(lab)
            for s in {2: [0, 1], 1: [1]}[x0]:
                for it in self.mask_items(0, s, -1) + self.max_items(0, s, -1):
                    P.extend(it.ins)
            site = len(self.ret_sites)
            self.ret_sites.append((site, f"RET_{site}", f"STUB_{site}", 0, x0))
            e("s_branch", f"STUB_{site}")
            P.label(f"RET_{site}")
            e("s_branch", "PRO_ENTRY")
        P.label("PRO_ENTRY")
        e("s_waitcnt", vmcnt=8, lgkmcnt=0)
        e("s_barrier")
        P.extend(self.stamp(5))
        e("s_branch", "DISP_0")

    def request_tile(self, kind, slot):
        """prologue: the four pieces of the next K (or V) tile -> ring slot `slot`"""
        e = self.e
        e("s_add_u32", S_KDMA, S_W1024, slot * SLOT)
        if self.paged:
            self.p.extend(self.paged_tile(S_VVT if kind == "v" else S_KVT, kind == "v"))
        else:
            desc, voffs, tile, rec, extra = ((S_VDESC, VOFV, S_VTILE, S_VREC, 3 * SLOT) if kind == "v"
                                             else (S_KDESC, VOFK, S_KTILE, S_KREC, 0))
            for it in self.dma_tile(desc, voffs, extra) + self.desc_advance(desc, tile, rec):
                self.p.extend(it.ins)

    # ---- epilogue -----------------------------------------------------------------------------------------------------------------------
    def epilogue(self):
        e = self.e
        P = self.p
        P.label("DRAIN_EPILOGUE")          # n_tiles = 0: nothing was computed, requests still in flight must land before the wave ends
        e("s_waitcnt", vmcnt=0)
        P.label("EPILOGUE")
        P.extend(self.stamp(4))
        self.nop(MFMA_SAFE)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        # parameters again (their VGPRs were reused)
        e("v_mov_b32", TMP[2], self.param_sgpr)
        for k in range(4):
            e("ds_read_b128", V(4 * k, 4), TMP[2], offset=16 * k)
        e("s_waitcnt", lgkmcnt=0)
        e("v_lshrrev_b32", TMP[1], 2, V_HI4)                     # hi
        for s in range(2):
            done = f"EPI_DONE_{s}"
            e("s_cmp_eq_u32", S_ROWS[s], 0)
            e("s_cbranch_scc1", done)
            self.rfl(S_ODESC[0], f"o{s}_lo")
            self.rfl(S_ODESC[1], f"o{s}_hi")
            e("s_sub_u32", S_TMP[0], S_ROWS[s], 1)
            e("s_mul_i32", S_TMP[0], S_TMP[0], S_OST)
            e("s_add_u32", S_ODESC[2], S_TMP[0], 256)            # rows past the slot's last valid row are out of range: dropped
            ltot, inv, t0, t1 = TMP[3], TMP[4], TMP[5], TMP[6]
            e("v_add_f32", t0, LS(s, 0), LS(s, 1))
            e("v_mov_b32", t1, t0)
            self.nop(2)
            e("v_permlane32_swap_b32", t0, t1)
            e("v_add_f32", ltot, t0, t1)
            e("v_rcp_f32", inv, ltot)
            e("v_cmp_gt_f32", VCC, ltot, 0)
            e("v_mov_b32", t0, 0)
            e("v_cndmask_b32", inv, t0, inv, VCC)                # rows that saw no key: 0
            e("v_mov_b32", t0, S_OST)
            e("v_mul_lo_u32", t0, V_LQ, t0)
            e("v_lshl_add_u32", TMP[7], TMP[1], 4, t0)           # row * stride + hi * 16
            for db in range(4):
                o = OA(s, db)
                for pr in range(2):                               # pairs of 8-column groups (r4 = 2pr, 2pr + 1)
                    w = [TMP[8 + 4 * ((2 * db + pr) % 4) + k] for k in range(4)]
                    f = [TMP[0], TMP[2]]
                    for half in range(2):
                        r4 = 2 * pr + half
                        for k2 in range(2):
                            e("v_accvgpr_read_b32", f[0], o[4 * r4 + 2 * k2])
                            e("v_accvgpr_read_b32", f[1], o[4 * r4 + 2 * k2 + 1])
                            e("v_mul_f32", f[0], f[0], inv)
                            e("v_mul_f32", f[1], f[1], inv)
                            e(self.cvt, w[2 * half + k2], f[0], f[1])
                    self.nop(2)
                    e("v_permlane32_swap_b32", w[0], w[2])
                    e("v_permlane32_swap_b32", w[1], w[3])
                    e("buffer_store_dwordx4", V(w[0].idx, 4), TMP[7], S_ODESC, 0, offen=True, offset=(32 * db + 16 * pr) * 2)
            # log-sum-exp (natural log), rows without keys: +inf
            nolse = f"EPI_NOLSE_{s}"
            self.rfl(S_PAIR[0], f"lse{s}_lo")
            self.rfl(S_PAIR[1], f"lse{s}_hi")
            e("s_or_b32", S_TMP[0], S_PAIR[0], S_PAIR[1])
            e("s_cmp_eq_u32", S_TMP[0], 0)
            e("s_cbranch_scc1", nolse)
            e("s_mov_b32", S_ODESC[0], S_PAIR[0])
            e("s_mov_b32", S_ODESC[1], S_PAIR[1])
            e("s_lshl_b32", S_ODESC[2], S_ROWS[s], 2)
            e("v_log_f32", t0, ltot)
            if self.exact:
                e("v_mul_f32", t1, S_SCALE, MREF[s])
                e("v_add_f32", t0, t0, t1)
            else:
                e("v_add_f32", t0, t0, MREF[s])
            e("v_mul_f32", t0, 0.6931471805599453, t0)
            e("v_mov_b32", t1, float("inf"))
            e("v_cndmask_b32", t0, t1, t0, VCC)
            e("v_lshlrev_b32", t1, 2, V_LQ)
            e("v_cmp_eq_u32", S_PAIR, 0, V_HI4)                  # one lane of each pair stores
            e("v_mov_b32", TMP[0], 0x7FFFFFF0)
            e("v_cndmask_b32", t1, TMP[0], t1, S_PAIR)
            e("buffer_store_dword", t0, t1, S_ODESC, 0, offen=True)
            P.label(nolse)
            P.label(done)
        e("s_waitcnt", vmcnt=0)
        if self.timing:
            P.extend(self.stamp(6))
            e("v_mov_b32", TMP[2], self.param_sgpr)
            e("ds_read_b128", V(0, 4), TMP[2], offset=16 * (PIDX["dbg_lo"] // 4))
            e("ds_read_b128", V(4, 4), TMP[2], offset=16 * (PIDX["dbg_lo"] // 4) + 16)
            e("s_waitcnt", lgkmcnt=0)
            e("v_readfirstlane_b32", S_ODESC[0], V(PIDX["dbg_lo"] % 4))
            e("v_readfirstlane_b32", S_ODESC[1], V(PIDX["dbg_lo"] % 4 + 1))
            e("s_mov_b32", S_ODESC[2], 32)
            e("s_or_b32", S_TT, S_ODESC[0], S_ODESC[1])
            e("s_cmp_eq_u32", S_TT, 0)
            e("s_cbranch_scc1", "TIMING_DONE")
            e("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
            e("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])
            e("v_lshlrev_b32", TMP[0], 16, TMP[0])                # only lane 0 is inside the 32-byte window
            for k in range(8):
                e("v_mov_b32", TMP[8 + k], S_TACC[k])
            e("buffer_store_dwordx4", V(TMP[8].idx, 4), TMP[0], S_ODESC, 0, offen=True)
            e("buffer_store_dwordx4", V(TMP[12].idx, 4), TMP[0], S_ODESC, 0, offen=True, offset=16)
            e("s_waitcnt", vmcnt=0)
            P.label("TIMING_DONE")

    # ---- the whole program ------------------------------------------------------------------------------------------------------------------
    def build(self):
        self.prologue()
        for p in (0, 1):
            self.dispatcher(p)
        for p in (0, 1):
            self.p.emit("p2align", 6)
            self.iteration(p, 2, 2, False)
        for p in (0, 1):
            for c, x, mk in ((2, 2, True), (2, 1, False), (2, 1, True), (2, 0, False), (1, 1, False), (1, 1, True), (1, 0, False)):
                self.iteration(p, c, x, mk)
        self.idle_iteration()
        # call stubs, rescale blocks
        for site, ret, stub, bank_n, x in self.ret_sites:
            self.p.label(stub)
            self.e("s_mov_b32", S_RET, site)
            self.e("s_branch", f"RESC_{bank_n}_{x}")
        for bank_n in (0, 1):
            for x in (2, 1):
                self.rescale(bank_n, x)
        self.epilogue()
        return self.p


MFMA_SAFE = 20
MFMA_SRCC_SAFE = 4


def build(dtype="bf16", paged=False, param_sgpr=S(4), exact=False, timing=False):
    b = Builder(dtype, paged, param_sgpr, exact, timing)
    prog = b.build()
    return prog, b


def clobbers():
    """registers the asm statement owns (csrc/prefill_asm.hip lists them as clobbered)"""
    return [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(SGPR_FIRST, SGPR_LAST + 1)] + ["vcc", "scc", "memory"]
