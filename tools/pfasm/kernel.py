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
import os

from .isa import Program, Ins, Reg, V, A, S, VCC, M0, f32_bits

NEG_INF = float("-inf")

# ---- parameter entry: 64 dwords per (query block, wavefront).  Contiguous K/V: an entry of the plan table in global memory
# (csrc/prefill_asm.hip pfa_plan_kernel), fetched with three s_load_dwordx16 straight into s48..: descriptors and scalars sit in
# the registers the code uses.  Paged K/V: the same layout in LDS, written by the launching kernel's C++ preamble. --------------
PARAMS = [
    "k_lo", "k_hi", "k_bytes", "k_flags",           # K descriptor (contiguous: row 0 of this sequence / kv head, bytes to the end of its last row)
    "v_lo", "v_hi", "v_bytes", "v_flags",           #   paged: lo / hi = cache base + head offset, bytes / flags rebuilt per piece
    "o0_lo", "o0_hi", "o0_bytes", "o0_flags",       # O descriptor of slot 0: its row 0 for this head; bytes = (rows - 1) * stride + 256
    "o1_lo", "o1_hi", "o1_bytes", "o1_flags",
    "q0_lo", "q0_hi", "q1_lo", "q1_hi",             # row 0 of the slot's q block (this head)
    "lse0_lo", "lse0_hi", "lse1_lo", "lse1_hi",     # row 0 of the slot in the LSE array (4-byte row stride), 0 = none
    "q_stride", "o_stride", "rows0", "rows1",       # row strides (bytes); valid rows of the slot (0 .. 32)
    "k_tile", "v_tile", "k_stride", "v_stride",     # 64 * stride; K/V row strides, bytes
    "len_k", "n_tiles", "n0", "n1",                 # keys; K/V tiles of the workgroup; tiles slot 0 / slot 1 take part in
    "n_steady", "tm0", "tm1", "tmm",                # leading iterations of the plain variant; first tile that needs a mask (slot 0, 1, min)
    "lim0", "lim1", "lim_step", "scale_log2",       # last visible key of the slot's row 0; +lim_step per row (1 causal, 0 not); f32 bits
    "thr", "flags", "mscale", "pad0",               # deferred-raise threshold in the domain of S' (8, or 8 / scale_log2 when exact); bit 0 exact, bit 1 valid; LSE: m * mscale
    "bt_lo", "bt_hi", "page_shift", "k_page",       # paged: this sequence's block table, log2(page size), page strides (bytes)
    "v_page", "wave", "dbg_lo", "dbg_hi",           # wave: 0 .. 3 (LDS entries only); timing builds: 32 bytes per wavefront
]
PIDX = {n: i for i, n in enumerate(PARAMS)}
PARAM_DWORDS = 64
FLAG_EXACT, FLAG_VALID = 1, 2
LDS_RING = 96 * 1024            # K slots at 0 / 16K / 32K, V slots at 48K / 64K / 80K
LDS_STAGE = LDS_RING            # 4 x 16 KiB: O^T -> O transposition in the epilogue; the paged kernels' parameter entries
LDS_PARAMS = LDS_RING           #   (4 x 256 bytes, read before the first barrier) share the first KiB
LDS_TOTAL = 160 * 1024
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
V_EPW = V(231)          # epilogue: LDS write address of this lane's row (staging region, swizzle folded in)
TMP = [V(232 + i) for i in range(24)]
EPR = [TMP[20], TMP[21], TMP[22], TMP[23]]      # epilogue: LDS read addresses (4 row groups); timing builds keep their timers in TMP[12..19]
TACC = [TMP[14 + i] for i in range(6)]        # phase 1, phase 2, waits + barrier, loop control, prologue, epilogue

# scalar registers: s16 .. s99 belong to the asm statement, except s32 (the compiler's stack pointer; s100 / s101 are reserved too)
S_T, S_T1, S_RBASE, S_DBASE, S_DELTA, S_KDMA, S_W1024, S_RET = (S(16 + i) for i in range(8))
S_TMP = [S(24 + i) for i in range(8)]
S_TLAST = S(33)
S_PAIR = S(34, 2)
S_KSOFF, S_VSOFF = S(36), S(37)                 # contiguous: byte offset of the next K / V tile to request
S_BLK, S_NENT = S(38), S(39)
S_SAVE = [S(40 + i) for i in range(8)]          # slot 1's epilogue parameters while the next block's entry is loaded (descriptor: 4-aligned)
S_G, S_G2, S_TAB = S(96), S(97), S(98, 2)       # persistent: s96..s99 (the paged fields of the entry) are not loaded
S_TNOW, S_TT = S(30, 2), S(29)
S_HASNEXT = S_RET
# paged kernels (not persistent): the registers of the block loop hold the paging state; the entry's paged fields sit in s96..s99
S_KVT, S_VVT, S_VPAGE = S(36), S(37), S(33)
S_KBASE, S_VBASE = S(40, 2), S(42, 2)
S_BT, S_PSHIFT, S_KPAGE = S(96, 2), S(98), S(99)
WIN = 48                        # the entry's first 52 dwords live in s48 .. s99


def W(name, n=1):
    return S(WIN + PIDX[name], n)


S_KDESC, S_VDESC, S_ODESC = W("k_lo", 4), W("v_lo", 4), [W("o0_lo", 4), W("o1_lo", 4)]
S_LENK, S_NT, S_N0, S_N1, S_NST, S_TM0, S_TM1, S_TMM = (W(n) for n in ("len_k", "n_tiles", "n0", "n1", "n_steady", "tm0", "tm1", "tmm"))
S_KTILE, S_VTILE, S_KSTRIDE, S_VSTRIDE = W("k_tile"), W("v_tile"), W("k_stride"), W("v_stride")
S_SCALE, S_THRV, S_QST, S_OST = W("scale_log2"), W("thr"), W("q_stride"), W("o_stride")
S_ROWS = [W("rows0"), W("rows1")]
SGPR_FIRST, SGPR_LAST = 16, 99


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
    def __init__(self, dtype="bf16", paged=False, inputs=None, timing=False):
        """paged = False: the persistent kernel (one workgroup per CU walks the plan table); inputs = dict(tab, first, n, g, wave[, dbg])
        of scalar operands.  paged = True: one workgroup per query block, parameters in LDS; inputs = dict(param).
        Both carry the two arithmetic variants (label suffix "" = fast, "X" = exact), chosen per query block by the entry's flags:
          fast  -- Q pre-multiplied by scale.log2(e) and rounded once (S' in the exp2 domain, 3 VALU per score);
          exact -- Q as it is, S' = s - m in the raw domain and one v_mul_f32 by scale.log2(e) in front of every v_exp_f32: the
                   reference's arithmetic to the last rounding, for rows whose softmax mass sits on a few keys."""
        assert dtype in ("bf16", "f16")
        assert not (timing and paged)
        self.dtype, self.paged, self.timing = dtype, paged, timing
        self.persistent = not paged
        self.split_req = self.persistent and os.environ.get("PFA_SPLIT_REQ", "0") != "0"   # (experiment, measured level to 2 % worse: profiles/r04_prefill_split_requests_ab.txt) the first five K/V tiles of a block: two with the epilogue, three under the prologue's products
        self.inp = inputs or (dict(param=S(4)) if paged else dict(tab=S(4, 2), first=S(6), n=S(7), g=S(8), wave=S(9), dbg=S(10, 2), g2=S(12)))
        self.exact, self.sfx = False, ""
        self.mfma = "v_mfma_f32_32x32x16_" + dtype
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.p = Program()
        self.sched_log = {}
        self.ret_sites = []       # RESC call sites: (id, return label, stub label, bank, slots, suffix)

    # ------------------------------------------------------------------------------------------------------------------------
    def e(self, op, *ops, **mods):
        return self.p.emit(op, *ops, **mods)

    def stamp(self, k, mode="phases"):
        """timing builds ("phases": the loop's phases and the per-block sums; "block": the regions between two blocks' loops): cycles
        since the previous stamp -> timer k; k = None only restarts the clock (s_memtime: the wait drains the LDS queue too)"""
        if self.timing != mode:
            return []
        out = [Ins("s_memtime", S_TNOW), Ins("s_waitcnt", lgkmcnt=0), Ins("s_sub_u32", S_TT, S_TNOW[0], S_TLAST), Ins("s_mov_b32", S_TLAST, S_TNOW[0])]
        return out + ([Ins("v_add_u32", TACC[k], S_TT, TACC[k])] if k is not None else [])

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
            x2 = lambda x: Ins("v_exp_f32", x, x)
            if self.exact:     # the multiplies run one pair ahead of their v_exp_f32
                m = lambda x: Ins("v_mul_f32", x, S_SCALE, x)
                seq += [m(e[0]), m(e[1]), m(e[2]), m(e[3]), x2(e[0]), x2(e[1]), x2(e[2]), x2(e[3]), m(e[4]), m(e[5]), ad(0), ad(1), cv(0),
                        x2(e[4]), x2(e[5]), m(e[6]), m(e[7]), ad(2), ad(3), cv(1), x2(e[6]), x2(e[7]), ad(4), ad(5), cv(2), ad(6), ad(7), cv(3)]
            else:
                seq += [x2(e[0]), x2(e[1]), x2(e[2]), x2(e[3]), ad(0), ad(1), cv(0), x2(e[4]), x2(e[5]), ad(2), ad(3), cv(1),
                        x2(e[6]), x2(e[7]), ad(4), ad(5), cv(2), ad(6), ad(7), cv(3)]
        return [Item(i, release, deadline) for i in seq]

    def v_read(self, o):
        h, kk, db = o >> 3, (o >> 2) & 1, o & 3
        off = (32 * h + 16 * kk) * 256
        dst = VV(o % 8)
        return [Ins("ds_read_b64_tr_b16", dst[0:2], VAD[db], offset=off), Ins("ds_read_b64_tr_b16", dst[2:4], VAD[db], offset=off + 2048)]

    def k_reads(self, extra=0):
        return [Ins("ds_read_b128", KA(h, j), KAD[j], offset=h * 8192 + extra) for h in range(2) for j in range(8)]

    def max_items(self, bank_n, s, release):
        out = []
        for h, acc in ((0, MX[s]), (1, MXB[s])):
            t = T(bank_n, s, h)
            out.append(Ins("v_max3_f32", acc, t[0], t[1], t[2]))
            for k in range(6):
                out.append(Ins("v_max3_f32", acc, acc, t[3 + 2 * k], t[4 + 2 * k]))
            out.append(Ins("v_max_f32", acc, acc, t[15]))
        a, b = out[:8], out[8:]
        seq = [x for pair in zip(a, b) for x in pair] + [Ins("v_max_f32", MX[s], MX[s], MXB[s])]
        return [Item(i, release) for i in seq]

    def mask_items(self, bank_n, s, release):
        """keys beyond the lane's limit -> -inf: key(r, h) = kv1 + 32h + (r&3) + 8(r>>2) + 4hi > LIM  <=>  const > LIMREL (S_T1 = kv1)"""
        seq = [Ins("v_subrev_u32", LIMREL[s], S_T1, LIM[s]), Ins("v_sub_u32", LIMREL[s], LIMREL[s], V_HI4)]
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

    def dma_tile(self, is_v):
        """contiguous K/V: the four 1 KiB pieces of this wavefront (rows 16u + 4w .. +3 of the tile) -> ring slot S_KDMA; the tile's byte
        offset rides in soffset (it takes part in the descriptor's range check: rows past the sequence arrive as zeros)"""
        desc, voffs, soff, tile, extra = (S_VDESC, VOFV, S_VSOFF, S_VTILE, 3 * SLOT) if is_v else (S_KDESC, VOFK, S_KSOFF, S_KTILE, 0)
        items = []
        exp = os.environ.get("PFA_EXP", "")
        if exp in ("regload", "regload_write", "nodma") and getattr(self, "in_loop", False):
            # TIMING EXPERIMENTS ONLY (results are wrong): the pieces as plain loads into registers [+ a 16-byte LDS write each] / no request at all
            for u in range(4):
                dst = V(TMP[0].idx + 4 * ((u + 4 * is_v) % 3), 4)
                if exp != "nodma":
                    items.append(Item([Ins("buffer_load_dwordx4", dst, voffs[u], desc, soff, offen=True)]))
                if exp == "regload_write":
                    items.append(Item([Ins("ds_write_b128", VAD[0], V(TMP[12].idx, 4), offset=0)]))
            items.append(Item(Ins("s_add_u32", soff, soff, tile)))
            return items
        for u in range(4):
            items.append(Item([Ins("s_add_u32", M0, S_KDMA, extra + u * 4096), Ins("s_nop", 0),
                               Ins("buffer_load_dwordx4", voffs[u], desc, soff, offen=True, lds=True)]))
        items.append(Item(Ins("s_add_u32", soff, soff, tile)))
        return items

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

    def request_tile(self, is_v, slot=None):
        """the four pieces of the next K (or V) tile, as straight code; slot: ring slot index (prologue), None: S_KDMA is set"""
        if slot is not None:
            self.e("s_add_u32", S_KDMA, S_W1024, slot * SLOT)
        if self.paged:
            self.p.extend(self.paged_tile(S_VVT if is_v else S_KVT, is_v))
        else:
            for it in self.dma_tile(is_v):
                self.p.extend(it.ins)

    def ring_rotate(self):
        return [Ins("s_mov_b32", S_DBASE, S_RBASE), Ins("s_add_u32", S_RBASE, S_RBASE, S_DELTA), Ins("s_add_u32", S_KDMA, S_DBASE, S_W1024)]

    # ---- one iteration -----------------------------------------------------------------------------------------------------------
    def iteration(self, p, c, x, mk):
        """tile t in bank p: cur slots (c = 2: both, 1: slot 1, 0: none) exponentiate + P.V; next slots (x) get S(t+1)."""
        sfx = self.sfx
        name = f"IT{sfx}_{p}_{c}{x}{'m' if mk else '0'}"
        cur = {2: [0, 1], 1: [1], 0: []}[c]
        nxt = {2: [0, 1], 1: [1], 0: []}[x]
        bank_c, bank_n = p, p ^ 1
        P = self.p
        P.label(name)
        # ---------------- phase 1 ----------------
        mf1 = self.qk_mfmas(bank_n, nxt)
        n1 = len(mf1)
        if len(cur) == 2:      # interleave the slots' streams so that both finish early (P of half 0 is needed first by P.V)
            a, b = self.exp_items(bank_c, 0, 0), self.exp_items(bank_c, 1, 0)
            exp_h0 = [y for pair in zip(a, b) for y in pair]
            a, b = self.exp_items(bank_c, 0, 1), self.exp_items(bank_c, 1, 1)
            exp_h1 = [y for pair in zip(a, b) for y in pair]
        else:
            exp_h0 = [it for s in cur for it in self.exp_items(bank_c, s, 0)]
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
        kr = [Item(i, 1 if n2 else -1) for i in self.k_reads()] if nxt else []
        ring = [Item(Ins("s_mov_b32", S_TMP[0], SLOT), -1, 1), Item(Ins("s_cmp_eq_u32", S_RBASE, 2 * SLOT), -1, 1),
                Item(Ins("s_cselect_b32", S_DELTA, -2 * SLOT, S_TMP[0]), -1, 1)]
        adv = [Item(Ins("v_add_u32", KAD[j], S_DELTA, KAD[j])) for j in range(8)]
        vadv = [Item(Ins("v_add_u32", VAD[db], S_DELTA, VAD[db]), release=max(15 * per - 1, -1)) for db in range(4)]
        self.in_loop = True
        dma = [] if self.paged else self.dma_tile(False) + self.dma_tile(True)      # paged: a scalar block in front of phase 2
        self.in_loop = False
        qpre = []
        if self.persistent and x == 0 and c > 0:         # the wavefront's last active iteration of the block: the NEXT block's Q rows
            v, loads = self.qpre_code()
            qpre1 = [Item(i) for i in v + loads[:8]]          # phase 1 of a draining iteration has no MFMAs at all: half of the loads there
            qpre = [Item(i, 2) for i in loads[8:]]
        for it in dma:
            it.release = max(it.release, 2 if n2 else -1)
        # ---- how much of the second half's softmax rides in phase 1: the same filler density in both phases ----
        cnt = lambda items: sum(len(it.ins) for it in items)
        f1 = cnt(exp_h0) + cnt(vreads)
        f2 = cnt(v2) + cnt(mm) + cnt(ring) + cnt(kr) + cnt(adv) + cnt(vadv) + cnt(dma) + cnt(qpre)
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
        exp_knob = os.environ.get("PFA_EXP", "")
        if exp_knob.startswith("drop:"):      # TIMING EXPERIMENTS ONLY (wrong results): drop every filler with this mnemonic from the loop
            ops = exp_knob[5:].split(",")
            flt = lambda items: [it for it in items if not any(i.op in ops for i in it.ins)]
            exp_h0, vreads, e1_in_p1, e1_in_p2, v2, mm, kr, adv, vadv = (flt(x) for x in (exp_h0, vreads, e1_in_p1, e1_in_p2, v2, mm, kr, adv, vadv))
        body1 = schedule(mf1, [exp_h0, vreads, e1_in_p1] + ([qpre1] if qpre else []), cap=cap1, log=log1)
        body2 = schedule(mf2, [e1_in_p2, v2, mm, ring + kr + adv, dma + qpre, vadv], cap=cap2, log=log2)
        self.sched_log[name] = (log1, log2)
        blk = self.stamp(3)
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
            ret, stub = f"RET_{site}", f"STUB_{site}"
            if len(nxt) == 2:
                self.e("v_cmp_gt_f32", S_PAIR, MX[0], THR[0])
                self.e("v_cmp_gt_f32", VCC, MX[1], THR[1])
                self.e("s_or_b64", S_PAIR, S_PAIR, VCC)
            else:
                self.e("v_cmp_gt_f32", S_PAIR, MX[1], THR[1])
                self.e("s_or_b64", S_PAIR, S_PAIR, S_PAIR)
            self.e("s_cbranch_scc1", stub)
            P.label(ret)
            self.ret_sites.append((site, ret, stub, bank_n, x, sfx))
        P.extend(self.ring_rotate())
        self.e("s_waitcnt", vmcnt=8 + (16 if qpre else 0), lgkmcnt=0)
        self.e("s_barrier")
        P.extend(self.stamp(2))
        self.e("s_add_u32", S_T, S_T, 1)
        if (c, x, mk) == (2, 2, False):
            self.e("s_cmp_lt_u32", S_T, S_NST)
            self.e("s_cbranch_scc1", f"IT{sfx}_{p ^ 1}_220")
        if qpre:                                               # the block ends here for most wavefronts: the Q loads stay in flight across its epilogue
            self.e("s_cmp_ge_u32", S_T, S_NT)
            self.e("s_cbranch_scc1", "BLOCK_END_Q")
        self.e("s_branch", f"DISP{sfx}_{p ^ 1}")

    def idle_iteration(self):
        """a wavefront with no active slot: its share of the LDS-DMA, the barrier (the read addresses keep following the ring: the
        next block of a persistent workgroup starts from them)"""
        P = self.p
        P.label("IT_00")
        self.request_tile(False)
        self.request_tile(True)
        self.e("s_mov_b32", S_TMP[0], SLOT)
        self.e("s_cmp_eq_u32", S_RBASE, 2 * SLOT)
        self.e("s_cselect_b32", S_DELTA, -2 * SLOT, S_TMP[0])
        for r in KAD + VAD:
            self.e("v_add_u32", r, S_DELTA, r)
        P.extend(self.ring_rotate())
        self.e("s_waitcnt", vmcnt=8, lgkmcnt=0)
        self.e("s_barrier")
        self.e("s_add_u32", S_T, S_T, 1)
        self.e("s_cmp_lt_u32", S_T, S_NT)
        self.e("s_cbranch_scc1", "IT_00")
        self.e("s_branch", "BLOCK_END")

    def dispatcher(self, p):
        P, e, x = self.p, self.e, self.sfx
        P.label(f"DISP{x}_{p}")
        e("s_cmp_ge_u32", S_T, S_NT)
        e("s_cbranch_scc1", "BLOCK_END")
        e("s_cmp_lt_u32", S_T, S_NST)
        e("s_cbranch_scc1", f"IT{x}_{p}_220")
        e("s_add_u32", S_T1, S_T, 1)
        e("s_cmp_lt_u32", S_T, S_N0)
        e("s_cbranch_scc0", f"D{x}{p}_CNOT2")
        e("s_cmp_lt_u32", S_T1, S_N0)
        e("s_cbranch_scc0", f"D{x}{p}_C2XN2")
        e("s_cmp_ge_u32", S_T1, S_TMM)
        e("s_cbranch_scc1", f"IT{x}_{p}_22m")
        e("s_branch", f"IT{x}_{p}_220")
        P.label(f"D{x}{p}_C2XN2")
        e("s_cmp_lt_u32", S_T1, S_N1)
        e("s_cbranch_scc0", f"IT{x}_{p}_200")
        e("s_cmp_ge_u32", S_T1, S_TM1)
        e("s_cbranch_scc1", f"IT{x}_{p}_21m")
        e("s_branch", f"IT{x}_{p}_210")
        P.label(f"D{x}{p}_CNOT2")
        e("s_cmp_lt_u32", S_T, S_N1)
        e("s_cbranch_scc0", "IT_00")
        e("s_cmp_lt_u32", S_T1, S_N1)
        e("s_cbranch_scc0", f"IT{x}_{p}_100")
        e("s_cmp_ge_u32", S_T1, S_TM1)
        e("s_cbranch_scc1", f"IT{x}_{p}_11m")
        e("s_branch", f"IT{x}_{p}_110")

    # ---- the rare block: raise the rows' reference --------------------------------------------------------------------------------
    def rescale(self, bank_n, x):
        """S'(t+1) of the slots in x sits in bank_n relative to the OLD references.  Per row (both lanes of a pair agree):
        r = row max';  raise = r > THR (THR = -inf while the row has no reference yet, S_THRV afterwards);  delta = raise ? r : 0;
        m += delta;  C-init = -m;  S' -= delta;  alpha = had a reference ? 2^-delta : 1;  O *= alpha;  l *= alpha."""
        e = self.e
        slots = {2: [0, 1], 1: [1]}[x]
        self.p.label(f"RESC{self.sfx}_{bank_n}_{x}")
        self.nop(MFMA_SAFE)
        for s in slots:
            r, r2, dl, al, one = VV(0)[0], VV(0)[1], VV(0)[2], VV(0)[3], VV(1)[0]     # the V^T window is dead here (all P.V MFMAs issued)
            tmp8 = [VV(2 + k // 4)[k % 4] for k in range(8)]
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
            e("v_mov_b32", VV(1)[1], S_THRV)
            e("v_cndmask_b32", THR[s], THR[s], VV(1)[1], VCC)      # raised rows have a reference from now on
            e("v_add_f32", MREF[s], MREF[s], dl)
            for k in range(16):
                e("v_sub_f32", CI(s)[k], 0, MREF[s])
            for h in range(2):
                t = T(bank_n, s, h)
                for k in range(16):
                    e("v_sub_f32", t[k], t[k], dl)
            e("v_mul_f32", LS(s, 0), LS(s, 0), al)
            e("v_mul_f32", LS(s, 1), LS(s, 1), al)
            # O *= alpha unless alpha == 1 everywhere
            skip = self.p.uniq("RESC_SKIP")
            e("v_cmp_neq_f32", VCC, al, one)
            e("s_cbranch_vccz", skip)
            for db in range(4):
                o = OA(s, db)
                for k0 in range(0, 16, 8):
                    for k in range(8):
                        e("v_accvgpr_read_b32", tmp8[k], o[k0 + k])
                    for k in range(8):
                        e("v_mul_f32", tmp8[k], tmp8[k], al)
                    for k in range(8):
                        e("v_accvgpr_write_b32", o[k0 + k], tmp8[k])
            self.p.label(skip)
        self.nop(MFMA_SRCC_SAFE)
        sites = [st for st in self.ret_sites if st[3] == bank_n and st[4] == x and st[5] == self.sfx]
        for site, ret, stub, _, _, _ in sites:
            e("s_cmp_eq_u32", S_RET, site)
            e("s_cbranch_scc1", ret)
        e("s_branch", sites[0][1])

    def first_reference(self, bank, s):
        """tile 0 of a block: the row's first reference is its maximum over the tile (rows that see no key yet keep none)"""
        e = self.e
        r, r2, dl = TMP[8], TMP[9], TMP[10]
        e("v_mov_b32", r, MX[s])
        e("v_mov_b32", r2, MX[s])
        self.nop(2)
        e("v_permlane32_swap_b32", r, r2)
        e("v_max_f32", r, r, r2)                                   # the row's max in both lanes of the pair
        e("v_cmp_gt_f32", VCC, r, V_NEGINF)                        # the row sees a key
        e("v_mov_b32", dl, 0)
        e("v_cndmask_b32", dl, dl, r, VCC)
        e("v_mov_b32", r2, S_THRV)
        e("v_cndmask_b32", THR[s], THR[s], r2, VCC)
        e("v_mov_b32", MREF[s], dl)
        for k in range(16):
            e("v_sub_f32", CI(s)[k], 0, dl)
        for h in range(2):
            t = T(bank, s, h)
            for k in range(16):
                e("v_sub_f32", t[k], t[k], dl)

    # ---- entry: lane constants that hold for the whole launch ----------------------------------------------------------------------
    def entry(self):
        e, P = self.e, self.p
        lane, hi = TMP[0], TMP[1]
        e("v_mbcnt_lo_u32_b32", lane, -1, 0)
        e("v_mbcnt_hi_u32_b32", lane, -1, lane)
        e("v_and_b32", V_LQ, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        e("v_lshlrev_b32", V_HI4, 2, hi)
        if self.timing:
            for a in TACC:
                e("v_mov_b32", a, 0)
            e("s_memtime", S_TNOW)
            e("s_waitcnt", lgkmcnt=0)
            e("s_mov_b32", S_TLAST, S_TNOW[0])
        if self.persistent:
            e("s_lshl_b32", S_W1024, self.inp["wave"], 10)
            e("s_mov_b32", S_BLK, self.inp["first"])
            e("s_mov_b32", S_NENT, self.inp["n"])
            e("s_mov_b32", S_G, self.inp["g"])                   # the workgroup's stride through the table alternates (snake order: a
            e("s_mov_b32", S_G2, self.inp["g2"])                 #  workgroup that took an early = long block of a round takes a late one next)
            e("s_mov_b64", S_TAB, self.inp["tab"])
            e("s_cmp_ge_u32", S_BLK, S_NENT)
            e("s_cbranch_scc1", "KERNEL_END")
            self.load_entry()                                     # strides are filled in every entry, valid or not
            e("s_waitcnt", lgkmcnt=0)
        else:
            e("v_mov_b32", TMP[2], self.inp["param"])
            for k in range(14):
                e("ds_read_b128", V(4 * k, 4), TMP[2], offset=16 * k)
            e("s_waitcnt", lgkmcnt=0)
            for i in range(52):
                e("v_readfirstlane_b32", S(WIN + i), V(i))
            for dst, name in ((S_VPAGE, "v_page"), (S_TMP[7], "wave")):
                e("v_readfirstlane_b32", dst, V(PIDX[name]))
            e("s_lshl_b32", S_W1024, S_TMP[7], 10)
            e("s_mov_b64", S_KBASE, S(S_KDESC.idx, 2))
            e("s_mov_b64", S_VBASE, S(S_VDESC.idx, 2))
            e("s_mov_b32", S_KDESC[3], 0x00020000)
            e("s_mov_b32", S_VDESC[3], 0x00020000)
        # LDS-DMA: lane -> row 4w + lane/16 of a 16-row group, 16-byte slot lane%16; source chunk = slot ^ swizzle(row)
        e("s_lshr_b32", S_TMP[7], S_W1024, 10)                    # wave
        e("v_lshrrev_b32", TMP[3], 4, lane)                       # lane / 16
        e("v_lshl_add_u32", TMP[3], S_TMP[7], 2, TMP[3])          # row16 = 4w + lane/16
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
        e("s_lshl_b32", S_TMP[0], S_KSTRIDE, 4)                   # 16 rows
        e("s_lshl_b32", S_TMP[1], S_VSTRIDE, 4)
        for u in range(1, 4):
            if self.paged:                                        # every piece has its own descriptor base: same offset
                e("v_mov_b32", VOFK[u], VOFK[0])
                e("v_mov_b32", VOFV[u], VOFV[0])
            else:
                e("v_add_u32", VOFK[u], S_TMP[0], VOFK[u - 1])
                e("v_add_u32", VOFV[u], S_TMP[1], VOFV[u - 1])
        # K reads: row lq, 16-byte chunk (2j + hi) ^ (lq & 15); the addresses start at ring slot 0
        e("v_and_b32", TMP[3], 15, V_LQ)
        e("v_lshlrev_b32", TMP[4], 8, V_LQ)                       # lq * 256
        for j in range(8):
            e("v_xor_b32", TMP[5], 2 * j, TMP[3])
            e("v_xor_b32", TMP[5], TMP[5], hi)                    # 2j is even: xor with hi = + hi
            e("v_lshl_add_u32", KAD[j], TMP[5], 4, TMP[4])
        # V^T reads: vrow = 4hi + ((lane & 15) >> 2); dcol = 32db + 16((lane >> 4) & 1) + 4(lane & 3); new_block() moves them to V slot 2
        e("v_and_b32", TMP[3], 15, lane)
        e("v_lshrrev_b32", TMP[3], 2, TMP[3])                     # jrow
        e("v_add_u32", TMP[4], V_HI4, TMP[3])                     # vrow
        e("v_lshlrev_b32", TMP[5], 2, TMP[3])                     # (vrow & 3) << 2 = jrow << 2
        e("v_lshrrev_b32", TMP[6], 4, lane)
        e("v_and_b32", TMP[6], 1, TMP[6])                         # (lane >> 4) & 1
        e("v_and_b32", TMP[7], 3, lane)                           # cc
        for db in range(4):
            e("v_lshrrev_b32", TMP[8], 1, TMP[7])                 # chunk = (dcol >> 3) = 4db + 2 g + (cc >> 1);  (dcol & 7) * 2 = (cc & 1) * 8
            e("v_lshl_add_u32", TMP[8], TMP[6], 1, TMP[8])
            e("v_add_u32", TMP[8], 4 * db, TMP[8])
            e("v_xor_b32", TMP[8], TMP[8], TMP[5])
            e("v_lshlrev_b32", TMP[8], 4, TMP[8])
            e("v_and_b32", TMP[9], 1, TMP[7])
            e("v_lshl_add_u32", TMP[8], TMP[9], 3, TMP[8])
            e("v_lshl_add_u32", VAD[db], TMP[4], 8, TMP[8])
            e("v_add_u32", VAD[db], 3 * SLOT, VAD[db])
        e("v_mov_b32", V_NEGINF, NEG_INF)
        # epilogue staging (this wavefront's 16 KiB at LDS_STAGE + 16K w): write address of the lane's row, 16-byte chunk index xor (row & 15)
        e("s_lshl_b32", S_TMP[0], S_TMP[7], 14)
        e("s_add_u32", S_TMP[0], S_TMP[0], LDS_STAGE)
        e("v_lshlrev_b32", TMP[3], 8, V_LQ)
        e("v_lshl_add_u32", TMP[3], hi, 3, TMP[3])                # lq * 256 + 8 hi
        e("v_and_b32", TMP[4], 15, V_LQ)
        e("v_lshlrev_b32", TMP[4], 4, TMP[4])
        e("v_xor_b32", TMP[3], TMP[3], TMP[4])
        e("v_add_u32", V_EPW, S_TMP[0], TMP[3])
        # ... and the read addresses: instruction k reads rows 4k + lane/16, chunk lane%16 (de-swizzled); 4 row groups (k & 3)
        e("v_lshrrev_b32", TMP[3], 4, lane)                       # rowl
        e("v_and_b32", TMP[4], 15, lane)                          # c
        for kk in range(4):
            e("v_add_u32", TMP[5], 4 * kk, TMP[3])
            e("v_xor_b32", TMP[5], TMP[5], TMP[4])                # c ^ (4 kk + rowl)
            e("v_lshlrev_b32", TMP[5], 4, TMP[5])
            e("v_lshl_add_u32", TMP[5], TMP[3], 8, TMP[5])        # + rowl * 256
            e("v_add_u32", TMP[5], kk * 1024, TMP[5])             # + 4 kk rows
            e("v_add_u32", EPR[kk], S_TMP[0], TMP[5])

    def load_entry(self):
        """persistent: the plan table's entry of (S_BLK, this wavefront) -> s48 .. s99 (asynchronous: s_waitcnt lgkmcnt(0) before use)"""
        e = self.e
        e("s_lshl_b32", S_TMP[0], S_BLK, 10)                      # 4 wavefronts x 256 bytes
        e("s_lshr_b32", S_TMP[1], S_W1024, 2)                     # wave * 256
        e("s_add_u32", S_TMP[0], S_TMP[0], S_TMP[1])
        e("s_lshr_b32", S_TMP[1], S_BLK, 22)
        e("s_add_u32", S_PAIR[0], S_TAB[0], S_TMP[0])
        e("s_addc_u32", S_PAIR[1], S_TAB[1], S_TMP[1])
        for k in range(3):
            e("s_load_dwordx16", S(WIN + 16 * k, 16), S_PAIR, 64 * k)

    def new_block(self):
        """A valid entry sits in s48..: read addresses back to K slot 0 / V slot 2 (they followed the ring: address = base + S_RBASE;
        S_RBASE = 0 before the first block), then the requests.  Persistent: entered at CHECK_ENTRY with the entry in flight; empty
        entries (a query block past the end of a short sequence, the padded tail of an XCD's list) are skipped; S_HASNEXT = 1 while
        the previous block's epilogue is still to come (block_end), S_T = 1 when a block has been set up."""
        e, P = self.e, self.p
        if self.persistent:
            P.label("CHECK_ENTRY")
            e("s_waitcnt", lgkmcnt=0)
            e("s_bitcmp1_b32", W("flags"), 1)
            e("s_cbranch_scc1", "NEW_BLOCK")
            P.extend(self.advance())
            e("s_mov_b32", S_T1, 0)                               # (the prefetched Q rows were this empty entry's: not the block that follows)
            e("s_cmp_lt_u32", S_BLK, S_NENT)
            e("s_cbranch_scc1", "FETCH_ENTRY")
            e("s_mov_b32", S_T, 0)                                # the table is exhausted
            e("s_cmp_eq_u32", S_HASNEXT, 0)
            e("s_cbranch_scc1", "KERNEL_END")
            e("s_branch", "EPI_BOTH")
            P.label("FETCH_ENTRY")
            self.load_entry()
            e("s_branch", "CHECK_ENTRY")
        P.label("NEW_BLOCK")
        P.extend(self.stamp(1, "block"))
        for j in range(8):
            e("v_subrev_u32", KAD[j], S_RBASE, KAD[j])
        e("s_sub_u32", S_TMP[0], S_RBASE, 2 * SLOT)
        for db in range(4):
            e("v_subrev_u32", VAD[db], S_TMP[0], VAD[db])
        if self.persistent:
            e("s_mov_b32", S_T, 1)
            e("s_cmp_eq_u32", S_HASNEXT, 1)
            e("s_cbranch_scc0", "COLD_START")
            e("s_cmp_eq_u32", S_T1, 1)                          # the requests ride inside the previous block's epilogue (block_end);
            e("s_cbranch_scc1", "EPI_WOVEN_NOQ")                #  the Q rows, usually, already went out with its last iteration
            e("s_branch", "EPI_WOVEN")
            P.label("COLD_START")
        self.pro_issue("A")

    def advance(self):
        return [Ins("s_add_u32", S_BLK, S_BLK, S_G), Ins("s_mov_b32", S_TMP[3], S_G), Ins("s_mov_b32", S_G, S_G2), Ins("s_mov_b32", S_G2, S_TMP[3])]

    # ---- per block: requests first ----------------------------------------------------------------------------------------------------
    def pro_issue(self, tag):
        """Q of the slots that see keys, then the first five K/V tiles.  K tile i lives in K slot i % 3, V tile i in V slot
        (i + 2) % 3; K(3) takes K(0)'s slot and is requested once every wavefront has fetched its K(0) fragments (pro_compute)."""
        e, P = self.e, self.p
        hi = TMP[1]
        e("v_lshrrev_b32", hi, 2, V_HI4)
        for s, sn in ((0, S_N0), (1, S_N1)):
            skip = f"PRO_NOQ_{tag}{s}"
            e("s_cmp_eq_u32", sn, 0)
            e("s_cbranch_scc1", skip)
            e("s_sub_u32", S_TMP[0], S_ROWS[s], 1)
            e("v_min_u32", TMP[10], S_TMP[0], V_LQ)               # clamp to the slot's last valid row
            e("v_mul_lo_u32", TMP[10], TMP[10], S_QST)
            e("v_lshl_add_u32", TMP[12], hi, 4, TMP[10])          # + hi * 16 bytes
            for j in range(8):
                e("global_load_dwordx4", QA(s, j), TMP[12], W(f"q{s}_lo", 2), offset=32 * j)
            P.label(skip)
        P.extend(self.stamp(2, "block"))
        if self.paged:
            e("s_mov_b32", S_KVT, 0)
            e("s_mov_b32", S_VVT, 0)
        else:
            e("s_mov_b32", S_KSOFF, 0)
            e("s_mov_b32", S_VSOFF, 0)
        self.request_tile(False, 0)
        self.request_tile(False, 1)
        if not self.split_req:
            self.request_tile(True, 2)
            self.request_tile(False, 2)
            self.request_tile(True, 0)

    def req_items(self, tiles):
        """the requests of K/V tiles [(is_v, ring slot)] as scheduler items (contiguous K/V): 4 pieces + the offset step per tile"""
        items = []
        for is_v, slot in tiles:
            its = self.dma_tile(is_v)
            its[0].ins = [Ins("s_add_u32", S_KDMA, S_W1024, slot * SLOT)] + its[0].ins
            items += its
        return items

    def q_offsets(self, regs, rows=None, ns=None, hi=None, tmp=None):
        """per-lane byte offsets of the Q rows of both slots -> regs[0..1]; a slot that sees no key reads (harmlessly) the head of
        the plan table instead: its base moves there and its offsets collapse to 0 -- no branch, so the loads can ride anywhere.
        Returns the instruction list (rows / ns: the registers that hold the slots' valid rows / tile counts)."""
        rows, ns = rows or S_ROWS, ns or (S_N0, S_N1)
        hi, t0, t1 = hi or TMP[1], (tmp or S_TMP)[0], (tmp or S_TMP)[1]
        out = [Ins("v_lshrrev_b32", hi, 2, V_HI4)]
        for s in range(2):
            q = W(f"q{s}_lo", 2)
            out += [Ins("s_sub_u32", t0, rows[s], 1), Ins("s_max_i32", t0, t0, 0),
                    Ins("v_min_u32", regs[s], t0, V_LQ),                        # clamp to the slot's last valid row
                    Ins("v_mul_lo_u32", regs[s], regs[s], S_QST),
                    Ins("v_lshl_add_u32", regs[s], hi, 4, regs[s]),             # + hi * 16 bytes
                    Ins("s_cmp_eq_u32", ns[s], 0), Ins("s_cselect_b32", t1, 0, 1),
                    Ins("v_mul_lo_u32", regs[s], regs[s], t1),
                    Ins("s_cselect_b32", q[0], S_TAB[0], q[0]), Ins("s_cselect_b32", q[1], S_TAB[1], q[1])]
        return out

    # ---- persistent: the NEXT block's Q rows are requested by the CURRENT block's last active iteration ---------------------------------
    # (16 row-strided loads cost ~230 cycles each when issued back to back between two blocks: 3.7k cycles per block, `profiles/r04_prefill_block_regions.txt`;
    #  under the P.V MFMAs of a draining iteration they cost their issue slots.)  Q^T lives in accumulator registers that the draining iteration
    #  no longer reads, so the loads land there directly.  The entry after this one is "peeked" at the start of the block: its Q pointers
    #  replace this block's (dead after its own loads) in s64..s67, rows / tile counts / flags go to S_SAVE[0..4]; S_SAVE[5] = 1 once the loads are out.
    def peek_next(self):
        e = self.e
        e("s_mov_b32", S_SAVE[4], 0)
        e("s_mov_b32", S_SAVE[5], 0)
        e("s_add_u32", S_TMP[2], S_BLK, S_G)
        e("s_cmp_ge_u32", S_TMP[2], S_NENT)
        e("s_cbranch_scc1", "PEEK_NONE")
        e("s_lshl_b32", S_TMP[0], S_TMP[2], 10)
        e("s_lshr_b32", S_TMP[1], S_W1024, 2)
        e("s_add_u32", S_TMP[0], S_TMP[0], S_TMP[1])
        e("s_lshr_b32", S_TMP[1], S_TMP[2], 22)
        e("s_add_u32", S_PAIR[0], S_TAB[0], S_TMP[0])
        e("s_addc_u32", S_PAIR[1], S_TAB[1], S_TMP[1])
        e("s_load_dwordx4", W("q0_lo", 4), S_PAIR, 4 * PIDX["q0_lo"])
        e("s_load_dwordx2", S(S_SAVE[0].idx, 2), S_PAIR, 4 * PIDX["rows0"])
        e("s_load_dwordx2", S(S_SAVE[2].idx, 2), S_PAIR, 4 * PIDX["n0"])
        e("s_load_dword", S_SAVE[4], S_PAIR, 4 * PIDX["flags"])
        self.p.label("PEEK_NONE")

    def qpre_code(self):
        """requests of the peeked entry's Q rows -> QA; an invalid / empty entry turns them into harmless reads of the table's head"""
        v = [Ins("s_bitcmp1_b32", S_SAVE[4], 1),                               # valid entry?
             Ins("s_cselect_b32", S_SAVE[2], S_SAVE[2], 0), Ins("s_cselect_b32", S_SAVE[3], S_SAVE[3], 0),
             Ins("s_cselect_b32", S_SAVE[5], 1, 0)]
        v += self.q_offsets(LIMREL, rows=(S_SAVE[0], S_SAVE[1]), ns=(S_SAVE[2], S_SAVE[3]), hi=MXB[0], tmp=(S_SAVE[6], S_SAVE[7]))
        loads = [Ins("global_load_dwordx4", QA(s, j), LIMREL[s], W(f"q{s}_lo", 2), offset=32 * j) for s in range(2) for j in range(8)]
        return v, loads


    def request_groups(self, regs, with_q=True):
        """the next block's requests as instruction groups (block_end weaves them into the epilogue): [16 Q loads,] K(0) K(1) V(0) K(2) V(1)"""
        groups = []
        for s in range(2 if with_q else 0):
            for j in range(8):
                groups.append([Ins("global_load_dwordx4", QA(s, j), regs[s], W(f"q{s}_lo", 2), offset=32 * j)])
        first = True
        if os.environ.get("PFA_EXP", "") == "noreq":          # TIMING EXPERIMENT ONLY (wrong results): what the K/V requests between two blocks cost
            return groups
        # (split_req: only K(0) and K(1) ride in the epilogue -- 20 pieces back to back stall the issue for 210 cycles each, the CU's LDS-DMA
        #  rate; V(0), K(2) go out under the S(0) products of pro_compute, V(1) with K(3) between its mask / max sections)
        for is_v, slot in (((False, 0), (False, 1)) if self.split_req else ((False, 0), (False, 1), (True, 2), (False, 2), (True, 0))):
            items = self.dma_tile(is_v)
            head = [Ins("s_add_u32", S_KDMA, S_W1024, slot * SLOT)]
            if first:
                head = [Ins("s_mov_b32", S_KSOFF, 0), Ins("s_mov_b32", S_VSOFF, 0)] + head
                first = False
            for k, it in enumerate(items):
                groups.append((head if k == 0 else []) + it.ins)
        return groups

    # ---- per block: S(0), the rows' first reference ----------------------------------------------------------------------------------------
    def pro_compute(self):
        """Barriers: K(0) landed, K(1) landed, entry -- the same count on every path."""
        e, P = self.e, self.p
        P.label("PRO_COMPUTE")
        if self.persistent:
            self.peek_next()
        # masks: last visible key of this lane's row, per slot
        e("s_sub_u32", S_TMP[5], S_LENK, 1)
        e("v_mul_lo_u32", TMP[3], V_LQ, W("lim_step"))
        for s in range(2):
            e("v_add_u32", LIM[s], W(f"lim{s}"), TMP[3])
            e("v_min_i32", LIM[s], S_TMP[5], LIM[s])
        # softmax state; O = 0 and C-init = 0 through the matrix pipe (0 x 0 + 0: 10 instructions instead of 160)
        for s in range(2):
            e("v_mov_b32", LS(s, 0), 0)
            e("v_mov_b32", LS(s, 1), 0)
            e("v_mov_b32", MREF[s], 0)
            e("v_mov_b32", THR[s], NEG_INF)
            e("v_mov_b32", MX[s], NEG_INF)
        z = VV(0)
        for k in range(4):
            e("v_mov_b32", z[k], 0)
        self.nop(2)
        for s in range(2):
            for db in range(4):
                e(self.mfma, OA(s, db), z, z, 0)
        for s in range(2):
            e(self.mfma, CI(s), z, z, 0)
        e("s_mov_b32", S_T, 0)
        e("s_cmp_eq_u32", S_NT, 0)
        e("s_cbranch_scc1", "BLOCK_DRAIN")
        e("s_cmp_eq_u32", S_N1, 0)
        e("s_cbranch_scc1", "PRO_IDLE")
        e("v_mov_b32", TMP[3], S_SCALE)

        def q_to_acc(s, exact, js=range(8)):
            """Q^T of slot s, d-steps js, in the accumulator file; fast: pre-multiplied by scale.log2(e) and rounded once, in place"""
            out = []
            for j in js:
                for k in range(4):
                    if exact:
                        continue                                  # Q^T is already where the MFMAs read it
                    x = TMP[8 + (k & 1)]
                    out.append(Ins("v_accvgpr_read_b32", x, QA(s, j)[k]))
                    lo, hi2 = TMP[4 + 2 * (k & 1)], TMP[5 + 2 * (k & 1)]
                    if self.dtype == "bf16":
                        out += [Ins("v_lshlrev_b32", lo, 16, x), Ins("v_and_b32", hi2, 0xFFFF0000, x)]
                    else:
                        out += [Ins("v_cvt_f32_f16", lo, x), Ins("v_lshrrev_b32", hi2, 16, x), Ins("v_cvt_f32_f16", hi2, hi2)]
                    out += [Ins("v_mul_f32", lo, lo, TMP[3]), Ins("v_mul_f32", hi2, hi2, TMP[3]), Ins(self.cvt, lo, lo, hi2),
                            Ins("v_accvgpr_write_b32", QA(s, j)[k], lo)]
            return out

        split = self.split_req

        def k0_fragments():
            e("s_waitcnt", vmcnt=4 if split else 16)
            e("s_barrier")                                         # K(0) landed
            P.extend(self.stamp(4, "block"))
            P.extend(self.k_reads())                               # KAD points at slot 0 here
            e("s_waitcnt", lgkmcnt=0)
            self.nop(2)
        e("s_waitcnt", vmcnt=8 if split else 20)                   # Q arrived (first on the queue: the K/V pieces were issued after it)
        for exact, tag in ((False, "F"), (True, "X")):
            if not exact:
                e("s_bitcmp1_b32", W("flags"), 0)
                e("s_cbranch_scc1", "PRO_QX")
            else:
                P.label("PRO_QX")
            one = f"PRO_ONE_{tag}"
            e("s_cmp_eq_u32", S_N0, 0)
            e("s_cbranch_scc1", one)
            # ---- both slots see keys: the d-steps of Q are converted just ahead of the S(0) MFMAs that take them (MFMA 2j of slot 0,
            # 16 + 2j of slot 1, with a margin of two instructions for the accumulator-write -> MFMA hazard) ----
            P.extend(q_to_acc(0, exact, (0, 1)))
            k0_fragments()
            a = [Item(i, -1, max(2 * j - 3, -1)) for j in range(2, 8) for i in q_to_acc(0, exact, (j,))]
            b = [Item(i, -1, 13 + 2 * j) for j in range(8) for i in q_to_acc(1, exact, (j,))]
            c = self.req_items(((True, 2), (False, 2))) if split else []          # V(0), K(2): spread over the 32 products
            for i, it in enumerate(c):
                it.deadline = 2 + (i * 28) // max(len(c), 1)
            P.extend(schedule(self.qk_mfmas(0, [0]) + self.qk_mfmas(0, [1]), [a, b, c], cap=12))
            e("s_branch", "PRO_K1")
            # ---- slot 1 only ----
            P.label(one)
            P.extend(q_to_acc(1, exact, (0, 1)))
            k0_fragments()
            a = [Item(i, -1, max(2 * j - 3, -1)) for j in range(2, 8) for i in q_to_acc(1, exact, (j,))]
            c = self.req_items(((True, 2), (False, 2))) if split else []
            for i, it in enumerate(c):
                it.deadline = 1 + (i * 14) // max(len(c), 1)
            P.extend(schedule(self.qk_mfmas(0, [1]), [a, c], cap=12))
            if not exact:
                e("s_branch", "PRO_K1")
        # ---- K(1) fragments (K(1) sits in slot 1); K(3) may now replace K(0) ----
        P.label("PRO_K1")
        e("s_waitcnt", vmcnt=8 if split else 12)
        e("s_barrier")
        P.extend(self.k_reads(extra=SLOT))
        for j in range(8):
            e("v_add_u32", KAD[j], 2 * SLOT, KAD[j])               # next K read: K(2) in slot 2
        # K(3) into K(0)'s slot [+ V(1)]: the pieces in three portions around the two slots' mask / max sections
        late = self.req_items(((False, 0), (True, 0))) if split else None
        if not split:
            self.request_tile(False, 0)
        portions = [late[0:3], late[3:7], late[7:]] if split else [[], [], []]
        for it in portions[0]:
            P.extend(it.ins)
        # ---- [mask] + row max of tile 0, the rows' first reference ----
        e("s_mov_b32", S_T1, 0)                                    # kv1 = 0 for the mask code
        self.nop(MFMA_SAFE)
        for s, sn, stm in ((0, S_N0, S_TM0), (1, S_N1, S_TM1)):
            skip, nomask = f"PRO_NOMAX_{s}", f"PRO_NOMASK_{s}"
            if s == 0:
                e("s_cmp_eq_u32", sn, 0)
                e("s_cbranch_scc1", skip)
            e("s_cmp_eq_u32", stm, 0)                              # tile 0 needs the mask only when the slot's first masked tile is tile 0
            e("s_cbranch_scc0", nomask)
            for it in self.mask_items(0, s, -1):
                P.extend(it.ins)
            P.label(nomask)
            for it in self.max_items(0, s, -1):
                P.extend(it.ins)
            self.first_reference(0, s)
            P.label(skip)
            for it in portions[1 + s]:
                P.extend(it.ins)
        self.nop(MFMA_SRCC_SAFE)
        e("s_branch", "PRO_ENTRY")
        # ---- no row of this wavefront sees a key: only the barriers and its share of the DMA ----
        P.label("PRO_IDLE")
        if split:
            self.request_tile(True, 2)                            # V(0), K(2): this wavefront's share (nobody else requests its rows)
            self.request_tile(False, 2)
        if self.persistent:                                       # this wavefront has no draining iteration: its share of the next block's Q goes out here
            e("s_waitcnt", vmcnt=16 if split else 20, lgkmcnt=0)   # (behind this block's own -- unused -- Q loads: same registers)
            v, loads = self.qpre_code()
            P.extend(v + loads)
        e("s_waitcnt", vmcnt=(12 if split else 16) + (16 if self.persistent else 0))
        e("s_barrier")
        e("s_waitcnt", vmcnt=(8 if split else 12) + (16 if self.persistent else 0))
        e("s_barrier")
        for j in range(8):
            e("v_add_u32", KAD[j], 2 * SLOT, KAD[j])
        self.request_tile(False, 0)
        if split:
            self.request_tile(True, 0)
        if self.persistent:
            e("s_waitcnt", vmcnt=8 + 16, lgkmcnt=0)                # (the 16 prefetch loads are younger than V(0), K(2); K(3), V(1) younger still)
            e("s_branch", "PRO_ENTRY_STATE")
        P.label("PRO_ENTRY")
        e("s_waitcnt", vmcnt=8, lgkmcnt=0)
        P.label("PRO_ENTRY_STATE")
        # state at the entry of iteration 0: read slot r = 2 (K(2) and V(0) live in slot 2), DMA slot d = 1
        e("s_mov_b32", S_RBASE, 2 * SLOT)
        e("s_mov_b32", S_DBASE, 1 * SLOT)
        e("s_add_u32", S_KDMA, S_DBASE, S_W1024)
        e("s_mov_b32", S_DELTA, 0)
        e("s_barrier")
        P.extend(self.stamp(4))
        P.extend(self.stamp(5, "block"))
        e("s_bitcmp1_b32", W("flags"), 0)
        e("s_cbranch_scc1", "DISPX_0")
        e("s_branch", "DISP_0")

    # ---- per block: normalise, transpose through LDS, store -----------------------------------------------------------------------------------
    def epi_slot(self, s, tag, odesc, lse, rows, ost, mscale, weave=()):
        """O^T of slot s (accumulators) -> 1/l -> storage type -> this wavefront's LDS staging rows (swizzled) -> whole 256-byte rows
        to memory, 4 rows per store instruction (a per-lane row-strided store would touch 32 partial lines per instruction)"""
        e, P = self.e, self.p
        done, norows = f"EPI_DONE_{tag}{s}", f"EPI_NOROWS_{tag}{s}"
        e("s_cmp_eq_u32", rows, 0)
        e("s_cbranch_scc1", norows)
        ltot, inv, t0, t1 = TMP[3], TMP[4], TMP[5], TMP[6]
        e("v_add_f32", t0, LS(s, 0), LS(s, 1))
        e("v_mov_b32", t1, t0)
        self.nop(2)
        e("v_permlane32_swap_b32", t0, t1)
        e("v_add_f32", ltot, t0, t1)
        e("v_rcp_f32", inv, ltot)
        e("v_cmp_gt_f32", VCC, ltot, 0)
        e("v_mov_b32", t0, 0)
        e("v_cndmask_b32", inv, t0, inv, VCC)                      # rows that saw no key: 0
        f = [TMP[0], TMP[2], TMP[7], TMP[8]]
        body = []
        for db in range(4):
            o = OA(s, db)
            for r4 in range(4):
                w = [TMP[10 + 2 * (r4 & 1)], TMP[11 + 2 * (r4 & 1)]]
                body += [Ins("v_accvgpr_read_b32", f[k], o[4 * r4 + k]) for k in range(4)]
                body += [Ins("v_mul_f32", f[k], f[k], inv) for k in range(4)]
                body += [Ins(self.cvt, w[0], f[0], f[1]), Ins(self.cvt, w[1], f[2], f[3]),
                         Ins("v_xor_b32", t1, (4 * db + r4) * 16, V_EPW),      # 16-byte chunk 4db + r4 of the row, swizzled; + 8 hi is in V_EPW
                         Ins("ds_write_b64", t1, V(w[0].idx, 2), offset=s * 8192)]
        # the next block's requests, evenly between these instructions
        n = len(weave)
        cut = [round((i + 1) * len(body) / (n + 1)) for i in range(n)]
        pos = 0
        for i in range(n):
            P.extend(body[pos:cut[i]])
            P.extend(weave[i])
            pos = cut[i]
        P.extend(body[pos:])
        e("s_waitcnt", lgkmcnt=0)
        # rows 4k + lane/16 of the slot, 16 bytes per lane
        e("v_lshrrev_b32", t0, 4, TMP[9])                          # TMP[9] = lane (set by the caller)
        e("v_mul_lo_u32", t0, t0, ost)
        e("v_and_b32", t1, 15, TMP[9])
        e("v_lshl_add_u32", t0, t1, 4, t0)                         # (lane / 16) * stride + (lane % 16) * 16
        e("s_mov_b32", S_TMP[0], 0)
        e("s_lshl_b32", S_TMP[1], ost, 2)
        for k in range(8):                                         # the V^T window (32 registers) is dead here: all reads, one wait, all stores
            e("ds_read_b128", VV(k), EPR[k & 3], offset=s * 8192 + (k >> 2) * 4096)
        for k in range(8):
            e("s_waitcnt", lgkmcnt=7 - k)
            e("buffer_store_dwordx4", VV(k), t0, odesc, S_TMP[0], offen=True)
            e("s_add_u32", S_TMP[0], S_TMP[0], S_TMP[1])
        # log-sum-exp (natural log), rows without keys: +inf
        nolse = f"EPI_NOLSE_{tag}{s}"
        e("s_or_b32", S_TMP[0], lse[0], lse[1])
        e("s_cmp_eq_u32", S_TMP[0], 0)
        e("s_cbranch_scc1", nolse)
        e("s_mov_b64", S_PAIR, S(lse[0].idx, 2))
        e("s_mov_b32", S_TMP[4], S_PAIR[0])
        e("s_mov_b32", S_TMP[5], S_PAIR[1])
        e("s_lshl_b32", S_TMP[6], rows, 2)
        e("s_mov_b32", S_TMP[7], 0x00020000)
        e("v_log_f32", t0, ltot)
        e("v_mul_f32", t1, mscale, MREF[s])
        e("v_add_f32", t0, t0, t1)
        e("v_mul_f32", t0, 0.6931471805599453, t0)
        e("v_mov_b32", t1, float("inf"))
        e("v_cndmask_b32", t0, t1, t0, VCC)
        e("v_lshlrev_b32", t1, 2, V_LQ)
        e("v_cmp_eq_u32", S_PAIR, 0, V_HI4)                        # one lane of each pair stores
        e("v_mov_b32", TMP[0], 0x7FFFFFF0)
        e("v_cndmask_b32", t1, TMP[0], t1, S_PAIR)
        e("buffer_store_dword", t0, t1, S(S_TMP[4].idx, 4), 0, offen=True)
        P.label(nolse)
        if weave:
            e("s_branch", done)
            P.label(norows)                # no row of this slot exists: only the requests
            for g in weave:
                P.extend(g)
        else:
            P.label(norows)
        P.label(done)

    def block_end(self):
        """Non-persistent: wait, barrier, both epilogues.  Persistent: the epilogue's parameters move to accumulator registers that
        are dead by now (the K fragments), the NEXT block's entry is fetched and its Q rows and first K/V tiles requested
        (new_block), and only then this block's O is normalised and stored: the requests' round trips ride under the stores."""
        e, P = self.e, self.p
        stash = ["o0_lo", "o0_hi", "o0_bytes", "o0_flags", "lse0_lo", "lse0_hi", "rows0", "mscale",
                 "o1_lo", "o1_hi", "o1_bytes", "o1_flags", "lse1_lo", "lse1_hi", "rows1", "o_stride"]
        if not self.persistent:
            P.label("BLOCK_DRAIN")
            P.label("BLOCK_END")
            P.extend(self.stamp(3))
            P.extend(self.stamp(None, "block"))
            self.nop(MFMA_SAFE)
            e("s_waitcnt", vmcnt=0, lgkmcnt=0)
            e("s_barrier")                     # every wavefront is done with the ring and with its requests
            P.extend(self.stamp(0, "block"))
            e("v_mbcnt_lo_u32_b32", TMP[9], -1, 0)
            e("v_mbcnt_hi_u32_b32", TMP[9], -1, TMP[9])
            self.epi_slot(0, "A", S_ODESC[0], [W("lse0_lo"), W("lse0_hi")], S_ROWS[0], S_OST, W("mscale"))
            self.epi_slot(1, "A", S_ODESC[1], [W("lse1_lo"), W("lse1_hi")], S_ROWS[1], S_OST, W("mscale"))
            P.label("KERNEL_END")
            return
        # Persistent.  Three ways in, differing in what must have landed before the ring is reused (S_T = 1: everything but the 16 youngest
        # requests -- the next block's Q rows, on their way into accumulator registers).  The epilogue's parameters move to the (dead) K
        # fragment registers and the NEXT entry's fetch goes out BEFORE the drain wait and the barrier: its round trip to memory (1.2k
        # cycles after the barrier, `profiles/r04_prefill_block_regions.txt`) rides under them.
        P.label("BLOCK_DRAIN")             # no K/V tile at all: every request in flight must land
        # The ring state a block leaves behind is "reads at slot 2" (K read addresses = slot 0 + S_RBASE, V = slot 2 + S_RBASE - 2 slots: new_block
        # rewinds both by S_RBASE).  A block without tiles never reaches PRO_K1 / PRO_ENTRY_STATE, which establish it: until round 6 it left the
        # PREVIOUS block's S_RBASE next to read addresses that new_block had already rewound, so the block AFTER it rewound them a second time
        # and read its K fragments from somewhere else in LDS (found by tests/fuzz_parity.py: kv_cache call, 3 query rows, a sequence without keys
        # in the middle of the batch; the simulator reproduces it: test_block_without_keys_between_two_blocks).
        for j in range(8):
            e("v_add_u32", KAD[j], 2 * SLOT, KAD[j])
        e("s_mov_b32", S_RBASE, 2 * SLOT)
        e("s_mov_b32", S_T, 0)
        e("s_branch", "BLOCK_END_COMMON")
        P.label("BLOCK_END_Q")
        P.extend(self.stamp(3))
        P.extend(self.stamp(None, "block"))
        e("s_mov_b32", S_T, 1)
        e("s_branch", "BLOCK_END_COMMON")
        P.label("BLOCK_END")
        P.extend(self.stamp(3))
        P.extend(self.stamp(None, "block"))
        e("s_mov_b32", S_T, 0)
        P.label("BLOCK_END_COMMON")
        self.nop(MFMA_SAFE)
        for k, name in enumerate(stash):
            e("v_accvgpr_write_b32", A(192 + k), W(name))
        e("s_mov_b32", S_HASNEXT, 1)
        e("s_mov_b32", S_T1, S_SAVE[5])                           # 1: the next entry's Q rows are already in (or on their way to) QA
        P.extend(self.advance())
        e("s_cmp_lt_u32", S_BLK, S_NENT)
        e("s_cbranch_scc0", "BLOCK_END_WAIT")
        e("s_waitcnt", lgkmcnt=0)                                  # (a peeked entry nobody consumed may still be on its way into the window)
        self.load_entry()
        P.label("BLOCK_END_WAIT")
        e("s_cmp_eq_u32", S_T, 1)
        e("s_cbranch_scc1", "BLOCK_END_WAITQ")
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        e("s_branch", "BLOCK_END_BAR")
        P.label("BLOCK_END_WAITQ")
        e("s_waitcnt", vmcnt=16, lgkmcnt=0)
        P.label("BLOCK_END_BAR")
        e("s_barrier")                     # every wavefront is done with the ring and with its requests
        P.extend(self.stamp(0, "block"))
        e("s_mov_b32", S_T, 0)
        e("s_cmp_lt_u32", S_BLK, S_NENT)
        e("s_cbranch_scc1", "CHECK_ENTRY")
        def epilogue(tag, woven, with_q=True):
            e("s_mov_b32", S_HASNEXT, 0)
            e("v_mbcnt_lo_u32_b32", TMP[9], -1, 0)
            e("v_mbcnt_hi_u32_b32", TMP[9], -1, TMP[9])
            groups = []
            if woven:
                if with_q:
                    P.extend(self.q_offsets(LIMREL))               # (LIMREL is rewritten by every masked tile: free between blocks)
                groups = self.request_groups(LIMREL, with_q)
            half = (len(groups) + 1) // 2
            for s_ in range(2):
                for k in range(8):
                    e("v_accvgpr_read_b32", TMP[k], A(192 + 8 * s_ + k))
                e("v_accvgpr_read_b32", TMP[8], A(192 + 15))
                e("v_accvgpr_read_b32", TMP[10], A(192 + 7))
                self.nop(1)
                for k in range(8):
                    e("v_readfirstlane_b32", S_SAVE[k], TMP[k])
                e("v_readfirstlane_b32", S_TMP[2], TMP[8])              # o_stride
                e("v_readfirstlane_b32", S_TMP[3], TMP[10])             # mscale (slot 0's stash holds it; slot 1's has the stride instead)
                self.epi_slot(s_, tag, S(S_SAVE[0].idx, 4), [S_SAVE[4], S_SAVE[5]], S_SAVE[6], S_TMP[2], S_TMP[3],
                              weave=groups[:half] if s_ == 0 else groups[half:])
                if woven:
                    P.extend(self.stamp(2 + s_, "block"))          # timer build: the epilogue of slot s_ with its share of the woven requests
        P.label("EPI_BOTH")                    # no next block: the plain epilogue
        P.extend(self.stamp(1, "block"))
        epilogue("B", False)
        P.extend(self.stamp(5))
        e("s_branch", "KERNEL_END")
        P.label("EPI_WOVEN")                   # the next block's entry is in s48..: its Q rows and first K/V tiles are requested between the epilogue's
        P.extend(self.stamp(1, "block"))       #  instructions.  (Taken when the Q rows did not go out with the last iteration: an empty entry was
        e("s_waitcnt", vmcnt=0)                #  skipped, or the block had no tile -- whatever that iteration requested into QA lands first.)
        epilogue("C", True)
        P.extend(self.stamp(5))
        e("s_branch", "PRO_COMPUTE")
        P.label("EPI_WOVEN_NOQ")
        P.extend(self.stamp(1, "block"))
        epilogue("D", True, with_q=False)
        P.extend(self.stamp(5))
        e("s_branch", "PRO_COMPUTE")
        P.label("KERNEL_END")
        if self.timing:
            e("s_waitcnt", vmcnt=0)
            e("s_mov_b64", S(S_TMP[4].idx, 2), self.inp["dbg"])
            e("s_mov_b32", S_TMP[6], 32)
            e("s_mov_b32", S_TMP[7], 0x00020000)
            e("s_or_b32", S_T1, S_TMP[4], S_TMP[5])
            e("s_cmp_eq_u32", S_T1, 0)
            e("s_cbranch_scc1", "TIMING_DONE")
            e("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
            e("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])
            e("v_lshlrev_b32", TMP[0], 16, TMP[0])                # only lane 0 is inside the 32-byte window
            e("buffer_store_dwordx4", V(TACC[0].idx, 4), TMP[0], S(S_TMP[4].idx, 4), 0, offen=True)
            e("v_mov_b32", TMP[20], 0)
            e("v_mov_b32", TMP[21], 0)
            e("buffer_store_dwordx4", V(TACC[4].idx, 4), TMP[0], S(S_TMP[4].idx, 4), 0, offen=True, offset=16)
            e("s_waitcnt", vmcnt=0)
            P.label("TIMING_DONE")

    # ---- the whole program ------------------------------------------------------------------------------------------------------------------
    def build(self):
        self.entry()
        self.e("s_mov_b32", S_RBASE, 0)
        self.e("s_mov_b32", S_HASNEXT, 0)
        self.new_block()
        self.pro_compute()
        for exact in (False, True):
            self.exact, self.sfx = exact, "X" if exact else ""
            for p in (0, 1):
                self.dispatcher(p)
            for p in (0, 1):
                self.p.emit("p2align", 6)
                self.iteration(p, 2, 2, False)
            for p in (0, 1):
                for c, x, mk in ((2, 2, True), (2, 1, False), (2, 1, True), (2, 0, False), (1, 1, False), (1, 1, True), (1, 0, False)):
                    self.iteration(p, c, x, mk)
        self.idle_iteration()
        for site, ret, stub, bank_n, x, sfx in self.ret_sites:
            self.p.label(stub)
            self.e("s_mov_b32", S_RET, site)
            self.e("s_branch", f"RESC{sfx}_{bank_n}_{x}")
        for exact in (False, True):
            self.exact, self.sfx = exact, "X" if exact else ""
            for bank_n in (0, 1):
                for x in (2, 1):
                    self.rescale(bank_n, x)
        self.exact, self.sfx = False, ""
        self.block_end()
        return self.p


MFMA_SAFE = 20
MFMA_SRCC_SAFE = 4


def build(dtype="bf16", paged=False, inputs=None, timing=False):
    b = Builder(dtype, paged, inputs, timing)
    prog = b.build()
    return prog, b


def clobbers():
    """registers the asm statement owns (csrc/prefill_asm.hip lists them as clobbered)"""
    return [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(SGPR_FIRST, SGPR_LAST + 1) if i != 32] + ["vcc", "scc", "memory"]
