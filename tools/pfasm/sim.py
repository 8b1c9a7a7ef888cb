"""Functional simulator for the instruction subset of tools/pfasm/isa.py: one workgroup of wave64 wavefronts, LDS,
a flat global memory, and the two asynchronous counters (vmcnt / lgkmcnt) with ADVERSARIAL timing --

  * a VGPR / AGPR / SGPR destination of a load is poisoned from issue until the s_waitcnt that covers it: any read or
    write of a poisoned register raises SimError (a missing or mis-counted wait);
  * LDS-DMA (buffer_load ... lds) lands either at the covering s_waitcnt of the issuing wave (late = True, the latest
    legal time: a consumer that reads before wait + barrier sees the OLD bytes) or at issue (late = False, the earliest:
    a DMA issued before every wave finished reading the slot corrupts those reads).  Tests run both, the second with
    the waves executed in reverse order between barriers;
  * MFMA results are not interlocked for non-MFMA consumers on gfx9: a VALU / DS / VMEM access to a register written by
    an MFMA fewer than MFMA_WAIT issued instructions ago, an MFMA reading as A/B a register written by VALU fewer than
    VALU_MFMA_WAIT ago, ... raise SimError (every instruction counts as one wait state, s_nop N as N + 1: conservative).

Test infrastructure only: nothing under atoma-infer_amd/ imports it.
"""
import numpy as np

from .isa import Reg, Special, Ins, f32_bits

MFMA_WAIT = 18          # 8-pass XDL write -> VALU read/write: 11 wait states; 16-pass would need 19: keep a margin
VALU_MFMA_WAIT = 2      # VALU write VGPR -> MFMA read
MFMA_SRCC_WAR = 16      # MFMA reads srcC late: a VALU write to its srcC registers must stay this far behind


class SimError(Exception):
    pass


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    return np.where(nan, np.uint16(0x7FC0), r).astype(np.uint16)


class Memory:
    """Flat global memory: named allocations at fake addresses; any access outside an allocation is a fault."""

    def __init__(self):
        self.allocs = []      # (base, bytes ndarray, name)
        self.next = 0x10_0000_0000

    def alloc(self, arr, name=""):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.allocs.append((base, b, name))
        self.next += (len(b) + 0xFFFF) & ~0xFFFF | 0x10000
        return base

    def find(self, addr, n):
        for base, b, name in self.allocs:
            if base <= addr and addr + n <= base + len(b):
                return b, addr - base
        raise SimError(f"memory access fault: {n} bytes at {addr:#x}")

    def read(self, addr, n):
        b, o = self.find(int(addr), n)
        return b[o:o + n]

    def write(self, addr, data):
        b, o = self.find(int(addr), len(data))
        b[o:o + len(data)] = data

    def get(self, base, dtype=np.uint8):
        for bb, b, name in self.allocs:
            if bb == base:
                return b.view(dtype)
        raise KeyError(base)


_CD_I = np.array([(r & 3) + 8 * (r >> 2) for r in range(16)])


class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(104, np.uint32)
        self.vcc = np.zeros(64, bool)
        self.scc = False
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.pending = {}            # (kind, idx) -> description
        self.vmq = []                # [complete_fn or None, [(kind, idx)...]]
        self.lgq = []                # [kind 'lds' | 'smem', complete_fn, regs]
        self.issue = 0               # wait-state clock
        self.mfma_w = {}             # (kind, idx) -> issue index of the last MFMA that wrote it
        self.valu_w = {}             # (kind, idx) -> issue index of the last VALU write
        self.mfma_c = {}             # (kind, idx) -> issue index of the last MFMA that read it as srcC
        self.stats = {}
        # poison patterns so that uninitialised registers are conspicuous
        self.v[:] = 0x7FC0DEAD
        self.a[:] = 0x7FC0DEAD

    # ---- register access -------------------------------------------------------------------------------
    def _file(self, kind):
        return self.v if kind == "v" else self.a

    def _chk(self, kind, idx, what):
        if (kind, idx) in self.pending:
            raise SimError(f"wave {self.wid} pc {self.pc}: {what} of {kind}{idx} while its load is in flight ({self.pending[(kind, idx)]})")

    def rd_vec(self, o, hazard=True):
        """64-lane uint32 view of a single-register operand / constant."""
        if isinstance(o, Reg):
            assert o.n == 1, o
            if o.kind == "s":
                self._chk("s", o.idx, "read")
                return np.full(64, self.s[o.idx], np.uint32)
            self._chk(o.kind, o.idx, "read")
            if hazard:
                self._mfma_hazard(o.kind, o.idx, "read")
            return self._file(o.kind)[o.idx]
        if isinstance(o, Special):
            if o.name == "m0":
                return np.full(64, self.m0, np.uint32)
            raise SimError(f"vector read of {o}")
        if isinstance(o, float):
            return np.full(64, f32_bits(o), np.uint32)
        return np.full(64, int(o) & 0xFFFFFFFF, np.uint32)

    def rd_f(self, o):
        return self.rd_vec(o).view(np.float32)

    def rd_tuple(self, o, hazard=True):
        assert isinstance(o, Reg) and o.kind in "va"
        for i in range(o.n):
            self._chk(o.kind, o.idx + i, "read")
            if hazard:
                self._mfma_hazard(o.kind, o.idx + i, "read")
        return self._file(o.kind)[o.idx:o.idx + o.n]

    def wr_vec(self, o, val, valu=True):
        assert isinstance(o, Reg) and o.n == 1 and o.kind in "va", o
        self._chk(o.kind, o.idx, "write")
        self._mfma_hazard(o.kind, o.idx, "write")
        if valu:
            t = self.mfma_c.get((o.kind, o.idx))
            if t is not None and self.issue - t < MFMA_SRCC_WAR:
                raise SimError(f"wave {self.wid} pc {self.pc}: VALU write of {o} {self.issue - t} states after an MFMA read it as srcC")
            self.valu_w[(o.kind, o.idx)] = self.issue
        self._file(o.kind)[o.idx] = np.asarray(val).view(np.uint32)

    def _mfma_hazard(self, kind, idx, what):
        t = self.mfma_w.get((kind, idx))
        if t is not None and self.issue - t < MFMA_WAIT:
            raise SimError(f"wave {self.wid} pc {self.pc}: {what} of {kind}{idx} only {self.issue - t} wait states after the MFMA that wrote it")

    def rs(self, o):
        if isinstance(o, Reg):
            assert o.kind == "s" and o.n == 1, o
            self._chk("s", o.idx, "read")
            return int(self.s[o.idx])
        if isinstance(o, Special):
            if o.name == "m0":
                return int(self.m0)
            if o.name == "scc":
                return int(self.scc)
            raise SimError(f"scalar read of {o}")
        if isinstance(o, float):
            return f32_bits(o)
        return int(o) & 0xFFFFFFFF

    def rs64(self, o):
        if isinstance(o, Special) and o.name == "vcc":
            return int(sum(1 << i for i in range(64) if self.vcc[i]))
        assert isinstance(o, Reg) and o.kind == "s" and o.n == 2
        return self.rs(o[0]) | (self.rs(o[1]) << 32)

    def ws(self, o, val):
        if isinstance(o, Special):
            if o.name == "m0":
                self.m0 = int(val) & 0xFFFFFFFF
                self.m0_set_at = self.issue
                return
            raise SimError(f"scalar write of {o}")
        assert isinstance(o, Reg) and o.kind == "s" and o.n == 1, o
        self._chk("s", o.idx, "write")
        self.s[o.idx] = int(val) & 0xFFFFFFFF

    def ws64(self, o, val):
        if isinstance(o, Special) and o.name == "vcc":
            self.vcc = np.array([(int(val) >> i) & 1 for i in range(64)], bool)
            return
        self.ws(o[0], val & 0xFFFFFFFF)
        self.ws(o[1], (val >> 32) & 0xFFFFFFFF)

    # ---- counters --------------------------------------------------------------------------------------------
    def wait(self, vmcnt=None, lgkmcnt=None):
        if vmcnt is not None:
            while len(self.vmq) > vmcnt:
                fn, regs = self.vmq.pop(0)
                for r in regs:
                    self.pending.pop(r, None)
                if fn:
                    fn()
        if lgkmcnt is not None:
            if lgkmcnt > 0 and any(k == "smem" for k, _, _ in self.lgq):
                raise SimError(f"wave {self.wid} pc {self.pc}: counted lgkmcnt({lgkmcnt}) with a scalar load outstanding (SMEM returns out of order)")
            while len(self.lgq) > lgkmcnt:
                _, fn, regs = self.lgq.pop(0)
                for r in regs:
                    self.pending.pop(r, None)
                if fn:
                    fn()


class Workgroup:
    def __init__(self, prog, mem, n_waves=4, lds_bytes=160 * 1024, late_dma=True, reverse=False, max_steps=2_000_000):
        self.ins = prog.ins if hasattr(prog, "ins") else list(prog)
        self.labels = {i.ops[0]: n for n, i in enumerate(self.ins) if i.op == "label"}
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.lds[:] = 0xFF      # NaN patterns: uninitialised LDS is conspicuous
        self.waves = [Wave(self, w) for w in range(n_waves)]
        self.late_dma, self.reverse, self.max_steps = late_dma, reverse, max_steps
        self.trace = None

    def run(self):
        steps = 0
        while True:
            order = list(reversed(self.waves)) if self.reverse else self.waves
            progressed = False
            for w in order:
                while not w.done and not w.at_barrier:
                    self.step(w)
                    progressed = True
                    steps += 1
                    if steps > self.max_steps:
                        raise SimError("step limit exceeded (runaway loop?)")
            live = [w for w in self.waves if not w.done]
            if not live:
                return
            if all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                continue
            if not progressed:
                raise SimError("deadlock: some waves at a barrier, others finished the interval without one")

    # ---- one instruction --------------------------------------------------------------------------------------
    def step(self, w):
        if w.pc >= len(self.ins):
            w.done = True
            # everything still in flight lands
            w.wait(vmcnt=0, lgkmcnt=0)
            return
        i = self.ins[w.pc]
        w.pc += 1
        op = i.op
        if op in ("label", "p2align"):
            return
        w.stats[op] = w.stats.get(op, 0) + 1
        h = getattr(self, "op_" + op, None)
        if h is None:
            h = self._generic(op)
        h(w, i)
        w.issue += 1

    def _generic(self, op):
        if op in VALU2:
            return lambda w, i: self._valu2(w, i, VALU2[op])
        if op in VALU1:
            return lambda w, i: self._valu1(w, i, VALU1[op])
        if op in VALU3:
            return lambda w, i: self._valu3(w, i, VALU3[op])
        if op in VCMP:
            return lambda w, i: self._vcmp(w, i, VCMP[op])
        if op in SALU2:
            return lambda w, i: self._salu2(w, i, SALU2[op])
        if op in SCMP:
            return lambda w, i: self._scmp(w, i, SCMP[op])
        raise SimError(f"unimplemented instruction {op}")

    @staticmethod
    def _valu1(w, i, f):
        w.wr_vec(i.ops[0], f(w.rd_vec(i.ops[1])))

    @staticmethod
    def _valu2(w, i, f):
        w.wr_vec(i.ops[0], f(w.rd_vec(i.ops[1]), w.rd_vec(i.ops[2])))

    @staticmethod
    def _valu3(w, i, f):
        w.wr_vec(i.ops[0], f(w.rd_vec(i.ops[1]), w.rd_vec(i.ops[2]), w.rd_vec(i.ops[3])))

    @staticmethod
    def _vcmp(w, i, f):
        r = f(w.rd_vec(i.ops[1]), w.rd_vec(i.ops[2]))
        d = i.ops[0]
        if isinstance(d, Special) and d.name == "vcc":
            w.vcc = np.asarray(r, bool)
        else:
            w.ws64(d, int(sum(1 << k for k in range(64) if r[k])))

    @staticmethod
    def _salu2(w, i, f):
        val, scc = f(w.rs(i.ops[1]), w.rs(i.ops[2]), int(w.scc))
        w.ws(i.ops[0], val)
        if scc is not None:
            w.scc = bool(scc)

    @staticmethod
    def _scmp(w, i, f):
        w.scc = bool(f(w.rs(i.ops[0]), w.rs(i.ops[1])))

    # ---- scalar ------------------------------------------------------------------------------------------------
    def op_s_mov_b32(self, w, i):
        w.ws(i.ops[0], w.rs(i.ops[1]))

    def op_s_mov_b64(self, w, i):
        src = i.ops[1]
        val = w.rs64(src) if isinstance(src, (Reg, Special)) else int(src) & 0xFFFFFFFFFFFFFFFF
        w.ws64(i.ops[0], val)

    def op_s_cselect_b32(self, w, i):
        w.ws(i.ops[0], w.rs(i.ops[1]) if w.scc else w.rs(i.ops[2]))

    def op_s_or_b64(self, w, i):
        val = w.rs64(i.ops[1]) | w.rs64(i.ops[2])
        w.ws64(i.ops[0], val)
        w.scc = val != 0

    def op_s_and_b64(self, w, i):
        val = w.rs64(i.ops[1]) & w.rs64(i.ops[2])
        w.ws64(i.ops[0], val)
        w.scc = val != 0

    def op_s_nop(self, w, i):
        w.issue += int(i.ops[0])

    def op_s_setprio(self, w, i):
        pass

    def op_s_waitcnt(self, w, i):
        w.wait(i.mods.get("vmcnt"), i.mods.get("lgkmcnt"))

    def op_s_barrier(self, w, i):
        w.at_barrier = True

    def op_s_endpgm(self, w, i):
        w.done = True
        w.wait(vmcnt=0, lgkmcnt=0)

    def _branch(self, w, i, cond):
        if cond:
            w.pc = self.labels[i.ops[0]]

    def op_s_branch(self, w, i):
        self._branch(w, i, True)

    def op_s_cbranch_scc0(self, w, i):
        self._branch(w, i, not w.scc)

    def op_s_cbranch_scc1(self, w, i):
        self._branch(w, i, w.scc)

    def op_s_cbranch_vccz(self, w, i):
        self._branch(w, i, not w.vcc.any())

    def op_s_cbranch_vccnz(self, w, i):
        self._branch(w, i, w.vcc.any())

    def _s_load(self, w, i, n):
        dst, base, off = i.ops[0], i.ops[1], i.ops[2]
        addr = w.rs64(base) + (w.rs(off) if isinstance(off, Reg) else int(off)) + int(i.mods.get("offset", 0))
        regs = [("s", dst.idx + k) for k in range(n)]
        for r in regs:
            w._chk(*r, "write")
            w.pending[r] = f"s_load pc {w.pc}"

        def done(addr=addr):
            data = self.mem.read(addr, 4 * n).view(np.uint32)
            for k in range(n):
                w.s[dst.idx + k] = data[k]
        w.lgq.append(("smem", done, regs))

    def op_s_memtime(self, w, i):
        dst = i.ops[0]
        regs = [("s", dst.idx), ("s", dst.idx + 1)]
        for r in regs:
            w.pending[r] = f"s_memtime pc {w.pc}"
        now = w.issue

        def done():
            w.s[dst.idx], w.s[dst.idx + 1] = now & 0xFFFFFFFF, 0
        w.lgq.append(("smem", done, regs))

    def op_s_load_dword(self, w, i):
        self._s_load(w, i, 1)

    def op_s_load_dwordx2(self, w, i):
        self._s_load(w, i, 2)

    def op_s_load_dwordx4(self, w, i):
        self._s_load(w, i, 4)

    def op_s_load_dwordx16(self, w, i):
        self._s_load(w, i, 16)

    def op_s_bitcmp1_b32(self, w, i):
        w.scc = bool((w.rs(i.ops[0]) >> (w.rs(i.ops[1]) & 31)) & 1)

    # ---- vector: special cases -----------------------------------------------------------------------------------
    def op_v_cndmask_b32(self, w, i):
        sel = i.ops[3]
        mask = w.vcc if (isinstance(sel, Special) and sel.name == "vcc") else np.array([(w.rs64(sel) >> k) & 1 for k in range(64)], bool)
        w.wr_vec(i.ops[0], np.where(mask, w.rd_vec(i.ops[2]), w.rd_vec(i.ops[1])))

    def op_v_readfirstlane_b32(self, w, i):
        src = i.ops[1]
        t = w.valu_w.get((src.kind, src.idx))
        if t is not None and w.issue - t < 1:
            raise SimError(f"wave {w.wid} pc {w.pc}: v_readfirstlane right behind the VALU write of its source")
        w.ws(i.ops[0], int(w.rd_vec(src)[0]))

    def op_v_mbcnt_lo_u32_b32(self, w, i):
        mask, add = w.rs(i.ops[1]) if not isinstance(i.ops[1], int) else int(i.ops[1]) & 0xFFFFFFFF, w.rd_vec(i.ops[2])
        lanes = np.arange(64)
        cnt = np.array([bin(mask & ((1 << min(l, 32)) - 1)).count("1") for l in lanes], np.uint32)
        w.wr_vec(i.ops[0], cnt + add)

    def op_v_mbcnt_hi_u32_b32(self, w, i):
        mask, add = int(i.ops[1]) & 0xFFFFFFFF, w.rd_vec(i.ops[2])
        lanes = np.arange(64)
        cnt = np.array([bin(mask & ((1 << max(l - 32, 0)) - 1)).count("1") for l in lanes], np.uint32)
        w.wr_vec(i.ops[0], cnt + add)

    def op_v_accvgpr_write_b32(self, w, i):
        w.wr_vec(i.ops[0], w.rd_vec(i.ops[1]))

    def op_v_accvgpr_read_b32(self, w, i):
        w.wr_vec(i.ops[0], w.rd_vec(i.ops[1]))

    def op_v_permlane32_swap_b32(self, w, i):
        d, s = i.ops[0], i.ops[1]
        for o in (d, s):
            t = w.valu_w.get((o.kind, o.idx))
            if t is not None and w.issue - t < 2:
                raise SimError(f"wave {w.wid} pc {w.pc}: v_permlane32_swap {w.issue - t} states after a VALU write of {o} (needs 2)")
        a, b = w.rd_vec(d).copy(), w.rd_vec(s).copy()
        a2, b2 = a.copy(), b.copy()
        a2[32:], b2[:32] = b[:32], a[32:]
        w.wr_vec(d, a2)
        w.wr_vec(s, b2)

    def op_v_cvt_pk_bf16_f32(self, w, i):
        lo, hi = f32_to_bf16(w.rd_f(i.ops[1])), f32_to_bf16(w.rd_f(i.ops[2]))
        w.wr_vec(i.ops[0], lo.astype(np.uint32) | (hi.astype(np.uint32) << 16))

    def op_v_cvt_pk_f16_f32(self, w, i):
        with np.errstate(over="ignore"):
            lo, hi = w.rd_f(i.ops[1]).astype(np.float16).view(np.uint16), w.rd_f(i.ops[2]).astype(np.float16).view(np.uint16)
        w.wr_vec(i.ops[0], lo.astype(np.uint32) | (hi.astype(np.uint32) << 16))

    def op_v_pk_mul_f16(self, w, i):
        a, b = w.rd_vec(i.ops[1]), w.rd_vec(i.ops[2])
        out = np.zeros(64, np.uint32)
        for sh in (0, 16):
            x = ((a >> sh) & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
            y = ((b >> sh) & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
            with np.errstate(over="ignore"):
                out |= (x * y).astype(np.float16).view(np.uint16).astype(np.uint32) << sh
        w.wr_vec(i.ops[0], out)

    def _mfma(self, w, i, conv):
        d, a, b, c = i.ops
        assert d.n == 16 and a.n == 4 and b.n == 4
        # operands written by VALU need two wait states; D of another MFMA as A/B needs the full MFMA_WAIT
        for o in (a, b):
            for k, idx in o.regs():
                w._chk(k, idx, "MFMA read")
                t = w.valu_w.get((k, idx))
                if t is not None and w.issue - t < VALU_MFMA_WAIT:
                    raise SimError(f"wave {w.wid} pc {w.pc}: MFMA reads {k}{idx} {w.issue - t} states after its VALU write")
                w._mfma_hazard(k, idx, "MFMA A/B read")
        am = conv(w._file(a.kind)[a.idx:a.idx + 4])      # [32 rows i][16 k]
        bm = conv(w._file(b.kind)[b.idx:b.idx + 4])      # [32 cols j][16 k]
        if isinstance(c, Reg):
            assert c.n == 16
            for k, idx in c.regs():
                w._chk(k, idx, "MFMA read")
                t = w.valu_w.get((k, idx))
                if t is not None and w.issue - t < VALU_MFMA_WAIT:
                    raise SimError(f"wave {w.wid} pc {w.pc}: MFMA reads srcC {k}{idx} {w.issue - t} states after its VALU write")
                if not (c == d):
                    w._mfma_hazard(k, idx, "MFMA srcC read (different tuple)")
                w.mfma_c[(k, idx)] = w.issue
            cm = w._file(c.kind)[c.idx:c.idx + 16].view(np.float32)     # [16 regs][64 lanes]
        else:
            assert c == 0
            cm = np.zeros((16, 64), np.float32)
        prod = (am.astype(np.float64) @ bm.astype(np.float64).T)      # [i][j]
        lanes = np.arange(64)
        ii = _CD_I[:, None] + 4 * (lanes[None, :] >> 5)              # [16][64]
        jj = np.broadcast_to(lanes[None, :] & 31, (16, 64))
        res = (cm.astype(np.float64) + prod[ii, jj]).astype(np.float32)
        for k, idx in d.regs():
            w._chk(k, idx, "MFMA write")
        w._file(d.kind)[d.idx:d.idx + 16] = res.view(np.uint32)
        for r in d.regs():
            w.mfma_w[r] = w.issue
            w.valu_w.pop(r, None)

    @staticmethod
    def _frag(regs4, to_f32):
        # regs4 [4][64] uint32 -> [32][16]: row = lane % 32, k = 8 (lane // 32) + element
        h = np.ascontiguousarray(regs4.T).view(np.uint16).reshape(64, 8)         # [lane][8 halves]
        f = to_f32(h)
        return f.reshape(2, 32, 8).transpose(1, 0, 2).reshape(32, 16)

    def op_v_mfma_f32_32x32x16_bf16(self, w, i):
        self._mfma(w, i, lambda r: self._frag(r, bf16_to_f32))

    def op_v_mfma_f32_32x32x16_f16(self, w, i):
        self._mfma(w, i, lambda r: self._frag(r, lambda h: h.view(np.float16).astype(np.float32)))

    # ---- LDS ---------------------------------------------------------------------------------------------------------
    def _ds_read(self, w, i, nbytes, gather):
        dst, addr = i.ops[0], i.ops[1]
        a = w.rd_vec(addr).astype(np.int64) + int(i.mods.get("offset", 0))
        if (a < 0).any() or (a + nbytes > len(self.lds)).any():
            raise SimError(f"wave {w.wid} pc {w.pc}: LDS read out of range")
        if (a % min(nbytes, 16) != 0).any() and nbytes != 8:
            raise SimError(f"wave {w.wid} pc {w.pc}: misaligned {nbytes}-byte LDS read")
        data = gather(a)                 # [n regs][64] uint32, LDS content at ISSUE time
        regs = dst.regs()
        for r in regs:
            w._chk(*r, "write")
            w._mfma_hazard(r[0], r[1], "LDS-load write")
            w.pending[r] = f"{i.op} pc {w.pc}"

        def done(data=data):
            w._file(dst.kind)[dst.idx:dst.idx + dst.n] = data
        w.lgq.append(("lds", done, regs))

    def op_ds_read_b128(self, w, i):
        def g(a):
            idx = a[:, None] + np.arange(16)[None, :]
            return np.ascontiguousarray(self.lds[idx]).view(np.uint32).T.copy()
        self._ds_read(w, i, 16, g)

    def op_ds_read_b32(self, w, i):
        def g(a):
            idx = a[:, None] + np.arange(4)[None, :]
            return np.ascontiguousarray(self.lds[idx]).view(np.uint32).T.copy()
        self._ds_read(w, i, 4, g)

    def op_ds_read_b64_tr_b16(self, w, i):
        # per 16-lane group: lane 4j + c supplies the address of row j, elements 4c .. 4c+3 (8 bytes) of a [4][16] b16
        # block; lane n of the group receives column n: (M[0][n], M[1][n], M[2][n], M[3][n])
        def g(a):
            idx = a[:, None] + np.arange(8)[None, :]
            raw = np.ascontiguousarray(self.lds[idx]).view(np.uint16).reshape(4, 4, 4, 4)   # [group][j][c][e]
            m = raw.reshape(4, 4, 16)                                                       # [group][j][col]
            out = m.transpose(0, 2, 1).reshape(64, 4)                                       # [lane][4 b16]
            return np.ascontiguousarray(out).view(np.uint32).T.copy()                       # [2][64]
        self._ds_read(w, i, 8, g)

    def op_ds_write_b128(self, w, i):
        addr, src = i.ops[0], i.ops[1]
        a = w.rd_vec(addr).astype(np.int64) + int(i.mods.get("offset", 0))
        data = np.ascontiguousarray(w.rd_tuple(src).T).view(np.uint8).reshape(64, 16)
        for l in range(64):
            self.lds[a[l]:a[l] + 16] = data[l]
        w.lgq.append(("lds", None, []))

    def op_ds_write_b64(self, w, i):
        addr, src = i.ops[0], i.ops[1]
        a = w.rd_vec(addr).astype(np.int64) + int(i.mods.get("offset", 0))
        if (a % 8 != 0).any():
            raise SimError(f"wave {w.wid} pc {w.pc}: misaligned ds_write_b64")
        data = np.ascontiguousarray(w.rd_tuple(src).T).view(np.uint8).reshape(64, 8)
        for l in range(64):
            self.lds[a[l]:a[l] + 8] = data[l]
        w.lgq.append(("lds", None, []))

    def op_ds_write_b32(self, w, i):
        addr, src = i.ops[0], i.ops[1]
        a = w.rd_vec(addr).astype(np.int64) + int(i.mods.get("offset", 0))
        data = np.ascontiguousarray(w.rd_vec(src)).view(np.uint8).reshape(64, 4)
        for l in range(64):
            self.lds[a[l]:a[l] + 4] = data[l]
        w.lgq.append(("lds", None, []))

    # ---- buffer / global -----------------------------------------------------------------------------------------------
    def _desc(self, w, rs):
        assert rs.kind == "s" and rs.n == 4 and rs.idx % 4 == 0, rs
        w0, w1, w2, w3 = (w.rs(rs[k]) for k in range(4))
        base = w0 | ((w1 & 0xFFFF) << 32)
        stride = (w1 >> 16) & 0x3FFF
        if stride != 0:
            raise SimError("only raw buffers (stride 0) are modelled")
        if w3 != 0x00020000:
            raise SimError(f"buffer descriptor word 3 = {w3:#x} (expected 0x00020000)")
        return base, w2

    def op_buffer_load_dwordx4(self, w, i):
        dst_or_off = i.ops
        if i.mods.get("lds"):
            voff, rs, soff = i.ops
            if w.issue - getattr(w, "m0_set_at", -9) < 2:      # one wait state between the SALU write of M0 and its use
                raise SimError(f"wave {w.wid} pc {w.pc}: LDS-DMA right behind the SALU write of M0")
            base, nrec = self._desc(w, rs)
            so = w.rs(soff)          # soffset takes part in the range check (measured on gfx950: tools/probes/lds_dma_oob_probe.hip)
            off = w.rd_vec(voff).astype(np.int64) + int(i.mods.get("offset", 0)) + so
            lds_base = int(w.m0) + int(i.mods.get("offset", 0))
            if lds_base % 16 or lds_base + 1024 > len(self.lds):
                raise SimError(f"wave {w.wid} pc {w.pc}: LDS-DMA destination {lds_base:#x}")
            inr = off + 16 <= nrec

            def land(off=off.copy(), inr=inr.copy(), base=base, lds_base=lds_base):
                for l in range(64):
                    dst = lds_base + 16 * l
                    if inr[l]:
                        self.lds[dst:dst + 16] = self.mem.read(base + off[l], 16)
                    else:
                        self.lds[dst:dst + 16] = 0
            if self.late_dma:
                w.vmq.append([land, []])
            else:
                land()
                w.vmq.append([None, []])
            return
        dst, voff, rs, soff = i.ops
        base, nrec = self._desc(w, rs)
        off = w.rd_vec(voff).astype(np.int64) + int(i.mods.get("offset", 0)) + w.rs(soff)
        inr = off + 16 <= nrec
        regs = dst.regs()
        for r in regs:
            w._chk(*r, "write")
            w.pending[r] = f"buffer_load pc {w.pc}"

        def done(off=off.copy(), inr=inr.copy()):
            out = np.zeros((64, 4), np.uint32)
            for l in range(64):
                if inr[l]:
                    out[l] = self.mem.read(base + off[l], 16).view(np.uint32)
            w._file(dst.kind)[dst.idx:dst.idx + 4] = out.T
        w.vmq.append([done, regs])

    def op_global_load_dwordx4(self, w, i):
        dst, voff, sbase = i.ops
        base = w.rs64(sbase)
        off = w.rd_vec(voff).astype(np.int64) + int(i.mods.get("offset", 0))
        regs = dst.regs()
        for r in regs:
            w._chk(*r, "write")
            w._mfma_hazard(r[0], r[1], "load write")
            w.pending[r] = f"global_load pc {w.pc}"

        def done(off=off.copy()):
            out = np.zeros((64, 4), np.uint32)
            for l in range(64):
                out[l] = self.mem.read(base + off[l], 16).view(np.uint32)
            w._file(dst.kind)[dst.idx:dst.idx + 4] = out.T
        w.vmq.append([done, regs])

    def _buffer_store(self, w, i, n):
        src, voff, rs, soff = i.ops
        base, nrec = self._desc(w, rs)
        off = w.rd_vec(voff).astype(np.int64) + int(i.mods.get("offset", 0)) + w.rs(soff)
        data = np.ascontiguousarray(w.rd_tuple(src).T if n > 1 else w.rd_vec(src)[:, None]).view(np.uint8).reshape(64, 4 * n)
        for l in range(64):
            if off[l] + 4 * n <= nrec:
                self.mem.write(base + off[l], data[l])
        w.vmq.append([None, []])

    def op_buffer_store_dwordx4(self, w, i):
        self._buffer_store(w, i, 4)

    def op_buffer_store_dword(self, w, i):
        self._buffer_store(w, i, 1)


# ---- tables of plain ALU semantics ------------------------------------------------------------------------------------------
def _f(fn):
    def g(*xs):
        with np.errstate(all="ignore"):
            return np.asarray(fn(*[x.view(np.float32) for x in xs]), np.float32).view(np.uint32)
    return g


def _i(fn):
    def g(*xs):
        return np.asarray(fn(*[x.view(np.int32).astype(np.int64) for x in xs])).astype(np.int64).astype(np.uint32, casting="unsafe")
    return g


def _u(fn):
    def g(*xs):
        return (np.asarray(fn(*[x.astype(np.uint64) for x in xs])) & 0xFFFFFFFF).astype(np.uint32)
    return g


def _exp2(x):
    return np.exp2(x.astype(np.float64)).astype(np.float32)


VALU1 = {
    "v_mov_b32": lambda a: a.copy(),
    "v_exp_f32": _f(_exp2),
    "v_log_f32": _f(lambda x: np.log2(x.astype(np.float64)).astype(np.float32)),
    "v_rcp_f32": _f(lambda x: (1.0 / x.astype(np.float64)).astype(np.float32)),
    "v_cvt_f32_f16": lambda a: (a & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32).view(np.uint32),
    "v_cvt_f32_u32": lambda a: a.astype(np.float32).view(np.uint32),
    "v_cvt_f32_i32": lambda a: a.view(np.int32).astype(np.float32).view(np.uint32),
}
VALU2 = {
    "v_add_u32": _u(lambda a, b: a + b),
    "v_sub_u32": _u(lambda a, b: a - b + (1 << 32)),
    "v_subrev_u32": _u(lambda a, b: b - a + (1 << 32)),
    "v_mul_lo_u32": _u(lambda a, b: a * b),
    "v_mul_u32_u24": _u(lambda a, b: (a & 0xFFFFFF) * (b & 0xFFFFFF)),
    "v_lshlrev_b32": _u(lambda a, b: b << (a & 31)),
    "v_lshrrev_b32": _u(lambda a, b: b >> (a & 31)),
    "v_and_b32": _u(lambda a, b: a & b),
    "v_or_b32": _u(lambda a, b: a | b),
    "v_xor_b32": _u(lambda a, b: a ^ b),
    "v_min_u32": _u(np.minimum),
    "v_max_u32": _u(np.maximum),
    "v_min_i32": _i(np.minimum),
    "v_max_i32": _i(np.maximum),
    "v_add_f32": _f(lambda a, b: a + b),
    "v_sub_f32": _f(lambda a, b: a - b),
    "v_mul_f32": _f(lambda a, b: a * b),
    "v_max_f32": _f(np.fmax),
    "v_min_f32": _f(np.fmin),
}
VALU3 = {
    "v_max3_f32": _f(lambda a, b, c: np.fmax(np.fmax(a, b), c)),
    "v_fma_f32": _f(lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)),
    "v_mad_u32_u24": _u(lambda a, b, c: (a & 0xFFFFFF) * (b & 0xFFFFFF) + c),
    "v_lshl_add_u32": _u(lambda a, b, c: (a << (b & 31)) + c),
    "v_lshl_or_b32": _u(lambda a, b, c: (a << (b & 31)) | c),
    "v_and_or_b32": _u(lambda a, b, c: (a & b) | c),
    "v_add3_u32": _u(lambda a, b, c: a + b + c),
}


def _cf(fn):
    return lambda a, b: fn(a.view(np.float32), b.view(np.float32))


def _ci(fn):
    return lambda a, b: fn(a.view(np.int32), b.view(np.int32))


VCMP = {
    "v_cmp_gt_f32": _cf(np.greater), "v_cmp_lt_f32": _cf(np.less), "v_cmp_ge_f32": _cf(np.greater_equal),
    "v_cmp_le_f32": _cf(np.less_equal), "v_cmp_neq_f32": _cf(np.not_equal), "v_cmp_eq_f32": _cf(np.equal),
    "v_cmp_gt_i32": _ci(np.greater), "v_cmp_lt_i32": _ci(np.less), "v_cmp_ge_i32": _ci(np.greater_equal),
    "v_cmp_le_i32": _ci(np.less_equal), "v_cmp_eq_u32": np.equal, "v_cmp_ne_u32": np.not_equal,
    "v_cmp_gt_u32": np.greater, "v_cmp_lt_u32": np.less, "v_cmp_ge_u32": np.greater_equal, "v_cmp_le_u32": np.less_equal,
}


def _s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


SALU2 = {
    "s_add_u32": lambda a, b, c: ((a + b) & 0xFFFFFFFF, (a + b) >> 32),
    "s_addc_u32": lambda a, b, c: ((a + b + c) & 0xFFFFFFFF, (a + b + c) >> 32),
    "s_sub_u32": lambda a, b, c: ((a - b) & 0xFFFFFFFF, int(b > a)),
    "s_subb_u32": lambda a, b, c: ((a - b - c) & 0xFFFFFFFF, int(b + c > a)),
    "s_add_i32": lambda a, b, c: ((a + b) & 0xFFFFFFFF, int(not -(1 << 31) <= _s32(a) + _s32(b) < (1 << 31))),
    "s_sub_i32": lambda a, b, c: ((a - b) & 0xFFFFFFFF, int(not -(1 << 31) <= _s32(a) - _s32(b) < (1 << 31))),
    "s_mul_i32": lambda a, b, c: ((a * b) & 0xFFFFFFFF, None),
    "s_mul_hi_u32": lambda a, b, c: ((a * b) >> 32, None),
    "s_lshl_b32": lambda a, b, c: ((a << (b & 31)) & 0xFFFFFFFF, int(((a << (b & 31)) & 0xFFFFFFFF) != 0)),
    "s_lshr_b32": lambda a, b, c: (a >> (b & 31), int((a >> (b & 31)) != 0)),
    "s_and_b32": lambda a, b, c: (a & b, int((a & b) != 0)),
    "s_or_b32": lambda a, b, c: (a | b, int((a | b) != 0)),
    "s_xor_b32": lambda a, b, c: (a ^ b, int((a ^ b) != 0)),
    "s_min_u32": lambda a, b, c: (min(a, b), int(a <= b)),
    "s_max_u32": lambda a, b, c: (max(a, b), int(a >= b)),
    "s_min_i32": lambda a, b, c: (min(_s32(a), _s32(b)) & 0xFFFFFFFF, int(_s32(a) <= _s32(b))),
    "s_max_i32": lambda a, b, c: (max(_s32(a), _s32(b)) & 0xFFFFFFFF, int(_s32(a) >= _s32(b))),
}
SCMP = {
    "s_cmp_lt_u32": lambda a, b: a < b, "s_cmp_le_u32": lambda a, b: a <= b, "s_cmp_gt_u32": lambda a, b: a > b,
    "s_cmp_ge_u32": lambda a, b: a >= b, "s_cmp_eq_u32": lambda a, b: a == b, "s_cmp_lg_u32": lambda a, b: a != b,
    "s_cmp_lt_i32": lambda a, b: _s32(a) < _s32(b), "s_cmp_le_i32": lambda a, b: _s32(a) <= _s32(b),
    "s_cmp_gt_i32": lambda a, b: _s32(a) > _s32(b), "s_cmp_ge_i32": lambda a, b: _s32(a) >= _s32(b),
    "s_cmp_eq_i32": lambda a, b: a == b, "s_cmp_lg_i32": lambda a, b: a != b,
}
