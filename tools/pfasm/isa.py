"""A tiny gfx950 instruction IR for the hand-scheduled prefill kernel (tools/pfasm/kernel.py).

One `Ins` = one machine instruction: mnemonic, operands in assembler order (destination first) and
modifiers.  The same list is printed as assembler text (emit) and executed by the functional simulator
(tools/pfasm/sim.py), so what is tested on the CPU is what hipcc assembles.

Operands:  V(i, n) / A(i, n) / S(i, n) register (ranges), the names in SPECIAL, python ints (inline constants
or 32-bit literals), python floats (printed as hex f32 literals), Label names (strings) for branches.
"""
import struct


class Reg:
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        assert kind in "vas" and idx >= 0 and n >= 1, (kind, idx, n)
        limit = 256 if kind in "va" else 102
        assert idx + n <= limit, (kind, idx, n)
        self.kind, self.idx, self.n = kind, int(idx), int(n)

    def __repr__(self):
        return f"{self.kind}{self.idx}" if self.n == 1 else f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def __getitem__(self, i):          # sub-register i of a range
        if isinstance(i, slice):
            start, stop, _ = i.indices(self.n)
            return Reg(self.kind, self.idx + start, stop - start)
        assert 0 <= i < self.n
        return Reg(self.kind, self.idx + i, 1)

    def __eq__(self, o):
        return isinstance(o, Reg) and (self.kind, self.idx, self.n) == (o.kind, o.idx, o.n)

    def __hash__(self):
        return hash((self.kind, self.idx, self.n))

    def regs(self):
        return [(self.kind, self.idx + i) for i in range(self.n)]


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


class Special:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


VCC, EXEC, M0, SCC, OFF = (Special(n) for n in ("vcc", "exec", "m0", "scc", "off"))


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


INLINE_F = {0.5: "0.5", -0.5: "-0.5", 1.0: "1.0", -1.0: "-1.0", 2.0: "2.0", -2.0: "-2.0", 4.0: "4.0", -4.0: "-4.0"}


def fmt_operand(o):
    if isinstance(o, (Reg, Special)):
        return repr(o)
    if isinstance(o, bool):
        raise TypeError(o)
    if isinstance(o, int):
        if -16 <= o <= 64:
            return str(o)
        return hex(o & 0xFFFFFFFF)
    if isinstance(o, float):
        if o == 0.0 and struct.pack("<f", o) == b"\0\0\0\0":
            return "0"
        if o in INLINE_F:
            return INLINE_F[o]
        return hex(f32_bits(o))
    if isinstance(o, str):
        return o
    raise TypeError(o)


class Ins:
    __slots__ = ("op", "ops", "mods", "note")

    def __init__(self, op, *ops, note=None, **mods):
        self.op, self.ops, self.mods, self.note = op, list(ops), mods, note

    def text(self):
        op = self.op
        if op == "label":
            return f"{self.ops[0]}:"
        if op == "s_waitcnt":
            parts = []
            if self.mods.get("vmcnt") is not None:
                parts.append(f"vmcnt({self.mods['vmcnt']})")
            if self.mods.get("lgkmcnt") is not None:
                parts.append(f"lgkmcnt({self.mods['lgkmcnt']})")
            return "s_waitcnt " + " ".join(parts)
        if op in ("s_nop", "s_setprio", "s_sleep"):
            return f"{op} {self.ops[0]}"
        if op in ("s_barrier", "s_endpgm"):
            return op
        if op == "p2align":
            return f".p2align {self.ops[0]}"
        s = op + " " + ", ".join(fmt_operand(o) for o in self.ops)
        m = self.mods
        if op.startswith("buffer_"):
            if m.get("offen"):
                s += " offen"
            if m.get("offset"):
                s += f" offset:{m['offset']}"
            if m.get("sc0"):
                s += " sc0"
            if m.get("sc1"):
                s += " sc1"
            if m.get("nt"):
                s += " nt"
            if m.get("lds"):
                s += " lds"
        elif op.startswith(("ds_", "global_", "s_load")):
            if m.get("offset"):
                s += f" offset:{m['offset']}"
            if m.get("nt"):
                s += " nt"
        return s.rstrip()


def L(name):
    return Ins("label", name)


class Program:
    """A list of instructions with label bookkeeping."""

    def __init__(self):
        self.ins = []
        self._uniq = 0

    def emit(self, op, *ops, **mods):
        i = Ins(op, *ops, **mods)
        self.ins.append(i)
        return i

    def extend(self, items):
        self.ins.extend(items)

    def label(self, name):
        self.ins.append(L(name))

    def uniq(self, stem):
        self._uniq += 1
        return f"{stem}_{self._uniq}"

    def text(self, prefix=""):
        """Assembler text.  `prefix` is put in front of every label (inline asm in a function that may be
        instantiated more than once needs unique labels: use the %= token)."""
        labels = {i.ops[0] for i in self.ins if i.op == "label"}
        out = []
        for i in self.ins:
            t = i.text()
            if i.op == "label":
                t = f"{prefix}{i.ops[0]}:"
            elif i.op.startswith(("s_cbranch", "s_branch")):
                tgt = i.ops[0]
                assert tgt in labels, f"undefined label {tgt}"
                t = f"{i.op} {prefix}{tgt}"
            out.append(t)
        return out
