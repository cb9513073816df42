"""Witness values that are either Python integers or recorded operations.

The chips of wnn_circuit.py compute their witness with the helpers below.  On integers they evaluate what the
reference's `Value::map` closures evaluate (same formulas, canonical integers mod r).  On `Sym` operands they append
one operation to a `Tape` instead: running WnnChip::predict once on a symbolic image yields the straight-line program
"image bytes -> every advice cell", which the device executes per image (csrc/witness.hip, include/zg_halo2.h
zg_witness_*) -- SURVEY.md 8f item 2: the witness of a batch of images without a host synthesis per image.

A slot holds an unsigned integer < 2^256 (every witness value of the circuit is a small non-negative integer or a
constant; nothing wraps modulo r in an honest witness, and the tape keeps that as its contract: SUB never
underflows).  Operations, dst = f(a, b, imm):

    CONST   imm (index into the 256-bit constant pool)      PIXEL   imm = byte offset in the image
    ADD     a + b          SUB  a - b         MUL  a * b (low 256 bits)
    ADDI    a + imm        RSUBI imm - a      MULI a * imm          (imm < 2^64)
    SHRI    a >> imm       SHLI a << imm      ANDI a & imm          SHRV a >> b
    GTI     a > imm        GEI  a >= imm      EQI  a == imm         (0 / 1)
    DIVI    a // imm       TABLE  table[imm + a]  (64-bit words)
"""
from __future__ import annotations

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

OPS = ["CONST", "PIXEL", "ADD", "SUB", "MUL", "ADDI", "RSUBI", "MULI", "SHRI", "SHLI", "ANDI", "SHRV", "GTI", "GEI",
       "EQI", "DIVI", "TABLE"]
OPCODE = {name: i for i, name in enumerate(OPS)}
M64 = (1 << 64) - 1
M256 = (1 << 256) - 1


class Tape:
    def __init__(self):
        self.ops = []      # (opcode, a, b, imm); the destination is the op's own index
        self.level = []    # dependency depth of every op
        self.consts = []   # 256-bit constant pool
        self.table = []    # 64-bit words (bloom filter words, one row after the other)
        self._const_slot = {}

    def emit(self, op: str, a: int = 0, b: int = 0, imm: int = 0, deps=()) -> "Sym":
        assert 0 <= imm <= M64, "immediate does not fit 64 bits"
        self.ops.append((OPCODE[op], a, b, imm))
        self.level.append(1 + max((self.level[d] for d in deps), default=-1))
        return Sym(self, len(self.ops) - 1)

    def const(self, value: int) -> "Sym":
        value %= R
        slot = self._const_slot.get(value)
        if slot is None:
            self.consts.append(value)
            slot = self.emit("CONST", imm=len(self.consts) - 1).slot
            self._const_slot[value] = slot
        return Sym(self, slot)

    def pixel(self, offset: int) -> "Sym":
        return self.emit("PIXEL", imm=offset)

    def add_table(self, words) -> int:
        base = len(self.table)
        self.table.extend(int(w) for w in words)
        return base

    # ---- reference interpreter (tests compare the device's result with it, and it with the concrete synthesis)
    def run(self, image_bytes) -> list:
        v = [0] * len(self.ops)
        for i, (op, a, b, imm) in enumerate(self.ops):
            name = OPS[op]
            if name == "CONST":
                r = self.consts[imm]
            elif name == "PIXEL":
                r = int(image_bytes[imm])
            elif name == "ADD":
                r = v[a] + v[b]
            elif name == "SUB":
                assert v[a] >= v[b], "tape contract: SUB does not underflow"
                r = v[a] - v[b]
            elif name == "MUL":
                r = v[a] * v[b]
            elif name == "ADDI":
                r = v[a] + imm
            elif name == "RSUBI":
                assert imm >= v[a], "tape contract: RSUBI does not underflow"
                r = imm - v[a]
            elif name == "MULI":
                r = v[a] * imm
            elif name == "SHRI":
                r = v[a] >> imm
            elif name == "SHLI":
                r = v[a] << imm
            elif name == "ANDI":
                r = v[a] & imm
            elif name == "SHRV":
                r = v[a] >> v[b] if v[b] < 256 else 0
            elif name == "GTI":
                r = int(v[a] > imm)
            elif name == "GEI":
                r = int(v[a] >= imm)
            elif name == "EQI":
                r = int(v[a] == imm)
            elif name == "DIVI":
                r = v[a] // imm
            elif name == "TABLE":
                r = self.table[imm + v[a]] if imm + v[a] < len(self.table) else 0  # (zero outside the table)
            else:
                raise AssertionError(name)
            v[i] = r & M256
        return v


class Sym:
    """One recorded value (a tape slot).  Only what the chips use is overloaded."""
    __slots__ = ("tape", "slot")

    def __init__(self, tape: Tape, slot: int):
        self.tape, self.slot = tape, slot

    def _bin(self, op, other):
        if isinstance(other, Sym):
            return self.tape.emit(op, self.slot, other.slot, deps=(self.slot, other.slot))
        return None

    def __add__(self, o):
        if isinstance(o, Sym):
            return self._bin("ADD", o)
        return self if o == 0 else self.tape.emit("ADDI", self.slot, imm=int(o), deps=(self.slot,))

    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, Sym):
            return self._bin("SUB", o)
        o = int(o)
        if o == 0:
            return self
        c = self.tape.const(o)
        return self.tape.emit("SUB", self.slot, c.slot, deps=(self.slot, c.slot))

    def __rsub__(self, o):
        return self.tape.emit("RSUBI", self.slot, imm=int(o), deps=(self.slot,))

    def __mul__(self, o):
        if isinstance(o, Sym):
            return self._bin("MUL", o)
        o = int(o)
        if o == 1:
            return self
        return self.tape.emit("MULI", self.slot, imm=o, deps=(self.slot,))

    __rmul__ = __mul__

    def __floordiv__(self, o):
        return self.tape.emit("DIVI", self.slot, imm=int(o), deps=(self.slot,))

    def __mod__(self, o):
        assert o == R, "only the reduction modulo r (a no-op on honest witness values)"
        return self

    def __rshift__(self, o):
        if isinstance(o, Sym):
            return self._bin("SHRV", o)
        return self if o == 0 else self.tape.emit("SHRI", self.slot, imm=int(o), deps=(self.slot,))

    def __lshift__(self, o):
        return self if o == 0 else self.tape.emit("SHLI", self.slot, imm=int(o), deps=(self.slot,))

    def __and__(self, o):
        return self.tape.emit("ANDI", self.slot, imm=int(o), deps=(self.slot,))


def is_sym(x) -> bool:
    return isinstance(x, Sym)


def _inv(x: int) -> int:
    return pow(x, -1, R)


# ---- the helpers the chips call
def gt(x, y: int):
    """1 if x > y else 0"""
    if is_sym(x):
        return x.tape.emit("GTI", x.slot, imm=y, deps=(x.slot,))
    return 1 if x > y else 0


def ge(x, y: int):
    if is_sym(x):
        return x.tape.emit("GEI", x.slot, imm=y, deps=(x.slot,))
    return 1 if x >= y else 0


def eq(x, y: int):
    if is_sym(x):
        return x.tape.emit("EQI", x.slot, imm=y, deps=(x.slot,))
    return 1 if x == y else 0


def exact_shr(x, bits: int):
    """x * (2^bits)^-1 mod r for an x whose low `bits` bits are zero: the running-sum step z' = (z - word) / 2^K of
    the decomposition gadgets (the reference multiplies by the field inverse; on such an x that IS the shift)."""
    if is_sym(x):
        # peephole: (z - (z & m)) >> bits with m inside the low `bits` bits is z >> bits -- one level of the replay instead of
        # three on the decomposition chains' critical path (the subtraction stays on the tape where a cell shows it)
        t = x.tape
        op, a, b, _ = t.ops[x.slot]
        if OPS[op] == "SUB":
            opb, ab, _, immb = t.ops[b]
            if OPS[opb] == "ANDI" and ab == a and immb < (1 << bits):
                return Sym(t, a) >> bits
        return x >> bits
    return x * _inv(1 << bits) % R


def byte_be(word, index, num_bytes: int):
    """bytes_be[index] of a num_bytes-byte word"""
    if is_sym(word) or is_sym(index):
        t = word.tape if is_sym(word) else index.tape
        w = word if is_sym(word) else t.const(word)
        if is_sym(index):
            amount = (num_bytes - 1 - index) * 8  # RSUBI, MULI
        else:
            return (w >> (8 * (num_bytes - 1 - index))) & 0xFF
        return (w >> amount) & 0xFF
    return (word >> (8 * (num_bytes - 1 - index))) & 0xFF


def bit_of_byte(byte, index):
    """(byte >> (7 - index)) & 1"""
    if is_sym(byte) or is_sym(index):
        t = byte.tape if is_sym(byte) else index.tape
        b = byte if is_sym(byte) else t.const(byte)
        if is_sym(index):
            return (b >> (7 - index)) & 1
        return (b >> (7 - int(index))) & 1
    return (byte >> (7 - index)) & 1


def as_int_check(x, predicate, message: str):
    """assertions on concrete values only (a recorded value has none yet)"""
    if not is_sym(x):
        assert predicate(x), message
