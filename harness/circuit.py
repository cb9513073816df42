"""Host-side circuit description for the prover ABI (include/zg_halo2.h `zg_circuit`).

A small mirror of halo2's `ConstraintSystem` (halo2_proofs v2023_04_20 src/plonk/circuit.rs) and of the
permutation assembly done by keygen (src/plonk/permutation/keygen.rs): columns, cell queries, gate
polynomials, lookup arguments, equality constraints -> the flat arrays `zg_prover_create` takes.
It feeds tests and bench.py: toy circuits (tests/circuits.py), the WNN-shaped circuit (wnn_shape.py)
and the restatement of zero_g's own `WnnChip::configure` / `predict` (wnn_circuit.py over layouter.py).

Polynomials are kept expanded: {tuple(sorted query indices): coefficient mod r}.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT = (1 << 256) % R
DELTA = pow(7, 1 << 28, R)
ROOT_OF_UNITY = pow(7, (R - 1) >> 28, R)

FIXED, ADVICE, INSTANCE = 0, 1, 2
MAX_FACTORS, MAX_LOOKUP_WIDTH = 8, 4
SELECTOR_BASE = 1 << 20   # provisional query indices of selectors before compress_selectors()


def omega_for(k: int) -> int:
    return pow(ROOT_OF_UNITY, 1 << (28 - k), R)


def to_mont_array(vals) -> np.ndarray:
    """list of canonical ints -> uint64[len, 4] Montgomery limbs"""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    m = (1 << 64) - 1
    for i, v in enumerate(vals):
        x = (v % R) * MONT % R
        out[i, 0] = x & m
        out[i, 1] = (x >> 64) & m
        out[i, 2] = (x >> 128) & m
        out[i, 3] = x >> 192
    return out


class Expr:
    """Multivariate polynomial over cell queries, coefficients mod r."""

    __slots__ = ("terms",)

    def __init__(self, terms=None):
        self.terms = {k: v % R for k, v in (terms or {}).items() if v % R}

    @staticmethod
    def const(c: int) -> "Expr":
        return Expr({(): c})

    @staticmethod
    def _lift(o) -> "Expr":
        return o if isinstance(o, Expr) else Expr.const(int(o))

    def __add__(self, o):
        o = Expr._lift(o)
        t = dict(self.terms)
        for k, v in o.terms.items():
            t[k] = (t.get(k, 0) + v) % R
        return Expr(t)

    __radd__ = __add__

    def __neg__(self):
        return Expr({k: -v for k, v in self.terms.items()})

    def __sub__(self, o):
        return self + (-Expr._lift(o))

    def __rsub__(self, o):
        return Expr._lift(o) - self

    def __mul__(self, o):
        o = Expr._lift(o)
        t = {}
        for k1, v1 in self.terms.items():
            for k2, v2 in o.terms.items():
                k = tuple(sorted(k1 + k2))
                t[k] = (t.get(k, 0) + v1 * v2) % R
        return Expr(t)

    __rmul__ = __mul__

    def degree(self) -> int:
        return max((len(k) for k in self.terms), default=0)

    def eval(self, cell):
        """cell(query_index) -> int"""
        acc = 0
        for k, v in self.terms.items():
            p = v
            for q in k:
                p = p * cell(q) % R
            acc += p
        return acc % R


class _Poly(ctypes.Structure):
    _fields_ = [("first", ctypes.c_uint32), ("count", ctypes.c_uint32)]


class _Query(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("column", ctypes.c_uint32), ("rotation", ctypes.c_int32)]


class _Monomial(ctypes.Structure):
    _fields_ = [("coeff", ctypes.c_uint64 * 4), ("n_factors", ctypes.c_uint32),
                ("factors", ctypes.c_uint32 * MAX_FACTORS)]


class _Lookup(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("inputs", _Poly * MAX_LOOKUP_WIDTH),
                ("tables", _Poly * MAX_LOOKUP_WIDTH)]


class CCircuit(ctypes.Structure):
    """ctypes image of `zg_circuit`."""
    _fields_ = [
        ("k", ctypes.c_uint32), ("cs_degree", ctypes.c_uint32), ("blinding_factors", ctypes.c_uint32),
        ("n_fixed", ctypes.c_uint32), ("n_advice", ctypes.c_uint32), ("n_instance", ctypes.c_uint32),
        ("n_queries", ctypes.c_uint32), ("queries", ctypes.POINTER(_Query)),
        ("n_monomials", ctypes.c_uint32), ("monomials", ctypes.POINTER(_Monomial)),
        ("n_gates", ctypes.c_uint32), ("gates", ctypes.POINTER(_Poly)),
        ("n_lookups", ctypes.c_uint32), ("lookups", ctypes.POINTER(_Lookup)),
        ("n_perm_columns", ctypes.c_uint32), ("perm_columns", ctypes.POINTER(_Query)),
        ("n_advice_queries", ctypes.c_uint32), ("advice_queries", ctypes.POINTER(_Query)),
        ("n_fixed_queries", ctypes.c_uint32), ("fixed_queries", ctypes.POINTER(_Query)),
    ]


@dataclass
class ConstraintSystem:
    k: int
    n_fixed: int = 0
    n_advice: int = 0
    n_instance: int = 0
    queries: list = field(default_factory=list)          # (kind, column, rotation)
    gates: list = field(default_factory=list)            # Expr, creation order
    lookups: list = field(default_factory=list)          # (inputs [Expr], tables [Expr])
    perm_columns: list = field(default_factory=list)     # (kind, column)
    advice_queries: list = field(default_factory=list)   # (column, rotation) in first-query order
    fixed_queries: list = field(default_factory=list)
    instance_queries: list = field(default_factory=list)
    n_selectors: int = 0
    selector_simple: list = field(default_factory=list)   # selector id -> created by selector() (True) / complex_selector()
    selector_assignment: list = field(default_factory=list)  # after compress_selectors: id -> (fixed column, value when enabled)

    # ---- selectors (circuit.rs `selector` / `complex_selector`).  halo2 keeps selectors virtual until keygen:
    # synthesis records on which rows each is enabled, then `ConstraintSystem::compress_selectors`
    # (circuit.rs + circuit/compress_selectors.rs) turns them into fixed columns appended AFTER the
    # circuit's own fixed columns, their cell queries after every other query, and substitutes them in
    # the gates and lookups.  compress_selectors() below restates that.
    def selector(self) -> int:
        self.n_selectors += 1
        self.selector_simple.append(True)
        return self.n_selectors - 1

    def complex_selector(self) -> int:
        self.n_selectors += 1
        self.selector_simple.append(False)
        return self.n_selectors - 1

    def query_selector(self, s: int) -> "Expr":
        assert not self.selector_assignment, "selectors already compressed"
        return Expr({(SELECTOR_BASE + s,): 1})

    def _exprs(self):
        return self.gates + [x for ins, tabs in self.lookups for x in ins + tabs]

    def compress_selectors(self, activations, merge: bool = True):
        """activations[s] = set of rows selector s is enabled on.  -> list of new fixed columns' values
        (each a list of n ints), in the order their columns were allocated.

        compress_selectors.rs `process`: a selector that is complex or appears in no gate keeps a column
        of its own (0/1), allocated first, in selector order.  The simple ones are merged greedily, in
        selector order: selector j joins the combination opened by selector i when it is never enabled
        on a row where a member already is, and max(member gate degree - 1) + members <= cs.degree().  A
        combination of m selectors is ONE column holding t on the rows where its t-th member is enabled
        (t = 1..m, 0 elsewhere), and member t is replaced by  q * prod_{u != t} (u - q).
        merge=False gives every selector its own column (halo2 with compress_selectors disabled)."""
        assert not self.selector_assignment and len(activations) == self.n_selectors
        n = 1 << self.k
        max_degree = self.degree()
        degrees = [0] * self.n_selectors
        if merge:
            for g in self.gates:  # extract_simple_selector: a gate polynomial mentions at most one
                simple = {qi - SELECTOR_BASE for key in g.terms for qi in key
                          if qi >= SELECTOR_BASE and self.selector_simple[qi - SELECTOR_BASE]}
                assert len(simple) <= 1, "two simple selectors cannot be in the same expression"
                for sel in simple:
                    degrees[sel] = max(degrees[sel], g.degree())
            for e in [x for ins, tabs in self.lookups for x in ins + tabs]:
                assert not any(qi >= SELECTOR_BASE and self.selector_simple[qi - SELECTOR_BASE]
                               for key in e.terms for qi in key), "simple selector in a lookup argument"
        columns, subst = [], {}
        self.selector_assignment = [None] * self.n_selectors

        def allocate():
            col = self.fixed_column()
            return col, self.q(FIXED, col, 0)

        for s in range(self.n_selectors):
            if degrees[s] == 0:
                col, q = allocate()
                subst[SELECTOR_BASE + s] = q
                self.selector_assignment[s] = (col, 1)
                columns.append([1 if r in activations[s] else 0 for r in range(n)])
        rest = [s for s in range(self.n_selectors) if degrees[s] > 0]
        added = set()
        for i, s in enumerate(rest):
            if s in added:
                continue
            added.add(s)
            assert degrees[s] <= max_degree
            d, combination = degrees[s] - 1, [s]
            for t in rest[i + 1:]:
                if d + len(combination) == max_degree:
                    break
                if t in added or any(activations[t] & activations[u] for u in combination):
                    continue
                new_d = max(d, degrees[t] - 1)
                if new_d + len(combination) + 1 > max_degree:
                    continue
                d = new_d
                combination.append(t)
                added.add(t)
            col, q = allocate()
            values = [0] * n
            for root, member in enumerate(combination, start=1):
                e = q
                for u in range(1, len(combination) + 1):
                    if u != root:
                        e = e * (Expr.const(u) - q)
                subst[SELECTOR_BASE + member] = e
                self.selector_assignment[member] = (col, root)
                for r in activations[member]:
                    values[r] = root
            columns.append(values)

        def fix(e: "Expr") -> "Expr":
            out = Expr()
            for key, v in e.terms.items():
                m = Expr({tuple(qi for qi in key if qi < SELECTOR_BASE): v})
                for qi in key:
                    if qi >= SELECTOR_BASE:
                        m = m * subst[qi]
                out = out + m
            return out

        self.gates = [fix(g) for g in self.gates]
        self.lookups = [([fix(e) for e in ins], [fix(e) for e in tabs]) for ins, tabs in self.lookups]
        for g in self.gates:
            assert g.degree() <= MAX_FACTORS
        return columns

    # ---- columns
    def fixed_column(self) -> int:
        self.n_fixed += 1
        return self.n_fixed - 1

    def advice_column(self) -> int:
        self.n_advice += 1
        return self.n_advice - 1

    def instance_column(self) -> int:
        self.n_instance += 1
        return self.n_instance - 1

    # ---- queries (query_any_index)
    def q(self, kind: int, column: int, rotation: int = 0) -> Expr:
        key = (kind, column, rotation)
        if key not in self.queries:
            self.queries.append(key)
            lst = {FIXED: self.fixed_queries, ADVICE: self.advice_queries, INSTANCE: self.instance_queries}[kind]
            lst.append((column, rotation))
        return Expr({(self.queries.index(key),): 1})

    def fixed(self, c, rot=0):
        return self.q(FIXED, c, rot)

    def advice(self, c, rot=0):
        return self.q(ADVICE, c, rot)

    def instance(self, c, rot=0):
        return self.q(INSTANCE, c, rot)

    def create_gate(self, polys):
        for p in polys:
            assert p.degree() <= MAX_FACTORS
            self.gates.append(p)

    def lookup(self, inputs, tables):
        assert len(inputs) == len(tables) <= MAX_LOOKUP_WIDTH
        self.lookups.append((list(inputs), list(tables)))

    def enable_equality(self, kind: int, column: int):
        """permutation.add_column + query at Rotation::cur() (halo2 does both)."""
        if (kind, column) not in self.perm_columns:
            self.perm_columns.append((kind, column))
            self.q(kind, column, 0)

    # ---- derived quantities (circuit.rs)
    def degree(self) -> int:
        d = 3 if self.perm_columns else 1  # permutation::Argument::required_degree
        for ins, tabs in self.lookups:
            ind = max([1] + [e.degree() for e in ins])
            td = max([1] + [e.degree() for e in tabs])
            d = max(d, 4, 2 + ind + td)
        for g in self.gates:
            d = max(d, g.degree())
        return d

    def blinding_factors(self) -> int:
        per_col = {}
        for (c, _r) in self.advice_queries:
            per_col[c] = per_col.get(c, 0) + 1
        return max(3, max(per_col.values(), default=0)) + 2

    def extended_k(self) -> int:
        ek = self.k
        while (1 << ek) < (1 << self.k) * (self.degree() - 1):
            ek += 1
        return ek

    def usable_rows(self) -> int:
        return (1 << self.k) - (self.blinding_factors() + 1)

    # ---- flattening to the ABI structs
    def to_c(self) -> "CircuitImage":
        monos, polys = [], []

        def add_poly(e: Expr) -> _Poly:
            first = len(monos)
            for key in sorted(e.terms):
                m = _Monomial()
                x = e.terms[key] * MONT % R
                for i in range(4):
                    m.coeff[i] = (x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
                m.n_factors = len(key)
                for i, qi in enumerate(key):
                    m.factors[i] = qi
                monos.append(m)
            return _Poly(first, len(monos) - first)

        gates = [add_poly(g) for g in self.gates]
        lks = []
        for ins, tabs in self.lookups:
            lk = _Lookup()
            lk.width = len(ins)
            for i, e in enumerate(ins):
                lk.inputs[i] = add_poly(e)
            for i, e in enumerate(tabs):
                lk.tables[i] = add_poly(e)
            lks.append(lk)

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))()
            for i, it in enumerate(items):
                a[i] = it
            return a

        img = CircuitImage()
        img.queries = arr(_Query, [_Query(k, c, r) for (k, c, r) in self.queries])
        img.monomials = arr(_Monomial, monos)
        img.gates = arr(_Poly, gates)
        img.lookups = arr(_Lookup, lks)
        img.perm = arr(_Query, [_Query(k, c, 0) for (k, c) in self.perm_columns])
        img.aq = arr(_Query, [_Query(ADVICE, c, r) for (c, r) in self.advice_queries])
        img.fq = arr(_Query, [_Query(FIXED, c, r) for (c, r) in self.fixed_queries])
        c = CCircuit()
        c.k, c.cs_degree, c.blinding_factors = self.k, self.degree(), self.blinding_factors()
        c.n_fixed, c.n_advice, c.n_instance = self.n_fixed, self.n_advice, self.n_instance
        c.n_queries, c.queries = len(self.queries), ctypes.cast(img.queries, ctypes.POINTER(_Query))
        c.n_monomials, c.monomials = len(monos), ctypes.cast(img.monomials, ctypes.POINTER(_Monomial))
        c.n_gates, c.gates = len(gates), ctypes.cast(img.gates, ctypes.POINTER(_Poly))
        c.n_lookups, c.lookups = len(lks), ctypes.cast(img.lookups, ctypes.POINTER(_Lookup))
        c.n_perm_columns, c.perm_columns = len(self.perm_columns), ctypes.cast(img.perm, ctypes.POINTER(_Query))
        c.n_advice_queries, c.advice_queries = len(self.advice_queries), ctypes.cast(img.aq, ctypes.POINTER(_Query))
        c.n_fixed_queries, c.fixed_queries = len(self.fixed_queries), ctypes.cast(img.fq, ctypes.POINTER(_Query))
        img.c = c
        return img


class CircuitImage:
    """Keeps the ctypes arrays alive next to the zg_circuit that points into them."""
    c: CCircuit

    def ptr(self):
        return ctypes.byref(self.c)


class Assignment:
    """Cell values + equality constraints -> fixed / advice / instance columns and the sigma
    polynomials keygen derives (permutation::keygen::Assembly)."""

    def __init__(self, cs: ConstraintSystem):
        self.cs = cs
        self.n = 1 << cs.k
        self.fixed = [[0] * self.n for _ in range(cs.n_fixed)]
        self.advice = [[0] * self.n for _ in range(cs.n_advice)]
        self.instance = [[0] * self.n for _ in range(cs.n_instance)]
        self.selectors = [set() for _ in range(cs.n_selectors)]  # rows each (virtual) selector is enabled on
        m = len(cs.perm_columns)
        # cycle representation as in halo2: mapping[col][row] = next cell of the cycle
        self.mapping = [[(c, r) for r in range(self.n)] for c in range(m)]
        self.aux = [[(c, r) for r in range(self.n)] for c in range(m)]
        self.sizes = [[1] * self.n for _ in range(m)]

    def compress_selectors(self, merge: bool = True):
        """keygen's step after synthesis: selectors -> fixed columns (appended), gates / lookups rewritten.
        On a constraint system that is already compressed (a second synthesis of the same circuit) only
        the columns are rebuilt, from cs.selector_assignment."""
        cs = self.cs
        if not cs.selector_assignment:
            self.fixed.extend(cs.compress_selectors(self.selectors, merge))
            return
        assert len(self.fixed) == cs.n_fixed
        for s, (col, value) in enumerate(cs.selector_assignment):
            for r in self.selectors[s]:
                assert self.fixed[col][r] == 0, "selectors of one combination enabled on the same row"
                self.fixed[col][r] = value

    def col(self, kind):
        return {FIXED: self.fixed, ADVICE: self.advice, INSTANCE: self.instance}[kind]

    def set(self, kind, column, row, value):
        self.col(kind)[column][row] = value % R

    def get(self, kind, column, row):
        return self.col(kind)[column][row]

    def copy(self, a, b):
        """a, b = (kind, column, row); merges their permutation cycles (Assembly::copy)."""
        pa = self.cs.perm_columns.index((a[0], a[1]))
        pb = self.cs.perm_columns.index((b[0], b[1]))
        assert self.get(*a) == self.get(*b), (a, b, self.get(*a), self.get(*b))
        lc, lr, rc, rr = pa, a[2], pb, b[2]
        if self.aux[lc][lr] == self.aux[rc][rr]:
            return
        if self.sizes[self.aux[lc][lr][0]][self.aux[lc][lr][1]] < self.sizes[self.aux[rc][rr][0]][self.aux[rc][rr][1]]:
            lc, lr, rc, rr = rc, rr, lc, lr
        la, ra = self.aux[lc][lr], self.aux[rc][rr]
        self.sizes[la[0]][la[1]] += self.sizes[ra[0]][ra[1]]
        i, j = rc, rr
        while True:
            self.aux[i][j] = la
            i, j = self.mapping[i][j]
            if (i, j) == (rc, rr):
                break
        self.mapping[lc][lr], self.mapping[rc][rr] = self.mapping[rc][rr], self.mapping[lc][lr]

    def sigma_values(self) -> np.ndarray:
        """[n_perm][n] Lagrange values: sigma_c(omega^r) = delta^c' * omega^r' for mapping[c][r] = (c', r')."""
        m = len(self.cs.perm_columns)
        w = omega_for(self.cs.k)
        wp = [1] * self.n
        for i in range(1, self.n):
            wp[i] = wp[i - 1] * w % R
        dp = [pow(DELTA, c, R) for c in range(m)]
        out = np.empty((max(m, 1), self.n, 4), dtype=np.uint64)
        for c in range(m):
            out[c] = to_mont_array([dp[pc] * wp[pr] % R for (pc, pr) in self.mapping[c]])
        return out[:m]

    def fixed_values(self) -> np.ndarray:
        return np.stack([to_mont_array(col) for col in self.fixed]) if self.fixed else np.zeros((0, self.n, 4), np.uint64)

    def advice_values(self) -> np.ndarray:
        return np.stack([to_mont_array(col) for col in self.advice]) if self.advice else np.zeros((0, self.n, 4), np.uint64)

    def instance_values(self, length: int) -> np.ndarray:
        if not self.instance:
            return np.zeros((0, length, 4), np.uint64)
        return np.stack([to_mont_array(col[:length]) for col in self.instance])

    def check(self):
        """MockProver-style check on the usable rows: gates vanish, lookup inputs are in the table,
        copies hold (asserted at copy time)."""
        cs, n = self.cs, self.n
        usable = cs.usable_rows()

        def cell_at(row):
            def f(qi):
                kind, c, rot = cs.queries[qi]
                return self.col(kind)[c][(row + rot) % n]
            return f

        for row in range(usable):
            f = cell_at(row)
            for gi, g in enumerate(cs.gates):
                v = g.eval(f)
                assert v == 0, f"gate {gi} fails on row {row}: {v}"
        for li, (ins, tabs) in enumerate(cs.lookups):
            table = {tuple(t.eval(cell_at(r)) for t in tabs) for r in range(usable)}
            for row in range(usable):
                tup = tuple(e.eval(cell_at(row)) for e in ins)
                assert tup in table, f"lookup {li} input {tup} on row {row} not in table"
