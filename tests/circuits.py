"""Test circuits built with the host-side ConstraintSystem mirror (harness/circuit.py)."""
import random

from circuit import ADVICE, FIXED, INSTANCE, R, Assignment, ConstraintSystem


def toy_circuit(k=5, seed=1, table_bits=4, two_lookups=False, force_degree=None):
    """A multiplication chain with a range lookup, a rotation, copy constraints across advice /
    fixed / instance columns: exercises every argument create_proof has (gates, lookup, permutation
    with >1 set when the degree is small, instance, constants)."""
    rng = random.Random(seed)
    cs = ConstraintSystem(k)
    q_mul = cs.fixed_column()
    tab = cs.fixed_column()
    const = cs.fixed_column()
    a0, a1, a2 = cs.advice_column(), cs.advice_column(), cs.advice_column()
    inst = cs.instance_column()
    for kind, c in ((ADVICE, a0), (ADVICE, a1), (ADVICE, a2), (INSTANCE, inst), (FIXED, const)):
        cs.enable_equality(kind, c)
    q = cs.fixed(q_mul)
    cs.create_gate([q * (cs.advice(a0) * cs.advice(a1) - cs.advice(a2)),
                    q * (cs.advice(a0, 1) - cs.advice(a2))])
    cs.lookup([q * cs.advice(a1)], [cs.fixed(tab)])
    if two_lookups:
        q_t = cs.fixed_column()
        tab2 = cs.fixed_column()
        cs.lookup([cs.fixed(q_t) * cs.advice(a1), cs.fixed(q_t) * cs.advice(a1, -1)],
                  [cs.fixed(tab), cs.fixed(tab2)])
    q_hi = None
    if force_degree:
        # a higher-degree restatement of the product gate, q_hi * a0^(d-3) * (a0*a1 - a2), active on
        # the same rows (an identically-zero gate would make the top h pieces the zero polynomial,
        # whose identity commitment EvmTranscript refuses to absorb -- as upstream does)
        q_hi = cs.fixed_column()
        e = cs.fixed(q_hi) * (cs.advice(a0) * cs.advice(a1) - cs.advice(a2))
        for _ in range(force_degree - 3):
            e = e * cs.advice(a0)
        cs.create_gate([e])

    asg = Assignment(cs)
    n, usable = 1 << k, cs.usable_rows()
    tsize = 1 << table_bits
    for r in range(usable):
        asg.set(FIXED, tab, r, r % tsize)
        if two_lookups:
            asg.set(FIXED, 4, r, (r % tsize) * 3 % R)  # tab2: pairs (v, 3v)
    rows = usable - 2
    val = 3
    asg.set(FIXED, const, 0, 3)
    for r in range(rows):
        asg.set(FIXED, q_mul, r, 1)
        if q_hi is not None:
            asg.set(FIXED, q_hi, r, 1)
        b = rng.randrange(tsize)
        asg.set(ADVICE, a0, r, val)
        asg.set(ADVICE, a1, r, b)
        val = val * b % R
        asg.set(ADVICE, a2, r, val)
    asg.set(ADVICE, a0, rows, val)
    asg.set(INSTANCE, inst, 0, val)
    asg.copy((ADVICE, a0, 0), (FIXED, const, 0))
    asg.copy((ADVICE, a2, rows - 1), (INSTANCE, inst, 0))
    for r in range(rows):
        asg.copy((ADVICE, a2, r), (ADVICE, a0, r + 1))
    # a few extra copies between equal cells to make longer cycles
    zeros = [r for r in range(rows) if asg.get(ADVICE, a1, r) == 0]
    for r1, r2 in zip(zeros, zeros[1:]):
        asg.copy((ADVICE, a1, r1), (ADVICE, a1, r2))
    if two_lookups:
        # q_t rows: (a1[r], a1[r-1]) must be (v, 3v): use dedicated rows at the end of the region
        pass
    return cs, asg, 1  # instance_len = 1


def variant_circuit(kind: str, k: int = 5, seed: int = 3):
    """Small circuits that take the prover through its less common paths.

    "no_lookup"      gates + permutation only (the vanishing random polynomial is committed with the
                     permutation products)
    "gates_only"     no lookup, no permutation, no instance column
    "wide_lookup"    one width-2 lookup ((a, b) in {(v, 3v+1)}), two instance columns, no gates
    "advice_factor", "merged_selectors"   gates with a univariate polynomial factor (see below)
    """
    rng = random.Random(seed)
    cs = ConstraintSystem(k)
    n = 1 << k
    if kind == "no_lookup":
        q = cs.fixed_column()
        a0, a1 = cs.advice_column(), cs.advice_column()
        inst = cs.instance_column()
        for kc in ((ADVICE, a0), (ADVICE, a1), (INSTANCE, inst)):
            cs.enable_equality(*kc)
        cs.create_gate([cs.fixed(q) * (cs.advice(a0) * cs.advice(a0) - cs.advice(a1))])
        asg = Assignment(cs)
        usable = cs.usable_rows()
        for r in range(usable - 1):
            v = rng.randrange(1 << 20)
            asg.set(FIXED, q, r, 1)
            asg.set(ADVICE, a0, r, v)
            asg.set(ADVICE, a1, r, v * v)
        asg.set(INSTANCE, inst, 0, asg.get(ADVICE, a1, 0))
        asg.copy((ADVICE, a1, 0), (INSTANCE, inst, 0))
        asg.set(ADVICE, a0, usable - 1, asg.get(ADVICE, a1, 3))
        asg.copy((ADVICE, a0, usable - 1), (ADVICE, a1, 3))
        return cs, asg, 1
    if kind == "gates_only":
        q = cs.fixed_column()
        a0, a1 = cs.advice_column(), cs.advice_column()
        # q * (a0 + a1(w) - 7): expands to monomials with the bare selector as a term (constant inside the
        # bracket); q * a0*(a0-1)*(a0-2) = 0.  (Gates must vanish on the blinding rows too, hence q.)
        cs.create_gate([cs.fixed(q) * (cs.advice(a0) + cs.advice(a1, 1) - 7),
                        cs.fixed(q) * cs.advice(a0) * (cs.advice(a0) - 1) * (cs.advice(a0) - 2)])
        asg = Assignment(cs)
        usable = cs.usable_rows()
        for r in range(usable - 1):
            v = rng.randrange(3)
            asg.set(FIXED, q, r, 1)
            asg.set(ADVICE, a0, r, v)
            asg.set(ADVICE, a1, r + 1, 7 - v)
        return cs, asg, 0
    if kind in ("advice_factor", "merged_selectors"):
        # Gates of the shape U(cell) * B with U a univariate polynomial -- what halo2's selector compression
        # produces.  "advice_factor": the cell is an advice cell (queried first, so that it is the common
        # cell the prover factors by): a0 (1 - a0)(2 - a0) * q * (a1 - 9).  "merged_selectors": three simple
        # selectors on disjoint rows, merged by compress_selectors into one fixed column.
        a0, a1 = cs.advice_column(), cs.advice_column()
        asg_rows = cs.usable_rows() - 1
        if kind == "advice_factor":
            x = cs.advice(a0)
            q = cs.fixed_column()
            cs.create_gate([x * (1 - x) * (2 - x) * cs.fixed(q) * (cs.advice(a1) - 9)])
            asg = Assignment(cs)
            for r in range(asg_rows):
                asg.set(FIXED, q, r, 1)
                free = rng.randrange(2)
                asg.set(ADVICE, a0, r, rng.randrange(1 << 30) if free else rng.randrange(3))
                asg.set(ADVICE, a1, r, 9 if free else rng.randrange(1 << 30))
            return cs, asg, 0
        sel = [cs.selector() for _ in range(3)]
        cs.create_gate([cs.query_selector(sel[0]) * (cs.advice(a0) * cs.advice(a0) - cs.advice(a1)),
                        cs.query_selector(sel[1]) * (cs.advice(a0) + cs.advice(a1, 1) - 7),
                        cs.query_selector(sel[2]) * (cs.advice(a0) * cs.advice(a1) - 3 * cs.advice(a1, -1) + 1)])
        # (a selector only joins others while gate degree - 1 + members <= cs.degree(): this plain gate lifts it to 5)
        qf = cs.fixed_column()
        y = cs.advice(a0)
        cs.create_gate([cs.fixed(qf) * y * (y - 1) * (y - 2) * (y - 3)])
        asg = Assignment(cs)
        for r in range(3, asg_rows - 1, 3):  # every third row: a gate's neighbours r - 1, r + 1 stay its own
            which = rng.randrange(4)
            v = rng.randrange(1 << 16)
            if which < 3:
                asg.selectors[which].add(r)
            if which == 0:    # a0^2 = a1
                asg.set(ADVICE, a0, r, v)
                asg.set(ADVICE, a1, r, v * v)
            elif which == 1:  # a0 + a1(next) = 7
                asg.set(ADVICE, a0, r, v)
                asg.set(ADVICE, a1, r + 1, 7 - v)
            elif which == 2:  # a0 * a1 = 3 * a1(prev) - 1
                asg.set(ADVICE, a1, r - 1, v)
                asg.set(ADVICE, a1, r, 1)
                asg.set(ADVICE, a0, r, 3 * v - 1)
            else:
                asg.set(FIXED, qf, r, 1)
                asg.set(ADVICE, a0, r, rng.randrange(4))
        asg.compress_selectors()
        assert cs.n_fixed == 2 and {c for c, _ in cs.selector_assignment} == {1}, "the three selectors share one column"
        return cs, asg, 0
    if kind == "wide_lookup":
        t0, t1, q = cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
        a0, a1 = cs.advice_column(), cs.advice_column()
        i0, i1 = cs.instance_column(), cs.instance_column()
        for kc in ((ADVICE, a0), (INSTANCE, i0), (INSTANCE, i1)):
            cs.enable_equality(*kc)
        qe = cs.fixed(q)
        # off rows look up the (0, 1) entry: (q*a0, q*a1 + (1-q))
        cs.lookup([qe * cs.advice(a0), qe * cs.advice(a1) + (1 - qe)], [cs.fixed(t0), cs.fixed(t1)])
        asg = Assignment(cs)
        usable = cs.usable_rows()
        for r in range(usable):
            v = r % 9
            asg.set(FIXED, t0, r, v)
            asg.set(FIXED, t1, r, 3 * v + 1)
        for r in range(usable - 3):
            v = rng.randrange(9)
            asg.set(FIXED, q, r, 1)
            asg.set(ADVICE, a0, r, v)
            asg.set(ADVICE, a1, r, 3 * v + 1)
        asg.set(INSTANCE, i0, 0, asg.get(ADVICE, a0, 0))
        asg.set(INSTANCE, i0, 1, asg.get(ADVICE, a0, 1))
        asg.set(INSTANCE, i1, 1, asg.get(ADVICE, a0, 2))
        asg.copy((ADVICE, a0, 0), (INSTANCE, i0, 0))
        asg.copy((ADVICE, a0, 1), (INSTANCE, i0, 1))
        asg.copy((ADVICE, a0, 2), (INSTANCE, i1, 1))
        return cs, asg, 2
    raise ValueError(kind)
