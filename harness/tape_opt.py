"""Depth reduction of a recorded witness program (harness/witness_tape.py): serial chains become parallel prefixes.

The device replays a program level by level (csrc/witness.hip) and its time is proportional to the NUMBER OF LEVELS, not to
the number of operations (a level costs one dependent-instruction chain and a barrier whatever it holds: DESIGN.md section 7).
The chips of zero_g's circuit record two kinds of long serial chains, each link of which is a witness cell:
  * Horner accumulations  x_j = c * x_{j-1} + y_j   (bits -> number in the filter-index gadget: 28 .. 49 links, MULI + ADD each),
  * running sums          x_j = x_{j-1} + y_j       (the response accumulator: one link per filter).
Every x_j is an affine function of x_0 with constant multipliers, so all of them come out of a Kogge-Stone prefix over the maps
(m_j, y_j): round d combines map j with map j - d  --  v_j <- v_{j-d} * M_j + v_j, M_j <- M_{j-d} * M_j, the M's being plain Python
integers  --  and x_j = M_j * x_0 + v_j at the end: 2 * ceil(log2 L) + 2 levels instead of 2 L (or L), for ~ L log2 L extra operations.
The integers are the same modulo 2^256 whichever way they are added up, so every slot keeps its value for EVERY image; slots
keep their identity (cells and instance rows refer to them), only their defining operation changes.

`parallelise_chains(tape, roots)` returns a new Tape in topological order, the map old slot -> new slot, and what it did.
"""
from __future__ import annotations

from symint import M64, OPCODE, Tape

_USES_B = {OPCODE[n] for n in ("ADD", "SUB", "MUL", "SHRV")}
_ADD, _MULI, _FIRST_WITH_A = OPCODE["ADD"], OPCODE["MULI"], OPCODE["ADD"]  # (opcodes >= ADD read operand a: csrc/witness.hip)
MIN_CHAIN = 4


def _deps(op):
    code, a, b, _ = op
    out = []
    if code >= _FIRST_WITH_A:
        out.append(a)
    if code in _USES_B:
        out.append(b)
    return out


def _levels(ops):
    """dependency depth of every op (operands may sit at larger indices after a rewrite): iterative, memoised"""
    level = [-1] * len(ops)
    for root in range(len(ops)):
        if level[root] >= 0:
            continue
        stack = [root]
        while stack:
            i = stack[-1]
            pending = [d for d in _deps(ops[i]) if level[d] < 0]
            if pending:
                stack.extend(pending)
                continue
            level[i] = 1 + max((level[d] for d in _deps(ops[i])), default=-1)
            stack.pop()
    return level


def parallelise_chains(tape: Tape):
    ops = list(tape.ops)
    level = list(tape.level)
    n0 = len(ops)
    _ADDI = OPCODE["ADDI"]
    # ---- links: op i = m * pred + y, pred = the operand on the critical path (through a MULI by a constant, or directly);
    #      y = ("s", slot) for an ADD, ("i", immediate) for an ADDI
    link = {}
    for i, (code, a, b, imm) in enumerate(ops):
        if code == _ADD:
            h, y = (a, ("s", b)) if level[a] >= level[b] else (b, ("s", a))
        elif code == _ADDI:
            h, y = a, ("i", imm)
        else:
            continue
        if ops[h][0] == _MULI:
            link[i] = (ops[h][1], ops[h][3], y)  # pred, multiplier, term
        else:
            link[i] = (h, 1, y)
    # ---- chains: follow the links; a node continues its predecessor's chain when it is the first to do so
    chain_of, chains, has_child = {}, [], set()
    for i in sorted(link):
        pred = link[i][0]
        if pred in chain_of and pred not in has_child and chains[chain_of[pred]][-1] == pred:
            chain_of[i] = chain_of[pred]
            chains[chain_of[pred]].append(i)
            has_child.add(pred)
        else:
            chain_of[i] = len(chains)
            chains.append([i])
    stats = {"chains": 0, "links": 0, "helper_ops": 0, "levels_before": max(level) + 1}

    def emit(code, a=0, b=0, imm=0):
        ops.append((code, a, b, imm))
        return len(ops) - 1

    def scaled(v, m):  # v * m
        if v[0] == "i":
            return ("i", v[1] * m)
        return v if m == 1 else ("s", emit(_MULI, v[1], imm=m))

    def added(t, v):  # t + v
        if t[0] == "i" and v[0] == "i":
            return ("i", t[1] + v[1])
        if t[0] == "i":
            t, v = v, t
        if v[0] == "i":
            return t if v[1] == 0 else ("s", emit(_ADDI, t[1], imm=v[1]))
        return ("s", emit(_ADD, t[1], v[1]))

    for ch in chains:
        L = len(ch)
        if L < MIN_CHAIN:
            continue
        base = link[ch[0]][0]
        m = [link[i][1] for i in ch]      # multiplier of map j (composition of maps j - span + 1 .. j after the rounds)
        v = [link[i][2] for i in ch]      # that map's constant term: a slot or an immediate
        # dry run on the integers: every multiplier and every immediate term a round or the final step needs must fit 64 bits
        ok, mm, vv, d = True, list(m), [x[1] if x[0] == "i" else 0 for x in v], 1
        while d < L and ok:
            vv = [vv[j] if j < d else vv[j - d] * mm[j] + vv[j] for j in range(L)]
            mm = [mm[j] if j < d else mm[j - d] * mm[j] for j in range(L)]
            ok = all(x <= M64 for x in mm) and all(x <= M64 for x in vv)
            d *= 2
        if not ok:
            continue
        d = 1
        while d < L:
            nm, nv = list(m), list(v)
            for j in range(d, L):
                nv[j] = added(scaled(v[j - d], m[j]), v[j])
                nm[j] = m[j - d] * m[j]
            m, v = nm, nv
            d *= 2
        for j, i in enumerate(ch):  # x_j = M_j * x_0 + V_j: the slot keeps its index, its definition changes
            if j == 0:
                continue  # (the first link already reads the base directly)
            t = base if m[j] == 1 else emit(_MULI, base, imm=m[j])
            ops[i] = (_ADDI, t, 0, v[j][1]) if v[j][0] == "i" else (_ADD, t, v[j][1], 0)
        stats["chains"] += 1
        stats["links"] += L
    stats["helper_ops"] = len(ops) - n0
    level = _levels(ops)
    return ops, level, stats


def rebuild(tape: Tape, ops, level, roots):
    """New Tape holding the operations reachable from `roots` (slots shown in cells / instance rows) in topological order;
    returns (tape, old slot -> new slot)."""
    live = [False] * len(ops)
    stack = list(roots)
    while stack:
        i = stack.pop()
        if live[i]:
            continue
        live[i] = True
        stack.extend(_deps(ops[i]))
    order = sorted((i for i in range(len(ops)) if live[i]), key=lambda i: (level[i], i))
    new = {old: k for k, old in enumerate(order)}
    out = Tape()
    out.consts, out.table = tape.consts, tape.table
    for old in order:
        code, a, b, imm = ops[old]
        dd = _deps(ops[old])
        out.ops.append((code, new[a] if a in dd or code >= _FIRST_WITH_A else a, new[b] if code in _USES_B else b, imm))
        out.level.append(level[old])
    return out, new
