#!/usr/bin/env python3
"""VERDICT r4 item 8, answered without a kernel: does adding a random multiple of r to every scalar remove the top-digit skew
of the free-position (width-w NAF) recoding that msm_digits_naf_kernel produces?  The recoding below is that kernel's
(odd signed digits of w bits at free positions, the carry folded into the next digit).  It does not: the LAST digit of any
bounded scalar covers whatever bits remain above the previous digit -- between 1 and w of them, about uniformly -- so its
magnitude is small with the same probability whatever the scalar's range is; s + t * r only adds a digit.
    python tools/top_digit_skew_sim.py [w] [samples]"""
import collections
import random
import sys

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
w = int(sys.argv[1]) if len(sys.argv) > 1 else 15
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000


def recode(s):
    digs, pos, mask, half = [], 0, (1 << w) - 1, 1 << (w - 1)
    while s:
        if s & 1 == 0:
            z = (s & -s).bit_length() - 1
            s >>= z
            pos += z
            continue
        v = s & mask
        neg = v > half
        digs.append((pos, (mask + 1 - v) if neg else v))
        s = (s >> w) + (1 if neg else 0)
        pos += w
    return digs


rnd = random.Random(1)
print(f"width-{w} free-position recoding, {N} uniform scalars each")
for label, gen in (("s < r (what the prover multiplies)", lambda: rnd.randrange(R)),
                   (f"s + t r, t uniform below 2^{w}", lambda: rnd.randrange(R) + rnd.randrange(1 << w) * R),
                   ("s + t r, t uniform below 2^30", lambda: rnd.randrange(R) + rnd.randrange(1 << 30) * R)):
    tops, nd, top_pos = collections.Counter(), 0, 0
    for _ in range(N):
        d = recode(gen())
        nd += len(d)
        tops[d[-1][1]] += 1
        top_pos = max(top_pos, d[-1][0])
    hot = sum(c for v, c in tops.items() if v < 16)
    print(f"  {label:42s} digits per scalar {nd / N:6.3f}   P(top digit = 1) {tops[1] / N:.4f}   = 3: {tops[3] / N:.4f}   < 16: {hot / N:.4f}   "
          f"highest position {top_pos}")
