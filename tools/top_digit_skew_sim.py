#!/usr/bin/env python3
"""The top digit of the free-position (width-w NAF) recoding that msm_digits_naf_kernel produces, in Python:
the kernel's own loop -- eight 32-bit limbs through a 64-bit sliding register, odd signed digits of w bits at free positions,
the carry folded into the register -- so that what is measured and checked here is what the kernel does.

1. Does adding a random multiple of r to every scalar remove the skew of the last digit (round 4's candidate)?  No: the last
   digit covers whatever bits remain above the previous one -- between 1 and w of them, about uniformly -- whatever range the
   scalar is drawn from; s + t r only adds a digit.
2. What does: cutting the LAST TWO digits evenly (`balanced`, ZG_MSM_TOPSPLIT = 1: once every limb is in the register and fewer
   than 2 (w - 1) bits are left, the next digit takes half of them; a remainder below 2^(w-1) is the last digit as it stands).
   Checked for w = 3 .. 16 on random and adversarial scalars: the digits sum to the scalar, are odd, at most 2^(w-1) in
   magnitude, at increasing positions <= 254, and never more than the 254 / w + 1 slots the launch reserves.
    python tools/top_digit_skew_sim.py [samples]"""
import collections
import random
import sys

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
M64 = (1 << 64) - 1
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def ctz(x):
    return (x & -x).bit_length() - 1


def recode(s, w, balanced, limbs=8):
    """msm_digits_naf_kernel's loop (csrc/msm.hip); `limbs` > 8 only to recode s + t r for question 1"""
    l = [(s >> (32 * i)) & 0xFFFFFFFF for i in range(limbs)]
    bits, have, nxt, pos, out = l[0], 32, 1, 0, []
    while True:
        if have <= 16 and nxt < limbs:
            bits = (bits + (l[nxt] << have)) & M64
            have += 32
            nxt += 1
        if bits == 0 and nxt >= limbs:
            break
        if bits & 1 == 0:
            z = ctz(bits) if bits else have
            z = max(1, min(z, have))
            bits >>= z
            have -= z
            pos += z
            continue
        wd = w
        if balanced and nxt >= limbs:
            rem = bits.bit_length()
            if rem <= w - 1:
                out.append((pos, bits))
                break
            if rem <= 2 * (w - 1):
                wd = (rem + 1) >> 1
        m, hf = (1 << wd) - 1, 1 << (wd - 1)
        v = bits & m
        neg = v > hf
        out.append((pos, -(m + 1 - v) if neg else v))
        bits = (bits >> wd) + (1 if neg else 0)
        have = have - wd if have >= wd else 0
        pos += wd
    return out


rnd = random.Random(1)
print(f"1. width-15 recoding as the prover runs it, {N} uniform scalars each")
for label, gen, limbs in (("s < r", lambda: rnd.randrange(R), 8), ("s + t r, t uniform below 2^15", lambda: rnd.randrange(R) + rnd.randrange(1 << 15) * R, 9),
                          ("s + t r, t uniform below 2^30", lambda: rnd.randrange(R) + rnd.randrange(1 << 30) * R, 9)):
    tops, nd = collections.Counter(), 0
    for _ in range(N):
        s = gen()
        d = recode(s, 15, False, limbs)
        assert sum(v << p for p, v in d) == s
        nd += len(d)
        tops[abs(d[-1][1])] += 1
    print(f"   {label:32s} digits per scalar {nd / N:6.3f}   P(top digit = 1) {tops[1] / N:.4f}   = 3: {tops[3] / N:.4f}   < 16: "
          f"{sum(c for v, c in tops.items() if v < 16) / N:.4f}")

adv = [R - 1, R - 2, 1, 2, 3, (1 << 253) + 1, int("10" * 127, 2), int("1" * 253, 2)] + [(1 << b) - 1 for b in range(1, 254)] + [(1 << b) + 1 for b in range(1, 253)] \
    + [((1 << b) - 1) << c for b in (1, 5, 14, 15, 16, 29, 30, 31, 47, 48) for c in (0, 17, 100, 200, 223)]
adv = [a for a in adv if 0 < a < R]
print(f"2. balanced last digits, {N // 4} uniform + {len(adv)} adversarial scalars per width")
print("   w  | digits per scalar: as is / balanced | most digits: as is / balanced / slots | P(top = 1): as is / balanced | heaviest bucket over the mean: as is / balanced")
for w in range(3, 17):
    row = {}
    for bal in (False, True):
        nd, mx, tops, load, cnt = 0, 0, collections.Counter(), collections.Counter(), 0
        for k in range(N // 4 + len(adv)):
            s = adv[k] if k < len(adv) else rnd.randrange(R)
            d = recode(s, w, bal)
            assert sum(v << p for p, v in d) == s, (w, bal, s)
            assert all(v % 2 and 0 < abs(v) <= 1 << (w - 1) for _, v in d) and all(p <= 254 for p, _ in d), (w, bal, s)
            assert all(d[i + 1][0] > d[i][0] for i in range(len(d) - 1)), (w, bal, s)
            mx = max(mx, len(d))
            if k >= len(adv):
                nd += len(d)
                cnt += 1
                tops[abs(d[-1][1])] += 1
                for _, v in d:
                    load[abs(v)] += 1
        assert mx <= 254 // w + 1, (w, bal, mx)
        row[bal] = (nd / cnt, mx, tops[1] / cnt, max(load.values()) / (nd / (1 << (w - 2))))
    a, b = row[False], row[True]
    print(f"   {w:2d} | {a[0]:7.3f} / {b[0]:7.3f} | {a[1]:3d} / {b[1]:3d} / {254 // w + 1:3d} | {a[2]:.4f} / {b[2]:.4f} | {a[3]:6.1f} / {b[3]:6.1f}")
