#!/usr/bin/env python3
"""Checks the product tools/fp64_probe.bin printed against Python integers: a = x * y * 2^-260 mod q and
b = y * a * 2^-260 mod q (lazy residues: compared modulo q), then prints the rate table and the gate's verdict.
    python tools/fp64_probe_check.py gpurun_out/fp64_probe.json"""
import json
import sys

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
d = json.load(open(sys.argv[1]))
val = lambda limbs: sum(int(v) << (52 * i) for i, v in enumerate(limbs))
x, y, a, b = (val(d[k]) for k in "xyab")
rinv = pow(1 << 260, -1, Q)
ok = a % Q == x * y * rinv % Q and b % Q == y * a * rinv % Q and a < (1 << 252) + Q and b < (1 << 252) + Q
print("product check:", "ok" if ok else "MISMATCH")
for r in d["rates"]:
    print(f"{r['waves_per_simd']} waves/SIMD: fp64 5x52 {r['fp64_5x52_Gmul_s']:7.1f} G products/s | int 9x29 {r['int_9x29_Gmul_s']:7.1f} | ratio {r['ratio']:.3f}")
r3 = [r for r in d["rates"] if r["waves_per_simd"] == 3][0]
print(f"gate (>= 1.2 at 3 waves/SIMD): {r3['ratio']:.3f} ->", "PASS" if ok and r3["ratio"] >= 1.2 else "REJECTED")
sys.exit(0 if ok else 1)
