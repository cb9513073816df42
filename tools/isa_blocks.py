#!/usr/bin/env python3
"""Basic blocks of a kernel in hipcc's -S output with their instruction mix:   python tools/isa_blocks.py FILE.s KERNEL_SUBSTRING [min_valu]
(the loop bodies of a kernel are its large blocks: what a change of the arithmetic did to them is read off here)"""
import re
import sys

src, want = open(sys.argv[1]).read().split("\n"), sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 100
inside, name, block, rows = False, None, None, []
for ln in src:
    m = re.match(r"^(_Z\S+):\s", ln)
    if m:
        inside = want in m.group(1)
        name, block = m.group(1), "entry"
        if inside:
            print(name)
            rows.append([block, []])
        continue
    if not inside:
        continue
    t = ln.strip()
    if t.startswith("s_endpgm"):
        inside = False
        for b, ins in rows:
            v = [i for i in ins if i.startswith("v_")]
            if len(v) < floor:
                continue
            mad = sum(i.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) for i in v)
            print(f"  {b:14s} valu {len(v):5d}  mad64 {mad:4d}  mul_lo/hi {sum(i.startswith(('v_mul_lo','v_mul_hi')) for i in v):3d}  "
                  f"ds {sum(i.startswith('ds_') for i in ins):3d}  global/scratch {sum(i.startswith(('global_','scratch_','buffer_')) for i in ins):3d}  "
                  f"salu {sum(i.startswith('s_') and not i.startswith(('s_waitcnt','s_nop')) for i in ins):4d}  waitcnt {sum(i.startswith('s_waitcnt') for i in ins):3d}")
        rows = []
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        rows.append([m.group(1), []])
        continue
    if t and not t.startswith((";", ".")):
        rows[-1][1].append(t)
