#!/usr/bin/env python3
"""Replace the two generated tables of DESIGN.md (the per-kernel table of section 4 and the family table of section 5) by
what tools/design_tables.py ROUND prints now:      python tools/design_splice.py r04
A table is found by its header row and runs to the first line that is not a table row."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), rnd], capture_output=True, text=True, check=True).stdout
path = os.path.join(ROOT, "DESIGN.md")
doc = open(path).read().split("\n")
gen = out.split("\n")


def table(lines, header_start):
    i = next(k for k, ln in enumerate(lines) if ln.startswith(header_start))
    j = i
    while j < len(lines) and lines[j].startswith("|"):
        j += 1
    return i, j


for header in ("| kernel | µs per proof | share |", "| family | share of device time |"):
    gi, gj = table(gen, header)
    di, dj = table(doc, header)
    doc[di:dj] = gen[gi:gj]
# the sentence under the family table
for k, ln in enumerate(doc):
    if ln.startswith("SURVEY 8d per proof:"):
        doc[k] = next(x for x in gen if x.startswith("SURVEY 8d per proof:"))
open(path, "w").write("\n".join(doc))
print("spliced")
