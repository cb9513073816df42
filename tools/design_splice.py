#!/usr/bin/env python3
"""Replace the generated tables of DESIGN.md by what tools/design_tables.py ROUND prints now:   python tools/design_splice.py r05
A table is found by the start of its header row and runs to the first line that is not a table row; where a header occurs
several times (the per-kernel tables of section 4: k = 14, 15, 17) the k-th table of the document takes the k-th generated one."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), rnd], capture_output=True, text=True, check=True).stdout
path = os.path.join(ROOT, "DESIGN.md")
doc = open(path).read().split("\n")
gen = out.split("\n")
HEADERS = ("| kernel | µs per proof | share |", "| family | share of device time |", "| model | k | provers × batch |", "| | digit tables | gate |",
           "| model | k | world | points per rank |", "| model | world | points per rank | digit width |", "| model | levels: recorded → parallel prefix |")


def tables(lines, header_start):
    found, k = [], 0
    while k < len(lines):
        if lines[k].startswith(header_start):
            j = k
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            found.append((k, j))
            k = j
        else:
            k += 1
    return found


n = 0
for header in HEADERS:
    g, d = tables(gen, header), tables(doc, header)
    for (gi, gj), (di, dj) in reversed(list(zip(g, d))):
        doc[di:dj] = gen[gi:gj]
        n += 1
for k, ln in enumerate(doc):  # the sentences under the tables
    for start in ("SURVEY 8d per proof:",):
        if ln.startswith(start):
            doc[k] = next(x for x in gen if x.startswith(start))
open(path, "w").write("\n".join(doc))
print("spliced", n, "tables")
