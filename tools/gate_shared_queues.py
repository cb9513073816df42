#!/usr/bin/env python3
"""Why ZG_LAT_GATE is opt-in: T concurrent lone provers (latency form, digit tables, gate on), one host thread each, in a
process with Q hardware queues:    GPU_MAX_HW_QUEUES=Q python tools/gate_shared_queues.py [T] [proofs]
Prints every thread's median and worst latency and how many proofs took longer than a second (a gate that ran into
its time limit: the proof is then made again in the plain order -- right bytes, late).  Run under `timeout -k 10 S`."""
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "harness"))
sys.path.insert(0, ROOT)
queues = os.environ.get("GPU_MAX_HW_QUEUES", "(runtime default: 4)")
import bench  # noqa: E402  (setdefault: keeps a GPU_MAX_HW_QUEUES given in the environment)

zg = bench.zg
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx0 = zg.Ctx(0)
c = bench.Circuit(ctx0, "tiny")
first = zg.Prover(ctx0, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
first.set_overlap("tables")
ctxs = [ctx0] + [zg.Ctx(0) for _ in range(T - 1)]
provers = [first] + [first.fork(x) for x in ctxs[1:]]
for p in provers:
    p.set_overlap(True)
want = [first.prove(c.advice, c.instance, 500 + i) for i in range(4)]
for p in provers[1:]:
    p.prove(c.advice, c.instance, 500)
zg.tuning_set("ZG_LAT_GATE", 1)
lat = [[] for _ in provers]
bad = []


def work(t):
    p = provers[t]
    for i in range(N):
        t0 = time.perf_counter()
        proof = p.prove_dev(p.advice_slot(0), c.instance, 500 + i % 4)
        lat[t].append((time.perf_counter() - t0) * 1e3)
        if proof != want[i % 4]:
            bad.append((t, i))


th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
t0 = time.perf_counter()
for x in th:
    x.start()
for x in th:
    x.join()
wall = time.perf_counter() - t0
zg.tuning_set("ZG_LAT_GATE", -1)
print(f"GPU_MAX_HW_QUEUES={queues}, {T} concurrent gated lone provers x {N} proofs: wall {wall:.2f} s, wrong proofs {len(bad)}")
for t in range(T):
    print(f"  prover {t}: median {statistics.median(lat[t]):.2f} ms, worst {max(lat[t]):.1f} ms, proofs over 1 s: {sum(x > 1000 for x in lat[t])}")
