#!/usr/bin/env python3
"""How long zg_prover_enable_digit_tables takes for the tiny model's three base sets (78 GB), and that a lone proof over the
fresh tables is the oracle's:   python tools/table_build_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402  (paths, torch before the library)

zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, "tiny")
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
t0 = time.time(); b = p.enable_digit_tables(); ctx.sync(); print(zg.LIB_PATH, "digit tables", b, "bytes in", round(time.time() - t0, 2), "s")
p.set_overlap(True)
import orc
proof = p.prove(c.advice, c.instance, 7)
prm = orc.params_from_scalar(c.k, c.s); pk = orc.ProvingKey(c.img, c.fixed, c.sigma, prm, c.vk_repr)
print("proof == oracle:", orc.create_proof(pk, c.advice, c.instance, 7)[1] == proof)
