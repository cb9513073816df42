#!/usr/bin/env python3
"""Concurrency soak of the bench configuration: P forked provers x batches of B on P host threads, S steps.  Every
(prover, step, slot) proves under a key that a DIFFERENT prover uses in a different slot and step, so every proof has a
twin made elsewhere, at another time, next to other neighbours: all twins must agree byte for byte, and a sample is
checked against the oracle.    python tools/soak.py [P] [B] [S] [images]
With a fourth argument every proof is for an image of bench.image_pool chosen by its key, the witness made on the device
(zg_prover_prove_images): twins then share image AND key, and the oracle sample proves the host-synthesised witness."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

zg = bench.zg
P, B, S = (int(a) for a in (sys.argv[1:4] + ["8", "16", "12"][len(sys.argv) - 1:])[:3])
IMAGES = len(sys.argv) > 4
ctx0 = zg.Ctx(0)
c = bench.Circuit(ctx0, "tiny")
ctxs, streams, _ = bench.make_streams(0, c, ctx0, P, B, 0)
out = [[None] * S for _ in range(P)]
errors = []
pool = plans = None
if IMAGES:
    import witness_tape

    pool = bench.image_pool(c)
    arrays = witness_tape.trace(c.wnn, c.k).arrays()
    plans = [zg.WitnessPlan(x, arrays) for x in ctxs]


def key(p, s, b):  # the twin of (p, s, b) is (p ^ 1, S - 1 - s, B - 1 - b): same key, other prover, slot and time
    lo = min((p, s, b), (p ^ 1 if (p ^ 1) < P else p, S - 1 - s, B - 1 - b))
    return 7_000_000 + (lo[0] * S + lo[1]) * B + lo[2]


def work(p):
    try:
        for s in range(S):
            seeds = [key(p, s, b) for b in range(B)]
            if IMAGES:
                out[p][s] = streams[p].prover.prove_images(plans[p], pool[[k % len(pool) for k in seeds]], seeds)[0]
            else:
                out[p][s] = streams[p].prover.prove_batch(None, [c.instance] * B, seeds, device=True)[0]
    except Exception as e:  # noqa: BLE001
        errors.append(e)


th = [threading.Thread(target=work, args=(p,)) for p in range(P)]
for t in th:
    t.start()
for t in th:
    t.join()
assert not errors, errors
by_key, mismatches = {}, 0
for p in range(P):
    for s in range(S):
        for b in range(B):
            k = key(p, s, b)
            if k in by_key and by_key[k] != out[p][s][b]:
                mismatches += 1
            by_key.setdefault(k, out[p][s][b])
print(f"{P * S * B} proofs, {len(by_key)} distinct keys, {mismatches} twin mismatches")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

params = orc.params_from_scalar(c.k, c.s)
pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
sample = sorted(by_key)[:: max(1, len(by_key) // 6)][:6]


def want(k):
    if not IMAGES:
        return orc.create_proof(pk, c.advice, c.instance, k)[1]
    im = pool[k % len(pool)].reshape(bench.wnn_model.load_test_image().shape)
    _, asg, ilen, _ = bench.wnn_circuit.build(c.wnn, im, c.k)
    return orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), k)[1]


bad = sum(want(k) != by_key[k] for k in sample)
print(f"oracle check of {len(sample)} sampled keys: {bad} differ")
sys.exit(1 if (mismatches or bad) else 0)
