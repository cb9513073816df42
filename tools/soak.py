#!/usr/bin/env python3
"""Concurrency soak of the bench configuration: P forked provers x batches of B on P host threads, S steps.  Every
(prover, step, slot) proves under a key that a DIFFERENT prover uses in a different slot and step, so every proof has a
twin made elsewhere, at another time, next to other neighbours: all twins must agree byte for byte, and a sample is
checked against the oracle (at least one proof of every prover).    python tools/soak.py [P] [B] [S] [images]
With a fourth argument every proof is for an image of bench.image_pool chosen by its key, the witness made on the device
(zg_prover_prove_images): twins then share image AND key, and the oracle sample proves the host-synthesised witness.
tests/test_gpu_soak.py runs a reduced form of this (12 x 16 x 2) inside `pytest -m gpu`."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def soak(P: int, B: int, S: int, images: bool = False, model: str = "tiny", oracle_per_prover: int = 1, log=print) -> dict:
    import bench

    zg = bench.zg
    ctx0 = zg.Ctx(0)
    c = bench.Circuit(ctx0, model)
    ctxs, streams, _ = bench.make_streams(0, c, ctx0, P, B, 0)
    out = [[None] * S for _ in range(P)]
    errors = []
    pool = plans = None
    if images:
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
                if images:
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
    by_key, mismatches, twins = {}, 0, 0
    produced = [[] for _ in range(P)]
    for p in range(P):
        for s in range(S):
            for b in range(B):
                k = key(p, s, b)
                if k in by_key:
                    twins += 1
                    if by_key[k] != out[p][s][b]:
                        mismatches += 1
                by_key.setdefault(k, out[p][s][b])
                produced[p].append(k)
    log(f"{P * S * B} proofs, {len(by_key)} distinct keys, {twins} twin pairs, {mismatches} twin mismatches")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(bench.host_cores())
    params = orc.params_from_scalar(c.k, c.s)
    pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
    sample = []
    for p in range(P):  # `oracle_per_prover` proofs of EVERY prover (spread over its steps and slots; twins count once)
        mine = [k for k in sorted(set(produced[p])) if k not in sample]
        sample += mine[:: max(1, len(mine) // max(1, oracle_per_prover))][:oracle_per_prover]

    def want(k):
        if not images:
            return orc.create_proof(pk, c.advice, c.instance, k)[1]
        im = pool[k % len(pool)].reshape(bench.wnn_model.load_test_image().shape)
        _, asg, ilen, _ = bench.wnn_circuit.build(c.wnn, im, c.k)
        return orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), k)[1]

    bad = sum(want(k) != by_key[k] for k in sample)
    log(f"oracle check of {len(sample)} sampled keys: {bad} differ")
    for s in streams:
        s.prover.close()
    if plans:
        for pl in plans:
            pl.close()
    c.g_bases.free()
    c.gl_bases.free()
    for x in ctxs:
        x.close()
    return {"proofs": P * S * B, "distinct_keys": len(by_key), "twin_pairs": twins, "twin_mismatches": mismatches,
            "oracle_checked": len(sample), "oracle_mismatches": bad}


if __name__ == "__main__":
    P, B, S = (int(a) for a in (sys.argv[1:4] + ["8", "16", "12"][len(sys.argv) - 1:])[:3])
    r = soak(P, B, S, images=len(sys.argv) > 4)
    sys.exit(1 if (r["twin_mismatches"] or r["oracle_mismatches"]) else 0)
