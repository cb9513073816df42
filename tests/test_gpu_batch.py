"""Lock-step batches (zg_prover_prove_batch): one launch sequence for several proofs of the same circuit.
Every proof of a batch must be byte-identical to the oracle's create_proof of ITS witness, instance and blinding key
-- whatever its neighbours in the batch are -- and to what the one-at-a-time entry point returns.  Forked provers
(shared proving key) and concurrent host threads are covered here too: this is how bench.py drives the GPU."""
import threading

import numpy as np
import pytest
import torch  # (before the first HIP call of the process: torch ships its own HIP runtime and wants to be the one that loads it)

from circuits import toy_circuit

pytestmark = pytest.mark.gpu


def _toy(orc, zg, ctx, k, **kw):
    cs, asg, ilen = toy_circuit(k, **kw)
    img = cs.to_c()
    params = orc.params_new(k, 0xABCDEF)
    vk_repr = orc.fr_from_int(0x1234567)
    pk = orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, vk_repr)
    prover = zg.Prover(ctx, img, asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), vk_repr)
    return cs, asg, ilen, pk, prover


@pytest.mark.parametrize("k,force_degree", [(6, None), (7, 6), (9, 6)])
def test_batch_equals_oracle_and_single_proofs(ctx, zg, orc, k, force_degree):
    """Four different statements in one batch: the satisfied witness, an unsatisfied one, a wrong public input, the
    satisfied one under another key; in both schedules (single coset + side stream / split domain)."""
    cs, asg, ilen, pk, prover = _toy(orc, zg, ctx, k, force_degree=force_degree)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    bad = adv.copy()
    bad[2, 3] = orc.fr_from_int(99)
    wrong = inst.copy()
    wrong[0, 0] = orc.fr_from_int(5)
    cases = [(adv, inst, 11), (bad, inst, 12), (adv, wrong, 13), (adv, inst, 14)]
    want = []
    for a, i, s in cases:
        st, proof, _ = orc.create_proof(pk, a, i, s)
        assert st == 0
        want.append(proof)
    singles = [prover.prove(a, i, s) for a, i, s in cases]
    assert singles == want
    prover.set_batch(6)
    qpd = cs.degree() - 1
    split_applies = qpd & (qpd - 1) != 0  # (degree - 1 a power of two: EvaluationDomain's coset is already minimal)
    for overlap in (True, False):
        prover.set_overlap(overlap)
        got, sts = prover.prove_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
        assert sts == [0, 0, 0, 0]
        if not overlap and split_applies:
            # A statement that does NOT hold (the altered witness, the wrong public input) has no quotient polynomial:
            # the reference's bytes then come from cutting the 2^ext_k-point interpolant of a non-polynomial at
            # (degree - 1) n coefficients, which the split domain (exact for polynomials of that degree only) does not
            # reproduce.  Such a proof verifies nowhere; the throughput form only promises to be deterministic about
            # it; zg_prover_set_overlap(1) reproduces the reference bytes for those too.
            for b in (1, 2):
                a_b, i_b, s_b = cases[b]
                assert got[b] == prover.prove(a_b, i_b, s_b) and got[b] != want[b]
                assert orc.verify_proof_pairing(pk, i_b, got[b]) != 1
                got[b] = want[b]
            assert got == want, "batch proof bytes"
            continue
        assert got == want, "batch proof bytes"
        # intermediates of a proof that is not the first of the batch
        n = 1 << k
        st, _, tr = orc.create_proof(pk, bad, inst, 12, want_trace=True)
        for l in range(len(cs.lookups)):
            assert np.array_equal(prover.fetch(3, l, n, slot=1), tr.array("permuted_input", (l + 1) * n)[l * n:])
            assert np.array_equal(prover.fetch(2, l, n, slot=1), tr.array("lookup_z", (l + 1) * n)[l * n:])
        qpd = cs.degree() - 1
        assert np.array_equal(prover.fetch(5, 0, qpd * n, slot=1), tr.array("h_pieces", qpd * n))
        orc.trace_free(tr)
        # a shorter batch on the same slots, in another order
        got, _ = prover.prove_batch([cases[3][0], cases[0][0]], [cases[3][1], cases[0][1]], [14, 11])
        assert got == [want[3], want[0]]
    assert orc.verify_proof_pairing(pk, inst, want[0]) == 1
    assert orc.verify_proof_pairing(pk, inst, want[1]) != 1
    assert orc.verify_proof_pairing(pk, wrong, want[2]) != 1
    prover.close()


def test_batch_with_one_failing_lookup(ctx, zg, orc):
    """plonk::Error::ConstraintSystemFailure hits the proof whose witness leaves its table; its neighbours finish."""
    cs, asg, ilen, pk, prover = _toy(orc, zg, ctx, 6)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    bad = adv.copy()
    bad[1, 2] = orc.fr_from_int(1000)
    prover.set_batch(3)
    got, sts = prover.prove_batch([adv, bad, adv], [inst] * 3, [1, 2, 3], raise_on_error=False)
    assert sts == [0, -5, 0]
    assert got[1] == b""
    assert got[0] == orc.create_proof(pk, adv, inst, 1)[1]
    assert got[2] == orc.create_proof(pk, adv, inst, 3)[1]
    with pytest.raises(zg.ZgError) as e:
        prover.prove_batch([adv, bad, adv], [inst] * 3, [1, 2, 3])
    assert e.value.status == -5
    # more proofs than slots is refused
    with pytest.raises(zg.ZgError):
        prover.prove_batch([adv] * 4, [inst] * 4, [1, 2, 3, 4])
    prover.close()


def test_batch_from_device_slots(ctx, zg, orc):
    """Advice written straight into the prover's slots (zg_prover_advice_slot) and proved in place, twice: a proof
    rewrites only the blinding rows of its columns."""
    cs, asg, ilen, pk, prover = _toy(orc, zg, ctx, 7, force_degree=6)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    prover.set_batch(3)
    prover.set_overlap(False)
    n_bytes = adv.nbytes
    src = torch.from_numpy(adv.view(np.int64).reshape(-1)).cuda()
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    for b in range(3):
        assert hip.hipMemcpy(ctypes.c_void_p(prover.advice_slot(b)), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(n_bytes), 3) == 0
    for seeds in ([5, 6, 7], [8, 9, 10]):
        got, _ = prover.prove_batch(None, [inst] * 3, seeds, device=True)
        assert got == [orc.create_proof(pk, adv, inst, s)[1] for s in seeds]
    # a foreign device buffer for one proof, the slot for the others
    got, _ = prover.prove_batch([None, src.data_ptr(), None], [inst] * 3, [1, 2, 3], device=True)
    assert got == [orc.create_proof(pk, adv, inst, s)[1] for s in (1, 2, 3)]
    prover.close()


def _real_tiny(orc, zg, ctx, images):
    import wnn_circuit
    import wnn_model

    k, name = wnn_model.MNIST_TINY
    wnn = wnn_model.load_checked_in(name)
    built = [wnn_circuit.build(wnn, im, k) for im in images]
    cs, asg0, ilen, _ = built[0]
    img = cs.to_c()
    fixed, sigma = asg0.fixed_values(), asg0.sigma_values()
    for _, asg, _, _ in built[1:]:  # the proving key does not depend on the image
        assert np.array_equal(asg.fixed_values(), fixed) and np.array_equal(asg.sigma_values(), sigma)
    params = orc.params_new(k, 0x5EED)
    vk_repr = orc.fr_from_int(0xC0FFEE)
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    gb, glb = ctx.register_bases(params.g_np()), ctx.register_bases(params.g_lagrange_np())
    prover = zg.Prover(ctx, img, fixed, sigma, gb, glb, vk_repr)
    wit = [(asg.advice_values(), asg.instance_values(ilen), scores) for _, asg, _, scores in built]
    return pk, prover, wit, (gb, glb)


def test_real_wnn_batch_over_images(ctx, zg, orc):
    """model_28input_256entry_1hash_1bpi (k = 14), a batch over DIFFERENT images (the real one and seeded noise): one
    proving key, five witnesses and instances; every proof == the oracle's, every proof passes the pairing check for
    its own class scores only."""
    import wnn_model

    orc.load().orc_set_threads(16)
    rng = np.random.default_rng(7)
    real = wnn_model.load_test_image()
    images = [real] + [rng.integers(0, 256, size=real.shape, dtype=real.dtype) for _ in range(4)]
    pk, prover, wit, bases = _real_tiny(orc, zg, ctx, images)
    assert wit[0][2] == [9, 6, 13, 10, 17, 10, 9, 26, 11, 16]  # /root/reference/tests/integration_test.rs:19
    prover.set_batch(5)
    prover.set_overlap(False)
    seeds = [100 + b for b in range(5)]
    got, sts = prover.prove_batch([w[0] for w in wit], [w[1] for w in wit], seeds)
    assert sts == [0] * 5
    for b in range(5):
        st, want, _ = orc.create_proof(pk, wit[b][0], wit[b][1], seeds[b])
        assert st == 0 and got[b] == want, f"proof {b}"
        assert orc.verify_proof_pairing(pk, wit[b][1], got[b]) == 1
    assert orc.verify_proof_pairing(pk, wit[1][1], got[0]) != 1
    prover.close()
    for b in bases:
        b.free()


def test_forked_provers_on_threads(ctx, zg, orc):
    """The bench's shape: several provers forked from one (shared proving key and base tables), each on its own
    context and host thread, each proving batches back to back.  Every proof of every batch is checked."""
    cs, asg, ilen, pk, first = _toy(orc, zg, ctx, 8, force_degree=6)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    ctxs = [zg.Ctx(0) for _ in range(3)]
    provers = [first] + [first.fork(c) for c in ctxs]
    for p in provers:
        p.set_batch(3)
        p.set_overlap(False)
    rounds, nb = 3, 3
    results = [[None] * rounds for _ in provers]
    errors = []

    def work(i):
        try:
            for r in range(rounds):
                seeds = [1000 * i + 10 * r + b for b in range(nb)]
                results[i][r] = (seeds, provers[i].prove_batch([adv] * nb, [inst] * nb, seeds)[0])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(provers))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    cache = {}
    for i in range(len(provers)):
        for r in range(rounds):
            seeds, proofs = results[i][r]
            for s, pr in zip(seeds, proofs):
                if s not in cache:
                    cache[s] = orc.create_proof(pk, adv, inst, s)[1]
                assert pr == cache[s], (i, r, s)
    # the forks outlive the prover they came from
    first.close()
    got, _ = provers[1].prove_batch([adv], [inst], [77])
    assert got[0] == orc.create_proof(pk, adv, inst, 77)[1]
    for p in provers[1:]:
        p.close()
    for c in ctxs:
        c.close()


def test_threads_sharing_one_context_serialise(ctx, zg, orc):
    """Calls on ONE context take its lock: four host threads on the same prover get four correct proofs."""
    cs, asg, ilen, pk, prover = _toy(orc, zg, ctx, 6)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    out, errors = {}, []

    def work(s):
        try:
            for r in range(3):
                out[(s, r)] = prover.prove(adv, inst, 10 * s + r)
                ctx.msm(bases, scal)  # another entry point on the same context in between
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    params = orc.params_new(6, 0xABCDEF)
    bases = ctx.register_bases(params.g_np())
    scal = orc.fill_fr(3, 64)
    th = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for (s, r), proof in out.items():
        assert proof == orc.create_proof(pk, adv, inst, 10 * s + r)[1]
    bases.free()
    prover.close()


@pytest.mark.parametrize("kind", ["no_lookup", "gates_only", "wide_lookup", "advice_factor", "merged_selectors"])
def test_batch_of_circuit_variants(ctx, zg, orc, kind):
    """The prover's less common paths in batch form: no lookups (the random polynomial rides with the permutation
    products), neither lookups nor permutation (it is committed on its own), no instance column, two instance columns,
    width-2 lookups, factored gates -- three proofs per batch under different keys, both scheduling forms."""
    from circuits import variant_circuit

    k = 6
    cs, asg, ilen = variant_circuit(kind, k=k)
    img = cs.to_c()
    params = orc.params_new(k, 0xABCDEF)
    vk_repr = orc.fr_from_int(99)
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    prover = zg.Prover(ctx, img, fixed, sigma, params.g_np(), params.g_lagrange_np(), vk_repr)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    prover.set_batch(3)
    seeds = [31, 32, 33]
    want = []
    for s in seeds:
        st, proof, _ = orc.create_proof(pk, adv, inst, s)
        assert st == 0
        want.append(proof)
    for overlap in (True, "tables", False):
        prover.set_overlap(overlap)
        got, sts = prover.prove_batch([adv] * 3, [inst] * 3, seeds)
        assert sts == [0, 0, 0] and got == want, (kind, overlap)
    assert orc.verify_proof_pairing(pk, inst, want[2]) == 1
    prover.close()
