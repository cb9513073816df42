"""GPU parity for the rest of create_proof: lookup/permutation grand products, evaluate_h, vanishing,
evaluations, GWC openings and the Keccak transcript.  The HIP prover (through the C ABI) must produce
the SAME PROOF BYTES as the oracle restatement for the same circuit, witness, SRS and blinding seed,
and the oracle's verifier must accept them."""
import numpy as np
import pytest
import torch

from circuits import toy_circuit

pytestmark = pytest.mark.gpu


def dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def setup(orc, zg, ctx, k, **kw):
    cs, asg, ilen = toy_circuit(k, **kw)
    img = cs.to_c()
    params = orc.params_new(k, 0xABCDEF)
    vk_repr = orc.fr_from_int(0x1234567)
    pk = orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, vk_repr)
    prover = zg.Prover(ctx, img, asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), vk_repr)
    return cs, asg, ilen, pk, prover


@pytest.mark.parametrize("k,force_degree", [(5, None), (5, 6), (6, 8), (8, 6), (10, None)])
def test_proof_bytes_match_oracle_and_verify(ctx, zg, orc, k, force_degree):
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, k, force_degree=force_degree)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    for seed in (1, 2):
        st, want, tr = orc.create_proof(pk, adv, inst, seed, want_trace=True)
        assert st == 0
        got = prover.prove(adv, inst, seed)
        n, en = 1 << k, 1 << cs.extended_k()
        sets = tr.n_sets
        # piecewise first, so that a mismatch names the stage
        for l in range(len(cs.lookups)):
            assert np.array_equal(prover.fetch(3, l, n), tr.array("permuted_input", (l + 1) * n)[l * n:]), "a'"
            assert np.array_equal(prover.fetch(4, l, n), tr.array("permuted_table", (l + 1) * n)[l * n:]), "s'"
            assert np.array_equal(prover.fetch(2, l, n), tr.array("lookup_z", (l + 1) * n)[l * n:]), "lookup z"
        for s in range(sets):
            assert np.array_equal(prover.fetch(1, s, n), tr.array("perm_z", (s + 1) * n)[s * n:]), "perm z"
        assert np.array_equal(prover.fetch(0, 0, en), tr.array("h_ext", en)), "h on the extended coset"
        qpd = cs.degree() - 1
        assert np.array_equal(prover.fetch(5, 0, qpd * n), tr.array("h_pieces", qpd * n)), "h pieces"
        assert got == want, "proof bytes"
        assert orc.verify_proof_pairing(pk, inst, got) == 1
        orc.trace_free(tr)
    prover.close()


def test_single_stream_schedule_gives_the_same_bytes(ctx, zg, orc):
    """zg_prover_set_overlap only changes which HIP stream the transforms run on."""
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 9)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    st, want, _ = orc.create_proof(pk, adv, inst, 5)
    assert st == 0
    for overlap in (False, True, "tables", False):  # ("tables": the latency form over its digit tables, built here)
        prover.set_overlap(overlap)
        assert prover.prove(adv, inst, 5) == want
    prover.close()


def test_gated_lone_proofs_give_the_same_bytes(ctx, zg, orc):
    """ZG_LAT_GATE = 1 (opt-in): a repeated lone proof in the latency form queues each phase behind a gate kernel before the
    previous phase's challenge exists (five gates per proof).  Same bytes as the oracle with the gate and without it; a proof
    whose signature differs from the last completed one's (first in a form, first after a zg_tuning_set) is never gated; a
    failing lookup under the gate is still ConstraintSystemFailure and stalls nothing.  Asserted on the MECHANISM
    (zg_prover_gate_stats: gated proofs, gates armed, proofs re-made, gates let go), not on wall-clock."""
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 9)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    want = [orc.create_proof(pk, adv, inst, s)[1] for s in range(4)]
    prover.set_overlap("tables")

    def prove(seed):
        before = prover.gate_stats()
        assert prover.prove(adv, inst, seed) == want[seed]
        after = prover.gate_stats()
        return {k_: after[k_] - before[k_] for k_ in after}

    plain = {"gated_proofs": 0, "gates_armed": 0, "remade_plain": 0, "yields": 0}
    gated = {"gated_proofs": 1, "gates_armed": 5, "remade_plain": 0, "yields": 0}
    try:
        zg.tuning_set("ZG_LAT_GATE", 1)
        assert prove(0) == plain  # (the first proof in this form: allocations and their synchronisations ahead)
        assert prove(1) == gated
        assert prove(2) == gated
        zg.tuning_set("ZG_LAT_GATE", 0)
        assert prove(2) == plain and prove(3) == plain
        zg.tuning_set("ZG_LAT_GATE", 1)
        assert prove(3) == plain  # (the first proof after a knob changed: it may size buffers differently)
        assert prove(3) == gated
        bad = adv.copy()
        bad[1, 2] = orc.fr_from_int(1000)
        before = prover.gate_stats()
        with pytest.raises(zg.ZgError) as e:
            prover.prove(bad, inst, 1)
        assert e.value.status == -5
        after = prover.gate_stats()
        assert after["gated_proofs"] == before["gated_proofs"] + 1 and after["remade_plain"] == before["remade_plain"]
        assert prove(2) == plain  # (a failed proof leaves no warm signature behind)
        assert prove(2) == gated
        # a gate nobody opens (knob value 2 leaves the proof's first gate closed): it gives up at its time limit (0.2 s here),
        # the phases behind it run on the previous challenge, and the library makes the proof again in the plain order
        zg.tuning_set("ZG_LAT_GATE", 2)
        assert prove(1) == plain
        assert prove(1) == {"gated_proofs": 1, "gates_armed": 5, "remade_plain": 1, "yields": 0}
        zg.tuning_set("ZG_LAT_GATE", 1)
        assert prove(0) == plain and prove(0) == gated
        prover.set_overlap(False)  # (another form: ungated again, and the throughput form never is)
        assert prove(1) == plain and prove(1) == plain
    finally:
        zg.tuning_set("ZG_LAT_GATE", -1)
    prover.close()


def test_nothing_that_allocates_or_synchronises_stands_behind_a_gate(ctx, zg, orc):
    """every call that can allocate, free or stream-synchronise on a WARM prover (DESIGN.md section 5 lists
    them), walked between gated proofs with ZG_LAT_GATE = 1.  After each, the next proofs have the oracle's bytes, NO proof is
    re-made (a gate that ran into its 4-s limit or had to be let go would count), every proof takes well under a second, and
    the gate is back at work by the second proof at the latest."""
    import time

    import ctypes

    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 9)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    want = {s: orc.create_proof(pk, adv, inst, s)[1] for s in range(3)}
    short = inst[:, : max(1, ilen - 1), :] if ilen > 1 else inst
    want_short = orc.create_proof(pk, adv, short, 1)[1]
    prover.set_overlap("tables")
    d_adv = dev(adv)
    torch.cuda.synchronize()
    state = {"n": 0}

    def proofs(count=2, instance=None, expect=None):
        """`count` lone proofs; returns how many of them were gated"""
        before = prover.gate_stats()
        for i in range(count):
            seed = i % 3
            t0 = time.perf_counter()
            got = prover.prove_dev(d_adv.data_ptr(), inst if instance is None else instance, seed if expect is None else 1)
            dt = time.perf_counter() - t0
            assert got == (want[seed] if expect is None else expect), state
            assert dt < 1.0, (state, dt)  # (a gate's time limit is 4 s)
            state["n"] += 1
        after = prover.gate_stats()
        assert after["remade_plain"] == before["remade_plain"] and after["yields"] == before["yields"], (state, before, after)
        return after["gated_proofs"] - before["gated_proofs"]

    try:
        zg.tuning_set("ZG_LAT_GATE", 1)
        assert proofs(3) == 2  # first proof plain, then gated
        # --- zg_prover_fetch* between proofs: synchronises an idle stream, allocates nothing the next proof asks for
        state["step"] = "fetch"
        prover.fetch(1, 0, 1 << 9)
        assert proofs(2) == 2
        # --- per-launch profiling events switched on and collected: events come from a pool, the collect synchronises between proofs
        state["step"] = "profile"
        ctx.profile(True)
        assert proofs(2) == 2
        assert ctx.profile_collect()["gate_pull"][0] == 10
        ctx.profile(False)
        # --- another user of the SAME context's workspace pool and pinned arena between proofs: a stand-alone MSM larger than
        #     anything the proof asks for (the pool only grows; blocks are matched by size), and a transform at a NEW size (a
        #     twiddle table of this device, made with a synchronisation -- between proofs)
        state["step"] = "workspace"
        g = pk.params.g_np() if hasattr(pk, "params") else None
        big = orc.fill_fr(3, 1 << 12)
        om, omi = zg.domain_omega(12)
        ctx.ntt(big, om)
        if g is not None:
            bases = ctx.register_bases(g)
            ctx.msm_batch(bases, np.stack([orc.fill_fr(5 + i, 1 << 9) for i in range(24)]))
            bases.free()
        assert proofs(2) == 2
        # --- a shorter instance: another signature, so a first proof again, then gated
        state["step"] = "instance_len"
        assert proofs(2, instance=short, expect=want_short) == (1 if ilen > 1 else 2)
        assert proofs(2) == (1 if ilen > 1 else 2)
        # --- zg_tuning_set of knobs that resize workspace or change the launch sequence: first proof again
        for knob, value in (("ZG_LAT_FULL_K", 8), ("ZG_MSM_K_LAT", 24), ("ZG_LAZY_DOT", 0), ("ZG_LAT_PULL", 0)):
            state["step"] = knob
            zg.tuning_set(knob, value)
            assert proofs(3) == 2
            zg.tuning_set(knob, -1)
            assert proofs(2) == 1
        # --- zg_prover_set_batch: the slots are reallocated (a free is a device synchronisation) -- first proof again
        state["step"] = "set_batch"
        prover.set_batch(3)
        assert proofs(3) == 2
        prover.set_batch(1)
        assert proofs(2) == 1
        # --- a fork of the warm prover on a context of its own: its first proof is a first proof, the parent stays warm
        state["step"] = "fork"
        ctx2 = zg.Ctx(0)
        child = prover.fork(ctx2)
        child.set_overlap(True)
        c0 = child.gate_stats()
        assert child.prove_dev(d_adv.data_ptr(), inst, 0) == want[0] and child.prove_dev(d_adv.data_ptr(), inst, 1) == want[1]
        c1 = child.gate_stats()
        assert c1["gated_proofs"] - c0["gated_proofs"] == 1 and c1["remade_plain"] == 0 and c1["yields"] == 0
        assert proofs(2) == 2
        child.close()
        ctx2.close()
        # --- the digit tables go (window form): another signature
        state["step"] = "overlap"
        prover.set_overlap(False)
        assert proofs(2) == 0
        prover.set_overlap(True)
        assert proofs(3) == 2
        total = prover.gate_stats()
        assert total["remade_plain"] == 0 and total["yields"] == 0 and total["gates_armed"] == 5 * total["gated_proofs"]
    finally:
        zg.tuning_set("ZG_LAT_GATE", -1)
    prover.close()


def test_a_blocking_call_behind_a_gate_lets_it_go_at_once(ctx, zg, orc):
    """The mechanism behind the enumeration: should a phase that was queued ahead of its challenge still have to allocate
    (here: the workspace pool of the prover's context is emptied behind its back by ZG_TEST_DROP_WORKSPACE), the gate is
    opened at once -- no 4-s stall --, the proof is re-made in the plain order and its bytes are right."""
    import time

    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 9)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    want = orc.create_proof(pk, adv, inst, 1)[1]
    prover.set_overlap(True)
    try:
        zg.tuning_set("ZG_LAT_GATE", 1)
        assert prover.prove(adv, inst, 1) == want and prover.prove(adv, inst, 1) == want
        assert prover.gate_stats()["gated_proofs"] == 1
        ctx.drop_workspace()  # (every free block of the pool is released: the next proof's requests must allocate)
        t0 = time.perf_counter()
        assert prover.prove(adv, inst, 1) == want
        dt = time.perf_counter() - t0
        st = prover.gate_stats()
        assert st["gated_proofs"] == 2 and st["yields"] >= 1 and st["remade_plain"] == 1, st
        assert dt < 1.0, dt
        assert prover.prove(adv, inst, 1) == want
        assert prover.gate_stats()["remade_plain"] == 1
    finally:
        zg.tuning_set("ZG_LAT_GATE", -1)
    prover.close()


@pytest.mark.parametrize("force_degree,parts", [(None, "one coset: 4n is exact"), (6, "4n + n"), (7, "4n + 2n"), (8, "one coset: 7 is no sum of two powers of two")])
def test_split_extended_domain_gives_the_same_bytes(ctx, zg, orc, force_degree, parts):
    """Throughput form (overlap off): the quotient comes from two cosets holding (degree - 1) * n points
    between them instead of one coset of 2^ext_k; same polynomial, same proof bytes."""
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 7, force_degree=force_degree)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    st, want, _ = orc.create_proof(pk, adv, inst, 11)
    assert st == 0
    for overlap in (False, True, "tables"):
        prover.set_overlap(overlap)
        assert prover.prove(adv, inst, 11) == want, parts
    assert orc.verify_proof_pairing(pk, inst, want) == 1
    prover.close()


def test_prover_reuse_with_a_shorter_instance(ctx, zg, orc):
    """The instance column lives on the device between proofs and only its first rows are refilled: a proof
    with fewer instance values after one with more must see zeros behind them (bytes == oracle each time; the
    shorter statement is not satisfied, which changes no code path of the prover)."""
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 7)
    adv = asg.advice_values()
    full, none = asg.instance_values(ilen), asg.instance_values(0)
    for inst, seed in ((full, 3), (none, 4), (full, 5)):
        st, want, _ = orc.create_proof(pk, adv, inst, seed)
        assert st == 0
        assert prover.prove(adv, inst, seed) == want
    prover.close()


def test_lookup_failure_is_constraint_system_failure(ctx, zg, orc):
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 5)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    bad = adv.copy()
    bad[1, 2] = orc.fr_from_int(1000)
    with pytest.raises(zg.ZgError) as e:
        prover.prove(bad, inst, 1)
    assert e.value.status == -5
    assert orc.create_proof(pk, bad, inst, 1)[0] == -5
    # the prover object stays usable
    assert prover.prove(adv, inst, 3) == orc.create_proof(pk, adv, inst, 3)[1]
    prover.close()


def test_unsatisfied_witness_still_proves_but_does_not_verify(ctx, zg, orc):
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, 5)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    bad = adv.copy()
    bad[2, 3] = orc.fr_from_int(99)
    got = prover.prove(bad, inst, 1)
    assert got == orc.create_proof(pk, bad, inst, 1)[1]
    assert orc.verify_proof(pk, inst, got) != 1
    prover.close()


@pytest.mark.parametrize("latency_form", [True, False])
@pytest.mark.parametrize("n", [1, 2, 5, 1000, 1024, 1025, 1 << 14, (1 << 17) + 3])
def test_grand_product(ctx, zg, orc, n, latency_form):
    """Both forms of the scan: block-local Hillis-Steele (latency) and one strip per lane (throughput)."""
    num, den = orc.fill_fr(1, n), orc.fill_fr(2, n)
    if n > 4:
        den[3] = 0  # BatchInvert leaves zeros alone -> that ratio is zero
    if n > 2000:
        den[1500] = 0
    z0 = orc.fill_fr(3, 1)[0]
    dn, dd = dev(num), dev(den)
    dz = torch.empty_like(dn)
    ctx.set_msm_latency(latency_form)
    try:
        ctx.grand_product_dev(dn.data_ptr(), dd.data_ptr(), z0, n, dz.data_ptr())
        assert np.array_equal(host(dz), orc.grand_product(num, den, z0))
        assert np.array_equal(ctx.grand_product(num, den, z0), orc.grand_product(num, den, z0))  # host-pointer entry
    finally:
        ctx.set_msm_latency(True)


@pytest.mark.parametrize("latency_form", [True, False])
@pytest.mark.parametrize("n", [1, 7, 1 << 10, (1 << 10) + 5, 1 << 14, (1 << 15) + 1])
def test_eval_polys_and_kate_division(ctx, zg, orc, n, latency_form):
    polys = np.stack([orc.fill_fr(10 + i, n) for i in range(3)])
    pts = orc.fill_fr(99, 4)
    d = dev(polys)
    idx = [0, 2, 1, 2]
    got = ctx.eval_polys_dev(d.data_ptr(), n, n, idx, pts)
    for j, pi in enumerate(idx):
        assert np.array_equal(got[j], orc.eval_poly(polys[pi], pts[j]))
    dq = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.set_msm_latency(latency_form)
    try:
        ctx.kate_division_dev(d.data_ptr(), n, pts[0], dq.data_ptr())
    finally:
        ctx.set_msm_latency(True)
    assert np.array_equal(host(dq), orc.kate_division(polys[0], pts[0]))


def test_wnn_shaped_circuit_parity_k12_and_verify_k14(ctx, zg, orc):
    """zero_g's circuit shape: byte parity with the oracle at k = 12, and at the BASELINE size
    (model_28input_256entry_1hash_1bpi, k = 14, extended domain 2^17) a proof the oracle's
    verifier accepts, deterministic in the seed."""
    import wnn_shape

    vk_repr = orc.fr_from_int(0xC0FFEE)
    for k, full_parity in ((12, True), (14, False)):
        cs, asg, ilen = wnn_shape.build("tiny", k=k, seed=k)
        img = cs.to_c()
        params = orc.params_new(k, 0x5EED)
        fixed, sigma = asg.fixed_values(), asg.sigma_values()
        pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
        prover = zg.Prover(ctx, img, fixed, sigma, params.g_np(), params.g_lagrange_np(), vk_repr)
        adv, inst = asg.advice_values(), asg.instance_values(ilen)
        got = prover.prove(adv, inst, 9)
        assert len(got) == 64 * 30 + 32 * (10 + len(cs.fixed_queries) + 1 + 8 + 5 + 20)
        assert prover.prove(adv, inst, 9) == got
        if full_parity:
            st, want, _ = orc.create_proof(pk, adv, inst, 9)
            assert st == 0 and got == want
        assert orc.verify_proof_pairing(pk, inst, got) == 1
        bad = bytearray(got)
        bad[100] ^= 4
        assert orc.verify_proof(pk, inst, bytes(bad)) != 1
        prover.close()


def test_wnn_shaped_circuit_k15_verifies(ctx, zg, orc):
    """BASELINE configs[2] size (model_28input_1024entry_2hash_2bpi shape: k = 15, extended domain 2^18):
    the GPU proof is accepted by the oracle's verifier; shared base tables give the same bytes."""
    import wnn_shape

    k = 15
    vk_repr = orc.fr_from_int(0xC0FFEE)
    cs, asg, ilen = wnn_shape.build("small", k=k, seed=5)
    img = cs.to_c()
    params = orc.params_new(k, 0x5EED)
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    g, gl = params.g_np(), params.g_lagrange_np()
    prover = zg.Prover(ctx, img, fixed, sigma, g, gl, vk_repr)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    got = prover.prove(adv, inst, 11)
    assert orc.verify_proof_pairing(pk, inst, got) == 1
    prover.close()
    gb, glb = ctx.register_bases(g), ctx.register_bases(gl)
    shared = zg.Prover(ctx, img, fixed, sigma, gb, glb, vk_repr)
    assert shared.prove(adv, inst, 11) == got
    shared.close()
    gb.free()
    glb.free()


@pytest.mark.parametrize("kind", ["no_lookup", "gates_only", "wide_lookup", "advice_factor", "merged_selectors"])
def test_circuit_variants_match_oracle(ctx, zg, orc, kind):
    """No-lookup / no-permutation / several-instance-column / width-2-lookup paths of the prover."""
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
    for seed, overlap in ((1, True), (2, False)):  # (latency form, then the throughput form with its split domain)
        prover.set_overlap(overlap)
        st, want, _ = orc.create_proof(pk, adv, inst, seed)
        assert st == 0
        got = prover.prove(adv, inst, seed)
        assert got == want
        assert orc.verify_proof_pairing(pk, inst, got) == 1
    prover.close()


def _real_model(orc, zg, ctx, which):
    import wnn_circuit
    import wnn_model

    k, name = which
    cs, asg, ilen, scores = wnn_circuit.build(wnn_model.load_checked_in(name), wnn_model.load_test_image(), k)
    img = cs.to_c()
    params = orc.params_new(k, 0x5EED)
    vk_repr = orc.fr_from_int(0xC0FFEE)
    pk = orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, vk_repr)
    prover = zg.Prover(ctx, img, asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), vk_repr)
    return cs, asg, ilen, scores, pk, prover


def _gated_twice(zg, prover, adv, inst, seed, want):
    """the GATED schedule's bytes at the sizes the bench quotes a lone proof at (k = 14 single coset, k = 15 / 17 split
    domain).  The prover's current latency form, ZG_LAT_GATE = 1: the first proof after the knob is plain, the next two are
    gated (five gates each, none let go, none re-made) and have the oracle's bytes."""
    zg.tuning_set("ZG_LAT_GATE", 1)
    try:
        assert prover.prove(adv, inst, seed) == want
        before = prover.gate_stats()
        assert prover.prove(adv, inst, seed) == want and prover.prove(adv, inst, seed) == want
        after = prover.gate_stats()
        assert {k_: after[k_] - before[k_] for k_ in after} == {"gated_proofs": 2, "gates_armed": 10, "remade_plain": 0, "yields": 0}
    finally:
        zg.tuning_set("ZG_LAT_GATE", -1)


def test_real_wnn_circuit_tiny_proof_bytes_match_oracle(ctx, zg, orc):
    """zero_g's WnnCircuit for model_28input_256entry_1hash_1bpi on example_image_7 (BASELINE configs[1]):
    the GPU proof is byte-identical to the oracle's and satisfies the public pairing equation."""
    import wnn_model

    orc.load().orc_set_threads(16)
    cs, asg, ilen, scores, pk, prover = _real_model(orc, zg, ctx, wnn_model.MNIST_TINY)
    assert scores == [9, 6, 13, 10, 17, 10, 9, 26, 11, 16]
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    # both configurations: overlap on = one 8n-point coset, two lanes per EC addition; overlap off (the bench's
    # throughput form) = the split 4n + n extended domain, one lane per addition
    # ("tables": zg_prover_enable_digit_tables first -- 78 GB at k = 14 -- then the lone proof as a flat sum of table points)
    for seed, overlap in ((1, True), (99, False), (7, "tables")):
        prover.set_overlap(overlap)
        got = prover.prove(adv, inst, seed)
        st, want, _ = orc.create_proof(pk, adv, inst, seed)
        assert st == 0 and got == want
        if overlap:  # (the gated order of the same form: plain latency form, then over the digit tables)
            _gated_twice(zg, prover, adv, inst, seed, want)
    assert orc.verify_proof_pairing(pk, inst, got) == 1
    wrong = inst.copy()
    wrong[0, 0] = orc.fr_from_int(scores[0] + 1)
    assert orc.verify_proof_pairing(pk, wrong, got) != 1
    prover.close()


def test_real_wnn_circuit_small_k15_verifies(ctx, zg, orc):
    """model_28input_1024entry_2hash_2bpi (BASELINE configs[2], k = 15): GPU proof bytes == oracle, pairing verifier."""
    import wnn_model

    orc.load().orc_set_threads(16)
    cs, asg, ilen, scores, pk, prover = _real_model(orc, zg, ctx, wnn_model.MNIST_SMALL)
    assert scores == [17, 13, 25, 27, 29, 21, 15, 55, 27, 32]
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    got = prover.prove(adv, inst, 5)
    st, want, _ = orc.create_proof(pk, adv, inst, 5)
    assert st == 0 and got == want  # byte parity at k = 15 as well
    assert orc.verify_proof_pairing(pk, inst, got) == 1
    prover.close()


def test_real_wnn_circuit_medium_k15_verifies(ctx, zg, orc):
    """model_28input_2048entry_2hash_3bpi (BASELINE configs[3], k = 15; the layout fills 32 738 of the 32 762
    usable rows): GPU proof bytes == oracle in both schedules (split 4n + n domain / single 8n coset), pairing verifier."""
    import wnn_model

    orc.load().orc_set_threads(16)
    cs, asg, ilen, scores, pk, prover = _real_model(orc, zg, ctx, wnn_model.MNIST_MEDIUM)
    assert scores == [29, 21, 40, 47, 45, 41, 28, 82, 35, 66]
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    st, want, _ = orc.create_proof(pk, adv, inst, 11)
    assert st == 0
    prover.set_overlap(False)
    assert prover.prove(adv, inst, 11) == want
    prover.set_overlap(True)
    assert prover.prove(adv, inst, 11) == want
    _gated_twice(zg, prover, adv, inst, 11, want)  # (k = 15: the lone proof takes the split domain, ZG_LAT_SPLIT_K)
    prover.set_overlap("tables")  # (k = 15: c = 10, 84 GB)
    assert prover.prove(adv, inst, 11) == want
    _gated_twice(zg, prover, adv, inst, 11, want)
    assert orc.verify_proof_pairing(pk, inst, want) == 1
    prover.close()


def test_large_shape_k17_verifies(ctx, zg, orc):
    """The k = 17 configuration (BASELINE configs[4]; seeded stand-in for the absent 49-input model): 2^17 rows,
    2^20-point extended domain, 15-bit MSM windows -- GPU proof bytes == oracle in both schedules and in a batch of two,
    pairing verifier."""
    import wnn_circuit
    import wnn_model

    orc.load().orc_set_threads(16)
    k = wnn_model.MNIST_LARGE[0]
    wnn = wnn_model.synthetic_wnn()
    cs, asg, ilen, scores = wnn_circuit.build(wnn, wnn_model.load_test_image(), k)
    assert scores == wnn.predict(wnn_model.load_test_image())
    img = cs.to_c()
    params = orc.params_new(k, 0x5EED)
    vk_repr = orc.fr_from_int(0xC0FFEE)
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    prover = zg.Prover(ctx, img, fixed, sigma, params.g_np(), params.g_lagrange_np(), vk_repr)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    st, want, _ = orc.create_proof(pk, adv, inst, 3)
    assert st == 0
    assert prover.prove(adv, inst, 3) == want          # latency form, side stream (split domain from k = 15 on)
    _gated_twice(zg, prover, adv, inst, 3, want)       # ... and its gated order (no digit tables at this size: buckets)
    prover.set_overlap(False)
    assert prover.prove(adv, inst, 3) == want          # split 4n + n domain
    prover.set_batch(2)
    got, _ = prover.prove_batch([adv, adv], [inst, inst], [3, 4])
    assert got[0] == want and got[1] == orc.create_proof(pk, adv, inst, 4)[1]
    prover.close()
    assert orc.verify_proof_pairing(pk, inst, want) == 1


@pytest.mark.parametrize("k,force_degree", [(6, None), (7, 6), (12, 6), (14, None)])
def test_stand_alone_evaluate_h_matches_oracle(ctx, zg, orc, k, force_degree):
    """zg_prover_evaluate_h (the arithmetic-level entry for Evaluator::evaluate_h + the division by X^n - 1): fed the
    coefficient forms of the oracle's own witness-side polynomials and its challenges, it must return the oracle's h on
    the extended coset."""
    cs, asg, ilen, pk, prover = setup(orc, zg, ctx, k, force_degree=force_degree)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    seed = 21
    st, _, tr = orc.create_proof(pk, adv, inst, seed, want_trace=True)
    assert st == 0
    n, en = 1 << k, 1 << cs.extended_k()
    bf = cs.blinding_factors()
    usable = n - (bf + 1)
    blinded = adv.copy()
    for c in range(cs.n_advice):  # create_proof's advice blinding: tag 1, index = column * (bf + 1) + j
        for j in range(bf + 1):
            blinded[c, usable + j] = orc.rand_fr(seed, 1, c * (bf + 1) + j)
    inst_cols = np.zeros((cs.n_instance, n, 4), np.uint64)
    inst_cols[:, :ilen] = inst
    d = orc.domain(cs.degree(), k)
    nl, sets = len(cs.lookups), tr.n_sets
    pin = tr.array("permuted_input", nl * n).reshape(nl, n, 4)
    ptab = tr.array("permuted_table", nl * n).reshape(nl, n, 4)
    permuted = np.stack([x for l in range(nl) for x in (pin[l], ptab[l])]) if nl else np.zeros((0, n, 4), np.uint64)

    def coeffs(cols):
        return np.stack([orc.lagrange_to_coeff(d, c) for c in cols]) if len(cols) else np.zeros((0, n, 4), np.uint64)

    got = prover.evaluate_h(coeffs(blinded), coeffs(inst_cols), coeffs(tr.array("perm_z", sets * n).reshape(sets, n, 4)),
                            coeffs(tr.array("lookup_z", nl * n).reshape(nl, n, 4)), coeffs(permuted), tr.fe("theta"),
                            tr.fe("beta"), tr.fe("gamma"), tr.fe("y"), en)
    assert np.array_equal(got, tr.array("h_ext", en))
    orc.trace_free(tr)
    # the prover still proves afterwards (slot 0 was used as scratch)
    assert prover.prove(adv, inst, 5) == orc.create_proof(pk, adv, inst, 5)[1]
    prover.close()
