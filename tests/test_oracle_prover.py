"""The oracle's create_proof restatement must produce proofs that its verify_proof restatement
accepts -- the only property the reference's own tests pin for this path (a proof verifies:
/root/reference/src/lib.rs:10-33, test_cli.sh:62-82) -- and reject tampered ones."""
import numpy as np
import pytest

from circuits import toy_circuit


def make_pk(orc, cs, asg, params, derive=True):
    img = cs.to_c()
    vk_repr = orc.fr_from_int(0x1234567)
    return orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, vk_repr, derive=derive)


@pytest.fixture(scope="module")
def params5(orc):
    return orc.params_new(5, 0xABCDEF)


def test_keccak_kat(orc):
    assert orc.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert orc.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # multi-block input (rate = 136 bytes)
    import hashlib
    assert orc.keccak256(b"a" * 136) != orc.keccak256(b"a" * 135)
    assert len(orc.keccak256(b"x" * 1000)) == 32


@pytest.mark.parametrize("force_degree", [None, 6, 8])
def test_toy_proof_verifies(orc, params5, force_degree):
    cs, asg, ilen = toy_circuit(5, force_degree=force_degree)
    asg.check()
    pk = make_pk(orc, cs, asg, params5)
    inst = asg.instance_values(ilen)
    st, proof, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=7)
    assert st == 0 and len(proof) > 0
    assert orc.verify_proof(pk, inst, proof) == 1 and orc.verify_proof_pairing(pk, inst, proof) == 1
    # deterministic in the seed, different for another seed
    st2, proof2, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=7)
    assert proof2 == proof
    st3, proof3, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=8)
    assert proof3 != proof and orc.verify_proof(pk, inst, proof3) == 1
    # the key's polys / cosets / l0, l_last, l_active_row kept with the key (keygen_pk) or computed inside the proof: same bytes
    assert pk.c.derived
    bare = make_pk(orc, cs, asg, params5, derive=False)
    assert orc.create_proof(bare, asg.advice_values(), inst, seed=7)[1] == proof and not bare.c.derived


def test_tampered_proofs_and_wrong_instance_fail(orc, params5):
    cs, asg, ilen = toy_circuit(5)
    pk = make_pk(orc, cs, asg, params5)
    inst = asg.instance_values(ilen)
    st, proof, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=1)
    assert orc.verify_proof(pk, inst, proof) == 1 and orc.verify_proof_pairing(pk, inst, proof) == 1
    # flip one bit in an evaluation (scalars start after the commitments) and in a commitment
    for pos in (len(proof) - 64 * 4 - 5, len(proof) - 10, 40):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert orc.verify_proof(pk, inst, bytes(bad)) != 1
    wrong = inst.copy()
    wrong[0, 0] = orc.fr_from_int(12345)
    assert orc.verify_proof(pk, wrong, proof) != 1
    assert orc.verify_proof(pk, inst, proof[:-1]) != 1
    assert orc.verify_proof(pk, inst, proof + b"\0") != 1


def test_unsatisfied_witness_is_rejected(orc, params5):
    cs, asg, ilen = toy_circuit(5)
    pk = make_pk(orc, cs, asg, params5)
    inst = asg.instance_values(ilen)
    adv = asg.advice_values()
    bad = adv.copy()
    bad[2, 3] = orc.fr_from_int(99)  # breaks a0*a1 = a2 on row 3 (and a copy constraint)
    st, proof, _ = orc.create_proof(pk, bad, inst, seed=1)
    assert st == 0  # the prover does not check gates (halo2 does not either)
    assert orc.verify_proof(pk, inst, proof) != 1
    # a lookup input outside the table -> ConstraintSystemFailure (-5), like halo2
    bad = adv.copy()
    bad[1, 2] = orc.fr_from_int(1000)
    st, proof, _ = orc.create_proof(pk, bad, inst, seed=1)
    assert st == -5


def test_wnn_shaped_circuit_proof_verifies(orc):
    """The circuit of zero_g's shape (12 gates, 4 lookups, 8 equality columns,
    degree 6) at a reduced k: witness satisfies every constraint, proof verifies."""
    import wnn_shape

    cs, asg, ilen = wnn_shape.build("tiny", k=12, seed=3)
    asg.check()
    assert (cs.n_fixed, cs.n_advice, cs.n_instance) == (23, 6, 1)
    assert cs.degree() == 6 and cs.extended_k() == 15 and cs.blinding_factors() == 5
    assert len(cs.gates) == 12 and len(cs.lookups) == 4 and len(cs.perm_columns) == 8
    assert sorted(cs.advice_queries) == sorted([(0, 0), (0, 1), (1, 0), (2, 0), (3, 0), (4, 0), (4, 1),
                                                (5, -1), (5, 0), (5, 1)])
    params = orc.params_new(12, 0x5EED)
    pk = make_pk(orc, cs, asg, params)
    inst = asg.instance_values(ilen)
    st, proof, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=5)
    assert st == 0
    assert orc.create_proof(make_pk(orc, cs, asg, params, derive=False), asg.advice_values(), inst, seed=5)[1] == proof
    # 6 advice + 8 permuted + 2 perm z + 4 lookup z + 1 random + 5 h + 4 W = 30 points
    n_scalars = 10 + len(cs.fixed_queries) + 1 + 8 + 5 + 20
    assert len(proof) == 64 * 30 + 32 * n_scalars
    assert orc.verify_proof(pk, inst, proof) == 1 and orc.verify_proof_pairing(pk, inst, proof) == 1


@pytest.mark.parametrize("kind", ["no_lookup", "gates_only", "wide_lookup", "advice_factor", "merged_selectors"])
def test_circuit_variants_verify(orc, params5, kind):
    from circuits import variant_circuit

    cs, asg, ilen = variant_circuit(kind)
    asg.check()
    pk = make_pk(orc, cs, asg, params5)
    inst = asg.instance_values(ilen)
    st, proof, _ = orc.create_proof(pk, asg.advice_values(), inst, seed=2)
    assert st == 0
    assert orc.verify_proof(pk, inst, proof) == 1 and orc.verify_proof_pairing(pk, inst, proof) == 1
    if ilen:
        wrong = inst.copy()
        wrong[0, 0] = orc.fr_from_int(424242)
        assert orc.verify_proof(pk, wrong, proof) != 1
    bad = bytearray(proof)
    bad[len(proof) // 2] ^= 0x10
    assert orc.verify_proof(pk, inst, bytes(bad)) != 1
