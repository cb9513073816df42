"""The artefact flow of zero_g's CLI around the proving path (/root/reference/test_cli.sh:44-82: generate-srs ->
generate-keys -> proof -> verify), with the files of 0g-halo2_amd/formats.py in the loop: everything the backend
consumes is READ BACK from the SRS file, the proving-key file and the circuit-parameter file, the proof leaves as a
ProofWithOutput file, and the verifier side (the oracle's pairing check of `verify_proof`) works from that file alone.
Layouts are from the published crates (formats.py docstring): the files are this repo's own, not the real CLI's."""
import numpy as np
import pytest
import torch  # (before the first HIP call of the process)

pytestmark = pytest.mark.gpu


def test_srs_keys_proof_verify_through_files(tmp_path, ctx, zg, orc):
    import formats
    import witness_tape
    import wnn_circuit
    import wnn_model

    orc.load().orc_set_threads(16)
    k, name = wnn_model.MNIST_TINY
    wnn = wnn_model.load_checked_in(name)
    image = wnn_model.load_test_image()

    # ---- `generate-srs` / `generate-keys`: params, circuit parameters and the proving key onto disk
    params = orc.params_new(k, 0x5EED)
    srs_path, pk_path, cp_path = (str(tmp_path / f) for f in ("srs.bin", "pk.bin", "circuit_params.json"))
    g2_0, s_g2_0 = (np.frombuffer(bytes(v), np.uint64) for v in (params.g2, params.s_g2))
    formats.write_srs(srs_path, k, params.g_np(), params.g_lagrange_np(), g2_0, s_g2_0)
    formats.write_circuit_params(cp_path, wnn.get_circuit_params())
    cs, asg0, ilen, _ = wnn_circuit.build(wnn, np.zeros_like(image), k)  # keygen synthesises a zero image (wnn.rs:222-229)
    n, en = 1 << k, 1 << cs.extended_k()
    fixed, sigma = asg0.fixed_values(), asg0.sigma_values()
    F, P = fixed.shape[0], sigma.shape[0]
    z = lambda *shape: np.zeros(shape + (4,), np.uint64)  # (families the backend derives itself: placeholders of the right shape)
    formats.write_pk(pk_path, formats.ProvingKeyFile(
        k, fixed_commitments=np.zeros((F, 8), np.uint64), permutation_commitments=np.zeros((P, 8), np.uint64),
        selectors=np.zeros((0, n), bool), l0=z(en), l_last=z(en), l_active_row=z(en), fixed_values=fixed, fixed_polys=z(F, n),
        fixed_cosets=z(F, en), permutations=sigma, permutation_polys=z(P, n), permutation_cosets=z(P, en)))

    # ---- `proof`: a fresh reader of those files
    k2, g, g_lagrange, g2, s_g2 = formats.read_srs(srs_path)
    cp = formats.read_circuit_params(cp_path)
    assert k2 == k and cp == wnn.get_circuit_params()
    assert np.array_equal(g2, g2_0) and np.array_equal(s_g2, s_g2_0)  # (the verifier below pairs against these two)
    pkf = formats.read_pk(pk_path, 0, P)
    vk_repr = orc.fr_from_int(0xC0FFEE)  # (vk.transcript_repr comes from the Rust side: formats.py)
    prover = zg.Prover(ctx, cs.to_c(), pkf.fixed_values, pkf.permutations, np.array(g), np.array(g_lagrange), vk_repr)
    plan = zg.WitnessPlan(ctx, witness_tape.trace(wnn, k).arrays())
    proofs, outputs, sts = prover.prove_images(plan, image[None], [2024])
    assert sts == [0]
    scores = [zg.fr_to_int(x) for x in outputs[0]]
    assert scores == [9, 6, 13, 10, 17, 10, 9, 26, 11, 16]  # /root/reference/tests/integration_test.rs:19
    for form in ("limbs", "hex"):
        proof_path = str(tmp_path / f"proof_{form}.json")
        formats.ProofWithOutput(proofs[0], scores).write(proof_path, form=form)

        # ---- `verify`: proof bytes and public output from the file, key material from the key files
        back = formats.ProofWithOutput.read(proof_path)
        assert back.proof == proofs[0] and back.output == scores
        pk = orc.ProvingKey(cs.to_c(), pkf.fixed_values, pkf.permutations, params, vk_repr)
        assert orc.verify_proof_pairing(pk, back.output_mont(), back.proof) == 1
        wrong = formats.ProofWithOutput(back.proof, [scores[1]] + scores[1:])
        assert orc.verify_proof_pairing(pk, wrong.output_mont(), back.proof) != 1
    plan.close()
    prover.close()
