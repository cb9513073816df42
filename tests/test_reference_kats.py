"""The reference's OWN small-circuit known-answer tests, reproduced on the caller mirror (harness/wnn_model.py,
harness/wnn_circuit.py) -- the component that produces every witness the GPU proves and records the device witness
program.  Every number below is reference-held data:

  * /root/reference/src/gadgets/wnn.rs:412-488 -- a 4x3 image, two thresholds per pixel, 2 classes; the comments there
    state the thermometer bits, the permuted bits, the filter indices 2237 / 3788, the MishMash hashes 825286 / 47598,
    the bloom indices 966 / 805 and 494 / 46, and the test asserts the class scores [1, 2] at k = 13;
  * /root/reference/src/gadgets/hash.rs:323-371 -- HashChip with p = 11, l = 3, n_bits = 8 at k = 9: 2 -> 0, 4 -> 1,
    42 -> 3, 255 -> 0;
  * /root/reference/src/gadgets/range_check.rs:230-288 -- le_constant: 1023 <= 1023, 1022 <= 1023, 4 <= 9,
    0 <= 0xffabcdef hold, 1024 <= 1023 fails verification;
  * /root/reference/src/gadgets/bloom_filter/byte_selector.rs:476-516 -- select_byte: 0xab[0] = 0xab, 0xabcdef[0] = 0xab,
    0xabcdef[1] = 0xcd (big-endian index) and a wrong public output is rejected.

`Assignment.check()` is the mirror's MockProver::assert_satisfied.  The -m gpu case replays the recorded witness program of
the 4x3 model on the device (zg_witness_run_dev) and proves it: same cells, same scores, proof == oracle."""
import numpy as np
import pytest
import torch  # noqa: F401  (before anything loads the HIP library: tests/conftest.py)

import wnn_circuit as wc
import wnn_model as wm
from circuit import ADVICE, FIXED, INSTANCE, ConstraintSystem
from layouter import Layouter

# ---- /root/reference/src/gadgets/wnn.rs:412-470, verbatim data
IMAGE = np.array([[70, 100, 150], [20, 110, 200], [27, 50, 211], [200, 100, 3]], dtype=np.uint8)
THRESHOLDS = np.array([[[50, 150], [0, 50], [200, 256]],
                       [[10, 80], [100, 200], [50, 150]],
                       [[0, 100], [100, 200], [0, 100]],
                       [[0, 100], [100, 200], [0, 100]]], dtype=np.uint16)
PERMUTATION = np.array([6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 0, 1, 2, 3, 4, 5], dtype=np.uint64)
BITS = [1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1,  # first threshold
        0, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 0]  # second threshold
PERMUTED_BITS = [1, 0, 1, 1, 1, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1]
K_WNN = 13


def reference_wnn() -> wm.Wnn:
    bloom = np.zeros((2, 2, 1024), dtype=bool)
    for c, f, e in [(0, 0, 966), (0, 0, 805), (0, 1, 494), (1, 0, 966), (1, 0, 805), (1, 1, 494), (1, 1, 46)]:
        bloom[c, f, e] = True
    # PARAMS of the test: p = (1 << 21) - 9, l = 20, n_hashes = 2, bits_per_hash = 10, bits_per_filter = 12, n_classes = 2
    return wm.Wnn(num_classes=2, num_filter_entries=1024, num_filter_hashes=2, num_filter_inputs=12, p=2097143,
                  bloom_filters=bloom, input_order=PERMUTATION, binarization_thresholds=THRESHOLDS)


def test_wnn_model_intermediate_values_are_the_references():
    wnn = reference_wnn()
    p = wnn.get_circuit_params()
    assert (p.p, p.l, p.n_hashes, p.bits_per_hash, p.bits_per_filter, p.n_classes) == (2097143, 20, 2, 10, 12, 2)
    bits = wnn.thermometer_encoding(IMAGE).astype(int).tolist()
    assert bits == BITS
    assert [bits[int(i)] for i in PERMUTATION] == PERMUTED_BITS
    assert wnn.encode_image(IMAGE) == [2237, 3788]
    assert [wnn.mish_mash_hash(x) for x in (2237, 3788)] == [825286, 47598]
    assert (2237 ** 3) % 2097143 % (1 << 20) == 825286 and (3788 ** 3) % 2097143 % (1 << 20) == 47598
    assert wnn.hash_indices(2237) == [966, 805] == [825286 % 1024, 825286 // 1024]
    assert wnn.hash_indices(3788) == [494, 46] == [47598 % 1024, 47598 // 1024]
    # one positive response for the first class, two for the second
    assert [[wnn.bloom_filter_lookup(wnn.bloom_filters[c, f], v) for f, v in enumerate((2237, 3788))] for c in range(2)] == \
        [[True, False], [True, True]]
    assert wnn.predict(IMAGE) == [1, 2]


def test_wnn_circuit_mock_proof_at_the_references_k():
    """`MockProver::run(13, &circuit, vec![vec![1, 2]])` + assert_satisfied (wnn.rs:478-488), and the values the chips lay
    out on the way: filter indices, hashes, bloom indices and responses as cells of the assignment."""
    wnn = reference_wnn()
    circuit = wc.WnnCircuit(wnn, K_WNN)
    asg, scores = circuit.synthesize(IMAGE)
    assert scores == [1, 2]
    asg.check()
    assert asg.instance[0][:2] == [1, 2]
    assert circuit.rows_used <= circuit.cs.usable_rows()
    hash_cfg = circuit.config["hash"]
    rows = sorted(asg.selectors[hash_cfg["selector"]])
    assert len(rows) == 2  # one hash region per filter
    assert [asg.advice[hash_cfg["input"]][r] for r in rows] == [2237, 3788]
    assert [asg.advice[hash_cfg["hash"]][r] for r in rows] == [825286, 47598]
    for r, x in zip(rows, (2237, 3788)):
        q, rem, msb = (asg.advice[hash_cfg[c]][r] for c in ("quotient", "remainder", "msb"))
        assert q * 2097143 + rem == x ** 3 and rem < 2097143 and rem == (msb << 20) + (x ** 3 % 2097143 % (1 << 20))
    # a wrong witness is not satisfied: the hash of the second filter altered
    asg.advice[hash_cfg["hash"]][rows[1]] += 1
    with pytest.raises(AssertionError):
        asg.check()


def test_wnn_circuit_keygen_is_image_independent_for_the_reference_model():
    wnn = reference_wnn()
    circuit = wc.WnnCircuit(wnn, K_WNN)
    asg, _ = circuit.synthesize(IMAGE)
    asg0, scores0 = circuit.synthesize(np.zeros_like(IMAGE))
    assert asg0.fixed == asg.fixed and asg0.mapping == asg.mapping
    assert scores0 == wnn.predict(np.zeros_like(IMAGE))


# ---- /root/reference/src/gadgets/hash.rs:236-371
def hash_circuit(k: int, value: int):
    """MyCircuit of hash.rs's tests: five advice columns, a constants column, an instance column, the bytes table;
    HashFunctionConfig { p: 11, l: 3, n_bits: 8 }."""
    cs = ConstraintSystem(k)
    input_, quotient, remainder, msb, hash_ = (cs.advice_column() for _ in range(5))
    constants = cs.fixed_column()
    cs.enable_equality(FIXED, constants)
    instance = cs.instance_column()
    cs.enable_equality(INSTANCE, instance)
    for a in (input_, quotient, remainder, msb, hash_):
        cs.enable_equality(ADVICE, a)
    table_column = cs.fixed_column()
    range_check = wc.RangeCheckConfig.configure(cs, input_, table_column)
    hash_config = wc.HashChip.configure(cs, input_, quotient, remainder, msb, hash_, p=11, l=3, n_bits=8)
    layouter = Layouter(cs, constants)
    assigned = layouter.assign_region(lambda region: region.assign_advice(input_, 0, value))
    layouter.assign_table((table_column,), [(i,) for i in range(256)])  # load_bytes_column (range_check.rs:131-149)
    out = wc.HashChip(hash_config, wc.RangeCheckConfig(range_check)).hash(layouter, assigned)
    return layouter, instance, out


@pytest.mark.parametrize("value,expected", [(2, 0), (4, 1), (42, 3), (255, 0)])
def test_hash_chip_kats(value, expected):
    assert (value ** 3 % 11) % 8 == expected  # (the comment in the reference's test)
    layouter, instance, out = hash_circuit(9, value)
    assert out.value == expected
    layouter.constrain_instance(out, instance, 0, expected)
    layouter.asg.compress_selectors(True)
    layouter.asg.check()


def test_hash_chip_rejects_a_wrong_output():
    layouter, instance, out = hash_circuit(9, 42)
    with pytest.raises(AssertionError):  # (the mirror asserts a copy constraint when it is made)
        layouter.constrain_instance(out, instance, 0, 4)  # (42^3 % 11) % 8 = 3, not 4
    # ... and a hash cell altered behind the chip's back fails the gate
    layouter.asg.compress_selectors(True)
    layouter.asg.advice[out.column][out.row] = 4
    with pytest.raises(AssertionError):
        layouter.asg.check()


# ---- /root/reference/src/gadgets/range_check.rs:165-288
def le_circuit(k: int, x: int, y: int):
    cs = ConstraintSystem(k)
    advice_column = cs.advice_column()
    table_column = cs.fixed_column()
    constants = cs.fixed_column()
    cs.enable_equality(ADVICE, advice_column)
    cs.enable_equality(FIXED, constants)
    config = wc.RangeCheckConfig.configure(cs, advice_column, table_column)
    layouter = Layouter(cs, constants)
    x_cell = layouter.assign_region(lambda region: region.assign_advice(advice_column, 0, x))
    layouter.assign_table((table_column,), [(i,) for i in range(256)])
    wc.RangeCheckConfig(config).le_constant(layouter, x_cell, y)
    layouter.asg.compress_selectors(True)
    return layouter.asg


@pytest.mark.parametrize("x,y", [(1023, 1023), (1022, 1023), (4, 9), (0, 0xFFABCDEF)])
def test_range_check_le_constant_holds(x, y):
    le_circuit(9, x, y).check()


def test_range_check_le_constant_fails_for_a_greater_value():
    """test_le_greater_10bit: x = 1024, y = 1023 -> `prover.verify().is_err()`."""
    with pytest.raises(AssertionError):
        le_circuit(9, 1024, 1023).check()


# ---- /root/reference/src/gadgets/bloom_filter/byte_selector.rs:380-516
def byte_selector_circuit(k: int, value: int, index: int, num_bytes: int):
    cs = ConstraintSystem(k)
    cols = [cs.advice_column() for _ in range(6)]  # byte_decomposition, lookup_index, byte_index, byte_selector, selector_acc, byte_acc
    instance = cs.instance_column()
    constants = cs.fixed_column()
    table_column = cs.fixed_column()
    cs.enable_equality(INSTANCE, instance)
    for a in cols:
        cs.enable_equality(ADVICE, a)
    cs.enable_equality(FIXED, constants)
    config = wc.ByteSelectorChip.configure(cs, *cols, table_column)
    layouter = Layouter(cs, constants)
    input_cell, index_cell = layouter.assign_region(
        lambda region: (region.assign_advice(cols[0], 0, value), region.assign_advice(cols[1], 0, index)))
    layouter.assign_table((table_column,), [(i,) for i in range(256)])
    out = wc.ByteSelectorChip(config).select_byte(layouter, input_cell, index_cell, num_bytes)
    return layouter, instance, out


@pytest.mark.parametrize("value,index,num_bytes,expected", [(0xAB, 0, 1, 0xAB), (0xABCDEF, 0, 3, 0xAB), (0xABCDEF, 1, 3, 0xCD)])
def test_byte_selector_kats(value, index, num_bytes, expected):
    layouter, instance, out = byte_selector_circuit(9, value, index, num_bytes)
    assert out.value == expected
    layouter.constrain_instance(out, instance, 0, expected)
    layouter.asg.compress_selectors(True)
    layouter.asg.check()


def test_byte_selector_rejects_a_wrong_output():
    layouter, instance, out = byte_selector_circuit(9, 0xABCDEF, 1, 3)
    with pytest.raises(AssertionError):  # (the mirror asserts a copy constraint when it is made)
        layouter.constrain_instance(out, instance, 0, 0xAB)
    layouter.asg.compress_selectors(True)
    layouter.asg.advice[out.column][out.row] = 0xAB
    with pytest.raises(AssertionError):
        layouter.asg.check()


# ---- the same 4x3 model through the DEVICE witness program and the prover
def test_witness_program_of_the_reference_model_replays_on_the_host():
    """harness/witness_tape.py records WnnChip::predict once on a symbolic image; replayed on the concrete image it must
    reproduce the synthesis above cell for cell (CPU; the device replay is the -m gpu case below)."""
    import witness_tape

    wnn = reference_wnn()
    prog = witness_tape.trace(wnn, K_WNN)
    circuit = wc.WnnCircuit(wnn, K_WNN)
    asg, scores = circuit.synthesize(IMAGE)
    adv, got_scores = prog.run(IMAGE)
    assert got_scores == scores == [1, 2]
    assert [list(col) for col in adv] == asg.advice


@pytest.mark.gpu
def test_reference_model_on_the_device_witness_and_proof(ctx, zg, orc):
    """zg_witness_run_dev on the reference's 4x3 image: advice columns == the host synthesis bit for bit, class scores
    [1, 2], and the proof made from the device-resident columns == the oracle's create_proof of the host witness
    (pairing-verified against the public inputs [1, 2])."""
    import witness_tape

    wnn = reference_wnn()
    cs, asg, ilen, scores = wc.build(wnn, IMAGE, K_WNN)
    assert scores == [1, 2]
    img = cs.to_c()
    params = orc.params_new(K_WNN, 0x5EED)
    vk_repr = orc.fr_from_int(0xC0FFEE)
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    prover = zg.Prover(ctx, img, fixed, sigma, params.g_np(), params.g_lagrange_np(), vk_repr)
    prover.set_batch(2)
    prover.set_overlap(False)
    plan = zg.WitnessPlan(ctx, witness_tape.trace(wnn, K_WNN).arrays())
    other = np.zeros_like(IMAGE)
    slots = [prover.advice_slot(b) for b in range(2)]
    inst = plan.run(np.stack([IMAGE.reshape(-1), other.reshape(-1)]), slots)
    want_inst = asg.instance_values(ilen)
    assert np.array_equal(inst[0], want_inst[0])
    assert [orc.fr_to_int(inst[0][i]) for i in range(2)] == [1, 2]
    import ctypes

    want_adv = asg.advice_values()
    got_adv = np.zeros(want_adv.shape, np.uint64)
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(got_adv.ctypes.data), ctypes.c_void_p(slots[0]), ctypes.c_size_t(want_adv.nbytes), 2) == 0
    assert np.array_equal(got_adv, want_adv)
    proofs, sts = prover.prove_batch(None, [i[None, :, :] for i in inst], [5, 6], device=True)
    assert sts == [0, 0]
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    st, want, _ = orc.create_proof(pk, asg.advice_values(), want_inst, 5)
    assert st == 0 and proofs[0] == want
    assert orc.verify_proof_pairing(pk, want_inst, proofs[0]) == 1
    _, asg0, _, scores0 = wc.build(wnn, other, K_WNN)
    st, want0, _ = orc.create_proof(pk, asg0.advice_values(), asg0.instance_values(ilen), 6)
    assert st == 0 and proofs[1] == want0 and scores0 == wnn.predict(other)
    plan.close()
    prover.close()
