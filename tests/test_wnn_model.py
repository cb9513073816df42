"""Host-side WNN model mirror (harness/wnn_model.py) against the values the reference's own tests
pin: tests/integration_test.rs:19,36,53 `snapshot_mnist_*_predictions` (Wnn::predict on
benches/example_image_7.png), via the fixtures extracted from the checked-in model files."""
import json
import os

import numpy as np
import pytest

import wnn_model as wm

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def snapshots():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)["reference"]["predictions"]


def test_test_image_is_the_reference_png():
    img = wm.load_test_image()
    assert img.shape == (28, 28) and img.dtype == np.uint8 and int(img.sum()) == 18454


@pytest.mark.parametrize("k,name", [wm.MNIST_TINY, wm.MNIST_SMALL, wm.MNIST_MEDIUM])
def test_snapshot_predictions(snapshots, k, name):
    wnn = wm.load_checked_in(name)
    pred = wnn.predict(wm.load_test_image())
    assert pred == snapshots[name]
    assert int(np.argmax(pred)) == 7  # the image is a seven


def test_circuit_params_follow_the_loader():
    p = wm.load_checked_in(wm.MNIST_SMALL[1]).get_circuit_params()
    assert (p.p, p.l, p.n_hashes, p.bits_per_hash, p.bits_per_filter, p.n_classes) == (2097143, 20, 2, 10, 28, 10)


def test_thresholds_quantisation_range():
    for _, name in (wm.MNIST_TINY, wm.MNIST_SMALL, wm.MNIST_MEDIUM):
        w = wm.load_checked_in(name)
        t = w.binarization_thresholds
        assert t.dtype == np.uint16 and t.min() >= 0 and t.max() <= 256
        assert sorted(w.input_permutation.tolist()) == list(range(t.size))
