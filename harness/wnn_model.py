"""Host-side mirror of zero_g's `Wnn` (/root/reference/src/wnn.rs) and of its loader
(/root/reference/src/io.rs:36-92): the BTHOWeN-style weightless neural network whose inference the
circuit proves.  Same names, same argument meaning; arrays are numpy.

Models come from the fixtures tools/extract_models.py writes (tests/golden/models/*.npz: the HDF5
attributes and datasets, untouched) -- the quantisation below is the loader's, restated.
"""
from __future__ import annotations

import math
import os

import numpy as np

from formats import WnnCircuitParams  # the product package's on-disk formats own this record (io.rs:149-156)

ATTRS = ["num_classes", "num_inputs", "bits_per_input", "num_filter_inputs", "num_filter_entries",
         "num_filter_hashes", "p"]

# (k, model name) -- /root/reference/src/lib.rs:48-51 `checked_in_test_data`
MNIST_TINY = (14, "model_28input_256entry_1hash_1bpi")
MNIST_SMALL = (15, "model_28input_1024entry_2hash_2bpi")
MNIST_MEDIUM = (15, "model_28input_2048entry_2hash_3bpi")
MNIST_LARGE = (17, "model_49input_8192entry_4hash_6bpi")  # file absent from the reference checkout


class Wnn:
    def __init__(self, num_classes: int, num_filter_entries: int, num_filter_hashes: int, num_filter_inputs: int,
                 p: int, bloom_filters: np.ndarray, input_order: np.ndarray, binarization_thresholds: np.ndarray):
        self.num_classes = num_classes
        self.num_filter_entries = num_filter_entries
        self.num_filter_hashes = num_filter_hashes
        self.num_filter_inputs = num_filter_inputs
        self.p = p
        self.bloom_filters = bloom_filters                    # bool (classes, filters, entries)
        self.input_permutation = input_order                  # (num_inputs * bits_per_input)
        self.binarization_thresholds = binarization_thresholds  # u16 (w, h, bits_per_input), in [0, 256]

    # wnn.rs:83-97
    def thermometer_encoding(self, image: np.ndarray) -> np.ndarray:
        thr = self.binarization_thresholds
        # bit order (b, i, j)
        return (image.astype(np.uint16)[None, :, :] >= np.moveaxis(thr, 2, 0)).reshape(-1)

    # wnn.rs:99-104:  x^3 % p % entries^hashes
    def mish_mash_hash(self, x: int) -> int:
        return (x * x * x % self.p) % (self.num_filter_entries ** self.num_filter_hashes)

    # wnn.rs:106-129
    def encode_image(self, image: np.ndarray) -> list:
        bits = self.thermometer_encoding(image)
        assert bits.shape[0] == self.input_permutation.shape[0]
        permuted = bits[self.input_permutation.astype(np.int64)]
        n = self.num_filter_inputs
        out = []
        for c in range(0, permuted.shape[0] - n + 1, n):  # chunks_exact, little-endian packing
            v = 0
            for b in permuted[c:c + n][::-1]:
                v = (v << 1) + int(b)
            out.append(v)
        return out

    # wnn.rs:131-150
    def hash_indices(self, filter_index: int) -> list:
        h = self.mish_mash_hash(filter_index)
        return [(h // self.num_filter_entries ** i) % self.num_filter_entries for i in range(self.num_filter_hashes)]

    def bloom_filter_lookup(self, bloom_array: np.ndarray, filter_index: int) -> bool:
        return all(bool(bloom_array[i]) for i in self.hash_indices(filter_index))

    # wnn.rs:152-169
    def predict(self, image: np.ndarray) -> list:
        idx = self.encode_image(image)
        assert len(idx) == self.bloom_filters.shape[1]
        return [sum(int(self.bloom_filter_lookup(self.bloom_filters[c, f], v)) for f, v in enumerate(idx))
                for c in range(self.num_classes)]

    # wnn.rs:171-181
    def get_circuit_params(self) -> WnnCircuitParams:
        bph = int(math.log2(self.num_filter_entries))
        return WnnCircuitParams(p=self.p, l=self.num_filter_hashes * bph, n_hashes=self.num_filter_hashes,
                                bits_per_hash=bph, bits_per_filter=self.num_filter_inputs,
                                n_classes=self.bloom_filters.shape[0])

    def img_shape(self):
        return self.binarization_thresholds.shape[0], self.binarization_thresholds.shape[1]


def load_wnn(path: str) -> Wnn:
    """io.rs:36-92 on an extracted fixture: shape checks and threshold quantisation as there."""
    z = np.load(path)
    a = dict(zip(ATTRS, (int(v) for v in z["attrs"])))
    shape = tuple(int(v) for v in z["bloom_shape"])
    expected = (a["num_classes"], a["num_inputs"] * a["bits_per_input"] // a["num_filter_inputs"], a["num_filter_entries"])
    assert shape == expected
    bloom = np.unpackbits(z["bloom_bits"])[: int(np.prod(shape))].reshape(shape).astype(bool)
    width = int(math.sqrt(a["num_inputs"]))
    thr = z["thresholds_f32"].astype(np.float32)
    assert thr.shape == (width, width, a["bits_per_input"])
    # ceil(f32 * 255) clamped to [0, 256]: u8 >= f32  <=>  u8 >= ceil(f32); 256 is never reached
    q = np.minimum(np.maximum(np.ceil(thr * np.float32(255.0)), 0.0), 256.0).astype(np.uint16)
    order = z["input_order"].astype(np.uint64)
    assert order.shape == (a["num_inputs"] * a["bits_per_input"],)
    return Wnn(a["num_classes"], a["num_filter_entries"], a["num_filter_hashes"], a["num_filter_inputs"], a["p"],
               bloom, order, q)


def synthetic_wnn(num_classes=10, num_inputs=784, bits_per_input=6, num_filter_inputs=49, num_filter_entries=8192,
                  num_filter_hashes=4, p=(1 << 53) - 111, seed=1) -> Wnn:
    """Stand-in for model_49input_8192entry_4hash_6bpi, whose file is not in the reference checkout
    (.MISSING_LARGE_BLOBS): same shape, seeded contents."""
    rng = np.random.default_rng(seed)
    filters = num_inputs * bits_per_input // num_filter_inputs
    bloom = rng.random((num_classes, filters, num_filter_entries)) < 0.3
    width = int(math.sqrt(num_inputs))
    thr = np.sort(rng.integers(0, 257, (width, width, bits_per_input)), axis=2).astype(np.uint16)
    order = rng.permutation(num_inputs * bits_per_input).astype(np.uint64)
    return Wnn(num_classes, num_filter_entries, num_filter_hashes, num_filter_inputs, p, bloom, order, thr)


def fixture_dir() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")


def load_checked_in(name: str) -> Wnn:
    return load_wnn(os.path.join(fixture_dir(), name + ".npz"))


def load_test_image() -> np.ndarray:
    """benches/example_image_7.png, first channel (io.rs:24-33)."""
    return np.load(os.path.join(fixture_dir(), "example_image_7.npy"))
