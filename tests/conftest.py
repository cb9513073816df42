"""pytest wiring: the `gpu` marker, import paths for the product binding and the oracle binding."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "harness"))  # the caller side restated (circuit, layouter, Wnn model)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc

    _orc.load()
    return _orc


@pytest.fixture(scope="session")
def zg():
    import zg_halo2

    zg_halo2.load()
    return zg_halo2


@pytest.fixture(scope="session")
def ctx(zg):
    c = zg.Ctx(0)
    yield c
    c.close()
