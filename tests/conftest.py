"""pytest wiring: the `gpu` marker, import paths for the product binding and the oracle binding."""
import os
import sys

import pytest
# PyTorch-ROCm ships its OWN HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7); libzg_halo2.so needs
# libamdhip64.so.7 as well.  Whichever is loaded first decides: with torch first the loader hands libzg torch's copy (one
# runtime in the process); with libzg first /opt/rocm's copy is loaded, torch brings its own beside it, and the SECOND
# runtime to initialise finds no device ("No HIP GPUs are available" / ZG_ERR_NO_DEVICE -- the round-2 incident, DESIGN.md
# toolchain notes).  So: torch before anything can load the library, in every process that uses both.
import torch  # noqa: F401

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
