import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def probe_lib():
    """The MEASUREMENT build of the library (libdreg_nerf_hip_probe.so, include/dreg_nerf_probe.h) for the duration of one test: inside,
    dreg_nerf_amd.lib.load() returns it, so the package's wrappers run its kernels and its process-global variant setters exist.  The product
    library has none of them (tests/test_abi_and_ddp.py)."""
    from dreg_nerf_amd import lib as L
    with L.probe() as pr:
        yield pr.lib


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
