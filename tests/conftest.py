import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _product_built():
    """The built libraries are git-ignored: in a fresh checkout compile them once (hipcc cross-compiles gfx950
    without a GPU).  Importing the package without its HIP library fails on purpose (there is no CPU fallback)."""
    lib = os.path.join(ROOT, "datacompressionfloat_amd", "lib", "libmrcz_hip.so")
    workers = os.path.join(ROOT, "datacompressionfloat_amd", "lib", "libmrcz_workers.so")
    if not (os.path.exists(lib) and os.path.exists(workers)):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so); built on demand with gcc."""
    import util
    return util.load_oracle()


@pytest.fixture(scope="session")
def simlib():
    """The product's HIP sources compiled with g++ against the test-only SIMT emulator."""
    import util
    return util.load_sim()
