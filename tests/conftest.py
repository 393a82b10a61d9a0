import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
