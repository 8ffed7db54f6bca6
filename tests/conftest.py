import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("orb_slam3-1_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle_api import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes mirror of the C ABI).  Loading fails loudly if the HIP library is missing."""
    return importlib.import_module("orb_slam3-1_amd")
