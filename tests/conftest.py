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


def _torch_first():
    """torch is imported BEFORE anything of the package: its wheel bundles its own libamdhip64.so.7, and a process must end up
    with exactly one HIP runtime.  With torch loaded first our library's DT_NEEDED libamdhip64.so.7 binds to that copy; the
    other order gives torch a second runtime that sees no GPU ("No HIP GPUs are available").  Any import of a package module
    (also orb_slam3-1_amd.synth / .synth_match) loads the library through the package's __init__."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


@pytest.fixture(scope="session")
def synth():
    _torch_first()
    return importlib.import_module("orb_slam3-1_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle_api import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes mirror of the C ABI).  Loading fails loudly if the HIP library is missing."""
    _torch_first()
    return importlib.import_module("orb_slam3-1_amd")


@pytest.fixture(scope="session")
def sm():
    _torch_first()
    return importlib.import_module("orb_slam3-1_amd.synth_match")

