import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib

    return importlib.import_module(graft.PKG_NAME + ".synth")


@pytest.fixture(scope="session")
def sor():
    """the oracle (oracle/liboracle.so), built on demand"""
    mod = graft.load_oracle()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()
