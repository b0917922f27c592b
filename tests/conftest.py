import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The tests are the consumers of the validation twins: with DKT_TWINS=1 a test that sets one of the variant switches (ops._VARIANT_SWITCHES) is served by
# libdkt_twins.so (the same sources with -DDKT_TWINS); every other call -- and every call of a user process -- goes to the product library, which has no switches.
os.environ.setdefault("DKT_TWINS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The in-tree HIP shared object (built on demand; hipcc cross-compiles without a GPU)."""
    import dkt_amd
    dkt_amd._lib.build()
    return dkt_amd._lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda", 0)
