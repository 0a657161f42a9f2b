import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Every device buffer the library allocates is filled with a pattern in the test processes (efx_api.cpp, DevBuf::reserve): a
# kernel that reads memory nobody wrote then fails parity deterministically, instead of only when a recycled allocation
# happens to hold a previous frame's data.
os.environ.setdefault("EFX_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
