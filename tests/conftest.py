import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import mi_oracle
    mi_oracle.build()
    return mi_oracle


@pytest.fixture(scope="session")
def engine_lib():
    import makisu_amd
    return makisu_amd.load_library()
