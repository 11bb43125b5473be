import os
import sys

import pytest

# the CPU oracle runs tiny grids in the tests: a few OpenMP threads beat one per core of a 256-core host
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def bz():
    import breeze_jl_amd
    return breeze_jl_amd


@pytest.fixture(scope="session")
def oc(oracle):
    from oracle import oracle_compressible
    return oracle_compressible
