import os
import sys

# The oracle can dlopen the MKL runtime; MKL's default Intel OpenMP layer must never share a process with
# torch's GNU OpenMP (dpotrf then returns silently wrong results), so pin the layer before anything loads MKL.
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def built():
    """make sure the product library and the oracle are built (hipcc cross-compiles without a GPU)"""
    import __graft_entry__ as g
    g.build_product()
    g.build_oracle()
    from dynadjust_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def orc(built):
    from tests import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from dynadjust_amd.device import DeviceContext
    ctx = DeviceContext(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
