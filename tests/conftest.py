import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU checker (test infrastructure): built on demand by oracle/Makefile."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Contexts on cuda:0 through the C ABI; fails loudly when the HIP library is absent."""
    from pbdagcon_amd import capi
    capi.load()
    made = []

    def make(**kw):
        c = capi.Context(**kw)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
