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
def qc():
    """The host layer (quantumcollocation.jl_amd/), built on demand."""
    import __graft_entry__ as g

    lib = os.path.join(g.CSRC, "libqcolloc_hip.so")
    if not os.path.exists(lib):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as g

    return g.load_oracle()


@pytest.fixture(scope="session")
def coracle():
    """oracle/qc_oracle_c.py: the C restatement's loader (test infrastructure)."""
    import oracle.qc_oracle_c as oc
    return oc
