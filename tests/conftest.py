import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def hiplib(pkg):
    """The loaded C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    if not os.path.exists(pkg.lib.LIB_PATH):
        g.build()
    return pkg.lib.load()
