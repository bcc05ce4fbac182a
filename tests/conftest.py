import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/_build on first use."""
    import orc

    orc.build()
    orc.load()
    return orc


@pytest.fixture(scope="session")
def native():
    """The HIP engine binding; builds the in-tree .so if it is missing."""
    from goldrush_amd import native as nat

    if not os.path.exists(nat.LIB_PATH):
        nat.build()
    nat.load()
    return nat
