import os
import sys

import pytest

# The oracle's OpenMP loops are small in these tests (a read, a tile): on the GPU box's host (128+ hardware threads)
# the default team made every parallel region cost more than its work — the oracle-bound GPU tests took 44 s each
# there against 2 s on the 8-core build container (round 4, profiles/r04_suite_durations.txt).  Set before any
# OpenMP runtime is loaded; the bench's CPU baseline sets its own thread count.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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
