import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The CPU oracle is OpenMP code.  The GPU boxes have 256 logical CPUs shared with other tenants: with one thread per CPU and
# libgomp's default active waiting, a busy neighbour turns every parallel region's barrier into a scheduler wait -- the same
# GPU suite took 4.4 min on a quiet box and > 15 min on a busy one (r03).  The checker does not need 256 threads: bound them,
# and let idle threads sleep.  (Set before libgomp is loaded; an explicit OMP_NUM_THREADS of the caller wins.)
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    return binding.load()
