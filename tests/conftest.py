import os
import sys

import pytest

# The ranks-as-threads tests (test_gpu_dist.py) keep one persistent round kernel per rank resident at the same time; HIP maps a
# process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on one queue run in order, so a fifth
# rank's kernel would wait behind a resident one that is itself waiting for that rank.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
