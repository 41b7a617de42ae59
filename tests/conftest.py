import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# A thread budget for the suite and everything it starts (set before torch is imported; inherited by mp.spawn workers and by the
# bench.py / launcher subprocesses): the GPU box has 128-256 hardware threads shared with other jobs on the pod, and the multi-process
# tests start up to 8 ranks — each defaulting to ALL cores for its host work — which made one two-rank test take 5 s or 79 s depending
# on the neighbours (profiles/r06/README.md).  The tests' own host work (oracle graphs at tiny sizes) does not need more.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, os.cpu_count() or 8))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
