import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")


def load_golden_mesh(name):
    m = json.load(open(os.path.join(GOLDEN, name + ".json")))
    v = np.array(m["vertices"], dtype=np.float64)
    c = np.array(m["connectivity"], dtype=np.uint64)
    return v, c


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (test infrastructure)."""
    from oracle import oracle as o

    o.lib()
    return o
