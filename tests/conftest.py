import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ten_by_ten():
    import json
    import numpy as np
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "ten_by_ten.json")))
    return np.array(d["occupancy_rows_y0_first"], dtype=np.int8)
