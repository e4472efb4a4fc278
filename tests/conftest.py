"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs here (no GPU): oracle vs the reference's golden vectors, host logic,
C-ABI symbol export checks, gloo world_size-2 tests. `-m gpu` runs on a real MI355X.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)
