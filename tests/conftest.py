import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "generative-models_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def train_golden():
    return {c["name"]: c for c in load_golden("train_step.json")}


@pytest.fixture(scope="session")
def metrics_golden():
    return load_golden("metrics.json")


@pytest.fixture(scope="session")
def extra_golden():
    """Round-2 fixtures from the imported reference: multi-step optimizer runs, eval-mode samples, the epoch loop,
    reference-written pickles (oracle/gen_golden.py gen_extra)."""
    return load_golden("train_extra.json")
