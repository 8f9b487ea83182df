import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_kats():
    import json
    with open(os.path.join(HERE, "golden", "ref_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def model_vectors():
    import json
    with open(os.path.join(HERE, "golden", "model_vectors.json")) as f:
        return json.load(f)
