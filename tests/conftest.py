import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "gorilla-tsc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    g = os.path.join(ROOT, "tests", "golden")
    return {
        "inputs": json.load(open(os.path.join(g, "reference_test_inputs.json"))),
        "answers": json.load(open(os.path.join(g, "known_answers.json"))),
    }
