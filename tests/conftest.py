import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    return _load("reference_kats.json")


@pytest.fixture(scope="session")
def murmur_vectors():
    return _load("murmur3_x64_128_vectors.json")["vectors"]


@pytest.fixture(scope="session")
def example_digests():
    return _load("example_fa_digests.json")


@pytest.fixture(scope="session")
def example_seq():
    with open(os.path.join(GOLDEN, "example.fa")) as f:
        return "".join(line.strip() for line in f if not line.startswith(">"))
