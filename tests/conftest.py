"""pytest config: registers the `gpu` marker and makes the repo root importable."""
import os
import sys

import pytest

# Tests that load the reference's example files by path (build container only) must never drop __pycache__ into
# the read-only reference tree.
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
