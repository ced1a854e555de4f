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
    # The CPU-oracle replays at the configuration sizes (tests/test_config_sizes_gpu.py, test_egemm_gpu.py) are torch CPU
    # ops; on the GPU box's 256 hardware threads torch's default (all of them) runs them ~4x SLOWER than 32 threads
    # (bench.py's thread sweeps: kNN 507 ms at 32 threads against 2010 ms at 256; arxiv aggregation 0.40 vs 0.11 M
    # edges/s).  DGCN_TEST_THREADS overrides.
    import torch
    want = int(os.environ.get("DGCN_TEST_THREADS", "32"))
    torch.set_num_threads(max(1, min(want, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
