import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the first HIP call: the handles drive two HIP streams per stream group
from alego_loader import load_package  # noqa: E402

load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle / synth / HIP libraries once if they are missing (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build(only_missing=True)


@pytest.fixture(scope="session")
def params_a():
    from alego_amd import synth
    return synth.default_params(16, 1800)
