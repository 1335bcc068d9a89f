import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# torch brings its own libamdhip64.so.7 and the library under test links the system's copy of the same name: whichever is
# loaded first serves both.  Tests that keep buffers in torch tensors need torch's to be the one, so it is loaded here,
# before any test can load the library (a torch initialised after the system's copy reports "No HIP GPUs are available").
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
