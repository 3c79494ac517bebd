import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


# The cuts are made for frames of >= 12 M rays by default; the suites render far smaller frames, so
# here every frame gets them -- every parity test then also tests the cuts.  (tests that want the default unset it)
os.environ.setdefault("RT_CUT_MIN_RAYS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(TESTS, "golden")
