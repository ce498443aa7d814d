import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# the Python twin of the host layer (tests/pyhost.py, test infrastructure) answers to SWP_HOST=py
import pyhost  # noqa: E402
from swarmkit_amd import host as _swhost  # noqa: E402
_swhost.register_twin(pyhost.PyHostScheduler)
