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


@pytest.hookimpl(hookwrapper=True)
def pytest_pyfunc_call(pyfuncitem):
    """tests/test_shardset_*.py run the imported scenario suites over shard sets of several shapes (SWP_SHARDSET): a scenario with more
    nodes than a shape has slots is skipped for that shape, not failed (the set refuses the node: SWP_ERANGE, 'the shard set is full')."""
    outcome = yield
    if os.environ.get("SWP_SHARDSET") and outcome.excinfo is not None and "the shard set is full" in str(outcome.excinfo[1]):
        outcome.force_exception(pytest.skip.Exception("the scenario needs more node slots than a shard set of shape %s has" % os.environ["SWP_SHARDSET"]))
