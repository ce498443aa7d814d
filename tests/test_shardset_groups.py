"""GPU: tests/test_engine_groups.py and tests/test_engine_generic.py again over a shard SET of 4 engines with 160 node slots each
(SWP_SHARDSET): task groups — nodeSet.tree with a heap of k per leaf, orderedNodes, the fill loop, spread preferences, leftovers and their
explanations (nodeset.go:50-124, scheduler.go:772-924) — placed on a node set that lives on four engines. A call for task groups runs
on the set's union engine (csrc/swp_shardset.hpp) and every placement goes back to the owner of its node; the one-off batches in
between are sharded batches: the two must keep each other's state exact."""
import pytest

from test_engine_groups import *    # noqa: F401,F403
from test_engine_generic import *   # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def shard_set(monkeypatch):
    monkeypatch.setenv("SWP_SHARDSET", "4:160")
