"""GPU: the 64 seeded random event scripts of tests/test_engine_fuzz.py — mixed grouped / one-off services, random filters and spread
preferences, several ticks with node drains / removals / re-adds and task deletions in between — again, with the engine behind the host
layer replaced by a shard SET of 5 engines with 160 node slots each (SWP_SHARDSET, read by swarmkit_amd.abi.Engine): the one-off batches
are sharded rounds, the group calls run on the union engine, node events and removals go to the owner of the node's range, recycled node
indices land in whatever range has the lowest free slot. Every tick's decisions must equal the oracle's. The tests are the imported
ones, unchanged; only the fixture differs."""
import pytest

from test_engine_fuzz import *   # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def shard_set(monkeypatch):
    monkeypatch.setenv("SWP_SHARDSET", "5:160")
