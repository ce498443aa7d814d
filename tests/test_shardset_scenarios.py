"""GPU: EVERY scenario of tests/test_engine_scenarios.py — the reference's scheduler_test.go scenarios and the filter truth tables — again,
with the engine behind the host layers replaced by a shard SET of 4 engines with 8 node slots each (SWP_SHARDSET, read by
swarmkit_amd.abi.Engine): the scenarios' handful of nodes straddle the range borders, so every path they take — one-off batches,
task groups (the union engine), preassigned tasks (taskFitNode on the owner), failures, host ports, generic resources, node removal and
index recycling — crosses shards. The tests are the imported ones, unchanged; only the fixture below differs."""
import pytest

from test_engine_scenarios import *   # noqa: F401,F403  (the tests, their factory and the host-layer fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def shard_set(monkeypatch):
    monkeypatch.setenv("SWP_SHARDSET", "4:8")
