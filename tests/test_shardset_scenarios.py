"""GPU: EVERY scenario of tests/test_engine_scenarios.py — the reference's scheduler_test.go scenarios and the filter truth tables — again,
with the engine behind the host layers replaced by a shard SET (SWP_SHARDSET, read by swarmkit_amd.abi.Engine) of two shapes — 8 engines
with 2 node slots each, 4 with 16 — so that the scenarios' handful of nodes straddle range borders (a scenario with more nodes than a
shape has slots is skipped for it: tests/conftest.py) and every path they take — one-off batches,
task groups (the union engine), preassigned tasks (taskFitNode on the owner), failures, host ports, generic resources, node removal and
index recycling — crosses shards. The tests are the imported ones, unchanged; only the fixture below differs."""
import pytest

from test_engine_scenarios import *   # noqa: F401,F403  (the tests, their factory and the host-layer fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["8:2", "4:16"])
def shard_set(request, monkeypatch):
    monkeypatch.setenv("SWP_SHARDSET", request.param)
