"""GPU parity of k_waterfill (csrc/swp_waterfill.hpp): runs of identical one-off tasks placed by water-filling over
(failure class, svcCount, ActiveTasksCount, node index) instead of task by task must give the oracle's placements bit for bit —
node order inside a level, nodes that fill up in the middle of a run, MaxReplicas, runs that end in "no suitable node", batches
that mix runs with single tasks (host ports keep a task out of a run)."""
import os

import numpy as np
import pytest

import parity_util as pu
from swarmkit_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def waterfill_env():
    old = os.environ.get("SWP_WATERFILL")
    yield
    if old is None:
        os.environ.pop("SWP_WATERFILL", None)
    else:
        os.environ["SWP_WATERFILL"] = old


def run_both(wl, mode):
    op, oe, _ = pu.oracle_run(wl)
    os.environ["SWP_WATERFILL"] = mode
    ep, ee, s, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)
    return s.e.stats()["waterfill_tasks"]


@pytest.mark.parametrize("name,T,N,services", [("cfg3", 3000, 300, 40), ("cfg3", 3000, 300, 3), ("cfg3", 2500, 700, 1), ("cfg4", 3000, 700, 25),
                                               ("cfg2", 3000, 50, 6), ("cfg1", 500, 40, 1), ("cfg4", 4000, 200, 1)])
def test_service_major_runs(waterfill_env, name, T, N, services):
    wl = synth.Workload(name, T=T, N=N, services=services, order="major")
    assert run_both(wl, "1") > T // 2   # the runs really went through k_waterfill


def test_round_robin_order_has_no_runs_and_still_agrees(waterfill_env):
    wl = synth.Workload("cfg3", T=2000, N=300)
    assert run_both(wl, "1") == 0


@pytest.mark.parametrize("N", [1, 63, 65, 1025, 5000, 20000])
def test_node_counts(waterfill_env, N):
    """one node per thread, several per thread, fewer nodes than threads; beyond the round resolver's range too"""
    wl = synth.Workload("cfg3", T=1500, N=N, services=4, order="major")
    run_both(wl, "1")


def test_default_policy_takes_the_reference_benchmark_shape(waterfill_env):
    """Without the knob: a batch that is one long run (the reference's benchScheduler: every task of ONE service) is water-filled."""
    wl = synth.Workload("cfg2", T=5000, N=120, services=1, order="major")
    os.environ.pop("SWP_WATERFILL", None)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, s, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)
    assert s.e.stats()["waterfill_tasks"] == wl.T


def test_off_switch(waterfill_env):
    wl = synth.Workload("cfg3", T=1500, N=200, services=2, order="major")
    assert run_both(wl, "0") == 0
