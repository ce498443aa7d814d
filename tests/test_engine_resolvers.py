"""GPU parity across the two resolver families and their geometries: 5, the round resolver (everything in one workgroup's LDS,
the default where it fits), and 6, the block resolver (bitmap rows in global memory; beyond k_resolve5's node limit, for batches with
more distinct reservations than k_resolve5 has rows for, for generic reservations) — at every owned-words-per-lane count K — must
reproduce the oracle bit for bit. (The round-1 generations k_resolve / 1 / 2 / 3 and the scan they fed on were retired in round 3.)"""
import os

import pytest

import parity_util as pu
from swarmkit_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def resolver_env():
    old = {k: os.environ.get(k) for k in ("SWP_RESOLVER", "SWP_R6_BLOCK", "SWP_R6_TASKROWS", "SWP_R6_COMPACT")}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def pick_variant(variant):
    os.environ.pop("SWP_R6_TASKROWS", None)
    os.environ.pop("SWP_R6_COMPACT", None)
    if variant == "6c":   # the block resolver with a compact index in front of every round (k_r6_compact; by itself only after the symptom)
        os.environ["SWP_RESOLVER"] = "6"
        os.environ["SWP_R6_COMPACT"] = "1"
    elif variant == "6t":   # the block resolver with ResourceFilter rows per task of the block instead of per demand class
        os.environ["SWP_RESOLVER"] = "6"
        os.environ["SWP_R6_TASKROWS"] = "1"
    else:
        os.environ["SWP_RESOLVER"] = str(variant)


CASES = [("cfg3", 2500, 300, {}), ("cfg4", 3000, 700, {}), ("cfg1", 500, 40, {}), ("cfg2", 3000, 50, {})]
@pytest.mark.parametrize("variant", [5, 6, "6t", "6c"])
@pytest.mark.parametrize("name,T,N,kw", CASES)
def test_variants_agree_with_oracle(resolver_env, variant, name, T, N, kw):
    wl = synth.Workload(name, T=T, N=N, **kw)
    op, oe, _ = pu.oracle_run(wl)
    pick_variant(variant)
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


# node counts that land on K = 1..6 words per lane (64 nodes per word, 64 lanes), incl. word-boundary sizes
@pytest.mark.parametrize("N", [1, 63, 64, 65, 4096, 4100, 8200, 12400, 16500, 20500, 26000, 30000, 40000])
def test_words_per_lane(N):
    T = 2500 if N <= 5000 else (1200 if N <= 21000 else 700)
    wl = synth.Workload("cfg3", T=T, N=N)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


@pytest.mark.parametrize("variant", [5, 6, "6t", "6c"])
@pytest.mark.parametrize("services,order", [(1, "rr"), (2, "rr"), (3, "major"), (40, "major"), (7, "rr")])
def test_same_service_runs(resolver_env, variant, services, order):
    """Consecutive tasks of one service: every commit must be visible to the next task of that service although
    its staged exception row is older (the resolver's commit ring)."""
    wl = synth.Workload("cfg3", T=2000, N=700, services=services, order=order)
    op, oe, _ = pu.oracle_run(wl)
    pick_variant(variant)
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


def test_level_spread_beyond_the_round_resolvers_planes():
    """One node keeps its count (it is DOWN) while the others take hundreds of tasks: the per-node spread outgrows the 255 levels
    the round resolver keeps in LDS in the middle of a batch. The engine must carry on with the block resolver (16 planes in global
    memory) from the task where the round resolver stopped — same placements as the oracle, no error."""
    import orc
    from swarmkit_amd import host as swhost
    o, e = orc.Oracle(), swhost.HostScheduler()
    docs = [{"ID": "n0", "Status": {"State": orc.READY}}, {"ID": "n1", "Status": {"State": orc.DOWN}}, {"ID": "n2", "Status": {"State": orc.READY}}]
    for s in (o, e):
        for d in docs:
            s.create_node(d)
        s.set_service("svc")
    for rnd, cnt in enumerate((350, 400, 250)):
        for j in range(cnt):
            t = {"ID": "t%d_%04d" % (rnd, j), "ServiceID": "svc", "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
            for s in (o, e):
                s.create_task(t)
        do = sorted((d["ID"], d["NodeID"], d["Err"]) for d in o.tick())
        de = sorted((d["ID"], d["NodeID"], d["Err"]) for d in e.tick())
        assert do == de
    assert e.node_info("n0")["ActiveTasksCount"] == o.node_info("n0")["ActiveTasksCount"] == 500
