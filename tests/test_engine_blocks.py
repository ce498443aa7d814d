"""GPU parity of the block resolver (csrc/swp_resolve6.hpp: candidate lists built by the whole chip from bitmap rows in global
memory, matched by one wave, two launches per round) — the path node sets beyond k_resolve5's LDS take by default, forced here
at small sizes too. Placements and explanations must equal the oracle's bit for bit, at every block size."""
import os

import pytest

import parity_util as pu
from swarmkit_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def env():
    keys = ("SWP_RESOLVER", "SWP_R6_BLOCK", "SWP_WATERFILL")
    old = {k: os.environ.get(k) for k in keys}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def run_both(wl, **engine_kw):
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, s, *_ = pu.engine_run(wl, **engine_kw)
    pu.assert_same(op, oe, ep, ee)
    return s


@pytest.mark.parametrize("block", [1, 7, 64, 256, 1024])
@pytest.mark.parametrize("name,T,N", [("cfg3", 3000, 900), ("cfg4", 4000, 700)])
def test_block_sizes(env, block, name, T, N):
    os.environ["SWP_RESOLVER"] = "6"
    os.environ["SWP_R6_BLOCK"] = str(block)
    s = run_both(synth.Workload(name, T=T, N=N))
    assert s.e.stats()["last_resolver"] == 6


@pytest.mark.parametrize("N", [1, 64, 65, 4100, 12400])
def test_forced_at_small_node_counts(env, N):
    os.environ["SWP_RESOLVER"] = "6"
    s = run_both(synth.Workload("cfg3", T=1500, N=N))
    assert s.e.stats()["last_resolver"] == 6


@pytest.mark.parametrize("name,T,N", [("cfg3", 1200, 16500), ("cfg3", 700, 40000), ("cfg4", 1500, 20500), ("cfg4", 1000, 70000)])
def test_default_beyond_the_round_resolver(name, T, N):
    """No knob set: a node set k_resolve5 cannot hold in LDS goes to the block resolver."""
    s = run_both(synth.Workload(name, T=T, N=N))
    assert s.e.stats()["last_resolver"] == 6


@pytest.mark.parametrize("services,order", [(1, "rr"), (3, "major"), (40, "major")])
def test_services_with_more_tasks_than_nodes(env, services, order):
    """Every node soon runs the service: the exception lists decide, one task per round."""
    os.environ["SWP_RESOLVER"] = "6"
    os.environ["SWP_WATERFILL"] = "0"
    run_both(synth.Workload("cfg3", T=1500, N=300, services=services, order=order))


def test_uncounted_tasks_cut_the_block(env):
    """A task whose DesiredState is beyond COMPLETED does not count on its node (nodeinfo.go:131-134): the node stays on its level
    and the next task may take it again — the block must be cut behind such a pick."""
    os.environ["SWP_RESOLVER"] = "6"
    wl = synth.Workload("cfg2", T=2000, N=300)
    wl.uncounted_every = 5
    run_both(wl)


def test_runs_and_blocks_share_a_batch(env):
    """Runs of identical tasks go through k_waterfill, the stretches between them through the block resolver, which rebuilds its
    bitmaps from the node rows the runs left."""
    os.environ["SWP_RESOLVER"] = "6"
    os.environ["SWP_WATERFILL"] = "1"
    run_both(synth.Workload("cfg3", T=3000, N=500, services=12, order="major"))


def test_level_spread_needs_more_than_eight_planes(env):
    """One node is DOWN while two others take a thousand tasks: levels beyond 255 above the base (the planes are 16 bits wide)."""
    import orc
    from swarmkit_amd import host as swhost
    os.environ["SWP_RESOLVER"] = "6"
    os.environ["SWP_WATERFILL"] = "0"
    o, e = orc.Oracle(), swhost.HostScheduler()
    docs = [{"ID": "n0", "Status": {"State": orc.READY}}, {"ID": "n1", "Status": {"State": orc.DOWN}}, {"ID": "n2", "Status": {"State": orc.READY}}]
    for s in (o, e):
        for d in docs:
            s.create_node(d)
        for k in range(3):
            s.set_service("svc%d" % k)
    for rnd, cnt in enumerate((350, 400, 250)):
        for j in range(cnt):
            t = {"ID": "t%d_%04d" % (rnd, j), "ServiceID": "svc%d" % (j % 3), "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
            for s in (o, e):
                s.create_task(t)
        do = sorted((d["ID"], d["NodeID"], d["Err"]) for d in o.tick())
        de = sorted((d["ID"], d["NodeID"], d["Err"]) for d in e.tick())
        assert do == de


@pytest.mark.parametrize("T,N,S", [(6000, 700, 1500), (3000, 13000, 900), (5000, 300, 4000)])
def test_many_distinct_reservations_go_to_the_block_resolver(T, N, S):
    """Every service its own NanoCPUs / MemoryBytes pair (synth cfg3m): far more demand classes than k_resolve5 has LDS rows for. No
    knob set: the batch must run through the block resolver (rows in global memory, class indices of 12 bits, the apply step bisects
    the thresholds a commit crosses) — not through a round-1 fall-back — and agree with the oracle."""
    s = run_both(synth.Workload("cfg3m", T=T, N=N, services=S))
    assert s.e.stats()["last_resolver"] == 6
