"""GPU: the reference's scheduler scenarios and filter truth tables through the HIP engine (C ABI +
host mirror). Same scenario code as tests/test_oracle_scheduler.py."""
import pytest

import kat_tables as kt
import orc
import scenarios as sc
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["cxx", "py"])
def host_kind(request, monkeypatch):
    """Every scenario runs through both host layers: swp::Scheduler inside libswp.so and its Python twin."""
    monkeypatch.setenv("SWP_HOST", request.param)


def factory():
    return swhost.HostScheduler()


def test_basic():
    sc.scenario_basic(factory)


def test_ha_one_off():
    sc.scenario_ha(factory, False)


def test_no_ready_nodes():
    sc.scenario_no_ready_nodes(factory)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint(with_generic):
    """with_generic: the reference's own version (scheduler_test.go:1617-1775) — orange x 2 + apple x 2 against Named / Discrete node
    resources; HasEnough as demand-class rows on the device, Claim in the apply step, the available LIST in the host layer."""
    sc.scenario_resource_constraint(factory, with_generic=with_generic)


def test_platform():
    sc.scenario_platform(factory)


def test_host_port():
    sc.scenario_host_port(factory)


def test_max_replicas():
    sc.scenario_max_replicas(factory)


def test_faulty_node():
    sc.scenario_faulty_node(factory)


def test_constraint_truth_tables_on_device():
    """constraint_test.go:62-350 through k_constraint_classes (swp_check_node)."""
    for cons, node, want in kt.constraint_cases():
        s = factory()
        s.create_node(node)
        t = sc.pending("t", "svc", Spec={"Placement": {"Constraints": cons}})
        d = s.task_desc(t)
        if want is None:
            assert int(d["constraint_set"][0]) == 0, cons
            continue
        assert int(d["constraint_set"][0]) != 0, cons
        ff = s.e.check_node(d, s.e.intern(0, node["ID"]))   # SWP_SPACE_NODE_ID
        assert (ff == -1) == want, (cons, node, ff)
        assert ff in (-1, 3), (cons, ff)


# ---- the remaining scheduler_test.go scenarios (same code as tests/test_oracle_scheduler.py) -------------
def test_faulty_node_spec_version():
    sc.scenario_faulty_node_spec_version(factory)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint_ha(with_generic):
    sc.scenario_resource_constraint_ha(factory, with_generic=with_generic)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint_dead_task(with_generic):
    sc.scenario_resource_constraint_dead_task(factory, with_generic=with_generic)


@pytest.mark.parametrize("with_generic", [False, True])
def test_preexisting_dead_task(with_generic):
    sc.scenario_preexisting_dead_task(factory, with_generic=with_generic)


def test_unassigned_map():
    sc.scenario_unassigned_map(factory)


def test_preassigned_tasks():
    sc.scenario_preassigned_tasks(factory)


def test_ignore_tasks():
    sc.scenario_ignore_tasks(factory)


def test_unscheduleable_task():
    sc.scenario_unscheduleable_task(factory)


def test_plugin_constraint():
    sc.scenario_plugin_constraint(factory)


def test_node_indices_are_recycled_and_the_tie_order_follows(host_kind):
    """nodeSet.remove (nodeset.go:46-48) frees the node's index: ten times the cluster's size in node arrivals and departures leaves the
    engine's node space no wider than the cluster (a removed node's slot was never recycled), and every tick still places
    exactly like the oracle — whose canonical scan order recycles its slots by the same lowest-free rule."""
    import random
    rng = random.Random(0x51075)
    o, e = orc.Oracle(), factory()
    alive, serial = [], 0

    def doc(i):
        return sc.node("node-%05d" % i, Spec={"Annotations": {"Labels": {"zone": "z%d" % (i % 3)}}},
                       Description={"Resources": {"NanoCPUs": int(4e9), "MemoryBytes": int(8e9)}})
    for _ in range(64):
        for s in (o, e):
            s.create_node(doc(serial))
        alive.append(serial)
        serial += 1
    for s in (o, e):
        s.set_service("svc")
    tid = 0
    for step in range(40):
        for _ in range(16):   # 16 nodes leave, 16 new ones arrive: 640 arrivals over the run
            gone = alive.pop(rng.randrange(len(alive)))
            for s in (o, e):
                s.delete_node("node-%05d" % gone)
            for s in (o, e):
                s.create_node(doc(serial))
            alive.append(serial)
            serial += 1
        for _ in range(24):
            t = sc.pending("t%05d" % tid, "svc", Spec={"Resources": {"Reservations": {"NanoCPUs": int(1e8)}}})
            tid += 1
            for s in (o, e):
                s.create_task(t)
        do = sorted((d["ID"], d["NodeID"], d["Err"], d["State"]) for d in o.tick())
        de = sorted((d["ID"], d["NodeID"], d["Err"], d["State"]) for d in e.tick())
        assert do == de, (step, [(a, b) for a, b in zip(do, de) if a != b][:5])
    st = e.e.stats()
    assert st["n_nodes"] == 64 and st["n_words"] <= 2, st   # 64 nodes alive: at most 80 indices were ever in use at once
