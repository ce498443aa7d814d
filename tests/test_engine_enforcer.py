"""GPU: swp_enforce (k_enforce) vs the oracle's rejectNoncompliantTasks — the reference's enforcer tests through the
engine, and seeded random clusters (labels, roles, constraints from current service specs, reservations, task states)."""
import os
import random

import pytest

import pyhost

import orc
import test_oracle_enforcer as kat
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu


def engine_enforce(node_docs, tasks, services=None):
    s = swhost.HostScheduler()
    for nd in node_docs:
        s.create_node(nd)
    tbn = {}
    for t in tasks:
        tbn.setdefault(t["NodeID"], []).append(t)
    return swhost.enforce(s, node_docs, tbn, services)


def oracle_enforce(node_docs, tasks, services=None):
    out = {}
    for nd in node_docs:
        if nd.get("Spec", {}).get("Availability", 0) not in (0, "ACTIVE", None):
            continue
        mine = sorted((t for t in tasks if t["NodeID"] == nd["ID"]), key=lambda t: t["ID"])
        out[nd["ID"]] = orc.enforce(nd, mine, services or {})
    return out


def test_reference_enforcer_sequence_on_the_device():
    n1, n2, tasks = kat._enforcer_fixture()
    assert engine_enforce([n1, n2], tasks) == {"id1": ["id0"], "id2": []}
    rest = [t for t in tasks if t["ID"] != "id0"]
    n1m = dict(n1, Role="MANAGER")
    n2s = dict(n2, Description={"Resources": {"NanoCPUs": 10**9, "MemoryBytes": 5 * 10**8}})
    assert engine_enforce([n1m, n2s], rest) == {"id1": ["id2"], "id2": ["id4"]}


def test_outdated_task_constraints_on_the_device():
    node = {"ID": "id0", "Spec": {"Annotations": {"Name": "node1", "Labels": {"foo": "bar"}}, "Availability": 0}, "Status": {"State": orc.READY}, "Role": "WORKER"}
    service = {"ID": "id1", "Spec": {"Task": {"Placement": {"Constraints": ["node.labels.foo == bar"]}}}}
    task = {"ID": "id2", "Spec": {}, "ServiceID": "id1", "NodeID": "id0", "Status": {"State": orc.RUNNING}, "DesiredState": orc.RUNNING}
    assert engine_enforce([node], [task], {"id1": service}) == {"id0": []}
    bare = dict(node, Spec={"Annotations": {"Name": "node1", "Labels": {}}, "Availability": 0})
    assert engine_enforce([bare], [task], {"id1": service}) == {"id0": ["id2"]}
    assert engine_enforce([bare], [task], {}) == {"id0": []}
    broken = {"ID": "id1", "Spec": {"Task": {"Placement": {"Constraints": ["node.labels.foo == bar", "what is this"]}}}}
    assert engine_enforce([bare], [task], {"id1": broken}) == {"id0": []}


def test_generic_resources_the_loop_ends_at_the_first_missing_assignment():
    """constraint_enforcer.go:186-200: a kept task claims its AssignedGenericResources from the node's list; the first task whose
    assignment is gone is rejected and `break loop` ends the node's walk — a later task that fails its constraint is NOT rejected."""
    node = {"ID": "id0", "Spec": {"Annotations": {"Labels": {"zone": "a"}}, "Availability": 0}, "Status": {"State": orc.READY},
            "Description": {"Resources": {"NanoCPUs": 10**10, "MemoryBytes": 10**10, "Generic": [{"Discrete": {"Kind": "gpu", "Value": 2}}, {"Named": {"Kind": "fpga", "Value": "f0"}}]}}}
    def task(i, gen, cons=None):
        t = {"ID": "t%d" % i, "NodeID": "id0", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING}, "Spec": {}}
        if gen is not None:
            t["AssignedGenericResources"] = gen
        if cons:
            t["Spec"]["Placement"] = {"Constraints": cons}
        return t
    tasks = [task(0, [{"Discrete": {"Kind": "gpu", "Value": 1}}]), task(1, [{"Named": {"Kind": "fpga", "Value": "f0"}}]), task(2, None, ["node.labels.zone==b"]),
             task(3, [{"Discrete": {"Kind": "gpu", "Value": 2}}]), task(4, None, ["node.labels.zone==b"]), task(5, [{"Named": {"Kind": "fpga", "Value": "f0"}}])]
    want = oracle_enforce([node], tasks)
    assert want == {"id0": ["t2", "t3"]}   # t2: constraint; t3: one gpu left, two assigned -> rejected, the loop ends: t4 (constraint) and t5 (fpga gone) stay
    assert engine_enforce([node], tasks) == want


STATES = [orc.NEW, orc.PENDING, orc.ASSIGNED, orc.READY_T, orc.RUNNING, orc.COMPLETE, orc.SHUTDOWN, orc.FAILED, orc.REJECTED]
CONS = ["node.labels.zone==a", "node.labels.zone!=b", "node.role==manager", "node.role != worker", "node.hostname==h3",
        "engine.labels.tier==gold", "node.platform.os==linux", "node.id!=n00002", "node.ip==10.0.0.0/24", "node.labels.disk == SSD",
        "bogus expr", "node.labels.zone==a"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWP_FUZZ_FIRST", "0")), int(os.environ.get("SWP_FUZZ_FIRST", "0")) + int(os.environ.get("SWP_FUZZ_SEEDS", "12"))))
def test_random_clusters(seed):
    rng = random.Random(0xE4F0 + seed)
    nodes = []
    for i in range(rng.choice([1, 5, 70, 300])):
        labels = {}
        if rng.random() < 0.8:
            labels["zone"] = rng.choice("abc")
        if rng.random() < 0.5:
            labels["disk"] = rng.choice(["ssd", "hdd", "SSD"])
        d = {"ID": "n%05d" % i, "Role": rng.choice(["WORKER", "WORKER", "MANAGER"]),
             "Spec": {"Annotations": {"Name": "x%d" % i, "Labels": labels}, "Availability": rng.choice([0, 0, 0, 0, 1, 2])},
             "Status": {"State": orc.READY, "Addr": "10.0.%d.%d" % (rng.randrange(2), i % 250)}}
        if rng.random() < 0.9:
            d["Description"] = {"Hostname": "h%d" % i, "Platform": {"Architecture": "amd64", "OS": rng.choice(["linux", "windows"])},
                                "Engine": {"Labels": {"tier": rng.choice(["gold", "tin"])}} if rng.random() < 0.5 else {}}
            if rng.random() < 0.85:
                d["Description"]["Resources"] = {"NanoCPUs": rng.choice([0, 1, 2, 4]) * 10**9, "MemoryBytes": rng.choice([0, 1, 4, 8]) << 30}
                if rng.random() < 0.6:
                    d["Description"]["Resources"]["Generic"] = [{"Discrete": {"Kind": "gpu", "Value": rng.randrange(0, 5)}}] + [{"Named": {"Kind": "fpga", "Value": "f%d" % q} } for q in range(rng.randrange(0, 3))]
        nodes.append(d)
    services = {}
    for k in range(rng.randrange(0, 8)):
        pl = {"Constraints": rng.sample(CONS, rng.randrange(0, 3))} if rng.random() < 0.7 else None
        services["s%d" % k] = {"ID": "s%d" % k, "Spec": {"Task": ({"Placement": pl} if pl is not None else {})}}
    tasks = []
    for j in range(rng.randrange(1, 12) * len(nodes)):
        t = {"ID": "t%06d" % rng.randrange(10**6), "NodeID": rng.choice(nodes)["ID"], "ServiceID": rng.choice(["s%d" % k for k in range(10)]),
             "DesiredState": rng.choice(STATES), "Status": {"State": rng.choice(STATES)}, "Spec": {}}
        if rng.random() < 0.6:
            t["Spec"]["Resources"] = {"Reservations": {"NanoCPUs": rng.choice([0, 5, 10, 20]) * 10**8, "MemoryBytes": rng.choice([0, 256, 1024, 3000]) << 20}}
        if rng.random() < 0.4:
            t["Spec"]["Placement"] = {"Constraints": rng.sample(CONS, rng.randrange(0, 3))}
        if rng.random() < 0.3:
            t["AssignedGenericResources"] = rng.choice([[], [{"Discrete": {"Kind": "gpu", "Value": rng.randrange(1, 3)}}], [{"Named": {"Kind": "fpga", "Value": "f%d" % rng.randrange(3)}}],
                                                        [{"Discrete": {"Kind": "gpu", "Value": 1}}, {"Named": {"Kind": "fpga", "Value": "f%d" % rng.randrange(3)}}]])
        tasks.append(t)
    tasks = list({t["ID"]: t for t in tasks}.values())
    assert engine_enforce(nodes, tasks, services) == oracle_enforce(nodes, tasks, services)


def test_node_matches_matrix_equals_per_pair_oracle():
    """Global orchestrator sweep (global.go:306,440,513): the (constraint set x node) NodeMatches matrix in one call,
    every cell against the oracle's ConstraintFilter on the same node doc; also the reference's constraint truth tables."""
    import numpy as np
    import kat_tables as kt
    rng = random.Random(77)
    s = swhost.HostScheduler()
    nodes = []
    for i in range(150):
        labels = {"zone": rng.choice("abc")} if rng.random() < 0.8 else {}
        d = {"ID": "n%05d" % i, "Role": rng.choice(["WORKER", "MANAGER"]), "Spec": {"Annotations": {"Name": "x", "Labels": labels}},
             "Status": {"State": orc.READY, "Addr": "10.0.%d.%d" % (rng.randrange(2), i)},
             "Description": {"Hostname": "h%d" % i, "Platform": {"Architecture": "amd64", "OS": rng.choice(["linux", "windows"])},
                             "Engine": {"Labels": {"tier": rng.choice(["gold", "tin"])}}}}
        nodes.append(d)
        s.create_node(d)
    lists = [[c] for c in CONS if c != "bogus expr"] + [["node.labels.zone==a", "node.role==manager"], ["node.labels.zone != a", "engine.labels.tier == GOLD"]]
    sets = [s.constraint_set(l) for l in lists] + [0]
    bm = s.e.node_matches(sets)
    for r, l in enumerate(lists + [[]]):
        for i, nd in enumerate(nodes):
            idx = s.e.intern(0, nd["ID"])   # SWP_SPACE_NODE_ID
            got = bool((int(bm[r, idx >> 6]) >> (idx & 63)) & 1)
            want = True if not l else orc.constraint_filter(l, nd)
            assert got == want, (l, nd["ID"])
    # constraint_test.go truth tables, one node at a time
    for cons, node, want in kt.constraint_cases():
        if want is None:
            continue
        s2 = swhost.HostScheduler()
        s2.create_node(node)
        parsed = pyhost.parse_constraints(cons)
        assert parsed is not None
        row = s2.e.node_matches([s2.constraint_set(cons)])
        assert bool(int(row[0, 0]) & 1) == want, (cons, node)
