"""Scheduler scenarios re-encoded from the reference's own tests
(/root/reference/manager/scheduler/scheduler_test.go; line numbers in each docstring).

Each scenario takes a factory returning a "scheduler under test" with the event-handler surface of
manager/scheduler.Scheduler (create_node / update_node / delete_node / create_task / update_task /
delete_task / set_service / tick). They run against the CPU oracle (tests/test_oracle_scheduler.py)
and, on a GPU box, against the HIP engine through its host mirror (tests/test_engine_scenarios.py).
A tick() returns decisions [{ID, NodeID, State, Err, Message}], which stands for the
EventUpdateTask stream the reference tests read with watchAssignment / watchAssignmentFailure.
"""
from collections import Counter

import orc  # enum constants only

READY_NODE = {"Status": {"State": orc.READY}}


def node(id_, **kw):
    n = {"ID": id_, "Status": {"State": orc.READY}}
    n.update(kw)
    return n


def labelled(id_, labels, **kw):
    return node(id_, Spec={"Annotations": {"Labels": labels}}, **kw)


def pending(id_, service="", spec_version=None, **kw):
    t = {"ID": id_, "ServiceID": service, "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
    if spec_version is not None:
        t["SpecVersion"] = {"Index": spec_version}
    t.update(kw)
    return t


def assignments(decisions):
    """watchAssignment filter: State in [ASSIGNED, RUNNING] and NodeID != ''."""
    return [d for d in decisions if orc.ASSIGNED <= d["State"] <= orc.RUNNING and d["NodeID"]]


def failures(decisions):
    """watchAssignmentFailure filter: State < ASSIGNED."""
    return [d for d in decisions if d["State"] < orc.ASSIGNED]


def commit(s, decisions):
    """The store commit echoes every assigned task back as EventUpdateTask (scheduler.go:192-195)."""
    return decisions


# ---------------------------------------------------------------------------------------------
def scenario_basic(factory):
    """TestScheduler, scheduler_test.go:22-369 (first phases)."""
    s = factory()
    for i in (1, 2, 3):
        s.create_node(node(f"id{i}"))
    s.create_task({"ID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.ASSIGNED}, "NodeID": "id1"})
    s.create_task(pending("id2"))
    s.create_task(pending("id3"))
    a = assignments(s.tick())
    # must assign to id2 / id3 since id1 already has a task (:125-133)
    assert sorted(d["NodeID"] for d in a) == ["id2", "id3"]
    # canonical order pins which one: lowest node index wins ties
    assert [d["NodeID"] for d in a] == ["id2", "id3"]

    # :135-168 — delete the task on id1 then add a task: goes to id1
    s.delete_task({"ID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.ASSIGNED}, "NodeID": "id1"})
    s.create_task(pending("id4"))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["id1"]

    # :170-191 — new node id4 is READY and empty → gets the next task
    s.create_node(node("id4"))
    s.create_task(pending("id5"))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["id4"]

    # :193-262 — a node that is not READY is never picked; once READY it is used
    s.create_node({"ID": "id5", "Status": {"State": orc.DOWN}})
    s.create_task(pending("id6"))
    a = assignments(s.tick())
    assert a[0]["NodeID"] != "id5"
    s.update_node(node("id5"))
    s.create_task(pending("id7"))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["id5"]

    # :333-368 — a node created READY and immediately taken DOWN never gets the next task
    s.create_node(node("id6"))
    s.update_node({"ID": "id6", "Status": {"State": orc.DOWN}})
    s.create_task(pending("id8"))
    a = assignments(s.tick())
    assert len(a) == 1 and a[0]["NodeID"] != "id6"
    # (extra) a deleted node never gets tasks either
    s.create_node(node("id7"))
    s.delete_node("id7")
    s.create_task(pending("id9"))
    a = assignments(s.tick())
    assert len(a) == 1 and a[0]["NodeID"] not in ("id6", "id7")


def scenario_ha(factory, use_spec_version):
    """testHA, scheduler_test.go:371-648."""
    s = factory()
    sv = 1 if use_spec_version else None
    for i in range(1, 6):
        s.create_node(node(f"id{i}"))
    for i in range(18):
        s.create_task(pending(f"t1id{i}", "service1", sv))
    t1 = Counter(d["NodeID"] for d in assignments(s.tick()))
    assert len(t1) == 5
    assert sorted(t1.values()) == [3, 3, 4, 4, 4]   # :495-496

    for i in range(2):
        s.create_task(pending(f"t2id{i}", "service2", sv))
    t2 = Counter(d["NodeID"] for d in assignments(s.tick()))
    assert len(t2) == 2
    for nid in t2:
        assert t1[nid] == 3   # :522-524

    for i in range(18, 21):
        s.create_task(pending(f"t1id{i}", "service1", sv))
    shared = []
    for d in assignments(s.tick()):
        assert t1[d["NodeID"]] != 5
        t1[d["NodeID"]] += 1
        if t2[d["NodeID"]]:
            shared.append(d["NodeID"])
    assert len(shared) == 2 and shared[0] != shared[1]
    assert sorted(t1.values()) == [4, 4, 4, 4, 5]   # :577-578

    s.create_task(pending("t2id4", "service2", sv))
    a = assignments(s.tick())
    assert len(a) == 1 and a[0]["ID"] == "t2id4"
    assert t2[a[0]["NodeID"]] == 0 and t1[a[0]["NodeID"]] != 5   # :593-598
    t2[a[0]["NodeID"]] += 1
    return s, t1, t2


def scenario_preferences(factory, use_spec_version):
    """testPreferences, scheduler_test.go:655-801 — fully deterministic outcome."""
    s = factory()
    sv = 1 if use_spec_version else None
    s.create_node(labelled("id1", {"az": "az1"}))
    for i in range(2, 6):
        s.create_node(labelled(f"id{i}", {"az": "az2"}))
    placement = {"Preferences": [{"Spread": {"SpreadDescriptor": "node.labels.az"}}]}
    for i in range(8):
        s.create_task(pending(f"t1id{i}", "service1", sv, Spec={"Placement": placement}))
    t1 = Counter(d["NodeID"] for d in assignments(s.tick()))
    assert dict(t1) == {"id1": 4, "id2": 1, "id3": 1, "id4": 1, "id5": 1}   # :795-800


def scenario_no_ready_nodes(factory):
    """TestSchedulerNoReadyNodes, scheduler_test.go:1263-1323."""
    s = factory()
    s.set_service("serviceID1")
    s.create_task(pending("id1", "serviceID1"))
    f = failures(s.tick())
    assert len(f) == 1 and f[0]["Err"] == "no suitable node"
    s.create_node(node("newnode"))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["newnode"]


def _res(cpu, mem, generic=None):
    r = {"NanoCPUs": int(cpu), "MemoryBytes": int(mem)}
    if generic is not None:
        r["Generic"] = generic
    return r


def named(kind, *vals):
    return [{"Named": {"Kind": kind, "Value": v}} for v in vals]


def discrete(kind, n):
    return [{"Discrete": {"Kind": kind, "Value": n}}]


def scenario_resource_constraint(factory, with_generic=True):
    """TestSchedulerResourceConstraint, scheduler_test.go:1617-1775."""
    s = factory()
    g = (lambda *a: sum(a, [])) if with_generic else (lambda *a: None)
    s.set_service("serviceID1")
    reservations = {"MemoryBytes": int(2e9)}
    if with_generic:
        reservations["Generic"] = discrete("orange", 2) + discrete("apple", 2)
    s.create_task(pending("id1", "serviceID1", Spec={"Resources": {"Reservations": reservations}}))
    s.create_node(node("underprovisioned", Description={"Resources": _res(1e9, 1e9, g(named("orange", "blue"), discrete("apple", 1)))}))
    for nid in ("nonready1", "nonready2"):
        s.create_node({"ID": nid, "Status": {"State": orc.UNKNOWN},
                       "Description": {"Resources": _res(2e9, 2e9, g(named("orange", "blue", "red"), discrete("apple", 2)))}})
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (2 nodes not available for new tasks; insufficient resources on 1 node)"   # :1742
    s.create_node(node("bignode", Description={"Resources": _res(4e9, 8e9, g(named("orange", "blue", "red", "green"), discrete("apple", 4)))}))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["bignode"]


def _plat(arch, os_):
    return {"Architecture": arch, "OS": os_}


def scenario_platform(factory):
    """TestSchedulerCompatiblePlatform, scheduler_test.go:2109-2340."""
    s = factory()
    s.set_service("serviceID1")
    s.create_node(node("node1", Description={"Platform": _plat("x86_64", "linux")}))
    s.create_node(node("node2", Description={"Platform": _plat("amd64", "windows")}))
    s.create_node(node("node3", Description={}))   # nil platform: cannot take anything with a platform constraint
    s.create_task(pending("id1", "serviceID1", Spec={"Placement": {"Platforms": [_plat("amd64", "linux")]}}))
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["node1"]   # x86_64 == amd64 after normalisation
    s.create_task(pending("id2", "serviceID1", Spec={"Placement": {"Platforms": [_plat("arm", "linux")]}}))
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (unsupported platform on 3 nodes)"   # :2311
    s.create_task(pending("id3", "serviceID1"))
    a = assignments(s.tick())
    assert a[0]["ID"] == "id3" and a[0]["NodeID"] in ("node2", "node3")
    s.create_task(pending("id4", "serviceID1", Spec={"Placement": {"Platforms": [_plat("", "linux")]}}))
    a = [d for d in assignments(s.tick()) if d["ID"] == "id4"]
    assert a[0]["NodeID"] == "node1"
    s.create_task(pending("id5", "serviceID1", Spec={"Placement": {"Platforms": [_plat("amd64", "linux"), _plat("x86_64", "windows")]}}))
    a = [d for d in assignments(s.tick()) if d["ID"] == "id5"]
    assert a[0]["NodeID"] in ("node1", "node2")


def _port(proto, port=58):
    return {"PublishMode": 1, "PublishedPort": port, "Protocol": proto}


def scenario_host_port(factory):
    """TestSchedulerHostPort, scheduler_test.go:3467-3626."""
    s = factory()
    s.set_service("serviceID1")
    s.create_task(pending("id1", "serviceID1", Endpoint={"Ports": [_port(0)]}))
    s.create_task(pending("id2", "serviceID1", Endpoint={"Ports": [_port(1)]}))
    assert len(failures(s.tick())) == 2
    s.create_node(node("nodeid1"))
    s.create_node(node("nodeid2"))
    a = assignments(s.tick())
    assert len(a) == 2 and a[0]["NodeID"] != a[1]["NodeID"]
    s.create_task(pending("id3", "serviceID1", Endpoint={"Ports": [_port(1), _port(0)]}))
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (host-mode port already in use on 2 nodes)"   # :3625


def scenario_max_replicas(factory):
    """TestSchedulerMaxReplicas, scheduler_test.go:3628-3879."""
    s = factory()
    s.set_service("serviceID1")
    mr1 = {"Placement": {"MaxReplicas": 1}}
    s.create_task(pending("id1", "serviceID1", Spec=mr1))
    s.create_task(pending("id2", "serviceID1", Spec=mr1))
    assert len(failures(s.tick())) == 2
    s.create_node(node("nodeid1"))
    s.create_node(node("nodeid2"))
    a = assignments(s.tick())
    assert len(a) == 2 and a[0]["NodeID"] != a[1]["NodeID"]
    s.create_task(pending("id3", "serviceID1", Spec=mr1))
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (max replicas per node limit exceed)"   # :3761
    s.create_node(node("nodeid3"))
    # id3 is still queued: it now fits on the new node
    a = assignments(s.tick())
    assert [d["NodeID"] for d in a] == ["nodeid3"]
    spec = {"Placement": {"Constraints": ["node.hostname==node1"], "MaxReplicas": 3}}
    for i in (4, 5, 6):
        s.create_task(pending(f"id{i}", "serviceID1", Spec=spec))
    s.tick()
    s.create_task(pending("id7", "serviceID1", Spec=spec))
    f = [d for d in failures(s.tick()) if d["ID"] == "id7"]
    assert f[0]["Err"] == "no suitable node (scheduling constraints not satisfied on 3 nodes)"   # :3878


def scenario_faulty_node(factory):
    """TestSchedulerFaultyNode, scheduler_test.go:1325-1476: ≥5 recent failures down-weight a node;
    pre-assigned tasks neither count nor care."""
    s = factory()
    s.create_node(node("id1"))
    s.create_node(node("id2"))
    s.create_task({"ID": "id1", "ServiceID": "service1", "NodeID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING}})
    s.create_task({"ID": "id2", "ServiceID": "service2", "NodeID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING}})
    for i in range(8):
        t = pending(f"r{i}", "service1")
        s.create_task(t)
        a = assignments(s.tick())
        assert len(a) == 1 and a[0]["ID"] == t["ID"]
        assert a[0]["NodeID"] == ("id2" if i < 5 else "id1"), (i, a)   # :1426-1430
        p = {"ID": f"p{i}", "ServiceID": "service2", "NodeID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
        s.create_task(p)
        pa = assignments(s.process_preassigned())
        assert len(pa) == 1 and pa[0]["NodeID"] == "id1"
        s.update_task(dict(t, NodeID=a[0]["NodeID"], Status={"State": orc.FAILED}))
        s.update_task(dict(p, Status={"State": orc.FAILED}))


def scenario_multiple_preferences(factory, use_spec_version, with_generic=True):
    """testMultiplePreferences, scheduler_test.go:808-1106: two spread levels + resources."""
    s = factory()
    sv = 1 if use_spec_version else None

    def n(i, az, rack, mem, apples):
        res = {"NanoCPUs": int(1e9), "MemoryBytes": int(mem)}
        if with_generic:
            res["Generic"] = discrete("apple", apples)
        return node(f"id{i}", Spec={"Annotations": {"Labels": {"az": az, "rack": rack}}}, Description={"Resources": res})
    s.create_node(n(0, "az1", "rack1", 1e8, 1))
    s.create_node(n(1, "az1", "rack1", 1e9, 10))
    for i in (2, 3, 4):
        s.create_node(n(i, "az2", "rack1", 1e9, 6))
    for i in (5, 6):
        s.create_node(n(i, "az2", "rack2", 1e9, 6))
    reservations = {"MemoryBytes": int(2e8)}
    if with_generic:
        reservations["Generic"] = discrete("apple", 2)
    spec = {"Placement": {"Preferences": [{"Spread": {"SpreadDescriptor": "node.labels.az"}},
                                          {"Spread": {"SpreadDescriptor": "node.labels.rack"}}]},
            "Resources": {"Reservations": reservations}}
    for i in range(12):
        s.create_task(pending(f"t1id{i}", "service1", sv, Spec=spec))
    t1 = Counter(d["NodeID"] for d in assignments(s.tick()))
    assert len(t1) == 6
    assert t1["id0"] == 0 and t1["id1"] == 5   # :1063-1068
    rack1 = t1["id2"] + t1["id3"] + t1["id4"]
    rack2 = t1["id5"] + t1["id6"]
    assert sorted((rack1, rack2)) == [3, 4]    # :1070-1105
    if rack1 == 4:
        assert sorted((t1["id2"], t1["id3"], t1["id4"])) == [1, 1, 2] and sorted((t1["id5"], t1["id6"])) == [1, 2]
    else:
        assert (t1["id2"], t1["id3"], t1["id4"]) == (1, 1, 1) and (t1["id5"], t1["id6"]) == (2, 2)
    return t1


# ---------------------------------------------------------------------------------------------
# The remaining scheduler_test.go scenarios (added after the first parity pass).
def running(id_, node_id, service="", spec_version=None, **kw):
    t = {"ID": id_, "ServiceID": service, "NodeID": node_id, "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING}}
    if spec_version is not None:
        t["SpecVersion"] = {"Index": spec_version}
    t.update(kw)
    return t


def scenario_multiple_preferences_scale_up(factory):
    """TestMultiplePreferencesScaleUp, scheduler_test.go:1115-1261: scaling a service up over an unbalanced two-level
    spread tree must terminate, and both new tasks land on id12 / id21."""
    s = factory()
    for nid, az, rack in (("id11", "dc1", "r1"), ("id12", "dc1", "r2"), ("id21", "dc2", "r1")):
        s.create_node(labelled(nid, {"az": az, "rack": rack}))
    s.set_service("service1", 1)
    spec = {"Placement": {"Preferences": [{"Spread": {"SpreadDescriptor": "node.labels.az"}}, {"Spread": {"SpreadDescriptor": "node.labels.rack"}}]}}
    for nid, cnt in (("id11", 3), ("id12", 1), ("id21", 3)):
        for i in range(cnt):
            s.create_task(running(f"t1running-{nid}-{i}", nid, "service1", 1, Spec=spec))
    for i in range(2):
        s.create_task(pending(f"t1id{i}", "service1", 1, Spec=spec))
    a = assignments(s.tick())
    assert len(a) == 2 and all(d["ID"].startswith("t1id") for d in a)
    c = Counter(d["NodeID"] for d in a)
    assert c["id12"] + c["id21"] == 2   # :1257-1260


def scenario_faulty_node_spec_version(factory):
    """TestSchedulerFaultyNodeSpecVersion, scheduler_test.go:1485-1615: failures are bucketed per (service, spec
    version); a service update starts a fresh bucket."""
    s = factory()
    s.create_node(node("id1"))
    s.create_node(node("id2"))
    s.set_service("service1", 1)
    s.create_task(running("id1", "id1", "service1", 1))
    for i in range(15):
        ver = 2 if i > 5 else 1
        if i == 6:
            s.set_service("service1", 2)
        t = pending(f"new{i}", "service1", ver)
        s.create_task(t)
        a = assignments(s.tick())
        assert len(a) == 1 and a[0]["ID"] == t["ID"]
        want = "id2" if (i < 5 or 5 < i < 11) else "id1"          # :1567-1578
        assert a[0]["NodeID"] == want, (i, a)
        n1, n2 = s.node_info("id1")["RecentFailures"], s.node_info("id2")["RecentFailures"]
        e11, e12, e21, e22 = 0, 0, i, 0                              # :1580-1591
        if i > 5:
            e11, e21, e22 = 1, 5, i - 6
        if i > 11:
            e12, e22 = i - 11, 5
        assert n1.get("service1@1", 0) == e11 and n1.get("service1@2", 0) == e12, (i, n1)
        assert n2.get("service1@1", 0) == e21 and n2.get("service1@2", 0) == e22, (i, n2)
        s.update_task(dict(t, NodeID=a[0]["NodeID"], Status={"State": orc.FAILED}))


def scenario_resource_constraint_ha(factory, with_generic=True):
    """TestSchedulerResourceConstraintHA, scheduler_test.go:1777-1917: id1 has fewer tasks but room for only one more."""
    s = factory()
    g = (lambda x: x) if with_generic else (lambda x: None)
    s.create_node(node("id1", Description={"Resources": _res(0, 1e9, g(discrete("apple", 2)))}))
    s.create_node(node("id2", Description={"Resources": _res(0, 1e11, g(discrete("apple", 5)))}))
    spec = {"Resources": {"Reservations": _res(0, 5e8, g(discrete("apple", 1)))}}
    for tid, nid in (("id1", "id1"), ("id2", "id2"), ("id3", "id2"), ("id4", "id2")):
        s.create_task(running(tid, nid, Spec=spec))
    s.create_task(pending("id5", Spec=spec))
    s.create_task(pending("id6", Spec=spec))
    a = assignments(s.tick())
    assert sorted(d["ID"] for d in a) == ["id5", "id6"]
    assert sorted(d["NodeID"] for d in a) == ["id1", "id2"]        # :1911-1915


def scenario_resource_constraint_dead_task(factory, with_generic=True):
    """TestSchedulerResourceConstraintDeadTask, scheduler_test.go:1919-2023: a task past RUNNING frees its reservations."""
    s = factory()
    g = (lambda x: x) if with_generic else (lambda x: None)
    s.set_service("serviceID1")
    s.create_node(node("id1", Description={"Resources": _res(1e9, 1e9, g(discrete("apple", 4)))}))
    spec = {"Resources": {"Reservations": _res(0, 8e8, g(discrete("apple", 3)))}}
    big1 = pending("id1", "serviceID1", Spec=spec)
    s.create_task(big1)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("id1", "id1")]
    s.create_task(pending("id2", "serviceID1", Spec=spec))
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (insufficient resources on 1 node)"   # :2008
    s.update_task(dict(big1, NodeID="id1", Status={"State": orc.SHUTDOWN}))
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("id2", "id1")]


def scenario_preexisting_dead_task(factory, with_generic=True):
    """TestSchedulerPreexistingDeadTask, scheduler_test.go:2025-2107: a task already past RUNNING never counted."""
    s = factory()
    g = (lambda x: x) if with_generic else (lambda x: None)
    s.create_node(node("id1", Description={"Resources": _res(1e9, 1e9, g(discrete("apple", 1)))}))
    spec = {"Resources": {"Reservations": _res(0, 8e8, g(discrete("apple", 1)))}}
    s.create_task({"ID": "id1", "NodeID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.SHUTDOWN}, "Spec": spec})
    s.create_task({"ID": "id2", "NodeID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}, "Spec": spec})
    a = assignments(s.process_preassigned())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("id2", "id1")]


def scenario_unassigned_map(factory):
    """TestSchedulerUnassignedMap, scheduler_test.go:2342-2419: a task that found no node stays queued until its service
    disappears; then it is dropped for good."""
    s = factory()
    s.set_service("serviceID1")
    s.create_task(pending("id1", "serviceID1", Spec={"Placement": {"Platforms": [_plat("amd64", "windows")]}}))
    s.create_node(node("node1", Description={"Platform": _plat("x86_64", "linux")}))
    f = failures(s.tick())
    assert [d["ID"] for d in f] == ["id1"] and f[0]["Err"] == "no suitable node (unsupported platform on 1 node)"
    f = failures(s.tick())
    assert [d["ID"] for d in f] == ["id1"]            # still in the unassigned map: tried again
    s.delete_service("serviceID1")
    assert s.tick() == []                              # noSuitableNode returns early and does not re-enqueue (:933-937)
    s.set_service("serviceID1")
    assert s.tick() == []                              # gone from the map


def scenario_preassigned_tasks(factory):
    """TestPreassignedTasks, scheduler_test.go:2421-2528."""
    s = factory()
    s.create_node(node("node1"))
    s.create_node(node("node2"))
    s.create_task(pending("task1"))
    s.create_task(pending("task2", NodeID="node1"))
    s.create_task(pending("task3", NodeID="node1"))
    pa = assignments(s.process_preassigned())
    assert sorted(d["ID"] for d in pa) == ["task2", "task3"] and {d["NodeID"] for d in pa} == {"node1"}
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task1", "node2")]   # node1 already holds two tasks


def scenario_ignore_tasks(factory):
    """TestIgnoreTasks, scheduler_test.go:2530-2616: desired state SHUTDOWN / REMOVE is not the scheduler's business."""
    s = factory()
    s.create_node(node("node1"))
    # the tasks are in the store before Run(): setupTasksList, not createTask events
    s.setup_task(pending("task1"))
    s.setup_task(dict(pending("task2", NodeID="node1"), DesiredState=orc.SHUTDOWN))
    s.setup_task(dict(pending("task3", NodeID="node1"), DesiredState=orc.REMOVE))
    assert assignments(s.process_preassigned()) == []
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task1", "node1")]


def scenario_unscheduleable_task(factory):
    """TestUnscheduleableTask, scheduler_test.go:2629-2832: a pending task of an outdated spec version whose desired
    state became terminal is shut down by the scheduler instead of waiting forever."""
    s = factory()
    s.create_node(node("nodeid1", Description={}))
    s.set_service("serviceid1", 0)
    spec = {"Placement": {"MaxReplicas": 1}}
    t1, t2 = pending("taskid1", "serviceid1", 0, Spec=spec), pending("taskid2", "serviceid1", 0, Spec=spec)
    s.create_task(t1)
    s.create_task(t2)
    d = s.tick()
    a, f = assignments(d), failures(d)
    assert len(a) == 1 and len(f) == 1
    assert f[0]["Err"] == "no suitable node (max replicas per node limit exceed)"   # :2754
    assigned, failed = (t1, t2) if a[0]["ID"] == "taskid1" else (t2, t1)
    s.update_task(dict(assigned, NodeID="nodeid1", Status={"State": orc.RUNNING}))
    s.update_task(dict(failed, DesiredState=orc.SHUTDOWN, Status={"State": orc.PENDING, "Err": f[0]["Err"]}))
    s.create_task(pending("taskid1update", "serviceid1", 1, Spec=spec))
    s.set_service("serviceid1", 1)
    d = s.tick()
    down = [x for x in d if x["ID"] == failed["ID"]]
    assert down and down[0]["State"] >= orc.SHUTDOWN            # :2806-2812


def _plugin_node(nid, plugins):
    return node(nid, Description={"Engine": {"Plugins": [{"Type": t, "Name": n} for t, n in plugins]}})


def _vol(source, driver):
    return {"Source": source, "Target": "/foo", "Type": "VOLUME", "VolumeOptions": {"DriverConfig": {"Name": driver}}}


def scenario_plugin_constraint(factory):
    """TestSchedulerPluginConstraint, scheduler_test.go:2865-3333: volume / network / log plugins incl. the Log-driver rules."""
    s = factory()
    n1 = _plugin_node("node1_ID", [("Volume", "plugin1"), ("Log", "default")])
    n2 = _plugin_node("node2_ID", [("Volume", "plugin1"), ("Volume", "plugin2"), ("Log", "default")])
    n3 = _plugin_node("node3_ID", [("Volume", "plugin1"), ("Network", "plugin1"), ("Log", "default")])
    n4 = _plugin_node("node4_ID", [("Log", "plugin1")])
    t0 = pending("task0_ID", "serviceID1", Spec={"Container": {"Mounts": [{"Source": "/src", "Target": "/foo", "Type": "BIND"}]}})
    t1 = pending("task1_ID", "serviceID1", Spec={"Container": {"Mounts": [_vol("testVol1", "plugin1")]}})
    t2 = pending("task2_ID", "serviceID1", Spec={"Container": {"Mounts": [_vol("testVol1", "plugin1"), _vol("testVol2", "plugin2")]}})
    t3 = pending("task3_ID", "serviceID1", Spec={"Container": {"Mounts": [_vol("testVol1", "plugin1")]}},
                 Networks=[{"Network": {"ID": "testNwID1", "DriverState": {"Name": "plugin1"}}}])
    t4 = pending("task4_ID", "serviceID1", Spec={"Container": {}, "LogDriver": {"Name": "plugin1"}})
    t5 = pending("task5_ID", "serviceID1", Spec={"Container": {}, "LogDriver": {"Name": "plugin1"}})
    t6 = pending("task6_ID", "serviceID1", Spec={"Container": {}, "LogDriver": {"Name": "none"}})
    t7 = pending("task7_ID", "serviceID1", Spec={"Container": {}, "LogDriver": {"Options": {"max-size": "50k"}}})
    s.set_service("serviceID1")
    s.create_task(t1)
    s.create_node(n1)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task1_ID", "node1_ID")]          # :3197-3198
    s.create_task(t0)                                                                   # bind mounts do not enable the filter
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task0_ID", "node1_ID")]          # :3208-3210
    s.create_task(t2)
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (missing plugin on 1 node)"               # :3245
    s.create_node(n2)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task2_ID", "node2_ID")]
    s.create_task(t3)
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (missing plugin on 2 nodes)"              # :3268
    s.create_node(n3)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task3_ID", "node3_ID")]
    s.create_task(t4)
    f = failures(s.tick())
    assert f[0]["Err"] == "no suitable node (missing plugin on 3 nodes)"              # :3292
    s.create_node(n4)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task4_ID", "node4_ID")]
    s.create_task(t5)
    a = assignments(s.tick())
    assert [(d["ID"], d["NodeID"]) for d in a] == [("task5_ID", "node4_ID")]
    for t in (t6, t7):                                                                  # "none" / unnamed log driver: any node
        s.create_task(t)
        a = assignments(s.tick())
        assert len(a) == 1 and a[0]["ID"] == t["ID"] and a[0]["NodeID"] != ""
