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

    # :264-331 — deleted node never gets tasks
    s.create_node(node("id6"))
    s.delete_node("id6")
    s.create_task(pending("id8"))
    a = assignments(s.tick())
    assert a[0]["NodeID"] != "id6"


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
