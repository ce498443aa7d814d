"""Oracle KATs: manager/orchestrator/constraintenforcer/constraint_enforcer_test.go re-encoded.
The reference tests drive a store + watch queue; here every rejectNoncompliantTasks call is made explicit
(one call per node state) and the expected REJECTED ids are the ones the reference asserts."""
import orc

ACTIVE, PAUSE, DRAIN = 0, 1, 2


def named(kind, *vals):
    return [{"Named": {"Kind": kind, "Value": v}} for v in vals]


def discrete(kind, v):
    return [{"Discrete": {"Kind": kind, "Value": v}}]


def test_completed_job_tasks_do_not_consume_reservations():
    # TestRejectNoncompliantTasksIgnoresCompletedJobTasksInReservations, constraint_enforcer_test.go:15-89
    node = {"ID": "node1", "Spec": {"Availability": ACTIVE}, "Description": {"Resources": {"MemoryBytes": 1024}}}
    running = {"ID": "running1", "NodeID": "node1", "ServiceID": "svc1", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
               "Spec": {"Resources": {"Reservations": {"MemoryBytes": 700}}}}
    job = {"ID": "job1", "NodeID": "node1", "ServiceID": "jobsvc", "DesiredState": orc.COMPLETE, "Status": {"State": orc.COMPLETE},
           "Spec": {"Resources": {"Reservations": {"MemoryBytes": 700}}}}
    assert orc.enforce(node, [job, running]) == []
    # and the counter-example: a live second task of the same size does not fit
    live = dict(job, ID="live2", DesiredState=orc.RUNNING, Status={"State": orc.RUNNING})
    assert orc.enforce(node, [live, running]) == ["running1"]


def _enforcer_fixture():
    n1 = {"ID": "id1", "Spec": {"Annotations": {"Name": "name1"}, "Availability": ACTIVE}, "Status": {"State": orc.READY}, "Role": "WORKER"}
    n2 = {"ID": "id2", "Spec": {"Annotations": {"Name": "name2"}, "Availability": ACTIVE}, "Status": {"State": orc.READY},
          "Description": {"Resources": {"NanoCPUs": 10**9, "MemoryBytes": 10**9}}}
    tasks = [
        {"ID": "id0", "DesiredState": orc.RUNNING, "Spec": {"Placement": {"Constraints": ["node.role == manager"]}}, "Status": {"State": orc.NEW}, "NodeID": "id1"},
        {"ID": "id1", "DesiredState": orc.RUNNING, "Status": {"State": orc.NEW}, "NodeID": "id1"},
        {"ID": "id5", "DesiredState": orc.COMPLETE, "Status": {"State": orc.NEW}, "NodeID": "id1"},
        {"ID": "id2", "DesiredState": orc.RUNNING, "Spec": {"Placement": {"Constraints": ["node.role == worker"]}}, "Status": {"State": orc.RUNNING}, "NodeID": "id1"},
        {"ID": "id3", "DesiredState": orc.NEW, "Status": {"State": orc.NEW}, "NodeID": "id2"},
        {"ID": "id4", "DesiredState": orc.READY_T, "Spec": {"Resources": {"Reservations": {"MemoryBytes": 9 * 10**8}}}, "Status": {"State": orc.PENDING}, "NodeID": "id2"},
    ]
    return n1, n2, tasks


def by_node(tasks, nid):
    return sorted((t for t in tasks if t["NodeID"] == nid), key=lambda t: t["ID"])


def test_constraint_enforcer_sequence():
    # TestConstraintEnforcer, constraint_enforcer_test.go:91-289
    n1, n2, tasks = _enforcer_fixture()
    assert orc.enforce(n1, by_node(tasks, "id1")) == ["id0"]          # :238-241 id0 rejected immediately (id1 is a worker)
    assert orc.enforce(n2, by_node(tasks, "id2")) == []
    rest = [t for t in tasks if t["ID"] != "id0"]
    n1m = dict(n1, Role="MANAGER")                                     # :243-252
    assert orc.enforce(n1m, by_node(rest, "id1")) == ["id2"]           # :256-258
    n2s = dict(n2, Description={"Resources": {"NanoCPUs": 10**9, "MemoryBytes": 5 * 10**8}})   # :260-270
    assert orc.enforce(n2s, by_node(rest, "id2")) == ["id4"]           # :272-274


def test_paused_and_drained_nodes_are_left_alone():
    # constraint_enforcer.go:66-72
    n1, _, tasks = _enforcer_fixture()
    for av in (PAUSE, DRAIN):
        assert orc.enforce(dict(n1, Spec=dict(n1["Spec"], Availability=av)), by_node(tasks, "id1")) == []


def test_outdated_task_placement_constraints():
    # TestOutdatedTaskPlacementConstraints, constraint_enforcer_test.go:290-373: the SERVICE's current constraints count
    node = {"ID": "id0", "Spec": {"Annotations": {"Name": "node1", "Labels": {"foo": "bar"}}, "Availability": ACTIVE}, "Status": {"State": orc.READY}, "Role": "WORKER"}
    service = {"ID": "id1", "Spec": {"Annotations": {"Name": "service1"}, "Task": {"Placement": {"Constraints": ["node.labels.foo == bar"]}}}}
    task = {"ID": "id2", "Spec": {}, "ServiceID": "id1", "NodeID": "id0", "Status": {"State": orc.RUNNING}, "DesiredState": orc.RUNNING}
    assert orc.enforce(node, [task], {"id1": service}) == []
    bare = dict(node, Spec={"Annotations": {"Name": "node1", "Labels": {}}, "Availability": ACTIVE})
    assert orc.enforce(bare, [task], {"id1": service}) == ["id2"]
    # a task whose service is gone falls back to its own (here: empty) spec
    assert orc.enforce(bare, [task], {}) == []
    # an unparsable service constraint list is ignored altogether (`constraints, _ := constraint.Parse`, :163)
    broken = {"ID": "id1", "Spec": {"Task": {"Placement": {"Constraints": ["node.labels.foo == bar", "what is this"]}}}}
    assert orc.enforce(bare, [task], {"id1": broken}) == []


def test_generic_resources_named_and_discrete():
    # TestGenericResourcesPlacementConstraints(:375-469) / ...Discrete(:471-586)
    node = {"ID": "id0", "Spec": {"Annotations": {"Name": "node1"}, "Availability": ACTIVE}, "Status": {"State": orc.READY}, "Role": "WORKER",
            "Description": {"Resources": {"Generic": named("mygeneric", "1")}}}
    task = {"ID": "id2", "Spec": {"Resources": {"Reservations": {"Generic": named("mygeneric", "1")}}}, "ServiceID": "id1", "NodeID": "id0",
            "Status": {"State": orc.RUNNING}, "DesiredState": orc.RUNNING, "AssignedGenericResources": named("mygeneric", "1")}
    assert orc.enforce(node, [task]) == []
    swapped = dict(node, Description={"Resources": {"Generic": named("mygeneric", "2")}})
    assert orc.enforce(swapped, [task]) == ["id2"]
    dnode = dict(node, Description={"Resources": {"Generic": discrete("mygeneric", 2)}})
    dtask = dict(task, AssignedGenericResources=discrete("mygeneric", 2), Spec={"Resources": {"Reservations": {"Generic": discrete("mygeneric", 2)}}})
    assert orc.enforce(dnode, [dtask]) == []
    assert orc.enforce(dict(node, Description={"Resources": {"Generic": discrete("mygeneric", 1)}}), [dtask]) == ["id2"]
    # `break loop` (:193): after a generic-resource rejection the remaining tasks of the node are not looked at
    late = {"ID": "id9", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING}, "NodeID": "id0", "Spec": {"Placement": {"Constraints": ["node.role == manager"]}}}
    assert orc.enforce(swapped, [task, late]) == ["id2"]
    assert orc.enforce(swapped, [late, task]) == ["id9", "id2"]


def test_resource_accounting_is_sequential_in_store_order():
    node = {"ID": "n", "Spec": {"Availability": ACTIVE}, "Description": {"Resources": {"NanoCPUs": 4 * 10**9, "MemoryBytes": 1000}}}
    def t(i, cpu, mem):
        return {"ID": "t%d" % i, "NodeID": "n", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
                "Spec": {"Resources": {"Reservations": {"NanoCPUs": cpu, "MemoryBytes": mem}}}}
    tasks = [t(0, 10**9, 400), t(1, 10**9, 400), t(2, 10**9, 400), t(3, 3 * 10**9, 100), t(4, 10**9, 200)]
    # t2: memory 400 > 200 left; t3: cpu 3e9 > 2e9 left; t4 fits what t2/t3 did not take
    assert orc.enforce(node, tasks) == ["t2", "t3"]
    # a node without Description.Resources has zero capacity (:101-106)
    assert orc.enforce({"ID": "n", "Spec": {"Availability": ACTIVE}}, tasks[:1]) == ["t0"]
