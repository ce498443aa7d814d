"""CPU: the host layers' generic-resource bookkeeping (csrc/swp_generic.hpp inside swp::Scheduler, swarmkit_amd/generic.py inside the
Python twin) against the oracle's restatement of api/genericresource.
  1. the Python twin's list functions, one by one, on random lists (claim / consume / reclaim / sanitize / counts vs HasEnough);
  2. both host layers driven through the reference's event handlers (nodes with generic resources, tasks that arrive already
     assigned — NodeInfo.addTask claims —, deletions that carry AssignedGenericResources — removeTask reclaims —, node updates
     that change the description — createOrUpdateNode consumes, sanitize resets) over the scripted engine double: the available list
     every node_info reports must equal the oracle's, and the two twins must make the same engine calls."""
import random

import pytest

import pyhost

import fakelib
import orc
from swarmkit_amd import abi
import pygeneric as gres
from swarmkit_amd import host as swhost
from swarmkit_amd import sched as swsched

KINDS = ["apple", "orange", "gpu"]
NAMES = ["red", "blue", "green", "x", "y"]


def rand_list(rng, n, discrete_once=True):
    out, have_d = [], set()
    for _ in range(n):
        k = rng.choice(KINDS)
        if rng.random() < 0.5:
            if discrete_once and k in have_d:
                continue
            have_d.add(k)
            out.append({"Discrete": {"Kind": k, "Value": rng.randrange(0, 7)}})
        else:
            out.append({"Named": {"Kind": k, "Value": rng.choice(NAMES)}})
    return out


def T(lst):
    return gres.decode(lst)[0]


@pytest.mark.parametrize("seed", range(200))
def test_python_list_functions_match_the_oracle(seed):
    rng = random.Random(seed)
    node = rand_list(rng, rng.randrange(0, 8), discrete_once=rng.random() < 0.8)
    res = rand_list(rng, rng.randrange(0, 5))
    node_res = rand_list(rng, rng.randrange(0, 7))
    # ConsumeNodeResources
    assert gres.encode(gres.consume(T(node), T(res))) == orc.generic("consume", node=node, res=res)["node"]
    # Claim with Discrete reservations (what ValidateTask lets through)
    want = [{"Discrete": {"Kind": k, "Value": rng.randrange(0, 4)}} for k in rng.sample(KINDS, rng.randrange(0, 3))]
    o = orc.generic("claim", node=node, res=want)
    avail, assigned = gres.claim(T(node), T(want))
    assert gres.encode(avail) == o["node"] and gres.encode(assigned) == o["assigned"]
    # Reclaim (reclaimResources + sanitize) and sanitize alone
    assert gres.encode(gres.reclaim(T(node), T(res), T(node_res))) == orc.generic("reclaim", node=node, assigned=res, node_res=node_res)["node"]
    assert gres.encode(gres.sanitize(T(node_res), T(node))) == orc.generic("sanitize", node=node, node_res=node_res)["node"]
    # counts: a request of v >= 1 fits iff HasEnough says so
    c = gres.counts(T(node))
    for k in KINDS:
        for v in (1, 2, 3, 6):
            assert (c.get(k, 0) >= v) == orc.generic("has_enough", node=node, res=[{"Discrete": {"Kind": k, "Value": v}}])["ok"], (k, v, node)


def node_doc(rng, i):
    return {"ID": "n%d" % i, "Status": {"State": orc.READY}, "Spec": {"Availability": 0},
            "Description": {"Resources": {"NanoCPUs": 8 * 10**9, "MemoryBytes": 1 << 34, "Generic": rand_list(rng, rng.randrange(0, 7))}}}


def hosts():
    lib = fakelib.build()
    return [orc.Oracle(), swsched.Scheduler(engine=abi.Engine(lib_path=lib)), pyhost.PyHostScheduler(engine=abi.Engine(lib_path=lib))]


@pytest.mark.parametrize("seed", range(60))
def test_event_scripts_keep_the_available_lists_of_both_twins_equal_to_the_oracle(seed):
    rng = random.Random(1000 + seed)
    hs = hosts()
    n_nodes = rng.randrange(1, 4)
    docs = {i: node_doc(rng, i) for i in range(n_nodes)}
    for d in docs.values():
        for s in hs:
            s.create_node(d)
    for s in hs:
        s.set_service("svc")
    live = {}   # task id -> doc as the store would hold it (with AssignedGenericResources)

    def check():
        for i in docs:
            infos = [s.node_info("n%d" % i) for s in hs]
            gen = [x["AvailableResources"]["Generic"] for x in infos]
            assert gen[0] == gen[1] == gen[2], (seed, i, gen)
            assert infos[0]["AvailableResources"]["NanoCPUs"] == infos[1]["AvailableResources"]["NanoCPUs"] == infos[2]["AvailableResources"]["NanoCPUs"]

    check()
    tid = 0
    for step in range(rng.randrange(5, 25)):
        act = rng.random()
        if act < 0.5:       # a task arrives already assigned and running: NodeInfo.addTask -> Claim
            i = rng.randrange(n_nodes)
            want = [{"Discrete": {"Kind": k, "Value": rng.randrange(1, 4)}} for k in rng.sample(KINDS, rng.randrange(0, 3))]
            t = {"ID": "t%03d" % tid, "ServiceID": "svc", "NodeID": "n%d" % i, "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
                 "Spec": {"Resources": {"Reservations": {"NanoCPUs": 10**9, "Generic": want}}}}
            tid += 1
            before = hs[0].node_info("n%d" % i)["AvailableResources"]["Generic"]
            for s in hs:
                s.create_task(t)
            # what the task holds now = what Claim took (the store object the next events would carry)
            got = orc.generic("claim", node=before, res=want)["assigned"]
            live[t["ID"]] = dict(t, AssignedGenericResources=got)
        elif act < 0.8 and live:   # the task is deleted: removeTask -> Reclaim + sanitize
            k = rng.choice(sorted(live))
            t = live.pop(k)
            for s in hs:
                s.delete_task(t)
        else:               # the node's description changes: createOrUpdateNode consumes the tasks' resources from the new list
            i = rng.randrange(n_nodes)
            docs[i] = node_doc(rng, i)
            for s in hs:
                s.update_node(docs[i])
        check()
    logs = [fakelib.take_log(s.e) for s in hs[1:]]
    assert logs[0] == logs[1]


@pytest.mark.parametrize("seed", range(20))
def test_preassigned_decisions_carry_what_claim_assigned(seed):
    """processPreassignedTasks' decisions must carry AssignedGenericResources (taskFitNode hands addTask the very task the
    decision holds, scheduler.go:676-688) — named values included — exactly as the oracle's decision.new does."""
    rng = random.Random(7000 + seed)
    hs = hosts()
    docs = {i: node_doc(rng, i) for i in range(3)}
    for d in docs.values():
        for s in hs:
            s.create_node(d)
    for s in hs:
        s.set_service("fits-svc")
    for tid in range(8):
        i = rng.randrange(3)
        avail = gres.counts(T(hs[0].node_info("n%d" % i)["AvailableResources"]["Generic"]))
        kinds = [k for k in KINDS if avail.get(k, 0) >= 1]
        want = [{"Discrete": {"Kind": k, "Value": 1}} for k in rng.sample(kinds, min(len(kinds), rng.randrange(0, 3)))]
        t = {"ID": "p%03d" % tid, "ServiceID": "fits-svc", "NodeID": "n%d" % i, "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING},
             "Spec": {"Resources": {"Reservations": {"NanoCPUs": 10**8, "Generic": want}}}}
        for s in hs:
            s.create_task(t)
        ds = [sorted((d["ID"], d["NodeID"], d["State"], str(d.get("AssignedGenericResources"))) for d in s.process_preassigned()) for s in hs]
        assert ds[0] == ds[1] == ds[2], (seed, tid, ds)
        if want:
            assert "Kind" in ds[0][0][3]
