"""CPU: the C++ host layer (csrc/swp_sched.cpp) against the ORACLE, end to end, without an engine. The seeded event scripts of the GPU fuzz
(tests/test_engine_fuzz.py: grouped and one-off services with random filters and spread preferences, several ticks, node drains /
removals / re-adds, task deletions) run through the oracle; every tick the oracle's decisions are scripted into the engine double
(fakelib.script: the node per task; fakelib.script_hist: the Pipeline counters behind a "no suitable node (...)" string, read back from
the oracle's own explanation), the host layer runs the same tick above the double and must report the same decisions — node, state and
error string. What is checked is everything the host layer does itself: event handling, queues and groups, descriptors and their
order, noSuitableNode and Explain, the bookkeeping behind the next tick. (What the engine decides is the GPU suite's business.)"""
import copy
import os
import random
import re

import pytest

import fakelib
import orc
import test_engine_fuzz as fz
from swarmkit_amd import abi, sched as swsched

# Filter.Explain (filter.go:46-58, 96-103, 204-210, 239-245, 308-314, 349-355, 377-379, 434-440), pipeline order
PHRASES = [(0, r"(\d+) nodes? not available for new tasks"), (1, r"insufficient resources on (\d+) nodes?"), (2, r"missing plugin on (\d+) nodes?"),
           (3, r"scheduling constraints not satisfied on (\d+) nodes?"), (4, r"unsupported platform on (\d+) nodes?"),
           (5, r"host-mode port already in use on (\d+) nodes?"), (6, r"max replicas per node limit exceed"),
           (7, r"cannot fulfill requested CSI volume mounts on (\d+) nodes?")]


def counters_of(err):
    """The Pipeline counters behind an explanation (pipeline.go:82-103: reasons by count, descending, stable in filter order). The
    MaxReplicas reason prints no count: any count that puts it where it stands will do."""
    hist = [0] * 8
    m = re.fullmatch(r"no suitable node(?: \((.*)\))?", err)
    assert m, err
    if not m.group(1):
        return hist
    parts = m.group(1).split("; ")
    items = []
    for part in parts:
        for f, pat in PHRASES:
            mm = re.fullmatch(pat, part)
            if mm:
                items.append((f, int(mm.group(1)) if mm.groups() else None))
                break
        else:
            raise AssertionError("unknown reason %r" % part)
    for pos, (f, c) in enumerate(items):
        if c is None:   # MaxReplicas: above everything behind it (behind a later filter an equal count is enough)
            nxt = items[pos + 1] if pos + 1 < len(items) else None
            c = 1 if nxt is None else (nxt[1] if nxt[0] > f else nxt[1] + 1)
        hist[f] = c
    return hist


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWP_REPLAY_SEEDS", "48"))))
def test_the_host_layer_replays_the_oracles_event_scripts(seed):
    rng = random.Random(0xC0FFEE + seed)   # the GPU fuzz's scripts, seed for seed
    o = orc.Oracle()
    e = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
    rng.choice([0, 0, 7, 64, 300])        # (the engine's window there: no engine here)
    both = (o, e)
    n_nodes = rng.choice([1, 3, 17, 64, 65, 200, 700])
    nodes = {i: fz.node_doc(rng, i) for i in range(n_nodes)}
    for d in nodes.values():
        for s in both:
            s.create_node(copy.deepcopy(d))
    n_svc = rng.randrange(1, 12)
    specs = [fz.service_spec(rng) for _ in range(n_svc)]
    grouped = [rng.random() < 0.5 for _ in range(n_svc)]
    for k in range(n_svc):
        for s in both:
            s.set_service("svc%02d" % k)
    placed, tid, docs = {}, 0, {}
    n_decisions = n_placed = n_explained = 0
    for rnd in range(rng.randrange(2, 6)):
        for _ in range(rng.randrange(1, 5)):
            k = rng.randrange(n_svc)
            for _ in range(rng.choice([1, 2, 5, 20, 60, 150])):
                t = {"ID": "t%06d" % tid, "ServiceID": "svc%02d" % k, "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
                if grouped[k]:
                    t["SpecVersion"] = {"Index": 1}
                t.update(specs[k])
                docs[t["ID"]] = t
                for s in both:
                    s.create_task(copy.deepcopy(t))
                tid += 1
        do = {d["ID"]: d for d in o.tick()}
        for t_id in sorted(do):   # the host layer hands the tasks of a service over in queue order = id order here
            d = do[t_id]
            if d["NodeID"]:
                fakelib.script(e.e, docs[t_id]["ServiceID"], d["NodeID"])
            else:
                fakelib.script_hist(e.e, docs[t_id]["ServiceID"], counters_of(d["Err"]))
        de = {d["ID"]: d for d in e.tick()}
        assert sorted(do) == sorted(de), (seed, rnd)
        for t_id, a in do.items():
            b = de[t_id]
            assert (a["NodeID"], a["State"], a["Err"]) == (b["NodeID"], b["State"], b["Err"]), (seed, rnd, t_id, a, b)
            n_decisions += 1
            if a["NodeID"] and a["State"] >= orc.ASSIGNED:
                placed[t_id] = a["NodeID"]
                n_placed += 1
            elif "(" in a["Err"]:
                n_explained += 1
        for _ in range(rng.randrange(0, 4)):
            act = rng.random()
            i = rng.randrange(n_nodes)
            if act < 0.35 and i in nodes:
                d = dict(nodes[i], Spec=dict(nodes[i]["Spec"], Availability=rng.choice([0, 1, 2])))
                nodes[i] = d
                for s in both:
                    s.update_node(copy.deepcopy(d))
            elif act < 0.5 and i in nodes:
                for s in both:
                    s.delete_node(nodes[i]["ID"])
                for t in [t for t, nid in placed.items() if nid == nodes[i]["ID"]]:
                    del placed[t]
                del nodes[i]
            elif act < 0.6 and i not in nodes:
                nodes[i] = fz.node_doc(rng, i)
                for s in both:
                    s.create_node(copy.deepcopy(nodes[i]))
            elif placed:
                t = rng.choice(sorted(placed))
                d = dict(docs[t], NodeID=placed[t], Status={"State": orc.RUNNING})
                for s in both:
                    s.delete_task(copy.deepcopy(d))
                del placed[t]
    # the node rows the two ended with (nodeSet.nodeInfo): task counts per node and service, residuals
    for i, d in nodes.items():
        io, ie = o.node_info(d["ID"]), e.node_info(d["ID"])
        assert (io is None) == (ie is None)
        if io is not None:
            nz = lambda m: {k: c for k, c in m.items() if c}   # (a service whose last task left a node reads 0 in the oracle's map, as in the reference's)
            assert (io["ActiveTasksCount"], nz(io["ActiveTasksCountByService"]), io["AvailableResources"]["NanoCPUs"], io["AvailableResources"]["MemoryBytes"], sorted(io["Tasks"])) == \
                (ie["ActiveTasksCount"], nz(ie["ActiveTasksCountByService"]), ie["AvailableResources"]["NanoCPUs"], ie["AvailableResources"]["MemoryBytes"], sorted(ie["Tasks"])), (seed, d["ID"], io, ie)
    print("decisions %d, placed %d, explained %d" % (n_decisions, n_placed, n_explained))
