"""GPU: seeded random event scripts through the oracle and the engine — mixed grouped / one-off services with
random filters and spread preferences, several ticks, node drains / removals / re-adds and task deletions in
between. Every tick's decisions (node, error string, state) must agree."""
import os
import random

import pytest

import orc
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu

GIB = 1 << 30
ZONES = ["a", "b", "c", ""]
RACKS = ["r1", "r2", "r3"]


def node_doc(rng, i):
    labels = {}
    z = rng.choice(ZONES)
    if z:
        labels["zone"] = z
    if rng.random() < 0.7:
        labels["rack"] = rng.choice(RACKS)
    labels["disk"] = rng.choice(["ssd", "hdd", "SSD"])
    eng = {}
    if rng.random() < 0.4:
        eng["Plugins"] = [{"Type": "Network", "Name": "overlay"}] + ([{"Type": "Volume", "Name": "nfs:latest"}] if rng.random() < 0.5 else [])
    if rng.random() < 0.3:
        eng["Labels"] = {"tier": rng.choice(["gold", "silver"])}
    d = {"ID": "n%05d" % i,
         "Spec": {"Annotations": {"Name": "node%d" % i, "Labels": labels}, "Availability": 0 if rng.random() < 0.93 else rng.choice([1, 2])},
         "Status": {"State": orc.READY if rng.random() < 0.95 else 1, "Addr": "10.0.%d.%d" % (i >> 8, i & 255)},
         "Description": {"Hostname": "h%d" % i,
                         "Platform": {"Architecture": rng.choice(["amd64", "x86_64", "arm64", "aarch64"]), "OS": rng.choice(["linux", "linux", "linux", "windows"])},
                         "Resources": {"NanoCPUs": rng.choice([2, 4, 8, 16]) * 10**9, "MemoryBytes": rng.choice([4, 8, 16, 64]) * GIB},
                         "Engine": eng}}
    if rng.random() < 0.04:
        del d["Description"]["Platform"]
    return d


def service_spec(rng):
    spec, t = {}, {}
    if rng.random() < 0.8:
        spec["Resources"] = {"Reservations": {"NanoCPUs": rng.choice([0, 250, 500, 1000, 3000]) * 10**6, "MemoryBytes": rng.choice([0, 256, 1024, 6144]) << 20}}
    pl = {}
    cons = []
    r = rng.random()
    if r < 0.25:
        cons.append("node.labels.zone==%s" % rng.choice(["a", "b", "c", "nowhere"]))
    elif r < 0.35:
        cons.append("node.labels.disk!=hdd")
    elif r < 0.42:
        cons.append("engine.labels.tier==gold")
    elif r < 0.47:
        cons.append("node.platform.os==linux")
    if cons:
        pl["Constraints"] = cons
    r = rng.random()
    if r < 0.35:
        pl["Platforms"] = [{"Architecture": "amd64", "OS": "linux"}]
    elif r < 0.5:
        pl["Platforms"] = [{"Architecture": "arm64", "OS": "linux"}, {"Architecture": "", "OS": "windows"}]
    if rng.random() < 0.15:
        pl["MaxReplicas"] = rng.choice([1, 2, 3])
    r = rng.random()
    if r < 0.2:
        pl["Preferences"] = [{"Spread": {"SpreadDescriptor": "node.labels.zone"}}]
    elif r < 0.3:
        pl["Preferences"] = [{"Spread": {"SpreadDescriptor": "node.labels.zone"}}, {"Spread": {"SpreadDescriptor": "node.labels.rack"}}]
    if pl:
        spec["Placement"] = pl
    if spec:
        t["Spec"] = spec
    if rng.random() < 0.1:
        t["Networks"] = [{"Network": {"DriverState": {"Name": "overlay"}}}]
    if rng.random() < 0.1:
        t["Endpoint"] = {"Ports": [{"Protocol": 0, "PublishedPort": 8000 + rng.randrange(4), "PublishMode": 1}]}
    return t


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWP_FUZZ_FIRST", "0")), int(os.environ.get("SWP_FUZZ_FIRST", "0")) + int(os.environ.get("SWP_FUZZ_SEEDS", "64"))))   # SWP_FUZZ_SEEDS=1000 for a soak
def test_random_event_scripts(seed):
    rng = random.Random(0xC0FFEE + seed)
    rng.choice([0, 0, 7, 64, 300])   # (the scan window of rounds 1-2: the draw stays so that the seeds keep their scripts)
    o, e = orc.Oracle(), swhost.HostScheduler()
    both = (o, e)
    n_nodes = rng.choice([1, 3, 17, 64, 65, 200, 700])
    nodes = {i: node_doc(rng, i) for i in range(n_nodes)}
    for d in nodes.values():
        for s in both:
            s.create_node(d)
    n_svc = rng.randrange(1, 12)
    specs = [service_spec(rng) for _ in range(n_svc)]
    grouped = [rng.random() < 0.5 for _ in range(n_svc)]
    for k in range(n_svc):
        for s in both:
            s.set_service("svc%02d" % k)
    placed, tid = {}, 0
    docs = {}

    def tick():
        do = sorted((d["ID"], d["NodeID"], d["Err"], d["State"]) for d in o.tick())
        de = sorted((d["ID"], d["NodeID"], d["Err"], d["State"]) for d in e.tick())
        assert do == de, (seed, [(a, b) for a, b in zip(do, de) if a != b][:5])
        for i, nid, err, st in do:
            if nid and st >= orc.ASSIGNED:
                placed[i] = nid

    for rnd in range(rng.randrange(2, 6)):
        # new tasks
        for _ in range(rng.randrange(1, 5)):
            k = rng.randrange(n_svc)
            for _ in range(rng.choice([1, 2, 5, 20, 60, 150])):
                t = {"ID": "t%06d" % tid, "ServiceID": "svc%02d" % k, "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
                if grouped[k]:
                    t["SpecVersion"] = {"Index": 1}
                t.update(specs[k])
                docs[t["ID"]] = t
                for s in both:
                    s.create_task(t)
                tid += 1
        tick()
        # churn between ticks
        for _ in range(rng.randrange(0, 4)):
            act = rng.random()
            i = rng.randrange(n_nodes)
            if act < 0.35 and i in nodes:       # drain / pause / reactivate
                d = dict(nodes[i], Spec=dict(nodes[i]["Spec"], Availability=rng.choice([0, 1, 2])))
                nodes[i] = d
                for s in both:
                    s.update_node(d)
            elif act < 0.5 and i in nodes:      # node leaves
                for s in both:
                    s.delete_node(nodes[i]["ID"])
                gone = [t for t, nid in placed.items() if nid == nodes[i]["ID"]]
                for t in gone:
                    del placed[t]
                del nodes[i]
            elif act < 0.6 and i not in nodes:  # node comes back empty
                nodes[i] = node_doc(rng, i)
                for s in both:
                    s.create_node(nodes[i])
            elif placed:                        # a running task is removed
                t = rng.choice(sorted(placed))
                d = dict(docs[t], NodeID=placed[t], Status={"State": orc.RUNNING})
                for s in both:
                    s.delete_task(d)
                del placed[t]
    tick()
    for i in list(nodes)[:5]:
        a, b = o.node_info(nodes[i]["ID"]), e.node_info(nodes[i]["ID"])
        assert a["ActiveTasksCount"] == b["ActiveTasksCount"], (seed, i)
        assert a["AvailableResources"]["NanoCPUs"] == b["AvailableResources"]["NanoCPUs"], (seed, i)
        assert a["AvailableResources"]["MemoryBytes"] == b["AvailableResources"]["MemoryBytes"], (seed, i)
