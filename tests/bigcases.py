"""BASELINE-size event scripts shared by the offline digest generator (tests/golden/make_golden_big.py, CPU oracle)
and the GPU parity tests (tests/test_engine_bigcases.py, HIP engine behind the host scheduler layer).

Every script drives a scheduler object through the reference's event-handler names (create_node / update_node /
set_service / create_task / delete_task / tick — orc.Oracle and swarmkit_amd.host.HostScheduler expose the same
methods) and folds every tick's decisions into SHA-256 digests, so both sides run EXACTLY the same protocol.

  cfg4_full     BASELINE.json configs[3]: 1M one-off tasks x 100k nodes, every filter (synth cfg4)
  cfg3m_*       cfg3's cluster with hundreds of distinct reservations per batch (synth cfg3m): the demand-class rows of the block resolver
  cfg5_churn    BASELINE.json configs[4]: 100k tasks x 10k nodes placed, then 100 rounds of {reactivate the previous
                round's drained nodes, drain 10 % of the nodes, delete the tasks on them, create as many new tasks, tick}
  refbench_*    the reference's own benchmark shape, benchScheduler (manager/scheduler/scheduler_test.go:3375-3465,
                sizes :3335-3373): tasks WITHOUT ServiceID / SpecVersion (one service "" for every task), nodes with an
                empty Engine description, every third node advertising the Network plugin "network", optionally every
                task attached to a network with that driver
"""
import hashlib

from swarmkit_amd import synth

READY, RUNNING, PENDING, ASSIGNED = 2, 512, 64, 192


def tick_digest(decisions):
    """(sha256 over the sorted decision lines, number of assigned tasks). A decision with VolumeAttachments carries them in its line."""
    def line(d):
        base = "%s|%s|%s|%d" % (d["ID"], d["NodeID"], d["Err"], d["State"])
        if d.get("Volumes"):
            base += "|" + ",".join("%s=%s@%s" % (v["ID"], v["Source"], v["Target"]) for v in d["Volumes"])
        return base
    lines = sorted(line(d) for d in decisions)
    h = hashlib.sha256("\n".join(lines).encode()).hexdigest()
    return h, sum(1 for d in decisions if d["NodeID"] and d["State"] >= ASSIGNED)


# ------------------------------------------------------------------------------------------------ cfg4
def run_cfg4(s, T=None, N=None, name="cfg4"):
    wl = synth.Workload(name, T=T, N=N)
    for i in range(wl.N):
        s.create_node(wl.node_doc(i))
    for k in range(wl.S):
        s.set_service(wl.service_id(k))
    for j in range(wl.T):
        s.create_task(wl.task_doc(j))
    h, placed = tick_digest(s.tick())
    return {"case": name, "T": wl.T, "N": wl.N, "seed": hex(wl.seed), "ticks": [h], "placed": [placed]}


# ------------------------------------------------------------------------------------------------ cfg5
def churn_drained(N, rnd):
    """10 % of the nodes, a different residue class every round (deterministic, no PRNG state to carry)."""
    return [i for i in range(N) if (i * 7 + rnd * 13) % 10 == 0]


def run_churn(s, T0=100_000, N=10_000, rounds=100, services=1000):
    total = T0 + rounds * (N // 10) * 12   # upper bound on the tasks ever created
    wl = synth.Workload("cfg3", T=total, N=N, services=services)
    for i in range(N):
        s.create_node(wl.node_doc(i))
    for k in range(wl.S):
        s.set_service(wl.service_id(k))
    for j in range(T0):
        s.create_task(wl.task_doc(j))
    placed = {}          # task index -> node index
    by_node = [[] for _ in range(N)]
    ticks, counts = [], []

    def do_tick():
        dec = s.tick()
        h, c = tick_digest(dec)
        ticks.append(h)
        counts.append(c)
        for d in dec:
            if d["NodeID"] and d["State"] >= ASSIGNED:
                j, n = int(d["ID"][1:]), int(d["NodeID"][1:])
                placed[j] = n
                by_node[n].append(j)

    do_tick()
    nxt, prev = T0, []
    for rnd in range(rounds):
        drained = churn_drained(N, rnd)
        for i in prev:
            s.update_node(wl.node_doc(i))
        for i in drained:
            doc = wl.node_doc(i)
            doc["Spec"] = dict(doc["Spec"], Availability=2)   # DRAIN
            s.update_node(doc)
        gone = []
        for i in drained:
            gone.extend(by_node[i])
            by_node[i] = []
        gone.sort()
        for j in gone:
            s.delete_task(dict(wl.task_doc(j), NodeID=wl.node_id(placed[j]), Status={"State": RUNNING}))
            del placed[j]
        for _ in range(len(gone)):
            s.create_task(wl.task_doc(nxt))
            nxt += 1
        do_tick()
        prev = drained
    return {"case": "cfg5_churn", "T0": T0, "N": N, "rounds": rounds, "services": services, "seed": hex(wl.seed), "created": nxt,
            "ticks": ticks, "placed": counts, "still_placed": len(placed)}


# ------------------------------------------------------------------------------------------------ benchScheduler
def ref_node(i):
    eng = {"Plugins": [{"Name": "network", "Type": "Network"}]} if i % 3 == 0 else {}
    return {"ID": "n%08d" % i, "Spec": {"Annotations": {"Name": "name%d" % i, "Labels": {}}}, "Status": {"State": READY},
            "Description": {"Engine": eng}}


def ref_task(i, net):
    t = {"ID": "task%d" % i, "DesiredState": RUNNING, "ServiceAnnotations": {"Name": "task%d" % i}, "Status": {"State": PENDING}}
    if net:
        t["Networks"] = [{"Network": {"DriverState": {"Name": "network"}}}]
    return t


def run_refbench(s, nodes, tasks, net):
    for i in range(nodes):
        s.create_node(ref_node(i))
    s.set_service("")   # the harness' stand-in for "the task's service lookup succeeds"; never consulted: every task is placed
    for i in range(tasks):
        s.create_task(ref_task(i, net))
    h, placed = tick_digest(s.tick())
    return {"case": "refbench", "nodes": nodes, "tasks": tasks, "net": bool(net), "ticks": [h], "placed": [placed]}


# ------------------------------------------------------------------------------------------------ task groups (SpecVersion set)
def run_grouped(s, name="cfg3", T=None, N=None, services=None):
    """SURVEY 8d's secondary mode: every task carries a SpecVersion, so a tick is S calls of scheduleTaskGroup with k = T / S."""
    wl = synth.Workload(name, T=T, N=N, services=services, grouped=True)
    for i in range(wl.N):
        s.create_node(wl.node_doc(i))
    for k in range(wl.S):
        s.set_service(wl.service_id(k))
    for j in range(wl.T):
        s.create_task(wl.task_doc(j))
    h, placed = tick_digest(s.tick())
    return {"case": "grouped_" + name, "T": wl.T, "N": wl.N, "services": wl.S, "seed": hex(wl.seed), "ticks": [h], "placed": [placed]}


def run_spread(s, N=6_000, groups=24, k=700, generic=False):
    """Three spread levels (node.labels.az / rack / engine.labels.slot: 16 x 9 x 8 values, ~1 100 leaves of ~5 nodes), a constraint, tight
    memory on a third of the services so that branches run out of room (noRoom, scheduler.go:810-813) and groups are left over; two
    ticks (the second one re-tries the leftovers against the residuals the first one left)."""
    GIB = 1 << 30
    for i in range(N):
        labels = {"az": "az%d" % (i % 16), "rack": "r%d" % ((i // 16) % 9), "tier": "a" if i % 4 else "b"}
        if i % 37 == 0:
            del labels["rack"]   # a missing label is the branch "" (nodeset.go:84-87)
        res = {"NanoCPUs": int((1 + i % 4) * 1e9), "MemoryBytes": (1 + (i * 5) % 7) * GIB}
        if generic:
            res["Generic"] = [{"Discrete": {"Kind": "gpu", "Value": i % 5}}] if i % 5 else []
        s.create_node({"ID": "n%08d" % i, "Spec": {"Annotations": {"Name": "node%d" % i, "Labels": labels}}, "Status": {"State": READY},
                       "Description": {"Hostname": "h%d" % i, "Resources": res, "Engine": {"Labels": {"slot": "s%d" % ((i // 144) % 8)}}}})
    prefs = [{"Spread": {"SpreadDescriptor": "node.labels.az"}}, {"Spread": {"SpreadDescriptor": "node.labels.rack"}},
             {"Spread": {"SpreadDescriptor": "engine.labels.slot"}}]
    j = 0
    for g in range(groups):
        sid = "svc%03d" % g
        s.set_service(sid)
        spec = {"Placement": {"Preferences": prefs[: 1 + g % 3] if g % 5 else prefs}}
        if g % 4 == 1:
            spec["Placement"]["Constraints"] = ["node.labels.tier==a"]
        if g % 3 == 0:
            spec["Resources"] = {"Reservations": {"MemoryBytes": (2 + g % 4) * GIB, "NanoCPUs": int(2e9)}}
        if g % 7 == 3:
            spec["Placement"]["MaxReplicas"] = 1 + g % 2
        if generic and g % 2 == 0:
            spec.setdefault("Resources", {}).setdefault("Reservations", {})["Generic"] = [{"Discrete": {"Kind": "gpu", "Value": 1 + g % 2}}]
        for _ in range(k + 17 * g):
            s.create_task({"ID": "t%08d" % j, "ServiceID": sid, "DesiredState": RUNNING, "Status": {"State": PENDING}, "SpecVersion": {"Index": 1}, "Spec": spec})
            j += 1
    ticks, counts = [], []
    for _ in range(2):
        h, c = tick_digest(s.tick())
        ticks.append(h)
        counts.append(c)
    return {"case": "spread3", "N": N, "tasks": j, "ticks": ticks, "placed": counts}


# ------------------------------------------------------------------------------------------------ CSI volumes
def run_volumes(s, N=20_000, T=40_000, services=400, grouped=False):
    """cfg4's cluster with CSI topologies (one plugin per node, zone + rack segments) and 600 volumes of every access mode in 40 groups; a
    quarter of the services mount a volume group, some a named volume too (read-only or not). Two ticks: after the first one every
    third placed task with mounts goes away, its volumes are free again, and as many new tasks of the same services arrive."""
    wl = synth.Workload("cfg4", T=T, N=N, services=services, grouped=grouped)
    for i in range(wl.N):
        d = wl.node_doc(i)
        seg = {"zone": "z%d" % wl.node_zone[i]}
        if i % 3:
            seg["rack"] = "r%d" % (i % 5)
        d["Description"]["CSIInfo"] = [{"PluginName": "csi-a" if i % 7 else "csi-b", "NodeID": "c%d" % i, "AccessibleTopology": {"Segments": seg}}] if i % 11 else []
        s.create_node(d)
    scopes, sharings = ["SINGLE_NODE", "MULTI_NODE"], ["NONE", "READ_ONLY", "ONE_WRITER", "ALL", "ALL", "ALL"]
    for v in range(600):
        acc = []
        if v % 4:
            acc.append({"Segments": {"zone": "z%d" % (v % 8)}})
        if v % 9 == 0:
            acc.append({"Segments": {"zone": "z%d" % ((v + 3) % 8), "rack": "r%d" % (v % 5)}})
        s.update_volume({"ID": "vol%04d" % v, "Spec": {"Annotations": {"Name": "name%04d" % v}, "Group": "g%02d" % (v % 40), "Driver": {"Name": "csi-a" if v % 5 else "csi-b"},
                                                       "AccessMode": {"Scope": scopes[(v // 3) % 2], "Sharing": sharings[v % 6]}, "Availability": "PAUSE" if v % 50 == 49 else "ACTIVE"},
                         "VolumeInfo": {"VolumeID": "plug%04d" % v, "AccessibleTopology": acc}})
    for k in range(wl.S):
        s.set_service(wl.service_id(k))

    def doc(j):
        t = wl.task_doc(j)
        k = wl.task_service(j)
        if k % 4 == 0:
            mounts = [{"Type": "CLUSTER", "Source": "group:g%02d" % (k % 40), "Target": "/data", "ReadOnly": k % 8 == 0}]
            if k % 12 == 0:
                mounts.append({"Type": "CLUSTER", "Source": "name%04d" % ((k * 7) % 600), "Target": "/named"})
            t.setdefault("Spec", {})["Container"] = {"Mounts": mounts}
        return t
    docs = {}
    for j in range(wl.T):
        docs[j] = doc(j)
        s.create_task(docs[j])
    ticks, counts = [], []
    dec = s.tick()
    h, c = tick_digest(dec)
    ticks.append(h)
    counts.append(c)
    gone = 0
    for d in dec:
        if d.get("Volumes") and d["NodeID"] and int(d["ID"][1:]) % 3 == 0:
            j = int(d["ID"][1:])
            s.delete_task(dict(docs[j], NodeID=d["NodeID"], Status={"State": ASSIGNED}, Volumes=d["Volumes"]))
            gone += 1
    total = synth.Workload("cfg4", T=T + gone, N=N, services=services, grouped=grouped)   # (same seed: the first T tasks are the same)
    wl.T = total.T
    for j in range(T, T + gone):
        s.create_task(doc(j))
    h, c = tick_digest(s.tick())
    ticks.append(h)
    counts.append(c)
    return {"case": "volumes", "T": T, "N": N, "services": services, "grouped": grouped, "released": gone, "ticks": ticks, "placed": counts}


CASES = {
    # CSI volumes: cfg4's cluster with topologies, 600 volumes in 40 groups, a quarter of the services with cluster mounts
    "volumes_mid": lambda s: run_volumes(s),
    "volumes_grouped_mid": lambda s: run_volumes(s, grouped=True),
    "volumes_small": lambda s: run_volumes(s, N=1_500, T=4_000, services=80),
    "volumes_grouped_small": lambda s: run_volumes(s, N=1_500, T=4_000, services=80, grouped=True),
    # task groups at BASELINE size: cfg3 as 1 000 groups of 100, cfg1 at its stated 1 000 x 10, ONE group of 20 000
    # tasks on 10 000 nodes (every heap takes its whole leaf; more than one task per node), three spread levels with > 512 leaves
    "grouped_cfg3_full": lambda s: run_grouped(s, "cfg3"),
    "grouped_cfg1_full": lambda s: run_grouped(s, "cfg1"),
    "grouped_one_20k": lambda s: run_grouped(s, "cfg3", T=20_000, N=10_000, services=1),
    "grouped_cfg4_mid": lambda s: run_grouped(s, "cfg4", T=60_000, N=20_000, services=300),
    "grouped_spread3": lambda s: run_spread(s),
    "grouped_spread3_generic": lambda s: run_spread(s, N=3_000, groups=12, k=800, generic=True),
    "grouped_small": lambda s: run_grouped(s, "cfg3", T=3_000, N=400),
    "cfg4_full": lambda s: run_cfg4(s),
    "cfg4_mid": lambda s: run_cfg4(s, T=200_000, N=40_000),
    # cfg3 with (almost) every service its own reservation pair: 1 000 / 1 750 distinct cpu and 1 000 / 2 000 distinct memory values per batch
    "cfg3m_full": lambda s: run_cfg4(s, T=100_000, N=10_000, name="cfg3m"),
    "cfg3m_mid": lambda s: run_cfg4(s, T=200_000, N=40_000, name="cfg3m"),
    "cfg3m_small": lambda s: run_cfg4(s, T=6_000, N=700, name="cfg3m"),
    "cfg5_churn": lambda s: run_churn(s),
    "cfg5_churn_small": lambda s: run_churn(s, T0=5_000, N=500, rounds=12, services=50),
    "cfg5_churn_mid": lambda s: run_churn(s, T0=20_000, N=2_000, rounds=100, services=200),   # all 100 rounds, a fifth of the cluster
    # cfg3's reservations fill the cluster at 100k x 10k (9.6 % of the first tick is unplaceable already), so draining 10 % of the nodes
    # leaves a backlog that every later tick re-evaluates and after ~10 rounds nothing moves any more: the oracle needs many hours
    # for that, and the rounds test little. The same script at 60 % load keeps re-placing ~10 % of the tasks in every round:
    "cfg5_churn_60k": lambda s: run_churn(s, T0=60_000, N=10_000, rounds=100, services=600),
    "cfg5_churn_12k": lambda s: run_churn(s, T0=12_000, N=2_000, rounds=100, services=120),
    "refbench_1k_100k": lambda s: run_refbench(s, 1_000, 100_000, False),
    "refbench_net_5k_100k": lambda s: run_refbench(s, 5_000, 100_000, True),
    "refbench_100k_100k": lambda s: run_refbench(s, 100_000, 100_000, False),
    "refbench_small": lambda s: run_refbench(s, 100, 3_000, True),
    # the reference's largest benchmark shape, BenchmarkScheduler100kNodes1MTasks (scheduler_test.go:3355-3357): 1M tasks of ONE service
    # on 100k nodes — ten tasks a node, the water-filling path (k_waterfill) at its extreme
    "refbench_100k_1m": lambda s: run_refbench(s, 100_000, 1_000_000, False),
}
