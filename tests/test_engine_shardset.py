"""GPU parity of the shard SET (include/swp.h swp_shardset_create, csrc/swp_shardset.hpp): G engines of this process behind ONE engine
handle — the node-range split as a drop-in for everything above the C ABI. The set owns the global node index space,
routes every node call to the owner of the range and runs a batch with swp_shard_run; the host layer (swp::Scheduler) and the event
scripts of tests/bigcases.py do not know that they talk to more than one engine.

  * one-off batches against the oracle, through the struct ABI and through the host layer;
  * BASELINE configs[4], the reschedule churn — drain 10 % of the nodes, NodeInfo.removeTask for the tasks on them (scheduler.go:350-366,
    nodeinfo.go:66-104), re-place, round after round — over 2 / 3 / 8 shards against the digests the oracle produced offline for the
    single sequential scheduler (tests/golden/big_cfg5_churn_*.json): the INCREMENTAL path across shards;
  * node removal and index recycling across range borders, preassigned tasks (taskFitNode on the owner), the constraint enforcer and
    NodeMatches sweeps split by owner, a set that is full."""
import json
import os

import numpy as np
import pytest

import bigcases
import orc
import parity_util as pu
from swarmkit_amd import abi, synth
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cap(n, g):
    return (n + g - 1) // g


@pytest.mark.parametrize("shards", [2, 3, 4])
@pytest.mark.parametrize("name,T,N,kw", [("cfg3", 2500, 300, {}), ("cfg4", 3000, 700, {}), ("cfg2", 3000, 50, {}), ("cfg3m", 3000, 500, {"services": 900})])
def test_set_agrees_with_oracle(shards, name, T, N, kw):
    wl = synth.Workload(name, T=T, N=N, **kw)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, s, out, hist = pu.engine_run(wl, shards=shards, nodes_per_shard=_cap(N, shards))
    pu.assert_same(op, oe, ep, ee)
    st = s.e.stats()
    assert st["last_resolver"] == 7 and st["n_nodes"] == N


def test_set_with_spare_slots_and_an_empty_range():
    """Ranges larger than the node set: the last shards hold few nodes or none (they are left out of the rounds)."""
    wl = synth.Workload("cfg3", T=1500, N=250)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, s, *_ = pu.engine_run(wl, shards=4, nodes_per_shard=100)   # 100 + 100 + 50 + 0
    pu.assert_same(op, oe, ep, ee)
    ep, ee, s, *_ = pu.engine_run(wl, shards=3, nodes_per_shard=1000)  # everything on shard 0: that engine's own batch path
    pu.assert_same(op, oe, ep, ee)
    assert s.e.stats()["last_resolver"] != 7


def test_a_full_set_refuses_the_next_node():
    s = swhost.HostScheduler(shards=2, nodes_per_shard=3)
    wl = synth.Workload("cfg2", T=10, N=7)
    for i in range(6):
        s.create_node(wl.node_doc(i))
    with pytest.raises(abi.SwpError) as ei:
        s.create_node(wl.node_doc(6))
    assert ei.value.code == abi.SWP_ERANGE and "full" in str(ei.value)
    s.delete_node(wl.node_id(4))             # a slot of shard 1 is free again: the next new node takes it (lowest free index first)
    s.create_node(wl.node_doc(6))
    assert s.node_index(wl.node_id(6)) == 4


@pytest.mark.parametrize("case,shards", [("cfg5_churn_small", 2), ("cfg5_churn_small", 3), ("cfg5_churn_small", 8), ("cfg5_churn_mid", 3), ("cfg5_churn_mid", 8), ("cfg5_churn_12k", 2)])
def test_churn_over_shards_matches_oracle_digests(case, shards):
    """BASELINE configs[4] over node-range shards: every round the drained nodes' rows change on their owners, NodeInfo.removeTask runs
    on the owner of every deleted task's node, and the re-placement batch is a sharded batch — every tick's digest must be the one the
    oracle's single sequential scheduler produced."""
    want = json.load(open(os.path.join(GOLD, "big_%s.json" % case)))
    sched = swhost.HostScheduler(shards=shards, nodes_per_shard=_cap(want["N"], shards))
    got = bigcases.CASES[case](sched)
    assert sched.e.stats()["last_resolver"] == 7
    assert got["placed"] == want["placed"]
    bad = [i for i, (a, b) in enumerate(zip(got["ticks"], want["ticks"])) if a != b]
    assert not bad, "tick digests differ at ticks %s" % bad[:10]
    for k in ("created", "still_placed", "rounds"):
        assert got[k] == want[k], k


def _script(s, wl, rounds=6):
    """Nodes leave and come back, tasks are deleted, new ones arrive: the same events into the oracle and into a shard set."""
    for i in range(wl.N):
        s.create_node(wl.node_doc(i))
    for k in range(wl.S):
        s.set_service(wl.service_id(k))
    nxt = 0
    placed = {}
    digests = []
    for rnd in range(rounds):
        for _ in range(wl.T // rounds):
            s.create_task(wl.task_doc(nxt))
            nxt += 1
        dec = s.tick()
        digests.append(bigcases.tick_digest(dec))
        for d in dec:
            if d["NodeID"]:
                placed[d["ID"]] = d["NodeID"]
        # every 9th node (another residue each round) leaves the cluster with its tasks; the nodes that left two rounds ago come back
        gone = [i for i in range(wl.N) if (i + rnd) % 9 == 0]
        for tid, nid in sorted(placed.items()):
            if int(nid[1:]) in gone:
                j = int(tid[1:])
                s.delete_task(dict(wl.task_doc(j), NodeID=nid, Status={"State": 512}))
                del placed[tid]
        for i in gone:
            s.delete_node(wl.node_id(i))
        if rnd >= 1:
            for i in range(wl.N):
                if (i + rnd - 1) % 9 == 0 and (i + rnd) % 9 != 0:
                    s.create_node(wl.node_doc(i))   # takes the lowest free index: usually a slot on ANOTHER shard than before
    return digests


@pytest.mark.parametrize("shards", [2, 5])
def test_nodes_leave_and_return_across_range_borders(shards):
    wl = synth.Workload("cfg4", T=3000, N=330)
    want = _script(orc.Oracle(), wl)
    got = _script(swhost.HostScheduler(shards=shards, nodes_per_shard=_cap(wl.N, shards) + 3), wl)
    assert got == want


def test_preassigned_tasks_and_node_info_on_the_owner():
    wl = synth.Workload("cfg3", T=400, N=60)
    o, s = orc.Oracle(), swhost.HostScheduler(shards=3, nodes_per_shard=20)
    for x in (o, s):
        for i in range(wl.N):
            x.create_node(wl.node_doc(i))
        for k in range(wl.S):
            x.set_service(wl.service_id(k))
        for j in range(100):
            x.create_task(dict(wl.task_doc(j), NodeID=wl.node_id((j * 7) % wl.N)))   # preassigned: taskFitNode on its node's owner
    a, b = o.process_preassigned(), s.process_preassigned()
    key = lambda d: d["ID"]
    assert [(d["ID"], d["NodeID"], d["State"], d["Err"]) for d in sorted(a, key=key)] == [(d["ID"], d["NodeID"], d["State"], d["Err"]) for d in sorted(b, key=key)]
    for x in (o, s):
        for j in range(100, 400):
            x.create_task(wl.task_doc(j))
    assert bigcases.tick_digest(o.tick()) == bigcases.tick_digest(s.tick())
    for i in (0, 19, 20, 41, 59):   # both sides of every range border
        ia, ib = o.node_info(wl.node_id(i)), s.node_info(wl.node_id(i))
        assert ia["ActiveTasksCount"] == ib["ActiveTasksCount"] and ia["AvailableResources"]["NanoCPUs"] == ib["AvailableResources"]["NanoCPUs"]
        assert ia["ActiveTasksCountByService"] == ib["ActiveTasksCountByService"]


def test_node_matches_and_enforce_are_split_by_owner():
    wl = synth.Workload("cfg3", T=800, N=150)
    one, many = swhost.HostScheduler(), swhost.HostScheduler(shards=4, nodes_per_shard=40)
    sets = []
    for s in (one, many):
        for i in range(wl.N):
            s.create_node(wl.node_doc(i))
        cs = [s.constraint_set(wl.service_spec(k).get("Spec", {}).get("Placement", {}).get("Constraints", [])) for k in range(min(wl.S, 8))]
        sets.append(s.e.node_matches(np.array(cs, dtype=np.uint32)))
    a, b = sets
    assert a.shape[0] == b.shape[0]
    for r in range(a.shape[0]):
        bits_a = [(int(a[r, i >> 6]) >> (i & 63)) & 1 for i in range(wl.N)]
        bits_b = [(int(b[r, i >> 6]) >> (i & 63)) & 1 for i in range(wl.N)]
        assert bits_a == bits_b
    # the enforcer's sweep: the same cluster state on both, the same verdicts
    for s in (one, many):
        for k in range(wl.S):
            s.set_service(wl.service_id(k))
        for j in range(wl.T):
            s.create_task(wl.task_doc(j))
    da, db = one.tick(), many.tick()
    assert bigcases.tick_digest(da) == bigcases.tick_digest(db)
    by_node = {}
    for d in da:
        if d["NodeID"]:
            j = int(d["ID"][1:])
            by_node.setdefault(d["NodeID"], []).append(dict(wl.task_doc(j), NodeID=d["NodeID"], Status={"State": 512}))
    docs = []
    for i in range(wl.N):
        doc = wl.node_doc(i)
        if i % 3 == 0:   # the node shrank: reservations no longer fit
            doc["Description"]["Resources"] = {"NanoCPUs": int(1e9), "MemoryBytes": 1 << 30}
        if i % 5 == 0:
            doc["Spec"]["Annotations"]["Labels"] = {}
        docs.append(doc)
    ra = swhost.enforce(one, docs, by_node)
    rb = swhost.enforce(many, docs, by_node)
    assert ra == rb and any(ra.values())


@pytest.mark.parametrize("case,N,shards", [("volumes_small", 1_500, 2), ("volumes_small", 1_500, 5), ("volumes_mid", 20_000, 4)])
def test_csi_volumes_over_shards_match_oracle_digests(case, N, shards):
    """CSI volumes through the node-range shards (VERDICT r4 missing #3; volumes.go:223-316): every shard holds the volume table, the owner
    of a placed task's node chooses and reserves its volumes (chooseTaskVolumes + reserveTaskVolumes on the device), the other shards
    learn the reservation from the trailer behind its next proposals — a usage pinned to a node of another range. The script of
    tests/bigcases.py (tasks with mounts placed, a third of them deleted, new ones placed against what is left) against the digests
    the oracle produced offline for one sequential scheduler."""
    want = json.load(open(os.path.join(GOLD, "big_%s.json" % case)))
    os.environ["SWP_HOST"] = "cxx"
    sched = swhost.HostScheduler(shards=shards, nodes_per_shard=_cap(N, shards))
    got = bigcases.CASES[case](sched)
    assert sched.e.stats()["last_resolver"] == 7
    assert got["placed"] == want["placed"] and got["released"] == want["released"]
    assert got["ticks"] == want["ticks"]


@pytest.mark.parametrize("case,N,shards", [("grouped_small", 400, 3), ("grouped_cfg3_full", 10_000, 4), ("grouped_cfg3_full", 10_000, 8), ("grouped_spread3", 6_000, 4),
                                           ("volumes_grouped_small", 1_500, 4), ("grouped_cfg1_full", 10, 2)])
def test_task_groups_over_shards_match_oracle_digests(case, N, shards):
    """A replicated service placed on a sharded node set (VERDICT r4 missing #2; nodeset.go:107-120, decision_tree.go:24-52): cfg3 as
    1 000 groups of 100 at 100k x 10k over 4 and 8 engines, three spread levels over 1 100 leaves, groups with cluster mounts — every
    tick's digest must be the one the oracle's single sequential scheduler produced."""
    want = json.load(open(os.path.join(GOLD, "big_%s.json" % case)))
    os.environ["SWP_HOST"] = "cxx"
    sched = swhost.HostScheduler(shards=shards, nodes_per_shard=_cap(N, shards))
    got = bigcases.CASES[case](sched)
    assert got["placed"] == want["placed"]
    assert got["ticks"] == want["ticks"]


def test_every_replica_of_the_volume_table_ends_a_batch_with_the_same_usage():
    """The struct ABI without a host layer in between: two batches of tasks with cluster mounts on a shard set, no swp_volume_set_usage
    between them — "swp_batch_fetch leaves the numbers as the batch made them" (include/swp.h) must hold on EVERY shard, the ones that
    did not place the last task with mounts included (its reservation reaches them behind the last round: k_r7_settle). Against one engine
    given the same calls."""
    os.environ["SWP_HOST"] = "cxx"
    wl = synth.Workload("cfg4", T=602, N=90, services=12)
    outs = []
    for kw in ({}, {"shards": 3, "nodes_per_shard": 30}):
        s = swhost.HostScheduler(**kw)
        for i in range(wl.N):
            d = wl.node_doc(i)
            seg = {"zone": "z%d" % wl.node_zone[i]}
            d["Description"]["CSIInfo"] = [{"PluginName": "csi-a", "NodeID": "c%d" % i, "AccessibleTopology": {"Segments": seg}}] if i % 5 else []
            s.create_node(d)
        for v in range(240):   # plenty of unshared volumes: every task with a mount takes the next unused one of its group
            acc = [{"Segments": {"zone": "z%d" % (v % 8)}}] if v % 3 == 0 else []
            s.update_volume({"ID": "vol%03d" % v,
                             "Spec": {"Annotations": {"Name": "name%03d" % v}, "Group": "g%d" % (v % 4), "Driver": {"Name": "csi-a"},
                                      "AccessMode": {"Scope": "SINGLE_NODE", "Sharing": "NONE"}, "Availability": "ACTIVE"},
                             "VolumeInfo": {"VolumeID": "p%03d" % v, "AccessibleTopology": acc}})
        for k in range(wl.S):
            s.set_service(wl.service_id(k))

        def descs(j0, n):
            out = []
            for j in range(j0, j0 + n):
                t = wl.task_doc(j)
                if j % 3 == 0:
                    t.setdefault("Spec", {})["Container"] = {"Mounts": [{"Type": "CLUSTER", "Source": "group:g%d" % (j % 4), "Target": "/d"}]}
                out.append(s.task_desc(t))
            return np.concatenate(out)
        res = []
        for j0 in (0, 301):   # (task 300 has a mount: the first batch ENDS on a reservation)
            b = s.e.batch_prepare(descs(j0, 301))
            b.run()
            out, hist = b.fetch()
            res.append((out.tolist(), hist.tolist(), b.attachments().tolist(), [s.e.volume_get_usage(s.e.intern(9, "vol%03d" % v)) for v in range(240)]))
            b.free()
        outs.append(res)
    assert outs[0] == outs[1]
    assert sum(1 for x in outs[0][1][0] if x >= 0) > 0 and any(a[0] != 0xFFFFFFFF for a in outs[0][1][2])
