"""CPU: the volume bookkeeping of the C++ host layer (csrc/swp_sched.cpp: reserveVolume / releaseVolume / reserveTaskVolumes /
freeVolumes — the half of volumes.go the engine does not hold) against the oracle, event by event. The engine under the host layer is
the scripted double (tests/fake_swp.cpp): no task with cluster mounts is ever SCHEDULED here — that is the GPU suite's
tests/test_engine_volumes.py — the events are the ones that move the reference counts: tasks of the store at start, tasks going away,
volumes being updated."""
import copy
import random

import pytest

import fakelib
import kat_volumes as kv
import orc
from swarmkit_amd import abi, sched as swsched


class Both:
    def __init__(self):
        self.o = orc.Oracle()
        self.e = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
        self.vols = []

    def __getattr__(self, name):
        def call(*a):
            ro, re = getattr(self.o, name)(*copy.deepcopy(a)), getattr(self.e, name)(*copy.deepcopy(a))
            if name == "update_volume" and a[0]["ID"] not in self.vols:
                self.vols.append(a[0]["ID"])
            return ro, re
        return call

    def check(self):
        for vid in self.vols:
            io, ie = self.o.volume_info(vid), self.e.volume_info(vid)
            assert (io is None) == (ie is None), vid
            if io is not None:
                assert io["Tasks"] == ie["Tasks"] and {k: c for k, c in io["Nodes"].items() if c} == {k: c for k, c in ie["Nodes"].items() if c}, (vid, io, ie)

    def free(self):
        fo, fe = self.o.free_volumes(), self.e.free_volumes()
        assert fo == fe, (fo, fe)
        return fo


def _cluster():
    b = Both()
    nodes, volumes, all_volume, tasks = kv.free_volumes_fixture()
    for n in nodes:
        b.create_node(dict(n, Status={"State": orc.READY}))
    for v in volumes + [all_volume]:
        b.update_volume(v)
    b.set_service("svc")
    for t in tasks:
        b.setup_task(t)
    b.check()
    return b, nodes, volumes, all_volume, tasks


def test_reference_counts_of_the_store_at_start():
    """volumes_test.go:644-661 through the host layer."""
    b, nodes, volumes, all_volume, tasks = _cluster()
    for i, v in enumerate(volumes):
        assert b.e.volume_info(v["ID"])["Nodes"] == {nodes[i]["ID"]: 1}
    assert b.e.volume_info(all_volume["ID"])["Nodes"] == {n["ID"]: 1 for n in nodes}
    assert b.free() == []


def test_free_volumes_that_are_no_longer_needed():
    """volumes_test.go:663-697 through the host layer."""
    b, nodes, volumes, all_volume, tasks = _cluster()
    b.delete_task(tasks[0])
    b.check()
    assert b.free() == [{"VolumeID": volumes[0]["ID"], "NodeIDs": ["node0"]}, {"VolumeID": all_volume["ID"], "NodeIDs": ["node0"]}]
    assert b.free() == []
    # the store's event brings the same statuses back: nothing changes; a fresh PUBLISHED status on a node without users is freed again
    v0 = copy.deepcopy(volumes[0])
    v0["PublishStatus"] = [{"NodeID": "node0", "State": "PENDING_NODE_UNPUBLISH"}, {"NodeID": "node3", "State": "PUBLISHED"}]
    b.update_volume(v0)
    assert b.free() == [{"VolumeID": volumes[0]["ID"], "NodeIDs": ["node3"]}]


def test_a_volume_reserved_twice_by_one_task_keeps_a_reference():
    """volumes.go:156-160 / :169-178: two mounts of one task on one volume count the node twice and give it back once."""
    b = Both()
    v = kv.canned_volume(1)
    v["PublishStatus"] = [{"NodeID": "n", "State": "PUBLISHED"}]
    b.update_volume(v)
    b.create_node({"ID": "n", "Status": {"State": orc.READY}, "Description": {}})
    b.set_service("svc")
    t = {"ID": "t", "ServiceID": "svc", "NodeID": "n", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
         "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume1", "/a"), kv.cluster_mount("volume1", "/b")]}},
         "Volumes": [{"ID": v["ID"], "Source": "volume1", "Target": "/a"}, {"ID": v["ID"], "Source": "volume1", "Target": "/b"}]}
    b.setup_task(t)
    b.check()
    assert b.e.volume_info(v["ID"])["Nodes"] == {"n": 2}
    b.delete_task(t)
    b.check()
    assert b.e.volume_info(v["ID"])["Nodes"] == {"n": 1} and b.free() == []


STATES = ["PENDING_PUBLISH", "PUBLISHED", "PENDING_NODE_UNPUBLISH", "PENDING_UNPUBLISH"]


@pytest.mark.parametrize("seed", range(12))
def test_seeded_event_scripts(seed):
    """Volumes with random publish statuses, tasks of the store at start holding random attachments (named and group mounts, read-only
    or not, sometimes the same volume twice, sometimes a volume the set does not know), tasks deleted or failing, volumes updated —
    the volumeSet's view of every volume and every freeVolumes batch must equal the oracle's."""
    rng = random.Random(0xF4EE + seed)
    b = Both()
    n_nodes, n_vols = rng.randrange(2, 7), rng.randrange(1, 7)
    nodes = ["n%d" % i for i in range(n_nodes)]
    for n in nodes:
        b.create_node({"ID": n, "Status": {"State": orc.READY}, "Description": {}})
    b.set_service("svc")

    def volume(i):
        v = kv.canned_volume(i, group=rng.choice(["g1", "g2", ""]))
        v["Spec"]["AccessMode"] = {"Scope": rng.choice([kv.SINGLE, kv.MULTI]), "Sharing": rng.choice([kv.ALL, "ONE_WRITER", "READ_ONLY", "NONE"])}
        v["PublishStatus"] = [{"NodeID": n, "State": rng.choice(STATES)} for n in rng.sample(nodes, rng.randrange(0, n_nodes + 1))]
        if rng.random() < 0.1:
            del v["VolumeInfo"]   # not created by the plugin yet: the scheduler ignores it
        return v

    for i in range(n_vols):
        b.update_volume(volume(i))
    live = {}
    next_task = freed = 0
    for step in range(60):
        r = rng.random()
        if r < 0.45 or not live:
            tid = "t%d" % next_task
            next_task += 1
            mounts, atts = [], []
            for k in range(rng.randrange(1, 4)):
                vi = rng.randrange(0, n_vols + 1)   # (n_vols: a volume the set never saw)
                src = rng.choice(["volume%d" % vi, "group:g1"])
                m = kv.cluster_mount(src, "/m%d" % k, rng.random() < 0.4)
                mounts.append(m)
                if rng.random() < 0.85:
                    atts.append({"ID": "volumeID%d" % vi, "Source": src, "Target": m["Target"]})
            if rng.random() < 0.2:
                mounts.append({"Type": "BIND", "Source": "/x", "Target": "/y"})
            t = {"ID": tid, "ServiceID": "svc", "NodeID": rng.choice(nodes), "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
                 "Spec": {"Container": {"Mounts": mounts}}, "Volumes": atts}
            b.setup_task(t)
            live[tid] = t
        elif r < 0.65:
            tid = rng.choice(sorted(live))
            b.delete_task(live.pop(tid))
        elif r < 0.75:   # the task fails: updateTask deletes the old task, its reservations with it (scheduler.go:286-316)
            tid = rng.choice(sorted(live))
            b.update_task(dict(live.pop(tid), Status={"State": rng.choice([orc.FAILED, orc.SHUTDOWN, orc.COMPLETE])}))
        elif r < 0.9:
            b.update_volume(volume(rng.randrange(0, n_vols)))
        else:
            freed += len(b.free())
        b.check()
    freed += len(b.free())
    for tid in sorted(live):
        b.delete_task(live[tid])
    b.check()
    freed += len(b.free())
    assert freed > 0   # (every script has volumes whose publications outlive their users)


# ---------------------------------------------------------------------------- placements with attachments, over the scripted double
def _usage_agrees(e, vids):
    """The engine's usage numbers of a volume — what the double counted for the placements it "made" plus what the host layer pushed —
    against the host layer's own maps: swp_volume_set_usage with the host's numbers must be idempotent behind a device call."""
    for vid in vids:
        info = e.volume_info(vid)
        if info is None:
            continue
        assert info["Engine"]["Tasks"] == len(info["Tasks"]), (vid, info)
        assert info["Engine"]["Writers"] == sum(1 for u in info["Tasks"].values() if not u["ReadOnly"]), (vid, info)
        counts = {}
        for u in info["Tasks"].values():
            counts[u["NodeID"]] = counts.get(u["NodeID"], 0) + 1
        assert {k: c for k, c in info["Nodes"].items() if c} == counts, (vid, info)   # (no task here mounts one volume twice)


@pytest.mark.parametrize("seed", range(10))
def test_placements_with_attachments_keep_the_books(seed):
    """One-off tasks and task groups with cluster mounts through tick(), preassigned ones through process_preassigned(), over the scripted
    engine double (its "choices" are pseudo-random volumes a mount could name): every decision's attachments are booked under the task on
    its node, a rejected decision and a deleted task give them back, the engine's usage numbers follow, and freeVolumes frees exactly
    the publications nobody uses."""
    import scenarios as sc
    rng = random.Random(0xA77A + seed)
    e = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
    nodes = ["n%d" % i for i in range(rng.randrange(2, 6))]
    for n in nodes:
        e.create_node({"ID": n, "Status": {"State": orc.READY}, "Description": {"CSIInfo": [{"PluginName": "driver", "NodeID": "csi-" + n}]}})
    vids = []
    for i in range(rng.randrange(2, 7)):
        v = kv.canned_volume(i, group=rng.choice(["g1", "g2"]))
        v["PublishStatus"] = [{"NodeID": n, "State": "PUBLISHED"} for n in nodes]
        e.update_volume(v)
        vids.append(v["ID"])
    for s in range(3):
        e.set_service("svc%d" % s, spec_version=7 if s == 2 else None)
    held = {}   # task id -> (node, [volume ids]) as decided
    tid = 0
    group_mounts = None
    for round_ in range(8):
        new = []
        for _ in range(rng.randrange(1, 6)):
            # (no task here ends up with one volume on two of its mounts — that case has its own test above: either ONE group mount,
            # or distinct named volumes, now and then one that does not exist)
            if rng.random() < 0.4:
                sources = [rng.choice(["group:g1", "group:g2"])]
            else:
                sources = ["volume%d" % i for i in rng.sample(range(len(vids)), rng.randrange(1, min(3, len(vids)) + 1))] + (["volume99"] if rng.random() < 0.15 else [])
            mounts = [kv.cluster_mount(src, "/m%d" % k, rng.random() < 0.4) for k, src in enumerate(sources)]
            svc = rng.randrange(0, 3)
            if svc == 2:   # the tasks of one (service, SpecVersion) share their spec: that is what makes them a group (scheduler.go:442-459)
                group_mounts = group_mounts or mounts
                mounts = group_mounts
            t = sc.pending("t%03d" % tid, "svc%d" % svc, spec_version=7 if svc == 2 else None, Spec={"Container": {"Mounts": mounts}})
            if rng.random() < 0.2:   # a preassigned task: decided by process_preassigned on its node
                t["NodeID"] = rng.choice(nodes)
            tid += 1
            new.append(t)
            e.create_task(t)
        e.process_preassigned()   # taskFitNode chooses volumes for the decision and reserves NOTHING (scheduler.go:663-677): not tracked here
        decisions = e.tick()
        for d in decisions:
            if d.get("Deferred"):
                continue
            vols = [v["ID"] for v in d.get("Volumes") or []]
            if d["NodeID"] and d["State"] == orc.ASSIGNED:
                held[d["ID"]] = (d["NodeID"], vols)
        for t_id, (nid, vols) in held.items():
            for v in vols:
                info = e.volume_info(v)
                assert info is not None and info["Tasks"].get(t_id, {}).get("NodeID") == nid, (t_id, nid, v, info)
        _usage_agrees(e, vids)
        # some of them go away again
        for t_id in sorted(held):
            r = rng.random()
            if r < 0.15:
                e.delete_task({"ID": t_id, "NodeID": held[t_id][0], "ServiceID": "svc0", "Volumes": [{"ID": v, "Source": "", "Target": ""} for v in held[t_id][1]]})
            elif r < 0.25 and any(d["ID"] == t_id for d in decisions):
                e.reject_decision(t_id)
            else:
                continue
            for v in held.pop(t_id)[1]:
                info = e.volume_info(v)
                assert t_id not in info["Tasks"], (t_id, v, info)
        _usage_agrees(e, vids)
        for upd in e.free_volumes():
            info = e.volume_info(upd["VolumeID"])
            for n in upd["NodeIDs"]:
                assert not info["Nodes"].get(n), (upd, info)


def test_the_commit_plan_names_the_publications_and_the_decisions_to_call_off():
    """applySchedulingDecisions' volume side (scheduler.go:548-610) as the commit plan reports it: an attachment on a volume that has no
    PublishStatus for the task's node wants a PENDING_PUBLISH one; a decision with an attachment on a volume that was paused or drained
    since is called off and publishes nothing."""
    e = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
    import scenarios as sc
    for n in ("n0", "n1"):
        e.create_node({"ID": n, "Status": {"State": orc.READY}, "Description": {"CSIInfo": [{"PluginName": "driver", "NodeID": "csi-" + n}]}})
    v0, v1 = kv.canned_volume(0), kv.canned_volume(1)
    v0["PublishStatus"] = [{"NodeID": "n0", "State": "PUBLISHED"}, {"NodeID": "n1", "State": "PUBLISHED"}]
    e.update_volume(v0)
    e.update_volume(v1)
    e.set_service("svc")
    placed = {}
    for i in range(12):   # (the double leaves a task without a node or without attachments now and then)
        e.create_task(sc.pending("t%02d" % i, "svc", Spec={"Container": {"Mounts": [kv.cluster_mount("volume0", "/a"), kv.cluster_mount("volume1", "/b")]}}))
    for d in e.tick():
        if d["NodeID"] and d.get("Volumes"):
            placed[d["ID"]] = d["NodeID"]
    assert placed
    plan = e.commit_plan()
    assert plan["VolumeFailed"] == []
    assert len(plan["Publish"]) == 1 and plan["Publish"][0]["VolumeID"] == "volumeID1"   # (volume0 is published on both nodes already)
    assert sorted(plan["Publish"][0]["NodeIDs"]) == sorted(set(placed.values()))
    # volume1 is drained before the commit: every decision that uses it is called off, nothing is published
    v1d = copy.deepcopy(v1)
    v1d["Spec"]["Availability"] = "DRAIN"
    e.update_volume(v1d)
    plan = e.commit_plan()
    assert sorted(plan["VolumeFailed"]) == sorted(placed) and plan["Publish"] == []
    for tid in plan["VolumeFailed"]:
        assert e.reject_decision(tid)
    assert e.volume_info("volumeID1")["Tasks"] == {} and e.volume_info("volumeID0")["Tasks"] == {}


# ---------------------------------------------------------------------------- REPLAY: the oracle's decisions through the host layer
@pytest.mark.parametrize("seed", range(10))
def test_the_host_layer_replays_the_oracles_ticks(seed):
    """End to end on CPU, without an engine: a seeded cluster with CSI volumes (topologies, access modes, groups, paused volumes), one-off
    tasks with and without cluster mounts over several ticks, a third of the placed tasks going away after every tick. The ORACLE decides
    each tick; its decisions — node and volumes per task — are scripted into the engine double (fakelib.script), the C++ host layer runs
    the same tick above it and must report the same assignments and attachments, book the same users under every volume and free the
    same publications. (What the double cannot replay: WHY a task found no node — those decisions are compared by node only.)"""
    import scenarios as sc
    rng = random.Random(0x7E91A + seed)
    o = orc.Oracle()
    e = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
    zones = ["z1", "z2", "z3"]
    n_nodes = rng.choice([3, 8, 30])
    for i in range(n_nodes):
        csi = []
        for plug in ("p1", "p2"):
            if rng.random() < 0.8:
                c = {"PluginName": plug}
                if rng.random() < 0.8:
                    c["AccessibleTopology"] = {"Segments": {"zone": rng.choice(zones)}}
                csi.append(c)
        n = {"ID": "n%04d" % i, "Status": {"State": orc.READY}, "Description": {"Resources": {"NanoCPUs": 8 * 10**9, "MemoryBytes": 16 << 30}, "CSIInfo": csi}}
        o.create_node(copy.deepcopy(n))
        e.create_node(copy.deepcopy(n))
    n_vol = rng.choice([2, 6, 14])
    vids = []
    for v in range(n_vol):
        acc = [{"Segments": {"zone": rng.choice(zones)}} for _ in range(rng.choice([0, 1, 1, 2]))]
        vol = {"ID": "vol%02d" % v, "Spec": {"Annotations": {"Name": "name%02d" % v}, "Group": rng.choice(["", "g1", "g2"]), "Driver": {"Name": rng.choice(["p1", "p2"])},
                                           "AccessMode": {"Scope": rng.choice([kv.SINGLE, kv.MULTI]), "Sharing": rng.choice([kv.NONE, kv.READ_ONLY, kv.ONE_WRITER, kv.ALL])},
                                           "Availability": rng.choice(["ACTIVE", "ACTIVE", "ACTIVE", "PAUSE"])},
               "VolumeInfo": {"VolumeID": "csi%02d" % v, "AccessibleTopology": acc},
               "PublishStatus": [{"NodeID": "n%04d" % k, "State": "PUBLISHED"} for k in range(0, n_nodes, 2)]}
        o.update_volume(copy.deepcopy(vol))
        e.update_volume(copy.deepcopy(vol))
        vids.append(vol["ID"])
    for s in range(4):   # svc3's tasks carry a SpecVersion: one task group per tick (scheduler.go:442-459), all with the same spec
        o.set_service("svc%d" % s, spec_version=3 if s == 3 else None)
        e.set_service("svc%d" % s, spec_version=3 if s == 3 else None)
    group_mounts = None
    placed, tid, docs, waiting = [], 0, {}, []
    n_placed = n_attached = 0
    for tick in range(5):
        for _ in range(rng.choice([5, 20, 60])):
            tid += 1
            mounts = []
            if rng.random() < 0.6:
                # a group mount or two, named volumes (one of them may not exist), the same volume or group for several mounts: a volume that
                # serves m of a task's mounts keeps m - 1 counts on the node for ever in the reference (chooseTaskVolumes reserves per call and
                # releases once per volume, volumes.go:104-131,162-178) — whether the choice succeeds or stops at a later mount
                r = rng.random()
                if r < 0.3:
                    sources = ["group:" + rng.choice(["", "g1", "g2", "g9"])]
                elif r < 0.45:
                    g = "group:" + rng.choice(["", "g1", "g2"])
                    sources = [g, g] + (["name%02d" % n_vol] if rng.random() < 0.3 else [])   # (the last one does not exist: the choice fails behind a doubled volume)
                elif r < 0.6:
                    nm = "name%02d" % rng.randrange(n_vol)
                    sources = [nm, nm, "group:" + rng.choice(["g1", "g2"])]
                else:
                    sources = ["name%02d" % i for i in rng.sample(range(n_vol + 1), rng.choice([1, 1, 2, min(3, n_vol)]))]
                mounts = [kv.cluster_mount(src, "/m%d" % m, rng.random() < 0.4) for m, src in enumerate(sources)]
                if rng.random() < 0.2:
                    mounts.insert(rng.randrange(len(mounts) + 1), {"Type": "BIND", "Source": "/x", "Target": "/y"})
            svc = rng.randrange(4)
            if svc == 3:
                group_mounts = mounts if group_mounts is None else group_mounts
                mounts = group_mounts
            t = sc.pending("t%05d" % tid, "svc%d" % svc, spec_version=3 if svc == 3 else None, **({"Spec": {"Container": {"Mounts": mounts}}} if mounts else {}))
            docs[t["ID"]] = t
            waiting.append(t["ID"])
            o.create_task(copy.deepcopy(t))
            e.create_task(copy.deepcopy(t))
        do = {d["ID"]: d for d in o.tick()}
        # the double answers the tasks of a service in the order the host layer hands them over: the queue's order, task id order here
        for t_id in sorted(do):
            d = do[t_id]
            # (a choice that stopped at a mount without a volume: the double reports the prefix it had chosen, as the engine does — "VolumePrefix" is
            # the oracle harness telling what chooseTaskVolumes had picked by then)
            fakelib.script(e.e, d["ServiceID"] if "ServiceID" in d else docs[t_id]["ServiceID"], d["NodeID"], [v["ID"] for v in d.get("Volumes") or []] or d.get("VolumePrefix") or [])
        de = {d["ID"]: d for d in e.tick()}
        assert sorted(do) == sorted(de)
        for t_id in do:
            a, b = do[t_id], de[t_id]
            assert (a["NodeID"], a["State"]) == (b["NodeID"], b["State"]), (t_id, a, b)
            assert [(v["ID"], v["Source"], v["Target"]) for v in a.get("Volumes") or []] == [(v["ID"], v["Source"], v["Target"]) for v in b.get("Volumes") or []], (t_id, a, b)
            if a["NodeID"]:
                placed.append(a)
                n_placed += 1
                n_attached += 1 if a.get("Volumes") else 0
        for vid in vids:
            io, ie = o.volume_info(vid), e.volume_info(vid)
            assert (io is None) == (ie is None), vid
            if io is None:
                continue
            assert io["Tasks"] == ie["Tasks"], (vid, io, ie)
            assert {k: c for k, c in io["Nodes"].items() if c} == {k: c for k, c in ie["Nodes"].items() if c}, (vid, io, ie)
        assert o.free_volumes() == e.free_volumes()
        rng.shuffle(placed)
        for d in placed[: len(placed) // 3]:
            doc = dict(docs[d["ID"]], NodeID=d["NodeID"], Status={"State": orc.ASSIGNED}, Volumes=d.get("Volumes") or [])
            o.delete_task(copy.deepcopy(doc))
            e.delete_task(copy.deepcopy(doc))
        placed = placed[len(placed) // 3:]
    assert n_placed > 0
    print("placed %d, with attachments %d" % (n_placed, n_attached))
