"""CPU: the oracle's restatement of volumes.go / topology.go / VolumesFilter against the reference's own test vectors (tests/kat_volumes.py),
and the scheduler-level scenarios of scheduler_ginkgo_test.go through the oracle."""
import pytest

import kat_volumes as kv
import orc
import scenarios as sc


@pytest.mark.parametrize("top,accessible,want", kv.TOPOLOGY)
def test_is_in_topology(top, accessible, want):
    assert orc.volumes(topology={"Top": top, "Accessible": accessible}) is want


@pytest.mark.parametrize("name,mode,in_use,in_top,ro,want", kv.CHECK_VOLUME, ids=[c[0] for c in kv.CHECK_VOLUME])
def test_check_volume_table(name, mode, in_use, in_top, ro, want):
    v, n, reserve = kv.check_volume_case(mode, in_use, in_top)
    assert orc.volumes([v], reserve, n, check={"ID": v["ID"], "ReadOnly": ro}) is want


def test_volume_or_group_availability_on_a_node():
    node, vols = kv.group_fixture()
    assert orc.volumes(vols, node=node, mount={"Source": "volumeName1"}) == "volume1"
    assert orc.volumes(vols, node=node, mount={"Source": "volumeNameNotReal"}) == ""
    assert orc.volumes(vols, node=node, mount={"Source": "group:someVolumeGroup"}) in ("volume3", "volume4")
    assert orc.volumes(vols, node=node, mount={"Source": "group:someVolumeGroup"}) == "volume3"   # canonical order: the first one added
    assert orc.volumes(vols, node=node, mount={"Source": "group:noSuchGroup"}) == ""
    # the first of the group in use on another node (single-node scope): the second one
    assert orc.volumes(vols, [("volume3", "t", "elsewhere", False)], node, mount={"Source": "group:someVolumeGroup"}) == "volume4"


def test_choose_task_volumes():
    """volumes_test.go:468-531: a group mount, two named mounts and a bind mount between them; a node without CSI info."""
    v1, v2, v3 = kv.canned_volume(1, "volumeGroup"), kv.canned_volume(2), kv.canned_volume(3)
    mounts = [kv.cluster_mount("group:volumeGroup", "/somedir", True), kv.cluster_mount("volume2", "/someOtherDir"),
              {"Type": "BIND", "Source": "/some/subdir", "Target": "/some/container/dir"}, kv.cluster_mount("volume3", "/some/third/dir")]
    task = {"ID": "taskID1", "Spec": {"Container": {"Mounts": mounts}}}
    node = {"ID": "node1", "Description": {}}
    got = orc.volumes([v1, v2, v3], node=node, task=task)
    assert got["Err"] == ""
    assert got["Attachments"] == [{"ID": "volumeID1", "Source": "group:volumeGroup", "Target": "/somedir"},
                                  {"ID": "volumeID2", "Source": "volume2", "Target": "/someOtherDir"},
                                  {"ID": "volumeID3", "Source": "volume3", "Target": "/some/third/dir"}]
    # a mount nothing satisfies: the reference's error string, no attachments (volumes.go:122-127)
    task2 = {"ID": "t2", "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume2", "/a"), kv.cluster_mount("nothing", "/b")]}}}
    got = orc.volumes([v1, v2, v3], node=node, task=task2)
    assert got == {"Attachments": [], "Err": "cannot find volume to satisfy mount with source nothing"}
    # two mounts of one task on a volume that cannot be shared: the second finds the first one's reservation (volumes.go:128)
    solo = dict(kv.canned_volume(7), Spec=dict(kv.canned_volume(7)["Spec"], AccessMode={"Scope": kv.SINGLE, "Sharing": kv.NONE}))
    task3 = {"ID": "t3", "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume7", "/a"), kv.cluster_mount("volume7", "/b")]}}}
    assert orc.volumes([solo], node=node, task=task3)["Err"] == "cannot find volume to satisfy mount with source volume7"


def _csi_node(i):
    return {"ID": "nodeID%d" % i, "Spec": {"Annotations": {"Name": "node%d" % i}}, "Status": {"State": orc.READY},
            "Description": {"Hostname": "nodeHost%d" % i, "CSIInfo": [{"PluginName": "somePlug", "NodeID": "nodeCSI%d" % i}]}}


def _vol(i, group, scope, sharing):
    return {"ID": "volumeID%d" % i, "Spec": {"Annotations": {"Name": "volume%d" % i}, "Group": group, "Driver": {"Name": "somePlug"},
                                             "AccessMode": {"Scope": scope, "Sharing": sharing}}, "VolumeInfo": {"VolumeID": "csi%d" % i}}


def test_scheduler_initialization_tracks_the_volumes_in_use():
    """scheduler_ginkgo_test.go:376-596: a running task reserves its attachments, a shut-down one and a pending preassigned one do not."""
    o = orc.Oracle()
    for i in range(3):
        o.create_node(_csi_node(i))
    for v in (_vol(1, "group1", kv.MULTI, kv.ALL), _vol(2, "group2", kv.SINGLE, kv.NONE), _vol(3, "group2", kv.SINGLE, kv.NONE)):
        o.update_volume(v)
    running = {"ID": "runningTask", "NodeID": "nodeID0", "Status": {"State": orc.RUNNING}, "DesiredState": orc.RUNNING,
               "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume1", "/var/"), kv.cluster_mount("group:group2", "/home/")]}},
               "Volumes": [{"Source": "volume1", "Target": "/var/", "ID": "volumeID1"}, {"Source": "group:group2", "Target": "/home/", "ID": "volumeID3"}]}
    shutdown = {"ID": "shutdownTask", "NodeID": "nodeID1", "Status": {"State": orc.SHUTDOWN}, "DesiredState": orc.SHUTDOWN,
                "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume1", "/foo/")]}}, "Volumes": [{"Source": "volume1", "Target": "/foo/", "ID": "volumeID1"}]}
    pending = {"ID": "pendingID", "NodeID": "nodeID2", "Status": {"State": orc.PENDING}, "DesiredState": orc.RUNNING,
               "Spec": {"Container": {"Mounts": [kv.cluster_mount("group:group2", "/foo/")]}}}
    for t in (running, shutdown, pending):
        o.setup_task(t)
    assert o.volume_info("volumeID1")["Tasks"] == {"runningTask": {"NodeID": "nodeID0", "ReadOnly": False}}
    assert o.volume_info("volumeID2")["Tasks"] == {}
    assert o.volume_info("volumeID3")["Tasks"] == {"runningTask": {"NodeID": "nodeID0", "ReadOnly": False}}
    # the pending preassigned task: volumeID3 is taken (sharing none), volumeID2 is free -> it fits its node with volumeID2
    d = o.process_preassigned()
    assert [(x["ID"], x["NodeID"], x["State"], x.get("Volumes")) for x in d] == [("pendingID", "nodeID2", orc.ASSIGNED, [{"ID": "volumeID2", "Source": "group:group2", "Target": "/foo/"}])]
    assert o.volume_info("volumeID2")["Tasks"] == {}   # taskFitNode chooses, it does not reserve (scheduler.go:663-677)
    # deleting the running task releases what it held (scheduler.go:355-358)
    o.delete_task(running)
    assert o.volume_info("volumeID3") == {"Tasks": {}, "Nodes": {"nodeID0": 0}}


def test_a_task_with_a_cluster_mount_through_the_tick():
    """scheduler_ginkgo_test.go:80-372: without the volume the task stays pending with the VolumesFilter's explanation; a volume that exists
    only as a spec (no VolumeInfo yet) does not count; once created, the task is assigned with its attachment and the volume is reserved."""
    o = orc.Oracle()
    o.create_node({"ID": "nodeID1", "Status": {"State": orc.READY}, "Description": {"CSIInfo": [{"PluginName": "somePlug", "NodeID": "nodeCSI1"}]}})
    o.set_service("service1")
    task = sc.pending("task1", "service1", Spec={"Container": {"Mounts": [kv.cluster_mount("volume1", "/var/")]}})
    o.create_task(task)
    d = o.tick()
    assert [(x["ID"], x["NodeID"], x["Err"]) for x in d] == [("task1", "", "no suitable node (cannot fulfill requested CSI volume mounts on 1 node)")]
    vol = {"ID": "volumeID1", "Spec": {"Annotations": {"Name": "volume1"}, "Driver": {"Name": "somePlug"}, "AccessMode": {"Scope": kv.SINGLE, "Sharing": kv.NONE}}}
    o.update_volume(vol)   # not created by the plugin yet: ignored (scheduler.go:207)
    assert o.volume_info("volumeID1") is None
    assert o.tick()[0]["NodeID"] == ""
    o.update_volume(dict(vol, VolumeInfo={"VolumeID": "csi1"}))
    d = o.tick()
    assert [(x["ID"], x["NodeID"], x["State"], x["Volumes"]) for x in d] == [("task1", "nodeID1", orc.ASSIGNED, [{"ID": "volumeID1", "Source": "volume1", "Target": "/var/"}])]
    assert o.volume_info("volumeID1") == {"Tasks": {"task1": {"NodeID": "nodeID1", "ReadOnly": False}}, "Nodes": {"nodeID1": 1}}
    # a second task of the service: the volume cannot be shared
    o.create_task(sc.pending("task2", "service1", Spec={"Container": {"Mounts": [kv.cluster_mount("volume1", "/var/")]}}))
    assert o.tick()[0]["Err"] == "no suitable node (cannot fulfill requested CSI volume mounts on 1 node)"


def test_any_requested_mount_passes_the_filter_but_every_mount_needs_a_volume():
    """filter.go:424-432 passes a node when ANY cluster mount is satisfiable; chooseTaskVolumes then fails for
    the other mount — the reference logs the error and assigns the task WITHOUT attachments (scheduler.go:862-872)."""
    o = orc.Oracle()
    o.create_node({"ID": "n1", "Status": {"State": orc.READY}, "Description": {}})
    o.set_service("svc")
    o.update_volume(kv.canned_volume(1))
    o.create_task(sc.pending("t1", "svc", Spec={"Container": {"Mounts": [kv.cluster_mount("volume1", "/a"), kv.cluster_mount("missing", "/b")]}}))
    d = o.tick()
    assert [(x["ID"], x["NodeID"], x["State"], x.get("Volumes")) for x in d] == [("t1", "n1", orc.ASSIGNED, None)]
    assert o.volume_info("volumeID1")["Tasks"] == {}


def _free_volumes_cluster(s):
    nodes, volumes, all_volume, tasks = kv.free_volumes_fixture()
    for n in nodes:
        s.create_node(dict(n, Status={"State": orc.READY}))
    for v in volumes + [all_volume]:
        s.update_volume(v)
    s.set_service("svc")
    for t in tasks:
        s.setup_task(t)
    return nodes, volumes, all_volume, tasks


def test_free_volumes_reference_counts():
    """volumes_test.go:644-661: every volume is referenced once per node that uses it."""
    o = orc.Oracle()
    nodes, volumes, all_volume, tasks = _free_volumes_cluster(o)
    for i, v in enumerate(volumes):
        assert o.volume_info(v["ID"])["Nodes"] == {nodes[i]["ID"]: 1}
    assert o.volume_info(all_volume["ID"])["Nodes"] == {n["ID"]: 1 for n in nodes}
    assert o.free_volumes() == []   # (nothing to free while every publication is in use)


def test_free_volumes_that_are_no_longer_needed():
    """volumes_test.go:663-697: task0 lets go of its two volumes; volume 0 and the shared volume are to be unpublished from node0, every
    other publication stays."""
    o = orc.Oracle()
    nodes, volumes, all_volume, tasks = _free_volumes_cluster(o)
    o.delete_task(tasks[0])   # releaseVolume(volumes[0].ID, task0) + releaseVolume(allVolume.ID, task0), scheduler.go:330-332
    got = o.free_volumes()
    assert got == [{"VolumeID": volumes[0]["ID"], "NodeIDs": ["node0"]}, {"VolumeID": all_volume["ID"], "NodeIDs": ["node0"]}]
    assert o.free_volumes() == []   # the store's copy says PENDING_NODE_UNPUBLISH now


def test_free_volumes_leaves_other_states_alone():
    """volumes.go:200: only a PUBLISHED status on a node with a zero reference count changes."""
    o = orc.Oracle()
    v = kv.canned_volume(1)
    v["PublishStatus"] = [{"NodeID": "a", "State": "PENDING_PUBLISH"}, {"NodeID": "b", "State": "PUBLISHED"}, {"NodeID": "c", "State": "PENDING_UNPUBLISH"},
                          {"NodeID": "d", "State": "PENDING_NODE_UNPUBLISH"}, {"NodeID": "e", "State": "PUBLISHED"}]
    o.update_volume(v)
    o.create_node({"ID": "e", "Status": {"State": orc.READY}, "Description": {}})
    o.set_service("svc")
    o.setup_task({"ID": "t", "ServiceID": "svc", "NodeID": "e", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
                  "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume1", "/m")]}}, "Volumes": [{"ID": v["ID"], "Source": "volume1", "Target": "/m"}]})
    assert o.free_volumes() == [{"VolumeID": v["ID"], "NodeIDs": ["b"]}]


def test_a_volume_reserved_twice_by_one_task_keeps_a_reference():
    """volumes.go:156-160 counts a node once per reserveVolume call, :169-178 gives back one per releaseVolume of a task that still holds
    the volume: a task with two mounts on ONE volume reserves twice and releases once — the publication is never freed (the
    reference's behaviour, restated as it is)."""
    o = orc.Oracle()
    v = kv.canned_volume(1)
    v["PublishStatus"] = [{"NodeID": "n", "State": "PUBLISHED"}]
    o.update_volume(v)
    o.create_node({"ID": "n", "Status": {"State": orc.READY}, "Description": {}})
    o.set_service("svc")
    t = {"ID": "t", "ServiceID": "svc", "NodeID": "n", "DesiredState": orc.RUNNING, "Status": {"State": orc.RUNNING},
         "Spec": {"Container": {"Mounts": [kv.cluster_mount("volume1", "/a"), kv.cluster_mount("volume1", "/b")]}},
         "Volumes": [{"ID": v["ID"], "Source": "volume1", "Target": "/a"}, {"ID": v["ID"], "Source": "volume1", "Target": "/b"}]}
    o.setup_task(t)
    assert o.volume_info(v["ID"])["Nodes"] == {"n": 2} and list(o.volume_info(v["ID"])["Tasks"]) == ["t"]
    o.delete_task(t)
    assert o.volume_info(v["ID"])["Nodes"] == {"n": 1} and o.volume_info(v["ID"])["Tasks"] == {}
    assert o.free_volumes() == []


def test_an_update_of_a_known_volume_keeps_its_first_object_and_joins_the_new_group():
    """volumes.go:62-82 to the letter: vs.volumes is a map of STRUCT values, so `info.volume = v` (:71) assigns to a copy — the volume
    object checkVolume reads stays the first one the set saw (a PAUSED volume that is updated to ACTIVE stays unusable; an ACTIVE one
    that is updated to PAUSE stays usable) — while byGroup gains the volume under the new object's group and keeps it under the old one
    (never pruned, :74-78) and byName gains the new name."""
    o = orc.Oracle()
    o.create_node({"ID": "n1", "Status": {"State": orc.READY}, "Description": {}})
    o.set_service("svc")
    paused = kv.canned_volume(1, group="g1")
    paused["Spec"]["Availability"] = "PAUSE"
    o.update_volume(paused)
    o.update_volume(kv.canned_volume(1, group="g1"))   # ACTIVE now, says the store — the scheduler's copy never hears of it
    o.create_task(sc.pending("t1", "svc", Spec={"Container": {"Mounts": [kv.cluster_mount("volume1", "/a")]}}))
    assert o.tick()[0]["Err"] == "no suitable node (cannot fulfill requested CSI volume mounts on 1 node)"
    active = kv.canned_volume(2, group="g1")
    o.update_volume(active)
    moved = kv.canned_volume(2, group="g2")
    moved["Spec"]["Availability"] = "PAUSE"
    moved["Spec"]["Annotations"]["Name"] = "renamed"
    o.update_volume(moved)
    # still ACTIVE for checkVolume, a member of g1 AND g2, known as volume2 AND renamed
    for k, src in enumerate(["group:g1", "group:g2", "volume2", "renamed"]):
        o.create_task(sc.pending("u%d" % k, "svc", Spec={"Container": {"Mounts": [kv.cluster_mount(src, "/m")]}}))
    got = {d["ID"]: (d["NodeID"], [v["ID"] for v in d.get("Volumes") or []]) for d in o.tick()}
    assert {k: got[k] for k in ("u0", "u1", "u2", "u3")} == {k: ("n1", ["volumeID2"]) for k in ("u0", "u1", "u2", "u3")}
