"""Known-answer vectors for the CSI-volume part of the path, re-encoded from the reference's own tests (no code copied: the tables
are data): topology_test.go:8-170 (IsInTopology), volumes_test.go:196-342 (checkVolume), :344-466 (isVolumeAvailableOnNode),
:468-531 (chooseTaskVolumes), :100-160 (reserveTaskVolumes), scheduler_ginkgo_test.go:376-596 (setupTasksList with volumes in use)
and :80-372 (a task with a cluster mount through the running scheduler). Shared by tests/test_oracle_volumes.py (the oracle) and the
host-layer / engine tests."""


def seg(**kw):
    return {"Segments": dict(kw)}


# (top, accessible, expected) — topology_test.go:14-166
TOPOLOGY = [
    (seg(region="R1", zone="Z1"), [seg(region="R1", zone="Z1")], True),
    (seg(region="R1", zone="Z2"), [seg(region="R1", zone="Z1"), seg(region="R1", zone="Z2")], True),
    (seg(region="R1", zone="Z3"), [seg(region="R1")], True),
    (seg(region="R1", zone="Z1"), [seg(region="R2", zone="Z1")], False),
    (seg(region="R1", zone="Z1", shelf="S1"), [seg(region="R1", zone="Z1"), seg(region="R1", zone="Z2")], True),
    (seg(region="R1", zone="Z1", shelf="S1"), [seg(region="R1", zone="Z1", shelf="S2"), seg(region="R1", zone="Z2", shelf="S1")], False),
    (seg(region="R1", zone="Z1", shelf="S1"), [seg(region="R1", zone="Z1", shelf="S2"), seg(region="R1", zone="Z2", shelf="S1"), seg(region="R1", zone="Z1", shelf="S1")], True),
    # topology.go:25-27: anything missing fits
    (None, [seg(region="R1")], True),
    (seg(region="R1"), [], True),
]

SINGLE, MULTI = "SINGLE_NODE", "MULTI_NODE"
NONE, READ_ONLY, ONE_WRITER, ALL = "NONE", "READ_ONLY", "ONE_WRITER", "ALL"
UNUSED, WRONG_NODE, ONLY_READERS, WRITER = range(4)

# volumes_test.go:270-341: (name, access mode or None (= single node / all), in use, in topology, read-only mount, expected)
CHECK_VOLUME = [
    ("volume outside of node topology", None, UNUSED, False, False, False),
    ("volume in use on a different node", None, WRONG_NODE, True, False, False),
    ("volume is read only, mount is not", (MULTI, READ_ONLY), UNUSED, True, False, False),
    ("volume is OneWriter, but already has a writer", (MULTI, ONE_WRITER), WRITER, True, False, False),
    ("volume is OneWriter, and has no writer", (MULTI, ONE_WRITER), ONLY_READERS, True, False, True),
    ("volume not in use and is in topology", None, UNUSED, True, False, True),
    ("in use on a different node, but the scope is multinode", (MULTI, ALL), WRONG_NODE, True, False, True),
    ("the volume is in use and cannot be shared", (SINGLE, NONE), ONLY_READERS, True, True, False),
    ("the volume is not in use and cannot be shared", (SINGLE, NONE), UNUSED, True, True, True),
]


def check_volume_case(mode, in_use, in_topology):
    """The volume, node and reservations of one checkVolume table entry (volumes_test.go:213-268)."""
    scope, sharing = mode or (SINGLE, ALL)
    v = {"ID": "someVolume", "Spec": {"AccessMode": {"Scope": scope, "Sharing": sharing}, "Driver": {"Name": "somePlugin"}},
         "VolumeInfo": {"VolumeID": "somePluginVolumeID", "AccessibleTopology": [seg(zone="z1")]}}
    n = {"ID": "someNode", "Description": {"CSIInfo": [{"PluginName": "somePlugin", "AccessibleTopology": seg(zone="z1" if in_topology else "z2")}]}}
    reserve = {UNUSED: [], WRONG_NODE: [("someVolume", "someTask", "someOtherNode", False)], ONLY_READERS: [("someVolume", "someTask", "someNode", True)],
               WRITER: [("someVolume", "someTask", "someNode", True), ("someVolume", "someWriter", "someNode", False)]}[in_use]
    return v, n, reserve


def canned_volume(i, group="group"):
    """volumes_test.go:17-38"""
    return {"ID": "volumeID%d" % i, "Spec": {"Annotations": {"Name": "volume%d" % i}, "Group": group, "Driver": {"Name": "driver"},
                                             "AccessMode": {"Scope": MULTI, "Sharing": ALL}}, "VolumeInfo": {"VolumeID": "volumePlugin%d" % i}}


def group_fixture():
    """volumes_test.go:350-441: one node with a CSI plugin and no topology, four single-node volumes, two of them in a group."""
    node = {"ID": "someNode", "Description": {"CSIInfo": [{"PluginName": "newPlugin", "NodeID": "newPluginSomeNode"}]}}

    def vol(i, group="", created=True):
        v = {"ID": "volume%d" % i, "Spec": {"Annotations": {"Name": "volumeName%d" % i}, "Driver": {"Name": "newPlugin"}, "Group": group,
                                            "AccessMode": {"Scope": SINGLE, "Sharing": ALL}}}
        if created:
            v["VolumeInfo"] = {"VolumeID": "newPluginVolume%d" % i}
        return v
    # (volume2 has no VolumeInfo in the reference's fixture; a volumeSet takes it all the same: only the scheduler's event handler filters)
    return node, [vol(1), vol(3, "someVolumeGroup"), vol(4, "someVolumeGroup")]


def cluster_mount(source, target, read_only=False):
    m = {"Type": "CLUSTER", "Source": source, "Target": target}
    if read_only:
        m["ReadOnly"] = True
    return m


def free_volumes_fixture():
    """volumes_test.go:530-642 (Describe "freeVolumes", BeforeEach): four nodes, four volumes each PUBLISHED on "its" node, a fifth volume
    PUBLISHED on all four; task i sits on node i and uses volume i and the fifth one. Returns (nodes, volumes, all_volume, tasks)."""
    all_volume = canned_volume(5)
    all_volume["PublishStatus"] = []
    nodes, volumes, tasks = [], [], []
    for i in range(4):
        v = canned_volume(i)
        n = {"ID": "node%d" % i, "Description": {}}
        v["PublishStatus"] = [{"NodeID": n["ID"], "State": "PUBLISHED"}]
        all_volume["PublishStatus"].append({"NodeID": n["ID"], "State": "PUBLISHED"})
        t = {"ID": "task%d" % i, "ServiceID": "svc", "NodeID": n["ID"], "DesiredState": 512, "Status": {"State": 512},
             "Spec": {"Container": {"Mounts": [cluster_mount(v["Spec"]["Annotations"]["Name"], "bar"), cluster_mount(all_volume["Spec"]["Annotations"]["Name"], "baz")]}},
             "Volumes": [{"Source": v["Spec"]["Annotations"]["Name"], "Target": "bar", "ID": v["ID"]},
                         {"Source": all_volume["Spec"]["Annotations"]["Name"], "Target": "baz", "ID": all_volume["ID"]}]}
        nodes.append(n)
        volumes.append(v)
        tasks.append(t)
    return nodes, volumes, all_volume, tasks
