"""GPU: examples/place_demo.c — a plain C program over include/swp.h + include/swp_sched.h — is compiled, run on the
device, and its decisions are compared with the oracle fed the same documents."""
import json
import os
import re
import subprocess

import pytest

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_c_consumer_matches_oracle(tmp_path):
    from swarmkit_amd import abi
    if not os.path.exists(abi.LIB_PATH):
        abi.build_library()
    exe = str(tmp_path / "place_demo")
    libdir = os.path.join(ROOT, "swarmkit_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "place_demo.c"),
                    "-L" + libdir, "-lswp", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.splitlines()
    got = {d["ID"]: (d["NodeID"], d["State"], d["Err"]) for d in json.loads(out[0])}
    # the same documents through the oracle
    src = open(os.path.join(ROOT, "examples", "place_demo.c")).read()
    o = orc.Oracle()
    zones = {"node-a": ("east", 4), "node-b": ("east", 2), "node-c": ("west", 8)}
    for nid, (zone, cpus) in zones.items():
        assert '\\"ID\\":\\"%s\\"' % nid in src and '\\"zone\\":\\"%s\\"' % zone in src
        o.create_node({"ID": nid, "Status": {"State": 2}, "Spec": {"Availability": 0, "Annotations": {"Labels": {"zone": zone}}},
                       "Description": {"Resources": {"NanoCPUs": cpus * 10**9, "MemoryBytes": 8 << 30}}})
    o.set_service("web")
    for i in range(5):
        o.create_task({"ID": "task-%d" % i, "ServiceID": "web", "DesiredState": 512, "Status": {"State": 64},
                       "Spec": {"Resources": {"Reservations": {"NanoCPUs": 10**9}}, "Placement": {"Constraints": ["node.labels.zone == east"]}}})
    want = {d["ID"]: (d["NodeID"], d["State"], d["Err"]) for d in o.tick()}
    assert got == want
    assert sorted(n for n, _, _ in got.values()) == ["node-a"] * 3 + ["node-b"] * 2
    assert re.match(r"placed 5, no suitable node 0, 3 nodes", out[1])


def _struct_demo_oracle():
    """The cluster of examples/struct_abi_demo.c as api.Node / api.Task / api.Volume documents through the oracle: the lines the demo must print."""
    GIB = 1 << 30
    o = orc.Oracle()

    def node(i, drain=False):
        d = {"ID": "n%d" % i, "Meta": {"Version": {"Index": 1}}, "Status": {"State": 2}, "Spec": {"Availability": 2 if drain else 0, "Annotations": {"Labels": {"zone": "z%d" % (i % 3)}}},
             "Description": {"Hostname": "h%d" % i, "Platform": {"OS": "linux", "Architecture": "arm64" if i % 4 == 3 else "amd64"},
                             "Resources": {"NanoCPUs": (2 + i % 3) * 10**9, "MemoryBytes": 8 * GIB}}}
        if i % 2 == 0:
            d["Description"]["CSIInfo"] = [{"PluginName": "csi", "AccessibleTopology": {"Segments": {"zone": "z%d" % (i % 3)}}}]
        return d
    for i in range(12):
        o.create_node(node(i))
    o.update_volume({"ID": "vol-a", "Spec": {"Annotations": {"Name": "vol-a"}, "Group": "g", "Driver": {"Name": "csi"}, "AccessMode": {"Scope": "MULTI_NODE", "Sharing": "ALL"}},
                     "VolumeInfo": {"VolumeID": "p-a", "AccessibleTopology": [{"Segments": {"zone": "z0"}}]}})
    o.update_volume({"ID": "vol-b", "Spec": {"Annotations": {"Name": "vol-b"}, "Group": "g", "Driver": {"Name": "csi"}, "AccessMode": {"Scope": "SINGLE_NODE", "Sharing": "NONE"}},
                     "VolumeInfo": {"VolumeID": "p-b"}})
    specs = [("web", {"Resources": {"Reservations": {"NanoCPUs": 10**9}}, "Placement": {"Constraints": ["node.labels.zone==z1"], "Platforms": [{"OS": "linux", "Architecture": "amd64"}]}}),
             ("db", {"Resources": {"Reservations": {"NanoCPUs": 5 * 10**8, "MemoryBytes": GIB}}, "Placement": {"Constraints": ["node.labels.zone!=z1"]},
                     "Container": {"Mounts": [{"Type": "CLUSTER", "Source": "group:g", "Target": "/data"}]}}),
             ("batch", {"Resources": {"Reservations": {"NanoCPUs": 2 * 10**9}}})]
    for svc, _ in specs:
        o.set_service(svc)

    def task(tid, k):
        return {"ID": "t%03d" % tid, "ServiceID": specs[k][0], "DesiredState": 512, "Status": {"State": 64}, "Spec": specs[k][1]}

    def lines(batch, decisions, first):
        out = []
        for d in sorted(decisions, key=lambda d: d["ID"]):
            i = int(d["ID"][1:]) - first
            if d["NodeID"]:
                out.append("B%d t%d %s%s" % (batch, i, d["NodeID"], " [%s]" % d["Volumes"][0]["ID"] if d.get("Volumes") else ""))
            else:
                why = d["Err"][len("no suitable node ("):-1] if d["Err"].startswith("no suitable node (") else ""
                out.append("B%d t%d - | %s" % (batch, i, why))
        return out
    docs = {i: task(i, i % 3) for i in range(40)}
    for i in range(40):
        o.create_task(docs[i])
    d1 = o.tick()
    want = lines(1, d1, 0)
    # n0 and n1 are drained; the tasks on them go away (their volumes with them); as many new tasks of the same services arrive
    for i in (0, 1):
        o.update_node(node(i, drain=True)) if hasattr(o, "update_node") else o.create_node(node(i, drain=True))
    gone = [d for d in sorted(d1, key=lambda d: d["ID"]) if d["NodeID"] in ("n0", "n1")]
    for d in gone:
        o.delete_task(dict(docs[int(d["ID"][1:])], NodeID=d["NodeID"], Status={"State": 512}, Volumes=d.get("Volumes") or []))
    for q, d in enumerate(gone):
        o.create_task(task(40 + q, int(d["ID"][1:]) % 3))
    # (the first batch's unplaceable tasks are still queued in the oracle: the demo's second batch only holds the replacements)
    d2 = [d for d in o.tick() if int(d["ID"][1:]) >= 40]
    return want + lines(2, d2, 40), len(gone)


@pytest.mark.parametrize("shards", [0, 3])
def test_struct_abi_consumer_matches_oracle(tmp_path, shards):
    """examples/struct_abi_demo.c: numeric rows and interned ids in, node indices and volume indices out — the calls of shim/go/swp_cgo.go in
    its order, incl. the incremental path (a drain, swp_commit(remove), a second batch) — on one engine and on a shard set of three."""
    from swarmkit_amd import abi
    if not os.path.exists(abi.LIB_PATH):
        abi.build_library()
    exe = str(tmp_path / "struct_abi_demo")
    libdir = os.path.join(ROOT, "swarmkit_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "struct_abi_demo.c"),
                    "-L" + libdir, "-lswp", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe] + ([str(shards)] if shards else []), check=True, capture_output=True, text=True, timeout=120).stdout.splitlines()
    want, n_gone = _struct_demo_oracle()
    assert n_gone > 0 and any("[vol-" in x for x in want) and any(" - | " in x for x in want)
    assert out[:-1] == want
    assert re.match(r"placed \d+, no suitable node \d+, 12 nodes, resolver %s" % ("7" if shards else r"\d+"), out[-1])
