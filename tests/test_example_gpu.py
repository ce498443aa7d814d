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
