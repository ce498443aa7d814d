"""The JSON host layer alone (swp::Scheduler, csrc/swp_sched.cpp) over the scripted engine double (tests/fake_swp.cpp): what a cfg3-sized
tick costs ABOVE the engine ABI — event parsing, task maps, descriptors, decisions as JSON. No GPU, no placement logic.
    python tools/host_layer_bench.py [--tasks 100000] [--nodes 10000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fakelib  # noqa: E402
from swarmkit_amd import abi, sched, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tasks", type=int, default=100000)
ap.add_argument("--nodes", type=int, default=10000)
ap.add_argument("--grouped", action="store_true")
args = ap.parse_args()
wl = synth.Workload("cfg3", T=args.tasks, N=args.nodes, grouped=args.grouped)
s = sched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
fakelib.quiet(s.e) if hasattr(fakelib, "quiet") else None
t0 = time.perf_counter()
for i in range(wl.N):
    s.create_node(wl.node_doc(i))
for k in range(wl.S):
    s.set_service(wl.service_id(k), spec_version=1 if args.grouped else None)
t1 = time.perf_counter()
docs = [json.dumps(wl.task_doc(j)).encode() for j in range(wl.T)]
t2 = time.perf_counter()
import ctypes as C
flag = C.c_int(0)
for b in docs:
    s.L.swp_sched_create_task(s.h, b, len(b), C.byref(flag))
t3 = time.perf_counter()
out = C.c_char_p()
rc = s.L.swp_sched_tick(s.h, C.byref(out))
t4 = time.perf_counter()
n = len(out.value)
print(json.dumps({"nodes": wl.N, "tasks": wl.T, "create_node_s": round(t1 - t0, 3), "create_task_s": round(t3 - t2, 3), "us_per_create_task": round((t3 - t2) / wl.T * 1e6, 2),
                  "tick_s": round(t4 - t3, 3), "us_per_decided_task": round((t4 - t3) / wl.T * 1e6, 2), "decisions_bytes": n, "rc": rc}))
