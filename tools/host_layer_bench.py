"""The JSON host layer alone (swp::Scheduler, csrc/swp_sched.cpp) over the scripted engine double (tests/fake_swp.cpp): what a cfg3-sized
tick costs ABOVE the engine ABI — event parsing, task maps, descriptors, decisions as JSON. No GPU, no placement logic.
    python tools/host_layer_bench.py [--tasks 100000] [--nodes 10000] [--log] [--dump DIR]
The double writes a line per task into its call log unless told not to (SWP_FAKE_QUIET, set here): --log leaves the log on, which is how
the numbers up to round 5's first half were taken (≈ 0.1 s of a 100k-task tick is the double formatting that log). --dump DIR writes the
events as files for tools/host_layer_prof.cpp (the same run as one native process, for gprof) and stops."""
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
ap.add_argument("--log", action="store_true", help="leave the double's call log on")
ap.add_argument("--dump", help="write nodes.jsonl / services.txt / tasks.jsonl into this directory and stop")
args = ap.parse_args()
if not args.log:
    os.environ["SWP_FAKE_QUIET"] = "1"
os.environ["SWP_FAKE_O3"] = "1"   # the host layer as the product builds it (-O3), not as the tests do (-O1)
wl = synth.Workload("cfg3", T=args.tasks, N=args.nodes, grouped=args.grouped)
if args.dump:
    os.makedirs(args.dump, exist_ok=True)
    with open(os.path.join(args.dump, "nodes.jsonl"), "w") as f:
        f.writelines(json.dumps(wl.node_doc(i)) + "\n" for i in range(wl.N))
    with open(os.path.join(args.dump, "services.txt"), "w") as f:
        f.writelines(wl.service_id(k) + "\n" for k in range(wl.S))
    with open(os.path.join(args.dump, "tasks.jsonl"), "w") as f:
        f.writelines(json.dumps(wl.task_doc(j)) + "\n" for j in range(wl.T))
    sys.exit(0)
s = sched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
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
                  "tick_s": round(t4 - t3, 3), "us_per_decided_task": round((t4 - t3) / wl.T * 1e6, 2), "decisions_bytes": n, "double_log": bool(args.log), "rc": rc}))
