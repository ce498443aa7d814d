#!/bin/bash
# round 4: short fresh batches with either resolver family (where is the crossover?) and the churn round's host-side pieces
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-c}; shift
O=gpurun_out/$TAG; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/crossover.txt
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from swarmkit_amd import abi, host, synth
for T in (256, 1024, 2048, 4096, 8192, 16000):
    for R in ("5", "6"):
        os.environ["SWP_RESOLVER"] = R
        wl = synth.Workload("cfg3", T=T, N=10000)
        s = host.HostScheduler(engine=abi.Engine(profile=True))
        descs = host.load_workload(s, wl)
        s.e.state_save()
        ms, wall = [], []
        for i in range(4):
            s.e.state_restore()
            t0 = time.perf_counter()
            out, _ = s.e.schedule_batch(descs, want_hist=False)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(s.e.stats()["ms_total"])
        print("fresh cfg3 T=%5d N=10000 resolver %s: device ms %.3f wall ms %.3f placed %d launches %d" % (T, R, min(ms[1:]), min(wall[1:]), int((out >= 0).sum()), s.e.stats()["resolve_launches"]))
os.environ.pop("SWP_RESOLVER", None)
# the churn round, piece by piece
wl = synth.Workload("cfg3")
s = host.HostScheduler(engine=abi.Engine(profile=True))
descs = host.load_workload(s, wl)
eng = s.e
out, _ = eng.schedule_batch(descs, want_hist=False)
assign = out.astype(np.int64).copy()
rng = np.random.default_rng(wl.seed)
prev = np.zeros(0, dtype=np.int64)
acc = {}
def lap(name, t0):
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
R = 10
for rnd in range(R):
    t0 = time.perf_counter(); drained = rng.choice(wl.N, size=wl.N // 10, replace=False); touched = np.concatenate([prev, drained]).astype(np.uint32); lap("choose", t0)
    t0 = time.perf_counter(); rows = eng.node_get_many(touched); lap("node_get_many", t0)
    t0 = time.perf_counter()
    upd = np.zeros(len(touched), dtype=abi.NODE_DYNAMIC_DTYPE)
    upd["node"], upd["cpu"], upd["mem"], upd["total"] = touched, rows["cpu"], rows["mem"], rows["total"]
    upd["flags"] = np.where(np.arange(len(touched)) < len(prev), rows["flags"] | abi.NODE_READY, rows["flags"] & ~np.uint32(abi.NODE_READY))
    lap("numpy rows", t0)
    t0 = time.perf_counter(); eng.node_update_dynamic_many(upd); lap("node_update_dynamic_many", t0)
    t0 = time.perf_counter(); gone = np.nonzero(np.isin(assign, drained))[0]; lap("numpy isin", t0)
    t0 = time.perf_counter()
    pl = np.zeros(len(gone), dtype=abi.PLACEMENT_DTYPE)
    pl["node"], pl["service"] = assign[gone], descs["service"][gone]
    pl["cpu"], pl["mem"], pl["counted"] = descs["cpu"][gone], descs["mem"][gone], 1
    lap("numpy placements", t0)
    t0 = time.perf_counter(); eng.commit(pl, add=False); lap("commit(remove)", t0)
    t0 = time.perf_counter(); d2 = descs[gone]; lap("numpy descs", t0)
    t0 = time.perf_counter(); b = eng.batch_prepare(d2); lap("batch_prepare", t0)
    t0 = time.perf_counter(); b.run(); lap("batch_run", t0)
    t0 = time.perf_counter(); new_out, _ = b.results(want_hist=False); lap("batch_results", t0)
    b.free()
    acc["device"] = acc.get("device", 0.0) + eng.stats()["ms_total"]
    acc["launches"] = acc.get("launches", 0.0) + eng.stats()["resolve_launches"]
    assign[gone] = new_out
    prev = drained
print("churn round pieces, ms per round:", {k: round(v / R, 3) for k, v in acc.items()}, "tasks per round", len(gone))
PY
