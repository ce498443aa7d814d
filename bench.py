#!/usr/bin/env python3
"""bench.py — placements/s of the batch task-placement hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" = one full pass of the hot path over one batch of synthetic pending tasks
(BASELINE.json configs[2]: 100k one-off tasks x 10k nodes, Resource + Constraint + Platform
filters, spread strategy), starting from the same cluster state every step, with the node rows and
task descriptors already resident in HBM. Inside the timed region, per step:
    device state restore (3 small D2D copies) -> class bitmaps -> [scan -> resolve/commit] per window
    -> explain pass -> placements copied back to the host (400 KB).
Host-side descriptor building / interning (what the Go shim does while it enqueues tasks) is
outside the timed region; its cost and the PCIe-inclusive rate are reported in DESIGN.md.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); see DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ROW_B = {"cfg2": 32, "cfg3": 48, "cfg4": 64}   # algorithmic node-row bytes per (task,node) pair, SURVEY.md §8a
TASK_B = 64 + 8                # descriptor + result per task, SURVEY.md §8d


def cpu_baseline(wl, budget_s=12.0):
    """The CPU oracle (single thread, like the reference's single scheduling goroutine) on a bounded
    sample of the SAME workload: the full node set, the first `sample` tasks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    o = orc.Oracle()
    for i in range(wl.N):
        o.create_node(wl.node_doc(i))
    for k in range(wl.S):
        o.set_service(wl.service_id(k))
    # calibrate on a small slice, then size the sample for ~budget_s of scheduling work
    probe = min(wl.T, max(50, 2_000_000 // max(wl.N, 1)))
    for j in range(probe):
        o.create_task(wl.task_doc(j))
    t0 = time.perf_counter()
    o.tick()
    dt = time.perf_counter() - t0
    rate = probe / max(dt, 1e-9)
    sample = int(min(wl.T - probe, max(0, rate * budget_s)))
    done, total_dt = probe, dt
    if sample > 0:
        for j in range(probe, probe + sample):
            o.create_task(wl.task_doc(j))
        t0 = time.perf_counter()
        o.tick()
        d2 = time.perf_counter() - t0
        done, total_dt = sample, d2   # steady-state slice (cluster already partly filled, like the GPU pass on average)
    return {"value": done / total_dt, "unit": "placements/s", "cores": 1, "kind": "port",
            "pair_evals_per_s": done * wl.N / total_dt,
            "sample": f"{done} consecutive one-off tasks of the same workload against all {wl.N} nodes "
                      f"(oracle tick() wall time {total_dt:.2f} s, after a {probe}-task warm-in)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--order", default="rr", choices=["rr", "major"], help="task order: round-robin over services (SURVEY 8d) or service-major")
    ap.add_argument("--rounds", type=int, default=20, help="churn rounds (--mode churn; BASELINE configs[4] uses 100)")
    ap.add_argument("--mode", default="one-off", choices=["one-off", "grouped", "enforce", "churn"],
                    help="one-off (headline, SURVEY 8d primary mode); grouped: S groups of T/S tasks through swp_schedule_groups (secondary mode); "
                         "enforce: the constraint enforcer's start-up sweep (SURVEY 8f-1) over the cluster the placement produced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus and world_env == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    import torch
    from swarmkit_amd import dist as swdist
    ranks = swdist.Ranks(backend="nccl" if world_env > 1 else None)   # nccl == RCCL on ROCm
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world

    from swarmkit_amd import abi, host, synth

    # Multi-GPU (round 1): independent replicas — every rank schedules its own cluster of the same shape
    # (seed offset by rank); no data-path collective. The node-sharded scan with an RCCL exchange is the
    # next row of SURVEY.md §8e.
    wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, seed=ranks.replica_seed(0x5EED0000), order=args.order)
    eng = abi.Engine(device=local_rank, window=args.window, profile=True)
    sched = host.HostScheduler(engine=eng)
    t0 = time.perf_counter()
    descs = host.load_workload(sched, wl)
    t_host_prep = time.perf_counter() - t0
    if args.mode == "churn":
        # BASELINE configs[4] / SURVEY 8d cfg5: place the batch once, then rounds of {reactivate the previous round's
        # drained nodes, drain a seeded random 10 % of the nodes, remove the tasks on them, re-place as many new tasks}.
        # Exercises the incremental path (swp_node_update_dynamic, swp_commit(remove), swp_schedule_batch), timed end to end.
        import numpy as np
        out, _h = eng.schedule_batch(descs, want_hist=False)
        assign = out.astype(np.int64).copy()          # task -> node (or -1)
        svc_of = np.array([wl.task_service(j) for j in range(wl.T)])
        rng = np.random.default_rng(wl.seed)
        prev = np.zeros(0, dtype=np.int64)
        replaced = 0
        t_rounds = []
        for rnd in range(args.rounds):
            t0 = time.perf_counter()
            for n in prev:                             # reactivate
                row = eng.node_get(int(n))
                eng.node_update_dynamic(int(n), row.flags | abi.NODE_READY, row.cpu, row.mem, row.total)
            drained = rng.choice(wl.N, size=max(wl.N // 10, 1), replace=False)
            for n in drained:                          # Availability = DRAIN
                row = eng.node_get(int(n))
                eng.node_update_dynamic(int(n), row.flags & ~abi.NODE_READY, row.cpu, row.mem, row.total)
            gone = np.nonzero(np.isin(assign, drained))[0]
            if len(gone):
                pl = np.zeros(len(gone), dtype=abi.PLACEMENT_DTYPE)
                pl["node"], pl["service"] = assign[gone], descs["service"][gone]
                pl["cpu"], pl["mem"], pl["counted"] = descs["cpu"][gone], descs["mem"][gone], 1
                eng.commit(pl, add=False)              # NodeInfo.removeTask for every task on a drained node
                new_out, _h = eng.schedule_batch(descs[gone], want_hist=False)   # as many new tasks of the same services
                assign[gone] = new_out
                replaced += len(gone)
            prev = drained
            t_rounds.append(time.perf_counter() - t0)
        tt = sum(t_rounds)
        print(json.dumps({"metric": "reschedule churn: placements/sec over rounds of {drain 10 % of the nodes, remove their tasks, re-place} (end to end)",
                          "value": replaced / tt if tt else 0.0, "unit": "placements/s", "n_gpus": 1, "steps": args.rounds, "warmup": 0,
                          "ms_per_step": 1e3 * tt / max(args.rounds, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "int64", "data": "synthetic",
                          "config": dict(wl.describe(), mode="churn", rounds=args.rounds, replaced=int(replaced)),
                          "still_placed": int((assign >= 0).sum())}))
        ranks.close()
        return
    if args.mode == "enforce":
        # constraintenforcer.rejectNoncompliantTasks for EVERY node (the enforcer's start-up sweep,
        # constraint_enforcer.go:45-52) after the batch has been placed: tasks-on-node x (current service constraints +
        # reservations against Description.Resources). Timed end to end around swp_enforce.
        import numpy as np
        out, _h = eng.schedule_batch(descs, want_hist=False)
        placed = np.nonzero(out >= 0)[0]
        order = placed[np.lexsort((placed, out[placed]))]          # by node, then task id (= canonical store order)
        node_of = out[order]
        svc_of = np.array([wl.task_service(int(j)) for j in order])
        cset = descs["constraint_set"][order]
        trec = np.zeros(len(order), dtype=abi.ENF_TASK_DTYPE)
        trec["cpu"], trec["mem"], trec["constraint_set"] = wl.svc_cpu[svc_of], wl.svc_mem[svc_of], cset
        trec["flags"], trec["desired_state"], trec["state"] = abi.ENF_RESERVATIONS, 512, 512
        firsts = np.searchsorted(node_of, np.arange(wl.N), side="left")
        counts = np.searchsorted(node_of, np.arange(wl.N), side="right") - firsts
        nrec = np.zeros(wl.N, dtype=abi.ENF_NODE_DTYPE)
        nrec["node"], nrec["first_task"], nrec["n_tasks"] = np.arange(wl.N), firsts, counts
        nrec["cpu"], nrec["mem"] = wl.node_cpu, wl.node_mem
        for _ in range(args.warmup):
            rej = eng.enforce(nrec, trec)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rej = eng.enforce(nrec, trec)
        t_step = (time.perf_counter() - t0) / max(args.steps, 1)
        res = {"metric": "constraint-enforcer sweep: (task, node) compliance checks/sec, end to end through swp_enforce",
               "value": len(order) / t_step, "unit": "task checks/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
               "data": "synthetic", "config": dict(wl.describe(), mode="enforce", tasks_checked=int(len(order)), nodes_swept=int(wl.N)),
               "rejected": int(rej.sum())}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc
            sample = min(wl.N, 400)
            docs = {}
            for idx, j in enumerate(order):
                n = int(node_of[idx])
                if n < sample:
                    docs.setdefault(n, []).append(dict(wl.task_doc(int(j)), NodeID=wl.node_id(n), DesiredState=512, Status={"State": 512}))
            t0 = time.perf_counter()
            nt = 0
            for n in range(sample):
                nt += len(docs.get(n, ()))
                orc.enforce(wl.node_doc(n), docs.get(n, []), {})
            dt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": nt / dt, "unit": "task checks/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle enforce_node on the first {sample} nodes ({nt} tasks, {dt:.2f} s incl. JSON marshalling)"}
        print(json.dumps(res))
        ranks.close()
        return
    if args.mode == "grouped":
        # secondary mode: every service is ONE group of T/S identical tasks -> S scans instead of T.
        # Timed end to end around swp_schedule_groups (descriptor upload, k_groups, results back).
        import numpy as np
        per_service = descs[:wl.S] if wl.order == "rr" else descs[::max(wl.T // wl.S, 1)][:wl.S]
        sizes = np.bincount(np.array([wl.task_service(j) for j in range(wl.T)]), minlength=wl.S).astype(np.uint32)
        eng.state_save()
        for _ in range(args.warmup):
            eng.state_restore()
            eng.schedule_groups(per_service, sizes)
        torch.cuda.synchronize(); ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.state_restore()
            out, _h = eng.schedule_groups(per_service, sizes)
        torch.cuda.synchronize(); ranks.barrier()
        t_step = ranks.max_over_ranks(time.perf_counter() - t0) / max(args.steps, 1)
        if rank == 0:
            print(json.dumps({"metric": "task placements/sec, grouped mode (S groups of T/S tasks, end to end through swp_schedule_groups)",
                              "value": world * wl.T / t_step, "unit": "placements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
                              "data": "synthetic", "config": dict(wl.describe(), mode="grouped", groups=int(wl.S)),
                              "pair_evals_per_s": world * wl.S * wl.N / t_step, "placed": int((out >= 0).sum())}))
        ranks.close()
        return
    t0 = time.perf_counter()
    batch = eng.batch_prepare(descs)
    t_prepare = time.perf_counter() - t0
    eng.state_save()

    def sync():
        torch.cuda.synchronize()
        ranks.barrier()
        torch.cuda.synchronize()

    def step():
        eng.state_restore()
        batch.run()
        return batch.results(want_hist=False)

    for _ in range(args.warmup):
        step()
    sync()
    ms_scan = ms_resolve = ms_explain = ms_classes = ms_dev = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, _ = step()
        st = eng.stats()
        ms_scan += st["ms_scan"]
        ms_resolve += st["ms_resolve"]
        ms_explain += st["ms_explain"]
        ms_classes += st["ms_classes"]
        ms_dev += st["ms_total"]
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = ranks.max_over_ranks(elapsed)

    st = eng.stats()
    K = max(args.steps, 1)
    windows = st["last_windows"]
    placed = int((out >= 0).sum())
    pairs = wl.T * wl.N
    row_b = ROW_B.get(args.workload, 48)
    alg_bytes_step = pairs * row_b + wl.T * TASK_B
    t_step = elapsed / K
    # dominant kernel = k_resolve3 (sequential argmin + residual-update commit): one launch per window
    res_launch_ms = ms_resolve / K / max(windows, 1)
    alg_bytes_launch = alg_bytes_step / max(windows, 1)
    achieved = alg_bytes_launch / (res_launch_ms * 1e-3) / 1e9 if res_launch_ms > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get("k_resolve_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": f"task placements/sec ({wl.T // 1000}k one-off tasks x {wl.N // 1000}k nodes, "
                  + {"cfg2": "Resource filter", "cfg3": "Resource+Constraint+Platform filters", "cfg4": "Resource+Constraint+Platform+HostPort+MaxReplicas+Plugin filters"}.get(args.workload, args.workload)
                  + ", spread)",
        "value": world * wl.T / t_step,
        "unit": "placements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": t_step * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": dict(wl.describe(), parallelism="replicas" if world > 1 else "single", control_backend=ranks.backend, window=int(st["last_windows"] and -(-wl.T // st["last_windows"])),
                       static_classes=st["last_static_classes"]),
        "pair_evals_per_s": world * pairs / t_step,
        "placed": placed,
        "unplaceable": wl.T - placed,
        "roofline": {"bound": "hbm", "kernel": {105: "k_resolve5<exact>", 5: "k_resolve5<scan>", 3: "k_resolve3", 2: "k_resolve2", 1: "k_resolve1", 0: "k_resolve"}.get(int(st.get("last_resolver", 3)), "k_resolve"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": res_launch_ms, "launches_per_step": windows},
        "kernels_ms_per_step": {"classes+init": ms_classes / K, "k_scan": ms_scan / K, "k_resolve": ms_resolve / K,
                                "k_explain": ms_explain / K, "device_total": ms_dev / K},
        "whole_job_algorithmic_GBs": alg_bytes_step / t_step / 1e9,
        "k_scan_algorithmic_GBs": (alg_bytes_step / (ms_scan / K * 1e-3) / 1e9) if ms_scan > 0 else None,
        "host_prep_s": {"intern+descriptors": t_host_prep, "swp_batch_prepare": t_prepare},
        "resolver_raw": {k: st[k] for k in ("generic_tasks", "resolver_spins", "verify_retries", "slow_path_tasks", "rebase_events", "batches")},
        "resolver": {"verify_retries": st["verify_retries"] // max(st["resolve_launches"] // max(windows, 1), 1),
                     "slow_path_tasks": st["slow_path_tasks"] // max(st["resolve_launches"] // max(windows, 1), 1)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only (bounded sample, ~15 s of one host core)
        result["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0:
        print(json.dumps(result))
    batch.free()
    ranks.close()


if __name__ == "__main__":
    main()
