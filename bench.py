#!/usr/bin/env python3
"""bench.py — placements/s of the batch task-placement hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" = one full pass of the hot path over one batch of synthetic pending tasks
(BASELINE.json configs[2]: 100k one-off tasks x 10k nodes, Resource + Constraint + Platform
filters, spread strategy), starting from the same cluster state every step, with the node rows and
task descriptors already resident in HBM. Inside the timed region, per step:
    device state restore (3 small D2D copies) -> class bitmaps -> resolve/commit (k_resolve5: one launch; k_resolve6: rounds)
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
ROW_B = {"cfg2": 32, "cfg3": 48, "cfg3m": 48, "cfg4": 64}   # algorithmic node-row bytes per (task,node) pair, SURVEY.md §8a
TASK_B = 64 + 8                # descriptor + result per task, SURVEY.md §8d


RESOLVER_NAMES = {105: "k_resolve5", 6: "k_resolve6 (one 'launch' = one ROUND: k_r6_propose + k_r6_commit)"}
SHADER_GHZ = 2.4               # MI355X peak engine clock, /opt/skills/guides/MI355X_MICROARCH.md: cycles_per_task is quoted at this clock
FILTERS = {"cfg2": "Resource filter", "cfg3": "Resource+Constraint+Platform filters",
           "cfg4": "Resource+Constraint+Platform+HostPort+MaxReplicas+Plugin filters"}


def emit_line(text):
    """The ONE JSON line of the run, as the LAST thing on stdout: whatever native libraries left in the C runtime's stdout buffer (RCCL
    prints a version banner there when a communicator is created) is flushed first, then the line goes out unbuffered."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    os.write(1, (text + "\n").encode())


def profile_traffic(kernel, run=None):
    """HBM bytes per launch of the dominant kernel(s) from the committed PMC summary (profiles/r06_pmc_summary.json, produced by
    tools/profile_round6.sh on the GPU box: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, corrected 2*FETCH + WRITE) — a
    rocprofv3 --pmc pass cannot run inside the timed bench. `kernel`: one kernel name, "k_resolve6" (one ROUND = k_r6_propose* +
    k_r6_commit*), or a list of kernel names whose per-launch bytes are summed (one round of that mode); `run`: the profiled run
    whose numbers to take (cfg3 / grouped / churn / shards4), default: whichever run saw the kernel first.
    Returns (bytes or None, provenance string)."""
    for tag in ("r06", "r05", "r04", "r03"):   # this round's summary, else the last one that measured the same kernels (its provenance string says which)
        path = os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")
        if not os.path.exists(path):
            continue
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        per = doc.get("hbm_bytes_per_launch", {})
        if run is not None and run in doc.get("runs", {}):
            per = {k.split("::")[-1].split("<")[0]: v["hbm_bytes_per_launch_corrected"] for k, v in doc["runs"][run].items()}
        src = "profiles/%s_pmc_summary.json%s (%s)" % (tag, " run " + run if run else "", doc.get("source", "rocprofv3 --pmc"))
        if isinstance(kernel, (list, tuple)):
            have = [k for k in kernel if k in per]
            if have:
                return sum(per[k] for k in have), " + ".join(have) + " per round, " + src
            continue
        base = kernel.split("<")[0]
        pk = "k_r6_propose_small" if "k_r6_propose_small" in per else "k_r6_propose"   # (the one-chunk instance is what runs up to 32 768 nodes: the headline)
        if kernel.startswith("k_resolve6") and pk in per and "k_r6_commit" in per:   # per ROUND: one launch of each
            return per[pk] + per["k_r6_commit"], "%s + k_r6_commit per round, %s" % (pk, src)
        if base in per:
            return per[base], src
    return None, "no PMC summary for %s under profiles/" % (kernel,)


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


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
    model, nproc = host_cpu()
    return {"value": done / total_dt, "unit": "placements/s", "cores": 1, "kind": "port", "cpu_model": model, "nproc": nproc,
            "pair_evals_per_s": done * wl.N / total_dt,
            "sample": f"{done} consecutive one-off tasks of the same workload against all {wl.N} nodes "
                      f"(oracle tick() wall time {total_dt:.2f} s, after a {probe}-task warm-in)"}


# SWP_BENCH_RANK_PATH=1 (a check, not a measurement mode): a job of ONE rank takes the code paths of a rank of many — DeviceRankShard,
# swp_shard_run_rank with its ncclAllGather, the churn script by owner rank — so that they can be exercised on a one-GPU box.
RANK_PATH = os.environ.get("SWP_BENCH_RANK_PATH") == "1"


def run_sharded(args, ranks, wl, eng, sched, descs, ranges, t_host_prep):
    """Node-range shards: the SAME 100k x 10k (or --workload / --tasks / --nodes) job, the node set split over the
    ranks. Per step: device state restore, then rounds of {k_propose over a block of tasks on every shard, all-gather of the
    proposal records (RCCL between ranks), the same merge on every rank, k_shard_apply on the owners} until the batch is
    placed, then the explain pass. Strong scaling: the job is fixed, the node rows per GPU shrink with N."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from swarmkit_amd import abi, host, shard as swshard
    rank, world = ranks.rank, ranks.world
    firsts = [r[0] for r in ranges]
    engines, batches = [eng], [eng.batch_prepare(descs)]
    if world == 1:   # --shards G: the other engines live on this GPU too
        for g in range(1, len(ranges)):
            e2 = abi.Engine(device=ranks.local_rank, profile=True, shard_rank=g, shard_count=len(ranges))
            s2 = host.HostScheduler(engine=e2)
            engines.append(e2)
            batches.append(e2.batch_prepare(host.load_workload(s2, wl, *ranges[g])))
            batches[-1]._sched = s2
    for e in engines:
        e.state_save()

    device_rounds = not args.host_merge and os.environ.get("SWP_SHARD_EXCHANGE") != "host"   # the rounds stay on the device: swp_shard_run (one process) / swp_shard_run_rank (RCCL between ranks)

    state = {"host_merge": bool(args.host_merge) or os.environ.get("SWP_SHARD_EXCHANGE") == "host"}

    def make_driver():
        if (world > 1 or RANK_PATH) and not state["host_merge"]:   # rounds on the device, an ncclAllGather of the block's proposals per round (swp_shard_run_rank)
            try:
                return swshard.DeviceRankShard(batches[0], rank, world, ranges, dist, ranks.device, fold=False)
            except swshard.RcclUnavailable as exc:   # raised on EVERY rank (the bootstrap agrees on its outcome): all take the same way out
                state["host_merge"] = True
                state["exchange_fallback"] = True
                if rank == 0:
                    print("bench: RCCL inside libswp.so is not usable here (%s): the proposals are exchanged through torch.distributed and merged on the host" % (exc,), file=sys.stderr)
                if args.strict:
                    raise SystemExit("bench --strict: the device-rounds exchange is not available (%s)" % (exc,))
        if world > 1:
            return swshard.RankShard(batches[0], rank, world, firsts, dist, ranks.device)   # cuda:<local rank> under RCCL, cpu if it fell back to gloo
        return swshard.DeviceShardGroup(batches, firsts, fold=False) if device_rounds else swshard.ShardGroup(batches, firsts)

    def sync():
        torch.cuda.synchronize()
        ranks.barrier()
        torch.cuda.synchronize()

    def step():
        for e in engines:
            e.state_restore()
        drv = make_driver()
        out, _ = drv.run(want_hist=False)
        return out, drv.rounds

    for _ in range(args.warmup):
        step()
    sync()
    ms_prop = ms_apply = 0.0
    n_launch = n_ptasks = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, rounds = step()
        st = engines[0].stats()
        ms_prop += st["ms_propose"]
        ms_apply += st["ms_apply"]
        n_launch += st["propose_launches"]
        n_ptasks += st["propose_tasks"]
    sync()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    device_rounds = device_rounds and not state["host_merge"]   # (what really ran: the labels below follow it)
    K = max(args.steps, 1)
    t_step = elapsed / K
    placed = int((out >= 0).sum())
    row_b = ROW_B.get(args.workload, 48)
    n_local = ranges[rank if world > 1 else 0][1]
    # dominant kernel of this path = k_propose (this rank's launches): tasks proposed x this shard's nodes x row bytes
    alg_launch = (n_ptasks / max(n_launch, 1)) * n_local * row_b + (n_ptasks / max(n_launch, 1)) * TASK_B
    launch_ms = ms_prop / max(n_launch, 1)
    achieved = alg_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    traffic, traffic_src = profile_traffic("k_propose")
    note = "rank 0's shard; a task cut off a block is proposed again, so the tasks proposed per step exceed the batch (%.2fx)" % (n_ptasks / K / max(wl.T, 1))
    kernels_ms = {"k_propose": ms_prop / K, "k_shard_apply": ms_apply / K}
    if device_rounds:   # no per-kernel events on this path: the step time over the rounds is what there is
        launch_ms, achieved, n_launch = t_step * 1e3 / max(rounds, 1), 0.0, rounds * K
        alg_launch = (wl.T / max(rounds, 1)) * wl.N * row_b + (wl.T / max(rounds, 1)) * TASK_B
        achieved = alg_launch / (launch_ms * 1e-3) / 1e9
        # (the PMC passes of tools/profile_round5.sh ran THIS shape: cfg4 200k x 40k over 4 shards — other shapes carry no traffic figure)
        profiled = args.workload == "cfg4" and wl.T == 200000 and wl.N == 40000 and len(ranges) == 4 and world == 1
        traffic, traffic_src = profile_traffic(["k_r7_propose", "k_r7_propose_small", "k_r7_commit"], run="shards4") if profiled else (None, "no PMC pass for this shape (tools/profile_round5.sh profiles cfg4 200k x 40k over 4 shards)")
        note = ("a 'launch' is one ROUND of the whole job: every shard's k_r7_propose, the exchange of the proposals, every shard's k_r7_commit (fold + match + apply); "
                "its time is the step time over the rounds (host gaps included), its bytes are the round's share of the batch's algorithmic bytes over ALL shards")
        kernels_ms = {"one round (k_r7_propose + exchange + k_r7_commit), wall": t_step * 1e3 / max(rounds, 1)}
    result = {
        "metric": f"task placements/sec ({wl.T // 1000}k one-off tasks x {wl.N // 1000}k nodes, " + FILTERS.get(args.workload, args.workload) + ", spread)",
        "value": wl.T / t_step, "unit": "placements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": dict(wl.describe(), parallelism="node-shard", shards=len(ranges), engines_per_gpu=1 if world > 1 else len(ranges),
                       exchange=("rounds on the device (%s): block-resolver proposals per shard, every shard folds them while it stages its lists, matches the same block and applies the picks in its own range (two launches per round and device)" % ("swp_shard_run_rank: an ncclAllGather of the proposals per round" if world > 1 else "swp_shard_run: peer reads")
                                 if device_rounds else "all_gather of %d-byte proposal records per task and shard (%s)" % (abi.PROPOSAL_DTYPE.itemsize, "RCCL" if world > 1 else "host arrays, one process")),
                       control_backend=ranks.backend, block=512 if device_rounds else swshard.BLOCK, rounds_per_step=rounds, tasks_decided_per_round=wl.T / max(rounds, 1)),
        "pair_evals_per_s": wl.T * wl.N / t_step, "placed": placed, "unplaceable": wl.T - placed,
        # (a fraction above 1 is no roofline: the rounds read bitmap rows, not a node row per pair — then the line says so instead)
        "roofline": {"bound": "hbm", "kernel": "one sharded round (k_r7_propose + k_r7_commit, one workgroup per shard)" if device_rounds else "k_propose",
                     # (as on the single-engine line: where algorithmic bytes / time passes the peak, the MEASURED HBM rate is what is reported)
                     "achieved": (achieved if achieved <= HBM_PEAK_GBS else (traffic / (launch_ms * 1e-3) / 1e9 if traffic and launch_ms > 0 else None)), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS if achieved <= HBM_PEAK_GBS else (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and launch_ms > 0 else None)),
                     "not_hbm_bound": achieved > HBM_PEAK_GBS, "algorithmic_GBs": achieved,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_launch, "avg_launch_ms": launch_ms,
                     "launches_per_step": n_launch / K, "note": note},
        "kernels_ms_per_step": kernels_ms,
        "host_prep_s": {"intern+descriptors": t_host_prep},
        # True: the RCCL path inside libswp.so could not be used and the ranks fell back to the host-merged round-2 protocol (an order
        # of magnitude slower: NOT the design's number). bench.py --strict exits non-zero instead of measuring that.
        "exchange_fallback": bool(state.get("exchange_fallback")),
        # the node-range split exists for node sets beyond one engine's LDS budget (~650k nodes). Below that a sharded round
        # pays a propose + a commit launch per device plus the exchange for work one engine does in the same two launches: the 1 -> N
        # curve points DOWN at every size one GPU holds.
        "expected_speedup": "<1 below ~650k nodes",
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0:
        emit_line(json.dumps(result))
    for b in batches:
        b.free()
    ranks.close()


def run_churn(args, ranks, wl, eng, descs, ranges):
    """BASELINE configs[4] / SURVEY 8d cfg5: place the batch once, then rounds of {reactivate the previous round's drained nodes, drain a
    seeded random 10 % of the nodes, remove the tasks on them, re-place as many new tasks}. The incremental path (scheduler.go:254-396,
    nodeinfo.go:66-154): swp_node_update_dynamic_many, swp_commit(remove), a re-placement batch — timed end to end, round by round.

      one engine            the calls as they are;
      --shards G (N = 1)    `eng` is a shard SET (swp_shardset_create): the same calls, routed to the owner of every node's range,
                            the re-placement batch by swp_shard_run;
      --gpus N   (N > 1)    one engine per rank holding ITS node range: every rank follows the same script, applies the drains and
                            swp_commit(remove) for its own nodes, and takes part in one sharded re-placement batch per round
                            (swp_shard_run_rank: an ncclAllGather of the block's proposals per round of the resolver)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from swarmkit_amd import abi, shard as swshard
    rank, world = ranks.rank, ranks.world
    by_rank = world > 1 or RANK_PATH
    first, cnt = (ranges[rank] if by_rank else (0, wl.N))
    state = {"rounds_total": 0}

    # the descriptors as (templates, template per task) — tasks of one service spec share theirs, which a caller knows (the shim keeps one
    # per (service, spec version)): the re-placement batches are prepared by swp_batch_prepare_templates, an array lookup per task
    tmpl, tmpl_of = np.unique(descs, return_inverse=True)
    tmpl_of = tmpl_of.astype(np.uint32)

    def place(which, timed=None):
        """one placement batch (the tasks `which` of the workload; None: all) over the whole node set -> (global node per task, device ms or None)"""
        of = tmpl_of if which is None else tmpl_of[which]
        if not by_rank:
            ta = time.perf_counter()
            bt = eng.batch_prepare_templates(tmpl, of)
            tb = time.perf_counter()
            bt.run()
            tc = time.perf_counter()
            out, _h = bt.fetch(want_hist=False)   # (fetch: the placements enter the engine's node mirror, as in swp_schedule_batch)
            bt.free()
            td = time.perf_counter()
            st = eng.stats()
            if st["last_resolver"] == 7:   # (the counter is cumulative: the rounds of THIS batch)
                state["rounds_total"] += st["resolve_launches"] - state.get("launches_seen", 0)
            state["launches_seen"] = st["resolve_launches"]
            if timed is not None:
                timed["swp_batch_prepare_templates"] += tb - ta
                timed["swp_batch_run"] += tc - tb
                timed["swp_batch_fetch"] += td - tc
            return out.astype(np.int64), st["ms_total"]
        ta = time.perf_counter()
        bt = eng.batch_prepare_templates(tmpl, of)
        tb = time.perf_counter()
        drv = swshard.DeviceRankShard(bt, rank, world, ranges, dist, ranks.device, fold=True)   # raises RcclUnavailable on EVERY rank: no silent fallback
        out, _h = drv.run(want_hist=False)
        tc = time.perf_counter()
        bt.free()
        state["rounds_total"] += drv.rounds
        if timed is not None:
            timed["swp_batch_prepare_templates"] += tb - ta
            timed["swp_batch_run"] += tc - tb
        return out.astype(np.int64), (tc - tb) * 1e3

    assign, _ms = place(None)           # task -> global node (or -1)
    assign = assign.copy()
    rng = np.random.default_rng(wl.seed)
    prev = np.zeros(0, dtype=np.int64)
    replaced = 0
    dev_ms = 0.0
    t_rounds = []
    phases = {"node calls (get_many + update_dynamic_many)": 0.0, "swp_commit(remove)": 0.0, "swp_batch_prepare_templates": 0.0, "swp_batch_run": 0.0,
              "swp_batch_fetch": 0.0, "the script itself (which tasks sat on the drained nodes, their descriptors)": 0.0}
    is_drained = np.zeros(wl.N + 1, dtype=bool)   # (index -1 = unplaced: the extra last entry)
    state["rounds_total"] = 0
    if by_rank:
        torch.cuda.synchronize()
        ranks.barrier()
    for rnd in range(args.rounds):
        t0 = time.perf_counter()
        drained = rng.choice(wl.N, size=max(wl.N // 10, 1), replace=False)
        touched = np.concatenate([prev, drained]).astype(np.int64)
        reactivate = np.arange(len(touched)) < len(prev)
        if by_rank:   # the rows of THIS rank's range, by local index
            mine = (touched >= first) & (touched < first + cnt)
            touched, reactivate = touched[mine] - first, reactivate[mine]
        ta = time.perf_counter()
        rows = eng.node_get_many(touched.astype(np.uint32))          # two calls per round instead of four per node
        upd = np.zeros(len(touched), dtype=abi.NODE_DYNAMIC_DTYPE)
        upd["node"], upd["cpu"], upd["mem"], upd["total"] = touched, rows["cpu"], rows["mem"], rows["total"]
        upd["flags"] = np.where(reactivate, rows["flags"] | abi.NODE_READY, rows["flags"] & ~np.uint32(abi.NODE_READY))
        eng.node_update_dynamic_many(upd)          # reactivate the previous round's nodes, Availability = DRAIN for this round's
        tb = time.perf_counter()
        is_drained[:] = False
        is_drained[drained] = True
        gone = np.nonzero(is_drained[assign])[0]
        tc = td = tg = tb
        if len(gone):
            own = gone if not by_rank else gone[(assign[gone] >= first) & (assign[gone] < first + cnt)]
            pl = np.zeros(len(own), dtype=abi.PLACEMENT_DTYPE)
            pl["node"], pl["service"] = assign[own] - first, descs["service"][own]
            pl["cpu"], pl["mem"], pl["counted"] = descs["cpu"][own], descs["mem"][own], 1
            tc = time.perf_counter()               # (as many new tasks of the same services: the same descriptors)
            if len(own):
                eng.commit(pl, add=False)          # NodeInfo.removeTask for every task on a drained node (on its owner)
            td = time.perf_counter()
            new_out, ms = place(gone, phases)
            tg = time.perf_counter()
            assign[gone] = new_out
            replaced += len(gone)
            dev_ms += ms
            phases["swp_commit(remove)"] += td - tc
        prev = drained
        t1 = time.perf_counter()
        t_rounds.append(t1 - t0)
        phases["node calls (get_many + update_dynamic_many)"] += tb - ta
        phases["the script itself (which tasks sat on the drained nodes, their descriptors)"] += (ta - t0) + (tc - tb) + ((t1 - tg) if len(gone) else 0.0)
    if by_rank:
        torch.cuda.synchronize()
        ranks.barrier()
    tt = ranks.max_over_ranks(sum(t_rounds)) if by_rank else sum(t_rounds)
    R = max(args.rounds, 1)
    row_b = ROW_B.get(args.workload, 48)
    alg = (replaced / R) * wl.N * row_b + (replaced / R) * TASK_B
    n_shards = world if by_rank else (eng.shards or 1)
    churn_traffic = (profile_traffic(["k_r7_propose", "k_r7_propose_small", "k_r7_commit"], run="churn_shards4") if n_shards > 1
                     else profile_traffic(["k_r6_propose_small_c", "k_r6_commit_c"], run="churn"))   # (the compact index of the next round is built at the end of k_r6_commit_c since round 6: its bytes are that kernel's)
    if by_rank:
        par, exch = "node-shard", "one engine per rank over its node range; per round of the resolver an ncclAllGather of the block's proposals (swp_shard_run_rank); drains and swp_commit(remove) on the owner rank"
    elif n_shards > 1:
        par, exch = "node-shard", "a shard SET of %d engines on one GPU behind one handle (swp_shardset_create): node calls and swp_commit(remove) routed to the owner, the re-placement batch by swp_shard_run" % n_shards
    else:
        par, exch = "single", None
    res = {"metric": "reschedule churn: placements/sec over rounds of {drain 10 % of the nodes, remove their tasks, re-place} (end to end)",
           "value": replaced / tt if tt else 0.0, "unit": "placements/s", "n_gpus": world, "steps": args.rounds, "warmup": 0,
           "ms_per_step": 1e3 * tt / R, "higher_is_better": True, "scaling": "strong" if n_shards > 1 else "weak", "vs_baseline": None,
           "dtype": "int64", "data": "synthetic",
           "config": dict(wl.describe(), mode="churn", rounds=args.rounds, replaced=int(replaced), parallelism=par, shards=n_shards),
           "still_placed": int((assign >= 0).sum()), "device_ms_per_round": dev_ms / R,
           "ms_per_round_by_phase": {k: 1e3 * v / R for k, v in phases.items()},
           "roofline": {"bound": "hbm", "kernel": "the round's re-placement batch (k_r6_propose + k_r6_commit rounds, or the sharded rounds k_r7_*) + explain",
                        "achieved": alg / (dev_ms / R * 1e-3) / 1e9 if dev_ms else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (alg / (dev_ms / R * 1e-3) / 1e9 / HBM_PEAK_GBS) if dev_ms else 0.0,
                        "traffic": churn_traffic[0], "traffic_source": "per resolver ROUND, " + churn_traffic[1],
                        "algorithmic_bytes_per_launch": alg, "avg_launch_ms": dev_ms / R,
                        "note": "device time of the re-placement batch of a round (engine events; over shards: the wall time of the sharded run); the round "
                                "itself also pays the host side: two bulk node calls, swp_commit(remove) and swp_batch_prepare_templates for ~9k tasks"}}
    if n_shards > 1:
        res["config"]["exchange"] = exch
        res["config"]["resolver_rounds_per_churn_round"] = state["rounds_total"] / R
        # (a sharded round costs a propose + a commit launch per device plus the exchange, against the same two launches on one
        # engine: below the node count one engine's LDS holds, sharding cannot be faster)
        res["expected_speedup"] = "<1 below ~650k nodes: the node-range split exists for node sets one engine cannot hold"
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(wl, budget_s=8.0)
    if rank == 0:
        emit_line(json.dumps(res))
    ranks.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--services", type=int, default=None, help="number of services of the synthetic workload (default T/100); 1 = the reference's benchScheduler shape: every task of ONE service")
    ap.add_argument("--order", default="rr", choices=["rr", "major"], help="task order: round-robin over services (SURVEY 8d) or service-major")
    ap.add_argument("--rounds", type=int, default=20, help="churn rounds (--mode churn; BASELINE configs[4] uses 100)")
    ap.add_argument("--mode", default="one-off", choices=["one-off", "grouped", "enforce", "churn"],
                    help="one-off (headline, SURVEY 8d primary mode); grouped: S groups of T/S tasks through swp_schedule_groups (secondary mode); "
                         "enforce: the constraint enforcer's start-up sweep (SURVEY 8f-1) over the cluster the placement produced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallelism", default="auto", choices=["auto", "single", "replicas", "node-shard"],
                    help="auto: one engine at N=1; at N>1 the node set is sharded over the ranks (SURVEY 8e: contiguous node ranges, RCCL "
                         "all-gather of the block's proposals, same merge on every rank). replicas: N independent clusters (no collective).")
    ap.add_argument("--shards", type=int, default=0, help="N=1 only: run the node-shard protocol over this many engines on the one GPU")
    ap.add_argument("--strict", action="store_true", help="--gpus N > 1: exit non-zero instead of falling back to the host-merged exchange when RCCL inside libswp.so is unusable")
    ap.add_argument("--host-merge", action="store_true", help="--shards: the round-2 protocol (proposals merged on the host) instead of the rounds on the device")
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

    if args.workload == "refbench":
        # The reference's own benchmark, benchScheduler (manager/scheduler/scheduler_test.go:3375-3465, sizes :3335-3373): tasks
        # WITHOUT ServiceID / SpecVersion — one service "" for every task — on nodes with an empty Engine description, every third
        # one advertising the Network plugin; timed around Scheduler.tick() of the C++ host layer (JSON decisions included), a
        # fresh scheduler per step as the Go benchmark builds a fresh store per iteration.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bigcases
        n_nodes, n_tasks = args.nodes or 1000, args.tasks or 100000
        ticks, placed = [], 0
        for it in range(args.warmup + args.steps):
            s = host.HostScheduler(engine=abi.Engine(device=local_rank, profile=True))
            for i in range(n_nodes):
                s.create_node(bigcases.ref_node(i))
            s.set_service("")
            for i in range(n_tasks):
                s.create_task(bigcases.ref_task(i, False))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec = s.tick()
            dt = time.perf_counter() - t0
            if it >= args.warmup:
                ticks.append(dt)
                placed = sum(1 for d in dec if d["NodeID"])
            st = s.e.stats()
        t_step = sum(ticks) / max(len(ticks), 1)
        emit_line(json.dumps({"metric": f"task placements/sec, the reference's benchScheduler shape ({n_tasks // 1000}k tasks of one service x {n_nodes} nodes), end to end through Scheduler.tick()",
                          "value": n_tasks / t_step, "unit": "placements/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                          "config": {"workload": "refbench", "tasks": n_tasks, "nodes": n_nodes, "services": 1, "path": "k_waterfill" if st["waterfill_tasks"] else "resolver"},
                          "placed": placed, "device_ms": st["ms_total"], "waterfill_tasks_total": st["waterfill_tasks"],
                          "note": "tick() = grouping + descriptors + one swp_schedule_batch + 100k decision documents as JSON; device_ms is the engine's share"}))
        ranks.close()
        return
    par = args.parallelism
    if par == "auto":
        par = "node-shard" if (world > 1 or args.shards > 1) else "single"
    if par == "node-shard" and args.mode not in ("one-off", "churn", "grouped"):
        # (the enforcer's sweep over node ranges goes through a shard SET in one process — tests/test_engine_shardset.py;
        # this harness measures it on one engine. Refuse rather than measure N independent replicas under a sharded label.)
        print(f"bench.py: --mode {args.mode} is not run over node-range shards here; use --parallelism replicas (or single)", file=sys.stderr)
        sys.exit(2)
    if par == "node-shard" and args.mode == "grouped" and not (world > 1 or RANK_PATH):
        print("bench.py: --mode grouped is not run over node-range shards of ONE rank here: a shard SET places groups on its union engine (tests/test_engine_shardset.py); "
              "this harness runs the RANK path (--gpus N, or SWP_BENCH_RANK_PATH=1)", file=sys.stderr)
        sys.exit(2)
    shard_mode = par == "node-shard"
    churn_set = shard_mode and args.mode == "churn" and world == 1 and not RANK_PATH   # --shards G: ONE handle over G engines on this GPU (swp_shardset_create)
    if churn_set:
        from swarmkit_amd import shard as swshard
        wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, order=args.order, services=args.services)
        shard_ranges_ = swshard.shard_ranges(wl.N, args.shards)
        eng = abi.Engine(device=local_rank, profile=True, shards=args.shards, nodes_per_shard=shard_ranges_[0][1])
        sched = host.HostScheduler(engine=eng)
        t0 = time.perf_counter()
        descs = host.load_workload(sched, wl)
        t_host_prep = time.perf_counter() - t0
    elif shard_mode:
        # every rank sees the SAME cluster and task list and owns one contiguous range of the canonical node order
        from swarmkit_amd import shard as swshard
        wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, order=args.order, services=args.services)
        n_shards = world if world > 1 else max(args.shards, 1)
        shard_ranges_ = swshard.shard_ranges(wl.N, n_shards)
        my = rank if world > 1 else 0
        eng = abi.Engine(device=local_rank, profile=True, shard_rank=my, shard_count=n_shards)
        sched = host.HostScheduler(engine=eng)
        t0 = time.perf_counter()
        descs = host.load_workload(sched, wl, *shard_ranges_[my])
        t_host_prep = time.perf_counter() - t0
    else:
        # replicas: every rank schedules its own cluster of the same shape (seed offset by rank); no data-path collective
        shard_ranges_ = None
        wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, seed=ranks.replica_seed(0x5EED0000), order=args.order, services=args.services)
        eng = abi.Engine(device=local_rank, profile=True)
        sched = host.HostScheduler(engine=eng)
        t0 = time.perf_counter()
        descs = host.load_workload(sched, wl)
        t_host_prep = time.perf_counter() - t0
    if args.mode == "churn":
        run_churn(args, ranks, wl, eng, descs, shard_ranges_)
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
        emit_line(json.dumps(res))
        ranks.close()
        return
    if args.mode == "grouped":
        # secondary mode: every service is ONE group of T/S identical tasks -> S scans instead of T.
        # Timed end to end around swp_schedule_groups (descriptor upload, k_groups, results back).
        import numpy as np
        per_service = descs[:wl.S] if wl.order == "rr" else descs[::max(wl.T // wl.S, 1)][:wl.S]
        sizes = np.bincount(np.array([wl.task_service(j) for j in range(wl.T)]), minlength=wl.S).astype(np.uint32)
        by_rank = shard_mode and (world > 1 or RANK_PATH)
        union = None
        if by_rank:
            # Task groups in a job of ranks (swarmkit_amd.shard.RankUnionGroups): every rank's engine holds ITS node range; rank 0 also keeps
            # the UNION engine (every node, global indices), places the groups there with k_groups2, ONE broadcast carries the placements,
            # every owner books its share. Groups are capacity-bound by one GPU; what the ranks add is that the one-off batches around a
            # grouped tick stay sharded.
            from swarmkit_amd import shard as swshard
            if rank == 0:
                union = abi.Engine(device=local_rank, profile=True)
                descs_u = host.load_workload(host.HostScheduler(engine=union), wl)
                assert np.array_equal(descs_u, descs), "the union engine and the rank's engine name services and predicate sets differently"
                union.state_save()
            my = rank if world > 1 else 0
            ru = swshard.RankUnionGroups(eng, union, my, world if world > 1 else 1, [r[0] for r in shard_ranges_], [r[1] for r in shard_ranges_], ranks.dist, ranks.device)
        eng.state_save()

        def one_tick():
            eng.state_restore()
            if union is not None:
                union.state_restore()
            return ru.schedule_groups(per_service, sizes) if by_rank else eng.schedule_groups(per_service, sizes)
        for _ in range(args.warmup):
            one_tick()
        torch.cuda.synchronize(); ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, _h = one_tick()
        torch.cuda.synchronize(); ranks.barrier()
        t_step = ranks.max_over_ranks(time.perf_counter() - t0) / max(args.steps, 1)
        if rank == 0:
            row_b = ROW_B.get(args.workload, 48)
            alg = wl.S * wl.N * row_b + wl.T * TASK_B          # pairs = S x N in grouped mode
            res = {"metric": "task placements/sec, grouped mode (S groups of T/S tasks, end to end through swp_schedule_groups)",
                   "value": (1 if by_rank else world) * wl.T / t_step, "unit": "placements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "strong" if by_rank else "weak", "vs_baseline": None, "dtype": "int64",
                   "data": "synthetic", "config": dict(wl.describe(), mode="grouped", groups=int(wl.S),
                                                         parallelism=("ranks: the groups on rank 0's union engine, one broadcast, owners book their share" if by_rank else par)),
                   "pair_evals_per_s": (1 if by_rank else world) * wl.S * wl.N / t_step, "placed": int((out >= 0).sum()),
                   "roofline": {"bound": "hbm", "kernel": "k_groups2", "achieved": alg / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg / t_step / 1e9 / HBM_PEAK_GBS, "traffic": profile_traffic(["k_g2_static", "k_groups2"], run="grouped")[0],
                                "traffic_source": profile_traffic(["k_g2_static", "k_groups2"], run="grouped")[1], "algorithmic_bytes_per_launch": alg,
                                "avg_launch_ms": t_step * 1e3,
                                "note": "one launch = the whole tick (k_g2_static: a wave per static class over the chip, then k_groups2: S groups, one workgroup: a machine wave + 15 helper waves); end-to-end step time (no separate kernel events on this path). "
                                        "The tick is bound by the machine wave's instruction issue while it replays container/heap in the reference's exact order (candidates from the group's static class list, 64 to a chunk; a full heap admits the candidates with its second key by counting them and scatters them by post-order rank, "
                                        "a push that moves is sifted up by the wave, two-key heaps are sorted by the wave; parallel appends / rotation / fill where the keys allow), not by bytes: the fraction says how far from a streaming scan that is"}}
            if world == 1 and not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import orc
                o = orc.Oracle()
                for i in range(wl.N):
                    o.create_node(wl.node_doc(i))
                gsz = max(wl.T // wl.S, 1)
                n_groups = min(wl.S, max(8, int(10.0 / (2.0e-7 * wl.N + 1.0e-6 * gsz))))   # ~10 s of oracle time at ~0.2 us per (group, node)
                for k in range(n_groups):
                    o.set_service(wl.service_id(k))
                wl_g = synth.Workload(args.workload, T=args.tasks, N=args.nodes, order=args.order, services=args.services, grouped=True)
                cnt = 0
                for j in range(wl.T):
                    if wl.task_service(j) < n_groups:
                        o.create_task(wl_g.task_doc(j))
                        cnt += 1
                t0 = time.perf_counter()
                o.tick()
                dt = time.perf_counter() - t0
                model, nproc = host_cpu()
                res["cpu_baseline"] = {"value": cnt / dt, "unit": "placements/s", "cores": 1, "kind": "port", "cpu_model": model, "nproc": nproc,
                                       "sample": f"the first {n_groups} groups of the same workload ({cnt} tasks) against all {wl.N} nodes, oracle tick() {dt:.2f} s"}
            emit_line(json.dumps(res))
        ranks.close()
        return
    if shard_mode:
        run_sharded(args, ranks, wl, eng, sched, descs, shard_ranges_, t_host_prep)
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
    ms_resolve = ms_explain = ms_classes = ms_dev = 0.0
    launches0 = eng.stats()["resolve_launches"]
    t0 = time.perf_counter()
    ms_scan = 0.0
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
    launches1 = eng.stats()["resolve_launches"]   # (before the end-to-end runs below add theirs)

    # end to end, outside the timed region: what a caller of swp_schedule_batch waits for — de-duplication of the predicate
    # sets + H2D of the descriptors (swp_batch_prepare), the device pass, D2H of placements AND explanation histograms
    e2e, prep = [], []
    for _ in range(3):
        eng.state_restore()
        t0 = time.perf_counter()
        b2 = eng.batch_prepare(descs)
        prep.append(time.perf_counter() - t0)
        b2.run()
        b2.results(want_hist=True)
        e2e.append(time.perf_counter() - t0)
        b2.free()
    t_e2e = min(e2e)
    t_prepare_warm = min(prep)   # (the first prepare of the process also pays for the device allocations: reported as "cold")
    # the same for a caller that knows which tasks share a descriptor (the tasks of one service spec: swp_batch_prepare_templates)
    import numpy as np
    tmpl, tmpl_of_task = np.unique(descs, return_inverse=True)
    e2e_t, prep_t = [], []
    for _ in range(3):
        eng.state_restore()
        t0 = time.perf_counter()
        b2 = eng.batch_prepare_templates(tmpl, tmpl_of_task)
        prep_t.append(time.perf_counter() - t0)
        b2.run()
        out_t, _h = b2.results(want_hist=True)
        e2e_t.append(time.perf_counter() - t0)
        b2.free()
    assert (out_t == out).all(), "swp_batch_prepare_templates: placements differ from swp_batch_prepare's"
    # ... and what swp_schedule_batch costs in all: swp_batch_fetch also books the placements in the engine's host-side node mirror
    # (per-node residuals and service counts, the service -> nodes index the next batch's exception lists are read off)
    e2e_f = []
    for _ in range(3):
        eng.state_restore()
        t0 = time.perf_counter()
        b2 = eng.batch_prepare(descs)
        b2.run()
        out_f, _h = b2.fetch(want_hist=True)
        e2e_f.append(time.perf_counter() - t0)
        b2.free()
    assert (out_f == out).all(), "swp_batch_fetch: placements differ from swp_batch_results'"
    eng.state_restore()

    st = eng.stats()
    K = max(args.steps, 1)
    windows = max(int(st["last_windows"]), 1)
    placed = int((out >= 0).sum())
    pairs = wl.T * wl.N
    row_b = ROW_B.get(args.workload, 48)
    alg_bytes_step = pairs * row_b + wl.T * TASK_B
    t_step = elapsed / K
    # dominant kernel = the resolver (sequential argmin + residual-update commit): `windows` launches per step (1 for
    # k_resolve5's exact mode, which needs no scan window)
    kernel = RESOLVER_NAMES.get(int(st.get("last_resolver", 105)), "k_resolve5")
    if kernel.startswith("k_resolve6"):   # the block resolver: a "launch" is one round (two kernel launches), decided tasks per round vary
        windows = max((launches1 - launches0) / 2.0 / K, 1.0)
    res_launch_ms = ms_resolve / K / windows
    alg_bytes_launch = alg_bytes_step / windows
    # A batch whose tasks have no plain candidates (a saturated small cluster) is mostly decided by the scan resolver (k_scanb / k_scan:
    # DESIGN 5b): it is then the dominant kernel, a "launch" is one stretch of tasks, its bytes the stretch's (task, node) pairs
    scan_dominant = kernel.startswith("k_resolve6") and ms_scan > 0.5 * ms_resolve and st["scan_launches"] > 0
    if scan_dominant:
        kernel = "k_scanb / k_scan (the scan resolver: one 'launch' = one STRETCH of tasks without plain candidates, k_scan_fill + k_scan_lists + the scan kernel)"
        windows = float(st["scan_launches"])
        res_launch_ms = ms_scan / K / windows
        alg_bytes_launch = (st["scan_tasks"] * wl.N * row_b + st["scan_tasks"] * TASK_B) / windows
    achieved = alg_bytes_launch / (res_launch_ms * 1e-3) / 1e9 if res_launch_ms > 0 else 0.0
    # (the PMC passes ran the headline shape — cfg3 100k x 10k, round-robin order; a line of another shape carries no traffic figure)
    profiled = args.workload == "cfg3" and wl.T == 100000 and wl.N == 10000 and getattr(wl, "order", "rr") == "rr" and wl.S == 1000
    traffic, traffic_src = profile_traffic(kernel) if profiled else (None, "no PMC pass for this shape (tools/profile_round5.sh profiles the headline: cfg3 100k x 10k)")
    measured_gbs = (traffic / (res_launch_ms * 1e-3) / 1e9) if (traffic and res_launch_ms > 0) else None
    # The algorithmic figure counts a node ROW per (task, node) pair; the resolver reads one BIT per pair and filter from L2-resident
    # rows, so on large node sets "algorithmic bytes / time" passes the HBM peak without the kernel being anywhere near it. A
    # fraction above 1 is not a roofline: such a line reports the measured HBM rate and says what the bound is instead.
    roof_frac = achieved / HBM_PEAK_GBS
    not_hbm = roof_frac > 1.0
    cycles_per_task = ms_resolve / K * 1e-3 * SHADER_GHZ * 1e9 / wl.T

    result = {
        "metric": f"task placements/sec ({wl.T // 1000}k one-off tasks x {wl.N // 1000}k nodes, "
                  + FILTERS.get(args.workload, args.workload) + ", spread)",
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
        "config": dict(wl.describe(), parallelism="replicas (independent clusters, no data-path collective)" if world > 1 else "single", control_backend=ranks.backend,
                       resolver_launches_per_step=windows, static_classes=st["last_static_classes"]),
        "pair_evals_per_s": world * pairs / t_step,
        "placed": placed,
        "unplaceable": wl.T - placed,
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": (measured_gbs if not_hbm else achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ((measured_gbs / HBM_PEAK_GBS) if (not_hbm and measured_gbs is not None) else (None if not_hbm else roof_frac)),
                     "not_hbm_bound": not_hbm, "algorithmic_GBs": achieved,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": res_launch_ms, "launches_per_step": windows, "tasks_per_launch": (st["scan_tasks"] if scan_dominant else wl.T) / windows,
                     "note": ("per STRETCH of the scan resolver (scan_tasks of the batch's tasks in scan_launches stretches; the rest by rounds of the block resolver): "
                              "algorithmic bytes = the stretch's (task, node) pairs x node-row bytes (SURVEY 8d). Everything a task reads is in LDS (node rows, the "
                              "(service, node) matrices, the class rows), so real HBM traffic is the task records and the logs; the kernel is bound by the "
                              "instruction issue of one workgroup (four waves, ~7 800 cycles per batch of four tasks: DESIGN 5b), and a task whose identical twin "
                              "found no node is answered without a look" if scan_dominant else
                              "per ROUND of the block resolver (launches_per_step rounds, tasks_per_launch decided each): algorithmic bytes = the round's "
                              "(task, node) pairs x node-row bytes (SURVEY 8d); traffic = PMC bytes of one k_r6_propose + one k_r6_commit. The resolver reads "
                              "bitmap ROWS (one bit per pair and filter, L2-resident) instead of a node row per pair, so real traffic is a small fraction "
                              "of the algorithmic figure; a round is bound by one wave's instruction issue in k_r6_commit (resolver.cycles_per_task), "
                              "not by HBM" if kernel.startswith("k_resolve6") else
                              "algorithmic bytes = pairs x node-row bytes (SURVEY 8d); the resolver decides from bitmaps held in LDS, so its real HBM "
                              "traffic is far below that (see traffic): the kernel is bound by one workgroup's instruction issue, not by HBM")},
        "kernels_ms_per_step": {"classes+init": ms_classes / K, "k_resolve": ms_resolve / K, "of which scan stretches": ms_scan / K, "k_explain": ms_explain / K, "device_total": ms_dev / K},
        # what really bounds the resolver: the instruction issue of ONE wavefront (the matcher's dependent chain), not bytes
        "issue_bound": {"matcher_instr_per_task": 13, "issue_cycles_per_instr": 5.0, "floor_cycles_per_task": 65.0, "cycles_per_task": cycles_per_task,
                        "note": "the serial chain of the batch is wv::match_seq64 on ONE wave: 13 instructions per task (tools/check_matcher_asm.sh), a lone wave issues one "
                                "every ~5 cycles (tools/mb/mb_issue.hip) -> 65 cycles per task is the floor of this design; cycles_per_task is the resolver's whole device time "
                                "(propose, launch gaps, staging, stops at emptied half-words, the last group's apply) over the tasks, at the peak engine clock"},
        "resolver": {"kernel": kernel.split(" ")[0], "ms_per_step": ms_resolve / K, "cycles_per_task": cycles_per_task,
                     "clock_GHz": SHADER_GHZ, "measured_HBM_GBs": measured_gbs,
                     "note": "cycles of the matching wave's CU per task of the batch, at the peak engine clock; measured_HBM_GBs = PMC bytes per launch (roofline.traffic) / launch time"},
        "whole_job_algorithmic_GBs": alg_bytes_step / t_step / 1e9,
        "end_to_end": {"ms": t_e2e * 1e3, "placements_per_s": wl.T / t_e2e, "with_mirror_fold_ms": min(e2e_f) * 1e3,
                       "includes": "swp_batch_prepare (predicate de-duplication + H2D of the task descriptors) + device pass + D2H of placements and Explain histograms",
                       "swp_batch_prepare_ms": t_prepare_warm * 1e3, "swp_batch_prepare_cold_ms": t_prepare * 1e3,
                       "with_templates": {"ms": min(e2e_t) * 1e3, "swp_batch_prepare_templates_ms": min(prep_t) * 1e3, "templates": int(len(tmpl)),
                                          "note": "the caller names the template (service spec) of every task: no per-task de-duplication pass"}},
        "host_prep_s": {"intern+descriptors": t_host_prep, "swp_batch_prepare": t_prepare},
        "resolver_raw": {k: st[k] for k in ("generic_tasks", "resolver_spins", "slow_path_tasks", "rebase_events", "batches")},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only (bounded sample, ~15 s of one host core)
        result["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0:
        emit_line(json.dumps(result))
    batch.free()
    ranks.close()


if __name__ == "__main__":
    main()
