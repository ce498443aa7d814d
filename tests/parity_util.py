"""Shared helpers: run one synthetic workload through the CPU oracle and through the HIP engine."""
import numpy as np

import orc
from swarmkit_amd import host as swhost


def oracle_run(wl, count=None):
    """Returns ({task id: node id or None}, {task id: Err})."""
    o = orc.Oracle()
    for i in range(wl.N):
        o.create_node(wl.node_doc(i))
    for k in range(wl.S):
        o.set_service(wl.service_id(k))
    n = wl.T if count is None else count
    for j in range(n):
        o.create_task(wl.task_doc(j))
    placed, errs = {}, {}
    for d in o.tick():
        if d["State"] >= orc.ASSIGNED and d["NodeID"]:
            placed[d["ID"]] = d["NodeID"]
        else:
            placed[d["ID"]] = None
            errs[d["ID"]] = d["Err"]
    return placed, errs, o


def engine_run(wl, count=None, **engine_kw):
    """Same workload through libswp.so. Returns (placed, errs, HostScheduler, raw out_node, hist)."""
    s = swhost.HostScheduler(**engine_kw)
    descs = swhost.load_workload(s, wl)
    n = wl.T if count is None else count
    out, hist = s.e.schedule_batch(descs[:n])
    placed, errs = {}, {}
    for j in range(n):
        tid = wl.task_id(j)
        if out[j] >= 0:
            placed[tid] = s.idx_to_id[int(out[j])]
        else:
            placed[tid] = None
            ex = s.explain(hist[j])
            errs[tid] = "no suitable node (" + ex + ")" if ex else "no suitable node"
    return placed, errs, s, out, hist


def assert_same(a_placed, a_errs, b_placed, b_errs):
    assert a_placed.keys() == b_placed.keys()
    diff = [(k, a_placed[k], b_placed[k]) for k in a_placed if a_placed[k] != b_placed[k]]
    assert not diff, f"{len(diff)} placements differ, first: {diff[:5]}"
    ediff = [(k, a_errs[k], b_errs.get(k)) for k in a_errs if a_errs[k] != b_errs.get(k)]
    assert not ediff, f"{len(ediff)} explanations differ, first: {ediff[:3]}"


def sharded_run(wl, n_shards, block=256, mode="host", **engine_kw):
    """The same workload over `n_shards` node-range shards (swarmkit_amd.shard.ShardGroup), every shard an engine of its own
    on this process' device. Returns (placed, errs, rounds) in the oracle's vocabulary."""
    from swarmkit_amd import shard as swshard
    ranges = swshard.shard_ranges(wl.N, n_shards)
    scheds, batches = [], []
    for g, (first, cnt) in enumerate(ranges):
        s = swhost.HostScheduler(shard_rank=g, shard_count=n_shards, **engine_kw)
        descs = swhost.load_workload(s, wl, first, cnt)
        scheds.append(s)
        batches.append(s.e.batch_prepare(descs))
    if mode == "device":   # rounds on the device (swp_shard_run); the block size is the block resolver's (SWP_R6_BLOCK)
        grp = swshard.DeviceShardGroup(batches, [r[0] for r in ranges])
    else:
        grp = swshard.ShardGroup(batches, [r[0] for r in ranges], block=block)
    out, hist = grp.run()
    placed, errs = {}, {}
    for j in range(wl.T):
        tid = wl.task_id(j)
        if out[j] >= 0:
            placed[tid] = wl.node_id(int(out[j]))
        else:
            placed[tid] = None
            ex = scheds[0].explain(hist[j])
            errs[tid] = "no suitable node (" + ex + ")" if ex else "no suitable node"
    for b in batches:
        b.free()
    return placed, errs, grp.rounds
