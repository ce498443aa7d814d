#!/usr/bin/env python3
"""Timing probe of the node-range shard protocol on ONE device (ShardGroup): rounds, accepted tasks per round, wall time.
usage: shard_probe.py <workload> <T> <N> <shards> [block]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from swarmkit_amd import host as swhost, shard as swshard, synth

name, T, N, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
block = int(sys.argv[5]) if len(sys.argv) > 5 else swshard.BLOCK
wl = synth.Workload(name, T=T, N=N)
ranges = swshard.shard_ranges(wl.N, G)
batches = []
for g, (first, cnt) in enumerate(ranges):
    s = swhost.HostScheduler(shard_rank=g, shard_count=G)
    batches.append(s.e.batch_prepare(swhost.load_workload(s, wl, first, cnt)))
    s.e.state_save()
    batches[-1]._sched = s
for rep in range(2):
    for b in batches:
        b._sched.e.state_restore()
    grp = swshard.ShardGroup(batches, [r[0] for r in ranges], block=block)
    t0 = time.perf_counter()
    out, hist = grp.run(want_hist=True)
    dt = time.perf_counter() - t0
    print(f"{name} T={T} N={N} shards={G} block={block}: {dt*1e3:.1f} ms, {grp.rounds} rounds, {T/grp.rounds:.1f} tasks/round, {T/dt/1e3:.1f} k placements/s, placed {(out>=0).sum()}", flush=True)
