"""GPU: seeded random BATCHES at sizes between the small event scripts of tests/test_engine_fuzz.py and the fixed BASELINE-size digests —
workload family, node count (1 … 14 000: both resolver families, node sets around word and chunk borders), task count (up to 30 000,
bounded so that the oracle's T x N stays within a second or two), number of services (one service of thousands of tasks: water-filling
and the scan resolver; hundreds: the block resolver's rounds, cuts and compact index), task order, uncounted tasks, and — a third of
the seeds — the same batch over a shard SET. Placements and explanations must equal the oracle's bit for bit.
SWP_FUZZ_SEEDS / SWP_FUZZ_FIRST: a soak (round 6: see docs/COVERAGE_r06.md)."""
import os
import random

import pytest

import parity_util as pu
from swarmkit_amd import synth

pytestmark = pytest.mark.gpu
FIRST = int(os.environ.get("SWP_FUZZ_FIRST", "0"))


def draw(seed):
    rng = random.Random(0xB10C + seed)
    name = rng.choice(["cfg2", "cfg3", "cfg3", "cfg4", "cfg4", "cfg3m"])
    N = rng.choice([rng.randrange(1, 200), 64 * rng.randrange(1, 40) + rng.choice([-1, 0, 1]), rng.randrange(200, 3000), rng.randrange(3000, 14000)])
    T = rng.choice([rng.randrange(1, 500), rng.randrange(500, 5000), rng.randrange(5000, 30000)])
    T = max(1, min(T, 60_000_000 // N))
    services = rng.choice([None, None, 1, rng.randrange(1, 20), rng.randrange(20, 400)])
    order = rng.choice(["rr", "rr", "major"])
    uncounted = rng.choice([0, 0, 0, 3, 17])
    shards = rng.choice([0, 0, 2, 5])
    wseed = rng.randrange(1 << 30)
    # engine knobs (drawn last: the shapes of the seeds stay what they were): the compact index forced on from the first round — with the
    # next round's index built at the end of k_r6_commit_c —, other block sizes
    knobs = {}
    if rng.random() < 0.35:
        knobs["SWP_R6_COMPACT"] = "1"
        if rng.random() < 0.3:
            knobs["SWP_R6_COMPACT_FUSED"] = "0"
    if rng.random() < 0.3:
        knobs["SWP_R6_BLOCK"] = str(rng.choice([1, 64, 192, 448, 1024]))
    return name, T, N, services, order, wseed, uncounted, shards, knobs


@pytest.mark.parametrize("seed", range(FIRST, FIRST + int(os.environ.get("SWP_FUZZ_SEEDS", "16"))))
def test_random_midsize_batches(seed, monkeypatch):
    name, T, N, services, order, wseed, uncounted, shards, knobs = draw(seed)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    if shards:
        monkeypatch.setenv("SWP_SHARDSET", "%d:%d" % (shards, (N + shards - 1) // shards + 3))
    wl = synth.Workload(name, T=T, N=N, seed=wseed, services=services, order=order)
    wl.uncounted_every = uncounted
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


# ---- event scripts at the same sizes: a first tick, then rounds of {reactivate, drain a random share of the nodes, remove their tasks,
# ---- sometimes a node leaves or comes back empty, re-place} — the incremental path (dirty rows, exception lists, the compact index kept
# ---- across batches) and, with SpecVersions, the task groups of k_groups2, under shapes no fixed script has
def draw_script(seed):
    rng = random.Random(0xC4A2 + seed)
    name = rng.choice(["cfg2", "cfg3", "cfg3", "cfg4"])
    N = rng.choice([rng.randrange(20, 300), rng.randrange(300, 2000), rng.randrange(2000, 5000)])
    per = rng.choice([1, 3, 8, 12])
    T0 = max(1, min(N * per, 30_000_000 // N))
    services = rng.choice([None, rng.randrange(1, 12), rng.randrange(12, 200)])
    p = dict(name=name, N=N, T0=T0, services=services, grouped=rng.random() < 0.4, rounds=rng.randrange(2, 6), wseed=rng.randrange(1 << 30),
             shards=rng.choice([0, 0, 0, 3]), rseed=rng.randrange(1 << 30))
    p["knobs"] = {"SWP_R6_COMPACT": "1"} if rng.random() < 0.35 else {}   # (drawn last, as above)
    return p


def run_script(s, p):
    import orc
    from bigcases import tick_digest
    rng = random.Random(p["rseed"])
    total = p["T0"] * (p["rounds"] + 2)
    wl = synth.Workload(p["name"], T=total, N=p["N"], seed=p["wseed"], services=p["services"], grouped=p["grouped"])
    N = p["N"]
    for i in range(N):
        s.create_node(wl.node_doc(i))
    for k in range(wl.S):
        s.set_service(wl.service_id(k))
    for j in range(p["T0"]):
        s.create_task(wl.task_doc(j))
    placed, by_node, present = {}, {i: [] for i in range(N)}, set(range(N))
    ticks = []

    def do_tick():
        dec = s.tick()
        ticks.append(tick_digest(dec))
        for d in dec:
            if d["NodeID"] and d["State"] >= orc.ASSIGNED:
                j, n = int(d["ID"][1:]), int(d["NodeID"][1:])
                placed[j] = n
                by_node[n].append(j)

    do_tick()
    nxt, prev = p["T0"], []
    for rnd in range(p["rounds"]):
        for i in prev:
            if i in present:
                s.update_node(wl.node_doc(i))
        share = rng.choice([0.03, 0.1, 0.1, 0.3])
        drained = sorted(rng.sample(sorted(present), max(1, int(len(present) * share))))
        for i in drained:
            doc = wl.node_doc(i)
            doc["Spec"] = dict(doc["Spec"], Availability=2)
            s.update_node(doc)
        gone = []
        for i in drained:
            gone.extend(by_node[i])
            by_node[i] = []
        if rng.random() < 0.4 and len(present) > 2:   # a node leaves with whatever runs there
            i = rng.choice(sorted(present - set(drained)))
            s.delete_node(wl.node_id(i))
            present.discard(i)
            for j in by_node[i]:
                placed.pop(j, None)   # (its tasks are orphaned, as in tests/test_engine_fuzz.py: nobody deletes them)
            by_node[i] = []
        if rng.random() < 0.3 and len(present) < N:   # a node comes back empty
            i = rng.choice(sorted(set(range(N)) - present))
            s.create_node(wl.node_doc(i))
            present.add(i)
        gone.sort()
        for j in gone:
            s.delete_task(dict(wl.task_doc(j), NodeID=wl.node_id(placed[j]), Status={"State": orc.RUNNING}))
            del placed[j]
        for _ in range(min(len(gone) + rng.randrange(0, 40), total - nxt)):
            s.create_task(wl.task_doc(nxt))
            nxt += 1
        do_tick()
        prev = drained
    return ticks


@pytest.mark.parametrize("seed", range(FIRST, FIRST + int(os.environ.get("SWP_FUZZ_SEEDS", "10"))))
def test_random_midsize_event_scripts(seed, monkeypatch):
    import orc
    from swarmkit_amd import host as swhost
    p = draw_script(seed)
    want = run_script(orc.Oracle(), p)
    for k, v in p["knobs"].items():
        monkeypatch.setenv(k, v)
    if p["shards"]:
        monkeypatch.setenv("SWP_SHARDSET", "%d:%d" % (p["shards"], (p["N"] + p["shards"] - 1) // p["shards"] + 3))
    got = run_script(swhost.HostScheduler(), p)
    assert want == got, (seed, p, [k for k, (a, b) in enumerate(zip(want, got)) if a != b])
