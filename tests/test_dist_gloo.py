"""world_size-2 gloo tests (CPU) of the multi-rank paths bench.py uses: the timing protocol (replica seeds, barrier,
max-over-ranks timing, whole-job rate) and the node-range shard protocol — all_gather of the block's proposals, the same
deterministic merge on every rank, owner-applied picks — with a numpy model standing in for the GPU kernels of each shard
(tests/shard_model.py) and the REAL merge (swp_shard_merge of libswp.so, a pure host function)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from swarmkit_amd import dist as swdist, synth
    r = swdist.Ranks(backend="gloo")
    wl = synth.Workload("cfg3", T=200, N=50, seed=r.replica_seed(0x5EED0000))
    r.barrier()
    elapsed = 0.25 * (rank + 1)            # rank 1 is the slow one
    rate, total, t = swdist.whole_job_rate(r, wl.T, elapsed)
    q.put((rank, wl.seed, float(wl.node_cpu.sum()), rate, total, t))
    r.barrier()
    r.close()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seed0, cpu0, rate0, tot0, t0), (r1, seed1, cpu1, rate1, tot1, t1) = got
    assert (r0, r1) == (0, 1)
    assert seed0 != seed1 and cpu0 != cpu1          # two different clusters of the same shape
    assert tot0 == tot1 == 400                      # units of all ranks
    assert t0 == t1 == pytest.approx(0.5)           # max over ranks
    assert rate0 == rate1 == pytest.approx(800.0)   # whole-job rate


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from swarmkit_amd import dist as swdist
    old = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        r = swdist.Ranks()
        assert (r.rank, r.world) == (0, 1) and r.replica_seed(5) is None
        assert swdist.whole_job_rate(r, 100, 0.5) == (200.0, 100, 0.5)
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def _shard_worker(rank, world, port, q, seed, block):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import shard_model
    from swarmkit_amd import shard as swshard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = shard_model.ToyProblem(seed, n_nodes=97, n_tasks=900, n_services=12)
    ranges = swshard.shard_ranges(prob.N, world)
    me = shard_model.ModelShard(prob, rank, *ranges[rank])
    drv = swshard.RankShard(me, rank, world, [r[0] for r in ranges], dist, "cpu", block=block)
    out, hist = drv.run()
    q.put((rank, out.tolist(), drv.rounds, int(hist.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,block", [(2, 64), (3, 7)])
def test_node_range_shards_over_gloo(world, block):
    """Every rank ends with the same full placement vector, and it is the sequential scan's."""
    import numpy as np
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_model
    seed = 1234 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(rk, world, port, q, seed, block)) for rk in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = shard_model.ToyProblem(seed, n_nodes=97, n_tasks=900, n_services=12).sequential()
    for rank, out, rounds, hsum in got:
        assert np.array_equal(np.asarray(out), want), rank
        assert rounds == got[0][2] and rounds < 900   # lock-step rounds; an exchange decides more than one task on average
    assert (want >= 0).sum() > 100 and (want < 0).sum() > 0   # the toy problem places most tasks and rejects some


# ------------------------------------------------------------------------------------------------------------------------------
# The DEVICE-ROUNDS protocol bench.py --gpus N runs (swp_shard_run_rank): one process per rank, each with ONE shard; per round the rank
# proposes over its own range with the product's kernel source, the blocks of R6Prop records of all ranks are all-gathered in rank order
# (the layout ncclAllGather leaves in d_all), EVERY rank folds + matches the gathered blocks itself and applies the picks of its range.
# Here the kernels run on CPU fibers (tests/emu/emu_resolve7.cpp, rank variant) and torch.distributed over gloo carries the blocks.
def _emu7():
    import subprocess
    here = os.path.join(ROOT, "tests")
    out = os.path.join(here, "_build", "emu_resolve7")
    csrc = os.path.join(ROOT, "swarmkit_amd", "csrc")
    srcs = [os.path.join(here, "emu", f) for f in ("emu_resolve7.cpp", "wv_emu.hpp", "emu_model.hpp")] + [os.path.join(csrc, f) for f in ("swp_resolve6.hpp", "swp_resolve7.hpp", "swp_shard.hpp", "swp_types.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, out)
    return out


def _rounds_worker(rank, world, port, q, emu, case):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import subprocess
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed, N, T, S, block, order, feat = case[:7]
    churn = len(case) > 7 and case[7] == "c"   # a second batch after node and task events (the emulation's "c" flag)
    prop_bytes = 8 + 12 * 16 + 24   # sizeof(R6Prop): level, n_cand, 32 half-word indices (16 bits) + 32 candidate words, the exception-list candidate
    send_bytes = block * prop_bytes + 144   # + sizeof(R7Tail): what a rank contributes to a round's exchange (swp_resolve7.hpp r7_send_bytes)
    proc = subprocess.Popen([emu] + [str(x) for x in (seed, N, T, S, block, order, feat, world)] + ["r%d" % rank] + (["c"] if churn else []), stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    rounds = 0
    batches_left = 2 if churn else 1
    try:
        while True:
            go = int.from_bytes(proc.stdout.read(4), "little")
            flags = torch.tensor([go], dtype=torch.int64)
            all_flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(all_flags, flags)
            assert len({int(f.item()) for f in all_flags}) == 1, "the ranks disagree on whether the batch is done: %s" % all_flags
            if not go:
                batches_left -= 1
                if batches_left == 0:
                    break
                continue   # the rank applies the drains and removals of ITS nodes, then the second sharded batch starts
            mine = torch.frombuffer(bytearray(proc.stdout.read(send_bytes)), dtype=torch.uint8)
            assert mine.numel() == send_bytes
            gathered = torch.empty(world * mine.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, mine)     # rank order: [rank 0's block][rank 1's block]...
            proc.stdin.write(gathered.numpy().tobytes())
            proc.stdin.flush()
            rounds += 1
        proc.stdin.close()
        err = proc.stderr.read().decode()
        rc = proc.wait(timeout=120)
    finally:
        if proc.poll() is None:
            proc.kill()
    q.put((rank, rc, rounds, err[-600:]))
    dist.barrier()
    dist.destroy_process_group()


# (seed, nodes, tasks, services, block, task order, feature level)
@pytest.mark.parametrize("world,case", [(2, (2, 700, 900, 30, 64, 0, 1)), (3, (3, 1000, 700, 40, 64, 2, 2)), (3, (7, 500, 600, 40, 32, 0, 3)), (2, (13, 401, 500, 8, 128, 1, 1)),
                                        (2, (3, 1000, 900, 40, 64, 2, 2, "c")), (3, (7, 500, 900, 40, 32, 0, 3, "c"))],
                         ids=["2ranks-maxrep", "3ranks-ports-uncounted", "3ranks-generic", "2ranks-service-major", "2ranks-churn-node-updates-between-two-batches", "3ranks-churn-generic"])
def test_device_rounds_protocol_over_gloo(world, case):
    """Every rank's shard ends exactly where the sequential model over the WHOLE node set puts it; the ranks take the same number of
    rounds and agree on the end of the batch in the same round."""
    import torch.multiprocessing as mp
    emu = _emu7()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rounds_worker, args=(rk, world, port, q, emu, case)) for rk in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rc, rounds, err in got:
        assert rc == 0 and "-> OK" in err, (rank, rc, err)
        assert rounds == got[0][2] and 0 < rounds < case[2]   # lock-step rounds; an exchange decides more than one task on average
        if len(case) > 7:   # the incremental path: drains + NodeInfo.removeTask on every rank's own nodes, then a second sharded batch
            assert err.count("-> OK") == 2, (rank, err)


# ------------------------------------------------------------------------------------------------------------------------------
# Task GROUPS in a job of ranks (swarmkit_amd.shard.RankUnionGroups): rank 0's union engine places the groups, one
# broadcast carries the placements, every owner books its share — BETWEEN two sharded one-off batches, whose results the union has to
# learn (note_batch), and with tasks going away in between (commit(remove) by global index). Engines: the persistent toy cluster of
# tests/shard_model.py; the exchange of the one-off batches is the real RankShard protocol with the real swp_shard_merge.
def _groups_worker(rank, world, port, q, seed):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    import shard_model
    from swarmkit_amd import abi, shard as swshard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = shard_model.ToyProblem(seed, n_nodes=97, n_tasks=600, n_services=12)
    ranges = swshard.shard_ranges(prob.N, world)
    firsts, counts = [r[0] for r in ranges], [r[1] for r in ranges]
    local = shard_model.ToyEngine(prob, rank, *ranges[rank])
    union = shard_model.ToyEngine(prob, 0, 0, prob.N) if rank == 0 else None
    ru = swshard.RankUnionGroups(local, union, rank, world, firsts, counts, dist, "cpu")
    A, B = np.arange(0, 300), np.arange(300, 600)
    drv = swshard.RankShard(local.batch(A), rank, world, firsts, dist, "cpu", block=64)
    out_a, _ = drv.run(want_hist=False)
    ru.note_batch(local.descs(A), out_a)
    gone = np.nonzero(out_a >= 0)[0][::7]                       # every seventh placed task goes away (a drain's removeTask)
    ru.commit(out_a[gone], local.descs(A[gone]), add=False)
    groups = np.zeros(6, dtype=abi.TASK_DTYPE)
    groups["service"] = np.arange(6)
    groups["cpu"] = prob.svc_need[np.arange(6)]
    out_g, hist = ru.schedule_groups(groups, np.full(6, 7, dtype=np.uint32))
    drv = swshard.RankShard(local.batch(B), rank, world, firsts, dist, "cpu", block=64)
    out_b, _ = drv.run(want_hist=False)
    ru.note_batch(local.descs(B), out_b)
    state = (local.total.tolist(), local.cpu.tolist(), local.cnt.sum(axis=0).tolist())
    ustate = (union.total.tolist(), union.cpu.tolist()) if union is not None else None
    q.put((rank, out_a.tolist(), out_g.tolist(), out_b.tolist(), state, ustate, hist.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_task_groups_between_two_sharded_batches_over_gloo(world):
    """Every rank ends with the placements and the node rows of ONE engine running the same script; rank 0's union holds them all."""
    import numpy as np
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_model
    from swarmkit_amd import abi, shard as swshard
    seed = 4321 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_groups_worker, args=(rk, world, port, q, seed)) for rk in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the same script on one engine over the whole node set
    prob = shard_model.ToyProblem(seed, n_nodes=97, n_tasks=600, n_services=12)
    ref = shard_model.ToyEngine(prob, 0, 0, prob.N)
    A, B = np.arange(0, 300), np.arange(300, 600)
    want_a = np.array([ref.place_one(int(s)) for s in prob.task_svc[A]])
    gone = np.nonzero(want_a >= 0)[0][::7]
    pl = np.zeros(len(gone), dtype=abi.PLACEMENT_DTYPE)
    pl["node"], pl["service"], pl["cpu"], pl["counted"] = want_a[gone], prob.task_svc[A[gone]], prob.svc_need[prob.task_svc[A[gone]]], 1
    ref.commit(pl, add=False)
    groups = np.zeros(6, dtype=abi.TASK_DTYPE)
    groups["service"] = np.arange(6)
    groups["cpu"] = prob.svc_need[np.arange(6)]
    want_g, _ = ref.schedule_groups(groups, np.full(6, 7, dtype=np.uint32))
    want_b = np.array([ref.place_one(int(s)) for s in prob.task_svc[B]])
    ranges = swshard.shard_ranges(prob.N, world)
    assert (want_g >= 0).sum() > 20 and (want_b >= 0).sum() > 100
    for rank, out_a, out_g, out_b, state, ustate, hshape in got:
        assert np.array_equal(np.asarray(out_a), want_a), rank
        assert np.array_equal(np.asarray(out_g), want_g), rank      # every rank holds the groups' placements (one broadcast)
        assert np.array_equal(np.asarray(out_b), want_b), rank      # ... and the batch behind them saw what they took
        first, cnt = ranges[rank]
        assert state[0] == ref.total[first:first + cnt].tolist() and state[1] == ref.cpu[first:first + cnt].tolist(), rank
        assert state[2] == ref.cnt.sum(axis=0)[first:first + cnt].tolist(), rank
        assert tuple(hshape) == (6, abi.NFILTERS)
        if rank == 0:
            assert ustate[0] == ref.total.tolist() and ustate[1] == ref.cpu.tolist()
