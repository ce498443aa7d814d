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
