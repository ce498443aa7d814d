"""world_size-2 gloo test (CPU) of the multi-rank protocol bench.py uses: replica seeds, barrier,
max-over-ranks timing and the whole-job rate."""
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
