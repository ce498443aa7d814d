"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle on identical seeded inputs — bit-exact
placements and identical 'no suitable node (...)' explanations."""
import numpy as np
import pytest

import parity_util as pu
from swarmkit_amd import abi, synth

pytestmark = pytest.mark.gpu


def test_abi_roundtrip():
    e = abi.Engine()
    e.reset(4)
    n0 = e.intern(abi.SPACE_NODE_ID, "a")
    n1 = e.intern(abi.SPACE_NODE_ID, "b")
    assert (n0, n1) == (0, 1) and e.intern(abi.SPACE_NODE_ID, "a") == 0
    assert e.intern(abi.SPACE_FOLDED, "SSD") == e.intern(abi.SPACE_FOLDED, "ssd")
    assert e.intern(abi.SPACE_ARCH, "x86_64") == e.intern(abi.SPACE_ARCH, "amd64")
    row = abi.NodeRow(node=n1, flags=abi.NODE_READY, cpu=5, mem=7, total=3)
    e.node_upsert(row)
    got = e.node_get(n1)
    assert (got.cpu, got.mem, got.total) == (5, 7, 3)
    assert e.node_get(n0) is None   # errNodeNotFound
    e.node_remove(n1)
    assert e.node_get(n1) is None


@pytest.mark.parametrize("name,T,N", [("cfg2", 2000, 300), ("cfg2", 10_000, 1_000), ("cfg3", 6000, 1000), ("cfg4", 6000, 1500)])
def test_parity_one_off(name, T, N):
    wl = synth.Workload(name, T=T, N=N)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, s, out, hist = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)
    # the host mirror folded the placements: counts agree with the oracle's nodeSet
    st = s.e.stats()
    assert st["placed"] == sum(v is not None for v in op.values())


@pytest.mark.parametrize("window", [64, 256, 4096])
def test_parity_windows(window):
    """Result must not depend on the scan window (freshness of the feasibility snapshot)."""
    wl = synth.Workload("cfg3", T=3000, N=500)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, *_ = pu.engine_run(wl, window=window)
    pu.assert_same(op, oe, ep, ee)


def test_parity_tight_capacity():
    """Nodes fill up inside the batch: exercises the touched-node re-check and the explain pass."""
    wl = synth.Workload("cfg2", T=4000, N=60)
    op, oe, _ = pu.oracle_run(wl)
    assert any(v is None for v in op.values())
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


def test_parity_single_service():
    """All tasks of ONE service (the shape of the reference's own benchScheduler): every node becomes an
    exception node after the first round, so the slow path carries the batch."""
    wl = synth.Workload("cfg1", T=700, N=40)
    op, oe, _ = pu.oracle_run(wl)
    ep, ee, *_ = pu.engine_run(wl)
    pu.assert_same(op, oe, ep, ee)


def test_second_batch_sees_first():
    wl = synth.Workload("cfg3", T=2000, N=400)
    o = None
    op1, oe1, o = pu.oracle_run(wl, count=1000)
    for j in range(1000, 2000):
        o.create_task(wl.task_doc(j))
    second = {d["ID"]: (d["NodeID"] or None) for d in o.tick() if d["ID"] >= wl.task_id(1000)}
    from swarmkit_amd import host as swhost
    s = swhost.HostScheduler()
    descs = swhost.load_workload(s, wl)
    out1, _ = s.e.schedule_batch(descs[:1000])
    out2, _ = s.e.schedule_batch(descs[1000:2000])
    for j in range(1000, 2000):
        want = second.get(wl.task_id(j))
        got = s.idx_to_id[int(out2[j - 1000])] if out2[j - 1000] >= 0 else None
        if want is not None or wl.task_id(j) in second:
            assert want == got, (j, want, got)
