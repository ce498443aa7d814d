"""GPU parity of the node-range sharded scan (include/swp.h "node-range shards"): the node set split over
2 / 3 / 4 engines — here all on one device — must place every task exactly where the oracle's single sequential scan does, with the
same explanation for every unplaceable task. Two drivers: "device" = the rounds on the device (swp_shard_run, csrc/swp_resolve7.hpp:
block-resolver proposals per shard, one matching wave over the folded records, every shard applies its picks — what a one-process
manager over the GPUs of a box runs), "host" = the round-2 protocol with the merge on the host (between processes the exchange is an
RCCL all-gather: swarmkit_amd.shard.RankShard, tests/test_dist_gloo.py)."""
import numpy as np
import pytest

import parity_util as pu
from swarmkit_amd import synth

pytestmark = pytest.mark.gpu


MODES = ["device", "host"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shards", [2, 3, 4])
@pytest.mark.parametrize("name,T,N,kw", [("cfg3", 2500, 300, {}), ("cfg4", 3000, 700, {}), ("cfg2", 3000, 50, {}), ("cfg1", 500, 40, {}), ("cfg3m", 3000, 500, {"services": 900})])
def test_shards_agree_with_oracle(mode, shards, name, T, N, kw):
    wl = synth.Workload(name, T=T, N=N, **kw)
    op, oe, _ = pu.oracle_run(wl)
    sp, se, rounds = pu.sharded_run(wl, shards, mode=mode)
    pu.assert_same(op, oe, sp, se)
    assert rounds <= T


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("services,order", [(1, "rr"), (3, "major"), (40, "major")])
def test_shards_same_service_runs(mode, services, order):
    """Few services: almost every task ends on its service's exception list (nodes where it already runs), the path where a
    block is cut after one task."""
    wl = synth.Workload("cfg3", T=1200, N=200, services=services, order=order)
    op, oe, _ = pu.oracle_run(wl)
    sp, se, _ = pu.sharded_run(wl, 3, mode=mode)
    pu.assert_same(op, oe, sp, se)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("block", [1, 7, 64, 1024])
def test_block_size_does_not_matter(mode, block, monkeypatch):
    monkeypatch.setenv("SWP_R6_BLOCK", str(block))
    wl = synth.Workload("cfg4", T=1500, N=400)
    op, oe, _ = pu.oracle_run(wl)
    sp, se, _ = pu.sharded_run(wl, 4, block=block, mode=mode)
    pu.assert_same(op, oe, sp, se)


@pytest.mark.parametrize("mode", MODES)
def test_uncounted_tasks_end_the_block(mode):
    """A task that does not count on its node (DesiredState beyond COMPLETED, nodeinfo.go:131-134) leaves the node on its level: the
    proposals of the later tasks of the block are stale about it, so the merge ends the block behind such a pick."""
    wl = synth.Workload("cfg2", T=1500, N=300)
    wl.uncounted_every = 4
    op, oe, _ = pu.oracle_run(wl)
    sp, se, _ = pu.sharded_run(wl, 3, mode=mode)
    pu.assert_same(op, oe, sp, se)


@pytest.mark.parametrize("mode", MODES)
def test_more_shards_than_nodes(mode):
    wl = synth.Workload("cfg2", T=200, N=3)
    op, oe, _ = pu.oracle_run(wl)
    sp, se, _ = pu.sharded_run(wl, 4, mode=mode)   # one shard is empty
    pu.assert_same(op, oe, sp, se)


@pytest.mark.parametrize("mode", MODES)
def test_sharded_equals_single_engine_at_scale(mode):
    """20k x 6k over 4 shards against the single-engine placement vector (itself pinned to the oracle by the other suites)."""
    wl = synth.Workload("cfg4", T=20000, N=6000)
    ep, ee, *_ = pu.engine_run(wl)
    sp, se, rounds = pu.sharded_run(wl, 4, mode=mode)
    pu.assert_same(ep, ee, sp, se)
    assert rounds < wl.T // 4   # an exchange decides many tasks


@pytest.mark.parametrize("mode", MODES)
def test_sharded_cfg4_200k_x_40k_equals_single_engine(mode):
    """BASELINE configs[3] at a fifth of its size — past the node range of the round resolver, so the single engine runs the block
    resolver — against 4 shards of 10k nodes. (The single-engine placement of exactly this case is pinned to the oracle's offline
    digest by tests/test_engine_bigcases.py::cfg4_mid.)"""
    wl = synth.Workload("cfg4", T=200_000, N=40_000)
    ep, ee, *_ = pu.engine_run(wl)
    sp, se, rounds = pu.sharded_run(wl, 4, mode=mode)
    pu.assert_same(ep, ee, sp, se)


@pytest.mark.parametrize("name,T,N", [("cfg3", 2500, 300), ("cfg4", 3000, 700), ("cfg3m", 3000, 500)])
def test_rank_variant_over_rccl_with_one_rank(name, T, N):
    """swp_shard_run_rank on a job of ONE rank: librccl.so loaded by libswp.so, communicator from a unique id, an ncclAllGather per
    round on the engine's stream (here: a copy), fold + match + apply — the path every rank of an 8-GPU job runs — against the oracle.
    (Two ranks cannot share the one GPU of the test box; the folding of several shards' records is covered by the one-process driver
    above, which runs the same kernels.)"""
    from swarmkit_amd import host as swhost
    from swarmkit_amd import shard as swshard
    wl = synth.Workload(name, T=T, N=N)
    op, oe, _ = pu.oracle_run(wl)
    s = swhost.HostScheduler()
    descs = swhost.load_workload(s, wl)
    b = s.e.batch_prepare(descs)
    drv = swshard.DeviceRankShard(b, 0, 1, [(0, wl.N)], None, None)
    out, hist = drv.run()
    placed, errs = {}, {}
    for j in range(wl.T):
        tid = wl.task_id(j)
        if out[j] >= 0:
            placed[tid] = wl.node_id(int(out[j]))
        else:
            placed[tid] = None
            ex = s.explain(hist[j])
            errs[tid] = "no suitable node (" + ex + ")" if ex else "no suitable node"
    b.free()
    s.e.rccl_finalize()
    pu.assert_same(op, oe, placed, errs)


class GenericWorkload(synth.Workload):
    """cfg3's cluster with Discrete generic resources on the nodes and Discrete reservations on two thirds of the services."""

    def node_doc(self, i):
        d = super().node_doc(i)
        gen = []
        if i % 4:
            gen.append({"Discrete": {"Kind": "gpu", "Value": i % 4}})
        if i % 7 == 0:
            gen.append({"Discrete": {"Kind": "fpga", "Value": 2}})
        d["Description"]["Resources"]["Generic"] = gen
        return d

    def service_spec(self, k):
        t = super().service_spec(k)
        if k % 3:
            gen = [{"Discrete": {"Kind": "gpu", "Value": 1 + k % 2}}] + ([{"Discrete": {"Kind": "fpga", "Value": 1}}] if k % 5 == 0 else [])
            t.setdefault("Spec", {}).setdefault("Resources", {}).setdefault("Reservations", {})["Generic"] = gen
        return t


@pytest.mark.parametrize("shards", [2, 4])
def test_shards_with_generic_reservations(shards):
    """Generic reservations through the sharded rounds (round 3 refused them): HasEnough as rows of every shard's propose, Claim in
    the owner's apply step of k_r7_commit — against the oracle and against the single engine."""
    wl = GenericWorkload("cfg3", T=3000, N=500)
    op, oe, _ = pu.oracle_run(wl)
    sp, se, _ = pu.sharded_run(wl, shards, mode="device")
    pu.assert_same(op, oe, sp, se)


def _digest(wl, placed, errs):
    """The tick digest of tests/bigcases.py from a placement map (what the host layer's decision lines would say)."""
    import hashlib
    lines = sorted("%s|%s|%s|%d" % (tid, placed[tid] or "", "" if placed[tid] else errs[tid], 192 if placed[tid] else 64) for tid in placed)
    return hashlib.sha256("\n".join(lines).encode()).hexdigest(), sum(1 for v in placed.values() if v)


def test_cfg4_200k_x_40k_over_8_shards_matches_the_oracle_digest():
    """BASELINE configs[3] at a fifth of its size over EIGHT shards (the job size SURVEY 8e names) against the oracle's offline digest."""
    import json
    import os
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_cfg4_mid.json")))
    wl = synth.Workload("cfg4", T=200_000, N=40_000)
    sp, se, _ = pu.sharded_run(wl, 8, mode="device")
    h, placed = _digest(wl, sp, se)
    assert placed == want["placed"][0]
    assert h == want["ticks"][0]


@pytest.mark.gpu
def test_rank_path_places_task_groups_between_two_sharded_batches():
    """VERDICT r5 row e3 on one GPU: a job of ONE rank through the rank code paths (swp_shard_run_rank + swarmkit_amd.shard.RankUnionGroups:
    the union engine, note_batch, the owners' commits) runs {one-off batch, tasks going away, grouped tick, one-off batch} and ends where a
    single engine running the same script ends — placements, Explain histograms of the groups, node rows."""
    from swarmkit_amd import abi, host, synth
    from swarmkit_amd import shard as swshard
    wl = synth.Workload("cfg3", T=6000, N=1500)

    def build(**kw):
        e = abi.Engine(**kw)
        return e, host.load_workload(host.HostScheduler(engine=e), wl)
    single, descs = build()
    local, descs_l = build(shard_rank=0, shard_count=1)
    union, descs_u = build()
    assert np.array_equal(descs, descs_l) and np.array_equal(descs, descs_u)
    A, B = np.arange(0, 2500), np.arange(3500, 6000)
    G = np.arange(2500, 3500)                       # the tasks in between go as groups: one per service, in service order
    svc = np.array([wl.task_service(int(j)) for j in G])
    order = np.argsort(svc, kind="stable")
    g_first = order[np.concatenate([[True], svc[order][1:] != svc[order][:-1]])]
    groups, sizes = descs[G][g_first], np.bincount(svc)[np.unique(svc)].astype(np.uint32)
    ru = swshard.RankUnionGroups(local, union, 0, 1, [0], [wl.N], None, None)

    def placements(nodes, d):
        pl = np.zeros(len(nodes), dtype=abi.PLACEMENT_DTYPE)
        pl["node"], pl["service"], pl["cpu"], pl["mem"], pl["port_set"], pl["counted"] = nodes, d["service"], d["cpu"], d["mem"], d["port_set"], 1
        return pl
    # one engine
    out_a, _ = single.schedule_batch(descs[A], want_hist=False)
    gone = np.nonzero(out_a >= 0)[0][::5]
    single.commit(placements(out_a[gone], descs[A][gone]), add=False)
    out_g, hist_g = single.schedule_groups(groups, sizes)
    out_b, _ = single.schedule_batch(descs[B], want_hist=False)
    # the rank path
    bt = local.batch_prepare(descs[A])
    drv = swshard.DeviceRankShard(bt, 0, 1, [(0, wl.N)], None, None)   # (opens the RCCL communicator of the one-rank job)
    la, _ = drv.run(want_hist=False)
    bt.free()
    assert np.array_equal(la, out_a)
    ru.note_batch(descs[A], la)
    ru.commit(la[gone], descs[A][gone], add=False)
    rg, rh = ru.schedule_groups(groups, sizes)
    assert np.array_equal(rg, out_g) and np.array_equal(rh, hist_g)
    bt = local.batch_prepare(descs[B])
    lb, _ = swshard.DeviceRankShard(bt, 0, 1, [(0, wl.N)], None, None).run(want_hist=False)
    bt.free()
    local.rccl_finalize()
    assert np.array_equal(lb, out_b)
    ru.note_batch(descs[B], lb)
    assert (out_g >= 0).sum() > 500 and (out_b >= 0).sum() > 1500
    nodes = np.arange(wl.N, dtype=np.uint32)
    want = single.node_get_many(nodes)
    for e in (local, union):
        got = e.node_get_many(nodes)
        for f in ("cpu", "mem", "total"):
            assert np.array_equal(got[f], want[f]), f
