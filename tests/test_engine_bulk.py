"""GPU: the bulk node calls (swp_node_update_dynamic_many / swp_node_get_many) are the single-node calls in a loop — same rows
back, same placements afterwards — and a drain round through them equals the oracle."""
import numpy as np
import pytest

import parity_util as pu
from swarmkit_amd import abi, host as swhost, synth

pytestmark = pytest.mark.gpu


def test_bulk_update_equals_single_updates():
    wl = synth.Workload("cfg3", T=1500, N=300)
    outs = []
    for bulk in (False, True):
        s = swhost.HostScheduler()
        descs = swhost.load_workload(s, wl)
        e = s.e
        drained = np.arange(0, wl.N, 7, dtype=np.uint32)
        rows = e.node_get_many(drained)
        for i, n in enumerate(drained):
            one = e.node_get(int(n))
            assert (one.cpu, one.mem, one.total, one.flags) == (int(rows["cpu"][i]), int(rows["mem"][i]), int(rows["total"][i]), int(rows["flags"][i]))
        if bulk:
            upd = np.zeros(len(drained), dtype=abi.NODE_DYNAMIC_DTYPE)
            upd["node"], upd["cpu"], upd["mem"], upd["total"] = drained, rows["cpu"], rows["mem"] // 2, rows["total"] + 3
            upd["flags"] = rows["flags"] & ~np.uint32(abi.NODE_READY)
            e.node_update_dynamic_many(upd)
        else:
            for i, n in enumerate(drained):
                e.node_update_dynamic(int(n), int(rows["flags"][i]) & ~abi.NODE_READY, int(rows["cpu"][i]), int(rows["mem"][i]) // 2, int(rows["total"][i]) + 3)
        out, hist = e.schedule_batch(descs)
        assert not np.isin(out[out >= 0], drained).any()      # nothing lands on a node that is not READY
        outs.append((out, hist))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_bulk_update_rejects_unknown_nodes():
    s = swhost.HostScheduler()
    swhost.load_workload(s, synth.Workload("cfg2", T=10, N=5))
    upd = np.zeros(2, dtype=abi.NODE_DYNAMIC_DTYPE)
    upd["node"] = [1, 99]
    with pytest.raises(abi.SwpError) as ei:
        s.e.node_update_dynamic_many(upd)
    assert ei.value.code == abi.SWP_ENOTFOUND and "row 1" in str(ei.value)
