"""GPU: swp_batch_prepare_templates (the caller names the template of every task) decides exactly what swp_batch_prepare decides —
placements and Explain rows — on the synthetic workloads, in both task orders, with templates that repeat and templates nobody uses."""
import numpy as np
import pytest

from swarmkit_amd import abi, host, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,T,N,order", [("cfg3", 6000, 700, "rr"), ("cfg3", 6000, 700, "major"), ("cfg4", 9000, 1500, "rr"), ("cfg2", 3000, 200, "rr")])
def test_templates_equal_plain_prepare(name, T, N, order):
    wl = synth.Workload(name, T=T, N=N, order=order)
    eng = abi.Engine()
    sched = host.HostScheduler(engine=eng)
    descs = host.load_workload(sched, wl)
    eng.state_save()
    b = eng.batch_prepare(descs)
    b.run()
    want_out, want_hist = b.results(want_hist=True)
    b.free()
    eng.state_restore()
    tmpl, idx = np.unique(descs, return_inverse=True)
    # an unused template in front and a duplicate of a used one behind: indices shift, nothing else may
    tmpl2 = np.concatenate([tmpl[:1], tmpl, tmpl[-1:]])
    idx2 = idx.astype(np.uint32) + 1
    idx2[::7] = np.where(idx[::7] == len(tmpl) - 1, len(tmpl2) - 1, idx2[::7])
    b = eng.batch_prepare_templates(tmpl2, idx2)
    b.run()
    out, hist = b.results(want_hist=True)
    b.free()
    assert (out == want_out).all()
    assert (hist == want_hist).all()
    with pytest.raises(abi.SwpError):
        eng.batch_prepare_templates(tmpl, np.full(3, len(tmpl), dtype=np.uint32))   # a template index out of range
