"""What the tests and bench.py drive above the C ABI: the host layer INSIDE libswp.so (swp::Scheduler, csrc/swp_sched.cpp, through
swarmkit_amd.sched) — the event handlers of manager/scheduler.Scheduler with the reference's names — plus the bulk loader of the
synthetic workloads. (An independent Python twin of that layer exists as test infrastructure: tests/pyhost.py; SWP_HOST=py selects it
once tests/conftest.py has registered it.)"""
import os

import numpy as np

from . import abi, sched

RUNNING = 512
Unsupported = abi.Unsupported   # the task/feature stays on the reference's own Go path (SWP_EUNSUPPORTED)
_twin = None


def register_twin(factory):
    """tests only: the factory of the Python twin (tests/pyhost.PyHostScheduler)."""
    global _twin
    _twin = factory


def HostScheduler(engine=None, **engine_kw):
    if os.environ.get("SWP_HOST", "cxx") == "py":
        if _twin is None:
            raise RuntimeError("SWP_HOST=py: the Python twin of the host layer is test infrastructure (tests/pyhost.py); the product's host layer is the one inside libswp.so")
        return _twin(engine, **engine_kw)
    return sched.Scheduler(engine, **engine_kw)


def enforce(s, node_docs, tasks_by_node, services=None):
    """constraintenforcer.rejectNoncompliantTasks (constraint_enforcer.go:65-196) for many nodes through swp_enforce."""
    return s.enforce(node_docs, tasks_by_node, services)


def load_workload(sched_, wl, first=0, count=None):
    """Bulk path used by bench.py / large parity tests: nodes through create_node, tasks as one
    descriptor array (per-service spec translated once, then tiled). first / count: only that range of the
    workload's nodes (a node-range shard, swarmkit_amd.shard); every shard sees every task."""
    for i in range(first, wl.N if count is None else first + count):
        sched_.create_node(wl.node_doc(i))
    for k in range(wl.S):
        sched_.set_service(wl.service_id(k))
    per_service = np.concatenate([sched_.task_desc(dict(wl.service_spec(k), ID="x", ServiceID=wl.service_id(k), DesiredState=RUNNING))
                                  for k in range(wl.S)])
    svc_of_task = np.array([wl.task_service(j) for j in range(wl.T)], dtype=np.int64) if getattr(wl, "order", "rr") != "rr" else np.arange(wl.T) % wl.S
    descs = per_service[svc_of_task]
    k = getattr(wl, "uncounted_every", 0)
    if k:
        descs["flags"][k - 1::k] |= abi.TASK_UNCOUNTED
    return descs
