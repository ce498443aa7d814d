"""ctypes wrapper around the CPU oracle (oracle/_build/libswkoracle.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Objects are plain dicts shaped like the Go structs (api.Node / api.Task),
so the known-answer tests read like the reference's own test literals.
"""
import ctypes
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libswkoracle.so")

# api/types.proto:510-539
NEW, PENDING, ASSIGNED, ACCEPTED, PREPARING, READY_T, STARTING, RUNNING = 0, 64, 192, 256, 320, 384, 448, 512
COMPLETE, SHUTDOWN, FAILED, REJECTED, REMOVE, ORPHANED = 576, 640, 704, 768, 800, 832
# NodeStatus.State / Availability
UNKNOWN, DOWN, READY, DISCONNECTED = 0, 1, 2, 3
ACTIVE, PAUSE, DRAIN = 0, 1, 2

_lib = None


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in ("swk_oracle.cpp", "swk_oracle_capi.cpp", "swk_oracle.hpp", "orc_json.hpp", "Makefile")]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src):
        return LIB_PATH
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.orc_new.restype = ctypes.c_void_p
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_result.restype = ctypes.c_char_p
        L.orc_set_now.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.orc_get_now.argtypes = [ctypes.c_void_p]
        L.orc_get_now.restype = ctypes.c_int64
        for name in ("orc_create_or_update_node", "orc_delete_node", "orc_create_task", "orc_setup_task", "orc_update_task",
                     "orc_delete_task", "orc_delete_service", "orc_node_info", "orc_update_volume", "orc_volume_info"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_set_service.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint64]
        L.orc_tick.argtypes = [ctypes.c_void_p]
        L.orc_free_volumes.argtypes = [ctypes.c_void_p]
        L.orc_process_preassigned.argtypes = [ctypes.c_void_p]
        L.orc_process_calls.argtypes = [ctypes.c_void_p]
        L.orc_process_calls.restype = ctypes.c_uint64
        L.orc_nodeless_calls.argtypes = [ctypes.c_void_p]
        L.orc_nodeless_calls.restype = ctypes.c_uint64
        L.orc_constraint_filter.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.orc_constraint_parse.argtypes = [ctypes.c_char_p]
        L.orc_constraint_match.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.orc_equal_fold.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.orc_pipeline_process.argtypes = [ctypes.c_char_p]
        L.orc_enforce.argtypes = [ctypes.c_char_p]
        L.orc_generic.argtypes = [ctypes.c_char_p]
        L.orc_nodeinfo_ops.argtypes = [ctypes.c_char_p]
        L.orc_tree.argtypes = [ctypes.c_char_p]
        L.orc_volumes.argtypes = [ctypes.c_char_p]
        _lib = L
    return _lib


def _j(obj):
    return json.dumps(obj).encode()


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise OracleError(lib().orc_last_error().decode())
    return rc


class Oracle:
    """The canonical-order restatement of manager/scheduler.Scheduler (no store: commits always succeed)."""

    SECOND = 1_000_000_000

    def __init__(self):
        self.L = lib()
        self.h = ctypes.c_void_p(self.L.orc_new())

    def __del__(self):
        try:
            self.L.orc_free(self.h)
        except Exception:
            pass

    # event handlers (scheduler.go Run loop)
    def create_node(self, node):
        _check(self.L.orc_create_or_update_node(self.h, _j(node)))

    update_node = create_node

    def delete_node(self, node_id):
        _check(self.L.orc_delete_node(self.h, node_id.encode()))

    def create_task(self, task):
        return bool(_check(self.L.orc_create_task(self.h, _j(task))))

    def setup_task(self, task):
        """A task that is already in the store when the scheduler starts (setupTasksList, scheduler.go:88-124)."""
        return bool(_check(self.L.orc_setup_task(self.h, _j(task))))

    def update_task(self, task):
        return bool(_check(self.L.orc_update_task(self.h, _j(task))))

    def delete_task(self, task):
        return bool(_check(self.L.orc_delete_task(self.h, _j(task))))

    def set_service(self, service_id, spec_version=None):
        _check(self.L.orc_set_service(self.h, service_id.encode(), 0 if spec_version is None else 1, spec_version or 0))

    def update_volume(self, volume):
        """EventUpdateVolume (scheduler.go:200-213) / a volume of the store at start (:70-81): taken only once the plugin has created it."""
        _check(self.L.orc_update_volume(self.h, _j(volume)))

    def volume_info(self, volume_id):
        _check(self.L.orc_volume_info(self.h, volume_id.encode()))
        return json.loads(self.L.orc_result().decode())

    def free_volumes(self):
        """freeVolumes (volumes.go:181-221), what tick defers (scheduler.go:501): [{"VolumeID", "NodeIDs"}] = the publish statuses to move
        from PUBLISHED to PENDING_NODE_UNPUBLISH."""
        _check(self.L.orc_free_volumes(self.h))
        return json.loads(self.L.orc_result().decode())

    def delete_service(self, service_id):
        _check(self.L.orc_delete_service(self.h, service_id.encode()))

    def tick(self):
        _check(self.L.orc_tick(self.h))
        return json.loads(self.L.orc_result().decode())

    def process_preassigned(self):
        _check(self.L.orc_process_preassigned(self.h))
        return json.loads(self.L.orc_result().decode())

    def node_info(self, node_id):
        rc = _check(self.L.orc_node_info(self.h, node_id.encode()))
        if rc == 1:
            return None   # errNodeNotFound
        return json.loads(self.L.orc_result().decode())

    def advance(self, seconds):
        self.L.orc_set_now(self.h, self.L.orc_get_now(self.h) + int(seconds * self.SECOND))

    @property
    def process_calls(self):
        return self.L.orc_process_calls(self.h)


def constraint_filter(constraints, node):
    """ConstraintFilter.SetTask+Check: None when SetTask returns false, else bool."""
    rc = lib().orc_constraint_filter(_j(constraints), _j(node))
    if rc == -2:
        raise OracleError(lib().orc_last_error().decode())
    return None if rc == -1 else bool(rc)


def constraint_parse(constraints):
    rc = lib().orc_constraint_parse(_j(constraints))
    if rc < 0:
        raise OracleError(lib().orc_last_error().decode())
    out = lib().orc_result().decode()
    if rc == 1:
        return None, out
    return [tuple(x) for x in json.loads(out)], None


def constraint_match(expr, what):
    rc = lib().orc_constraint_match(expr.encode(), what.encode())
    if rc < 0:
        raise OracleError("unparsable constraint " + expr)
    return bool(rc)


def equal_fold(a, b):
    return bool(lib().orc_equal_fold(a.encode(), b.encode()))


def generic(op, node=(), assigned=(), res=(), node_res=()):
    """api/genericresource pure functions. Returns {"node": [...], "assigned": [...], "ok": bool}."""
    _check(lib().orc_generic(_j({"op": op, "node": list(node), "assigned": list(assigned), "res": list(res), "nodeRes": list(node_res)})))
    return json.loads(lib().orc_result().decode())


def enforce(node, tasks, services=None):
    """constraintenforcer.rejectNoncompliantTasks for one node: ids of the tasks that would be REJECTED.
    tasks: api.Task docs in store order (canonical: ascending ID); services: {ServiceID: api.Service doc}."""
    _check(lib().orc_enforce(_j({"Node": node, "Tasks": list(tasks), "Services": services or {}})))
    return json.loads(lib().orc_result().decode())


def volumes(volumes=(), reserve=(), node=None, check=None, mount=None, task=None, topology=None):
    """volumes.go / topology.go in isolation (a fresh volumeSet per call): checkVolume, isVolumeAvailableOnNode, chooseTaskVolumes, IsInTopology."""
    doc = {"Volumes": list(volumes), "Reserve": [list(r) for r in reserve]}
    if node is not None:
        doc["Node"] = node
    for k, v in (("Check", check), ("Mount", mount), ("Task", task), ("Topology", topology)):
        if v is not None:
            doc[k] = v
    _check(lib().orc_volumes(_j(doc)))
    return json.loads(lib().orc_result().decode())


def pipeline_process(task, node, available=None, by_service=None, used_ports=None):
    doc = {"Task": task, "Node": node, "Available": available, "ByService": by_service, "UsedPorts": used_ports}
    _check(lib().orc_pipeline_process(_j(doc)))
    return json.loads(lib().orc_result().decode())


def nodeinfo_ops(node, available, tasks, ops):
    _check(lib().orc_nodeinfo_ops(_j({"Node": node, "Available": available, "Tasks": tasks, "Ops": ops})))
    return json.loads(lib().orc_result().decode())


def tree(nodes, service_id, preferences, max_assignments):
    _check(lib().orc_tree(_j({"Nodes": nodes, "ServiceID": service_id, "Preferences": preferences, "MaxAssignments": max_assignments})))
    return json.loads(lib().orc_result().decode())
