"""ctypes face of the C++ host layer (include/swp_sched.h, csrc/swp_sched.cpp): swp::Scheduler — the restatement of
manager/scheduler.Scheduler's event handlers and tick above the engine ABI — with the method names the parity tests
use. Documents cross as JSON with the Go field names. No logic here: every method is one call into libswp.so."""
import ctypes as C
import json

import numpy as np

from . import abi


def _b(s):
    return s.encode() if isinstance(s, str) else s


class Scheduler:
    SECOND = 1_000_000_000

    def __init__(self, engine=None, **engine_kw):
        self.e = engine or abi.Engine(**engine_kw)
        self.L = self.e.L
        h = C.c_void_p()
        rc = self.L.swp_sched_create(self.e.h, C.byref(h))
        if rc != 0:
            raise abi.SwpError(rc, (self.L.swp_sched_last_error(None) or b"").decode())
        self.h = h
        self.idx_to_id = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.swp_sched_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc == 0:
            return
        msg = (self.L.swp_sched_last_error(self.h) or b"").decode()
        if rc == abi.SWP_EUNSUPPORTED:
            raise abi.Unsupported(msg)
        raise abi.SwpError(rc, msg)

    def _doc_call(self, fn, doc):
        b = json.dumps(doc).encode()
        flag = C.c_int(0)
        self._ck(fn(self.h, b, len(b), C.byref(flag)))
        return bool(flag.value)

    def _json_result(self, fn, *args):
        out = C.c_char_p()
        self._ck(fn(self.h, *args, C.byref(out)))
        return json.loads(out.value.decode())

    # ---- nodes ----
    def create_node(self, doc):
        b = json.dumps(doc).encode()
        self._ck(self.L.swp_sched_create_or_update_node(self.h, b, len(b)))
        self.idx_to_id[self.e.intern(abi.SPACE_NODE_ID, doc["ID"])] = doc["ID"]

    update_node = create_node

    def delete_node(self, nid):
        b = _b(nid)
        self._ck(self.L.swp_sched_delete_node(self.h, b, len(b)))

    def node_info(self, nid):
        b = _b(nid)
        out = C.c_char_p()
        rc = self.L.swp_sched_node_info(self.h, b, len(b), C.byref(out))
        if rc == abi.SWP_ENOTFOUND:
            return None   # errNodeNotFound
        self._ck(rc)
        return json.loads(out.value.decode())

    def node_index(self, nid):
        return self.e.intern(abi.SPACE_NODE_ID, nid)

    # ---- services / clock ----
    def update_volume(self, doc):
        """EventUpdateVolume (scheduler.go:200-213): taken once the plugin has created the volume (VolumeInfo.VolumeID)."""
        b = _b(json.dumps(doc))
        self._ck(self.L.swp_sched_update_volume(self.h, b, len(b)))

    def volume_info(self, vid):
        b = _b(vid)
        out = C.c_char_p()
        rc = self.L.swp_sched_volume_info(self.h, b, len(b), C.byref(out))
        if rc == abi.SWP_ENOTFOUND:
            return None
        self._ck(rc)
        return json.loads(out.value.decode())

    def free_volumes(self):
        """volumeSet.freeVolumes (volumes.go:181-221), what the reference's tick defers (scheduler.go:501): [{"VolumeID", "NodeIDs"}] — the
        publish statuses to move from PUBLISHED to PENDING_NODE_UNPUBLISH in the caller's store."""
        out = C.c_char_p()
        self._ck(self.L.swp_sched_free_volumes(self.h, C.byref(out)))
        return json.loads(out.value.decode())

    def set_service(self, sid, spec_version=None):
        b = _b(sid)
        self._ck(self.L.swp_sched_set_service(self.h, b, len(b), 0 if spec_version is None else 1, int(spec_version or 0)))

    def delete_service(self, sid):
        b = _b(sid)
        self._ck(self.L.swp_sched_delete_service(self.h, b, len(b)))

    def advance(self, seconds):
        self._ck(self.L.swp_sched_advance(self.h, int(seconds * self.SECOND)))

    def counts(self):
        """What the scheduler holds: tasks, queued tasks, rejectable decisions, task templates (swp_sched_counts)."""
        out = (C.c_uint64 * 4)()
        self._ck(self.L.swp_sched_counts(self.h, out))
        return {"tasks": out[0], "queued": out[1], "decisions": out[2], "templates": out[3]}

    # ---- task events ----
    def create_task(self, t):
        return self._doc_call(self.L.swp_sched_create_task, t)

    def setup_task(self, t):
        return self._doc_call(self.L.swp_sched_setup_task, t)

    def update_task(self, t):
        return self._doc_call(self.L.swp_sched_update_task, t)

    def delete_task(self, t):
        return self._doc_call(self.L.swp_sched_delete_task, t)

    # ---- the path ----
    def tick(self):
        return self._json_result(self.L.swp_sched_tick)

    def reject_decision(self, tid):
        """The failed half of applySchedulingDecisions (scheduler.go:472-487): undo the decision of the last tick for `tid`."""
        b = _b(tid)
        found = C.c_int(0)
        self._ck(self.L.swp_sched_reject_decision(self.h, b, len(b), C.byref(found)))
        return bool(found.value)

    def commit_plan(self, max_changes=0):
        """The last tick's decisions in commit order: grouped by node with the node's Meta.Version, cut into transactions
        (applySchedulingDecisions, scheduler.go:490-643; 200 changes per store transaction)."""
        return self._json_result(self.L.swp_sched_commit_plan, max_changes)

    def reject_decisions(self, tids):
        b = _b(json.dumps(list(tids)))
        n = C.c_uint32(0)
        self._ck(self.L.swp_sched_reject_decisions(self.h, b, len(b), C.byref(n)))
        return n.value

    def reject_node(self, nid):
        """Every decision of the last tick that landed on `nid` is undone (a failed Meta.Version check, scheduler.go:540-545)."""
        b = _b(nid)
        n = C.c_uint32(0)
        self._ck(self.L.swp_sched_reject_node(self.h, b, len(b), C.byref(n)))
        return n.value

    def process_preassigned(self):
        return self._json_result(self.L.swp_sched_process_preassigned)

    def task_desc(self, t):
        """One swp_task_desc as a numpy record array of length 1 (Pipeline.SetTask, pipeline.go:76-81)."""
        b = json.dumps(t).encode()
        d = abi.TaskDesc()
        self._ck(self.L.swp_sched_task_desc(self.h, b, len(b), C.byref(d)))
        return np.frombuffer(bytes(d), dtype=abi.TASK_DTYPE).copy()

    def constraint_set(self, exprs):
        """ConstraintFilter.SetTask for a list of expressions: predicate-set id, 0 when empty / unparsable."""
        b = json.dumps(list(exprs)).encode()
        out = C.c_uint32()
        self._ck(self.L.swp_sched_constraint_set(self.h, b, len(b), C.byref(out)))
        return out.value

    def explain(self, hist):
        """Pipeline.Explain (pipeline.go:84-103) for one per-filter failure histogram."""
        return _explain_with(self.L, hist)

    def enforce(self, node_docs, tasks_by_node, services=None):
        b = json.dumps({"nodes": list(node_docs), "tasks_by_node": {k: list(v) for k, v in tasks_by_node.items()}, "services": services or {}}).encode()
        return self._json_result(self.L.swp_sched_enforce, b, len(b))


# ---- pure string helpers (no engine; usable on CPU) ----
def parse_constraints(exprs, lib_path=None):
    """constraint.Parse → [(key, op, value)] or None."""
    L = abi.load_library(lib_path)
    b = json.dumps(list(exprs)).encode()
    out = C.c_char_p()
    rc = L.swp_constraint_parse(b, len(b), C.byref(out))
    if rc != 0:
        return None
    return [tuple(x) for x in json.loads(out.value.decode())]


def key_equal_fold(a, b, lib_path=None):
    L = abi.load_library(lib_path)
    a, b = a.encode(), b.encode()
    return bool(L.swp_key_equal_fold(a, len(a), b, len(b)))


def _explain_with(L, hist):
    h = (C.c_uint32 * abi.NFILTERS)(*[int(x) for x in hist])
    buf = C.create_string_buffer(1024)
    L.swp_explain(h, buf, len(buf))
    return buf.value.decode()


def explain(hist, lib_path=None):
    return _explain_with(abi.load_library(lib_path), hist)


def parse_ip(s, lib_path=None):
    L = abi.load_library(lib_path)
    b = s.encode()
    out = (C.c_uint8 * 16)()
    v4 = C.c_int(0)
    if not L.swp_parse_ip(b, len(b), out, C.byref(v4)):
        return None
    return bytes(out), bool(v4.value)
