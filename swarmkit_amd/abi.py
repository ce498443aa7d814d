"""ctypes binding of include/swp.h (libswp.so). Mirrors the header 1:1; no placement logic here."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "swarmkit_amd", "csrc")
LIB_PATH = os.environ.get("SWP_LIB_PATH") or os.path.join(ROOT, "swarmkit_amd", "lib", "libswp.so")

SWP_OK, SWP_EINVAL, SWP_ENOTFOUND, SWP_ENOMEM, SWP_EHIP, SWP_EUNSUPPORTED, SWP_ERANGE, SWP_ENODEVICE = 0, -1, -2, -3, -4, -5, -6, -7
(SPACE_NODE_ID, SPACE_SERVICE, SPACE_LABEL_KEY, SPACE_FOLDED, SPACE_OS, SPACE_ARCH, SPACE_PLUGIN, SPACE_RAW, SPACE_GENERIC_KIND) = range(9)
NODE_READY, NODE_HAS_DESC, NODE_HAS_PLATFORM, NODE_HAS_ENGINE = 0x1, 0x2, 0x4, 0x8
NODE_HAS_LABELS, NODE_HAS_ELABELS, NODE_MANAGER, NODE_HAS_LOGPLUG, NODE_IP_VALID, NODE_IP_V4 = 0x10, 0x20, 0x40, 0x80, 0x100, 0x200
(CK_NODE_ID, CK_HOSTNAME, CK_IP, CK_ROLE, CK_PLATFORM_OS, CK_PLATFORM_ARCH, CK_NODE_LABEL, CK_ENGINE_LABEL, CK_INVALID) = range(9)
OP_EQ, OP_NE = 0, 1
IP_SINGLE, IP_CIDR, IP_MALFORMED = 0, 1, 2
TASK_RES_ENABLED, TASK_UNCOUNTED = 0x1, 0x2
CFG_PROFILE = 1
NFILTERS = 8


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("resolver_threads", C.c_uint32), ("flags", C.c_uint32),
                ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class NodeRow(C.Structure):
    _fields_ = [("node", C.c_uint32), ("flags", C.c_uint32), ("cpu", C.c_int64), ("mem", C.c_int64), ("total", C.c_uint32),
                ("os", C.c_uint32), ("arch", C.c_uint32), ("os_fold", C.c_uint32), ("arch_fold", C.c_uint32),
                ("hostname_fold", C.c_uint32), ("id_fold", C.c_uint32), ("reserved", C.c_uint32), ("ip", C.c_uint8 * 16),
                ("version", C.c_uint64)]


class KV(C.Structure):
    _fields_ = [("key", C.c_uint32), ("value", C.c_uint32), ("raw", C.c_uint32)]


class Constraint(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("op", C.c_uint32), ("key", C.c_uint32), ("value", C.c_uint32), ("ip", C.c_uint8 * 16),
                ("ip_kind", C.c_uint32), ("prefix_len", C.c_uint32), ("ip_is_v4", C.c_uint32), ("reserved", C.c_uint32)]


class Platform(C.Structure):
    _fields_ = [("os", C.c_uint32), ("arch", C.c_uint32)]


class Port(C.Structure):
    _fields_ = [("protocol", C.c_uint32), ("port", C.c_uint32)]


class Spread(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("key", C.c_uint32)]


class TaskDesc(C.Structure):
    _fields_ = [("service", C.c_uint32), ("flags", C.c_uint32), ("cpu", C.c_int64), ("mem", C.c_int64),
                ("constraint_set", C.c_uint32), ("platform_set", C.c_uint32), ("plugin_set", C.c_uint32), ("port_set", C.c_uint32),
                ("max_replicas", C.c_uint64), ("spec_version", C.c_uint64), ("spread_set", C.c_uint32), ("generic_set", C.c_uint32)]


class Placement(C.Structure):
    _fields_ = [("node", C.c_uint32), ("service", C.c_uint32), ("cpu", C.c_int64), ("mem", C.c_int64), ("port_set", C.c_uint32),
                ("counted", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("tasks", C.c_uint64), ("placed", C.c_uint64), ("infeasible", C.c_uint64),
                ("pair_evals", C.c_uint64), ("verify_retries", C.c_uint64), ("slow_path_tasks", C.c_uint64), ("rebase_events", C.c_uint64),
                ("generic_tasks", C.c_uint64), ("resolver_spins", C.c_uint64),
                ("n_nodes", C.c_uint32), ("n_words", C.c_uint32), ("last_windows", C.c_uint32), ("last_static_classes", C.c_uint32),
                ("ms_classes", C.c_float), ("ms_scan", C.c_float), ("ms_resolve", C.c_float), ("ms_explain", C.c_float), ("ms_total", C.c_float),
                ("scan_launches", C.c_uint32), ("resolve_launches", C.c_uint32), ("last_resolver", C.c_uint32),
                ("ms_propose", C.c_float), ("ms_apply", C.c_float), ("propose_launches", C.c_uint32), ("propose_tasks", C.c_uint32),
                ("waterfill_tasks", C.c_uint32), ("scan_tasks", C.c_uint32)]


# numpy views of the POD structs, for bulk construction
TASK_DTYPE = np.dtype([("service", "<u4"), ("flags", "<u4"), ("cpu", "<i8"), ("mem", "<i8"), ("constraint_set", "<u4"),
                       ("platform_set", "<u4"), ("plugin_set", "<u4"), ("port_set", "<u4"), ("max_replicas", "<u8"),
                       ("spec_version", "<u8"), ("spread_set", "<u4"), ("generic_set", "<u4")])
PLACEMENT_DTYPE = np.dtype([("node", "<u4"), ("service", "<u4"), ("cpu", "<i8"), ("mem", "<i8"), ("port_set", "<u4"), ("counted", "<u4")])
ENF_NODE_DTYPE = np.dtype([("node", "<u4"), ("first_task", "<u4"), ("n_tasks", "<u4"), ("reserved", "<u4"), ("cpu", "<i8"), ("mem", "<i8")])
ENF_TASK_DTYPE = np.dtype([("cpu", "<i8"), ("mem", "<i8"), ("constraint_set", "<u4"), ("flags", "<u4"), ("desired_state", "<u4"), ("state", "<u4")])
ENF_RESERVATIONS = 1
NODE_DYNAMIC_DTYPE = np.dtype([("node", "<u4"), ("flags", "<u4"), ("cpu", "<i8"), ("mem", "<i8"), ("total", "<u4"), ("reserved", "<u4")])
NODE_ROW_DTYPE = np.dtype([("node", "<u4"), ("flags", "<u4"), ("cpu", "<i8"), ("mem", "<i8"), ("total", "<u4"), ("os", "<u4"), ("arch", "<u4"), ("os_fold", "<u4"),
                           ("arch_fold", "<u4"), ("hostname_fold", "<u4"), ("id_fold", "<u4"), ("reserved", "<u4"), ("ip", "u1", (16,)), ("version", "<u8")])
assert NODE_DYNAMIC_DTYPE.itemsize == 32 and NODE_ROW_DTYPE.itemsize == C.sizeof(NodeRow)
# node-range shards (include/swp.h): swp_proposal / swp_shard_pick
SHARD_CAND = 4
PROPOSAL_DTYPE = np.dtype([("level", "<u4"), ("n_cand", "<u4"), ("word", "<u4", (SHARD_CAND,)), ("bits", "<u8", (SHARD_CAND,)), ("exc_hi", "<u8"),
                           ("exc_lo", "<u8"), ("exc_entry", "<u4"), ("flags", "<u4")])
PICK_DTYPE = np.dtype([("shard", "<i4"), ("node", "<u4"), ("entry", "<u4"), ("reserved", "<u4")])
assert PROPOSAL_DTYPE.itemsize == 80 and PICK_DTYPE.itemsize == 16
assert ENF_NODE_DTYPE.itemsize == 32 and ENF_TASK_DTYPE.itemsize == 32
assert TASK_DTYPE.itemsize == C.sizeof(TaskDesc) == 64
assert PLACEMENT_DTYPE.itemsize == C.sizeof(Placement) == 32

class Generic(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("value", C.c_int64)]


EXPORTS = [
    "swp_generic_set", "swp_node_set_generic", "swp_node_get_generic",
    "swp_node_set_csi", "swp_volume_upsert", "swp_volume_set_usage", "swp_volume_get_usage", "swp_mount_set", "swp_choose_volumes", "swp_batch_attachments", "swp_schedule_groups_volumes", "swp_batch_prepare_templates",
    "swp_create", "swp_destroy", "swp_reset", "swp_intern", "swp_intern_lookup", "swp_node_upsert", "swp_node_update_dynamic",
    "swp_node_remove", "swp_node_get", "swp_node_set_svc_count", "swp_node_get_svc_count", "swp_node_set_failures", "swp_node_port",
    "swp_constraint_set", "swp_platform_set", "swp_plugin_set", "swp_port_set", "swp_spread_set", "swp_schedule_groups", "swp_schedule_batch", "swp_batch_prepare",
    "swp_batch_run", "swp_batch_fetch", "swp_batch_results", "swp_batch_free", "swp_state_save", "swp_state_restore", "swp_commit", "swp_check_node", "swp_enforce", "swp_node_matches",
    "swp_stats", "swp_strerror", "swp_last_error", "swp_abi_check", "swp_node_update_dynamic_many", "swp_node_get_many", "swp_shardset_create",
    "swp_shard_begin", "swp_shard_propose", "swp_shard_merge", "swp_shard_commit", "swp_shard_end", "swp_shard_run", "swp_rccl_available", "swp_rccl_unique_id", "swp_rccl_init", "swp_rccl_finalize", "swp_shard_run_rank", "swp_shard_verdict",
    # include/swp_sched.h — the host layer above the engine
    "swp_sched_create", "swp_sched_destroy", "swp_sched_last_error", "swp_sched_create_or_update_node", "swp_sched_delete_node", "swp_sched_node_info",
    "swp_sched_set_service", "swp_sched_delete_service", "swp_sched_advance", "swp_sched_counts", "swp_sched_create_task", "swp_sched_setup_task", "swp_sched_update_task",
    "swp_sched_delete_task", "swp_sched_tick", "swp_sched_process_preassigned", "swp_sched_reject_decision", "swp_sched_commit_plan", "swp_sched_reject_decisions", "swp_sched_reject_node", "swp_sched_task_desc", "swp_sched_constraint_set", "swp_sched_enforce", "swp_sched_update_volume", "swp_sched_volume_info", "swp_sched_free_volumes",
    "swp_constraint_parse", "swp_key_equal_fold", "swp_explain", "swp_parse_ip",
]


class Unsupported(NotImplementedError):
    """The task/feature stays on the reference's own Go path (SWP_EUNSUPPORTED)."""


class SwpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"swp error {code}: {msg}")
        self.code = code
        self.msg = msg


def build_library(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU). In-tree output: swarmkit_amd/lib/libswp.so."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile"] \
        + [os.path.join(ROOT, "include", h) for h in ("swp.h", "swp_sched.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        if os.path.exists(LIB_PATH):
            return LIB_PATH
        raise RuntimeError("hipcc not found and no prebuilt libswp.so")
    r = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libswp.so build failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_libs = {}


def load_library(path=None):
    """libswp.so (default) or another library with the same exports (tests/_build/libswpfake.so: the host layer's CPU test double)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        build_library()
    L = C.CDLL(path)
    vp, u32, u64, i32, i64, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64, C.c_char_p
    P = C.POINTER
    sig = {
        "swp_create": ([P(Config), P(vp)], C.c_int),
        "swp_destroy": ([vp], None),
        "swp_shardset_create": ([P(Config), P(i32), u32, u32, P(vp)], C.c_int),
        "swp_reset": ([vp, u32], C.c_int),
        "swp_intern": ([vp, C.c_int, cp, C.c_size_t, P(u32)], C.c_int),
        "swp_intern_lookup": ([vp, C.c_int, u32, cp, C.c_size_t], C.c_int),
        "swp_node_upsert": ([vp, P(NodeRow), P(KV), u32, P(KV), u32, P(u32), u32], C.c_int),
        "swp_node_update_dynamic": ([vp, u32, u32, i64, i64, u32], C.c_int),
        "swp_node_remove": ([vp, u32], C.c_int),
        "swp_node_update_dynamic_many": ([vp, vp, u32], C.c_int),
        "swp_node_get_many": ([vp, vp, u32, vp], C.c_int),
        "swp_node_get": ([vp, u32, P(NodeRow)], C.c_int),
        "swp_node_set_svc_count": ([vp, u32, u32, u32], C.c_int),
        "swp_node_get_svc_count": ([vp, u32, u32, P(u32)], C.c_int),
        "swp_node_set_failures": ([vp, u32, u32, u64, u32], C.c_int),
        "swp_node_port": ([vp, u32, u32, u32, C.c_int], C.c_int),
        "swp_constraint_set": ([vp, P(Constraint), u32, P(u32)], C.c_int),
        "swp_platform_set": ([vp, P(Platform), u32, P(u32)], C.c_int),
        "swp_plugin_set": ([vp, P(u32), u32, u32, P(u32)], C.c_int),
        "swp_port_set": ([vp, P(Port), u32, P(u32)], C.c_int),
        "swp_spread_set": ([vp, P(Spread), u32, P(u32)], C.c_int),
        "swp_generic_set": ([vp, P(Generic), u32, P(u32)], C.c_int),
        "swp_node_set_generic": ([vp, u32, P(Generic), u32], C.c_int),
        "swp_node_get_generic": ([vp, u32, u32, P(C.c_int64)], C.c_int),
        "swp_schedule_groups": ([vp, vp, vp, u32, vp, vp], C.c_int),
        "swp_schedule_batch": ([vp, vp, u32, vp, vp], C.c_int),
        "swp_batch_prepare": ([vp, vp, u32, P(vp)], C.c_int),
        "swp_batch_prepare_templates": ([vp, vp, u32, vp, u32, P(vp)], C.c_int),
        "swp_batch_run": ([vp, vp], C.c_int),
        "swp_batch_fetch": ([vp, vp, vp, vp], C.c_int),
        "swp_batch_results": ([vp, vp, vp, vp], C.c_int),
        "swp_batch_free": ([vp, vp], None),
        "swp_state_save": ([vp], C.c_int),
        "swp_state_restore": ([vp], C.c_int),
        "swp_commit": ([vp, vp, u32, C.c_int], C.c_int),
        "swp_shard_begin": ([vp, vp], C.c_int),
        "swp_shard_propose": ([vp, vp, u32, u32, vp], C.c_int),
        "swp_shard_merge": ([P(vp), P(u32), u32, u32, vp, P(u32)], C.c_int),
        "swp_shard_commit": ([vp, vp, u32, vp, u32], C.c_int),
        "swp_shard_end": ([vp, vp, vp, vp], C.c_int),
        "swp_shard_run": ([vp, vp, u32, u32, vp, vp, vp], C.c_int),
        "swp_rccl_available": ([vp], C.c_int),
        "swp_shard_verdict": ([vp, u32, vp], C.c_int),
        "swp_rccl_unique_id": ([vp, vp], C.c_int),
        "swp_rccl_init": ([vp, vp, u32, u32], C.c_int),
        "swp_rccl_finalize": ([vp], C.c_int),
        "swp_shard_run_rank": ([vp, vp, vp, u32, vp, vp], C.c_int),
        "swp_check_node": ([vp, P(TaskDesc), u32, P(i32)], C.c_int),
        "swp_enforce": ([vp, vp, u32, vp, u32, vp], C.c_int),
        "swp_node_matches": ([vp, vp, u32, vp, u32], C.c_int),
        "swp_stats": ([vp, P(Stats)], C.c_int),
        "swp_strerror": ([C.c_int], cp),
        "swp_last_error": ([vp], cp),
        "swp_abi_check": ([P(u32), u32], C.c_int),
    }
    sz = C.c_size_t
    PP = P(cp)
    sig.update({   # include/swp_sched.h — the host layer above the engine ABI
        "swp_sched_create": ([vp, P(vp)], C.c_int),
        "swp_sched_destroy": ([vp], None),
        "swp_sched_last_error": ([vp], cp),
        "swp_sched_create_or_update_node": ([vp, cp, sz], C.c_int),
        "swp_sched_delete_node": ([vp, cp, sz], C.c_int),
        "swp_sched_node_info": ([vp, cp, sz, PP], C.c_int),
        "swp_sched_set_service": ([vp, cp, sz, C.c_int, u64], C.c_int),
        "swp_sched_delete_service": ([vp, cp, sz], C.c_int),
        "swp_sched_advance": ([vp, i64], C.c_int),
        "swp_sched_counts": ([vp, C.POINTER(C.c_uint64)], C.c_int),
        "swp_sched_create_task": ([vp, cp, sz, P(C.c_int)], C.c_int),
        "swp_sched_setup_task": ([vp, cp, sz, P(C.c_int)], C.c_int),
        "swp_sched_update_task": ([vp, cp, sz, P(C.c_int)], C.c_int),
        "swp_sched_delete_task": ([vp, cp, sz, P(C.c_int)], C.c_int),
        "swp_sched_tick": ([vp, PP], C.c_int),
        "swp_sched_process_preassigned": ([vp, PP], C.c_int),
        "swp_sched_reject_decision": ([vp, cp, sz, P(C.c_int)], C.c_int),
        "swp_sched_commit_plan": ([vp, u32, PP], C.c_int),
        "swp_sched_reject_decisions": ([vp, cp, sz, P(u32)], C.c_int),
        "swp_sched_reject_node": ([vp, cp, sz, P(u32)], C.c_int),
        "swp_sched_task_desc": ([vp, cp, sz, P(TaskDesc)], C.c_int),
        "swp_sched_constraint_set": ([vp, cp, sz, P(u32)], C.c_int),
        "swp_sched_enforce": ([vp, cp, sz, PP], C.c_int),
        "swp_constraint_parse": ([cp, sz, PP], C.c_int),
        "swp_key_equal_fold": ([cp, sz, cp, sz], C.c_int),
        "swp_explain": ([P(u32), cp, sz], C.c_int),
        "swp_parse_ip": ([cp, sz, P(C.c_uint8), P(C.c_int)], C.c_int),
    })
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    sizes = (u32 * 16)()
    n = L.swp_abi_check(sizes, 16)
    want = [C.sizeof(x) for x in (Config, NodeRow, KV, Constraint, Platform, Port, TaskDesc, Placement, Stats, Spread, Generic)]
    if list(sizes[:n]) != want:
        raise RuntimeError(f"ABI struct size mismatch: lib {list(sizes[:n])} vs binding {want}")
    _libs[path] = L
    return L


def shard_run(batches, want_hist=True, fold=True):
    """swp_shard_run: the node-range shards of ONE process with the rounds on the device. batches[g] = the Batch prepared on
    engine g from the same task list. Returns (shard int32[T], shard-local node int32[T], hist uint32[T, 8] or None)."""
    G, T = len(batches), batches[0].n
    L = batches[0].eng.L
    eh = (C.c_void_p * G)(*[b.eng.h for b in batches])
    bh = (C.c_void_p * G)(*[b.h for b in batches])
    shard = np.empty(T, dtype=np.int32)
    node = np.empty(T, dtype=np.int32)
    hist = np.zeros((T, NFILTERS), dtype=np.uint32) if want_hist else None
    rc = L.swp_shard_run(eh, bh, G, 0 if fold else 1, shard.ctypes.data, node.ctypes.data, hist.ctypes.data if want_hist else None)
    if rc != 0:
        batches[0].eng._ck(rc)
    return shard, node, hist


def shard_merge(proposals, first_nodes, lib_path=None):
    """swp_shard_merge: proposals = one PROPOSAL_DTYPE array per shard (range order, same length), first_nodes = the global
    index of each shard's first node. Returns the decided prefix as a PICK_DTYPE array (length >= 1 for a non-empty block)."""
    L = load_library(lib_path)
    G, count = len(proposals), len(proposals[0])
    props = [np.ascontiguousarray(p, dtype=PROPOSAL_DTYPE) for p in proposals]
    ptrs = (C.c_void_p * G)(*[p.ctypes.data for p in props])
    firsts = (C.c_uint32 * G)(*[int(x) for x in first_nodes])
    picks = np.zeros(count, dtype=PICK_DTYPE)
    acc = C.c_uint32(0)
    rc = L.swp_shard_merge(ptrs, firsts, G, count, picks.ctypes.data, C.byref(acc))
    if rc != 0:
        raise SwpError(rc, "swp_shard_merge")
    return picks[:acc.value]


class Batch:
    def __init__(self, eng, handle, n):
        self.eng, self.h, self.n = eng, handle, n

    def run(self):
        self.eng._ck(self.eng.L.swp_batch_run(self.eng.h, self.h))

    def fetch(self, want_hist=True):
        out = np.empty(self.n, dtype=np.int32)
        hist = np.zeros((self.n, NFILTERS), dtype=np.uint32) if want_hist else None
        self.eng._ck(self.eng.L.swp_batch_fetch(self.eng.h, self.h, out.ctypes.data, hist.ctypes.data if want_hist else None))
        return out, hist

    def attachments(self, tasks=None):
        """swp_batch_attachments: [len(tasks), SWP_MAX_MOUNTS] volume indices (NO_VOLUME beyond a task's mounts), after fetch / results."""
        tasks = np.arange(self.n, dtype=np.uint32) if tasks is None else np.ascontiguousarray(tasks, dtype=np.uint32)
        out = np.empty((len(tasks), 8), dtype=np.uint32)
        self.eng.L.swp_batch_attachments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        self.eng._ck(self.eng.L.swp_batch_attachments(self.eng.h, self.h, tasks.ctypes.data, len(tasks), out.ctypes.data))
        return out

    def results(self, want_hist=True):
        out = np.empty(self.n, dtype=np.int32)
        hist = np.zeros((self.n, NFILTERS), dtype=np.uint32) if want_hist else None
        self.eng._ck(self.eng.L.swp_batch_results(self.eng.h, self.h, out.ctypes.data, hist.ctypes.data if want_hist else None))
        return out, hist

    # ---- node-range shard protocol (include/swp.h "node-range shards") ----
    def shard_begin(self):
        self.eng._ck(self.eng.L.swp_shard_begin(self.eng.h, self.h))

    def shard_propose(self, j0, count):
        out = np.empty(count, dtype=PROPOSAL_DTYPE)
        self.eng._ck(self.eng.L.swp_shard_propose(self.eng.h, self.h, j0, count, out.ctypes.data))
        return out

    def shard_commit(self, j0, picks):
        picks = np.ascontiguousarray(picks, dtype=PICK_DTYPE)
        self.eng._ck(self.eng.L.swp_shard_commit(self.eng.h, self.h, j0, picks.ctypes.data, len(picks)))

    def shard_end(self, want_hist=True):
        out = np.empty(self.n, dtype=np.int32)
        hist = np.zeros((self.n, NFILTERS), dtype=np.uint32) if want_hist else None
        self.eng._ck(self.eng.L.swp_shard_end(self.eng.h, self.h, out.ctypes.data, hist.ctypes.data if want_hist else None))
        return out, hist

    def free(self):
        if self.h:
            self.eng.L.swp_batch_free(self.eng.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One swp_engine handle. Raises SwpError(SWP_ENODEVICE) when no gfx950 is present: there is no CPU path."""

    def __init__(self, device=0, resolver_threads=0, profile=False, lib_path=None, shard_rank=0, shard_count=0,
                 shards=None, nodes_per_shard=None, devices=None):
        """shards=G, nodes_per_shard=cap: a shard SET (swp_shardset_create) — G engines of this process behind the one handle, on
        `devices` (one ordinal per shard; default: all on `device`). SWP_SHARDSET="G:cap" in the environment turns every engine
        created without shard arguments into such a set (how the whole scenario suite is run over node-range shards)."""
        self.L = load_library(lib_path)
        cfg = Config(device=device, resolver_threads=resolver_threads, flags=CFG_PROFILE if profile else 0,
                     shard_rank=shard_rank, shard_count=shard_count)
        env = os.environ.get("SWP_SHARDSET")
        if shards is None and env and not shard_count and lib_path is None:
            shards, nodes_per_shard = (int(x) for x in env.split(":"))
        h = C.c_void_p()
        self.shards = int(shards or 0)
        if self.shards:
            if not nodes_per_shard:
                raise ValueError("a shard set needs nodes_per_shard (the node slots of every range)")
            self.nodes_per_shard = int(nodes_per_shard)
            dev = (C.c_int32 * self.shards)(*[int(d) for d in devices]) if devices is not None else None
            rc = self.L.swp_shardset_create(C.byref(cfg), dev, self.shards, self.nodes_per_shard, C.byref(h))
        else:
            rc = self.L.swp_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise SwpError(rc, self.L.swp_last_error(None).decode())
        self.h = h

    def close(self):
        if self.h:
            self.L.swp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise SwpError(rc, self.L.swp_strerror(rc).decode() + ": " + self.L.swp_last_error(self.h).decode())

    def reset(self, hint=0):
        self._ck(self.L.swp_reset(self.h, hint))

    def intern(self, space, s):
        b = s.encode() if isinstance(s, str) else s
        out = C.c_uint32()
        self._ck(self.L.swp_intern(self.h, space, b, len(b), C.byref(out)))
        return out.value

    def node_upsert(self, row, labels=(), engine_labels=(), plugins=()):
        la = (KV * max(1, len(labels)))(*[KV(*kv) for kv in labels])
        ea = (KV * max(1, len(engine_labels)))(*[KV(*kv) for kv in engine_labels])
        pa = (C.c_uint32 * max(1, len(plugins)))(*plugins)
        self._ck(self.L.swp_node_upsert(self.h, C.byref(row), la, len(labels), ea, len(engine_labels), pa, len(plugins)))

    def node_update_dynamic(self, node, flags, cpu, mem, total):
        self._ck(self.L.swp_node_update_dynamic(self.h, node, flags, cpu, mem, total))

    def node_update_dynamic_many(self, rows):
        """rows: NODE_DYNAMIC_DTYPE array (swp_node_update_dynamic_many: a burst of node events in one call)."""
        rows = np.ascontiguousarray(rows, dtype=NODE_DYNAMIC_DTYPE)
        self._ck(self.L.swp_node_update_dynamic_many(self.h, rows.ctypes.data, len(rows)))

    def node_get_many(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
        out = np.zeros(len(nodes), dtype=NODE_ROW_DTYPE)
        self._ck(self.L.swp_node_get_many(self.h, nodes.ctypes.data, len(nodes), out.ctypes.data))
        return out

    def node_remove(self, node):
        self._ck(self.L.swp_node_remove(self.h, node))

    def node_get(self, node):
        row = NodeRow()
        rc = self.L.swp_node_get(self.h, node, C.byref(row))
        if rc == SWP_ENOTFOUND:
            return None
        self._ck(rc)
        return row

    def node_set_svc_count(self, node, service, count):
        self._ck(self.L.swp_node_set_svc_count(self.h, node, service, count))

    def node_get_svc_count(self, node, service):
        out = C.c_uint32()
        self._ck(self.L.swp_node_get_svc_count(self.h, node, service, C.byref(out)))
        return out.value

    def node_set_failures(self, node, service, spec_version, count):
        self._ck(self.L.swp_node_set_failures(self.h, node, service, spec_version, count))

    def node_port(self, node, protocol, port, set_=True):
        self._ck(self.L.swp_node_port(self.h, node, protocol, port, 1 if set_ else 0))

    def volume_get_usage(self, volume):
        """swp_volume_get_usage: (n_tasks, n_writers, pin). (On a shard set the call also checks that every shard holds the same numbers.)"""
        u = (C.c_uint32 * 4)()
        self.L.swp_volume_get_usage.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        self._ck(self.L.swp_volume_get_usage(self.h, volume, u))
        return u[0], u[1], u[2]

    def constraint_set(self, cs):
        arr = (Constraint * max(1, len(cs)))(*cs)
        out = C.c_uint32()
        self._ck(self.L.swp_constraint_set(self.h, arr, len(cs), C.byref(out)))
        return out.value

    def platform_set(self, ps):
        arr = (Platform * max(1, len(ps)))(*[Platform(o, a) for o, a in ps])
        out = C.c_uint32()
        self._ck(self.L.swp_platform_set(self.h, arr, len(ps), C.byref(out)))
        return out.value

    def plugin_set(self, required, log_plugin=0):
        arr = (C.c_uint32 * max(1, len(required)))(*required)
        out = C.c_uint32()
        self._ck(self.L.swp_plugin_set(self.h, arr, len(required), log_plugin, C.byref(out)))
        return out.value

    def port_set(self, ports):
        arr = (Port * max(1, len(ports)))(*[Port(p, q) for p, q in ports])
        out = C.c_uint32()
        self._ck(self.L.swp_port_set(self.h, arr, len(ports), C.byref(out)))
        return out.value

    def generic_set(self, items):
        """items: (GENERIC_KIND id, value) pairs — the task's Discrete reservations."""
        arr = (Generic * max(1, len(items)))(*[Generic(k, 0, v) for k, v in items])
        out = C.c_uint32()
        self._ck(self.L.swp_generic_set(self.h, arr, len(items), C.byref(out)))
        return out.value

    def node_set_generic(self, node, counts):
        """counts: (GENERIC_KIND id, count) pairs of the node's available generic resources (replaces)."""
        arr = (Generic * max(1, len(counts)))(*[Generic(k, 0, v) for k, v in counts])
        self._ck(self.L.swp_node_set_generic(self.h, node, arr, len(counts)))

    def node_get_generic(self, node, kind):
        out = C.c_int64()
        self._ck(self.L.swp_node_get_generic(self.h, node, kind, C.byref(out)))
        return out.value

    # ---- the rank variant of the node-range shards: one engine per process / GPU, RCCL between them ----
    def rccl_available(self):
        """librccl.so loads with every symbol the engine uses (what the ranks exchange before any of them calls rccl_init)."""
        return self.L.swp_rccl_available(self.h) == 0

    def rccl_unique_id(self):
        buf = (C.c_uint8 * 128)()
        self._ck(self.L.swp_rccl_unique_id(self.h, buf))
        return bytes(buf)

    def rccl_init(self, uid, rank, n_ranks):
        buf = (C.c_uint8 * 128)(*uid)
        self._ck(self.L.swp_rccl_init(self.h, buf, rank, n_ranks))

    def rccl_finalize(self):
        self._ck(self.L.swp_rccl_finalize(self.h))

    def shard_run_rank(self, batch, shard_nodes, want_hist=True, fold=True):
        """swp_shard_run_rank: this rank's share of a sharded batch. Returns (local node int32[T] (-1 elsewhere), hist or None)."""
        nodes = np.ascontiguousarray(shard_nodes, dtype=np.uint32)
        out = np.empty(batch.n, dtype=np.int32)
        hist = np.zeros((batch.n, NFILTERS), dtype=np.uint32) if want_hist else None
        self._ck(self.L.swp_shard_run_rank(self.h, batch.h, nodes.ctypes.data, 0 if fold else 1, out.ctypes.data, hist.ctypes.data if want_hist else None))
        return out, hist

    def spread_set(self, levels):
        arr = (Spread * max(1, len(levels)))(*[Spread(k, key) for k, key in levels])
        out = C.c_uint32()
        self._ck(self.L.swp_spread_set(self.h, arr, len(levels), C.byref(out)))
        return out.value

    def schedule_groups(self, groups, sizes):
        """groups: TASK_DTYPE array (one descriptor per group), sizes: tasks per group.
        Returns (out_node int32[sum sizes], hist uint32[n_groups, 8])."""
        groups = np.ascontiguousarray(groups, dtype=TASK_DTYPE)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        out = np.empty(int(sizes.sum()), dtype=np.int32)
        hist = np.zeros((len(groups), NFILTERS), dtype=np.uint32)
        self._ck(self.L.swp_schedule_groups(self.h, groups.ctypes.data, sizes.ctypes.data, len(groups), out.ctypes.data, hist.ctypes.data))
        return out, hist

    def enforce(self, nodes, tasks):
        """constraintenforcer.rejectNoncompliantTasks over many nodes. nodes: ENF_NODE_DTYPE, tasks: ENF_TASK_DTYPE
        (grouped by node, store order). Returns uint8[len(tasks)]: 1 = the task would be REJECTED."""
        nodes = np.ascontiguousarray(nodes, dtype=ENF_NODE_DTYPE)
        tasks = np.ascontiguousarray(tasks, dtype=ENF_TASK_DTYPE)
        out = np.zeros(len(tasks), dtype=np.uint8)
        self._ck(self.L.swp_enforce(self.h, nodes.ctypes.data, len(nodes), tasks.ctypes.data, len(tasks), out.ctypes.data))
        return out

    def node_matches(self, constraint_sets):
        """constraint.NodeMatches for every (set, node) pair: uint64[len(sets), n_words] bitmaps (bit i of word w = node 64w+i)."""
        sets = np.ascontiguousarray(constraint_sets, dtype=np.uint32)
        nw = int(self.stats()["n_words"])
        out = np.zeros((len(sets), nw), dtype=np.uint64)
        if len(sets):
            self._ck(self.L.swp_node_matches(self.h, sets.ctypes.data, len(sets), out.ctypes.data, nw))
        return out

    def schedule_batch(self, tasks, want_hist=True):
        """tasks: numpy array of TASK_DTYPE. Returns (out_node int32[T], hist uint32[T,8] or None)."""
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        n = len(tasks)
        out = np.empty(n, dtype=np.int32)
        hist = np.zeros((n, NFILTERS), dtype=np.uint32) if want_hist else None
        self._ck(self.L.swp_schedule_batch(self.h, tasks.ctypes.data, n, out.ctypes.data, hist.ctypes.data if want_hist else None))
        return out, hist

    def batch_prepare(self, tasks):
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        h = C.c_void_p()
        self._ck(self.L.swp_batch_prepare(self.h, tasks.ctypes.data, len(tasks), C.byref(h)))
        return Batch(self, h, len(tasks))

    def batch_prepare_templates(self, templates, template_of_task):
        """swp_batch_prepare for tasks given as (templates, index per task): no per-task de-duplication pass."""
        templates = np.ascontiguousarray(templates, dtype=TASK_DTYPE)
        idx = np.ascontiguousarray(template_of_task, dtype=np.uint32)
        h = C.c_void_p()
        self._ck(self.L.swp_batch_prepare_templates(self.h, templates.ctypes.data, len(templates), idx.ctypes.data, len(idx), C.byref(h)))
        return Batch(self, h, len(idx))

    def state_save(self):
        self._ck(self.L.swp_state_save(self.h))

    def state_restore(self):
        self._ck(self.L.swp_state_restore(self.h))

    def commit(self, placements, add=True):
        p = np.ascontiguousarray(placements, dtype=PLACEMENT_DTYPE)
        self._ck(self.L.swp_commit(self.h, p.ctypes.data, len(p), 1 if add else 0))

    def check_node(self, task, node):
        t = np.ascontiguousarray(task, dtype=TASK_DTYPE).reshape(1)
        ff = C.c_int32()
        self._ck(self.L.swp_check_node(self.h, C.cast(t.ctypes.data, C.POINTER(TaskDesc)), node, C.byref(ff)))
        return ff.value

    def stats(self):
        s = Stats()
        self._ck(self.L.swp_stats(self.h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}
