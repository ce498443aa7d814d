"""TEST INFRASTRUCTURE: an independent Python twin of the product's host layer (swarmkit_amd/csrc/swp_sched.cpp, swp::Scheduler inside
libswp.so). The two are compared call by call on CPU against a scripted test double of the engine (tests/test_sched_cpu.py) and the
GPU scenario suites run through both (SWP_HOST=py selects this one: tests/conftest.py registers it with swarmkit_amd.host). Nothing in
swarmkit_amd/ imports this file.

Host-side mirror of manager/scheduler.Scheduler ABOVE the C ABI (include/swp.h).

What a cgo shim would do inside swarmkit, written in Python for the test bed:
it keeps the string-typed half of the nodeSet (api.Node docs, NodeInfo.Tasks, failure timestamps),
turns every mutator of nodeSet / NodeInfo into ABI calls, translates Filter.SetTask into
predicate-set registrations, and turns the engine's numeric answers back into scheduling decisions
(NodeID, Status.Err strings). It contains NO placement logic: which node a task lands on is decided
by the kernels behind swp_schedule_batch.

Mirrors (paths under /root/reference/manager/scheduler/):
  scheduler.go:254-396  createTask / updateTask / deleteTask / createOrUpdateNode
  scheduler.go:429-488  tick            (one-off tasks → swp_schedule_batch)
  scheduler.go:646-690  taskFitNode     (swp_check_node)
  scheduler.go:928-971  noSuitableNode  (Status.Err from the failure histogram, pipeline.go:84-103)
  nodeinfo.go:66-221    addTask / removeTask / taskFailed / countRecentFailures
"""
import ipaddress
import os
import re

import numpy as np

import pygeneric as gres
from swarmkit_amd import abi

# api/types.proto:510-539
NEW, PENDING, ASSIGNED, RUNNING, COMPLETE, SHUTDOWN, FAILED, REJECTED = 0, 64, 192, 512, 576, 640, 704, 768
NODE_READY, AVAIL_ACTIVE = 2, 0
MOUNT_VOLUME, MOUNT_CLUSTER = 1, 4
PUBLISH_HOST = 1

MONITOR_FAILURES = 5 * 60 * 1_000_000_000   # scheduler.go:19
MAX_FAILURES = 5                            # scheduler.go:23

_TASK_STATES = {"NEW": 0, "PENDING": 64, "ASSIGNED": 192, "ACCEPTED": 256, "PREPARING": 320, "READY": 384, "STARTING": 448,
                "RUNNING": 512, "COMPLETE": 576, "SHUTDOWN": 640, "FAILED": 704, "REJECTED": 768, "REMOVE": 800, "ORPHANED": 832}


Unsupported = abi.Unsupported   # the task/feature stays on the reference's own Go path (SWP_EUNSUPPORTED)


def _get(d, *path, default=None):
    for p in path:
        if d is None:
            return default
        d = d.get(p)
    return default if d is None else d


def _state(v, table=_TASK_STATES):
    if v is None:
        return 0
    return table[v] if isinstance(v, str) else int(v)


# ---- constraint.Parse (constraint.go:40-81) ---------------------------------------------------------
# key `^(?i)[a-z_][a-z0-9\-_.]+$`, value pattern constraint.go:23-26; under (?i) RE2 also folds
# U+212A (KELVIN SIGN) and U+017F (LONG S) onto k / s.
_KEY_RE = re.compile("[a-zA-Z_\u212a\u017f][a-zA-Z0-9\\-_.\u212a\u017f]+")
_VAL_RE = re.compile("[a-zA-Z0-9:\\-_\t\n\f\r .*()?+\\[\\]\\\\^$|/\u212a\u017f]+")
# strings.TrimSpace: Unicode White_Space
_GO_SPACE = "\t\n\v\f\r \x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000"


def parse_constraints(exprs):
    """Returns [(key, op, exp)] or None when constraint.Parse would return an error."""
    out = []
    for e in exprs:
        found = False
        for i, op in enumerate(("==", "!=")):
            at = e.find(op)
            if at < 0:
                continue
            key, val = e[:at].strip(_GO_SPACE), e[at + 2:].strip(_GO_SPACE)
            if not _KEY_RE.fullmatch(key) or not _VAL_RE.fullmatch(val):
                return None
            out.append((key, i, val))
            found = True
            break
        if not found:
            return None
    return out


def _fold_eq(a, b):
    """strings.EqualFold restricted to what a constraint KEY can contain (ASCII + K-sign + long-s)."""
    def canon(s):
        return s.replace("K", "k").replace("ſ", "s").lower()
    return canon(a) == canon(b)


def _parse_ip(s):
    """net.ParseIP → (16 bytes, is_v4) or None."""
    if "%" in s:
        return None
    try:
        ip = ipaddress.ip_address(s)
    except ValueError:
        return None
    if ip.version == 4:
        return b"\x00" * 10 + b"\xff\xff" + ip.packed, True
    return ip.packed, ip.ipv4_mapped is not None


class PyHostScheduler:
    """The Python twin of swp::Scheduler (csrc/swp_sched.cpp): the same event handlers, kept as an independent second
    implementation of the host layer — tests/test_sched_cpu.py runs both against the same scripted engine and compares
    every ABI call. Surface: create_node/update_node/delete_node/create_task/update_task/delete_task/tick."""

    SECOND = 1_000_000_000

    def __init__(self, engine=None, **engine_kw):
        self.e = engine or abi.Engine(**engine_kw)
        self.e.reset(0)
        self.now = 1_000_000_000_000
        self.nodes = {}            # id -> {"doc", "idx", "tasks": {task id: task doc}, "failures": {(svc, ver): [ts]}, "last_cleanup"}
        self.idx_to_id = {}
        self.services = {}         # id -> spec version index or None
        self.unassigned = {}       # insertion-ordered: id -> task doc
        self.pending_preassigned = {}
        self.last_decisions = {}   # task id -> (old task, preassigned?) of the last tick / process_preassigned (reject_decision)
        self.pushed_failures = {}  # (node index, service, spec version) -> count the engine holds
        self.irregular_generic = {}  # node id -> generic kinds its available list holds more than once, not all Named (_push_generic)
        self.last_error = ""
        self.preassigned = set()
        self.all_tasks = {}
        self._desc_cache = {}
        self._generic_touched = {}   # nodes whose available generic list _place changed during the current device call

    # ------------------------------------------------------------------------------ interning helpers
    def _folded(self, s):
        return self.e.intern(abi.SPACE_FOLDED, s or "")

    def _node_row(self, doc, idx, cpu, mem, total):
        flags = 0
        st = _get(doc, "Status", "State", default=0)
        st = {"UNKNOWN": 0, "DOWN": 1, "READY": 2, "DISCONNECTED": 3}.get(st, st)
        av = _get(doc, "Spec", "Availability", default=0)
        av = {"ACTIVE": 0, "PAUSE": 1, "DRAIN": 2}.get(av, av)
        if st == NODE_READY and av == AVAIL_ACTIVE:
            flags |= abi.NODE_READY
        role = doc.get("Role", 0)
        if role in (1, "MANAGER"):
            flags |= abi.NODE_MANAGER
        row = abi.NodeRow(node=idx, cpu=cpu, mem=mem, total=total)
        row.id_fold = self._folded(doc.get("ID", ""))
        labels = _get(doc, "Spec", "Annotations", "Labels")
        lab = []
        if labels is not None:
            flags |= abi.NODE_HAS_LABELS
            lab = [(self.e.intern(abi.SPACE_LABEL_KEY, k), self._folded(v), self.e.intern(abi.SPACE_RAW, v or "")) for k, v in labels.items()]
        elab, plugins = [], []
        desc = doc.get("Description")
        if desc is not None:
            flags |= abi.NODE_HAS_DESC
            row.hostname_fold = self._folded(desc.get("Hostname", ""))
            plat = desc.get("Platform")
            if plat is not None:
                flags |= abi.NODE_HAS_PLATFORM
                row.os = self.e.intern(abi.SPACE_OS, plat.get("OS", "") or "")
                row.arch = self.e.intern(abi.SPACE_ARCH, plat.get("Architecture", "") or "")
                row.os_fold = self._folded(plat.get("OS", ""))
                row.arch_fold = self._folded(plat.get("Architecture", ""))
            eng = desc.get("Engine")
            if eng is not None:
                flags |= abi.NODE_HAS_ENGINE
                el = eng.get("Labels")
                if el is not None:
                    flags |= abi.NODE_HAS_ELABELS
                    elab = [(self.e.intern(abi.SPACE_LABEL_KEY, k), self._folded(v), self.e.intern(abi.SPACE_RAW, v or "")) for k, v in el.items()]
                for p in eng.get("Plugins") or []:
                    typ, name = p.get("Type", ""), p.get("Name", "")
                    if typ == "Log":
                        flags |= abi.NODE_HAS_LOGPLUG
                    plugins.append(self.e.intern(abi.SPACE_PLUGIN, typ + "\0" + name))
                    if name.endswith(":latest"):   # filter.go:189-199: "name" also matches "name:latest"
                        plugins.append(self.e.intern(abi.SPACE_PLUGIN, typ + "\0" + name[:-7]))
        ip = _parse_ip(_get(doc, "Status", "Addr", default="") or "")
        if ip is not None:
            flags |= abi.NODE_IP_VALID | (abi.NODE_IP_V4 if ip[1] else 0)
            row.ip = (abi.C.c_uint8 * 16)(*ip[0])
        row.flags = flags
        row.version = _get(doc, "Meta", "Version", "Index", default=0)
        return row, lab, elab, plugins

    # ------------------------------------------------------------------------------ nodeSet mutators
    @staticmethod
    def _reservations(task):
        r = _get(task, "Spec", "Resources", "Reservations")
        if r is None:
            return 0, 0
        return int(r.get("NanoCPUs", 0) or 0), int(r.get("MemoryBytes", 0) or 0)

    def create_node(self, doc):
        """createOrUpdateNode, scheduler.go:368-396."""
        nid = doc["ID"]
        ent = self.nodes.get(nid)
        res = _get(doc, "Description", "Resources")
        cpu = mem = 0
        avail = []
        if res is not None:
            cpu, mem = int(res.get("NanoCPUs", 0) or 0), int(res.get("MemoryBytes", 0) or 0)
            avail, _ = gres.decode(res.get("Generic"))
            if ent is not None:   # :376-384: the reservations of the tasks already on the node, their generic resources taken out
                for t in ent["tasks"].values():
                    c, m = self._reservations(t)
                    cpu -= c
                    mem -= m
                    avail = gres.consume(avail, gres.decode(t.get("AssignedGenericResources"))[0])
        idx = self.e.intern(abi.SPACE_NODE_ID, nid)
        total = 0
        if ent is not None:
            cur = self.e.node_get(idx)
            total = cur.total if cur is not None else 0
        else:
            ent = {"doc": doc, "idx": idx, "tasks": {}, "failures": {}, "last_cleanup": self.now}
            self.nodes[nid] = ent
            self.idx_to_id[idx] = nid
        ent["doc"] = doc
        ent["generic"] = avail
        row, lab, elab, plugins = self._node_row(doc, idx, cpu, mem, total)
        self.e.node_upsert(row, lab, elab, plugins)
        self._push_generic(ent)

    def _push_generic(self, ent):
        """the node's available generic list as the engine sees it: one count per kind (swp_node_set_generic)"""
        items = []
        for kind, c in gres.counts(ent["generic"]).items():
            if c >= 1 << 31:
                raise Unsupported("a generic resource count of 2^31 or more stays on the Go path")
            items.append((self.e.intern(abi.SPACE_GENERIC_KIND, kind), c))
        self.e.node_set_generic(ent["idx"], items)
        irr = gres.irregular_kinds(ent["generic"])   # (a list one count per kind cannot stand for: tick() looks at this before it schedules)
        nid = ent["doc"].get("ID", "")
        if irr:
            self.irregular_generic[nid] = irr
        else:
            self.irregular_generic.pop(nid, None)

    def _refuse_irregular_generic(self, queue):
        """csrc/swp_sched.cpp refuseIrregularGeneric: a queued task reserves a kind some node lists more than once — the whole tick stays
        on the Go path."""
        if not self.irregular_generic:
            return
        for _, t in queue:
            r, _ = gres.decode(_get(t, "Spec", "Resources", "Reservations", "Generic"))
            for _, kind, _ in r:
                for nid in sorted(self.irregular_generic):
                    if kind in self.irregular_generic[nid]:
                        raise abi.Unsupported("node %s lists the generic kind '%s' more than once (its type changed under a running task): a tick with tasks that reserve it stays on the Go path" % (nid, kind))

    @staticmethod
    def _generic_reservations(t):
        """What swp_generic_set takes: Discrete entries, one per kind, values >= 1, at most 8 kinds; anything else stays on the Go path."""
        r, _ = gres.decode(_get(t, "Spec", "Resources", "Reservations", "Generic"))
        if len(r) > 8:
            raise Unsupported("more than 8 generic reservations in one task stay on the Go path")
        kinds = set()
        for named, kind, val in r:
            if named:
                raise Unsupported("a Named generic reservation stays on the Go path")
            if val < 1:
                raise Unsupported("a generic reservation below 1 stays on the Go path")
            if kind in kinds:
                raise Unsupported("a generic kind reserved twice stays on the Go path")
            kinds.add(kind)
        return r

    update_node = create_node

    def delete_node(self, nid):
        """nodeSet.remove, nodeset.go:46-48."""
        ent = self.nodes.pop(nid, None)
        self.irregular_generic.pop(nid, None)
        if ent is not None:
            self.e.node_remove(ent["idx"])
            # the engine hands the index to the next node that is new to it: nothing here may remember it as this node's
            if ent["idx"] < len(self.idx_to_id):
                self.idx_to_id[ent["idx"]] = ""
            for key in [k for k in self.pushed_failures if k[0] == ent["idx"]]:
                del self.pushed_failures[key]

    def node_info(self, nid):
        ent = self.nodes.get(nid)
        if ent is None:
            return None   # errNodeNotFound
        row = self.e.node_get(ent["idx"])
        by_service = {}
        for t in ent["tasks"].values():
            sid = t.get("ServiceID", "")
            c = self.e.node_get_svc_count(ent["idx"], self.e.intern(abi.SPACE_SERVICE, sid))
            if c:
                by_service[sid] = c
        return {"ID": nid, "ActiveTasksCount": row.total, "ActiveTasksCountByService": by_service,
                "AvailableResources": {"NanoCPUs": row.cpu, "MemoryBytes": row.mem, "Generic": gres.encode(ent["generic"])},
                "Tasks": sorted(ent["tasks"]),
                "RecentFailures": {"%s@%d" % (sid, ver): len(ts) for (sid, ver), ts in ent["failures"].items()}}

    # ------------------------------------------------------------------------------ NodeInfo.addTask/removeTask
    def _port_set(self, task):
        ports = [(int(p.get("Protocol", 0) if not isinstance(p.get("Protocol", 0), str) else {"TCP": 0, "UDP": 1, "SCTP": 2}[p["Protocol"]]),
                  int(p.get("PublishedPort", 0)))
                 for p in (_get(task, "Endpoint", "Ports") or [])
                 if p.get("PublishMode", 0) in (PUBLISH_HOST, "HOST") and int(p.get("PublishedPort", 0)) != 0]
        return self.e.port_set(ports) if ports else 0

    def _placement(self, ent, task, counted, with_resources=True):
        cpu, mem = self._reservations(task) if with_resources else (0, 0)
        p = np.zeros(1, dtype=abi.PLACEMENT_DTYPE)
        p["node"] = ent["idx"]
        p["service"] = self.e.intern(abi.SPACE_SERVICE, task.get("ServiceID", ""))
        p["cpu"], p["mem"] = cpu, mem
        p["port_set"] = self._port_set(task) if with_resources else 0
        p["counted"] = 1 if counted else 0
        return p

    def _add_task(self, ent, t):
        """nodeinfo.go:108-154; returns True when nodeInfo was modified."""
        old = ent["tasks"].get(t["ID"])
        ds = _state(t.get("DesiredState"))
        if old is not None:
            ods = _state(old.get("DesiredState"))
            if ds <= COMPLETE < ods:
                ent["tasks"][t["ID"]] = t
                self.e.commit(self._placement(ent, t, True, with_resources=False), add=True)
                return True
            if ods <= COMPLETE < ds:
                ent["tasks"][t["ID"]] = t
                self.e.commit(self._placement(ent, t, True, with_resources=False), add=False)
                return True
            return False
        # :128-137: a fresh AssignedGenericResources, then Claim against the node's available list
        stored = dict(t)
        ent["generic"], assigned = gres.claim(ent["generic"], gres.decode(_get(t, "Spec", "Resources", "Reservations", "Generic"))[0])
        stored["AssignedGenericResources"] = gres.encode(assigned)
        ent["tasks"][t["ID"]] = stored
        if t["ID"] in self.all_tasks:
            self.all_tasks[t["ID"]] = stored   # (the reference writes through the one *api.Task both maps point to)
        self.e.commit(self._placement(ent, t, ds <= COMPLETE), add=True)
        self._push_generic(ent)
        return True

    def _remove_task(self, ent, t):
        """nodeinfo.go:66-104."""
        old = ent["tasks"].pop(t["ID"], None)
        if old is None:
            return False
        self.e.commit(self._placement(ent, t, _state(old.get("DesiredState")) <= COMPLETE), add=False)
        # :95-104: the task's AssignedGenericResources go back — unless the node's description lists no generic resources at all
        node_res, desc_nil = gres.decode(_get(ent["doc"], "Description", "Resources", "Generic"))
        if not desc_nil:
            ent["generic"] = gres.reclaim(ent["generic"], gres.decode(t.get("AssignedGenericResources"))[0], node_res)
            self._push_generic(ent)
        return True

    def _task_failed(self, ent, t):
        """nodeinfo.go:177-202."""
        if self.now - ent["last_cleanup"] >= MONITOR_FAILURES:
            for k in [k for k, ts in ent["failures"].items() if not any(self.now - x < MONITOR_FAILURES for x in ts)]:
                del ent["failures"][k]
            ent["last_cleanup"] = self.now
        key = (t.get("ServiceID", ""), _get(t, "SpecVersion", "Index", default=0))
        lst = ent["failures"].get(key, [])
        expired = 0
        for ts in lst:
            if self.now - ts < MONITOR_FAILURES:
                break
            expired += 1
        ent["failures"][key] = lst[expired:] + [self.now]

    def _count_recent_failures(self, ent, key):
        """nodeinfo.go:206-221."""
        lst = ent["failures"].get(key, [])
        count = len(lst)
        for i in range(count - 1, -1, -1):
            if self.now - lst[i] > MONITOR_FAILURES:
                count -= i + 1
                break
        return count

    # ------------------------------------------------------------------------------ task event handlers
    def set_service(self, sid, spec_version=None):
        self.services[sid] = spec_version

    def delete_service(self, sid):
        self.services.pop(sid, None)

    def advance(self, seconds):
        self.now += int(seconds * self.SECOND)

    @staticmethod
    def _require_supported(t):
        """Tasks the engine cannot judge (generic resources, CSI cluster volumes) are refused at the event boundary —
        the shim leaves them to the Go scheduler's own path — so that a tick never meets one half-way through a batch."""
        PyHostScheduler._generic_reservations(t)   # refuses what swp_generic_set would refuse
        host_ports = sum(1 for p in (_get(t, "Endpoint", "Ports") or []) if p.get("PublishMode") in (1, "HOST") and p.get("PublishedPort"))
        if host_ports > 32:
            raise Unsupported("more than 32 host-mode ports in one task stay on the Go path")   # swp_port_set's limit
        for m in _get(t, "Spec", "Container", "Mounts") or []:
            if m.get("Type") in (MOUNT_CLUSTER, "CLUSTER"):
                raise Unsupported("CSI cluster volumes stay on the Go path")

    def create_task(self, t):
        """scheduler.go:254-283."""
        st = _state(_get(t, "Status", "State"))
        if st < PENDING or st > RUNNING:
            return False
        self._require_supported(t)
        self.all_tasks[t["ID"]] = t
        if not t.get("NodeID"):
            self.unassigned[t["ID"]] = t
            return True
        if st == PENDING:
            self.preassigned.add(t["ID"])
            self.pending_preassigned[t["ID"]] = t
            return False
        ent = self.nodes.get(t["NodeID"])
        if ent is not None:
            self._add_task(ent, t)
        return False

    def setup_task(self, t):
        """setupTasksList, scheduler.go:88-124: a task found in the store when the scheduler starts. Differs from the
        createTask event in one rule: a task still PENDING whose desired state is already past COMPLETED is ignored."""
        if _state(_get(t, "Status", "State")) == PENDING and _state(t.get("DesiredState")) > COMPLETE:
            return False
        return self.create_task(t)

    def update_task(self, t):
        """scheduler.go:285-349."""
        st = _state(_get(t, "Status", "State"))
        if st < PENDING:
            return False
        old = self.all_tasks.get(t["ID"])
        if st > RUNNING:
            if old is None:
                return False
            if st != _state(_get(old, "Status", "State")) and st in (FAILED, REJECTED):
                if t["ID"] not in self.preassigned:
                    ent = self.nodes.get(t.get("NodeID", ""))
                    if ent is not None:
                        self._task_failed(ent, t)
            self._delete_task(old)
            return True
        self._require_supported(t)
        if not t.get("NodeID"):
            if old is not None:
                self._delete_task(old)
            self.all_tasks[t["ID"]] = t
            self.unassigned[t["ID"]] = t
            return True
        if st == PENDING:
            if old is not None:
                self._delete_task(old)
            self.preassigned.add(t["ID"])
            self.all_tasks[t["ID"]] = t
            self.pending_preassigned[t["ID"]] = t
            return False
        self.all_tasks[t["ID"]] = t
        ent = self.nodes.get(t["NodeID"])
        if ent is not None:
            self._add_task(ent, t)
        return False

    def _delete_task(self, t):
        """scheduler.go:351-366."""
        self.all_tasks.pop(t["ID"], None)
        self.preassigned.discard(t["ID"])
        self.pending_preassigned.pop(t["ID"], None)
        ent = self.nodes.get(t.get("NodeID", ""))
        if ent is not None and self._remove_task(ent, t):
            return True
        return False

    def delete_task(self, t):
        return self._delete_task(t)

    # ------------------------------------------------------------------------------ Filter.SetTask → predicate sets
    def constraint_set(self, exprs):
        """ConstraintFilter.SetTask for a list of expressions: predicate-set id, 0 when empty / unparsable."""
        parsed = parse_constraints(list(exprs)) if exprs else None
        return self.e.constraint_set(self._constraint_structs(parsed)) if parsed else 0

    def _constraint_structs(self, parsed):
        out = []
        for key, op, exp in parsed:
            c = abi.Constraint(kind=abi.CK_INVALID, op=op)
            c.value = self._folded(exp)
            if _fold_eq(key, "node.id"):
                c.kind = abi.CK_NODE_ID
            elif _fold_eq(key, "node.hostname"):
                c.kind = abi.CK_HOSTNAME
            elif _fold_eq(key, "node.ip"):
                c.kind = abi.CK_IP
                ip = _parse_ip(exp)
                if ip is not None:
                    c.ip_kind, c.ip_is_v4 = abi.IP_SINGLE, 1 if ip[1] else 0
                    c.ip = (abi.C.c_uint8 * 16)(*ip[0])
                else:
                    net = None
                    if "/" in exp:
                        addr, _, plen = exp.partition("/")
                        a = _parse_ip(addr)
                        if a is not None and plen.isdigit() and len(plen) <= 3:
                            syntactic_v4 = ":" not in addr
                            bits = 32 if syntactic_v4 else 128
                            if int(plen) <= bits:
                                net = (a[0], syntactic_v4, int(plen) + (96 if syntactic_v4 else 0))
                    if net is None:
                        c.ip_kind = abi.IP_MALFORMED
                    else:
                        c.ip_kind, c.ip_is_v4, c.prefix_len = abi.IP_CIDR, 1 if net[1] else 0, net[2]
                        c.ip = (abi.C.c_uint8 * 16)(*net[0])
            elif _fold_eq(key, "node.role"):
                c.kind = abi.CK_ROLE
            elif _fold_eq(key, "node.platform.os"):
                c.kind = abi.CK_PLATFORM_OS
            elif _fold_eq(key, "node.platform.arch"):
                c.kind = abi.CK_PLATFORM_ARCH
            elif len(key) > len("node.labels.") and _fold_eq(key[:len("node.labels.")], "node.labels."):
                c.kind = abi.CK_NODE_LABEL
                c.key = self.e.intern(abi.SPACE_LABEL_KEY, key[len("node.labels."):])
            elif len(key) > len("engine.labels.") and _fold_eq(key[:len("engine.labels.")], "engine.labels."):
                c.kind = abi.CK_ENGINE_LABEL
                c.key = self.e.intern(abi.SPACE_LABEL_KEY, key[len("engine.labels."):])
            out.append(c)
        return out

    def task_desc(self, t):
        """One swp_task_desc (numpy record) from an api.Task doc == Pipeline.SetTask (pipeline.go:76-81)."""
        d = np.zeros(1, dtype=abi.TASK_DTYPE)
        d["service"] = self.e.intern(abi.SPACE_SERVICE, t.get("ServiceID", ""))
        res = _get(t, "Spec", "Resources", "Reservations")
        if res is not None:
            cpu, mem = int(res.get("NanoCPUs", 0) or 0), int(res.get("MemoryBytes", 0) or 0)
            d["cpu"], d["mem"] = cpu, mem
            any_generic = isinstance(res.get("Generic"), list) and len(res["Generic"]) > 0
            if cpu != 0 or mem != 0 or any_generic:   # ResourceFilter.SetTask, filter.go:61-74
                d["flags"] |= abi.TASK_RES_ENABLED
            items = [(self.e.intern(abi.SPACE_GENERIC_KIND, kind), val) for _, kind, val in self._generic_reservations(t)]
            if items:
                d["generic_set"] = self.e.generic_set(items)
        if _state(t.get("DesiredState")) > COMPLETE:
            d["flags"] |= abi.TASK_UNCOUNTED
        pl = _get(t, "Spec", "Placement")
        if pl is not None:
            cons = pl.get("Constraints") or []
            if cons:
                parsed = parse_constraints(cons)
                if parsed is not None:   # parse error ⇒ filter disabled (filter.go:223-229)
                    d["constraint_set"] = self.e.constraint_set(self._constraint_structs(parsed))
            plats = pl.get("Platforms") or []
            if plats:
                d["platform_set"] = self.e.platform_set(
                    [(self.e.intern(abi.SPACE_OS, p.get("OS", "") or ""), self.e.intern(abi.SPACE_ARCH, p.get("Architecture", "") or "")) for p in plats])
            d["max_replicas"] = int(pl.get("MaxReplicas", 0) or 0)
            levels = []
            for pref in pl.get("Preferences") or []:   # nodeset.go:59-82: only label spreads create a tree level
                sp = pref.get("Spread")
                if sp is None:
                    continue
                sd = sp.get("SpreadDescriptor", "") or ""
                if len(sd) > len("node.labels.") and _fold_eq(sd[:len("node.labels.")], "node.labels."):
                    levels.append((abi.CK_NODE_LABEL, self.e.intern(abi.SPACE_LABEL_KEY, sd[len("node.labels."):])))
                elif len(sd) > len("engine.labels.") and _fold_eq(sd[:len("engine.labels.")], "engine.labels."):
                    levels.append((abi.CK_ENGINE_LABEL, self.e.intern(abi.SPACE_LABEL_KEY, sd[len("engine.labels."):])))
            if levels:
                d["spread_set"] = self.e.spread_set(levels)
        # PluginFilter.SetTask, filter.go:119-131
        mounts = _get(t, "Spec", "Container", "Mounts") or []
        for m in mounts:
            if m.get("Type") in (MOUNT_CLUSTER, "CLUSTER"):
                raise Unsupported("CSI cluster volumes stay on the Go path")
        vol = [_get(m, "VolumeOptions", "DriverConfig", "Name") for m in mounts
               if m.get("Type") in (MOUNT_VOLUME, "VOLUME") and _get(m, "VolumeOptions", "DriverConfig") is not None
               and _get(m, "VolumeOptions", "DriverConfig", "Name") not in (None, "", "local")]
        nets = t.get("Networks") or []
        logd = _get(t, "Spec", "LogDriver")
        if nets or logd is not None or vol:
            req = [self.e.intern(abi.SPACE_PLUGIN, "Volume\0" + v) for v in vol]
            for na in nets:
                name = _get(na, "Network", "DriverState", "Name")
                if name:
                    req.append(self.e.intern(abi.SPACE_PLUGIN, "Network\0" + name))
            log = 0
            if logd is not None and logd.get("Name") not in (None, "", "none"):
                log = self.e.intern(abi.SPACE_PLUGIN, "Log\0" + logd["Name"])
            if req or log:
                d["plugin_set"] = self.e.plugin_set(req, log)
        d["port_set"] = self._port_set(t)
        d["spec_version"] = _get(t, "SpecVersion", "Index", default=0)
        return d

    # ------------------------------------------------------------------------------ Explain
    _EXPLAIN = [("1 node not available for new tasks", "%d nodes not available for new tasks"),
                ("insufficient resources on 1 node", "insufficient resources on %d nodes"),
                ("missing plugin on 1 node", "missing plugin on %d nodes"),
                ("scheduling constraints not satisfied on 1 node", "scheduling constraints not satisfied on %d nodes"),
                ("unsupported platform on 1 node", "unsupported platform on %d nodes"),
                ("host-mode port already in use on 1 node", "host-mode port already in use on %d nodes"),
                ("max replicas per node limit exceed", "max replicas per node limit exceed"),
                ("cannot fulfill requested CSI volume mounts on 1 node", "cannot fulfill requested CSI volume mounts on %d nodes")]

    @classmethod
    def explain(cls, hist):
        """Pipeline.Explain, pipeline.go:84-103: stable sort by failure count, descending."""
        order = sorted(range(len(hist)), key=lambda i: -int(hist[i]))
        parts = []
        for i in order:
            n = int(hist[i])
            if n > 0:
                one, many = cls._EXPLAIN[i]
                parts.append(one if n == 1 else (many % n if "%d" in many else many))
        return "; ".join(parts)

    # ------------------------------------------------------------------------------ tick
    def _push_failures(self, service_ids):
        """The failure counts nodeLess reads (scheduler.go:706-735) for the services of the coming batch, at its `now`. A bucket
        the engine still holds a count for but the node no longer has (erased by cleanupFailures, nodeinfo.go:163-183, or the
        node left) is reset to 0: that is what countRecentFailures would say."""
        now = {}
        for _nid, ent in self.nodes.items():
            for (sid, ver) in ent["failures"]:
                if sid in service_ids:
                    now[(ent["idx"], sid, ver)] = self._count_recent_failures(ent, (sid, ver))
        for key in sorted(self.pushed_failures):
            idx, sid, ver = key
            if sid in service_ids and key not in now:
                if self.pushed_failures[key] != 0 and idx < len(self.idx_to_id) and self.idx_to_id[idx] in self.nodes:
                    self.e.node_set_failures(idx, self.e.intern(abi.SPACE_SERVICE, sid), ver, 0)
                del self.pushed_failures[key]
        for key in sorted(now):
            idx, sid, ver = key
            self.e.node_set_failures(idx, self.e.intern(abi.SPACE_SERVICE, sid), ver, now[key])
            self.pushed_failures[key] = now[key]

    def process_preassigned(self):
        """processPreassignedTasks + taskFitNode, scheduler.go:398-426, 646-690."""
        decisions = []
        self.last_decisions = {k: v for k, v in self.last_decisions.items() if not v[1]}
        for tid, t in list(self.pending_preassigned.items()):
            ent = self.nodes.get(t.get("NodeID", ""))
            if ent is None:
                continue
            new_t = dict(t)
            try:
                ff = self.e.check_node(self.task_desc(t), ent["idx"])
            except (abi.SwpError, abi.Unsupported) as err:   # the engine cannot judge this task: it stays pending, the loop carries on
                self.last_error = str(err)
                d = self._decision(t, t)
                d["Err"] = "swp: deferred to the host scheduler: " + self._err_text(err)
                d["Deferred"] = True
                decisions.append(d)
                continue
            if ff >= 0:
                hist = [0] * abi.NFILTERS
                hist[ff] = 1
                new_t["Status"] = dict(t.get("Status", {}), Err=self.explain(hist))
                self.all_tasks[tid] = new_t
            else:
                new_t["Status"] = {"State": ASSIGNED, "Message": "scheduler confirmed task can run on preassigned node"}
                self.all_tasks[tid] = new_t
                self._add_task(ent, new_t)
                del self.pending_preassigned[tid]
                # taskFitNode hands addTask the task the decision carries (scheduler.go:676-688): what Claim assigned is part of decision.new
                stored = self.all_tasks.get(tid, {})
                if stored.get("AssignedGenericResources"):
                    new_t["AssignedGenericResources"] = stored["AssignedGenericResources"]
            self.last_decisions[tid] = (t, True)
            d = self._decision(t, new_t)
            if new_t.get("AssignedGenericResources"):
                d["AssignedGenericResources"] = new_t["AssignedGenericResources"]
            decisions.append(d)
        return decisions

    @staticmethod
    def _decision(old, new):
        return {"ID": new["ID"], "ServiceID": new.get("ServiceID", ""), "NodeID": new.get("NodeID", ""),
                "State": _state(_get(new, "Status", "State")), "Message": _get(new, "Status", "Message", default=""),
                "Err": _get(new, "Status", "Err", default=""), "OldState": _state(_get(old, "Status", "State"))}

    def _place(self, tid, t, n, decisions):
        nid = self.idx_to_id[int(n)]
        new_t = dict(t)
        new_t["NodeID"] = nid
        new_t["Status"] = {"State": ASSIGNED, "Message": "scheduler assigned task to node"}
        # nodeInfo.addTask(&newT) (:886-888): the counts moved on the device already; WHICH resources the task holds is decided here
        want, _ = gres.decode(_get(t, "Spec", "Resources", "Reservations", "Generic"))
        ent = self.nodes[nid]
        if want:
            ent["generic"], assigned = gres.claim(ent["generic"], want)
            new_t["AssignedGenericResources"] = gres.encode(assigned)
            self._generic_touched[nid] = True   # pushed once the whole call's placements are booked (_push_touched)
        self.all_tasks[tid] = new_t
        ent["tasks"][tid] = new_t   # numeric addTask already happened on the device
        self.last_decisions[tid] = (t, False)
        d = self._decision(t, new_t)
        if want:
            d["AssignedGenericResources"] = new_t["AssignedGenericResources"]
        decisions.append(d)

    def _push_touched(self):
        for nid in sorted(self._generic_touched):
            if nid in self.nodes:
                self._push_generic(self.nodes[nid])
        self._generic_touched = {}

    def _no_suitable_node(self, tid, t, hist, decisions):
        """noSuitableNode, scheduler.go:928-971."""
        sid = t.get("ServiceID", "")
        if sid not in self.services:
            return
        new_t = dict(t)
        sv, tv = self.services[sid], _get(t, "SpecVersion", "Index")
        if sv is not None and tv is not None and sv > tv:
            if _state(_get(t, "Status", "State")) == PENDING and _state(t.get("DesiredState")) >= SHUTDOWN:
                new_t["Status"] = dict(t.get("Status", {}), State=SHUTDOWN, Err="")
        else:
            ex = self.explain(hist)
            new_t["Status"] = dict(t.get("Status", {}), Err="no suitable node (" + ex + ")" if ex else "no suitable node")
            self.unassigned[tid] = new_t
        self.all_tasks[tid] = new_t
        self.last_decisions[tid] = (t, False)
        decisions.append(self._decision(t, new_t))

    def _defer(self, tid, t, err, decisions):
        """A device call failed for this task (a group beyond the engine's capacity, ...): nothing of the call was applied, the
        task goes back on the queue and the tick carries on; the Go shim routes a deferred task to the reference's own path."""
        self.unassigned[tid] = t
        self.last_error = str(err)
        d = self._decision(t, t)
        d["Err"] = "swp: deferred to the host scheduler: " + self._err_text(err)
        d["Deferred"] = True
        decisions.append(d)

    @staticmethod
    def _err_text(err):
        return getattr(err, "msg", None) or str(err)

    def reject_decision(self, tid):
        """The failed half of applySchedulingDecisions (scheduler.go:472-487 after tick, :416-425 for preassigned tasks)."""
        if tid not in self.last_decisions:
            return False
        old, preassigned = self.last_decisions.pop(tid)
        new_t = self.all_tasks.get(tid)
        if new_t is not None:
            ent = self.nodes.get(new_t.get("NodeID", ""))
            if ent is not None and not old.get("NodeID"):
                self._remove_task(ent, new_t)
            elif ent is not None and preassigned and _state(_get(new_t, "Status", "State")) == ASSIGNED:
                self._remove_task(ent, new_t)
        self.all_tasks[tid] = old
        if preassigned:
            self.pending_preassigned[tid] = old
        else:
            self.unassigned[tid] = old
        return True

    def commit_plan(self, max_changes=0):
        """swp_sched_commit_plan: the last call's decisions grouped by node (node index order) with the node's Meta.Version, the
        decisions that name no node, and transactions of at most max_changes (200) updates."""
        max_changes = max_changes or 200
        by_node, unassigned = {}, []
        for tid in sorted(self.last_decisions):
            nid = (self.all_tasks.get(tid) or {}).get("NodeID", "")
            ent = self.nodes.get(nid) if nid else None
            if ent is None:
                unassigned.append(tid)
            else:
                by_node.setdefault(ent["idx"], []).append(tid)
        nodes, order = [], []
        for idx in sorted(by_node):
            row = self.e.node_get(idx)
            nodes.append({"NodeID": self.idx_to_id[idx], "Version": int(row.version), "Tasks": by_node[idx]})
            order.extend(by_node[idx])
        order.extend(unassigned)
        txs = [order[i:i + max_changes] for i in range(0, len(order), max_changes)]
        return {"Nodes": nodes, "Unassigned": unassigned, "Transactions": txs, "VolumeFailed": [], "Publish": []}   # (this twin takes no task with cluster mounts)

    def reject_decisions(self, tids):
        return sum(1 for t in tids if self.reject_decision(t))

    def reject_node(self, nid):
        ids = [tid for tid in sorted(self.last_decisions) if (self.all_tasks.get(tid) or {}).get("NodeID", "") == nid]
        return sum(1 for t in ids if self.reject_decision(t))

    def _run_groups(self, groups, decisions):
        """groups: list of [(tid, task)...] sharing a spec; one swp_schedule_groups call, groups in order."""
        if not groups:
            return
        # one device call must not mix spec versions of one service (the failure buckets are per (service, version)):
        # split the ordered group list into consecutive runs that respect this, keeping the order
        seen, cut = {}, None
        for i, g in enumerate(groups):
            sid, ver = g[0][1].get("ServiceID", ""), _get(g[0][1], "SpecVersion", "Index", default=0)
            if seen.setdefault(sid, ver) != ver:
                cut = i
                break
        if cut is not None:
            self._run_groups(groups[:cut], decisions)
            try:
                self._push_failures({g[0][1].get("ServiceID", "") for g in groups[cut:]})
            except (abi.SwpError, abi.Unsupported) as err:   # the other spec version's counts did not reach the engine: those groups wait
                for g in groups[cut:]:
                    for tid, t in g:
                        self._defer(tid, t, err, decisions)
                return
            self._run_groups(groups[cut:], decisions)
            return
        descs = []
        for i, g in enumerate(groups):
            try:
                descs.append(self.task_desc(g[0][1]))
            except (abi.SwpError, abi.Unsupported) as err:   # a predicate set the engine refuses: this group is deferred, the others run
                self._run_groups(groups[:i], decisions)
                for tid, t in g:
                    self._defer(tid, t, err, decisions)
                self._run_groups(groups[i + 1:], decisions)
                return
        descs = np.concatenate(descs)
        sizes = np.array([len(g) for g in groups], dtype=np.uint32)
        try:
            out, hist = self.e.schedule_groups(descs, sizes)
        except (abi.SwpError, abi.Unsupported) as err:
            if len(groups) > 1:   # find the group(s) the engine cannot take: run them one by one
                for g in groups:
                    self._run_groups([g], decisions)
                return
            for tid, t in groups[0]:
                self._defer(tid, t, err, decisions)
            return
        off = 0
        for gi, g in enumerate(groups):
            for i, (tid, t) in enumerate(g):
                n = out[off + i]
                if n >= 0:
                    self._place(tid, t, n, decisions)
                else:
                    self._no_suitable_node(tid, t, hist[gi], decisions)
            off += len(g)
        self._push_touched()   # groups with generic reservations: the nodes' available lists after _place's Claim

    def _run_one_offs(self, run, decisions):
        if not run:
            return
        descs = np.concatenate([d for _, _, d in run])
        try:
            out, hist = self.e.schedule_batch(descs)
        except (abi.SwpError, abi.Unsupported) as err:
            for tid, t, _ in run:
                self._defer(tid, t, err, decisions)
            return
        for (tid, t, _), n, h in zip(run, out, hist):
            if n >= 0:
                self._place(tid, t, n, decisions)
            else:
                self._no_suitable_node(tid, t, h, decisions)
        self._push_touched()

    def tick(self):
        """scheduler.go:429-488: groups (ServiceID, SpecVersion) in first-seen order, then the one-off tasks in
        queue order; every scheduling step is a device call (swp_schedule_groups / swp_schedule_batch)."""
        queue = [(tid, t) for tid, t in self.unassigned.items() if t is not None and not t.get("NodeID")]
        self.unassigned.clear()
        decisions = []
        self.last_decisions = {k: v for k, v in self.last_decisions.items() if v[1]}   # the previous tick's decisions are final now
        if not queue:
            return decisions
        try:
            self._refuse_irregular_generic(queue)
            self._push_failures({t.get("ServiceID", "") for _, t in queue})
        except (abi.SwpError, abi.Unsupported) as err:   # nothing was scheduled: the whole queue stays queued
            for tid, t in queue:
                self._defer(tid, t, err, decisions)
            return decisions
        # whatever else fails below (every device call has its own handler): the tasks without a decision line go back on the queue and
        # the decisions made so far are still returned — they are applied to all_tasks and the node rows already
        try:
            self._schedule_queue(queue, decisions)
        except (abi.SwpError, abi.Unsupported) as err:
            have = {d["ID"] for d in decisions}
            for tid, t in queue:   # (a task deleted while it was queued is not in all_tasks: its queued document goes back, as the normal path would have scheduled it)
                if tid not in have:
                    self._defer(tid, self.all_tasks.get(tid, t), err, decisions)
        return decisions

    def _schedule_queue(self, queue, decisions):
        grouped, one_off = {}, []
        for tid, t in queue:
            if t.get("SpecVersion") is not None:
                grouped.setdefault((t.get("ServiceID", ""), _get(t, "SpecVersion", "Index", default=0)), []).append((tid, t))
            else:
                one_off.append((tid, t))
        self._run_groups(list(grouped.values()), decisions)
        # one-off tasks: a task with spread preferences is a group of one and must keep its place in the order
        run = []
        for tid, t in one_off:
            try:
                d = self.task_desc(t)   # Pipeline.SetTask once per one-off task
            except (abi.SwpError, abi.Unsupported) as err:
                self._defer(tid, t, err, decisions)
                continue
            if int(d["spread_set"][0]):
                self._run_one_offs(run, decisions)
                run = []
                self._run_groups([[(tid, t)]], decisions)
            else:
                run.append((tid, t, d))
        self._run_one_offs(run, decisions)


def _py_enforce(sched, node_docs, tasks_by_node, services=None):
    """constraintenforcer.rejectNoncompliantTasks (constraint_enforcer.go:65-196) for many nodes through swp_enforce.
    node_docs: api.Node docs already known to `sched` (create_node); tasks_by_node: {node id: [api.Task docs]} (sorted here
    by task ID = the canonical store order); services: {ServiceID: api.Service doc} — the CURRENT specs.
    Returns {node id: [rejected task ids]} for the ACTIVE nodes (others are skipped, :70-72)."""
    services = services or {}
    nrec, trec, owners, walks = [], [], [], []
    for nd in node_docs:
        avail = _get(nd, "Spec", "Availability")
        if avail not in (None, 0, "ACTIVE"):
            continue
        nid = nd["ID"]
        tasks = sorted(tasks_by_node.get(nid, ()), key=lambda t: t["ID"])
        res = _get(nd, "Description", "Resources") or {}
        first = len(trec)
        if any(t.get("AssignedGenericResources") is not None for t in tasks):   # the generic half is walked after the device call
            walks.append((nid, first, gres.decode(res.get("Generic"))[0], [gres.decode(t.get("AssignedGenericResources")) for t in tasks]))
        for t in tasks:
            svc = services.get(t.get("ServiceID", ""))
            pl = _get(svc, "Spec", "Task", "Placement") if svc is not None else _get(t, "Spec", "Placement")
            cset = 0
            cons = (pl or {}).get("Constraints") or []
            if cons:
                parsed = parse_constraints(cons)
                if parsed is not None:   # `constraints, _ := constraint.Parse(...)`: an error leaves no constraints (:163)
                    cset = sched.e.constraint_set(sched._constraint_structs(parsed))
            r = _get(t, "Spec", "Resources", "Reservations")
            trec.append((int((r or {}).get("NanoCPUs", 0) or 0), int((r or {}).get("MemoryBytes", 0) or 0), cset,
                         abi.ENF_RESERVATIONS if r is not None else 0, _state(t.get("DesiredState")), _state(_get(t, "Status", "State"))))
            owners.append((nid, t["ID"]))
        nrec.append((sched.nodes[nid]["idx"], first, len(trec) - first, 0, int(res.get("NanoCPUs", 0) or 0), int(res.get("MemoryBytes", 0) or 0)))
    out = {nd["ID"]: [] for nd in node_docs if _get(nd, "Spec", "Availability") in (None, 0, "ACTIVE")}
    if not trec:
        return out
    rej = np.array(sched.e.enforce(np.array(nrec, dtype=abi.ENF_NODE_DTYPE), np.array(trec, dtype=abi.ENF_TASK_DTYPE)), dtype=np.uint8)
    # constraint_enforcer.go:186-200 for the nodes whose tasks hold generic resources: a task the device kept claims what it was assigned
    # from the node's list; the first task whose assignment is no longer there is rejected and ends the node's loop (`break loop`)
    for nid, first, avail, assigned in walks:
        broke = False
        for k, (lst, is_nil) in enumerate(assigned):
            i = first + k
            if broke:
                rej[i] = 0
                continue
            ds, st = trec[i][4], trec[i][5]
            if ds < ASSIGNED or ds > COMPLETE or st >= COMPLETE or rej[i] or is_nil:
                continue
            if any(not gres.has_resource(ta, avail) for ta in lst):
                rej[i] = 1
                broke = True
                continue
            avail = gres.consume(avail, lst)
    for (nid, tid), r in zip(owners, rej):
        if r:
            out[nid].append(tid)
    return out


PyHostScheduler.enforce = lambda self, node_docs, tasks_by_node, services=None: _py_enforce(self, node_docs, tasks_by_node, services)
