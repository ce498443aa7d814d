"""CPU: the C++ host layer (csrc/swp_sched.cpp, include/swp_sched.h).

1. Its pure string helpers (constraint.Parse, key folding, net.ParseIP, Pipeline.Explain) against the oracle's
   restatement of the Go behaviour — called in the PRODUCT library (libswp.so loads without a GPU; only swp_create needs one).
2. The scheduler logic itself, event by event, against its independent Python twin (swarmkit_amd/host.py, the
   implementation the GPU parity suite has validated against the oracle): both drive the same scripted test double of
   the engine (tests/fake_swp.cpp — no placement logic, pseudo-random answers) and must issue the same ABI calls with
   the same arguments and report the same decisions."""
import random

import pytest

import pyhost

import fakelib
import orc
import test_engine_fuzz as fz
from swarmkit_amd import abi, host as swhost, sched as swsched


# ------------------------------------------------------------------------------------------------ helpers vs the oracle
@pytest.mark.parametrize("seed", range(8))
def test_cxx_parse_constraints_matches_the_oracle(seed):
    import test_host_fuzz_cpu as hf
    rng = random.Random(0xFADE + seed)
    for _ in range(400):
        exprs = [hf.rand_expr(rng) for _ in range(rng.randrange(1, 4))]
        want, err = orc.constraint_parse(exprs)
        got = swsched.parse_constraints(exprs)
        if want is None:
            assert got is None, (exprs, err, got)
        else:
            assert got is not None, (exprs, want)
            assert [(k, o, v) for k, o, v in got] == [(k, o, v) for k, o, v in want], exprs


@pytest.mark.parametrize("expr,ok,key,exp", __import__("kat_tables").PARSE_CASES)
def test_cxx_constraint_parse_kat(expr, ok, key, exp):
    p = swsched.parse_constraints([expr])
    assert (p is not None) == ok
    if ok:
        assert p[0][0] == key and p[0][2] == exp


def test_cxx_key_fold_matches_equal_fold():
    rng = random.Random(5)
    chars = list("abkKsS.-_09") + ["K", "ſ"]
    for _ in range(3000):
        a = "".join(rng.choice(chars) for _ in range(rng.randrange(0, 6)))
        b = "".join(rng.choice(chars) for _ in range(rng.randrange(0, 6)))
        if rng.random() < 0.5:
            b = "".join(rng.choice([c, c.upper(), c.lower()]) for c in a)
        assert swsched.key_equal_fold(a, b) == orc.equal_fold(a, b), (a, b)


def test_cxx_parse_ip_matches_the_python_twin():
    rng = random.Random(11)
    fixed = ["10.0.0.1", "255.255.255.255", "256.1.1.1", "1.2.3", "1.2.3.4.5", "01.2.3.4", "1.2.3.04", " 1.2.3.4", "1.2.3.4 ", "", "::", "::1", "fe80::1%eth0",
             "::ffff:1.2.3.4", "::1.2.3.4", "2001:db8::8a2e:370:7334", "2001:db8:0:0:0:0:2:1", "1:2:3:4:5:6:7:8", "1:2:3:4:5:6:7", "1::2::3", "12345::1", "g::1",
             "1:2:3:4:5:6:1.2.3.4", "::ffff:01.2.3.4", "1.2.3.4/24", "a.b.c.d", "0.0.0.0", "::ffff:0:0", "FE80::ABCD"]
    corpus = list(fixed)
    for _ in range(2000):
        if rng.random() < 0.5:
            corpus.append(".".join(str(rng.choice([0, 1, 9, 10, 99, 100, 255, 256, 300])) for _ in range(rng.choice([3, 4, 4, 4, 5]))))
        else:
            groups = ["%x" % rng.randrange(0, 1 << rng.choice([4, 8, 16])) for _ in range(rng.choice([2, 4, 7, 8, 8, 9]))]
            s = ":".join(groups)
            if rng.random() < 0.4:
                cut = rng.randrange(0, len(groups))
                s = ":".join(groups[:cut]) + "::" + ":".join(groups[cut + 1:])
            corpus.append(s)
    for s in corpus:
        assert swsched.parse_ip(s) == pyhost._parse_ip(s), s


# ------------------------------------------------------------------------------------------------ twin test
class Pair:
    """The two host layers over two instances of the scripted engine."""

    def __init__(self):
        lib = fakelib.build()
        self.py = pyhost.PyHostScheduler(engine=abi.Engine(lib_path=lib))
        self.cx = swsched.Scheduler(engine=abi.Engine(lib_path=lib))
        self.steps = 0

    def both(self, name, *args):
        self.steps += 1
        res = []
        for s in (self.py, self.cx):
            try:
                res.append(("ok", getattr(s, name)(*args)))
            except abi.Unsupported:
                res.append(("unsupported", None))
        assert res[0] == res[1], (self.steps, name, args[:1], res)
        lp, lc = fakelib.take_log(self.py.e), fakelib.take_log(self.cx.e)
        assert lp == lc, (self.steps, name, [(a, b) for a, b in zip(lp, lc) if a != b][:3], len(lp), len(lc))
        return res[0][1]

    def enforce(self, node_docs, tbn, services):
        self.steps += 1
        res = []
        for s in (self.py, self.cx):
            try:
                res.append(swhost.enforce(s, node_docs, tbn, services))
            except abi.Unsupported:
                res.append("unsupported")
        a, b = res
        assert a == b, (self.steps, a, b)
        lp, lc = fakelib.take_log(self.py.e), fakelib.take_log(self.cx.e)
        assert lp == lc, (self.steps, [(x, y) for x, y in zip(lp, lc) if x != y][:3])
        return a


def odd_task_bits(rng, t):
    """Fields the GPU fuzz does not draw: string enums, mounts, log drivers, IP constraints, odd preferences."""
    r = rng.random()
    spec = dict(t.get("Spec") or {})
    if r < 0.10:
        spec["LogDriver"] = {"Name": rng.choice(["syslog", "none", "", "json-file"])}
    elif r < 0.18:
        spec["Container"] = {"Mounts": [{"Type": rng.choice([1, "VOLUME", 0, "BIND"]), "VolumeOptions": {"DriverConfig": {"Name": rng.choice(["nfs", "local", "", "ceph"])}}},
                                        {"Type": 1, "VolumeOptions": {}}]}
    elif r < 0.22:
        spec["Container"] = {"Mounts": [{"Type": rng.choice([4, "CLUSTER"]), "Source": "v%d" % q, "Target": "/m%d" % q} for q in range(9)]}   # more cluster mounts than swp_mount_set takes: unsupported (the Python twin knows no volumes at all)
    elif r < 0.25:
        spec["Resources"] = {"Reservations": {"NanoCPUs": 10**9, "Generic": [{"DiscreteResourceSpec": {"Kind": "gpu", "Value": 1}}]}}   # unsupported
    pl = dict(spec.get("Placement") or {})
    r = rng.random()
    if r < 0.12:
        pl["Constraints"] = [rng.choice(["node.ip==10.0.0.0/24", "node.ip!=10.0.0.7", "node.ip==fe80::/10", "node.ip==10.0.0.0/33", "NODE.ID==n00001", "node.hostname!=H3",
                                         "node.role==manager", "node.platform.arch==AMD64", "Node.Labels.Zone == a", "bogus", "node.labels.zone=a", "engine.labels.tier!=gold ",
                                         "node.labelsſ==x", "node.ip==::ffff:10.0.0.1", "node.labels.==x"])]
        if rng.random() < 0.3:
            pl["Constraints"].append("node.labels.disk != hdd")
    if r > 0.9:
        pl["Preferences"] = [{"Spread": {"SpreadDescriptor": rng.choice(["NODE.LABELS.zone", "engine.labels.tier", "node.labels.", "node.id", "", "Node.Labelſ.rack"])}}, {"Other": 1}]
    if rng.random() < 0.05:
        pl["MaxReplicas"] = rng.choice([0, 1, 2**63, 2**64 - 1])
    if pl:
        spec["Placement"] = pl
    if spec:
        t["Spec"] = spec
    if rng.random() < 0.08:
        t["Endpoint"] = {"Ports": [{"Protocol": rng.choice([0, 1, "TCP", "UDP", "SCTP"]), "PublishedPort": rng.choice([0, 80, 8080]), "PublishMode": rng.choice([0, 1, "HOST", "INGRESS"])}
                                   for _ in range(rng.randrange(1, 4))]}
    if rng.random() < 0.08:
        t["Networks"] = [{"Network": {"DriverState": {"Name": rng.choice(["overlay", "", "weave"])}}}, {"Network": {}}]
    return t


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SWP_TWIN_SEEDS", "24"))))
def test_cxx_host_is_the_twin_of_the_python_host(seed):
    rng = random.Random(0xBEE5 + seed)
    p = Pair()
    n_nodes = rng.choice([1, 3, 17, 40])
    nodes = {}
    for i in range(n_nodes):
        d = fz.node_doc(rng, i)
        if rng.random() < 0.3:   # string enums, roles, versions, IPv6 / broken addresses
            d["Status"] = {"State": rng.choice(["READY", "DOWN", "UNKNOWN", 2]), "Addr": rng.choice(["fe80::%d" % i, "10.1.2.%d" % i, "not-an-ip", "::ffff:10.0.0.%d" % i, ""])}
            d["Spec"] = dict(d["Spec"], Availability=rng.choice(["ACTIVE", "DRAIN", "PAUSE", 0]))
            d["Role"] = rng.choice(["MANAGER", "WORKER", 1, 0])
            d["Meta"] = {"Version": {"Index": rng.randrange(1, 1000)}}
        if rng.random() < 0.1:
            d["Description"]["Engine"]["Plugins"] = [{"Type": "Log", "Name": "syslog"}, {"Type": "Volume", "Name": "ceph:latest"}]
        if rng.random() < 0.05:
            del d["Description"]
        if rng.random() < 0.05:
            d["Description"] = dict(d.get("Description") or {}, Resources={"NanoCPUs": 4 * 10**9, "Generic": [{"DiscreteResourceSpec": {"Kind": "gpu", "Value": 2}}]})   # unsupported
        nodes[i] = d
        p.both("create_node", d)
    n_svc = rng.randrange(1, 8)
    specs = [fz.service_spec(rng) for _ in range(n_svc)]
    grouped = [rng.random() < 0.5 for _ in range(n_svc)]
    version = [1] * n_svc
    for k in range(n_svc):
        p.both("set_service", "svc%02d" % k, version[k] if grouped[k] and rng.random() < 0.7 else None)
    docs, placed, tid = {}, {}, 0

    def note(decisions):
        for d in decisions:
            if d["NodeID"] and d["State"] >= orc.ASSIGNED:
                placed[d["ID"]] = d["NodeID"]

    for rnd in range(rng.randrange(2, 6)):
        for _ in range(rng.randrange(1, 5)):
            k = rng.randrange(n_svc)
            if grouped[k] and rng.random() < 0.2:   # service update: a newer spec version (old tasks of the service may still be queued)
                version[k] += 1
                p.both("set_service", "svc%02d" % k, version[k])
            for _ in range(rng.choice([1, 2, 5, 20])):
                t = {"ID": "t%06d" % tid, "ServiceID": "svc%02d" % k, "DesiredState": rng.choice([orc.RUNNING, orc.RUNNING, "RUNNING", orc.SHUTDOWN]),
                     "Status": {"State": rng.choice([orc.PENDING, orc.PENDING, "PENDING", orc.NEW, orc.RUNNING])}}
                if grouped[k]:
                    t["SpecVersion"] = {"Index": rng.choice([version[k], version[k], max(1, version[k] - 1)])}
                t.update(specs[k])
                t = odd_task_bits(rng, t)
                r = rng.random()
                if r < 0.15 and nodes:   # preassigned (global-service style) task
                    t["NodeID"] = rng.choice(sorted(d["ID"] for d in nodes.values()))
                elif r < 0.18:
                    t["NodeID"] = "n-unknown"
                docs[t["ID"]] = t
                p.both(rng.choice(["create_task", "create_task", "create_task", "setup_task", "update_task"]), t)
                tid += 1
        if rng.random() < 0.6:
            note(p.both("process_preassigned"))
        note(p.both("tick"))
        for _ in range(rng.randrange(0, 5)):   # churn between ticks
            act = rng.random()
            i = rng.randrange(n_nodes)
            if act < 0.25 and i in nodes:
                d = dict(nodes[i], Spec=dict(nodes[i]["Spec"], Availability=rng.choice([0, 1, 2])))
                nodes[i] = d
                p.both("update_node", d)
            elif act < 0.35 and i in nodes:
                p.both("delete_node", nodes[i]["ID"])
                for t in [t for t, nid in placed.items() if nid == nodes[i]["ID"]]:
                    del placed[t]
                del nodes[i]
            elif act < 0.45 and i not in nodes:
                nodes[i] = fz.node_doc(rng, i)
                p.both("create_node", nodes[i])
            elif act < 0.70 and placed:   # the task fails on its node: failure bookkeeping + re-queue by the orchestrator
                t = rng.choice(sorted(placed))
                d = dict(docs[t], NodeID=placed[t], Status={"State": rng.choice([orc.FAILED, orc.REJECTED, "FAILED", orc.COMPLETE, orc.RUNNING])})
                p.both("update_task", d)
                if orc_state(d["Status"]["State"]) > orc.RUNNING:
                    del placed[t]
            elif act < 0.8:
                p.both("advance", rng.choice([1, 30, 200, 400]))
            elif act < 0.9 and placed:
                t = rng.choice(sorted(placed))
                p.both("delete_task", dict(docs[t], NodeID=placed[t], Status={"State": orc.RUNNING}))
                del placed[t]
            elif placed:   # desired state flips past COMPLETE and back: the task stops / resumes counting
                t = rng.choice(sorted(placed))
                p.both("update_task", dict(docs[t], NodeID=placed[t], DesiredState=rng.choice([orc.SHUTDOWN, orc.RUNNING]), Status={"State": orc.RUNNING}))
        if rng.random() < 0.5 and nodes:   # constraint-enforcer sweep over the current cluster
            tbn = {}
            for t, nid in placed.items():
                tbn.setdefault(nid, []).append(dict(docs[t], NodeID=nid, Status={"State": orc.RUNNING}))
            services = {"svc%02d" % k: {"Spec": {"Task": {"Placement": (specs[k].get("Spec") or {}).get("Placement")}}} for k in range(n_svc) if rng.random() < 0.7}
            live = [d for d in nodes.values() if not ((d.get("Description") or {}).get("Resources") or {}).get("Generic")]
            if live:
                p.enforce(live, tbn, services)
    note(p.both("tick"))
    for i in list(nodes)[:6]:
        p.both("node_info", nodes[i]["ID"])
    p.both("node_info", "n-unknown")


def orc_state(v):
    return {"FAILED": orc.FAILED, "REJECTED": orc.REJECTED}.get(v, v) if isinstance(v, str) else v


def test_cxx_wrapper_has_the_twins_surface(monkeypatch):
    """Everything the parity tests and bench.py call on a host scheduler exists on both implementations, and the bulk
    workload path (host.load_workload / parity_util.engine_run) runs through the C++ layer (scripted engine: the
    placements themselves mean nothing here)."""
    public = {n for n in dir(pyhost.PyHostScheduler) if not n.startswith("_") and callable(getattr(pyhost.PyHostScheduler, n))}
    missing = sorted(n for n in public if not hasattr(swsched.Scheduler, n))
    assert not missing, missing
    import parity_util as pu
    from swarmkit_amd import synth
    monkeypatch.setenv("SWP_HOST", "cxx")
    wl = synth.Workload("cfg4", T=300, N=40)
    placed, errs, s, out, hist = pu.engine_run(wl, lib_path=fakelib.build())
    assert isinstance(s, swsched.Scheduler) and len(placed) == wl.T
    assert all(e.startswith("no suitable node") for e in errs.values())


def test_cxx_host_error_behaviour():
    """No exception crosses the C boundary: malformed input comes back as a negative SWP_E* with a message, and the
    scheduler stays usable."""
    import ctypes as C
    lib = fakelib.build()
    s = swsched.Scheduler(engine=abi.Engine(lib_path=lib))
    L, flag, out = s.L, C.c_int(7), C.c_char_p()
    bad = b'{"ID": "t1", "Status": {"State": '
    assert L.swp_sched_create_task(s.h, bad, len(bad), C.byref(flag)) == abi.SWP_EINVAL and b"json" in L.swp_sched_last_error(s.h)
    no_id = b'{"Status": {"State": 64}}'
    assert L.swp_sched_create_task(s.h, no_id, len(no_id), C.byref(flag)) == abi.SWP_EINVAL and b"ID" in L.swp_sched_last_error(s.h)
    odd = b'{"ID": "t1", "Status": {"State": "SLEEPING"}}'
    assert L.swp_sched_create_task(s.h, odd, len(odd), C.byref(flag)) == abi.SWP_EINVAL and b"SLEEPING" in L.swp_sched_last_error(s.h)
    assert L.swp_sched_create_task(None, no_id, len(no_id), C.byref(flag)) == abi.SWP_EINVAL
    assert L.swp_sched_create_task(s.h, None, 0, C.byref(flag)) == abi.SWP_EINVAL
    assert L.swp_sched_node_info(s.h, b"nope", 4, C.byref(out)) == abi.SWP_ENOTFOUND      # errNodeNotFound
    generic = {"ID": "n1", "Description": {"Resources": {"NanoCPUs": 1, "Generic": [{"DiscreteResourceSpec": {"Kind": "gpu", "Value": 1}}]}}}
    s.create_node(generic)                                                                # generic resources are mirrored as counts (round 3)
    assert s.node_info("n1")["AvailableResources"]["Generic"] == [{"Discrete": {"Kind": "gpu", "Value": 1}}]
    named = {"ID": "tg", "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64},
             "Spec": {"Resources": {"Reservations": {"Generic": [{"Named": {"Kind": "gpu", "Value": "a"}}]}}}}
    with pytest.raises(abi.Unsupported):                                                  # a Named reservation: HasEnough's error return, Go path
        s.create_task(named)
    s.delete_node("n1")
    with pytest.raises(abi.SwpError) as ei:                                               # enforcer on a node the nodeSet does not hold
        s.enforce([{"ID": "ghost", "Spec": {"Availability": 0}}], {"ghost": [{"ID": "t", "DesiredState": 512, "Status": {"State": 512}}]})
    assert ei.value.code == abi.SWP_ENOTFOUND
    # still usable afterwards
    s.create_node({"ID": "n2", "Status": {"State": 2}, "Spec": {"Availability": 0}})
    s.set_service("svc")
    assert s.create_task({"ID": "t9", "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64}}) is True
    d = s.tick()
    assert [x["ID"] for x in d] == ["t9"]
    # the pure helpers
    assert swsched.explain([0, 2, 0, 0, 0, 0, 0, 1]) == "insufficient resources on 2 nodes; cannot fulfill requested CSI volume mounts on 1 node"
    assert swsched.parse_constraints(["node.labels.a==b", "oops"]) is None
    assert swsched.parse_constraints([" Node.ID  !=  x y "]) == [("Node.ID", 1, "x y")]


def test_cxx_json_boundary_round_trips_strings_and_numbers():
    """Documents cross the host boundary as JSON: strings (escapes, surrogate pairs, control characters, raw UTF-8) and
    64-bit integers must survive parser and writer unchanged."""
    import ctypes as C
    import json
    s = swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))
    ids = ["plain", "quote\"back\\slash/", "tab\tnl\ncr\r", "ctl\x01\x1f", "é-ſ-K", "astral-\U0001F680-\U00010348", "mixed     end", ""]
    for i, nid in enumerate(ids):
        doc = {"ID": nid, "Status": {"State": 2}, "Spec": {"Availability": 0},
               "Description": {"Resources": {"NanoCPUs": 2**62 + i, "MemoryBytes": 2**53 + 1}}}
        for text in (json.dumps(doc), json.dumps(doc, ensure_ascii=False)):   # \\uXXXX escapes and raw UTF-8
            b = text.encode()
            assert s.L.swp_sched_create_or_update_node(s.h, b, len(b)) == 0, nid
            info = s.node_info(nid)
            assert info["ID"] == nid
            assert info["AvailableResources"] == {"NanoCPUs": 2**62 + i, "MemoryBytes": 2**53 + 1, "Generic": []}
    # numbers: negative, uint64 above int64 (MaxReplicas), exponent form; unknown members are ignored
    t = {"ID": "t", "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64}, "Unknown": [1, {"x": None}, 2.5e3, True],
         "Spec": {"Placement": {"MaxReplicas": 2**64 - 1}, "Resources": {"Reservations": {"NanoCPUs": -5, "MemoryBytes": 1e3}}}}
    d = s.task_desc(t)
    assert int(d["max_replicas"][0]) == 2**64 - 1 and int(d["cpu"][0]) == -5 and int(d["mem"][0]) == 1000
    for bad in [b"", b"{", b'{"ID": "x",}', b'{"ID": "\\ud800"} trailing', b'[1, 2', b'{"a": tru}', b'"\\x"', b"{" * 100 + b"}" * 100]:
        flag = C.c_int()
        assert s.L.swp_sched_create_task(s.h, bad, len(bad), C.byref(flag)) == abi.SWP_EINVAL, bad


# ------------------------------------------------------------------------------------------------ round-2 host-layer fixes
def _both_hosts():
    lib = fakelib.build()
    return [pyhost.PyHostScheduler(engine=abi.Engine(lib_path=lib)), swsched.Scheduler(engine=abi.Engine(lib_path=lib))]


def _task(tid, sid, ver=None, **kw):
    t = dict({"ID": tid, "ServiceID": sid, "DesiredState": 512, "Status": {"State": 64}}, **kw)
    if ver is not None:
        t["SpecVersion"] = {"Index": ver}
    return t


def test_tick_survives_a_refused_device_call():
    """A device call the engine refuses (here: the test double refuses every call that carries a task of service "boom*")
    must not lose the tick: the other groups / tasks are placed, the refused ones come back as Deferred decision lines and
    stay queued (tick was not failure-atomic)."""
    logs = []
    for s in _both_hosts():
        for i in range(4):
            s.create_node({"ID": "n%d" % i, "Status": {"State": 2}, "Spec": {"Availability": 0}})
        for sid in ("good1", "boom-group", "good2", "boom-oneoff", "plain"):
            s.set_service(sid)
        tasks = [_task("g1a", "good1", 1), _task("g1b", "good1", 1), _task("b1", "boom-group", 1), _task("b2", "boom-group", 1),
                 _task("g2a", "good2", 3), _task("o1", "plain"), _task("ob", "boom-oneoff"), _task("o2", "plain")]
        for t in tasks:
            s.create_task(t)
        d1 = s.tick()
        by = {d["ID"]: d for d in d1}
        assert set(by) == {t["ID"] for t in tasks}
        for tid in ("b1", "b2"):
            assert by[tid].get("Deferred") is True and by[tid]["NodeID"] == "" and "refused" in by[tid]["Err"]
        for tid in ("g1a", "g1b", "g2a"):
            assert not by[tid].get("Deferred")
        # one-off run: the refused batch defers every task of that run (one device call), nothing is lost
        assert by["ob"].get("Deferred") is True
        d2 = s.tick()                      # the deferred tasks are still queued
        assert {"b1", "b2", "ob"} <= {d["ID"] for d in d2}
        logs.append((d1, d2, fakelib.take_log(s.e)))
    assert logs[0][0] == logs[1][0] and logs[0][1] == logs[1][1] and logs[0][2] == logs[1][2]   # the twins agree call by call


def test_preassigned_loop_survives_a_refused_check():
    """a throwing swp_check_node in processPreassignedTasks aborted the loop and lost the decisions made so far. The
    refused task stays pending with a Deferred decision line, the others are confirmed."""
    logs = []
    for s in _both_hosts():
        s.create_node({"ID": "n0", "Status": {"State": 2}, "Spec": {"Availability": 0}})
        for sid in ("ok", "boom-pre"):
            s.set_service(sid)
        for tid, sid in (("p1", "ok"), ("pb", "boom-pre"), ("p2", "ok")):
            s.create_task(_task(tid, sid, NodeID="n0"))
        d1 = s.process_preassigned()
        by = {d["ID"]: d for d in d1}
        assert set(by) == {"p1", "pb", "p2"}
        assert by["pb"].get("Deferred") is True and "refused" in by["pb"]["Err"]
        assert not by["p1"].get("Deferred") and not by["p2"].get("Deferred")
        d2 = s.process_preassigned()       # still pending: tried again
        assert "pb" in {d["ID"] for d in d2}
        logs.append((d1, d2, fakelib.take_log(s.e)))
    assert logs[0] == logs[1]


def test_failure_buckets_are_reset_when_cleanup_erases_them():
    """after cleanupFailures (nodeinfo.go:163-183) erased a (service, version) bucket the engine kept the old count
    (>= 5 down-ranks the node for ever). The next tick of that service must push 0."""
    logs = []
    for s in _both_hosts():
        s.create_node({"ID": "n0", "Status": {"State": 2}, "Spec": {"Availability": 0}})
        s.set_service("svc")
        s.set_service("other")
        for k in range(5):                 # five failures of svc on n0
            t = _task("f%d" % k, "svc", NodeID="n0")
            t["Status"] = {"State": 512}
            s.create_task(t)
            s.update_task(dict(t, Status={"State": 704}))   # FAILED
        s.create_task(_task("q1", "svc"))
        s.tick()
        pushed = [l for l in fakelib.take_log(s.e) if l.startswith("failures ")]
        assert pushed and pushed[-1].rstrip().endswith("5"), pushed
        s.advance(6 * 60)                  # past monitorFailures (5 min)
        t = _task("x", "other", NodeID="n0")
        t["Status"] = {"State": 512}
        s.create_task(t)
        s.update_task(dict(t, Status={"State": 704}))       # taskFailed -> cleanupFailures erases svc's bucket
        s.create_task(_task("q2", "svc"))
        s.tick()
        pushed = [l for l in fakelib.take_log(s.e) if l.startswith("failures ") and "svc@" in l]
        assert pushed and pushed[-1].rstrip().endswith("0"), pushed
        logs.append(pushed)
    assert logs[0] == logs[1]


def test_reject_decision_rolls_a_placement_back():
    """scheduler.go:472-487: a decision whose store commit failed is undone — old task back in allTasks and on the queue,
    NodeInfo.removeTask(new) in the engine (one swp_commit(remove))."""
    logs = []
    for s in _both_hosts():
        for i in range(2):
            s.create_node({"ID": "n%d" % i, "Status": {"State": 2}, "Spec": {"Availability": 0}, "Description": {"Resources": {"NanoCPUs": 8 * 10**9, "MemoryBytes": 2**34}}})
        s.set_service("svc")
        s.create_task(_task("t1", "svc", Spec={"Resources": {"Reservations": {"NanoCPUs": 10**9}}}))
        s.create_task(_task("t2", "svc"))
        d = {x["ID"]: x for x in s.tick()}
        assert d["t1"]["NodeID"]
        fakelib.take_log(s.e)
        assert s.reject_decision("t1") is True
        log = fakelib.take_log(s.e)
        assert len(log) == 1 and "commit" in log[0] and "remove" in log[0], log
        assert s.reject_decision("t1") is False            # only once
        assert s.reject_decision("nope") is False
        info = s.node_info(d["t1"]["NodeID"])
        assert "t1" not in info["Tasks"]
        d2 = {x["ID"]: x for x in s.tick()}                # t1 is queued again, t2 is not
        assert "t1" in d2 and (("t2" in d2) == (not d["t2"]["NodeID"]))   # (the double answers at random: t2 may have been unplaceable)
        if "t2" not in d2:
            assert s.reject_decision("t2") is False        # the earlier tick's decisions are final
        logs.append((d, d2, log))
    assert logs[0] == logs[1]


def test_commit_plan_groups_by_node_and_a_stale_node_is_rolled_back_once():
    """SURVEY 8f-3, applySchedulingDecisions (scheduler.go:490-643). The plan hands the tick's decisions back grouped by node with the
    Meta.Version the scheduler's NodeInfo holds (:540), cut into transactions of at most 200 updates (store/memory.go:47); a node whose
    version moved in the store fails ALL its decisions (:533-545): reject_node undoes them in one call — NodeInfo.removeTask for each,
    old task back in allTasks and on the queue — and the next tick sees them again. Both host layers make the same engine calls."""
    logs = []
    for s in _both_hosts():
        for i in range(5):
            s.create_node({"ID": "n%d" % i, "Meta": {"Version": {"Index": 100 + i}}, "Status": {"State": 2}, "Spec": {"Availability": 0},
                           "Description": {"Resources": {"NanoCPUs": 64 * 10**9, "MemoryBytes": 2**40}}})
        s.set_service("svc")
        for j in range(450):
            s.create_task(_task("t%03d" % j, "svc", Spec={"Resources": {"Reservations": {"NanoCPUs": 10**6}}}))
        d = {x["ID"]: x for x in s.tick()}
        plan = s.commit_plan()
        placed = {tid: x["NodeID"] for tid, x in d.items() if x["NodeID"]}
        # every decision appears exactly once, node groups in node order, each with the version the node document carried
        assert sorted(t for g in plan["Nodes"] for t in g["Tasks"]) == sorted(placed)
        assert sorted(plan["Unassigned"]) == sorted(set(d) - set(placed))
        assert [g["NodeID"] for g in plan["Nodes"]] == sorted({n for n in placed.values()})
        for g in plan["Nodes"]:
            assert g["Version"] == 100 + int(g["NodeID"][1:]) and all(placed[t] == g["NodeID"] for t in g["Tasks"])
        flat = [t for tx in plan["Transactions"] for t in tx]
        assert flat == [t for g in plan["Nodes"] for t in g["Tasks"]] + plan["Unassigned"]
        assert all(len(tx) <= 200 for tx in plan["Transactions"]) and len(plan["Transactions"]) == -(-len(d) // 200)
        assert [len(tx) for tx in s.commit_plan(64)["Transactions"]][:-1] == [64] * (len(d) // 64)
        # the store's copy of one node moved on: the version check fails once, for the whole group
        stale = plan["Nodes"][0]
        fakelib.take_log(s.e)
        assert s.reject_node(stale["NodeID"]) == len(stale["Tasks"])
        log = fakelib.take_log(s.e)
        assert len(log) == len(stale["Tasks"]) and all("commit" in l and "remove" in l for l in log)
        assert s.reject_node(stale["NodeID"]) == 0
        info = s.node_info(stale["NodeID"])
        assert not set(stale["Tasks"]) & set(info["Tasks"])
        other = plan["Nodes"][1] if len(plan["Nodes"]) > 1 else None
        if other:
            assert s.reject_decisions(other["Tasks"][:3] + ["nope"]) == min(3, len(other["Tasks"]))
        d2 = {x["ID"] for x in s.tick()}
        assert set(stale["Tasks"]) <= d2
        logs.append((plan, log))
    assert logs[0] == logs[1]


def test_a_generic_kind_listed_twice_keeps_the_tick_on_the_go_path():
    """A node's generic kind changes its type (Named -> Discrete) under a running task that holds a named value; when that task goes
    away, Reclaim + sanitize (resource_management.go:75-153) leave the kind in the node's available list TWICE. HasEnough reads the first
    entry, a claim is subtracted from every entry: one count per kind cannot stand for that list inside a device call, so a tick with a
    task that reserves the kind is handed back whole (every line Deferred, nothing placed) — and ticks go on as before once the list is
    regular again. (Found by the 20 000-seed soak of round 6.)"""
    GIB = 1 << 30
    logs = []
    for s in _both_hosts():
        def node(gen):
            return {"ID": "n0", "Status": {"State": 2}, "Spec": {"Availability": 0},
                    "Description": {"Resources": {"NanoCPUs": 16 * 10**9, "MemoryBytes": 64 * GIB, "Generic": gen}}}
        s.create_node(node([{"Named": {"Kind": "ssd", "Value": "ssd0"}}]))
        for sid in ("want-ssd", "plain"):
            s.set_service(sid)
        spec = {"Spec": {"Resources": {"Reservations": {"Generic": [{"Discrete": {"Kind": "ssd", "Value": 1}}]}}}}
        # a task that holds the named value runs on the node
        s.create_task(dict(_task("held", "want-ssd", NodeID="n0", AssignedGenericResources=[{"Named": {"Kind": "ssd", "Value": "ssd0"}}]), Status={"State": 512}, **spec))
        s.update_node(node([{"Discrete": {"Kind": "ssd", "Value": 2}}]))   # the kind changes its type: the named assignment is ignored (helpers.go remove())
        s.create_task(dict(_task("a", "want-ssd"), **spec))
        d1 = s.tick()
        assert [(d["ID"], d["NodeID"], bool(d.get("Deferred"))) for d in d1] == [("a", "n0", False)]
        placed_a = [d for d in d1 if d["ID"] == "a"][0]
        s.delete_task(dict(_task("held", "want-ssd", NodeID="n0", AssignedGenericResources=[{"Named": {"Kind": "ssd", "Value": "ssd0"}}]), Status={"State": 512}, **spec))
        gen = s.node_info("n0")["AvailableResources"]["Generic"]
        assert [g["Discrete"]["Kind"] for g in gen if "Discrete" in g].count("ssd") == 2, gen   # the list the reference ends up with
        s.create_task(dict(_task("b", "want-ssd"), **spec))
        s.create_task(_task("p", "plain"))
        d2 = s.tick()
        by = {d["ID"]: d for d in d2}
        assert set(by) == {"b", "p"} and all(d.get("Deferred") and d["NodeID"] == "" for d in d2), d2
        assert "more than once" in by["b"]["Err"] and "ssd" in by["b"]["Err"]
        d3 = s.tick()                                                   # still so: the tasks stay queued, nothing is lost
        assert {d["ID"] for d in d3} == {"b", "p"} and all(d.get("Deferred") for d in d3)
        s.update_node(node([{"Discrete": {"Kind": "ssd", "Value": 2}}]))   # createOrUpdateNode rebuilds the list from the description: regular again
        s.create_task(dict(_task("c", "want-ssd"), **spec))
        d4 = s.tick()
        assert {d["ID"] for d in d4} == {"b", "p", "c"} and not any(d.get("Deferred") for d in d4), d4
        logs.append((d1, d2, d3, d4, fakelib.take_log(s.e), placed_a["NodeID"]))
    assert logs[0] == logs[1]   # the twins agree call by call
