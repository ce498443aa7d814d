"""CPU: the host layer's own containers (csrc/swp_json.hpp, the IdTable of csrc/swp_tables.hpp) from the outside — raw JSON text through the
C boundary (the Python wrapper would re-serialise it), and the task maps under create / delete churn that closes the holes of the table.
The engine behind the scheduler is the scripted double (tests/fake_swp.cpp)."""
import ctypes as C
import random

import pytest

import fakelib
from swarmkit_amd import abi, sched as swsched


@pytest.fixture()
def s():
    return swsched.Scheduler(engine=abi.Engine(lib_path=fakelib.build()))


def raw_desc(s, text):
    b = text.encode() if isinstance(text, str) else text
    d = abi.TaskDesc()
    rc = s.L.swp_sched_task_desc(s.h, b, len(b), C.byref(d))
    return rc, d


def task_text(cpu="1", mem="2", extra=""):
    return '{"ID":"t","ServiceID":"s","DesiredState":512,"Status":{"State":64},"Spec":{"Resources":{"Reservations":{"NanoCPUs":%s,"MemoryBytes":%s%s}}}}' % (cpu, mem, extra)


# ------------------------------------------------------------------------------------------------ numbers
@pytest.mark.parametrize("text,want", [
    ("0", 0), ("7", 7), ("-5", -5), ("250000000", 250000000), ("12345678901234567", 12345678901234567), ("-12345678901234567", -12345678901234567),
    ("123456789012345678", 123456789012345678),                      # 18 characters: still the short path
    ("1234567890123456789", 1234567890123456789),                    # 19 digits: strtoll
    ("9223372036854775807", 2**63 - 1), ("-9223372036854775808", -2**63),
    ("1e3", 1000), ("1.5", 1), ("2.5E2", 250), ("-0", 0),
    ("1e308", 2**63 - 1), ("-1e308", -2**63), ("9.3e18", 2**63 - 1),   # a real beyond int64 saturates (the conversion itself would be undefined)
])
def test_integers_and_reals_reach_the_descriptor(s, text, want):
    rc, d = raw_desc(s, task_text(cpu=text))
    assert rc == 0, s.L.swp_sched_last_error(s.h)
    assert d.cpu == want and d.mem == 2


def test_a_uint64_above_int64_keeps_its_bit_pattern(s):
    # MaxReplicas is a uint64 (api/specs.proto): 2^64 - 1 must not become a float
    t = '{"ID":"t","ServiceID":"s","DesiredState":512,"Status":{"State":64},"Spec":{"Placement":{"MaxReplicas":18446744073709551615}}}'
    rc, d = raw_desc(s, t)
    assert rc == 0 and d.max_replicas == 2**64 - 1


@pytest.mark.parametrize("bad", ["-", "--1", "+1", "1-2x", ".", "-e"])
def test_malformed_numbers_are_refused_or_read_as_a_real(s, bad):
    # "-" is refused; the others go to strtod like before (whatever it makes of them): the scheduler answers, it does not crash
    rc, _ = raw_desc(s, task_text(cpu=bad))
    assert rc in (0, abi.SWP_EINVAL)
    if bad == "-":
        assert rc == abi.SWP_EINVAL


# ------------------------------------------------------------------------------------------------ objects
def test_a_repeated_member_keeps_its_last_value(s):
    rc, d = raw_desc(s, task_text(cpu="1", mem="2", extra=',"NanoCPUs":5'))
    assert rc == 0 and d.cpu == 5 and d.mem == 2
    # ... on every level, and a repeated sub-document replaces the earlier one as a whole
    t = '{"ID":"t","ServiceID":"s","Status":{"State":64},"Spec":{"Resources":{"Reservations":{"NanoCPUs":1}}},"Spec":{"Resources":{"Reservations":{"MemoryBytes":9}}}}'
    rc, d = raw_desc(s, t)
    assert rc == 0 and d.cpu == 0 and d.mem == 9


def test_whitespace_and_empty_containers(s):
    t = ' {\n "ID" : "t" ,\t"ServiceID":"s","Status":{ },"Networks":[ ],"Spec":{"Resources":{"Reservations":{ "NanoCPUs" : 3 }},"Placement":{"Constraints":[]}}}\r\n'
    rc, d = raw_desc(s, t)
    assert rc == 0 and d.cpu == 3 and d.constraint_set == 0


@pytest.mark.parametrize("bad", ['{"ID":"t"', '{"ID":"t"}}', '{"ID":"t",}', '{"ID" "t"}', '{"ID":"t\\x"}', '{"ID":"t\\u12"}', '{"ID":"t', '[1,2', '{"a":[1,2,}', 'nul', '',
                                 '{"ID":"t"} x', '{ID:1}'])
def test_malformed_documents_are_refused_and_leave_the_parser_usable(s, bad):
    rc, _ = raw_desc(s, bad)
    assert rc == abi.SWP_EINVAL
    assert b"json" in s.L.swp_sched_last_error(s.h)
    # the members gathered before the failure are gone: the next document is read on its own
    rc, d = raw_desc(s, task_text(cpu="11", mem="12"))
    assert rc == 0 and d.cpu == 11 and d.mem == 12


def test_nesting_is_bounded(s):
    ok = '{"ID":"t","x":' + "[" * 60 + "]" * 60 + "}"
    assert raw_desc(s, ok)[0] == 0
    assert raw_desc(s, '{"ID":"t","x":' + "[" * 70 + "]" * 70 + "}")[0] == abi.SWP_EINVAL
    assert raw_desc(s, '{"ID":"t","x":' + '{"a":' * 70 + "1" + "}" * 70 + "}")[0] == abi.SWP_EINVAL


# ------------------------------------------------------------------------------------------------ strings
BS = chr(92)   # (escapes are put together here so that no tool on the way to this file reads them)
ESCAPED_IDS = [
    ("plain", "plain"),
    (BS + "u0041" + BS + "u00e9", "A" + chr(0xE9)),                 # \u escapes, one of them two UTF-8 bytes
    (BS + "ud83d" + BS + "ude00", chr(0x1F600)),                    # a surrogate pair
    (BS + "ud83dx", chr(0xFFFD) + "x"),                             # half a surrogate pair on its own: U+FFFD, as Go's encoding/json decodes it
    ("x" + BS + "/y" + BS + "n" + BS + "t" + BS + "b" + BS + "f" + BS + "r", "x/y\n\t\b\f\r"),
    (chr(0xE9) + chr(0x4E2D), chr(0xE9) + chr(0x4E2D)),             # raw UTF-8 passes through
]


@pytest.mark.parametrize("raw,value", ESCAPED_IDS)
def test_string_escapes_survive_parse_and_dump(s, raw, value):
    """A task id written with escapes comes back, in the decision line, as the same string."""
    s.create_node(node_doc(0))
    s.set_service("svc")
    text = ('{"ID":"%s","ServiceID":"svc","DesiredState":512,"Status":{"State":64},"Spec":{}}' % raw).encode()
    flag = C.c_int(0)
    assert s.L.swp_sched_create_task(s.h, text, len(text), C.byref(flag)) == 0 and flag.value == 1
    out = s.tick()
    assert [d["ID"] for d in out] == [value]
    assert s.node_info("n000")["Tasks"] == [value]


def _go_coerce(raw):
    """Go's encoding/json on a string that is not UTF-8 (decode.go, unquote): utf8.DecodeRune — an ill-formed sequence is ONE byte long
    and reads as U+FFFD."""
    out, i = [], 0
    while i < len(raw):
        for n in (1, 2, 3, 4):
            try:
                out.append(raw[i:i + n].decode("utf-8"))
                if len(out[-1]) == 1:
                    i += n
                    break
                out.pop()
            except UnicodeDecodeError:
                pass
        else:
            out.append(chr(0xFFFD))
            i += 1
    return "".join(out)


@pytest.mark.parametrize("raw", [b"\xac", b"a\xc3", b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xf8\x88\x80\x80\x80", b"\x80x", b"k\xe2\x82", b"\xe2\x82\xacok\xff"])
def test_strings_that_are_not_utf8_are_coerced_as_go_does(s, raw):
    """What goes in comes out again in the decisions: a string that is not UTF-8 (a stray byte, a cut sequence, an overlong form, a
    surrogate, something beyond U+10FFFF) reads as Go's encoding/json reads it — U+FFFD for every ill-formed byte — so the event is
    scheduled and the output stays JSON (until round 6 the document was refused: ADVICE r5)."""
    s.create_node({"ID": "n000", "Status": {"State": 2}, "Spec": {"Availability": 0}, "Description": {"Resources": {"NanoCPUs": 10**10, "MemoryBytes": 2**34}}})
    s.set_service("svc")
    text = b'{"ID":"' + raw + b'","ServiceID":"svc","DesiredState":512,"Status":{"State":64},"Spec":{}}'
    flag = C.c_int(0)
    assert s.L.swp_sched_create_task(s.h, text, len(text), C.byref(flag)) == 0 and flag.value == 1
    want = _go_coerce(raw)
    assert chr(0xFFFD) in want
    out = s.tick()   # json.loads inside: text that is not JSON would raise
    assert [d["ID"] for d in out] == [want]
    assert s.node_info("n000")["Tasks"] == [want]


def test_escaped_quotes_and_backslashes_are_written_back_escaped(s):
    # a node id with a quote, a backslash and a control character: the decision line must be valid JSON that names it again
    nid = 'n"1\\x\ty'
    s.create_node({"ID": nid, "Status": {"State": 2}, "Spec": {"Availability": 0}, "Description": {"Resources": {"NanoCPUs": 10**10, "MemoryBytes": 2**34}}})
    s.set_service("svc")
    s.create_task({"ID": "t\"1", "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64}, "Spec": {}})
    out = s.tick()   # json.loads inside: invalid text would raise
    assert [(d["ID"], d["NodeID"]) for d in out] == [('t"1', nid)]
    assert s.node_info(nid)["Tasks"] == ['t"1']


# ------------------------------------------------------------------------------------------------ the task maps under churn
def node_doc(i):
    return {"ID": "n%03d" % i, "Status": {"State": 2}, "Spec": {"Availability": 0}, "Description": {"Resources": {"NanoCPUs": 10**13, "MemoryBytes": 2**50}}}


def running(tid, nid):
    return {"ID": tid, "ServiceID": "svc", "NodeID": nid, "DesiredState": 512, "Status": {"State": 512}, "Spec": {"Resources": {"Reservations": {"NanoCPUs": 1000}}}}


@pytest.mark.parametrize("seed", range(3))
def test_all_tasks_survives_the_closing_of_its_holes(s, seed):
    """4 000 tasks known to the scheduler, most of them deleted again in random order (the table closes its holes when more than half of it —
    and more than 1 024 entries — are gone), new ones created in between: afterwards the scheduler knows exactly the survivors."""
    rng = random.Random(seed)
    nodes = ["n%03d" % i for i in range(20)]
    for i in range(20):
        s.create_node(node_doc(i))
    s.set_service("svc")
    alive = {}
    for j in range(4000):
        tid, nid = "t%05d" % j, rng.choice(nodes)
        s.create_task(running(tid, nid))
        alive[tid] = nid
    order = list(alive)
    rng.shuffle(order)
    for k, tid in enumerate(order[:3400]):
        assert s.delete_task(running(tid, alive.pop(tid))) is True
        if k % 400 == 0:   # a newcomer in the middle of the deletions
            tid2, nid2 = "u%05d" % k, rng.choice(nodes)
            s.create_task(running(tid2, nid2))
            alive[tid2] = nid2
    # deleting a task a second time finds nothing on its node
    gone = order[0]
    assert s.delete_task(running(gone, nodes[0])) is False
    by_node = {n: sorted(t for t, x in alive.items() if x == n) for n in nodes}
    for n in nodes:
        info = s.node_info(n)
        assert info["Tasks"] == by_node[n]
        assert info["ActiveTasksCount"] == len(by_node[n])
    # updateTask of a task that failed: true exactly for the tasks allTasks still holds (scheduler.go:283-300)
    for tid in order[:50]:
        t = running(tid, nodes[0])
        t["Status"] = {"State": 640}
        assert s.update_task(t) is False
    for tid in sorted(alive)[:50]:
        t = running(tid, alive[tid])
        t["Status"] = {"State": 640}
        assert s.update_task(t) is True


def test_decisions_of_the_last_tick_are_found_after_many_ticks(s):
    """The decision log is emptied by every tick (one fill, not an erase per entry) and filled again: reject_decision finds exactly the last
    tick's tasks, and the commit plan lists them in id order whatever order they were decided in."""
    for i in range(4):
        s.create_node(node_doc(i))
    s.set_service("svc")
    final = None
    for r in range(6):
        ids = ["r%d-%03d" % (r, k) for k in range(300)]
        random.Random(r).shuffle(ids)
        for tid in ids:
            s.create_task({"ID": tid, "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64}, "Spec": {}})
        out = s.tick()
        decided = sorted(d["ID"] for d in out if d["NodeID"])
        plan = s.commit_plan()
        listed = [t for n in plan["Nodes"] for t in n["Tasks"]]
        assert sorted(listed) == decided
        for n in plan["Nodes"]:
            assert n["Tasks"] == sorted(n["Tasks"])
        if final is not None:
            assert s.reject_decision(final) is False   # a decision of the tick before is final
        final = decided[1] if len(decided) > 1 else None
        if decided:
            assert s.reject_decision(decided[0]) is True
            assert s.reject_decision(decided[0]) is False


# ------------------------------------------------------------------------------------------------ task templates come and go
def spec_task(tid, k):
    # (every k is a service revision of its own: the reservation differs)
    return {"ID": tid, "ServiceID": "svc", "DesiredState": 512, "Status": {"State": 64}, "Spec": {"Resources": {"Reservations": {"NanoCPUs": 1000 + k}}}}


def test_templates_of_revisions_nothing_is_queued_of_are_swept(s):
    """One template per service revision with queued tasks (swp_sched_counts): once there are more than 1 024 and most of them belong to
    revisions the queue holds nothing of, a tick keeps the ones in use — and the tasks it re-queues (no suitable node) are still
    scheduled with THEIR spec afterwards."""
    for i in range(8):
        s.create_node(node_doc(i))
    s.set_service("svc")
    for k in range(1500):
        s.create_task(spec_task("a%04d" % k, k))
    assert s.counts() == {"tasks": 1500, "queued": 1500, "decisions": 0, "templates": 1500}
    out = s.tick()
    waiting = sorted(d["ID"] for d in out if not d["NodeID"])
    assert 100 < len(waiting) < 600            # the double leaves one task in five without a node: they are queued again
    c = s.counts()
    assert c["templates"] == 1500 and c["queued"] == len(waiting) and c["decisions"] == 1500
    for k in range(10):
        s.create_task(spec_task("b%04d" % k, 5000 + k))
    assert s.counts()["templates"] == 1510
    out = s.tick()
    assert sorted(d["ID"] for d in out) == sorted(waiting + ["b%04d" % k for k in range(10)])
    assert s.counts()["templates"] == len(waiting) + 10   # swept: what this tick's queue was of
    # a task keeps the descriptor of ITS revision through the sweep: the double logs the cpu it was asked for
    log = "\n".join(fakelib.take_log(s.e))
    for tid in waiting[:20]:
        assert "cpu=%d " % (1000 + int(tid[1:])) in log
    # a revision that was swept is recognised again from its next event
    before = s.counts()["templates"]
    s.create_task(spec_task("c0000", 7))
    assert s.counts()["templates"] == before + (0 if "a0007" in waiting else 1)   # (kept if a0007 was in the second tick's queue)
    assert [d["ID"] for d in s.tick() if d["ID"] == "c0000"] == ["c0000"]
