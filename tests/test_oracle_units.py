"""Pins the CPU oracle's pure functions against the reference's unit tests (CPU only)."""
import pytest

import kat_tables as kt
import orc


@pytest.mark.parametrize("i", range(len(kt.constraint_cases())))
def test_constraint_filter(i):
    cons, node, want = kt.constraint_cases()[i]
    assert orc.constraint_filter(cons, node) == want, (cons, node)


@pytest.mark.parametrize("expr,ok,key,exp", kt.PARSE_CASES)
def test_parse(expr, ok, key, exp):
    parsed, err = orc.constraint_parse([expr])
    assert (parsed is not None) == ok, err
    if ok:
        assert parsed[0][0] == key and parsed[0][2] == exp


@pytest.mark.parametrize("expr,what,want", kt.MATCH_CASES)
def test_match(expr, what, want):
    assert orc.constraint_match(expr, what) == want


def test_equal_fold_specials():
    assert orc.equal_fold("Kelvin", "Kelvin") and orc.equal_fold("sS", "ſs")
    assert not orc.equal_fold("abc", "abd") and not orc.equal_fold("a", "ab")


def _gen(named=(), discrete=()):
    return [{"Named": {"Kind": k, "Value": v}} for k, v in named] + [{"Discrete": {"Kind": k, "Value": v}} for k, v in discrete]


def test_remove_task():
    """TestRemoveTask, nodeinfo_test.go:11-99."""
    node = {"Description": {"Resources": {"NanoCPUs": 100000, "MemoryBytes": 1000000,
                                          "Generic": _gen([("orange", c) for c in ("blue", "red", "green")] + [("orange", "x")][:0], [("apple", 6)])}}}
    node["Description"]["Resources"]["Generic"] = _gen([("orange", "orange"), ("orange", "blue"), ("orange", "red"), ("orange", "green")][1:], [("apple", 6)])
    avail = {"NanoCPUs": 100000, "MemoryBytes": 1000000, "Generic": _gen([("orange", "blue"), ("orange", "red")], [("apple", 5)])}
    # reference: node has orange{blue,red,green}+... ; available orange{blue,red}... the test uses NewSet("orange","blue","red","green") = kind orange, values blue/red/green
    node["Description"]["Resources"]["Generic"] = _gen([("orange", "blue"), ("orange", "red"), ("orange", "green")], [("apple", 6)])
    task1 = {"ID": "task1", "Spec": {"Resources": {"Reservations": {"NanoCPUs": 5000, "MemoryBytes": 5000,
                                                                    "Generic": _gen((), [("apple", 1), ("orange", 1)])}}},
             "AssignedGenericResources": _gen([("orange", "green")], [("apple", 1)])}
    r = orc.nodeinfo_ops(node, avail, [], [["remove", task1]])
    assert r["results"] == [False]   # nodeInfo has no tasks
    r = orc.nodeinfo_ops(node, avail, [{"ID": "task1"}, {"ID": "task2"}], [["remove", task1], ["remove", {"ID": "task3"}]])
    assert r["results"] == [True, False]
    ar = r["info"]["AvailableResources"]
    assert ar["NanoCPUs"] == 105000 and ar["MemoryBytes"] == 1005000
    apples = [g for g in ar["Generic"] if "Discrete" in g and g["Discrete"]["Kind"] == "apple"]
    oranges = sorted(g["Named"]["Value"] for g in ar["Generic"] if "Named" in g and g["Named"]["Kind"] == "orange")
    assert len(apples) == 1 and apples[0]["Discrete"]["Value"] == 6
    assert oranges == ["blue", "green", "red"]


def test_add_task():
    """TestAddTask, nodeinfo_test.go:101-172."""
    avail = {"NanoCPUs": 100000, "MemoryBytes": 1000000, "Generic": _gen([("orange", "blue"), ("orange", "red")], [("apple", 5)])}
    task3 = {"ID": "task3", "Spec": {"Resources": {"Reservations": {"NanoCPUs": 5000, "MemoryBytes": 5000,
                                                                    "Generic": _gen((), [("apple", 2), ("orange", 1)])}}}}
    r = orc.nodeinfo_ops({}, avail, [{"ID": "task1"}, {"ID": "task2"}], [["add", {"ID": "task1"}], ["add", task3], ["add", task3]])
    assert r["results"] == [False, True, False]
    ar = r["info"]["AvailableResources"]
    assert ar["NanoCPUs"] == 95000 and ar["MemoryBytes"] == 995000
    apples = [g for g in ar["Generic"] if "Discrete" in g]
    oranges = [g for g in ar["Generic"] if "Named" in g]
    assert len(apples) == 1 and apples[0]["Discrete"]["Value"] == 3
    assert len(oranges) == 1 and oranges[0]["Named"]["Value"] in ("blue", "red")


def test_tree_task_counts():
    """TestTreeTaskCountConsistency, nodeset_test.go:9-163."""
    def n(i, labels, c):
        return {"Node": {"ID": f"node{i}", "Spec": {"Annotations": {"Labels": labels}}}, "ByService": {"service1": c}}
    nodes = [n(1, {"datacenter": "dc1", "rack": "r1"}, 3), n(2, {"datacenter": "dc1", "rack": "r2"}, 2),
             n(3, {"datacenter": "dc2", "rack": "r2"}, 4), n(4, {}, 2), n(5, {}, 1)]
    t = orc.tree(nodes, "service1", ["node.labels.datacenter", "node.labels.rack"], 10)
    assert t["tasks"] == 12
    assert t["next"]["dc1"]["tasks"] == 5 and t["next"]["dc1"]["next"]["r1"]["tasks"] == 3 and t["next"]["dc1"]["next"]["r2"]["tasks"] == 2
    assert t["next"]["dc2"]["tasks"] == 4 and t["next"]["dc2"]["next"]["r2"]["tasks"] == 4
    assert t["next"][""]["tasks"] == 3 and t["next"][""]["next"][""]["tasks"] == 3

    def check(d):
        if d["next"] is None:
            return d["tasks"]
        s = sum(check(c) for c in d["next"].values())
        assert s == d["tasks"]
        return s
    check(t)
