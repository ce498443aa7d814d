"""Known-answer tables re-encoded from the reference's unit tests; shared by the oracle tests (CPU)
and the engine tests (GPU). Source lines are given per table."""
import copy

import orc


def base_node():
    """setupEnv(), manager/scheduler/constraint_test.go:16-60."""
    return {"ID": "nodeid-1", "Spec": {"Annotations": {"Labels": {}}, "DesiredRole": 0},
            "Description": {"Engine": {"Labels": {}}}, "Status": {"State": orc.READY, "Addr": "186.17.9.41"}}


def _with(node, **edits):
    n = copy.deepcopy(node)
    for path, v in edits.items():
        d = n
        keys = path.split("__")
        for k in keys[:-1]:
            d = d[k]
        d[keys[-1]] = v
    return n


def constraint_cases():
    """(constraints, node, expected): expected None = SetTask false, else Check result.
    constraint_test.go:62-350."""
    n0 = base_node()
    out = []
    A = out.append
    # TestConstraintSetTask :62-78
    A(([], n0, None))
    # TestWrongSyntax :80-96
    A((["node.abc.bcd == high"], n0, False))
    A((["node.abc.bcd != high"], n0, False))
    # TestNodeHostname :98-118
    c = ["node.hostname != node-1"]
    A((c, n0, True))
    A((c, _with(n0, Description__Hostname="node-2"), True))
    A((c, _with(n0, Description__Hostname="node-1"), False))
    A((c, _with(n0, Description__Hostname="NODe-1"), False))
    # TestNodeIP :120-175
    for cs, req, res in [("node.ip == 186.17.9.41", True, True), ("node.ip != 186.17.9.41", True, False),
                         ("node.ip == 186.17.9.42", True, False), ("node.ip == 186.17.9.4/24", True, True),
                         ("node.ip == 186.17.8.41/24", True, False), ("node.ip == 186.17.9.41/34", True, False),
                         ("node.ip != 266.17.9.41", True, False), ("node.ip != 0.0.0.0", True, True),
                         ("node.ip == ", False, True), ("node.ip == not_ip_addr", True, False)]:
        A(([cs], n0, res if req else None))
    n6 = _with(n0, Status__Addr="2001:db8::2")
    for cs, res in [("node.ip == 2001:db8::2", True), ("node.ip == 2001:db8:0::2", True), ("node.ip != 2001:db8::2/128", False),
                    ("node.ip == 2001:db8::/64", True), ("node.ip == 2001:db9::/64", False), ("node.ip != 2001:db9::/64", True)]:
        A(([cs], n6, res))
    ne = _with(n0, Status__Addr="")
    A((["node.ip == 0.0.0.0"], ne, False))
    A((["node.ip != 0.0.0.0"], ne, True))
    # TestNodeID :177-199
    A((["node.id == nodeid-1"], n0, True))
    A((["node.id == nodeid-1-extra"], n0, False))
    A((["node.id == nodeid-"], n0, False))
    # TestNodeRole :201-223
    A((["node.role == worker"], n0, True))
    A((["node.role == manager"], n0, False))
    A((["node.role == worker-manager"], n0, False))
    A((["node.role == manager"], dict(n0, Role=1), True))
    # TestNodePlatform :225-259
    A((["node.platform.os == linux"], n0, False))
    nl = _with(n0, Description__Platform={"Architecture": "x86_64", "OS": "linux"})
    nw = _with(n0, Description__Platform={"Architecture": "x86_64", "OS": "windows"})
    A((["node.platform.os == linux"], nl, True))
    A((["node.platform.os == linux"], nw, False))
    A((["node.platform.arch == amd64"], nw, False))   # constraint compare is NOT arch-normalised
    A((["node.platform.arch != amd64"], nw, True))
    # TestNodeLabel :261-276
    c = ["node.labels.security == high"]
    A((c, n0, False))
    n1 = _with(n0, Description__Engine__Labels={"security": "high"})
    A((c, n1, False))
    A((c, _with(n1, Spec__Annotations__Labels={"security": "high"}), True))
    # TestEngineLabel :278-297
    c = ["engine.labels.disk != ssd"]
    A((c, n0, True))
    n1 = _with(n0, Spec__Annotations__Labels={"disk": "ssd"})
    A((c, n1, True))
    n2 = _with(n1, Description__Engine__Labels={"disk": "ssd"})
    A((c, n2, False))
    A((c, _with(n2, Description__Engine__Labels={"disk": "ssd", "memory": "large"}), False))
    # TestMultipleConstraints :299-350
    c = ["node.hostname == node-1", "engine.labels.operatingsystem != Ubuntu 14.04"]
    A((c, n0, False))
    n1 = _with(n0, Description__Hostname="node-1")
    A((c, n1, True))
    A((c, _with(n1, Description__Engine__Labels={"operatingsystem": "Ubuntu 14.04"}), False))
    A((c, _with(n1, Description__Engine__Labels={"operatingsystem": "ubuntu 14.04"}), False))
    n2 = _with(n1, Description__Engine__Labels={"operatingsystem": "ubuntu 15.04"})
    A((c, n2, True))
    c3 = c + ["node.labels.security == high"]
    A((c3, n2, False))
    A((c3, _with(n2, Spec__Annotations__Labels={"security": "low"}), False))
    n3 = _with(n2, Spec__Annotations__Labels={"security": "high"})
    A((c3, n3, True))
    A((c3, _with(n3, Description__Engine__Labels={"operatingsystem": "ubuntu 15.04", "memory": "large"}), True))
    # nil-map rules, constraint.go:172-198
    nn = {"ID": "x", "Status": {"State": orc.READY}}
    A((["node.labels.a != b"], nn, True))
    A((["node.labels.a == b"], nn, False))
    A((["engine.labels.a != b"], nn, True))
    A((["node.hostname != h"], nn, True))
    A((["node.platform.os != linux"], nn, True))
    # key match is case-insensitive, label name case-sensitive (constraint.go:109-203)
    A((["NODE.LABELS.Sec == high"], _with(n0, Spec__Annotations__Labels={"Sec": "HIGH"}), True))
    A((["node.labels.sec == high"], _with(n0, Spec__Annotations__Labels={"Sec": "high"}), False))
    return out


# manager/constraint/constraint_test.go:9-72 — (expr, ok, key, exp)
PARSE_CASES = [
    ("", False, None, None), (" ", False, None, None), ("nodeabc", False, None, None), ("node ~ abc", False, None, None),
    ("1node==a2", False, None, None), (" node == node1", True, "node", "node1"), ("no de== node1", False, None, None),
    ("no*de==node1", False, None, None), ("==node1", False, None, None), ("node==", False, None, None),
    ("node== ", False, None, None), ("no$de==node1", False, None, None), ("NoDe==node1", True, "NoDe", "node1"),
    ("no.de==node1", True, "no.de", "node1"), ("_node==_node1", True, "_node", "_node1"),
    ("node==[a-b]+c*(n|b)/", True, "node", "[a-b]+c*(n|b)/"), ("node==node 1", True, "node", "node 1"),
]

# constraint_test.go:74-117 — (expr, what, expected)
MATCH_CASES = [
    ("node.name==foo", "foo", True), ("node.name==foo", "fo", False), ("node.name==foo", "fooE", False),
    ("node.name!=foo", "foo", False), ("node.name!=foo", "bar", True), ("node.name!=foo", "fo", True),
    ("node.name!=foo", "fooExtra", True), ("node.name==f*o", "fo", False), ("node.name==f*o", "f*o", True),
    ("node.name==f*o", "F*o", True), ("node.name==f*o", "foo", False), ("node.name==f.-$o", "fa-$o", False),
    ("node.name==f.-$o", "f.-$o", True),
]
