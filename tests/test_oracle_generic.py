"""Oracle KATs: api/genericresource/{resource_management,helpers,validate}_test.go re-encoded, test by test.
(parse_test.go covers ParseCmd, the CLI flag parser — not on the scheduling path, not restated.)"""
import orc


def S(kind, *vals):      # genericresource.NewSet
    return [{"Named": {"Kind": kind, "Value": v}} for v in vals]


def D(kind, n):          # genericresource.NewDiscrete
    return {"Discrete": {"Kind": kind, "Value": n}}


def get(kind, lst):      # genericresource.GetResource
    return [r for r in lst if (r.get("Named") or r.get("Discrete"))["Kind"] == kind]


def has(kind, val, lst):  # HasResource(NewString(kind, val), lst)
    return orc.generic("has_resource", node=lst, res=S(kind, val))["ok"]


def dval(r):
    return r["Discrete"]["Value"]


# ---- helpers_test.go ---------------------------------------------------------------------------
def test_consume_resources_single():      # :10-29
    node = S("apple", "red", "orange", "blue")
    node = orc.generic("consume", node=node, res=S("apple", "red"))["node"]
    assert len(node) == 2
    node = orc.generic("consume", node=node + [D("apple", 1)], res=[D("apple", 1)])["node"]
    assert len(node) == 2
    node = orc.generic("consume", node=node + [D("apple", 4)], res=[D("apple", 1)])["node"]
    assert len(node) == 3 and dval(node[2]) == 3


def test_consume_resources_multiple():    # :31-66
    node = S("apple", "red", "orange", "blue", "green", "yellow") + [D("orange", 5), D("banana", 3)]
    node += S("grape", "red", "orange", "blue", "green", "yellow") + [D("cakes", 3)]
    res = S("apple", "red") + [D("banana", 2)] + S("apple", "green", "blue", "red") + S("grape", "red", "blue", "red") + [D("cakes", 3)]
    node = orc.generic("consume", node=node, res=res)["node"]
    assert len(node) == 7
    apples, oranges, bananas, grapes = get("apple", node), get("orange", node), get("banana", node), get("grape", node)
    assert (len(apples), len(oranges), len(bananas), len(grapes)) == (2, 1, 1, 3)
    assert all(has("apple", k, apples) for k in ("yellow", "orange"))
    assert all(has("grape", k, grapes) for k in ("yellow", "orange", "green"))
    assert dval(oranges[0]) == 5 and dval(bananas[0]) == 1


# ---- validate_test.go --------------------------------------------------------------------------
def test_has_resource_discrete():         # :10-29
    assert orc.generic("has_resource", node=[D("apple", 5)], res=[D("apple", 1)])["ok"]
    assert orc.generic("has_resource", node=[D("apple", 5)], res=[D("apple", 5)])["ok"]
    assert not orc.generic("has_resource", node=[D("apple", 5)], res=[D("apple", 6)])["ok"]


# ---- resource_management_test.go ---------------------------------------------------------------
def test_claim_single_discrete():         # :10-24
    r = orc.generic("claim", node=[D("apple", 3)], res=[D("apple", 2)])
    assert len(r["node"]) == 1 and len(r["assigned"]) == 1
    assert dval(r["node"][0]) == 1 and dval(r["assigned"][0]) == 2


def test_claim_multiple_discrete():       # :26-51
    r = orc.generic("claim", node=[D("apple", 3), D("orange", 4), D("banana", 2), D("cake", 1)], res=[D("orange", 4), D("apple", 2)])
    assert len(r["node"]) == 3 and len(r["assigned"]) == 2
    apples, oranges = get("apple", r["assigned"]), get("orange", r["assigned"])
    assert len(apples) == 1 and len(oranges) == 1 and dval(apples[0]) == 2 and dval(oranges[0]) == 4


def test_claim_single_str():              # :53-68
    r = orc.generic("claim", node=S("apple", "red", "orange", "blue", "green"), res=[D("apple", 2)])
    assert len(r["node"]) == 2 and len(r["assigned"]) == 2
    assert all(has("apple", k, r["assigned"]) for k in ("red", "orange"))


def test_claim_multiple_str():            # :70-94
    node = S("apple", "red", "orange", "blue", "green") + S("oranges", "red", "orange", "blue", "green") + S("bananas", "red", "orange", "blue", "green")
    r = orc.generic("claim", node=node, res=[D("oranges", 4), D("apple", 2)])
    assert len(r["node"]) == 6 and len(r["assigned"]) == 6
    assert all(has("apple", k, get("apple", r["assigned"])) for k in ("red", "orange"))
    assert all(has("oranges", k, get("oranges", r["assigned"])) for k in ("red", "orange", "blue", "green"))


def test_reclaim_single_discrete():       # :96-112
    node = orc.generic("reclaim_resources", node=[], assigned=[D("apple", 2)])["node"]
    assert len(node) == 1 and dval(node[0]) == 2
    node = orc.generic("reclaim_resources", node=node, assigned=[D("apple", 2)])["node"]
    assert len(node) == 1 and dval(node[0]) == 4


def test_reclaim_multiple_discrete():     # :114-139
    node = orc.generic("reclaim_resources", node=[D("apple", 3), D("banana", 2)], assigned=[D("orange", 4), D("apple", 2)])["node"]
    assert len(node) == 3
    assert [dval(get(k, node)[0]) for k in ("apple", "orange", "banana")] == [5, 4, 2]


def test_reclaim_single_str():            # :141-162
    node = orc.generic("reclaim_resources", node=[], assigned=S("apple", "red", "orange"))["node"]
    assert len(node) == 2 and all(has("apple", k, node) for k in ("red", "orange"))
    node = orc.generic("reclaim_resources", node=node, assigned=S("apple", "blue", "red"))["node"]
    assert len(node) == 4 and all(has("apple", k, node) for k in ("red", "orange", "blue"))


def test_reclaim_multiple_str():          # :164-185
    node = orc.generic("reclaim_resources", node=S("orange", "green"), assigned=S("apple", "red", "orange") + S("orange", "red", "orange"))["node"]
    assert len(node) == 5
    apples, oranges = get("apple", node), get("orange", node)
    assert len(apples) == 2 and len(oranges) == 3
    assert all(has("apple", k, apples) for k in ("red", "orange")) and all(has("orange", k, oranges) for k in ("red", "orange", "green"))


def test_reclaim_resources():             # :187-230
    node = S("orange", "green", "blue") + [D("apple", 3)] + S("banana", "red", "orange", "green") + [D("cake", 2)]
    assigned = S("orange", "red", "orange") + S("grape", "red", "orange") + [D("apple", 3), D("coffe", 2)]
    node = orc.generic("reclaim_resources", node=node, assigned=assigned)["node"]
    assert len(node) == 12
    assert [len(get(k, node)) for k in ("apple", "orange", "banana", "cake", "grape", "coffe")] == [1, 4, 3, 1, 2, 1]
    assert dval(get("apple", node)[0]) == 6 and dval(get("cake", node)[0]) == 2 and dval(get("coffe", node)[0]) == 2
    assert all(has("orange", k, get("orange", node)) for k in ("red", "orange", "green", "blue"))
    assert all(has("banana", k, get("banana", node)) for k in ("red", "orange", "green"))
    assert all(has("grape", k, get("grape", node)) for k in ("red", "orange"))


def test_sanitize_discrete():             # :232-269
    avail = orc.generic("sanitize", node_res=[], node=[D("orange", 4)])["node"]
    assert avail == []
    avail = orc.generic("sanitize", node_res=[D("orange", 6)], node=[D("orange", 4)])["node"]
    assert len(avail) == 1 and dval(avail[0]) == 4
    avail = orc.generic("sanitize", node_res=[D("orange", 4)], node=avail)["node"]
    assert len(avail) == 1 and dval(avail[0]) == 4
    avail = orc.generic("sanitize", node_res=[D("orange", 2)], node=avail)["node"]
    assert len(avail) == 1 and dval(avail[0]) == 2
    node_res = [D("orange", 2), D("banana", 6), D("cake", 6)]
    avail = orc.generic("sanitize", node_res=node_res, node=avail + [D("cake", 2), D("apple", 4), D("banana", 8)])["node"]
    assert [dval(x) for x in avail] == [2, 2, 6]      # oranges, cake, banana (apple is gone)


def test_sanitize_str():                  # :271-290
    assert orc.generic("sanitize", node_res=[], node=S("apple", "red", "orange", "blue"))["node"] == []
    avail = orc.generic("sanitize", node_res=S("apple", "red", "orange", "blue", "green"), node=S("apple", "red", "orange", "blue"))["node"]
    assert len(avail) == 3
    avail = orc.generic("sanitize", node_res=S("apple", "red", "orange", "blue"), node=avail)["node"]
    assert len(avail) == 3
    avail = orc.generic("sanitize", node_res=S("apple", "red", "orange"), node=avail)["node"]
    assert len(avail) == 2


def test_sanitize_change_discrete_to_set():   # :292-338
    avail = orc.generic("sanitize", node_res=S("apple", "red"), node=[D("apple", 5)])["node"]
    assert len(avail) == 1 and avail[0]["Named"]["Value"] == "red"
    node_res = S("apple", "red", "orange", "green")
    avail = orc.generic("sanitize", node_res=node_res, node=[D("apple", 5)])["node"]
    assert len(avail) == 3 and all(has("apple", k, avail) for k in ("red", "orange", "green"))
    node_res = node_res + S("orange", "red", "orange", "green") + S("cake", "red", "orange", "green")
    avail = orc.generic("sanitize", node_res=node_res, node=S("apple", "green") + [D("cake", 3)] + S("orange", "orange", "blue"))["node"]
    assert len(avail) == 5
    apples, oranges, cakes = get("apple", avail), get("orange", avail), get("cake", avail)
    assert (len(apples), len(oranges), len(cakes)) == (1, 1, 3)
    assert has("apple", "green", apples) and has("orange", "orange", oranges) and all(has("cake", k, cakes) for k in ("red", "orange", "green"))


def test_sanitize_change_set_to_discrete():   # :340-378
    avail = orc.generic("sanitize", node_res=[D("apple", 5)], node=S("apple", "red"))["node"]
    assert len(avail) == 1 and dval(avail[0]) == 5
    avail = orc.generic("sanitize", node_res=[D("apple", 5)], node=S("apple", "red", "orange", "green"))["node"]
    assert len(avail) == 1 and dval(avail[0]) == 5
    node_res = [D("apple", 5), D("orange", 3), D("cake", 1)]
    avail = orc.generic("sanitize", node_res=node_res, node=avail + [D("cake", 2)] + S("orange", "orange", "blue"))["node"]
    assert len(avail) == 3
    assert [dval(get(k, avail)[0]) for k in ("apple", "orange", "cake")] == [5, 3, 1]
