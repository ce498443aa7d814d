"""The host layer's half of generic resources, Python twin of csrc/swp_generic.hpp: the node's AvailableResources.Generic LIST and
what NodeInfo.addTask / removeTask / createOrUpdateNode do to it. Restated from api/genericresource:
  helpers.go             Kind, GetResource, ConsumeNodeResources (:58-85), remove (:87-111)
  resource_management.go Claim (:11-39), selectNodeResources (:41-72), Reclaim (:75-85), reclaimResources (:87-117),
                         sanitize (:119-153), sanitizeResource (:155-203)
  validate.go            HasEnough's view of a list (:24-52) as counts()
An entry is a tuple (named: bool, kind: str, value: str | int). Lists are plain Python lists of such tuples; every function returns
new lists (the reference mutates entries through pointers inside one list only)."""


def decode(v):
    """JSON spelling -> (list, is_nil). {"Named": {"Kind", "Value"}} / {"Discrete": {"Kind", "Value"}} (the *ResourceSpec names too)."""
    if not isinstance(v, list):
        return [], True
    out = []
    for x in v:
        n = x.get("Named", x.get("NamedResourceSpec"))
        d = x.get("Discrete", x.get("DiscreteResourceSpec"))
        if isinstance(n, dict):
            out.append((True, n.get("Kind", ""), n.get("Value", "")))
        elif isinstance(d, dict):
            out.append((False, d.get("Kind", ""), int(d.get("Value", 0))))
    return out, False


def encode(lst):
    return [{"Named" if named else "Discrete": {"Kind": kind, "Value": val}} for named, kind, val in lst]


def _remove(na, r):
    """remove, helpers.go:87-111 -> (entry after the call, leaves the list)."""
    if not r[0]:
        if na[0]:
            return na, False
        left = na[2] - r[2]
        return (False, na[1], left), left <= 0
    if not na[0]:
        return na, False
    return na, r[2] == na[2]


def has_resource(res, resources):
    """HasResource, validate.go:53-85: is there enough of `res` in `resources` (entries are (named, kind, value))."""
    for r in resources:
        if res[1] != r[1]:
            continue
        if not r[0]:                      # DiscreteResourceSpec
            if res[0]:
                return False
            return not (res[2] > r[2])
        if not res[0]:                    # NamedResourceSpec
            return False
        if res[2] != r[2]:
            continue
        return True
    return False


def consume(avail, res):
    """ConsumeNodeResources, helpers.go:58-85."""
    kept = []
    for na in avail:
        gone = False
        for r in res:
            if na[1] != r[1]:
                continue
            na, gone = _remove(na, r)
            if gone:
                break
        if not gone:
            kept.append(na)
    return kept


def select_node_resources(node_res, kind, value):
    """selectNodeResources, resource_management.go:41-72 -> list, or None for the error return."""
    out = []
    for res in node_res:
        if res[1] != kind:
            continue
        if not res[0]:
            if res[2] >= value and value != 0:
                out.append((False, kind, value))
            return out
        out.append(res)
        if len(out) == value:
            return out
    return out if out else None


def claim(avail, reservations):
    """Claim, resource_management.go:11-39 -> (available list afterwards, what the task was assigned)."""
    selected = []
    for res in reservations:
        if res[0]:
            return avail, []
        nrs = select_node_resources(avail, res[1], res[2])
        if nrs is None:
            return avail, []
        selected.extend(nrs)
    return consume(avail, selected), selected


def _sanitize_resource(node_res, res):
    """sanitizeResource, resource_management.go:155-203 -> (sane, replacement)."""
    nrs = [r for r in node_res if r[1] == res[1]]
    if not res[0]:
        if len(nrs) != 1 or nrs[0][0] or res[2] > nrs[0][2]:
            return False, nrs
        return True, []
    if not nrs:
        return False, []
    for nr in nrs:
        if not nr[0]:
            return False, nrs
        if res[2] == nr[2]:
            return True, []
    return False, []


def sanitize(node_res, avail):
    """sanitize, resource_management.go:119-153."""
    kept, sanitized, seen = [], [], set()
    for na in avail:
        ok, nrs = _sanitize_resource(node_res, na)
        if not ok:
            if na[1] in seen:
                continue
            seen.add(na[1])
            sanitized.extend(nrs)
            continue
        kept.append(na)
    return kept + sanitized


def reclaim(avail, assigned, node_res):
    """Reclaim = reclaimResources (:87-117) + sanitize, resource_management.go:75-85."""
    avail = list(avail)
    for res in assigned:
        if res[0]:
            avail.append(res)
            continue
        idx = [i for i, r in enumerate(avail) if r[1] == res[1]]
        if not idx:
            avail.append(res)
        if len(idx) != 1:
            continue
        i = idx[0]
        if avail[i][0]:
            continue
        avail[i] = (False, avail[i][1], avail[i][2] + res[2])
    return sanitize(node_res, avail)


def irregular_kinds(avail):
    """The kinds ONE count cannot stand for (csrc/swp_generic.hpp irregular_kinds): more than one entry of the kind and not all of them
    Named, or a Named value listed twice — what Reclaim + sanitize leave behind when a node's description changed under a running task."""
    n, names, out = {}, set(), set()
    for named, kind, val in avail:
        c = n.setdefault(kind, [0, 0])
        c[0] += 1
        c[1] += 0 if named else 1
        if named:
            if (kind, val) in names:
                out.add(kind)
            names.add((kind, val))
    return out | {k for k, c in n.items() if c[0] > 1 and c[1] > 0}


def counts(avail):
    """kind -> what HasEnough (validate.go:24-52) compares a request with: the first entry of the kind decides — Discrete: its
    value, Named: how many entries the kind has. Kinds whose count is <= 0 are left out. Sorted by kind."""
    first, n = {}, {}
    for named, kind, val in avail:
        if kind not in first:
            first[kind] = (named, val)
        n[kind] = n.get(kind, 0) + 1
    out = {}
    for kind in sorted(n):
        c = n[kind] if first[kind][0] else first[kind][1]
        if c > 0:
            out[kind] = c
    return out
