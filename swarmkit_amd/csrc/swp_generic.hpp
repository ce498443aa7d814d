// swp_generic.hpp — the host layer's half of generic resources: the node's AvailableResources.Generic LIST and what
// NodeInfo.addTask / removeTask / createOrUpdateNode do to it (api/genericresource). The engine decides placements from one
// count per (node, kind) (include/swp.h, swp_node_set_generic); which named values a task is given, what comes back when it
// leaves and what survives a node update is list bookkeeping and stays here, restated from
//   api/genericresource/helpers.go            Kind, GetResource, ConsumeNodeResources (:58-85), remove (:87-111)
//   api/genericresource/resource_management.go Claim (:11-39), selectNodeResources (:41-72), Reclaim (:75-85),
//                                              reclaimResources (:87-117), sanitize (:119-153), sanitizeResource (:155-203)
//   api/genericresource/validate.go            HasEnough's view of a list (:24-52) as counts()
// Lists are value vectors (the reference mutates entries through pointers inside one list only; no entry is shared between lists:
// Claim and Reclaim copy).
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "swp_json.hpp"

namespace swp {
namespace generic {

struct Res {   // api.GenericResource: oneof {NamedResourceSpec{Kind, Value string}, DiscreteResourceSpec{Kind, Value int64}}
    bool named = false;
    std::string kind;
    std::string sval;
    int64_t ival = 0;
    bool operator==(const Res& o) const { return named == o.named && kind == o.kind && sval == o.sval && ival == o.ival; }
};
using List = std::vector<Res>;

// JSON spelling: {"Named": {"Kind": k, "Value": "v"}} / {"Discrete": {"Kind": k, "Value": n}} (the protobuf oneof, Go field names;
// "NamedResourceSpec" / "DiscreteResourceSpec" are accepted too). *is_nil: the field is absent or null (a nil slice in Go).
inline List decode(const json::Value* v, bool* is_nil = nullptr) {
    List out;
    if (is_nil) *is_nil = v == nullptr || !v->is_arr();
    if (v == nullptr || !v->is_arr()) return out;
    for (const json::Value& x : *v->a) {
        Res r;
        const json::Value* n = x.get("Named");
        if (n == nullptr) n = x.get("NamedResourceSpec");
        const json::Value* d = x.get("Discrete");
        if (d == nullptr) d = x.get("DiscreteResourceSpec");
        if (n != nullptr && n->is_obj()) {
            r.named = true;
            r.kind = json::as_str(n->get("Kind"));
            r.sval = json::as_str(n->get("Value"));
        } else if (d != nullptr && d->is_obj()) {
            r.kind = json::as_str(d->get("Kind"));
            r.ival = json::as_i64(d->get("Value"));
        } else
            continue;   // an empty oneof: Kind() is "" and nothing ever matches it
        out.push_back(std::move(r));
    }
    return out;
}
inline json::Value encode(const List& l) {
    json::Value a = json::Value::array();
    for (const Res& r : l) {
        json::Value spec = json::Value::object();
        spec.set("Kind", json::Value::str(r.kind));
        spec.set("Value", r.named ? json::Value::str(r.sval) : json::Value::integer(r.ival));
        json::Value x = json::Value::object();
        x.set(r.named ? "Named" : "Discrete", std::move(spec));
        a.push(std::move(x));
    }
    return a;
}

// GetResource, helpers.go:43-56 — as indices
inline std::vector<size_t> get_resource(const std::string& kind, const List& l) {
    std::vector<size_t> out;
    for (size_t i = 0; i < l.size(); ++i)
        if (l[i].kind == kind) out.push_back(i);
    return out;
}

// remove, helpers.go:87-111: true when `na` is to leave the list
inline bool remove_one(Res& na, const Res& r) {
    if (!r.named) {
        if (na.named) return false;   // type change, ignore
        na.ival -= r.ival;
        return na.ival <= 0;
    }
    if (!na.named) return false;      // type change, ignore
    return r.sval == na.sval;         // not the right item otherwise
}

// HasResource, validate.go:53-85: is there enough of `res` in `resources`
inline bool has_resource(const Res& res, const List& resources) {
    for (const Res& r : resources) {
        if (res.kind != r.kind) continue;
        if (!r.named) {                       // DiscreteResourceSpec
            if (res.named) return false;
            return !(res.ival > r.ival);
        }
        if (!res.named) return false;         // NamedResourceSpec
        if (res.sval != r.sval) continue;
        return true;
    }
    return false;
}

// ConsumeNodeResources, helpers.go:58-85
inline void consume(List* avail, const List& res) {
    List kept;
    for (Res na : *avail) {
        bool gone = false;
        for (const Res& r : res) {
            if (na.kind != r.kind) continue;
            if (remove_one(na, r)) { gone = true; break; }
        }
        if (!gone) kept.push_back(std::move(na));
    }
    *avail = std::move(kept);
}

// selectNodeResources, resource_management.go:41-72; false = the error return
inline bool select_node_resources(const List& node_res, const std::string& kind, int64_t value, List* out) {
    out->clear();
    for (const Res& res : node_res) {
        if (res.kind != kind) continue;
        if (!res.named) {
            if (res.ival >= value && value != 0) {
                Res d;
                d.kind = kind;
                d.ival = value;
                out->push_back(d);
            }
            return true;
        }
        out->push_back(res);
        if ((int64_t)out->size() == value) return true;
    }
    return !out->empty();
}

// Claim, resource_management.go:11-39: an error (a Named reservation, a kind the node has nothing of) leaves everything untouched
inline void claim(List* avail, List* assigned, const List& reservations) {
    List selected;
    for (const Res& res : reservations) {
        if (res.named) return;   // "task should only hold Discrete type"
        List nrs;
        if (!select_node_resources(*avail, res.kind, res.ival, &nrs)) return;
        selected.insert(selected.end(), nrs.begin(), nrs.end());
    }
    assigned->insert(assigned->end(), selected.begin(), selected.end());   // ClaimResources, :31-39
    consume(avail, selected);
}

// sanitizeResource, resource_management.go:155-203: true = the entry is in nodeRes and sane; false + what replaces it
inline bool sanitize_resource(const List& node_res, const Res& res, List* replacement) {
    replacement->clear();
    const std::vector<size_t> nrs = get_resource(res.kind, node_res);
    if (!res.named) {
        auto all = [&]() { for (size_t i : nrs) replacement->push_back(node_res[i]); };
        if (nrs.size() != 1) { all(); return false; }            // type change or removed: reset
        if (node_res[nrs[0]].named) { all(); return false; }      // type change: reset
        if (res.ival > node_res[nrs[0]].ival) { all(); return false; }   // amount change: reset
        return true;
    }
    if (nrs.empty()) return false;   // type change (nothing replaces it)
    for (size_t i : nrs) {
        if (!node_res[i].named) {    // type change: reset
            for (size_t q : nrs) replacement->push_back(node_res[q]);
            return false;
        }
        if (res.sval == node_res[i].sval) return true;
    }
    return false;   // removed
}

// sanitize, resource_management.go:119-153
inline void sanitize(const List& node_res, List* avail) {
    List kept, sanitized;
    std::map<std::string, bool> kind_sanitized;
    for (const Res& na : *avail) {
        List nrs;
        if (!sanitize_resource(node_res, na, &nrs)) {
            if (kind_sanitized.count(na.kind)) continue;
            kind_sanitized[na.kind] = true;
            sanitized.insert(sanitized.end(), nrs.begin(), nrs.end());
            continue;
        }
        kept.push_back(na);
    }
    kept.insert(kept.end(), sanitized.begin(), sanitized.end());
    *avail = std::move(kept);
}

// Reclaim = reclaimResources (:87-117) + sanitize, resource_management.go:75-85
inline void reclaim(List* avail, const List& assigned, const List& node_res) {
    for (const Res& res : assigned) {
        if (res.named) {
            avail->push_back(res);
            continue;
        }
        const std::vector<size_t> nrs = get_resource(res.kind, *avail);
        if (nrs.empty()) avail->push_back(res);   // went down to 0: no longer in the available list
        if (nrs.size() != 1) continue;            // (appended just now, or a type change)
        if ((*avail)[nrs[0]].named) continue;     // type change
        (*avail)[nrs[0]].ival += res.ival;
    }
    sanitize(node_res, avail);
}

// What HasEnough (validate.go:24-52) reads of a list, per kind: the first entry of the kind decides — Discrete: its value,
// Named: how many entries the kind has. Kinds whose count is <= 0 are left out (no request >= 1 fits either way).
inline std::map<std::string, int64_t> counts(const List& avail) {
    std::map<std::string, int64_t> first_discrete, out;
    std::map<std::string, bool> seen_named_first;
    std::map<std::string, int64_t> n_of_kind;
    for (const Res& r : avail) {
        if (!n_of_kind.count(r.kind)) {
            seen_named_first[r.kind] = r.named;
            if (!r.named) first_discrete[r.kind] = r.ival;
        }
        n_of_kind[r.kind] += 1;
    }
    for (const auto& kv : n_of_kind) {
        const int64_t c = seen_named_first[kv.first] ? kv.second : first_discrete[kv.first];
        if (c > 0) out[kv.first] = c;
    }
    return out;
}

// The kinds of a list that ONE count cannot stand for: (a) more than one entry of the kind and not all of them Named, (b) a Named
// value listed twice. Reclaim + sanitize (resource_management.go:75-153) leave such lists behind when a node's description changed
// under a running task and that task goes away — a second Discrete entry next to the first (the kind changed its type), or a task's
// named values appended next to the same values of the fresh description. HasEnough (validate.go:24-52) reads the FIRST entry (or counts
// the entries) while ConsumeNodeResources (helpers.go:87-111) subtracts a claim from EVERY Discrete entry of the kind and removes EVERY
// entry with a claimed name: the list's answer to a second request on the same node is not "count - request" any more. counts() is
// exact for ONE request; a device call that may place several tasks on the node is not: the host layer keeps a tick with tasks that
// reserve such a kind on the Go path (swp_sched.cpp refuseIrregularGeneric).
inline std::set<std::string> irregular_kinds(const List& avail) {
    std::map<std::string, std::pair<int, int>> n;   // kind -> (entries, Discrete entries)
    std::set<std::pair<std::string, std::string>> names;
    std::set<std::string> out;
    for (const Res& r : avail) {
        auto& c = n[r.kind];
        c.first += 1;
        c.second += r.named ? 0 : 1;
        if (r.named && !names.insert({r.kind, r.sval}).second) out.insert(r.kind);
    }
    for (const auto& kv : n)
        if (kv.second.first > 1 && kv.second.second > 0) out.insert(kv.first);
    return out;
}

}  // namespace generic
}  // namespace swp
