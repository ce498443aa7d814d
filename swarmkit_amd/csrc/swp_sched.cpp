// swp_sched.cpp — host side above the C ABI (include/swp_sched.h): manager/scheduler.Scheduler's event handlers and
// tick over the engine, in C++ with the reference's names. In swarmkit itself this layer is a cgo shim in Go
// (INTEGRATION.md); Go is not available in this build environment, so it is restated here.
//
// NO placement logic lives in this file: every "which node" decision is made by the kernels behind
// swp_schedule_batch / swp_schedule_groups / swp_check_node / swp_enforce. This layer keeps what a nodeSet keeps
// beside the numeric columns (node documents, NodeInfo.Tasks, failure timestamps), mirrors the mutators into the
// engine and converts numeric answers into decisions and Status.Err strings.
//
// Mirrors (paths under /root/reference/manager/scheduler/):
//   scheduler.go:254-366  createTask / updateTask / deleteTask          scheduler.go:368-396  createOrUpdateNode
//   scheduler.go:398-426  processPreassignedTasks   :646-690 taskFitNode     :429-488 tick     :928-971 noSuitableNode
//   nodeinfo.go:66-221    removeTask / addTask / taskFailed / countRecentFailures
//   pipeline.go:76-103    SetTask / Explain          filter.go  every Filter.SetTask
//   manager/constraint/constraint.go:40-81 Parse;   :109-203 the key dispatch of Match
#include <arpa/inet.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <list>
#include <map>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/swp_sched.h"
#include "swp_generic.hpp"
#include "swp_json.hpp"
#include "swp_tables.hpp"

namespace swp {

using json::Value;
using json::as_i64;
using json::as_str;
using json::at;
using json::truthy;

// api/types.proto:510-539 (TaskState), :190-207 (NodeStatus.State), specs.proto (Availability, Role, Mount.Type, PublishMode)
enum : int64_t { NEW = 0, PENDING = 64, ASSIGNED = 192, RUNNING = 512, COMPLETE = 576, SHUTDOWN = 640, FAILED = 704, REJECTED = 768 };
static constexpr int64_t NODE_STATE_READY = 2, AVAILABILITY_ACTIVE = 0;
static constexpr int64_t MOUNT_VOLUME = 1, MOUNT_CLUSTER = 4, PUBLISH_HOST = 1;
static constexpr int64_t MONITOR_FAILURES = 5LL * 60 * 1000000000LL;   // scheduler.go:19

struct Fail {   // carries an SWP_E* code to the C boundary
    int code;
    std::string msg;
};
[[noreturn]] static void fail(int code, std::string msg) { throw Fail{code, std::move(msg)}; }
[[noreturn]] static void unsupported(const char* what) { fail(SWP_EUNSUPPORTED, what); }

// ------------------------------------------------------------------------------------------------ small helpers
// protobuf enums travel either as their number or as their name; an unknown name is an error for the enums whose
// value is ordered (task states, protocols) and simply "none of the values tested" (-1) for the others
static int64_t enum_value(const Value* v, std::initializer_list<std::pair<const char*, int64_t>> names, int64_t def = 0, bool strict = false) {
    if (v == nullptr) return def;
    if (v->is_str()) {
        for (const auto& n : names)
            if (v->s == n.first) return n.second;
        if (strict) fail(SWP_EINVAL, "unknown enum name '" + v->s + "'");
        return -1;
    }
    return as_i64(v, def);
}
// a - b as Go computes it on int64: two's complement, wrapping
static inline int64_t wrap_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static int64_t task_state(const Value* v) {
    return enum_value(v, {{"NEW", 0}, {"PENDING", 64}, {"ASSIGNED", 192}, {"ACCEPTED", 256}, {"PREPARING", 320}, {"READY", 384}, {"STARTING", 448},
                          {"RUNNING", 512}, {"COMPLETE", 576}, {"SHUTDOWN", 640}, {"FAILED", 704}, {"REJECTED", 768}, {"REMOVE", 800}, {"ORPHANED", 832}}, 0, true);
}
static const std::string& task_id(const Value& t) {
    const Value* id = t.get("ID");
    if (id == nullptr || !id->is_str()) fail(SWP_EINVAL, "document without ID");
    return id->s;
}

// UTF-8 ⇄ code points (documents are valid UTF-8: they come from a JSON parser)
static std::u32string decode(const std::string& s) {
    std::u32string out;
    size_t i = 0;
    while (i < s.size()) {
        unsigned char c = (unsigned char)s[i];
        uint32_t cp = c;
        int extra = 0;
        if (c >= 0xF0) { cp = c & 0x07; extra = 3; }
        else if (c >= 0xE0) { cp = c & 0x0F; extra = 2; }
        else if (c >= 0xC0) { cp = c & 0x1F; extra = 1; }
        ++i;
        for (int k = 0; k < extra && i < s.size(); ++k, ++i) cp = (cp << 6) | ((unsigned char)s[i] & 0x3F);
        out.push_back((char32_t)cp);
    }
    return out;
}
static std::string encode(const std::u32string& s, size_t from = 0, size_t to = std::string::npos) {
    std::string out;
    to = std::min(to, s.size());
    for (size_t i = from; i < to; ++i) {
        uint32_t cp = s[i];
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    return out;
}

// strings.EqualFold restricted to what it is applied to on this path: one side is always an ASCII literal
// ("node.id", "node.labels." …), so only code points that fold ONTO ASCII matter: A-Z, U+212A KELVIN SIGN → k and
// U+017F LATIN SMALL LETTER LONG S → s (Unicode simple case folding, constraint.go:110-188).
static char32_t fold_cp(char32_t c) {
    if (c >= U'A' && c <= U'Z') return c + 32;
    if (c == 0x212A) return U'k';
    if (c == 0x017F) return U's';
    return c;
}
static bool equal_fold(const std::u32string& a, size_t a_len, const char* lit) {
    size_t n = std::strlen(lit);
    if (a_len != n) return false;
    for (size_t i = 0; i < n; ++i)
        if (fold_cp(a[i]) != (char32_t)(unsigned char)lit[i]) return false;
    return true;
}
static bool equal_fold(const std::u32string& a, const std::u32string& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (fold_cp(a[i]) != fold_cp(b[i])) return false;
    return true;
}

// ---- constraint.Parse (manager/constraint/constraint.go:40-81) ----------------------------------------------------
// key   `^(?i)[a-z_][a-z0-9\-_.]+$`   value `^(?i)[a-z0-9:\-_\s\.\*\(\)\?\+\[\]\\\^\$\|\/]+$`  (constraint.go:23-26);
// under (?i) RE2 also lets U+212A and U+017F match k / s. strings.TrimSpace trims Unicode White_Space.
static bool is_go_space(char32_t c) {
    switch (c) {
        case 0x09: case 0x0A: case 0x0B: case 0x0C: case 0x0D: case 0x20: case 0x85: case 0xA0: case 0x1680:
        case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
            return true;
        default:
            return c >= 0x2000 && c <= 0x200A;
    }
}
static bool is_alpha_i(char32_t c) { return (c >= U'a' && c <= U'z') || (c >= U'A' && c <= U'Z') || c == 0x212A || c == 0x017F; }
static bool is_digit(char32_t c) { return c >= U'0' && c <= U'9'; }
static bool key_ok(const std::u32string& k) {
    if (k.size() < 2) return false;
    if (!(is_alpha_i(k[0]) || k[0] == U'_')) return false;
    for (size_t i = 1; i < k.size(); ++i) {
        char32_t c = k[i];
        if (!(is_alpha_i(c) || is_digit(c) || c == U'-' || c == U'_' || c == U'.')) return false;
    }
    return true;
}
static bool value_ok(const std::u32string& v) {
    if (v.empty()) return false;
    for (char32_t c : v) {
        if (is_alpha_i(c) || is_digit(c)) continue;
        switch (c) {
            case U':': case U'-': case U'_': case U'\t': case U'\n': case U'\f': case U'\r': case U' ': case U'.': case U'*':
            case U'(': case U')': case U'?': case U'+': case U'[': case U']': case U'\\': case U'^': case U'$': case U'|': case U'/':
                continue;
            default:
                return false;
        }
    }
    return true;
}
struct Expr {
    std::u32string key;
    int op;   // 0 "==", 1 "!="
    std::u32string exp;
};
static std::u32string trim(const std::u32string& s, size_t from, size_t to) {
    while (from < to && is_go_space(s[from])) ++from;
    while (to > from && is_go_space(s[to - 1])) --to;
    return s.substr(from, to - from);
}
static bool parse_constraints(const std::vector<std::string>& exprs, std::vector<Expr>& out) {
    out.clear();
    for (const std::string& raw : exprs) {
        const std::u32string e = decode(raw);
        bool found = false;
        for (int op = 0; op < 2 && !found; ++op) {   // eq first, then noteq (constraint.go:51-68: first operator that splits)
            const char32_t c0 = op == 0 ? U'=' : U'!';
            size_t pos = std::u32string::npos;
            for (size_t i = 0; i + 1 < e.size(); ++i)
                if (e[i] == c0 && e[i + 1] == U'=') { pos = i; break; }
            if (pos == std::u32string::npos) continue;
            Expr x;
            x.key = trim(e, 0, pos);
            x.op = op;
            x.exp = trim(e, pos + 2, e.size());
            if (!key_ok(x.key) || !value_ok(x.exp)) return false;
            out.push_back(std::move(x));
            found = true;
        }
        if (!found) return false;
    }
    return true;
}

// net.ParseIP (constraint.go:128-146): dotted decimal without leading zeros, or RFC 4291 text; no zone.
static bool parse_ip(const std::string& s, uint8_t out[16], bool* is_v4) {
    if (s.empty() || s.find('%') != std::string::npos || std::strlen(s.c_str()) != s.size()) return false;
    unsigned char b4[4];
    if (s.find(':') == std::string::npos) {
        if (inet_pton(AF_INET, s.c_str(), b4) != 1) return false;
        std::memset(out, 0, 10);
        out[10] = out[11] = 0xFF;
        std::memcpy(out + 12, b4, 4);
        *is_v4 = true;
        return true;
    }
    if (inet_pton(AF_INET6, s.c_str(), out) != 1) return false;
    bool mapped = out[10] == 0xFF && out[11] == 0xFF;
    for (int i = 0; i < 10; ++i) mapped = mapped && out[i] == 0;
    *is_v4 = mapped;
    return true;
}

// Pipeline.Explain (pipeline.go:84-103): entries sorted by failure count, descending, stable for ≤ 12 elements
// (insertion sort inside sort.Sort), zero counts skipped; each filter's Explain(nodes) text (filter.go).
static std::string explain(const uint32_t* hist) {
    // (text, then whether a count goes in front of it / in its middle — "%u nodes …" and "… on %u nodes")
    struct Reason { const char* one; const char* head; const char* tail; };
    static const Reason reasons[SWP_NFILTERS] = {
        {"1 node not available for new tasks", "", " nodes not available for new tasks"},
        {"insufficient resources on 1 node", "insufficient resources on ", " nodes"},
        {"missing plugin on 1 node", "missing plugin on ", " nodes"},
        {"scheduling constraints not satisfied on 1 node", "scheduling constraints not satisfied on ", " nodes"},
        {"unsupported platform on 1 node", "unsupported platform on ", " nodes"},
        {"host-mode port already in use on 1 node", "host-mode port already in use on ", " nodes"},
        {"max replicas per node limit exceed", nullptr, nullptr},
        {"cannot fulfill requested CSI volume mounts on 1 node", "cannot fulfill requested CSI volume mounts on ", " nodes"}};
    // most failures first, pipeline order among equals (pipeline.go:86-98: sort.Sort over eight entries is Go's insertion sort, equals keep
    // their order): an insertion sort here too
    int order[SWP_NFILTERS];
    for (int i = 0; i < SWP_NFILTERS; ++i) {
        int k = i;
        while (k > 0 && hist[order[k - 1]] < hist[i]) { order[k] = order[k - 1]; --k; }
        order[k] = i;
    }
    std::string out;
    out.reserve(128);
    for (int k = 0; k < SWP_NFILTERS; ++k) {
        const int i = order[k];
        const uint32_t n = hist[i];
        if (n == 0) continue;
        if (!out.empty()) out += "; ";
        if (n == 1 || reasons[i].head == nullptr) out += reasons[i].one;
        else {
            char tmp[12];
            char* q = tmp + sizeof tmp;
            uint32_t u = n;
            do { *--q = (char)('0' + u % 10); u /= 10; } while (u);
            out += reasons[i].head;
            out.append(q, (size_t)(tmp + sizeof tmp - q));
            out += reasons[i].tail;
        }
    }
    return out;
}

// structural hash / equality of a (sub)document: what recognises two tasks as carrying the same spec
static uint64_t hash_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001B3ull + 0x9E3779B97F4A7C15ull; }
static uint64_t hash_bytes(uint64_t h, const std::string& s) {
    for (unsigned char c : s) h = (h ^ c) * 0x100000001B3ull;
    return hash_mix(h, s.size());
}
static uint64_t hash_value(uint64_t h, const Value* v) {
    if (v == nullptr) return hash_mix(h, 0xA5);
    h = hash_mix(h, (uint64_t)v->kind + 1);
    switch (v->kind) {
        case Value::Null: break;
        case Value::Bool: h = hash_mix(h, v->b ? 1 : 0); break;
        case Value::Int: h = hash_mix(h, (uint64_t)v->i); break;
        case Value::Real: { uint64_t bits; std::memcpy(&bits, &v->d, 8); h = hash_mix(h, bits); break; }
        case Value::Str: h = hash_bytes(h, v->s); break;
        case Value::Arr:
            for (const Value& x : *v->a) h = hash_value(h, &x);
            break;
        case Value::Obj:
            for (const json::Member& m : *v->o) h = hash_value(hash_bytes(h, m.first), &m.second);
            break;
    }
    return h;
}
static bool equal_value(const Value* a, const Value* b) {
    if (a == nullptr || b == nullptr) return a == b;
    if (a->kind != b->kind) return false;
    switch (a->kind) {
        case Value::Null: return true;
        case Value::Bool: return a->b == b->b;
        case Value::Int: return a->i == b->i;
        case Value::Real: return a->d == b->d;
        case Value::Str: return a->s == b->s;
        case Value::Arr:
            if (a->a == b->a) return true;
            if (a->a->size() != b->a->size()) return false;
            for (size_t i = 0; i < a->a->size(); ++i)
                if (!equal_value(&(*a->a)[i], &(*b->a)[i])) return false;
            return true;
        case Value::Obj:
            if (a->o == b->o) return true;
            if (a->o->size() != b->o->size()) return false;
            for (size_t i = 0; i < a->o->size(); ++i)
                if ((*a->o)[i].first != (*b->o)[i].first || !equal_value(&(*a->o)[i].second, &(*b->o)[i].second)) return false;
            return true;
    }
    return false;
}

using FailureKey = std::pair<std::string, int64_t>;   // versionedService: (ServiceID, SpecVersion.Index) — nodeinfo.go:18-26

// The decisions of a tick / of processPreassignedTasks as the JSON text the C boundary hands out — what applySchedulingDecisions
// (scheduler.go:490-643) would write, one object per task — written straight into the output buffer: a 100k-task tick is 15 MB of
// decisions, and a document tree per decision (seven strings, an object, a vector) was most of what the tick cost (round 4: 15-24 µs
// per decided task).
class Decisions {
  public:
    Decisions() { buf_.reserve(1 << 16); buf_.push_back('['); }
    void expect(size_t n) {   // a line of an assigned task is 150 bytes with ids of 10 characters: room for n of them at once
        buf_.reserve(n * 192 + 64);
        ids_.reserve(n);
    }
    // {"ID","ServiceID","NodeID","State","Message","Err","OldState"} from (decision.old, decision.new); err: instead of new's Status.Err
    void begin(const Value& old, const Value& neu, const std::string* err = nullptr) {
        const Value* st = neu.get("Status");
        begin(task_id(neu), as_str(neu.get("ServiceID")), as_str(neu.get("NodeID")), task_state(st ? st->get("State") : nullptr), as_str(st ? st->get("Message") : nullptr),
              err ? *err : as_str(st ? st->get("Err") : nullptr), task_state(at(&old, "Status", "State")));
    }
    void begin(const std::string& id, const std::string& service, const std::string& node, int64_t state, const std::string& message, const std::string& err, int64_t old_state) {
        if (n_++) buf_.push_back(',');
        ids_.push_back(id);
        // The usual line — ids, a fixed message, no character to escape — is written through a pointer into room made once: seven members
        // were fourteen appends with a capacity check each.
        if (plain(id) && plain(service) && plain(node) && plain(message) && plain(err)) {
            const size_t at = buf_.size();
            buf_.resize(at + 160 + id.size() + service.size() + node.size() + message.size() + err.size());
            char* p = &buf_[at];
            p = lit(p, "{\"ID\":\"");
            p = raw(p, id);
            p = lit(p, "\",\"ServiceID\":\"");
            p = raw(p, service);
            p = lit(p, "\",\"NodeID\":\"");
            p = raw(p, node);
            p = lit(p, "\",\"State\":");
            p = num(p, state);
            p = lit(p, ",\"Message\":\"");
            p = raw(p, message);
            p = lit(p, "\",\"Err\":\"");
            p = raw(p, err);
            p = lit(p, "\",\"OldState\":");
            p = num(p, old_state);
            buf_.resize((size_t)(p - buf_.data()));
            return;
        }
        buf_ += "{\"ID\":";
        json::dump_string(buf_, id);
        buf_ += ",\"ServiceID\":";
        json::dump_string(buf_, service);
        buf_ += ",\"NodeID\":";
        json::dump_string(buf_, node);
        buf_ += ",\"State\":";
        number(state);
        buf_ += ",\"Message\":";
        json::dump_string(buf_, message);
        buf_ += ",\"Err\":";
        json::dump_string(buf_, err);
        buf_ += ",\"OldState\":";
        number(old_state);
    }
    static bool plain(const std::string& s) {   // nothing json::dump_string would escape
        bool ok = true;
        for (unsigned char c : s) ok = ok && c >= 0x20 && c != '"' && c != '\\';
        return ok;
    }
    template <size_t N> static char* lit(char* p, const char (&text)[N]) { std::memcpy(p, text, N - 1); return p + (N - 1); }
    static char* raw(char* p, const std::string& s) { std::memcpy(p, s.data(), s.size()); return p + s.size(); }
    static char* num(char* p, int64_t v) {   // at most 20 characters
        char tmp[24];
        char* q = tmp + sizeof tmp;
        uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
        do { *--q = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *--q = '-';
        const size_t n = (size_t)(tmp + sizeof tmp - q);
        std::memcpy(p, q, n);
        return p + n;
    }
    void number(int64_t v) {
        char tmp[24];
        char* p = tmp + sizeof tmp;
        uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
        do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *--p = '-';
        buf_.append(p, (size_t)(tmp + sizeof tmp - p));
    }
    void field(const char* key, const Value& v) {
        buf_ += ",\"";
        buf_ += key;
        buf_ += "\":";
        json::dump(buf_, v);
    }
    void end() { buf_.push_back('}'); }
    const std::vector<std::string>& ids() const { return ids_; }   // the tasks that have a line
    std::string finish() {
        buf_.push_back(']');
        return std::move(buf_);
    }

  private:
    std::string buf_;
    std::vector<std::string> ids_;
    size_t n_ = 0;
};

// The non-numeric half of scheduler.NodeInfo (nodeinfo.go:28-44); the numeric half lives in the engine's node row.
struct NodeInfo {
    Value node;                                           // *api.Node
    uint32_t idx = 0;                                     // engine node index
    NodeTasks Tasks;                                      // NodeInfo.Tasks
    std::map<FailureKey, std::vector<int64_t>> recentFailures;
    int64_t lastCleanup = 0;
    generic::List availGeneric;                           // AvailableResources.Generic (the engine holds one count per kind of it)
};

// SWP_SCHED_PROF=1: where a tick's host time goes (a line per tick on stderr)
struct TickProf {
    bool on = std::getenv("SWP_SCHED_PROF") != nullptr;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        ms[k] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
    }
    void report(size_t n) {
        if (!on) return;
        std::fprintf(stderr, "[swp_sched] tick of %zu tasks: queue %.1f ms, failure counts %.1f, grouping %.1f, descriptors %.1f, engine %.1f, decisions + bookkeeping %.1f, output %.1f\n", n, ms[0], ms[1],
                     ms[2], ms[3], ms[4], ms[5], ms[6]);
        for (double& m : ms) m = 0;
    }
};

class Scheduler {
  public:
    explicit Scheduler(swp_engine* e) : e_(e) { ck(swp_reset(e_, 0), "swp_reset"); }
    TickProf prof_;

    std::string scratch;   // result storage handed out through const char**
    std::string last_error;

    // ---------------------------------------------------------------------------------------------- nodes
    // createOrUpdateNode, scheduler.go:368-396
    void createOrUpdateNode(const Value& n) {
        const std::string& nid = task_id(n);
        auto it = nodes_.find(nid);
        NodeInfo* ni = it == nodes_.end() ? nullptr : &it->second;
        const Value* res = at(&n, "Description", "Resources");
        int64_t cpu = 0, mem = 0;
        generic::List avail;
        if (res != nullptr) {
            cpu = as_i64(res->get("NanoCPUs"));
            mem = as_i64(res->get("MemoryBytes"));
            avail = generic::decode(res->get("Generic"));
            if (ni != nullptr) {   // :376-384: subtract the reservations of the tasks already on the node, take their generic resources out
                ni->Tasks.each_sorted([&](const std::string&, const Value& t) {
                    int64_t c, m;
                    taskReservations(t, c, m);
                    cpu = wrap_sub(cpu, c);   // (Go's int64 wraps; documents at the boundary may say anything)
                    mem = wrap_sub(mem, m);
                    generic::consume(&avail, generic::decode(t.get("AssignedGenericResources")));
                });
            }
        }
        const uint32_t idx = intern(SWP_SPACE_NODE_ID, nid);
        uint32_t total = 0;
        if (ni != nullptr) {
            swp_node_row cur;
            int rc = swp_node_get(e_, idx, &cur);
            if (rc == SWP_OK) total = cur.total;
            else if (rc != SWP_ENOTFOUND) ck(rc, "swp_node_get");
        } else {
            NodeInfo fresh;
            fresh.idx = idx;
            fresh.lastCleanup = now_;
            ni = &nodes_.emplace(nid, std::move(fresh)).first->second;
            if (idx_to_id_.size() <= idx) idx_to_id_.resize(idx + 1);
            idx_to_id_[idx] = nid;
            if (node_by_idx_.size() <= idx) node_by_idx_.resize(idx + 1, nullptr);
            node_by_idx_[idx] = ni;   // (std::map nodes never move)
            repin_after_ = nid;   // volumes in use on a node of this id learn its index below
        }
        ni->node = n;
        ni->availGeneric = std::move(avail);
        upsertRow(n, idx, cpu, mem, total);
        pushGeneric(*ni);
        if (!repin_after_.empty()) {
            repinVolumes(repin_after_);
            repin_after_.clear();
        }
    }
    std::string repin_after_;
    void repinVolumes(const std::string& nid) {
        for (const auto& kv : volumes_)
            for (const auto& u : kv.second.tasks)
                if (u.second.node == nid) { pushVolumeUsage(kv.second); break; }
    }
    // nodeSet.remove, nodeset.go:46-48
    void deleteNode(const std::string& nid) {
        auto it = nodes_.find(nid);
        if (it == nodes_.end()) return;
        const uint32_t idx = it->second.idx;
        nodes_.erase(it);
        irregularGeneric_.erase(nid);
        repinVolumes(nid);   // (before the index is free again: no usage number may name it)
        ck(swp_node_remove(e_, idx), "swp_node_remove");
        // the engine hands the index to the next node that is new to it: nothing here may remember it as this node's
        if (idx < idx_to_id_.size()) idx_to_id_[idx].clear();
        if (idx < node_by_idx_.size()) node_by_idx_[idx] = nullptr;
        for (auto pf = pushedFailures_.begin(); pf != pushedFailures_.end();) pf = std::get<0>(pf->first) == idx ? pushedFailures_.erase(pf) : std::next(pf);
    }
    // nodeSet.nodeInfo, nodeset.go:23-29
    bool nodeInfo(const std::string& nid, std::string& out) {
        auto it = nodes_.find(nid);
        if (it == nodes_.end()) return false;
        NodeInfo& ni = it->second;
        swp_node_row row;
        ck(swp_node_get(e_, ni.idx, &row), "swp_node_get");
        Value by_service = Value::object(), tasks = Value::array(), fails = Value::object();
        ni.Tasks.each_sorted([&](const std::string& id, const Value& t) {
            const std::string& sid = as_str(t.get("ServiceID"));
            uint32_t c = 0;
            ck(swp_node_get_svc_count(e_, ni.idx, intern(SWP_SPACE_SERVICE, sid), &c), "swp_node_get_svc_count");
            if (c) by_service.set(sid, Value::integer(c));
            tasks.push(Value::str(id));
        });
        for (const auto& kv : ni.recentFailures) fails.set(kv.first.first + "@" + std::to_string(kv.first.second), Value::integer((int64_t)kv.second.size()));
        Value avail = Value::object();
        avail.set("NanoCPUs", Value::integer(row.cpu));
        avail.set("MemoryBytes", Value::integer(row.mem));
        avail.set("Generic", generic::encode(ni.availGeneric));
        Value info = Value::object();
        info.set("ID", Value::str(nid));
        info.set("ActiveTasksCount", Value::integer(row.total));
        info.set("ActiveTasksCountByService", by_service);
        info.set("AvailableResources", avail);
        info.set("Tasks", tasks);
        info.set("RecentFailures", fails);
        out = json::dump(info);
        return true;
    }

    // ---------------------------------------------------------------------------------------------- CSI volumes
    // EventUpdateVolume (scheduler.go:200-213) and the volumes of the store at start (:70-81): addOrUpdateVolume (volumes.go:62-82) once
    // the plugin has created the volume. The engine gets what checkVolume reads; the maps behind its usage numbers stay here.
    void updateVolume(const Value& v) {
        const std::string& plugin_id = as_str(at(&v, "VolumeInfo", "VolumeID"));
        if (v.get("VolumeInfo") == nullptr || plugin_id.empty()) return;
        const std::string& vid = task_id(v);
        auto it = volumes_.find(vid);
        if (it == volumes_.end()) {
            VolumeRec rec;
            rec.idx = intern(SWP_SPACE_VOLUME, vid);
            it = volumes_.emplace(vid, std::move(rec)).first;
            if (vol_idx_to_id_.size() <= it->second.idx) vol_idx_to_id_.resize(it->second.idx + 1);
            vol_idx_to_id_[it->second.idx] = vid;
        }
        it->second.doc = v;
        swp_volume sv;
        std::memset(&sv, 0, sizeof sv);
        sv.group = intern(SWP_SPACE_VOLUME_GROUP, as_str(at(&v, "Spec", "Group")));
        sv.driver = intern(SWP_SPACE_CSI, as_str(at(&v, "Spec", "Driver", "Name")));
        sv.scope = (uint32_t)enum_value(at(&v, "Spec", "AccessMode", "Scope"), {{"SINGLE_NODE", 0}, {"MULTI_NODE", 1}});
        sv.sharing = (uint32_t)enum_value(at(&v, "Spec", "AccessMode", "Sharing"), {{"NONE", 0}, {"READ_ONLY", 1}, {"ONE_WRITER", 2}, {"ALL", 3}});
        sv.active = enum_value(at(&v, "Spec", "Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}}) == 0 ? 1u : 0u;
        std::vector<uint32_t> topo_off(1, 0);
        std::vector<swp_seg> segs;
        if (const Value* at_ = at(&v, "VolumeInfo", "AccessibleTopology"))
            if (at_->is_arr())
                for (const Value& top : *at_->a) {
                    if (const Value* sg = top.get("Segments"))
                        if (sg->is_obj())
                            for (const json::Member& m : *sg->o) segs.push_back({intern(SWP_SPACE_CSI, m.first), intern(SWP_SPACE_CSI, as_str(&m.second))});
                    topo_off.push_back((uint32_t)segs.size());
                }
        sv.n_topologies = (uint32_t)topo_off.size() - 1;
        static const swp_seg no_seg = {0, 0};
        ck(swp_volume_upsert(e_, it->second.idx, &sv, topo_off.data(), segs.empty() ? &no_seg : segs.data()), "swp_volume_upsert");
        volByName_[as_str(at(&v, "Spec", "Annotations", "Name"))] = vid;
        ++tmplGen_;   // (a mount's Source may resolve differently from now on)
    }
    // volumeSet.reserveVolume / releaseVolume (volumes.go:156-187) + the usage numbers the engine judges by
    void reserveVolume(const std::string& vid, const std::string& tid, const std::string& nid, bool ro) {
        auto it = volumes_.find(vid);
        if (it == volumes_.end()) return;
        it->second.tasks[tid] = VolumeUse{nid, ro};
        it->second.nodes[nid] += 1;   // (per CALL, volumes.go:159: a task with two mounts on one volume counts its node twice)
        pushVolumeUsage(it->second);
    }
    void releaseVolume(const std::string& vid, const std::string& tid) {
        auto it = volumes_.find(vid);
        if (it == volumes_.end()) return;
        auto ut = it->second.tasks.find(tid);
        if (ut == it->second.tasks.end()) return;
        int64_t& c = it->second.nodes[ut->second.node];
        if (c > 0) c -= 1;   // volumes.go:171-176
        it->second.tasks.erase(ut);
        pushVolumeUsage(it->second);
    }
    // freeVolumes (volumes.go:181-221), what tick defers (scheduler.go:501): the PUBLISHED statuses of a volume on nodes whose reference
    // count is zero become PENDING_NODE_UNPUBLISH. The store is the caller's: this returns the updates to write — [{"VolumeID", "NodeIDs"}]
    // by volume id — and moves the statuses of the volume documents kept here (they stand for the store's copy, :189; the store's
    // EventUpdateVolume brings the same back), so a second call reports nothing new.
    void freeVolumes(std::string& out) {
        Value updates = Value::array();
        for (auto& kv : volumes_) {
            const Value* ps = kv.second.doc.get("PublishStatus");   // (this document is the private copy updateVolume parsed)
            if (ps == nullptr || !ps->is_arr()) continue;
            Value changed = Value::array();
            for (Value& st : *ps->a) {
                if (!st.is_obj()) continue;
                const std::string& nid = as_str(st.get("NodeID"));
                auto n = kv.second.nodes.find(nid);
                if ((n == kv.second.nodes.end() || n->second == 0) && enum_value(st.get("State"), {{"PENDING_PUBLISH", 0}, {"PUBLISHED", 1}, {"PENDING_NODE_UNPUBLISH", 2}, {"PENDING_UNPUBLISH", 3}}) == 1) {
                    st.set("State", Value::str("PENDING_NODE_UNPUBLISH"));
                    changed.push(Value::str(nid));
                }
            }
            if (changed.a->empty()) continue;
            Value u = Value::object();
            u.set("VolumeID", Value::str(kv.first));
            u.set("NodeIDs", changed);
            updates.push(u);
        }
        out = json::dump(updates);
    }
    // What chooseTaskVolumes leaves behind in the counts (volumes.go:104-131): it reserves every volume it picks for the task — a count on
    // the node per CALL (:159) — and releases them all on return, but a release finds the task only ONCE per volume (:169-178), so a
    // volume that serves m of ONE task's mounts keeps m − 1 counts on the node for ever, whether the choice succeeds or stops at a later
    // mount. The engine chooses on the device and reserves nothing on the way; the remainder is booked here from what it chose (`att`:
    // the first n entries, the prefix in front of a failing mount included) so that freeVolumes frees exactly what the reference frees.
    void bookChooseRemainder(const uint32_t* att, size_t n, const std::string& nid) {
        for (size_t q = 0; q < n; ++q) {
            size_t m = 0;
            bool first = true;
            for (size_t z = 0; z < n; ++z)
                if (att[z] == att[q]) {
                    if (z < q) first = false;
                    ++m;
                }
            if (!first || m < 2 || att[q] >= vol_idx_to_id_.size()) continue;
            auto it = volumes_.find(vol_idx_to_id_[att[q]]);
            if (it != volumes_.end()) it->second.nodes[nid] += (int64_t)(m - 1);
        }
    }
    // reserveTaskVolumes, volumes.go:144-154
    void reserveTaskVolumes(const Value& t) {
        const Value* vols = t.get("Volumes");
        const Value* mounts = at(&t, "Spec", "Container", "Mounts");
        if (vols == nullptr || !vols->is_arr() || mounts == nullptr || !mounts->is_arr()) return;
        for (const Value& va : *vols->a)
            for (const Value& m : *mounts->a)
                if (as_str(m.get("Source")) == as_str(va.get("Source")) && as_str(m.get("Target")) == as_str(va.get("Target")))
                    reserveVolume(as_str(va.get("ID")), task_id(t), as_str(t.get("NodeID")), truthy(m.get("ReadOnly")));
    }
    void releaseTaskVolumes(const Value& t) {
        const Value* vols = t.get("Volumes");
        if (vols != nullptr && vols->is_arr())
            for (const Value& va : *vols->a) releaseVolume(as_str(va.get("ID")), task_id(t));
    }
    bool volumeInfo(const std::string& vid, std::string& out) {
        auto it = volumes_.find(vid);
        if (it == volumes_.end()) return false;
        Value tasks = Value::object(), nodes = Value::object();
        for (const auto& kv : it->second.tasks) {
            Value u = Value::object();
            u.set("NodeID", Value::str(kv.second.node));
            u.set("ReadOnly", Value::boolean(kv.second.read_only));
            tasks.set(kv.first, u);
        }
        for (const auto& kv : it->second.nodes) nodes.set(kv.first, Value::integer(kv.second));   // (volumeInfo.nodes as the reference keeps it: zeros stay)
        swp_volume_usage use;
        ck(swp_volume_get_usage(e_, it->second.idx, &use), "swp_volume_get_usage");
        Value eng = Value::object();
        eng.set("Tasks", Value::integer(use.n_tasks));
        eng.set("Writers", Value::integer(use.n_writers));
        Value info = Value::object();
        info.set("Tasks", tasks);
        info.set("Nodes", nodes);
        info.set("Engine", eng);
        out = json::dump(info);
        return true;
    }

    // ---------------------------------------------------------------------------------------------- services / clock
    void setService(const std::string& sid, bool has_version, uint64_t version) {
        services_[sid] = has_version ? std::optional<uint64_t>(version) : std::nullopt;
    }
    void deleteService(const std::string& sid) { services_.erase(sid); }
    void advance(int64_t ns) { now_ += ns; }

    // ---------------------------------------------------------------------------------------------- task events
    // The task's Reservations.Generic as the engine takes them (swp_generic_set): Discrete entries, one per kind, values >= 1, at
    // most 8 kinds. Anything else — a Named reservation (HasEnough's error return, validate.go:26-29: such a task fits nowhere in
    // the reference), a request of 0 (selectNodeResources would claim every named value, resource_management.go:52-66), a kind
    // listed twice — stays on the Go path.
    static generic::List genericReservations(const Value& t) {
        generic::List r = generic::decode(at(&t, "Spec", "Resources", "Reservations", "Generic"));
        if (r.size() > 8) unsupported("more than 8 generic reservations in one task stay on the Go path");
        std::set<std::string> kinds;
        for (const generic::Res& g : r) {
            if (g.named) unsupported("a Named generic reservation stays on the Go path");
            if (g.ival < 1) unsupported("a generic reservation below 1 stays on the Go path");
            if (!kinds.insert(g.kind).second) unsupported("a generic kind reserved twice stays on the Go path");
        }
        return r;
    }
    // Tasks the engine cannot judge (generic resources, CSI cluster volumes) are refused at the event boundary — the shim
    // leaves them to the Go scheduler's own path — so that a tick never meets one half-way through a batch.
    static void requireSupported(const Value& t) {
        (void)genericReservations(t);   // refuses what swp_generic_set would refuse
        if (const Value* ports = at(&t, "Endpoint", "Ports"))
            if (ports->is_arr()) {
                size_t host_ports = 0;
                for (const Value& p : *ports->a)
                    if (enum_value(p.get("PublishMode"), {{"INGRESS", 0}, {"HOST", 1}}) == PUBLISH_HOST && as_i64(p.get("PublishedPort")) != 0) ++host_ports;
                if (host_ports > 32) unsupported("more than 32 host-mode ports in one task stay on the Go path");   // swp_port_set's limit
            }
        if (clusterMounts(t).size() > SWP_MAX_MOUNTS) unsupported("more than 8 cluster mounts in one task stay on the Go path");   // swp_mount_set's limit
    }
    // the task's MountTypeCluster mounts in spec order (VolumesFilter.SetTask, filter.go:412-420)
    static bool isCluster(const Value& m) { return enum_value(m.get("Type"), {{"BIND", 0}, {"VOLUME", 1}, {"TMPFS", 2}, {"NPIPE", 3}, {"CLUSTER", 4}}, -1) == MOUNT_CLUSTER; }
    static std::vector<const Value*> clusterMounts(const Value& t) {
        std::vector<const Value*> out;
        const Value* mounts = at(&t, "Spec", "Container", "Mounts");
        if (mounts != nullptr && mounts->is_arr())
            for (const Value& m : *mounts->a)
                if (isCluster(m)) out.push_back(&m);
        return out;
    }
    // createTask, scheduler.go:254-283
    bool createTask(const Value& t) {
        const int64_t st = task_state(at(&t, "Status", "State"));
        if (st < PENDING || st > RUNNING) return false;
        requireSupported(t);
        const std::string& id = task_id(t);
        allTasks_[id] = t;
        if (!truthy(t.get("NodeID"))) {
            unassignedTasks_.put(id, t, templateOf(t));   // enqueue, :250-252
            return true;
        }
        if (st == PENDING) {
            preassignedTasks_.insert(id);
            pendingPreassignedTasks_.put(id, t);
            return false;   // preassigned tasks are processed by processPreassignedTasks, not tick
        }
        auto it = nodes_.find(as_str(t.get("NodeID")));
        if (it != nodes_.end()) addTask(it->second, t);
        return false;
    }
    // setupTasksList, scheduler.go:68-126: a task found in the store at start-up. One rule differs from the createTask
    // event: a task still PENDING whose desired state is already past COMPLETED is ignored (:93-101).
    bool setupTask(const Value& t) {
        const int64_t st = task_state(at(&t, "Status", "State"));
        if (st == PENDING && task_state(t.get("DesiredState")) > COMPLETE) return false;
        const bool r = createTask(t);
        // :115-116: the volumes in use by a task that sits on its node already (NOT what the createTask event does)
        if (st > PENDING && st <= RUNNING && truthy(t.get("NodeID"))) reserveTaskVolumes(t);
        return r;
    }
    // updateTask, scheduler.go:283-348
    bool updateTask(const Value& t) {
        const int64_t st = task_state(at(&t, "Status", "State"));
        if (st < PENDING) return false;
        const std::string& id = task_id(t);
        auto old_it = allTasks_.find(id);
        const bool has_old = old_it != allTasks_.end();
        if (st > RUNNING) {
            if (!has_old) return false;
            const Value old = old_it->second;
            if (st != task_state(at(&old, "Status", "State")) && (st == FAILED || st == REJECTED)) {
                if (preassignedTasks_.count(id) == 0) {   // preassigned tasks do not count against the node, :303-310
                    auto n = nodes_.find(as_str(t.get("NodeID")));
                    if (n != nodes_.end()) taskFailed(n->second, t);
                }
            }
            deleteTask(old);
            return true;
        }
        requireSupported(t);
        if (!truthy(t.get("NodeID"))) {
            if (has_old) {
                const Value old = old_it->second;
                deleteTask(old);
            }
            allTasks_[id] = t;
            unassignedTasks_.put(id, t);
            return true;
        }
        if (st == PENDING) {
            if (has_old) {
                const Value old = old_it->second;
                deleteTask(old);
            }
            preassignedTasks_.insert(id);
            allTasks_[id] = t;
            pendingPreassignedTasks_.put(id, t);
            return false;
        }
        allTasks_[id] = t;
        auto n = nodes_.find(as_str(t.get("NodeID")));
        if (n != nodes_.end()) addTask(n->second, t);
        return false;
    }
    // deleteTask, scheduler.go:350-366
    bool deleteTask(const Value& t) {
        const std::string id = task_id(t);
        allTasks_.erase(id);
        if (unassignedTasks_.find(id)) deletedWhileQueued_.push_back(id);   // (the queue keeps it, as the reference's does: tick's catch-all path below)
        preassignedTasks_.erase(id);
        pendingPreassignedTasks_.erase(id);
        releaseTaskVolumes(t);   // :355-358
        auto n = nodes_.find(as_str(t.get("NodeID")));
        if (n != nodes_.end() && removeTask(n->second, t)) return true;
        return false;
    }

    // ---------------------------------------------------------------------------------------------- task templates
    // Everything Pipeline.SetTask reads of a task (taskDesc below) is shared by the tasks of one service revision: ServiceID, whether the
    // task still counts (DesiredState), SpecVersion, Spec, Networks, Endpoint. A task is matched with its template when its EVENT comes in
    // — the document is hot then: one structural hash and one comparison with the template's exemplar — and a tick takes descriptors,
    // services and groups from the templates without touching 100 000 documents again (round 4: 3 µs of the 15 µs per task).
    struct Template {
        Value exemplar;           // a task of the template (its documents are shared sub-trees: nothing is copied)
        std::string service;
        bool has_version = false;
        int64_t version = 0;
        swp_task_desc desc{};
        uint64_t gen = ~0ull;     // tmplGen_ the descriptor was computed at (a mount's Source resolves through byName NOW, volumes.go:252)
        bool wants_generic = false;   // Spec.Resources.Reservations.Generic is there: place() decides WHICH resources the task holds
        bool has_mounts = false;      // Spec.Container.Mounts is there: place() looks for the cluster mounts' attachments
    };
    std::vector<Template> templates_;
    std::unordered_map<uint64_t, std::vector<uint32_t>> tmplIndex_;
    uint64_t tmplGen_ = 0;
    static uint64_t templateHash(const Value& t) {
        const Value *spec = t.get("Spec"), *nets = t.get("Networks"), *endp = t.get("Endpoint"), *sv = t.get("SpecVersion");
        uint64_t h = hash_bytes(0xCBF29CE484222325ull, as_str(t.get("ServiceID")));
        h = hash_mix(h, task_state(t.get("DesiredState")) > COMPLETE ? 1 : 0);
        return hash_value(hash_value(hash_value(hash_value(h, sv), spec), nets), endp);
    }
    uint32_t templateOf(const Value& t) {
        const Value *spec = t.get("Spec"), *nets = t.get("Networks"), *endp = t.get("Endpoint"), *sv = t.get("SpecVersion");
        const std::string& sid = as_str(t.get("ServiceID"));
        const bool uncounted = task_state(t.get("DesiredState")) > COMPLETE;
        std::vector<uint32_t>& cands = tmplIndex_[templateHash(t)];
        for (uint32_t id : cands) {
            const Value& x = templates_[id].exemplar;
            if (as_str(x.get("ServiceID")) == sid && (task_state(x.get("DesiredState")) > COMPLETE) == uncounted && equal_value(x.get("SpecVersion"), sv) &&
                equal_value(x.get("Spec"), spec) && equal_value(x.get("Networks"), nets) && equal_value(x.get("Endpoint"), endp))
                return id;
        }
        Template tm;
        tm.exemplar = t;
        tm.service = sid;
        tm.has_version = sv != nullptr;
        tm.version = as_i64(sv != nullptr ? sv->get("Index") : nullptr);
        tm.wants_generic = at(spec, "Resources", "Reservations", "Generic") != nullptr;
        tm.has_mounts = at(spec, "Container", "Mounts") != nullptr;
        templates_.push_back(std::move(tm));
        cands.push_back((uint32_t)templates_.size() - 1);
        return (uint32_t)templates_.size() - 1;
    }
    // Templates are only ever referred to by QUEUED tasks (QItem::tmpl), and a tick holds the whole queue in its hands: when most of
    // the templates belong to service revisions nothing in the queue is of any more, the ones in use move to the front and the rest
    // — their exemplar documents, their descriptors — go (a scheduler that lives for months sees revisions come and never sees them
    // leave otherwise). A revision that comes back is recognised afresh from its next event.
    static constexpr size_t TEMPLATES_KEPT = 1024;   // below this many nothing is swept
    void sweepTemplates(std::vector<QItem>& queue, size_t in_use) {
        if (templates_.size() <= TEMPLATES_KEPT || templates_.size() <= 4 * in_use) return;
        std::vector<uint32_t> remap(templates_.size(), NO_TMPL);
        std::vector<Template> kept;
        kept.reserve(in_use);
        for (QItem& it : queue) {
            if (remap[it.tmpl] == NO_TMPL) {
                remap[it.tmpl] = (uint32_t)kept.size();
                kept.push_back(std::move(templates_[it.tmpl]));
            }
            it.tmpl = remap[it.tmpl];
        }
        templates_.swap(kept);
        tmplIndex_.clear();
        for (size_t i = 0; i < templates_.size(); ++i) tmplIndex_[templateHash(templates_[i].exemplar)].push_back((uint32_t)i);
    }
    // the template's descriptor (Pipeline.SetTask once per template and volume generation); throws what taskDesc throws
    // INVARIANT: a cached descriptor carries engine-side set ids (constraint, platform, plugin, port, generic, spread, mount sets) and
    // interned ids; they stay valid because this scheduler's engine is reset ONCE, in the constructor, and sets are never collected.
    // Whatever invalidates them later — an swp_reset of the engine, a shard set rebuilt under the same scheduler — must bump tmplGen_
    // (as updateVolume does for mounts), or every tick afterwards schedules with stale ids and nothing says so.
    const swp_task_desc& descOf(uint32_t tmpl) {
        Template& tm = templates_[tmpl];
        if (tm.gen != tmplGen_) {
            tm.desc = taskDesc(tm.exemplar);
            tm.gen = tmplGen_;
        }
        return tm.desc;
    }

    // ---------------------------------------------------------------------------------------------- Pipeline.SetTask
    // One swp_task_desc from an api.Task: every Filter.SetTask of the pipeline (pipeline.go:76-81, filter.go).
    swp_task_desc taskDesc(const Value& t) {
        swp_task_desc d;
        std::memset(&d, 0, sizeof d);
        d.service = intern(SWP_SPACE_SERVICE, as_str(t.get("ServiceID")));
        if (const Value* res = at(&t, "Spec", "Resources", "Reservations")) {   // ResourceFilter.SetTask, filter.go:61-74
            d.cpu = as_i64(res->get("NanoCPUs"));
            d.mem = as_i64(res->get("MemoryBytes"));
            const bool any_generic = res->get("Generic") != nullptr && res->get("Generic")->is_arr() && res->get("Generic")->size() > 0;
            if (d.cpu != 0 || d.mem != 0 || any_generic) d.flags |= SWP_TASK_RES_ENABLED;
            std::vector<swp_generic> items;
            for (const generic::Res& g : genericReservations(t)) items.push_back({intern(SWP_SPACE_GENERIC_KIND, g.kind), 0u, g.ival});
            if (!items.empty()) ck(swp_generic_set(e_, items.data(), (uint32_t)items.size(), &d.generic_set), "swp_generic_set");
        }
        if (task_state(t.get("DesiredState")) > COMPLETE) d.flags |= 0x2u;   // SWP_TASK_UNCOUNTED: nodeinfo.go:148
        if (const Value* pl = at(&t, "Spec", "Placement")) {
            d.constraint_set = constraintSet(pl->get("Constraints"));          // ConstraintFilter.SetTask, filter.go:218-232
            if (const Value* plats = pl->get("Platforms")) {                    // PlatformFilter.SetTask, filter.go:253-263
                if (plats->is_arr() && plats->size() > 0) {
                    std::vector<swp_platform> ps;
                    for (const Value& p : *plats->a) ps.push_back({intern(SWP_SPACE_OS, as_str(p.get("OS"))), intern(SWP_SPACE_ARCH, as_str(p.get("Architecture")))});
                    ck(swp_platform_set(e_, ps.data(), (uint32_t)ps.size(), &d.platform_set), "swp_platform_set");
                }
            }
            d.max_replicas = (uint64_t)as_i64(pl->get("MaxReplicas"));          // MaxReplicasFilter.SetTask, filter.go:363-370
            if (const Value* prefs = pl->get("Preferences")) {                  // nodeset.go:59-82: only label spreads make a level
                std::vector<swp_spread> levels;
                if (prefs->is_arr())
                    for (const Value& pref : *prefs->a) {
                        const Value* sp = pref.get("Spread");
                        if (sp == nullptr) continue;
                        const std::u32string sd = decode(as_str(sp->get("SpreadDescriptor")));
                        uint32_t kind, key;
                        if (labelKey(sd, kind, key)) levels.push_back({kind, key});
                    }
                if (!levels.empty()) ck(swp_spread_set(e_, levels.data(), (uint32_t)levels.size(), &d.spread_set), "swp_spread_set");
            }
        }
        // PluginFilter.SetTask, filter.go:119-131 (+ Check :133-177 decides what counts as a requirement)
        std::vector<std::string> vol;
        if (const Value* mounts = at(&t, "Spec", "Container", "Mounts")) {
            if (mounts->is_arr()) {
                {   // VolumesFilter.SetTask, filter.go:392-422: the cluster mounts as a mount set (a name resolves through byName NOW, volumes.go:252)
                    std::vector<swp_mount> ms;
                    for (const Value* m : clusterMounts(t)) {
                        swp_mount mm;
                        std::memset(&mm, 0, sizeof mm);
                        const std::string& src = as_str(m->get("Source"));
                        static const std::string prefix = "group:";
                        if (src.compare(0, prefix.size(), prefix) == 0) {
                            mm.is_group = 1;
                            mm.ref = intern(SWP_SPACE_VOLUME_GROUP, src.substr(prefix.size()));
                        } else {
                            auto bn = volByName_.find(src);
                            mm.ref = bn == volByName_.end() ? SWP_NO_VOLUME : volumes_.at(bn->second).idx;
                        }
                        mm.read_only = truthy(m->get("ReadOnly")) ? 1u : 0u;
                        mm.reserve_read_only = mm.read_only;   // what reserveTaskVolumes records: the LAST mount with this (Source, Target) speaks (volumes.go:148-151)
                        for (const Value& other : *mounts->a)
                            if (as_str(other.get("Source")) == src && as_str(other.get("Target")) == as_str(m->get("Target"))) mm.reserve_read_only = truthy(other.get("ReadOnly")) ? 1u : 0u;
                        ms.push_back(mm);
                    }
                    if (ms.size() > SWP_MAX_MOUNTS) unsupported("more than 8 cluster mounts in one task stay on the Go path");
                    if (!ms.empty()) {
                        uint32_t set = 0;
                        ck(swp_mount_set(e_, ms.data(), (uint32_t)ms.size(), &set), "swp_mount_set");
                        d.flags |= SWP_TASK_MOUNTS(set);
                    }
                }
                for (const Value& m : *mounts->a) {
                    if (enum_value(m.get("Type"), {{"BIND", 0}, {"VOLUME", 1}, {"TMPFS", 2}, {"NPIPE", 3}, {"CLUSTER", 4}}, -1) != MOUNT_VOLUME) continue;
                    const Value* dc = at(&m, "VolumeOptions", "DriverConfig");
                    if (dc == nullptr) continue;
                    const Value* name = dc->get("Name");
                    if (name == nullptr || !name->is_str() || name->s.empty() || name->s == "local") continue;
                    vol.push_back(name->s);
                }
            }
        }
        const Value* nets = t.get("Networks");
        const bool has_nets = truthy(nets) && nets->is_arr();
        const Value* logd = at(&t, "Spec", "LogDriver");
        if (has_nets || logd != nullptr || !vol.empty()) {
            std::vector<uint32_t> req;
            for (const std::string& v : vol) req.push_back(intern(SWP_SPACE_PLUGIN, std::string("Volume") + '\0' + v));
            if (has_nets)
                for (const Value& na : *nets->a) {
                    const std::string& name = as_str(at(&na, "Network", "DriverState", "Name"));
                    if (!name.empty()) req.push_back(intern(SWP_SPACE_PLUGIN, std::string("Network") + '\0' + name));
                }
            uint32_t log = 0;
            if (logd != nullptr) {
                const Value* name = logd->get("Name");
                if (name != nullptr && name->is_str() && !name->s.empty() && name->s != "none") log = intern(SWP_SPACE_PLUGIN, std::string("Log") + '\0' + name->s);
            }
            if (!req.empty() || log != 0) ck(swp_plugin_set(e_, req.empty() ? &log : req.data(), (uint32_t)req.size(), log, &d.plugin_set), "swp_plugin_set");
        }
        d.port_set = portSet(t);                                                 // HostPortFilter.SetTask, filter.go:322-333
        d.spec_version = (uint64_t)as_i64(at(&t, "SpecVersion", "Index"));
        return d;
    }
    // Placement.Constraints → predicate set id; 0 = filter disabled (empty list, or constraint.Parse failed: filter.go:223-229)
    uint32_t constraintSet(const Value* cons) {
        if (cons == nullptr || !cons->is_arr() || cons->size() == 0) return 0;
        std::vector<std::string> exprs;
        for (const Value& c : *cons->a) exprs.push_back(as_str(&c));
        std::vector<Expr> parsed;
        if (!parse_constraints(exprs, parsed)) return 0;
        std::vector<swp_constraint> cs;
        for (const Expr& x : parsed) cs.push_back(constraintStruct(x));
        uint32_t id = 0;
        ck(swp_constraint_set(e_, cs.data(), (uint32_t)cs.size(), &id), "swp_constraint_set");
        return id;
    }

    // ---------------------------------------------------------------------------------------------- preassigned
    // processPreassignedTasks + taskFitNode, scheduler.go:398-426, 646-690
    std::string processPreassignedTasks() {
        Decisions decisions;
        lastDecisions_.erase_if([](const PendingDecision& d) { return d.preassigned; });
        for (auto& kv : pendingPreassignedTasks_.snapshot()) {
            const std::string& tid = kv.first;
            const Value& t = kv.second;
            auto n = nodes_.find(as_str(t.get("NodeID")));
            if (n == nodes_.end()) continue;   // node not (yet) known: the task stays pending, :651-656
            Value newT = t.shallow_copy();
            int32_t ff = -1;
            try {
                const swp_task_desc d = taskDesc(t);
                ck(swp_check_node(e_, &d, n->second.idx, &ff), "swp_check_node");
            } catch (const Fail& f) {   // the engine cannot judge this task: it stays pending, the loop carries on
                last_error = f.msg;
                const bool from_engine = !engine_detail_.empty() && f.msg.size() >= engine_detail_.size() &&
                                         f.msg.compare(f.msg.size() - engine_detail_.size(), engine_detail_.size(), engine_detail_) == 0;
                const std::string why = "swp: deferred to the host scheduler: " + (from_engine ? engine_detail_ : f.msg);
                decisions.begin(t, t, &why);
                decisions.field("Deferred", Value::boolean(true));
                decisions.end();
                continue;
            }
            if (ff >= 0) {
                uint32_t hist[SWP_NFILTERS] = {0};
                hist[ff] = 1;
                Value status = statusCopy(t);
                status.set("Err", Value::str(explain(hist)));   // newT.Status.Err = s.pipeline.Explain(), :660
                newT.set("Status", status);
                allTasks_[tid] = newT;
            } else if (!chooseForPreassigned(t, n->second, newT)) {   // scheduler.go:663-674: the error string is the task's new status
                allTasks_[tid] = newT;
            } else {
                Value status = Value::object();
                status.set("State", Value::integer(ASSIGNED));
                status.set("Message", Value::str("scheduler confirmed task can run on preassigned node"));
                newT.set("Status", status);
                allTasks_[tid] = newT;
                addTask(n->second, newT);
                pendingPreassignedTasks_.erase(tid);
                // taskFitNode hands addTask the task the decision carries (scheduler.go:676-688): what Claim assigned is part of decision.new
                auto st = allTasks_.find(tid);
                if (st != allTasks_.end() && st->second.get("AssignedGenericResources")) newT.set("AssignedGenericResources", *st->second.get("AssignedGenericResources"));
            }
            lastDecisions_[tid] = PendingDecision{t, true};
            decisions.begin(t, newT);
            const Value* ag = newT.get("AssignedGenericResources");
            if (ag && ag->is_arr() && ag->size() > 0) decisions.field("AssignedGenericResources", *ag);
            if (newT.get("Volumes") != nullptr && task_state(at(&newT, "Status", "State")) == ASSIGNED) decisions.field("Volumes", *newT.get("Volumes"));
            decisions.end();
        }
        return decisions.finish();
    }

    // chooseTaskVolumes for a preassigned task on its node (scheduler.go:663-677): the attachments go into newT — nothing is reserved,
    // the reference reserves only in scheduleNTasksOnNodes and at start-up. false: a mount found no volume, newT carries the error.
    bool chooseForPreassigned(const Value& t, const NodeInfo& ni, Value& newT) {
        const std::vector<const Value*> cms = clusterMounts(t);
        if (cms.empty()) return true;
        const swp_task_desc d = taskDesc(t);
        uint32_t att[SWP_MAX_MOUNTS], n_out = 0, failed = 0;
        ck(swp_choose_volumes(e_, d.flags >> SWP_TASK_MOUNTS_SHIFT, ni.idx, att, &n_out, &failed), "swp_choose_volumes");
        bookChooseRemainder(att, n_out ? n_out : std::min<size_t>(failed, cms.size()), as_str(ni.node.get("ID")));
        if (n_out == 0) {
            Value status = statusCopy(t);
            status.set("Err", Value::str("cannot find volume to satisfy mount with source " + as_str(cms[std::min<size_t>(failed, cms.size() - 1)]->get("Source"))));
            newT.set("Status", status);
            return false;
        }
        Value vols = Value::array();
        for (size_t m = 0; m < cms.size(); ++m) {
            if (att[m] >= vol_idx_to_id_.size()) fail(SWP_EINVAL, "engine returned an unknown volume index");
            Value va = Value::object();
            va.set("ID", Value::str(vol_idx_to_id_[att[m]]));
            va.set("Source", Value::str(as_str(cms[m]->get("Source"))));
            va.set("Target", Value::str(as_str(cms[m]->get("Target"))));
            vols.push(va);
        }
        newT.set("Volumes", vols);
        return true;
    }

    // ---------------------------------------------------------------------------------------------- tick
    // tick, scheduler.go:429-488: task groups (ServiceID, SpecVersion) in first-seen order, then the one-off tasks in
    // queue order; every scheduling step is a device call.
    std::string tick() {
        using Item = QItem;
        prof_.start();
        std::vector<Item> queue = unassignedTasks_.take_all();   // (only tasks without a NodeID are ever queued: createTask / updateTask / enqueue)
        Decisions decisions;
        lastDecisions_.erase_if([](const PendingDecision& d) { return !d.preassigned; });   // the previous tick's decisions are final now
        if (queue.empty()) return decisions.finish();
        decisions.expect(queue.size());
        lastDecisions_.reserve(lastDecisions_.size() + queue.size());
        std::set<std::string> sids;
        {
            std::vector<char> seen(templates_.size() + queue.size(), 0);   // (a task re-queued without a template gets one here)
            size_t in_use = 0;
            for (Item& it : queue) {
                if (it.tmpl == NO_TMPL) it.tmpl = templateOf(it.second);
                if (it.tmpl >= seen.size()) seen.resize(it.tmpl + 1, 0);
                if (seen[it.tmpl]) continue;
                seen[it.tmpl] = 1;
                ++in_use;
                sids.insert(templates_[it.tmpl].service);
            }
            sweepTemplates(queue, in_use);
        }
        prof_.lap(0);
        try {
            refuseIrregularGeneric(queue);
            pushFailures(sids);
            prof_.lap(1);
        } catch (const Fail& f) {   // nothing was scheduled: the whole queue stays queued
            for (const Item& it : queue) defer(it.first, it.second, f, decisions);
            return decisions.finish();
        }
        // Whatever else fails below (every device call has its own handler; this is for the ones nobody thought of): the tasks without
        // a decision line go back on the queue and the decisions made so far are still returned — they are applied to allTasks_ and
        // the node rows already.
        std::vector<std::string> ids;
        for (const Item& it : queue) ids.push_back(it.first);
        // (a task deleted while it was queued is scheduled all the same — deleteTask leaves the queue alone, scheduler.go:350-366 — but it is
        // not in allTasks_, where the handler below finds the others' documents once scheduleQueue has moved the queue's away: its queued
        // document is kept aside. Rare: nothing is copied unless a deletion met a queued task since the last tick.)
        std::map<std::string, Value> orphans;
        if (!deletedWhileQueued_.empty()) {
            const std::set<std::string> gone(deletedWhileQueued_.begin(), deletedWhileQueued_.end());
            deletedWhileQueued_.clear();
            for (const Item& it : queue)
                if (gone.count(it.first) && allTasks_.find(it.first) == allTasks_.end()) orphans.emplace(it.first, it.second);
        }
        try {
            scheduleQueue(queue, decisions);
        } catch (const Fail& f) {
            const std::set<std::string> have(decisions.ids().begin(), decisions.ids().end());
            for (const std::string& id : ids) {
                if (have.count(id)) continue;
                auto t = allTasks_.find(id);
                if (t != allTasks_.end()) defer(id, t->second, f, decisions);
                else {
                    auto o = orphans.find(id);
                    if (o != orphans.end()) defer(id, o->second, f, decisions);
                }
            }
        }
        std::string out = decisions.finish();
        prof_.lap(6);
        prof_.report(ids.size());
        return out;
    }
    void scheduleQueue(std::vector<QItem>& queue, Decisions& decisions) {
        using Item = QItem;
        std::vector<std::vector<Item>> groups;
        std::map<FailureKey, size_t> group_of;
        std::vector<size_t> group_of_tmpl(templates_.size(), ~(size_t)0);   // (the key is looked up once per template, not once per task)
        std::vector<Item> one_off;
        one_off.reserve(queue.size());
        for (Item& it : queue) {
            const Template& tm = templates_[it.tmpl];
            if (tm.has_version) {   // :442-459: tasks with a spec version are grouped
                size_t& g = group_of_tmpl[it.tmpl];
                if (g == ~(size_t)0) {
                    auto ins = group_of.emplace(FailureKey{tm.service, tm.version}, groups.size());
                    if (ins.second) groups.emplace_back();
                    g = ins.first->second;
                }
                groups[g].push_back(std::move(it));
            } else one_off.push_back(std::move(it));
        }
        prof_.lap(2);
        runGroups(groups, 0, groups.size(), decisions);
        // one-off tasks (:460-466): a task with spread preferences is a group of one and keeps its place in the order
        std::vector<Item> run;
        std::vector<swp_task_desc> run_descs;   // Pipeline.SetTask once per task
        run.reserve(one_off.size());
        run_descs.reserve(one_off.size());
        for (Item& it : one_off) {
            swp_task_desc d;
            try {
                d = descOf(it.tmpl);
            } catch (const Fail& f) {
                defer(it.first, it.second, f, decisions);
                continue;
            }
            if (d.spread_set != 0) {
                runOneOffs(run, run_descs, decisions);
                run.clear();
                run_descs.clear();
                std::vector<std::vector<Item>> single(1);
                single[0].push_back(std::move(it));
                runGroups(single, 0, 1, decisions);
            } else {
                run.push_back(std::move(it));
                run_descs.push_back(d);
            }
        }
        prof_.lap(3);
        runOneOffs(run, run_descs, decisions);
    }

    // The failed half of applySchedulingDecisions (scheduler.go:472-487 for tick, :416-425 for preassigned tasks): the store
    // commit of a decision did not go through (stale Meta.Version :533-545, node no longer READY :560-567, a conflicting
    // write). Undo it: allTasks gets the old task back, NodeInfo.removeTask(new) returns the resources, and the old task is
    // queued again (tick) or stays pending (preassigned). Valid until the next tick / processPreassignedTasks.
    bool rejectDecision(const std::string& tid) {
        auto it = lastDecisions_.find(tid);
        if (it == lastDecisions_.end()) return false;
        const PendingDecision pd = it->second;
        lastDecisions_.erase(tid);
        auto cur = allTasks_.find(tid);
        if (cur != allTasks_.end()) {
            const Value newT = cur->second;
            releaseTaskVolumes(newT);   // scheduler.go:480-483 and :422-424 (a preassigned task's attachments were never reserved: releasing them finds nothing)
            auto n = nodes_.find(as_str(newT.get("NodeID")));
            if (n != nodes_.end() && !truthy(pd.old.get("NodeID")) ) removeTask(n->second, newT);                 // tick: the node was chosen by this decision
            else if (n != nodes_.end() && pd.preassigned && task_state(at(&newT, "Status", "State")) == ASSIGNED) removeTask(n->second, newT);
        }
        allTasks_[tid] = pd.old;
        if (pd.preassigned) pendingPreassignedTasks_.put(tid, pd.old);
        else unassignedTasks_.put(tid, pd.old);   // enqueue(decision.old), :486
        return true;
    }

    // the commit path (include/swp_sched.h swp_sched_commit_plan): the last call's decisions in commit order
    Value commitPlan(uint32_t max_changes) {
        if (max_changes == 0) max_changes = 200;   // MaxChangesPerTransaction, manager/state/store/memory.go:47
        std::map<uint32_t, std::vector<std::string>> by_node;   // engine node index -> task ids (map order: ascending id)
        Value unassigned = Value::array();
        for (const auto* dp : lastDecisions_.sorted()) {
            const auto& kv = *dp;
            auto cur = allTasks_.find(kv.first);
            const std::string nid = cur != allTasks_.end() ? as_str(cur->second.get("NodeID")) : std::string();
            auto n = nid.empty() ? nodes_.end() : nodes_.find(nid);
            if (n == nodes_.end()) unassigned.push(Value::str(kv.first));
            else by_node[n->second.idx].push_back(kv.first);
        }
        Value nodes = Value::array(), txs = Value::array(), tx = Value::array();
        auto add = [&](const std::string& tid) {
            if (tx.size() == max_changes) {
                txs.push(tx);
                tx = Value::array();
            }
            tx.push(Value::str(tid));
        };
        for (const auto& kv : by_node) {
            swp_node_row row;
            ck(swp_node_get(e_, kv.first, &row), "swp_node_get");
            Value g = Value::object(), ids = Value::array();
            for (const std::string& tid : kv.second) {
                ids.push(Value::str(tid));
                add(tid);
            }
            g.set("NodeID", Value::str(idx_to_id_[kv.first]));
            g.set("Version", Value::integer((int64_t)row.version));
            g.set("Tasks", ids);
            nodes.push(g);
        }
        for (const Value& v : *unassigned.a) add(v.s);
        if (tx.size() > 0) txs.push(tx);
        // the volume side of a commit (scheduler.go:548-610): a decision whose attachment names a volume that is not ACTIVE any more (or
        // that this scheduler does not hold) is called off; every other attachment wants its volume published on the task's node — a
        // PENDING_PUBLISH status where the volume has none for that node yet. Computed from the volume documents as EventUpdateVolume left
        // them; the caller repeats the "already published?" look at its store's copy inside its transaction, as :591-606 does.
        Value vol_failed = Value::array(), publish = Value::array();
        std::vector<std::pair<std::string, std::vector<std::string>>> pub;   // volume -> nodes, in commit order
        for (const auto* dp : lastDecisions_.sorted()) {
            const auto& kv = *dp;
            auto cur = allTasks_.find(kv.first);
            if (cur == allTasks_.end()) continue;
            const Value* vols = cur->second.get("Volumes");
            const std::string& nid = as_str(cur->second.get("NodeID"));
            if (vols == nullptr || !vols->is_arr() || nid.empty()) continue;
            bool ok = true;
            for (const Value& va : *vols->a) {
                auto v = volumes_.find(as_str(va.get("ID")));
                if (v == volumes_.end() || enum_value(at(&v->second.doc, "Spec", "Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}}) != 0) ok = false;
            }
            if (!ok) {
                vol_failed.push(Value::str(kv.first));
                continue;
            }
            for (const Value& va : *vols->a) {
                const std::string& vid = as_str(va.get("ID"));
                bool published = false;
                if (const Value* ps = volumes_.at(vid).doc.get("PublishStatus"))
                    if (ps->is_arr())
                        for (const Value& st : *ps->a) published = published || as_str(st.get("NodeID")) == nid;
                if (published) continue;
                auto it = std::find_if(pub.begin(), pub.end(), [&](const auto& p) { return p.first == vid; });
                if (it == pub.end()) it = pub.insert(pub.end(), {vid, {}});
                if (std::find(it->second.begin(), it->second.end(), nid) == it->second.end()) it->second.push_back(nid);
            }
        }
        for (const auto& p : pub) {
            Value u = Value::object(), ns = Value::array();
            for (const std::string& n : p.second) ns.push(Value::str(n));
            u.set("VolumeID", Value::str(p.first));
            u.set("NodeIDs", ns);
            publish.push(u);
        }
        Value plan = Value::object();
        plan.set("Nodes", nodes);
        plan.set("Unassigned", unassigned);
        plan.set("Transactions", txs);
        plan.set("VolumeFailed", vol_failed);
        plan.set("Publish", publish);
        return plan;
    }
    uint32_t rejectNode(const std::string& nid) {
        std::vector<std::string> ids;
        for (const auto* dp : lastDecisions_.sorted()) {
            const auto& kv = *dp;
            auto cur = allTasks_.find(kv.first);
            if (cur != allTasks_.end() && as_str(cur->second.get("NodeID")) == nid) ids.push_back(kv.first);
        }
        uint32_t n = 0;
        for (const std::string& tid : ids) n += rejectDecision(tid) ? 1u : 0u;
        return n;
    }

    // what the scheduler holds (swp_sched_counts)
    void counts(uint64_t out[4]) const {
        out[0] = allTasks_.size();
        out[1] = unassignedTasks_.size();
        out[2] = lastDecisions_.size();
        out[3] = templates_.size();
    }

    // ---------------------------------------------------------------------------------------------- constraint enforcer
    // constraintenforcer.rejectNoncompliantTasks for many nodes (constraint_enforcer.go:65-196) through swp_enforce.
    Value enforce(const Value& req) {
        const Value* node_docs = req.get("nodes");
        const Value* tbn = req.get("tasks_by_node");
        const Value* services = req.get("services");
        std::vector<swp_enforce_node> nrec;
        std::vector<swp_enforce_task> trec;
        std::vector<std::pair<std::string, std::string>> owners;
        struct GenericWalk { std::string nid; uint32_t first; generic::List avail; std::vector<generic::List> assigned; std::vector<bool> has; };
        std::vector<GenericWalk> walks;
        Value out = Value::object();
        if (node_docs == nullptr || !node_docs->is_arr()) return out;
        for (const Value& nd : *node_docs->a) {
            if (enum_value(at(&nd, "Spec", "Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}}) != AVAILABILITY_ACTIVE) continue;   // :70-72
            const std::string& nid = task_id(nd);
            std::vector<const Value*> tasks;
            if (const Value* lst = tbn != nullptr ? tbn->get_key(nid) : nullptr)
                if (lst->is_arr())
                    for (const Value& t : *lst->a) tasks.push_back(&t);
            std::stable_sort(tasks.begin(), tasks.end(), [](const Value* a, const Value* b) { return task_id(*a) < task_id(*b); });
            const Value* res = at(&nd, "Description", "Resources");
            // The generic half (constraint_enforcer.go:186-200) is walked on the host AFTER the device call (below): the device's verdicts
            // — constraints, then memory and cpu accounted task by task — do not depend on it.
            bool generic = false;
            for (const Value* t : tasks) generic = generic || t->get("AssignedGenericResources") != nullptr;
            if (generic) {
                GenericWalk w;
                w.nid = nid;
                w.first = (uint32_t)trec.size();
                w.avail = generic::decode(res != nullptr ? res->get("Generic") : nullptr);
                for (const Value* t : tasks) {
                    bool nil = false;
                    generic::List a = generic::decode(t->get("AssignedGenericResources"), &nil);
                    w.assigned.push_back(std::move(a));
                    w.has.push_back(t->get("AssignedGenericResources") != nullptr && !nil);
                }
                walks.push_back(std::move(w));
            }
            auto ni = nodes_.find(nid);
            if (ni == nodes_.end()) fail(SWP_ENOTFOUND, "enforce: node " + nid + " is not in the nodeSet");
            const uint32_t first = (uint32_t)trec.size();
            for (const Value* t : tasks) {
                const Value* svc = services != nullptr ? services->get_key(as_str(t->get("ServiceID"))) : nullptr;
                // :152-168: the CURRENT service spec decides; the task's own placement only when the service is gone
                const Value* pl = svc != nullptr ? at(svc, "Spec", "Task", "Placement") : at(t, "Spec", "Placement");
                swp_enforce_task r;
                std::memset(&r, 0, sizeof r);
                r.constraint_set = constraintSet(pl != nullptr ? pl->get("Constraints") : nullptr);
                const Value* rsv = at(t, "Spec", "Resources", "Reservations");
                if (rsv != nullptr) {
                    r.cpu = as_i64(rsv->get("NanoCPUs"));
                    r.mem = as_i64(rsv->get("MemoryBytes"));
                    r.flags = SWP_ENF_RESERVATIONS;
                }
                r.desired_state = (uint32_t)task_state(t->get("DesiredState"));
                r.state = (uint32_t)task_state(at(t, "Status", "State"));
                trec.push_back(r);
                owners.emplace_back(nid, task_id(*t));
            }
            swp_enforce_node n;
            std::memset(&n, 0, sizeof n);
            n.node = ni->second.idx;
            n.first_task = first;
            n.n_tasks = (uint32_t)trec.size() - first;
            n.cpu = res != nullptr ? as_i64(res->get("NanoCPUs")) : 0;
            n.mem = res != nullptr ? as_i64(res->get("MemoryBytes")) : 0;
            nrec.push_back(n);
            out.set(nid, Value::array());
        }
        if (trec.empty()) return out;
        std::vector<uint8_t> rej(trec.size(), 0);
        ck(swp_enforce(e_, nrec.data(), (uint32_t)nrec.size(), trec.data(), (uint32_t)trec.size(), rej.data()), "swp_enforce");
        // :186-200 for the nodes whose tasks hold generic resources: a task the device kept claims what it was assigned from the node's
        // list (ClaimResources against a throw-away store); the first task whose assignment is no longer there is rejected and ENDS the
        // node's loop (`break loop`): nothing behind it is looked at, whatever the device said about it.
        for (GenericWalk& w : walks) {
            bool broke = false;
            for (size_t k = 0; k < w.assigned.size(); ++k) {
                uint8_t& verdict = rej[w.first + k];
                if (broke) { verdict = 0; continue; }
                const swp_enforce_task& tr = trec[w.first + k];
                if (tr.desired_state < (uint32_t)ASSIGNED || tr.desired_state > (uint32_t)COMPLETE || tr.state >= (uint32_t)COMPLETE) continue;   // :118-126 (never rejected)
                if (verdict || !w.has[k]) continue;   // rejected by constraints / reservations (`continue`), or AssignedGenericResources == nil
                bool gone = false;
                for (const generic::Res& ta : w.assigned[k])
                    if (!generic::has_resource(ta, w.avail)) { gone = true; break; }
                if (gone) { verdict = 1; broke = true; continue; }
                generic::consume(&w.avail, w.assigned[k]);
            }
        }
        for (size_t i = 0; i < rej.size(); ++i)
            if (rej[i])
                for (json::Member& m : *out.o)
                    if (m.first == owners[i].first) { m.second.push(Value::str(owners[i].second)); break; }
        return out;
    }

  private:
    using Item = QItem;
    swp_engine* e_;
    int64_t now_ = 1000000000000LL;
    std::map<std::string, NodeInfo> nodes_;                               // nodeSet.nodes (ordered: deterministic call order)
    std::vector<std::string> idx_to_id_;
    std::vector<NodeInfo*> node_by_idx_;                                   // engine node index -> its NodeInfo (place(): no look-up by id)
    std::unordered_map<std::string, std::optional<uint64_t>> services_;   // store.GetService: existence + SpecVersion
    OrderedTasks unassignedTasks_;                                         // Scheduler.unassignedTasks
    std::map<std::string, std::set<std::string>> irregularGeneric_;          // node id -> generic kinds its available list holds more than once, not all Named (pushGeneric)
    std::vector<std::string> deletedWhileQueued_;                           // ids deleteTask met in the queue since the last tick (tick's catch-all path)
    OrderedTasks pendingPreassignedTasks_;                                 // Scheduler.pendingPreassignedTasks
    // the `old` half of every decision of the last tick / processPreassignedTasks (schedulingDecision.old, scheduler.go:53-56),
    // kept until the next one so that the caller can roll a decision back when its store commit failed (rejectDecision)
    struct PendingDecision { Value old; bool preassigned = false; };
    IdTable<PendingDecision> lastDecisions_;
    std::string engine_detail_;   // "<strerror>: <engine message>" of the last failed engine call
    // failure counts pushed to the engine: (node index, service, spec version) -> count, so that a bucket the clean-up
    // erased (nodeinfo.go:163-183) is reset there too
    std::map<std::tuple<uint32_t, std::string, int64_t>, uint32_t> pushedFailures_;
    std::set<std::string> preassignedTasks_;                               // Scheduler.preassignedTasks
    // Scheduler.volumes (volumes.go:19-46): the task -> usage maps; the engine holds the numbers checkVolume derives from them
    struct VolumeUse { std::string node; bool read_only; };
    struct VolumeRec { Value doc; uint32_t idx = 0; std::map<std::string, VolumeUse> tasks; std::map<std::string, int64_t> nodes; };   // volumeInfo, volumes.go:19-36
    std::map<std::string, VolumeRec> volumes_;
    std::map<std::string, std::string> volByName_;
    std::vector<std::string> vol_idx_to_id_;
    void pushVolumeUsage(const VolumeRec& r) {
        swp_volume_usage u{0, 0, SWP_PIN_NONE, 0};
        for (const auto& kv : r.tasks) {
            u.n_tasks += 1;
            if (!kv.second.read_only) u.n_writers += 1;
            auto n = nodes_.find(kv.second.node);
            const uint32_t idx = n == nodes_.end() ? SWP_PIN_MANY : n->second.idx;   // (a user on a node the nodeSet does not hold: no node of the set is that node)
            u.pin = u.pin == SWP_PIN_NONE ? idx : (u.pin == idx ? idx : SWP_PIN_MANY);
        }
        ck(swp_volume_set_usage(e_, r.idx, &u), "swp_volume_set_usage");
    }
    IdTable<Value> allTasks_;                                              // Scheduler.allTasks

    void ck(int rc, const char* what) {
        if (rc == SWP_OK) return;
        engine_detail_ = std::string(swp_strerror(rc)) + ": " + swp_last_error(e_);   // what a deferred decision line reports
        fail(rc, std::string(what) + ": " + engine_detail_);
    }
    uint32_t intern(int space, const std::string& s) {
        uint32_t id = 0;
        ck(swp_intern(e_, space, s.data(), s.size(), &id), "swp_intern");
        return id;
    }
    uint32_t folded(const Value* v) { return intern(SWP_SPACE_FOLDED, as_str(v)); }

    // "node.labels.<name>" / "engine.labels.<name>" (prefix compared with EqualFold, the name is case-sensitive:
    // constraint.go:177-199, nodeset.go:69-81)
    bool labelKey(const std::u32string& key, uint32_t& kind, uint32_t& id) {
        static const size_t nl = std::strlen("node.labels."), el = std::strlen("engine.labels.");
        if (key.size() > nl && equal_fold(key, nl, "node.labels.")) {
            kind = SWP_CK_NODE_LABEL;
            id = intern(SWP_SPACE_LABEL_KEY, encode(key, nl));
            return true;
        }
        if (key.size() > el && equal_fold(key, el, "engine.labels.")) {
            kind = SWP_CK_ENGINE_LABEL;
            id = intern(SWP_SPACE_LABEL_KEY, encode(key, el));
            return true;
        }
        return false;
    }
    // the key dispatch of Constraint.Match, constraint.go:109-203
    swp_constraint constraintStruct(const Expr& x) {
        swp_constraint c;
        std::memset(&c, 0, sizeof c);
        c.kind = SWP_CK_INVALID;
        c.op = (uint32_t)x.op;
        const std::string exp = encode(x.exp);
        c.value = intern(SWP_SPACE_FOLDED, exp);
        const size_t n = x.key.size();
        if (equal_fold(x.key, n, "node.id")) c.kind = SWP_CK_NODE_ID;
        else if (equal_fold(x.key, n, "node.hostname")) c.kind = SWP_CK_HOSTNAME;
        else if (equal_fold(x.key, n, "node.ip")) {
            c.kind = SWP_CK_IP;
            bool v4 = false;
            if (parse_ip(exp, c.ip, &v4)) {
                c.ip_kind = SWP_IP_SINGLE;
                c.ip_is_v4 = v4 ? 1 : 0;
            } else {
                // net.ParseCIDR: "<address>/<decimal prefix length>", the length bounded by the address family as written
                c.ip_kind = SWP_IP_MALFORMED;
                std::memset(c.ip, 0, sizeof c.ip);
                const size_t slash = exp.find('/');
                if (slash != std::string::npos) {
                    const std::string addr = exp.substr(0, slash), plen = exp.substr(slash + 1);
                    uint8_t a[16];
                    bool dummy = false;
                    const bool digits = !plen.empty() && plen.size() <= 3 && std::all_of(plen.begin(), plen.end(), [](char ch) { return ch >= '0' && ch <= '9'; });
                    if (digits && parse_ip(addr, a, &dummy)) {
                        const bool syntactic_v4 = addr.find(':') == std::string::npos;
                        const int bits = syntactic_v4 ? 32 : 128, len = std::atoi(plen.c_str());
                        if (len <= bits) {
                            c.ip_kind = SWP_IP_CIDR;
                            c.ip_is_v4 = syntactic_v4 ? 1 : 0;
                            c.prefix_len = (uint32_t)(len + (syntactic_v4 ? 96 : 0));
                            std::memcpy(c.ip, a, 16);
                        }
                    }
                }
            }
        } else if (equal_fold(x.key, n, "node.role")) c.kind = SWP_CK_ROLE;
        else if (equal_fold(x.key, n, "node.platform.os")) c.kind = SWP_CK_PLATFORM_OS;
        else if (equal_fold(x.key, n, "node.platform.arch")) c.kind = SWP_CK_PLATFORM_ARCH;
        else {
            uint32_t kind, key;
            if (labelKey(x.key, kind, key)) {
                c.kind = kind;
                c.key = key;
            }
        }
        return c;
    }

    // node document → numeric row + interned labels / plugins (every field the path reads)
    void upsertRow(const Value& doc, uint32_t idx, int64_t cpu, int64_t mem, uint32_t total) {
        swp_node_row row;
        std::memset(&row, 0, sizeof row);
        row.node = idx;
        row.cpu = cpu;
        row.mem = mem;
        row.total = total;
        uint32_t flags = 0;
        const int64_t st = enum_value(at(&doc, "Status", "State"), {{"UNKNOWN", 0}, {"DOWN", 1}, {"READY", 2}, {"DISCONNECTED", 3}});
        const int64_t av = enum_value(at(&doc, "Spec", "Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}});
        if (st == NODE_STATE_READY && av == AVAILABILITY_ACTIVE) flags |= SWP_NODE_READY;   // ReadyFilter, filter.go:41-44
        if (enum_value(doc.get("Role"), {{"WORKER", 0}, {"MANAGER", 1}}) == 1) flags |= SWP_NODE_MANAGER;
        row.id_fold = folded(doc.get("ID"));
        std::vector<swp_kv> lab, elab;
        std::vector<uint32_t> plugins;
        auto kvs = [&](const Value* labels, std::vector<swp_kv>& out) {
            if (!labels->is_obj()) return;
            for (const json::Member& m : *labels->o)
                out.push_back({intern(SWP_SPACE_LABEL_KEY, m.first), folded(&m.second), intern(SWP_SPACE_RAW, as_str(&m.second))});
        };
        if (const Value* labels = at(&doc, "Spec", "Annotations", "Labels")) {
            flags |= SWP_NODE_HAS_LABELS;
            kvs(labels, lab);
        }
        if (const Value* desc = doc.get("Description")) {
            flags |= SWP_NODE_HAS_DESC;
            row.hostname_fold = folded(desc->get("Hostname"));
            if (const Value* plat = desc->get("Platform")) {
                flags |= SWP_NODE_HAS_PLATFORM;
                row.os = intern(SWP_SPACE_OS, as_str(plat->get("OS")));
                row.arch = intern(SWP_SPACE_ARCH, as_str(plat->get("Architecture")));
                row.os_fold = folded(plat->get("OS"));
                row.arch_fold = folded(plat->get("Architecture"));
            }
            if (const Value* eng = desc->get("Engine")) {
                flags |= SWP_NODE_HAS_ENGINE;
                if (const Value* el = eng->get("Labels")) {
                    flags |= SWP_NODE_HAS_ELABELS;
                    kvs(el, elab);
                }
                const Value* pl = eng->get("Plugins");
                if (pl != nullptr && pl->is_arr())
                    for (const Value& p : *pl->a) {
                        const std::string &typ = as_str(p.get("Type")), &name = as_str(p.get("Name"));
                        if (typ == "Log") flags |= SWP_NODE_HAS_LOGPLUG;
                        plugins.push_back(intern(SWP_SPACE_PLUGIN, typ + '\0' + name));
                        static const std::string latest = ":latest";   // filter.go:189-199: "name" also matches "name:latest"
                        if (name.size() >= latest.size() && name.compare(name.size() - latest.size(), latest.size(), latest) == 0)
                            plugins.push_back(intern(SWP_SPACE_PLUGIN, typ + '\0' + name.substr(0, name.size() - latest.size())));
                    }
            }
        }
        bool v4 = false;
        if (parse_ip(as_str(at(&doc, "Status", "Addr")), row.ip, &v4)) flags |= SWP_NODE_IP_VALID | (v4 ? SWP_NODE_IP_V4 : 0u);
        else std::memset(row.ip, 0, sizeof row.ip);
        row.flags = flags;
        row.version = (uint64_t)as_i64(at(&doc, "Meta", "Version", "Index"));
        static const swp_kv no_kv = {0, 0, 0};
        static const uint32_t no_plugin = 0;
        ck(swp_node_upsert(e_, &row, lab.empty() ? &no_kv : lab.data(), (uint32_t)lab.size(), elab.empty() ? &no_kv : elab.data(), (uint32_t)elab.size(),
                           plugins.empty() ? &no_plugin : plugins.data(), (uint32_t)plugins.size()),
           "swp_node_upsert");
        // Description.CSIInfo: the node's topology per CSI plugin (volumes.go:272-278)
        std::vector<swp_csi> infos;
        std::vector<swp_seg> segs;
        if (const Value* cs = at(&doc, "Description", "CSIInfo"))
            if (cs->is_arr())
                for (const Value& c : *cs->a) {
                    swp_csi ci;
                    std::memset(&ci, 0, sizeof ci);
                    ci.plugin = intern(SWP_SPACE_CSI, as_str(c.get("PluginName")));
                    ci.seg_off = (uint32_t)segs.size();
                    if (const Value* top = c.get("AccessibleTopology")) {
                        ci.has_topology = 1;
                        if (const Value* sg = top->get("Segments"))
                            if (sg->is_obj())
                                for (const json::Member& m : *sg->o) segs.push_back({intern(SWP_SPACE_CSI, m.first), intern(SWP_SPACE_CSI, as_str(&m.second))});
                    }
                    ci.n_seg = (uint32_t)segs.size() - ci.seg_off;
                    infos.push_back(ci);
                }
        static const swp_csi no_csi = {0, 0, 0, 0};
        static const swp_seg no_seg = {0, 0};
        ck(swp_node_set_csi(e_, idx, infos.empty() ? &no_csi : infos.data(), (uint32_t)infos.size(), segs.empty() ? &no_seg : segs.data(), (uint32_t)segs.size()), "swp_node_set_csi");
    }

    // taskReservations, nodeinfo.go:156-161
    static void taskReservations(const Value& t, int64_t& cpu, int64_t& mem) {
        const Value* r = at(&t, "Spec", "Resources", "Reservations");
        cpu = r != nullptr ? as_i64(r->get("NanoCPUs")) : 0;
        mem = r != nullptr ? as_i64(r->get("MemoryBytes")) : 0;
    }
    // host-mode published ports of the task (filter.go:322-333, nodeinfo.go:78-84,139-145) → port-set id, 0 = none
    uint32_t portSet(const Value& t) {
        const Value* ports = at(&t, "Endpoint", "Ports");
        if (ports == nullptr || !ports->is_arr()) return 0;
        std::vector<swp_port> ps;
        for (const Value& p : *ports->a) {
            if (enum_value(p.get("PublishMode"), {{"INGRESS", 0}, {"HOST", 1}}) != PUBLISH_HOST) continue;
            const int64_t port = as_i64(p.get("PublishedPort"));
            if (port == 0) continue;
            ps.push_back({(uint32_t)enum_value(p.get("Protocol"), {{"TCP", 0}, {"UDP", 1}, {"SCTP", 2}}, 0, true), (uint32_t)port});
        }
        if (ps.empty()) return 0;
        uint32_t id = 0;
        ck(swp_port_set(e_, ps.data(), (uint32_t)ps.size(), &id), "swp_port_set");
        return id;
    }
    // the node's available generic list as the engine sees it: one count per kind (swp_node_set_generic)
    void pushGeneric(const NodeInfo& ni) {
        std::vector<swp_generic> items;
        for (const auto& kv : generic::counts(ni.availGeneric)) {
            if (kv.second >= (1ll << 31)) unsupported("a generic resource count of 2^31 or more stays on the Go path");
            items.push_back({intern(SWP_SPACE_GENERIC_KIND, kv.first), 0u, kv.second});
        }
        ck(swp_node_set_generic(e_, ni.idx, items.data(), (uint32_t)items.size()), "swp_node_set_generic");
        // (a list one count per kind cannot stand for — swp_generic.hpp irregular_kinds: tick() looks at this before it schedules)
        const std::string& nid = as_str(ni.node.get("ID"));
        std::set<std::string> irr = generic::irregular_kinds(ni.availGeneric);
        if (irr.empty()) irregularGeneric_.erase(nid);
        else irregularGeneric_[nid] = std::move(irr);
    }
    // A node lists a generic kind in a way the engine's one count per kind cannot stand for, and a queued task reserves that kind: the
    // engine's arithmetic inside a device call (count -= request) would not be the reference's for a second task on that node. Such a
    // tick stays on the Go path as a whole — a task left out would change the others' answers. Throws; tick() defers the queue.
    void refuseIrregularGeneric(const std::vector<QItem>& queue) {
        if (irregularGeneric_.empty()) return;
        std::vector<char> seen(templates_.size(), 0);
        for (const QItem& it : queue) {
            if (it.tmpl >= templates_.size() || seen[it.tmpl]) continue;
            seen[it.tmpl] = 1;
            const Template& tm = templates_[it.tmpl];
            if (!tm.wants_generic) continue;
            for (const generic::Res& r : generic::decode(at(&tm.exemplar, "Spec", "Resources", "Reservations", "Generic")))
                for (const auto& kv : irregularGeneric_)
                    if (kv.second.count(r.kind))
                        fail(SWP_EUNSUPPORTED, ("node " + kv.first + " lists the generic kind '" + r.kind + "' more than once (its type changed under a running task): a tick with tasks that reserve it stays on the Go path").c_str());
        }
    }
    void commit(const NodeInfo& ni, const Value& t, bool counted, bool with_resources, bool add) {
        swp_placement p;
        std::memset(&p, 0, sizeof p);
        p.node = ni.idx;
        p.service = intern(SWP_SPACE_SERVICE, as_str(t.get("ServiceID")));
        if (with_resources) {
            taskReservations(t, p.cpu, p.mem);
            p.port_set = portSet(t);
        }
        p.counted = counted ? 1 : 0;
        ck(swp_commit(e_, &p, 1, add ? 1 : 0), "swp_commit");
    }
    // NodeInfo.addTask, nodeinfo.go:108-154; true when nodeInfo was modified
    bool addTask(NodeInfo& ni, const Value& t) {
        const std::string& id = task_id(t);
        const int64_t ds = task_state(t.get("DesiredState"));
        Value* old = ni.Tasks.find(id);
        if (old != nullptr) {
            const int64_t ods = task_state(old->get("DesiredState"));
            if (ds <= COMPLETE && ods > COMPLETE) {          // :113-119: the task counts again
                *old = t;
                commit(ni, t, true, false, true);
                return true;
            }
            if (ods <= COMPLETE && ds > COMPLETE) {          // :120-126: the task stops counting
                *old = t;
                commit(ni, t, true, false, false);
                return true;
            }
            return false;
        }
        // :128-137: a fresh AssignedGenericResources, then Claim against the node's available list
        Value stored = t.shallow_copy();
        generic::List assigned;
        generic::claim(&ni.availGeneric, &assigned, generic::decode(at(&t, "Spec", "Resources", "Reservations", "Generic")));
        stored.set("AssignedGenericResources", generic::encode(assigned));
        ni.Tasks.put(id, stored);
        auto all = allTasks_.find(id);
        if (all != allTasks_.end()) all->second = stored;   // (the reference writes through the one *api.Task both maps point to)
        commit(ni, t, ds <= COMPLETE, true, true);
        pushGeneric(ni);
        return true;
    }
    // NodeInfo.removeTask, nodeinfo.go:66-104
    bool removeTask(NodeInfo& ni, const Value& t) {
        const Value* old = ni.Tasks.find(task_id(t));
        if (old == nullptr) return false;
        const bool counted = task_state(old->get("DesiredState")) <= COMPLETE;
        ni.Tasks.erase(task_id(t));
        commit(ni, t, counted, true, false);
        // :95-104: the task's AssignedGenericResources go back — unless the node's description lists no generic resources at all
        bool desc_nil = true;
        const generic::List node_res = generic::decode(at(&ni.node, "Description", "Resources", "Generic"), &desc_nil);
        if (!desc_nil) {
            generic::reclaim(&ni.availGeneric, generic::decode(t.get("AssignedGenericResources")), node_res);
            pushGeneric(ni);
        }
        return true;
    }
    // NodeInfo.taskFailed (+ cleanupFailures), nodeinfo.go:163-202
    void taskFailed(NodeInfo& ni, const Value& t) {
        if (now_ - ni.lastCleanup >= MONITOR_FAILURES) {
            for (auto it = ni.recentFailures.begin(); it != ni.recentFailures.end();) {
                bool recent = false;
                for (int64_t ts : it->second) recent = recent || (now_ - ts < MONITOR_FAILURES);
                if (!recent) it = ni.recentFailures.erase(it);
                else ++it;
            }
            ni.lastCleanup = now_;
        }
        FailureKey k{as_str(t.get("ServiceID")), as_i64(at(&t, "SpecVersion", "Index"))};
        std::vector<int64_t>& lst = ni.recentFailures[k];
        size_t expired = 0;
        for (int64_t ts : lst) {
            if (now_ - ts < MONITOR_FAILURES) break;
            ++expired;
        }
        lst.erase(lst.begin(), lst.begin() + (std::ptrdiff_t)expired);
        lst.push_back(now_);
    }
    // NodeInfo.countRecentFailures, nodeinfo.go:206-221
    uint32_t countRecentFailures(const NodeInfo& ni, const FailureKey& k) const {
        auto it = ni.recentFailures.find(k);
        if (it == ni.recentFailures.end()) return 0;
        const std::vector<int64_t>& lst = it->second;
        int64_t count = (int64_t)lst.size();
        for (int64_t i = count - 1; i >= 0; --i)
            if (now_ - lst[(size_t)i] > MONITOR_FAILURES) {
                count -= i + 1;
                break;
            }
        return (uint32_t)count;
    }
    // the failure counts nodeLess reads (scheduler.go:706-735) for the services of the coming batch, at its `now`
    void pushFailures(const std::set<std::string>& sids) {
        std::map<std::tuple<uint32_t, std::string, int64_t>, uint32_t> now;
        for (auto& kv : nodes_)
            for (auto& f : kv.second.recentFailures)
                if (sids.count(f.first.first)) now[{kv.second.idx, f.first.first, f.first.second}] = countRecentFailures(kv.second, f.first);
        // buckets the engine still holds a count for but the node no longer has (erased by cleanupFailures, or the node left):
        // countRecentFailures would say 0 (nodeinfo.go:206-221), so must the engine
        for (auto it = pushedFailures_.begin(); it != pushedFailures_.end();) {
            if (sids.count(std::get<1>(it->first)) && now.find(it->first) == now.end()) {
                if (it->second != 0 && std::get<0>(it->first) < idx_to_id_.size() && nodes_.count(idx_to_id_[std::get<0>(it->first)]))
                    ck(swp_node_set_failures(e_, std::get<0>(it->first), intern(SWP_SPACE_SERVICE, std::get<1>(it->first)), (uint64_t)std::get<2>(it->first), 0), "swp_node_set_failures");
                it = pushedFailures_.erase(it);
            } else ++it;
        }
        for (auto& kv : now) {
            ck(swp_node_set_failures(e_, std::get<0>(kv.first), intern(SWP_SPACE_SERVICE, std::get<1>(kv.first)), (uint64_t)std::get<2>(kv.first), kv.second), "swp_node_set_failures");
            pushedFailures_[kv.first] = kv.second;
        }
    }

    static Value statusCopy(const Value& t) {
        const Value* s = t.get("Status");
        return (s != nullptr && s->is_obj()) ? s->shallow_copy() : Value::object();
    }
    // The decisions of a device call are booked task by task, and every one of them lands in memory nothing has touched since the
    // task's event (its entry in allTasks, its node's task list): the answer is known for the WHOLE call, so the lines a task will
    // need are asked for a few tasks ahead — stage 1: the table cells and the NodeInfo, stage 2: what those point to.
    static constexpr size_t PREFETCH_FAR = 12, PREFETCH_NEAR = 6;
    void prefetchBooking(const Item& item, int32_t n, int stage) const {
        const uint64_t h = id_hash(item.first);
        allTasks_.prefetch(h, stage);
        lastDecisions_.prefetch(h, stage);
        if (n < 0 || (size_t)n >= node_by_idx_.size() || node_by_idx_[(size_t)n] == nullptr) return;
        const NodeInfo* ni = node_by_idx_[(size_t)n];
        if (stage == 1) {
            __builtin_prefetch(&ni->Tasks);
            __builtin_prefetch(idx_to_id_.data() + n);
        } else ni->Tasks.prefetch();
    }
    // scheduleNTasksOnNodes' bookkeeping for one placed task (scheduler.go:868-897); the numeric addTask already
    // happened on the device
    void place(const Item& item, int32_t n, Decisions& decisions, const uint32_t* att = nullptr) {
        const std::string& tid = item.first;
        const Value& t = item.second;
        const Template& tm = templates_[item.tmpl];   // (what follows reads of the Spec is the same for every task of the template)
        const uint64_t tid_hash = id_hash(tid);
        if ((size_t)n >= idx_to_id_.size()) fail(SWP_EINVAL, "engine returned an unknown node index");
        const std::string& nid = idx_to_id_[(size_t)n];
        // (every assigned task's Status is the same two fields: ONE shared sub-document — sub-documents are immutable once built, a
        // handler that changes a status copies it first: statusCopy)
        static const Value assigned_status = [] {
            Value st = Value::object();
            st.set("State", Value::integer(ASSIGNED));
            st.set("Message", Value::str("scheduler assigned task to node"));
            return st;
        }();
        static const std::string assigned_message = "scheduler assigned task to node", no_err;
        // newT = the task with NodeID and Status set: ONE pass over the document's members, one allocation; what the decision line and the
        // checks below need of the document is picked up on the way
        if (t.kind != Value::Obj || !t.o) fail(SWP_EINVAL, "a queued task document is not a JSON object");   // (create_task / update_task only queue objects: a guard, not a path)
        Value newT;
        newT.kind = Value::Obj;
        newT.o = std::make_shared<std::vector<json::Member>>();
        newT.o->reserve(t.o->size() + 2);
        const Value *v_service = nullptr, *v_status = nullptr, *v_spec = nullptr;
        for (const json::Member& m : *t.o) {
            if (Value::key_is(m.first, "NodeID")) continue;
            if (Value::key_is(m.first, "Status")) { v_status = &m.second; continue; }
            if (Value::key_is(m.first, "ServiceID")) v_service = &m.second;
            else if (Value::key_is(m.first, "Spec")) v_spec = &m.second;
            newT.o->push_back(m);
        }
        newT.o->emplace_back("NodeID", Value::str(nid));
        newT.o->emplace_back("Status", assigned_status);
        NodeInfo* nip = (size_t)n < node_by_idx_.size() ? node_by_idx_[(size_t)n] : nullptr;
        if (nip == nullptr) fail(SWP_EINVAL, "engine placed a task on a node the nodeSet does not hold");
        NodeInfo& ni_second = *nip;
        // nodeInfo.addTask(&newT) (:886-888): the counts moved on the device already; WHICH resources the task holds is decided here
        const generic::List want = tm.wants_generic ? generic::decode(at(v_spec, "Resources", "Reservations", "Generic")) : generic::List();
        if (!want.empty()) {
            generic::List assigned;
            generic::claim(&ni_second.availGeneric, &assigned, want);
            newT.set("AssignedGenericResources", generic::encode(assigned));
            genericTouched_.insert(nid);   // pushed once the whole call's placements are booked (pushTouched): then the counts equal
                                           // what the engine's own arithmetic left and the call changes nothing
        }
        // newT.Volumes = attachments; reserveTaskVolumes(&newT) (scheduler.go:862-874): what the engine chose on the node, in mount order; a
        // mount that found no volume leaves the task without attachments (the reference logs the error and assigns it all the same)
        const std::vector<const Value*> cms = tm.has_mounts ? clusterMounts(t) : std::vector<const Value*>();
        size_t n_chosen = 0;   // the mounts chooseTaskVolumes found a volume for, in order (all of them: the task gets its attachments)
        if (!cms.empty() && att != nullptr) {
            while (n_chosen < cms.size() && att[n_chosen] != SWP_NO_VOLUME) ++n_chosen;
            bookChooseRemainder(att, n_chosen, nid);
        }
        if (!cms.empty() && n_chosen == cms.size()) {
            Value vols = Value::array();
            for (size_t m = 0; m < cms.size(); ++m) {
                if (att[m] >= vol_idx_to_id_.size()) fail(SWP_EINVAL, "engine returned an unknown volume index");
                Value va = Value::object();
                va.set("ID", Value::str(vol_idx_to_id_[att[m]]));
                va.set("Source", Value::str(as_str(cms[m]->get("Source"))));
                va.set("Target", Value::str(as_str(cms[m]->get("Target"))));
                vols.push(va);
            }
            newT.set("Volumes", vols);
            reserveTaskVolumes(newT);
        }
        decisions.begin(tid, as_str(v_service), nid, ASSIGNED, assigned_message, no_err, task_state(v_status != nullptr ? v_status->get("State") : nullptr));
        if (!want.empty()) decisions.field("AssignedGenericResources", *newT.get("AssignedGenericResources"));
        if (newT.get("Volumes") != nullptr) decisions.field("Volumes", *newT.get("Volumes"));
        decisions.end();
        ni_second.Tasks.put(tid, tid_hash, newT);
        lastDecisions_.at(tid, tid_hash) = PendingDecision{t, false};
        allTasks_.at(tid, tid_hash) = std::move(newT);
    }
    // lastDecisions_[tid] = {old, preassigned}; a tick decides its tasks in id order more often than not: try the end of the map first
    void rememberDecision(const std::string& tid, const Value& old, bool preassigned) {
        lastDecisions_[tid] = PendingDecision{old, preassigned};
    }
    // noSuitableNode, scheduler.go:928-971
    void noSuitableNode(const Item& item, const uint32_t* hist, Decisions& decisions) {
        const std::string& tid = item.first;
        const Value& t = item.second;
        const uint64_t tid_hash = id_hash(tid);
        auto svc = services_.find(as_str(t.get("ServiceID")));
        if (svc == services_.end()) return;   // :935-939: the service is gone, the task is dropped
        Value newT = t.shallow_copy();
        const Value* tv = at(&t, "SpecVersion", "Index");
        if (svc->second.has_value() && tv != nullptr && *svc->second > (uint64_t)as_i64(tv)) {
            // :940-953: a task of an old revision that is meant to shut down anyway is moved to SHUTDOWN instead of retried
            if (task_state(at(&t, "Status", "State")) == PENDING && task_state(t.get("DesiredState")) >= SHUTDOWN) {
                Value status = statusCopy(t);
                status.set("State", Value::integer(SHUTDOWN));
                status.set("Err", Value::str(""));
                newT.set("Status", status);
            }
        } else {
            // The tasks of a batch that find no node do so for a handful of reasons and from the same status: the new Status is ONE shared
            // sub-document per (old status, histogram) — sub-documents are immutable once built — kept in a small direct-mapped memo.
            const Value* old_status = t.get("Status");
            uint64_t hh = 0xCBF29CE484222325ull;
            for (int k = 0; k < SWP_NFILTERS; ++k) hh = hash_mix(hh, hist[k]);
            NsnMemo& memo = nsn_[(hh >> 20) % (sizeof nsn_ / sizeof nsn_[0])];
            if (!(memo.valid && std::memcmp(memo.hist, hist, sizeof memo.hist) == 0 && equal_value(old_status, memo.old.is_null() ? nullptr : &memo.old))) {
                const std::string ex = explain(hist);
                std::string err;
                err.reserve(ex.size() + 20);
                err += "no suitable node";
                if (!ex.empty()) { err += " ("; err += ex; err += ')'; }
                // the old Status with Err replaced (or added behind the other members): one vector of the right size
                Value status = Value::object();
                bool had_err = false;
                if (old_status != nullptr && old_status->is_obj()) {
                    status.o->reserve(old_status->o->size() + 1);
                    for (const json::Member& m : *old_status->o) {
                        if (Value::key_is(m.first, "Err")) { status.o->emplace_back(m.first, Value::str(std::move(err))); had_err = true; }
                        else status.o->push_back(m);
                    }
                }
                if (!had_err) status.o->emplace_back("Err", Value::str(std::move(err)));
                memo.status = std::move(status);
                memo.old = old_status != nullptr ? *old_status : Value();
                std::memcpy(memo.hist, hist, sizeof memo.hist);
                memo.valid = true;
            }
            newT.set("Status", memo.status);
            unassignedTasks_.put(tid, newT, item.tmpl);   // enqueue again, :968 (its template is what it was: nothing SetTask reads changed)
        }
        decisions.begin(t, newT);
        decisions.end();
        lastDecisions_.at(tid, tid_hash) = PendingDecision{t, false};
        allTasks_.at(tid, tid_hash) = std::move(newT);
    }
    struct NsnMemo { Value status, old; uint32_t hist[SWP_NFILTERS]; bool valid = false; };   // noSuitableNode: a new Status, and what it was made from
    NsnMemo nsn_[64];
    // A device call failed for these tasks (a predicate set the engine refuses, a device error): nothing of the call was applied, so the tasks go back on the queue — the Go shim routes a deferred task to the
    // reference's own scheduleTaskGroup — and the tick carries on with the rest. One decision line per task says why.
    void defer(const std::string& tid, const Value& t, const Fail& f, Decisions& decisions) {
        unassignedTasks_.put(tid, t);
        last_error = f.msg;
        const bool from_engine = !engine_detail_.empty() && f.msg.size() >= engine_detail_.size() &&
                                 f.msg.compare(f.msg.size() - engine_detail_.size(), engine_detail_.size(), engine_detail_) == 0;
        const std::string why = "swp: deferred to the host scheduler: " + (from_engine ? engine_detail_ : f.msg);
        decisions.begin(t, t, &why);
        decisions.field("Deferred", Value::boolean(true));
        decisions.end();
    }
    // groups[from, to): one swp_schedule_groups call, groups in order. One device call must not mix spec versions of one
    // service (the failure buckets are per (service, version)): cut the ordered list where that would happen.
    void runGroups(std::vector<std::vector<Item>>& groups, size_t from, size_t to, Decisions& decisions) {
        if (from >= to) return;
        std::map<std::string, int64_t> seen;
        size_t cut = to;
        for (size_t i = from; i < to; ++i) {
            const Value& t0 = groups[i][0].second;
            const std::string& sid = as_str(t0.get("ServiceID"));
            const int64_t ver = as_i64(at(&t0, "SpecVersion", "Index"));
            auto ins = seen.emplace(sid, ver);
            if (!ins.second && ins.first->second != ver) { cut = i; break; }
        }
        if (cut != to) {
            runGroups(groups, from, cut, decisions);
            std::set<std::string> sids;
            for (size_t i = cut; i < to; ++i) sids.insert(as_str(groups[i][0].second.get("ServiceID")));
            try {
                pushFailures(sids);
            } catch (const Fail& f) {   // the counts of the other spec version did not reach the engine: those groups wait for the next tick
                for (size_t i = cut; i < to; ++i)
                    for (const Item& it : groups[i]) defer(it.first, it.second, f, decisions);
                return;
            }
            runGroups(groups, cut, to, decisions);
            return;
        }
        std::vector<swp_task_desc> descs;
        std::vector<uint32_t> sizes;
        size_t total = 0;
        for (size_t i = from; i < to; ++i) {
            try {
                descs.push_back(descOf(groups[i][0].tmpl));
            } catch (const Fail& f) {   // a predicate set the engine refuses: this group is deferred, the others run
                runGroups(groups, from, i, decisions);
                for (const Item& it : groups[i]) defer(it.first, it.second, f, decisions);
                runGroups(groups, i + 1, to, decisions);
                return;
            }
            sizes.push_back((uint32_t)groups[i].size());
            total += groups[i].size();
        }
        std::vector<int32_t> out(total, -1);
        std::vector<uint32_t> hist(descs.size() * SWP_NFILTERS, 0);
        prof_.lap(3);
        bool any_mounts = false;
        for (const swp_task_desc& d : descs) any_mounts = any_mounts || (d.flags >> SWP_TASK_MOUNTS_SHIFT) != 0;
        std::vector<uint32_t> att(any_mounts ? total * SWP_MAX_MOUNTS : 0, SWP_NO_VOLUME);
        try {
            if (any_mounts) ck(swp_schedule_groups_volumes(e_, descs.data(), sizes.data(), (uint32_t)descs.size(), out.data(), hist.data(), att.data()), "swp_schedule_groups");
            else ck(swp_schedule_groups(e_, descs.data(), sizes.data(), (uint32_t)descs.size(), out.data(), hist.data()), "swp_schedule_groups");
            prof_.lap(4);
        } catch (const Fail& f) {
            if (to - from > 1) {   // find the group(s) the engine cannot take: run them one by one
                for (size_t g = from; g < to; ++g) runGroups(groups, g, g + 1, decisions);
                return;
            }
            for (const Item& it : groups[from]) defer(it.first, it.second, f, decisions);
            return;
        }
        size_t off = 0;
        for (size_t g = from; g < to; ++g) {
            for (size_t i = 0; i < groups[g].size(); ++i) {
                const int32_t n = out[off + i];
                if (i + PREFETCH_FAR < groups[g].size()) prefetchBooking(groups[g][i + PREFETCH_FAR], out[off + i + PREFETCH_FAR], 1);
                if (i + PREFETCH_NEAR < groups[g].size()) prefetchBooking(groups[g][i + PREFETCH_NEAR], out[off + i + PREFETCH_NEAR], 2);
                if (n >= 0) place(groups[g][i], n, decisions, any_mounts ? &att[(off + i) * SWP_MAX_MOUNTS] : nullptr);
                else noSuitableNode(groups[g][i], &hist[(g - from) * SWP_NFILTERS], decisions);
            }
            off += groups[g].size();
        }
        pushTouched();   // groups with generic reservations: the nodes' available lists after place()'s Claim
        prof_.lap(5);
    }
    void runOneOffs(const std::vector<Item>& run, const std::vector<swp_task_desc>& descs, Decisions& decisions) {
        if (run.empty()) return;
        std::vector<int32_t> out(run.size(), -1);
        std::vector<uint32_t> hist(run.size() * SWP_NFILTERS, 0);
        std::vector<uint32_t> with_mounts;   // tasks whose attachments are read back
        for (size_t i = 0; i < run.size(); ++i)
            if (descs[i].flags >> SWP_TASK_MOUNTS_SHIFT) with_mounts.push_back((uint32_t)i);
        std::vector<uint32_t> att(with_mounts.size() * SWP_MAX_MOUNTS, SWP_NO_VOLUME);
        try {
            if (with_mounts.empty()) ck(swp_schedule_batch(e_, descs.data(), (uint32_t)descs.size(), out.data(), hist.data()), "swp_schedule_batch");
            else {   // the same in three steps: the batch is needed for its attachments
                swp_batch* b = nullptr;
                ck(swp_batch_prepare(e_, descs.data(), (uint32_t)descs.size(), &b), "swp_batch_prepare");
                int rc = swp_batch_run(e_, b);
                if (rc == SWP_OK) rc = swp_batch_fetch(e_, b, out.data(), hist.data());
                if (rc == SWP_OK) rc = swp_batch_attachments(e_, b, with_mounts.data(), (uint32_t)with_mounts.size(), att.data());
                if (rc != SWP_OK) engine_detail_ = std::string(swp_strerror(rc)) + ": " + swp_last_error(e_);
                swp_batch_free(e_, b);
                if (rc != SWP_OK) fail(rc, "swp_batch_run: " + engine_detail_);
            }
        } catch (const Fail& f) {
            for (const Item& it : run) defer(it.first, it.second, f, decisions);
            return;
        }
        prof_.lap(4);
        size_t wm = 0;
        for (size_t i = 0; i < run.size(); ++i) {
            const uint32_t* a = nullptr;
            if (wm < with_mounts.size() && with_mounts[wm] == i) a = &att[wm++ * SWP_MAX_MOUNTS];
            if (i + PREFETCH_FAR < run.size()) prefetchBooking(run[i + PREFETCH_FAR], out[i + PREFETCH_FAR], 1);
            if (i + PREFETCH_NEAR < run.size()) prefetchBooking(run[i + PREFETCH_NEAR], out[i + PREFETCH_NEAR], 2);
            if (out[i] >= 0) place(run[i], out[i], decisions, a);
            else noSuitableNode(run[i], &hist[i * SWP_NFILTERS], decisions);
        }
        pushTouched();
        prof_.lap(5);
    }
    void pushTouched() {
        for (const std::string& nid : genericTouched_) {
            auto ni = nodes_.find(nid);
            if (ni != nodes_.end()) pushGeneric(ni->second);
        }
        genericTouched_.clear();
    }
    std::set<std::string> genericTouched_;   // nodes whose available generic list place() changed during the current device call
};

}   // namespace swp

// ===================================================================================================== C boundary
struct swp_sched {
    swp::Scheduler impl;
    explicit swp_sched(swp_engine* e) : impl(e) {}
};

namespace {
thread_local std::string g_create_error;
thread_local std::string g_parse_buffer;

template <class F>
int guarded(swp_sched* s, F&& body) {
    if (s == nullptr) return SWP_EINVAL;
    try {
        return body(s->impl);
    } catch (const swp::Fail& f) {
        s->impl.last_error = f.msg;
        return f.code;
    } catch (const swp::json::ParseError& e) {
        s->impl.last_error = e.what();
        return SWP_EINVAL;
    } catch (const std::bad_alloc&) {
        s->impl.last_error = "out of memory";
        return SWP_ENOMEM;
    } catch (const std::exception& e) {
        s->impl.last_error = e.what();
        return SWP_EINVAL;
    }
}
int task_event(swp_sched* s, const char* doc, size_t len, int* flag, bool (swp::Scheduler::*handler)(const swp::json::Value&)) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (doc == nullptr) return (int)SWP_EINVAL;
        const swp::json::Value t = swp::json::parse(doc, len);
        const bool r = (impl.*handler)(t);
        if (flag != nullptr) *flag = r ? 1 : 0;
        return (int)SWP_OK;
    });
}
}   // namespace

extern "C" {

int swp_sched_create(swp_engine* engine, swp_sched** out) {
    if (engine == nullptr || out == nullptr) return SWP_EINVAL;
    try {
        *out = new swp_sched(engine);
        return SWP_OK;
    } catch (const swp::Fail& f) {
        g_create_error = f.msg;
        return f.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return SWP_ENOMEM;
    }
}
void swp_sched_destroy(swp_sched* s) { delete s; }
const char* swp_sched_last_error(swp_sched* s) { return s != nullptr ? s->impl.last_error.c_str() : g_create_error.c_str(); }

int swp_sched_create_or_update_node(swp_sched* s, const char* node_json, size_t len) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (node_json == nullptr) return (int)SWP_EINVAL;
        impl.createOrUpdateNode(swp::json::parse(node_json, len));
        return (int)SWP_OK;
    });
}
int swp_sched_delete_node(swp_sched* s, const char* node_id, size_t len) {
    return guarded(s, [&](swp::Scheduler& impl) {
        impl.deleteNode(std::string(node_id != nullptr ? node_id : "", node_id != nullptr ? len : 0));
        return (int)SWP_OK;
    });
}
int swp_sched_node_info(swp_sched* s, const char* node_id, size_t len, const char** json_out) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (node_id == nullptr || json_out == nullptr) return (int)SWP_EINVAL;
        if (!impl.nodeInfo(std::string(node_id, len), impl.scratch)) return (int)SWP_ENOTFOUND;
        *json_out = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_update_volume(swp_sched* s, const char* volume_json, size_t len) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (volume_json == nullptr) return (int)SWP_EINVAL;
        impl.updateVolume(swp::json::parse(volume_json, len));
        return (int)SWP_OK;
    });
}
int swp_sched_volume_info(swp_sched* s, const char* volume_id, size_t len, const char** json_out) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (volume_id == nullptr || json_out == nullptr) return (int)SWP_EINVAL;
        if (!impl.volumeInfo(std::string(volume_id, len), impl.scratch)) return (int)SWP_ENOTFOUND;
        *json_out = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_free_volumes(swp_sched* s, const char** json_out) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (json_out == nullptr) return (int)SWP_EINVAL;
        impl.freeVolumes(impl.scratch);
        *json_out = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_set_service(swp_sched* s, const char* service_id, size_t len, int has_version, uint64_t version) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (service_id == nullptr) return (int)SWP_EINVAL;
        impl.setService(std::string(service_id, len), has_version != 0, version);
        return (int)SWP_OK;
    });
}
int swp_sched_delete_service(swp_sched* s, const char* service_id, size_t len) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (service_id == nullptr) return (int)SWP_EINVAL;
        impl.deleteService(std::string(service_id, len));
        return (int)SWP_OK;
    });
}
int swp_sched_counts(swp_sched* s, uint64_t out[4]) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (out == nullptr) return (int)SWP_EINVAL;
        impl.counts(out);
        return (int)SWP_OK;
    });
}
int swp_sched_advance(swp_sched* s, int64_t ns) {
    return guarded(s, [&](swp::Scheduler& impl) {
        impl.advance(ns);
        return (int)SWP_OK;
    });
}
int swp_sched_create_task(swp_sched* s, const char* j, size_t n, int* f) { return task_event(s, j, n, f, &swp::Scheduler::createTask); }
int swp_sched_setup_task(swp_sched* s, const char* j, size_t n, int* f) { return task_event(s, j, n, f, &swp::Scheduler::setupTask); }
int swp_sched_update_task(swp_sched* s, const char* j, size_t n, int* f) { return task_event(s, j, n, f, &swp::Scheduler::updateTask); }
int swp_sched_delete_task(swp_sched* s, const char* j, size_t n, int* f) { return task_event(s, j, n, f, &swp::Scheduler::deleteTask); }

int swp_sched_tick(swp_sched* s, const char** decisions_json) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (decisions_json == nullptr) return (int)SWP_EINVAL;
        impl.scratch = impl.tick();
        *decisions_json = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_reject_decision(swp_sched* s, const char* task_id, size_t len, int* found) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (task_id == nullptr) return (int)SWP_EINVAL;
        const bool r = impl.rejectDecision(std::string(task_id, len));
        if (found != nullptr) *found = r ? 1 : 0;
        return (int)SWP_OK;
    });
}
int swp_sched_commit_plan(swp_sched* s, uint32_t max_changes, const char** plan_json) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (plan_json == nullptr) return (int)SWP_EINVAL;
        impl.scratch = swp::json::dump(impl.commitPlan(max_changes));
        *plan_json = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_reject_decisions(swp_sched* s, const char* ids_json, size_t len, uint32_t* n_undone) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (ids_json == nullptr || n_undone == nullptr) return (int)SWP_EINVAL;
        const swp::json::Value ids = swp::json::parse(ids_json, len);
        if (!ids.is_arr()) swp::fail(SWP_EINVAL, "a JSON array of task ids is expected");
        *n_undone = 0;
        for (const swp::json::Value& v : *ids.a)
            if (v.is_str() && impl.rejectDecision(v.s)) *n_undone += 1;
        return (int)SWP_OK;
    });
}
int swp_sched_reject_node(swp_sched* s, const char* node_id, size_t len, uint32_t* n_undone) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (node_id == nullptr || n_undone == nullptr) return (int)SWP_EINVAL;
        *n_undone = impl.rejectNode(std::string(node_id, len));
        return (int)SWP_OK;
    });
}
int swp_sched_process_preassigned(swp_sched* s, const char** decisions_json) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (decisions_json == nullptr) return (int)SWP_EINVAL;
        impl.scratch = impl.processPreassignedTasks();
        *decisions_json = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}
int swp_sched_task_desc(swp_sched* s, const char* task_json, size_t len, swp_task_desc* out) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (task_json == nullptr || out == nullptr) return (int)SWP_EINVAL;
        *out = impl.taskDesc(swp::json::parse(task_json, len));
        return (int)SWP_OK;
    });
}
int swp_sched_constraint_set(swp_sched* s, const char* exprs_json, size_t len, uint32_t* set_out) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (exprs_json == nullptr || set_out == nullptr) return (int)SWP_EINVAL;
        const swp::json::Value v = swp::json::parse(exprs_json, len);
        *set_out = impl.constraintSet(&v);
        return (int)SWP_OK;
    });
}
int swp_sched_enforce(swp_sched* s, const char* request_json, size_t len, const char** rejected_json) {
    return guarded(s, [&](swp::Scheduler& impl) {
        if (request_json == nullptr || rejected_json == nullptr) return (int)SWP_EINVAL;
        const swp::json::Value out = impl.enforce(swp::json::parse(request_json, len));
        impl.scratch = swp::json::dump(out);
        *rejected_json = impl.scratch.c_str();
        return (int)SWP_OK;
    });
}

int swp_constraint_parse(const char* exprs_json, size_t len, const char** parsed_json) {
    if (exprs_json == nullptr || parsed_json == nullptr) return SWP_EINVAL;
    try {
        const swp::json::Value v = swp::json::parse(exprs_json, len);
        if (!v.is_arr()) return SWP_EINVAL;
        std::vector<std::string> exprs;
        for (const swp::json::Value& e : *v.a) exprs.push_back(swp::json::as_str(&e));
        std::vector<swp::Expr> parsed;
        if (!swp::parse_constraints(exprs, parsed)) return SWP_EINVAL;
        swp::json::Value out = swp::json::Value::array();
        for (const swp::Expr& x : parsed) {
            swp::json::Value triple = swp::json::Value::array();
            triple.push(swp::json::Value::str(swp::encode(x.key)));
            triple.push(swp::json::Value::integer(x.op));
            triple.push(swp::json::Value::str(swp::encode(x.exp)));
            out.push(triple);
        }
        g_parse_buffer = swp::json::dump(out);
        *parsed_json = g_parse_buffer.c_str();
        return SWP_OK;
    } catch (const std::exception&) {
        return SWP_EINVAL;
    }
}
int swp_key_equal_fold(const char* a, size_t la, const char* b, size_t lb) {
    if ((a == nullptr && la) || (b == nullptr && lb)) return 0;
    try {
        return swp::equal_fold(swp::decode(std::string(a ? a : "", la)), swp::decode(std::string(b ? b : "", lb))) ? 1 : 0;
    } catch (const std::exception&) {
        return 0;
    }
}
int swp_explain(const uint32_t* hist, char* out, size_t cap) {
    if (hist == nullptr) return SWP_EINVAL;
    try {
        const std::string s = swp::explain(hist);
        if (out != nullptr && cap > 0) {
            const size_t n = std::min(cap - 1, s.size());
            std::memcpy(out, s.data(), n);
            out[n] = 0;
        }
        return (int)s.size();
    } catch (const std::exception&) {
        return SWP_ENOMEM;
    }
}
int swp_parse_ip(const char* s, size_t len, uint8_t out16[16], int* is_v4) {
    if (s == nullptr || out16 == nullptr) return 0;
    try {
        bool v4 = false;
        if (!swp::parse_ip(std::string(s, len), out16, &v4)) return 0;
        if (is_v4 != nullptr) *is_v4 = v4 ? 1 : 0;
        return 1;
    } catch (const std::exception&) {
        return 0;
    }
}

}   // extern "C"
