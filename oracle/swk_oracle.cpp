// swk_oracle.cpp — CPU ORACLE (test infrastructure; see swk_oracle.hpp header).
// Each function cites the reference lines it restates (paths under /root/reference/).
#include "swk_oracle.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace orc {

// ============================================================================
// Go stdlib pieces restated (not under /root/reference: go 1.25 stdlib)
// ============================================================================

// UTF-8 decode one rune; invalid bytes decode as U+FFFD width 1 (Go's utf8.DecodeRuneInString).
static uint32_t decode_rune(const std::string& s, size_t i, size_t* w) {
    unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && (((unsigned char)s[i + k]) & 0xC0) == 0x80; };
    if (c < 0x80) { *w = 1; return c; }
    if (c >= 0xC2 && c <= 0xDF && cont(1)) { *w = 2; return ((c & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); }
    if (c >= 0xE0 && c <= 0xEF && cont(1) && cont(2)) {
        uint32_t r = ((c & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu);
        if (r >= 0x800 && !(r >= 0xD800 && r <= 0xDFFF)) { *w = 3; return r; }
    }
    if (c >= 0xF0 && c <= 0xF4 && cont(1) && cont(2) && cont(3)) {
        uint32_t r = ((c & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu);
        if (r >= 0x10000 && r <= 0x10FFFF) { *w = 4; return r; }
    }
    *w = 1;
    return 0xFFFD;
}

// Canonical representative of a rune's simple-case-folding orbit (unicode.SimpleFold):
// the smallest rune of the orbit. Covers ASCII, Latin-1, Latin Extended-A, Greek and
// Cyrillic basic blocks plus the ASCII-touching specials (K U+212A, ſ U+017F, µ U+00B5, Å U+212B).
// Runes outside these ranges fold to themselves — a documented limit of the restatement;
// constraint expressions are ASCII(+K,+ſ) by the value regexp (constraint.go:23-26), so only
// these orbits can ever compare equal to one.
static uint32_t fold_rune(uint32_t r) {
    if (r < 0x80) return (r >= 'a' && r <= 'z') ? r - 32 : r;
    if (r == 0x212A) return 'K';
    if (r == 0x017F) return 'S';
    if (r == 0x00B5 || r == 0x03BC) return 0x039C;   // µ, μ → Μ
    if (r == 0x212B || r == 0x00E5) return 0x00C5;   // Å(angstrom), å → Å
    if (r >= 0x00E0 && r <= 0x00FE && r != 0x00F7) return r - 32;           // Latin-1 lower → upper
    if (r == 0x00FF) return 0x0178;
    if (r >= 0x0100 && r <= 0x017E) {                                        // Latin Extended-A pairs
        if (r == 0x0130 || r == 0x0131 || r == 0x0138 || r == 0x0149) return r;
        if ((r >= 0x0139 && r <= 0x0148) || (r >= 0x0179 && r <= 0x017E)) return (r & 1) ? r : r - 1;
        return (r & 1) ? r - 1 : r;
    }
    if (r >= 0x03B1 && r <= 0x03C9 && r != 0x03C2) return r - 32;           // Greek lower → upper
    if (r == 0x03C2) return 0x03A3;                                          // final sigma orbit
    if (r >= 0x0430 && r <= 0x044F) return r - 32;                           // Cyrillic
    if (r >= 0x0450 && r <= 0x045F) return r - 80;
    return r;
}

// strings.EqualFold (call sites constraint.go:90,110-188; nodeset.go:69,73)
// Two pure-ASCII strings are fold-equal iff they are equal ignoring ASCII case: the simple-folding orbits that tie an ASCII letter to
// a non-ASCII rune (K / U+212A, s / U+017F) need a non-ASCII byte on one side. Same verdicts as the rune walk below, which stays the
// path for everything else; NodeMatches compares every constraint key with six literals per node, so this is the oracle's hot spot.
static inline bool all_ascii(const char* p, size_t n) {
    unsigned char acc = 0;
    for (size_t i = 0; i < n; ++i) acc |= (unsigned char)p[i];
    return acc < 0x80;
}
static inline bool ascii_equal_fold(const char* a, const char* b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        unsigned char x = (unsigned char)a[i], y = (unsigned char)b[i];
        if (x == y) continue;
        if (x >= 'a' && x <= 'z') x -= 32;
        if (y >= 'a' && y <= 'z') y -= 32;
        if (x != y) return false;
    }
    return true;
}
static inline bool equal_fold(const std::string& a, const char* lit) {   // comparisons with the key literals: no temporary string
    const size_t n = std::strlen(lit);
    if (all_ascii(a.data(), a.size()) && all_ascii(lit, n)) return a.size() == n && ascii_equal_fold(a.data(), lit, n);
    return equal_fold(a, std::string(lit));
}
bool equal_fold(const std::string& a, const std::string& b) {
    if (all_ascii(a.data(), a.size()) && all_ascii(b.data(), b.size()))
        return a.size() == b.size() && ascii_equal_fold(a.data(), b.data(), a.size());
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
        size_t wa, wb;
        uint32_t ra = decode_rune(a, i, &wa), rb = decode_rune(b, j, &wb);
        i += wa;
        j += wb;
        if (ra == rb) continue;
        if (fold_rune(ra) != fold_rune(rb)) return false;
    }
    return i == a.size() && j == b.size();
}

// strings.TrimSpace: Unicode White_Space; ASCII set + U+0085, U+00A0 and the common wide ones.
static bool is_space_rune(uint32_t r) {
    switch (r) {
    case '\t': case '\n': case '\v': case '\f': case '\r': case ' ': case 0x85: case 0xA0:
    case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
        return true;
    }
    return r >= 0x2000 && r <= 0x200A;
}
static std::string trim_space(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e) {
        size_t w;
        uint32_t r = decode_rune(s, b, &w);
        if (!is_space_rune(r)) break;
        b += w;
    }
    while (e > b) {
        size_t k = e - 1;
        while (k > b && (((unsigned char)s[k]) & 0xC0) == 0x80) --k;
        size_t w;
        uint32_t r = decode_rune(s, k, &w);
        if (k + w != e || !is_space_rune(r)) break;
        e = k;
    }
    return s.substr(b, e - b);
}

// regexp `^(?i)[a-z_][a-z0-9\-_.]+$` and the value pattern (constraint.go:23-26).
// Under (?i) RE2 folds [a-z] to include U+212A (K) and U+017F (ſ).
static bool is_alpha_i(uint32_t r) {
    return (r >= 'a' && r <= 'z') || (r >= 'A' && r <= 'Z') || r == 0x212A || r == 0x017F;
}
static bool key_valid(const std::string& s) {
    size_t i = 0, n = 0;
    while (i < s.size()) {
        size_t w;
        uint32_t r = decode_rune(s, i, &w);
        if (r == 0xFFFD && w == 1) return false;
        bool ok = is_alpha_i(r) || r == '_';
        if (n > 0) ok = ok || (r >= '0' && r <= '9') || r == '-' || r == '.';
        if (!ok) return false;
        i += w;
        ++n;
    }
    return n >= 2;
}
static bool value_valid(const std::string& s) {
    size_t i = 0, n = 0;
    while (i < s.size()) {
        size_t w;
        uint32_t r = decode_rune(s, i, &w);
        if (r == 0xFFFD && w == 1) return false;
        bool ok = is_alpha_i(r) || (r >= '0' && r <= '9') || r == ':' || r == '-' || r == '_' ||
                  r == '\t' || r == '\n' || r == '\f' || r == '\r' || r == ' ' ||   // \s
                  r == '.' || r == '*' || r == '(' || r == ')' || r == '?' || r == '+' || r == '[' || r == ']' ||
                  r == '\\' || r == '^' || r == '$' || r == '|' || r == '/';
        if (!ok) return false;
        i += w;
        ++n;
    }
    return n >= 1;
}

// net.ParseIP / net.ParseCIDR (call sites constraint.go:128-146). 16-byte form; v4 as v4-in-v6.
struct IPAddr { bool ok = false; bool is4 = false; unsigned char b[16] = {0}; };
static bool parse_ipv4_fields(const std::string& s, unsigned char out[4]) {
    size_t i = 0;
    for (int f = 0; f < 4; ++f) {
        if (i >= s.size() || s[i] < '0' || s[i] > '9') return false;
        size_t start = i;
        int v = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') {
            v = v * 10 + (s[i] - '0');
            if (v > 255) return false;
            ++i;
        }
        if (i - start > 1 && s[start] == '0') return false;   // leading zeros rejected (netip.ParseAddr)
        out[f] = (unsigned char)v;
        if (f < 3) {
            if (i >= s.size() || s[i] != '.') return false;
            ++i;
        }
    }
    return i == s.size();
}
static IPAddr parse_ip(const std::string& s) {
    IPAddr ip;
    if (s.find(':') == std::string::npos) {
        unsigned char v4[4];
        if (!parse_ipv4_fields(s, v4)) return ip;
        ip.ok = true;
        ip.is4 = true;
        ip.b[10] = 0xff;
        ip.b[11] = 0xff;
        std::memcpy(ip.b + 12, v4, 4);
        return ip;
    }
    if (s.find('%') != std::string::npos) return ip;   // zones are rejected by net.ParseIP
    // IPv6
    int groups[8];
    int ng = 0, ellipsis = -1;
    size_t i = 0;
    if (s.size() >= 2 && s[0] == ':' && s[1] == ':') {
        ellipsis = 0;
        i = 2;
        if (i == s.size()) { ip.ok = true; return ip; }
    } else if (!s.empty() && s[0] == ':') return ip;
    while (i < s.size()) {
        // hex group
        size_t start = i;
        unsigned v = 0;
        while (i < s.size() && std::isxdigit((unsigned char)s[i]) && i - start < 4) {
            char c = s[i];
            v = v * 16 + (c <= '9' ? unsigned(c - '0') : unsigned((c | 0x20) - 'a' + 10));
            ++i;
        }
        if (i == start) return ip;
        if (i < s.size() && s[i] == '.') {
            // embedded IPv4 in the last 32 bits
            unsigned char v4[4];
            if (!parse_ipv4_fields(s.substr(start), v4)) return ip;
            if (ng > 6) return ip;
            groups[ng++] = (v4[0] << 8) | v4[1];
            groups[ng++] = (v4[2] << 8) | v4[3];
            i = s.size();
            break;
        }
        if (ng >= 8) return ip;
        groups[ng++] = int(v);
        if (i == s.size()) break;
        if (s[i] != ':') return ip;
        ++i;
        if (i == s.size()) return ip;   // trailing single colon
        if (s[i] == ':') {
            if (ellipsis >= 0) return ip;
            ellipsis = ng;
            ++i;
            if (i == s.size()) break;
        }
    }
    if (ng < 8) {
        if (ellipsis < 0) return ip;
        int fill = 8 - ng;
        for (int k = ng - 1; k >= ellipsis; --k) groups[k + fill] = groups[k];
        for (int k = ellipsis; k < ellipsis + fill; ++k) groups[k] = 0;
    } else if (ellipsis >= 0) return ip;
    for (int k = 0; k < 8; ++k) {
        ip.b[2 * k] = (unsigned char)(groups[k] >> 8);
        ip.b[2 * k + 1] = (unsigned char)(groups[k] & 0xff);
    }
    ip.ok = true;
    return ip;
}
static bool ip_is_v4mapped(const IPAddr& a) {
    for (int i = 0; i < 10; ++i)
        if (a.b[i]) return false;
    return a.b[10] == 0xff && a.b[11] == 0xff;
}
// IP.Equal: an IPv4 address and its IPv4-in-IPv6 form are the same.
static bool ip_equal(const IPAddr& a, const IPAddr& b) {
    if (!a.ok || !b.ok) return false;
    return std::memcmp(a.b, b.b, 16) == 0;
}
struct IPNet { bool ok = false; bool is4 = false; unsigned char ip[16]; unsigned char mask[16]; };
static IPNet parse_cidr(const std::string& s) {
    IPNet n;
    size_t slash = s.find('/');
    if (slash == std::string::npos) return n;
    std::string addr = s.substr(0, slash), m = s.substr(slash + 1);
    IPAddr ip = parse_ip(addr);
    if (!ip.ok) return n;
    bool is4 = ip.is4;
    int bits = is4 ? 32 : 128;
    if (m.empty() || m.size() > 3) return n;
    int len = 0;
    for (char c : m) {
        if (c < '0' || c > '9') return n;
        len = len * 10 + (c - '0');
    }
    if (len > bits) return n;
    n.ok = true;
    n.is4 = is4;
    std::memset(n.mask, 0, 16);
    int off = is4 ? 12 : 0;
    if (is4) std::memset(n.mask, 0xff, 12);
    for (int k = 0; k < len; ++k) n.mask[off + k / 8] |= (unsigned char)(0x80 >> (k % 8));
    for (int k = 0; k < 16; ++k) n.ip[k] = ip.b[k] & n.mask[k];
    return n;
}
// IPNet.Contains: a v4 network only contains v4 (or v4-mapped) addresses; a v6 network only v6.
static bool cidr_contains(const IPNet& n, const IPAddr& ip) {
    if (!n.ok || !ip.ok) return false;
    bool ip4 = ip.is4 || ip_is_v4mapped(ip);
    if (n.is4 != ip4) return false;
    for (int k = 0; k < 16; ++k)
        if ((ip.b[k] & n.mask[k]) != n.ip[k]) return false;
    return true;
}

// ============================================================================
// manager/constraint/constraint.go
// ============================================================================

// Parse, constraint.go:40-81. Operators are tried in order "==" then "!=" (constraint.go:29).
bool constraint_parse(const std::vector<std::string>& env, std::vector<Constraint>* out, std::string* err) {
    static const char* operators[2] = {"==", "!="};
    out->clear();
    for (const std::string& e : env) {
        bool found = false;
        for (int i = 0; i < 2; ++i) {
            size_t at = e.find(operators[i]);
            if (at == std::string::npos) continue;
            // strings.SplitN(e, op, 2)
            std::string part0 = trim_space(e.substr(0, at));
            if (!key_valid(part0)) {
                if (err) *err = "key '" + part0 + "' is invalid";
                return false;
            }
            std::string part1 = trim_space(e.substr(at + 2));
            if (!value_valid(part1)) {
                if (err) *err = "value '" + part1 + "' is invalid";
                return false;
            }
            Constraint c;
            c.key = part0;
            c.op = i;
            c.exp = part1;
            out->push_back(std::move(c));
            found = true;
            break;
        }
        if (!found) {
            if (err) *err = "constraint expected one operator from ==, !=";
            return false;
        }
    }
    return true;
}

// Constraint.Match, constraint.go:84-105 (single target; the reference only ever passes one).
bool constraint_match(const Constraint& c, const std::string& what) {
    bool match = equal_fold(c.exp, what);
    return c.op == 0 ? match : !match;
}

static const char kNodeLabelPrefix[] = "node.labels.";      // constraint.go:15
static const char kEngineLabelPrefix[] = "engine.labels.";  // constraint.go:17

static bool has_prefix_fold(const std::string& s, const char* prefix) {
    size_t n = std::strlen(prefix);
    // len(s) > len(prefix) && EqualFold(s[:len(prefix)], prefix)   (byte slice, as in Go)
    if (s.size() <= n) return false;
    if (all_ascii(s.data(), n) && all_ascii(prefix, n)) return ascii_equal_fold(s.data(), prefix, n);
    return equal_fold(s.substr(0, n), prefix);
}

// NodeMatches, constraint.go:107-207. The switch over the key (`case strings.EqualFold(constraint.key, "node.id")`, … in this
// order, then the two label prefixes) depends on the key alone: constraint_kind() evaluates it once per constraint.
enum { CK_ID = 0, CK_HOSTNAME, CK_IP, CK_ROLE, CK_OS, CK_ARCH, CK_NODE_LABEL, CK_ENGINE_LABEL, CK_OTHER };
static int constraint_kind(const Constraint& c) {
    if (c.kind >= 0) return c.kind;
    int k = CK_OTHER;
    if (equal_fold(c.key, "node.id")) k = CK_ID;
    else if (equal_fold(c.key, "node.hostname")) k = CK_HOSTNAME;
    else if (equal_fold(c.key, "node.ip")) k = CK_IP;
    else if (equal_fold(c.key, "node.role")) k = CK_ROLE;
    else if (equal_fold(c.key, "node.platform.os")) k = CK_OS;
    else if (equal_fold(c.key, "node.platform.arch")) k = CK_ARCH;
    else if (has_prefix_fold(c.key, kNodeLabelPrefix)) {
        k = CK_NODE_LABEL;
        c.label = c.key.substr(sizeof(kNodeLabelPrefix) - 1);   // label name is case sensitive
    } else if (has_prefix_fold(c.key, kEngineLabelPrefix)) {
        k = CK_ENGINE_LABEL;
        c.label = c.key.substr(sizeof(kEngineLabelPrefix) - 1);
    }
    c.kind = k;
    return k;
}
bool node_matches(const std::vector<Constraint>& cs, const Node& n) {
    static const std::string kEmpty;
    for (const Constraint& c : cs) {
        switch (constraint_kind(c)) {
        case CK_ID:
            if (!constraint_match(c, n.id)) return false;
            break;
        case CK_HOSTNAME:
            if (!n.has_description) {
                if (!constraint_match(c, kEmpty)) return false;
                continue;
            }
            if (!constraint_match(c, n.hostname)) return false;
            break;
        case CK_IP: {
            IPAddr node_ip = parse_ip(n.addr);
            IPAddr ip = parse_ip(c.exp);
            if (ip.ok) {
                bool eq = ip_equal(ip, node_ip);
                if ((eq && c.op != 0) || (!eq && c.op == 0)) return false;
                continue;
            }
            IPNet subnet = parse_cidr(c.exp);
            if (subnet.ok) {
                bool within = cidr_contains(subnet, node_ip);
                if ((within && c.op != 0) || (!within && c.op == 0)) return false;
                continue;
            }
            return false;   // malformed address/network: both operators fail
        }
        case CK_ROLE:
            if (!constraint_match(c, n.role == NodeRoleManager ? "MANAGER" : (n.role == NodeRoleWorker ? "WORKER" : std::to_string(n.role))))
                return false;
            break;
        case CK_OS:
            if (!n.has_description || !n.has_platform) {
                if (!constraint_match(c, kEmpty)) return false;
                continue;
            }
            if (!constraint_match(c, n.platform.os)) return false;
            break;
        case CK_ARCH:
            if (!n.has_description || !n.has_platform) {
                if (!constraint_match(c, kEmpty)) return false;
                continue;
            }
            if (!constraint_match(c, n.platform.arch)) return false;
            break;
        case CK_NODE_LABEL: {
            if (n.labels_nil) {
                if (!constraint_match(c, kEmpty)) return false;
                continue;
            }
            auto it = n.labels.find(c.label);
            if (!constraint_match(c, it == n.labels.end() ? kEmpty : it->second)) return false;
            break;
        }
        case CK_ENGINE_LABEL: {
            if (!n.has_description || !n.has_engine || n.engine_labels_nil) {
                if (!constraint_match(c, kEmpty)) return false;
                continue;
            }
            auto it = n.engine_labels.find(c.label);
            if (!constraint_match(c, it == n.engine_labels.end() ? kEmpty : it->second)) return false;
            break;
        }
        default:
            return false;   // key doesn't match predefined syntax
        }
    }
    return true;
}

// ============================================================================
// api/genericresource
// ============================================================================

static std::vector<size_t> get_resource_idx(const std::string& kind, const GenericList& rs) {   // helpers.go:43-55
    std::vector<size_t> out;
    for (size_t i = 0; i < rs.size(); ++i)
        if (rs[i].kind == kind) out.push_back(i);
    return out;
}

// HasEnough, validate.go:24-52. *err mirrors the non-nil error return.
bool generic_has_enough(const Resources& node_avail, const GenericResource& task_res, bool* err) {
    *err = false;
    if (task_res.named) { *err = true; return false; }   // "task should only hold Discrete type"
    if (node_avail.generic_nil) return false;
    auto nrs = get_resource_idx(task_res.kind, node_avail.generic);
    if (nrs.empty()) return false;
    const GenericResource& first = node_avail.generic[nrs[0]];
    if (!first.named) {
        if (task_res.ivalue > first.ivalue) return false;
    } else {
        if (task_res.ivalue > int64_t(nrs.size())) return false;
    }
    return true;
}

// remove(), helpers.go:88-111
static bool generic_remove(GenericResource& na, const GenericResource& r) {
    if (!r.named) {
        if (na.named) return false;
        na.ivalue -= r.ivalue;
        return na.ivalue <= 0;
    }
    if (!na.named) return false;
    return r.svalue == na.svalue;
}

// ConsumeNodeResources, helpers.go:58-85
void generic_consume(GenericList* node_avail, const GenericList& res) {
    if (!node_avail) return;
    size_t w = 0;
    for (size_t i = 0; i < node_avail->size(); ++i) {
        GenericResource& na = (*node_avail)[i];
        bool removed = false;
        for (const GenericResource& r : res) {
            if (na.kind != r.kind) continue;
            if (generic_remove(na, r)) { removed = true; break; }
        }
        if (removed) continue;
        if (w != i) (*node_avail)[w] = na;
        ++w;
    }
    node_avail->resize(w);
}

// selectNodeResources, resource_management.go:41-72. Returns false on the error return.
static bool select_node_resources(const GenericList& node_res, const GenericResource& tr, GenericList* out) {
    GenericList nrs;
    for (const GenericResource& res : node_res) {
        if (res.kind != tr.kind) continue;
        if (!res.named) {
            if (res.ivalue >= tr.ivalue && tr.ivalue != 0) {
                GenericResource d;
                d.kind = tr.kind;
                d.ivalue = tr.ivalue;
                nrs.push_back(d);
            }
            *out = nrs;
            return true;
        }
        nrs.push_back(res);
        if (int64_t(nrs.size()) == tr.ivalue) { *out = nrs; return true; }
    }
    if (nrs.empty()) return false;
    *out = nrs;
    return true;
}

// Claim, resource_management.go:11-39 (error returns leave everything untouched).
void generic_claim(GenericList* node_avail, GenericList* task_assigned, const GenericList& reservations) {
    GenericList selected;
    for (const GenericResource& res : reservations) {
        if (res.named) return;
        GenericList nrs;
        if (!select_node_resources(*node_avail, res, &nrs)) return;
        selected.insert(selected.end(), nrs.begin(), nrs.end());
    }
    task_assigned->insert(task_assigned->end(), selected.begin(), selected.end());
    generic_consume(node_avail, selected);
}

// sanitizeResource, resource_management.go:155-203
static bool sanitize_resource(const GenericList& node_res, const GenericResource& res, GenericList* replacement) {
    replacement->clear();
    auto idx = get_resource_idx(res.kind, node_res);
    auto fill = [&]() { for (size_t i : idx) replacement->push_back(node_res[i]); };
    if (!res.named) {
        if (idx.size() != 1) { fill(); return false; }
        const GenericResource& nr = node_res[idx[0]];
        if (nr.named) { fill(); return false; }
        if (res.ivalue > nr.ivalue) { fill(); return false; }
        return true;
    }
    if (idx.empty()) return false;
    for (size_t i : idx) {
        if (!node_res[i].named) { fill(); return false; }
        if (res.svalue == node_res[i].svalue) return true;
    }
    return false;   // removed
}

// reclaimResources, resource_management.go:87-117
void generic_reclaim_resources(GenericList* node_avail, const GenericList& task_assigned) {
    for (const GenericResource& res : task_assigned) {
        if (!res.named) {
            auto nrs = get_resource_idx(res.kind, *node_avail);
            if (nrs.empty()) node_avail->push_back(res);
            if (nrs.size() != 1) continue;
            GenericResource& nr = (*node_avail)[nrs[0]];
            if (nr.named) continue;
            nr.ivalue += res.ivalue;
        } else {
            node_avail->push_back(res);
        }
    }
}

// sanitize, resource_management.go:119-153
void generic_sanitize(const GenericList& node_res, GenericList* node_avail) {
    GenericList sanitized;
    std::map<std::string, bool> kind_sanitized;
    size_t w = 0;
    for (size_t i = 0; i < node_avail->size(); ++i) {
        GenericResource na = (*node_avail)[i];
        GenericList nrs;
        if (!sanitize_resource(node_res, na, &nrs)) {
            if (kind_sanitized.count(na.kind)) continue;
            kind_sanitized[na.kind] = true;
            sanitized.insert(sanitized.end(), nrs.begin(), nrs.end());
            continue;
        }
        (*node_avail)[w++] = na;
    }
    node_avail->resize(w);
    node_avail->insert(node_avail->end(), sanitized.begin(), sanitized.end());
}

// Reclaim = reclaimResources + sanitize, resource_management.go:75-85
void generic_reclaim(GenericList* node_avail, const GenericList& task_assigned, const GenericList& node_res) {
    generic_reclaim_resources(node_avail, task_assigned);
    generic_sanitize(node_res, node_avail);
}

// ============================================================================
// manager/scheduler/nodeinfo.go
// ============================================================================

static const int64_t kMonitorFailures = 5LL * 60 * 1'000'000'000LL;   // scheduler.go:19
static const int64_t kMaxFailures = 5;                                 // scheduler.go:23

// the process-wide service-id table behind SvcCounts (ids are never forgotten: a handful of bytes per service ever seen)
namespace {
struct ServiceTable {
    std::unordered_map<std::string, uint32_t> key;
    std::vector<std::string> name;
};
ServiceTable& service_table() {
    static ServiceTable t;
    return t;
}
}  // namespace
uint32_t service_key(const std::string& service_id) {
    ServiceTable& t = service_table();
    auto it = t.key.find(service_id);
    if (it != t.key.end()) return it->second;
    const uint32_t k = uint32_t(t.name.size());
    t.name.push_back(service_id);
    t.key.emplace(service_id, k);
    return k;
}
const std::string& service_of_key(uint32_t key) { return service_table().name[key]; }
static uint32_t task_service_key(const Task& t) {
    if (t.service_key_cache == 0xFFFFFFFFu) t.service_key_cache = service_key(t.service_id);
    return t.service_key_cache;
}

int64_t NodeInfo::svc_count(const std::string& s) const {
    if (!by_service) return 0;   // nil map read
    return by_service->get(service_key(s));
}
int64_t NodeInfo::svc_count_key(uint32_t key) const {
    if (!by_service) return 0;
    return by_service->get(key);
}

// taskReservations, nodeinfo.go:156-161
static Resources task_reservations(const Task& t) {
    return t.has_reservations ? t.reservations : Resources();
}

// newNodeInfo, nodeinfo.go:46-62
NodeInfo new_node_info(const NodePtr& n, const std::vector<TaskPtr>& tasks, const Resources& avail, int64_t now) {
    NodeInfo ni;
    ni.node = n;
    ni.tasks = std::make_shared<std::map<std::string, TaskPtr>>();
    ni.by_service = std::make_shared<SvcCounts>();
    ni.available = std::make_shared<Resources>(avail);
    ni.used_ports = std::make_shared<std::map<HostPortSpec, int>>();
    ni.recent_failures = std::make_shared<std::map<VersionedService, std::vector<int64_t>>>();
    ni.last_cleanup = now;
    for (const TaskPtr& t : tasks) ni.add_task(t);
    return ni;
}

// removeTask, nodeinfo.go:66-104
bool NodeInfo::remove_task(const Task& t) {
    auto it = tasks->find(t.id);
    if (it == tasks->end()) return false;
    TaskPtr old_task = it->second;
    tasks->erase(it);
    if (old_task->desired_state <= TaskStateCompleted) {
        active_tasks_count--;
        by_service->ref(task_service_key(t))--;
    }
    if (t.has_endpoint) {
        for (const PortConfig& p : t.ports)
            if (p.publish_mode == PublishModeHost && p.published_port != 0)
                used_ports->erase(HostPortSpec{p.protocol, p.published_port});
    }
    Resources r = task_reservations(t);
    available->memory_bytes += r.memory_bytes;
    available->nano_cpus += r.nano_cpus;
    if (!node || !node->has_description || !node->has_resources || node->resources.generic_nil) return true;
    generic_reclaim(&available->generic, t.assigned_generic, node->resources.generic);
    available->generic_nil = false;
    return true;
}

// addTask, nodeinfo.go:108-154
bool NodeInfo::add_task(const TaskPtr& t) {
    auto it = tasks->find(t->id);
    if (it != tasks->end()) {
        TaskPtr old_task = it->second;
        if (t->desired_state <= TaskStateCompleted && old_task->desired_state > TaskStateCompleted) {
            it->second = t;
            active_tasks_count++;
            by_service->ref(task_service_key(*t))++;
            return true;
        } else if (t->desired_state > TaskStateCompleted && old_task->desired_state <= TaskStateCompleted) {
            it->second = t;
            active_tasks_count--;
            by_service->ref(task_service_key(*t))--;
            return true;
        }
        return false;
    }
    (*tasks)[t->id] = t;
    Resources r = task_reservations(*t);
    available->memory_bytes -= r.memory_bytes;
    available->nano_cpus -= r.nano_cpus;
    t->assigned_generic.clear();
    generic_claim(&available->generic, &t->assigned_generic, r.generic);
    if (t->has_endpoint) {
        for (const PortConfig& p : t->ports)
            if (p.publish_mode == PublishModeHost && p.published_port != 0)
                (*used_ports)[HostPortSpec{p.protocol, p.published_port}] = 1;
    }
    if (t->desired_state <= TaskStateCompleted) {
        active_tasks_count++;
        by_service->ref(task_service_key(*t))++;
    }
    return true;
}

// cleanupFailures, nodeinfo.go:163-174
void NodeInfo::cleanup_failures(int64_t now) {
    for (auto it = recent_failures->begin(); it != recent_failures->end();) {
        bool keep = false;
        for (int64_t ts : it->second)
            if (now - ts < kMonitorFailures) { keep = true; break; }
        if (keep) ++it;
        else it = recent_failures->erase(it);
    }
    last_cleanup = now;
}

// taskFailed, nodeinfo.go:177-202
void NodeInfo::task_failed(int64_t now, const Task& t) {
    if (now - last_cleanup >= kMonitorFailures) cleanup_failures(now);
    VersionedService vs{t.service_id, t.has_spec_version ? t.spec_version : 0};
    std::vector<int64_t>& list = (*recent_failures)[vs];
    size_t expired = 0;
    for (int64_t ts : list) {
        if (now - ts < kMonitorFailures) break;
        expired++;
    }
    list.erase(list.begin(), list.begin() + long(expired));
    list.push_back(now);
}

// countRecentFailures, nodeinfo.go:206-221
int64_t NodeInfo::count_recent_failures(int64_t now, const Task& t) const {
    if (!recent_failures || recent_failures->empty()) return 0;
    return count_recent_failures_key(now, VersionedService{t.service_id, t.has_spec_version ? t.spec_version : 0});
}
int64_t NodeInfo::count_recent_failures_key(int64_t now, const VersionedService& vs) const {
    if (!recent_failures || recent_failures->empty()) return 0;   // Go: lookups in an empty map return at once
    auto it = recent_failures->find(vs);
    if (it == recent_failures->end()) return 0;
    const std::vector<int64_t>& list = it->second;
    int64_t count = int64_t(list.size());
    for (int64_t i = count - 1; i >= 0; --i) {
        if (now - list[size_t(i)] > kMonitorFailures) {
            count -= i + 1;
            break;
        }
    }
    return count;
}

// ============================================================================
// manager/scheduler/filter.go + pipeline.go
// ============================================================================

// referencesVolumePlugin, filter.go:109-116
static bool references_volume_plugin(const Mount& m) {
    return m.type == MountTypeVolume && m.has_driver_config && !m.driver_name.empty() && m.driver_name != "local";
}

// pluginExistsOnNode, filter.go:179-202 → (typeFound-or-true, exists)
static std::pair<bool, bool> plugin_exists(const std::string& type, const std::string& name, const std::vector<Plugin>& plugins) {
    bool type_found = false;
    for (const Plugin& np : plugins) {
        if (type != np.type) continue;
        type_found = true;
        if (name == np.name) return {true, true};
        if (np.name.size() >= name.size() && np.name.compare(0, name.size(), name) == 0 && np.name.substr(name.size()) == ":latest")
            return {true, true};
    }
    return {type_found, false};
}

// platformEqual, filter.go:283-306
static bool platform_equal(Platform img, Platform node) {
    if (img.arch == "x86_64") img.arch = "amd64";
    if (node.arch == "x86_64") node.arch = "amd64";
    if (img.arch == "aarch64") img.arch = "arm64";
    if (node.arch == "aarch64") node.arch = "arm64";
    return (img.arch.empty() || img.arch == node.arch) && (img.os.empty() || img.os == node.os);
}

// Filter.SetTask for each filter in checklist order (pipeline.go:9-20), pipeline.go:76-81.
void Pipeline::set_task(const Task* task) {
    t = task;
    for (auto& e : checklist) { e.enabled = false; e.failure_count = 0; }
    // ReadyFilter.SetTask filter.go:36-38
    checklist[F_READY].enabled = true;
    // ResourceFilter.SetTask filter.go:61-74
    if (t->has_reservations) {
        const Resources& r = t->reservations;
        if (!(r.nano_cpus == 0 && r.memory_bytes == 0 && r.generic.empty())) checklist[F_RESOURCE].enabled = true;
    }
    // PluginFilter.SetTask filter.go:119-131
    if (!t->networks.empty() || t->has_log_driver) checklist[F_PLUGIN].enabled = true;
    else if (t->has_container && std::any_of(t->mounts.begin(), t->mounts.end(), references_volume_plugin))
        checklist[F_PLUGIN].enabled = true;
    // ConstraintFilter.SetTask filter.go:218-232 (parse failure ⇒ filter disabled)
    constraints.clear();
    if (t->has_placement && !t->constraints.empty()) {
        if (constraint_parse(t->constraints, &constraints, nullptr)) checklist[F_CONSTRAINT].enabled = true;
        else constraints.clear();
    }
    // PlatformFilter.SetTask filter.go:253-263
    if (t->has_placement && !t->platforms.empty()) checklist[F_PLATFORM].enabled = true;
    // HostPortFilter.SetTask filter.go:322-333
    if (t->has_endpoint)
        for (const PortConfig& p : t->ports)
            if (p.publish_mode == PublishModeHost && p.published_port != 0) { checklist[F_HOSTPORT].enabled = true; break; }
    // MaxReplicasFilter.SetTask filter.go:363-370
    if (t->has_placement && t->max_replicas > 0) checklist[F_MAXREPLICAS].enabled = true;
    // VolumesFilter.SetTask filter.go:392-422 (the filter Run appends, scheduler.go:132): enabled iff the task has a MountTypeCluster mount
    if (vs && t->has_container)
        for (const Mount& m : t->mounts)
            if (m.type == MountTypeCluster) { checklist[F_VOLUMES].enabled = true; break; }
}

bool Pipeline::check(int f, const NodeInfo& n) const {
    switch (f) {
    case F_READY:   // filter.go:41-44
        return n.node->state == NodeStatusReady && n.node->availability == NodeAvailabilityActive;
    case F_RESOURCE: {   // filter.go:77-94
        const Resources& res = t->reservations;
        if (res.nano_cpus > n.available->nano_cpus) return false;
        if (res.memory_bytes > n.available->memory_bytes) return false;
        for (const GenericResource& v : res.generic) {
            bool err;
            bool enough = generic_has_enough(*n.available, v, &err);
            if (err || !enough) return false;
        }
        return true;
    }
    case F_PLUGIN: {   // filter.go:135-176
        if (!n.node->has_description || !n.node->has_engine) return true;
        const std::vector<Plugin>& np = n.node->plugins;
        if (t->has_container)
            for (const Mount& m : t->mounts)
                if (references_volume_plugin(m) && !plugin_exists("Volume", m.driver_name, np).second) return false;
        for (const NetworkAttachment& tn : t->networks)
            if (tn.has_network && tn.has_driver_state && !tn.driver_name.empty())
                if (!plugin_exists("Network", tn.driver_name, np).second) return false;
        if (t->has_log_driver && t->log_driver != "none" && !t->log_driver.empty()) {
            auto r = plugin_exists("Log", t->log_driver, np);
            if (!r.second && r.first) return false;
        }
        return true;
    }
    case F_CONSTRAINT:   // filter.go:235-237
        return node_matches(constraints, *n.node);
    case F_PLATFORM: {   // filter.go:266-281
        if (t->platforms.empty()) return true;
        if (n.node->has_description && n.node->has_platform)
            for (const Platform& p : t->platforms)
                if (platform_equal(p, n.node->platform)) return true;
        return false;
    }
    case F_HOSTPORT:   // filter.go:336-347
        for (const PortConfig& p : t->ports)
            if (p.publish_mode == PublishModeHost && p.published_port != 0)
                if (n.used_ports && n.used_ports->count(HostPortSpec{p.protocol, p.published_port})) return false;
        return true;
    case F_MAXREPLICAS:   // filter.go:373-375
        return uint64_t(n.svc_count_key(task_service_key(*t))) < t->max_replicas;
    case F_VOLUMES:   // filter.go:424-432: passes when ANY requested cluster mount has an available volume on the node
        for (const Mount& m : t->mounts)
            if (m.type == MountTypeCluster && !vs->is_available_on_node(m, n).empty()) return true;
        return false;
    }
    return true;
}

// Process, pipeline.go:56-68
bool Pipeline::process(const NodeInfo& n) {
    ++process_calls;
    for (int i = 0; i < F_COUNT; ++i) {
        if (checklist[i].enabled && !check(i, n)) {
            checklist[i].failure_count++;
            return false;
        }
    }
    for (auto& e : checklist) e.failure_count = 0;
    return true;
}

// Filter.Explain strings, filter.go:47-52,97-102,205-210,240-245,309-314,350-355,378-380
std::string filter_explain(int f, int64_t nodes) {
    auto plural = [&](const char* one, const char* many_fmt_suffix, const char* prefix_many) {
        (void)prefix_many;
        return nodes == 1 ? std::string(one) : std::to_string(nodes) + many_fmt_suffix;
    };
    switch (f) {
    case F_READY: return plural("1 node not available for new tasks", " nodes not available for new tasks", "");
    case F_RESOURCE: return nodes == 1 ? "insufficient resources on 1 node" : "insufficient resources on " + std::to_string(nodes) + " nodes";
    case F_PLUGIN: return nodes == 1 ? "missing plugin on 1 node" : "missing plugin on " + std::to_string(nodes) + " nodes";
    case F_CONSTRAINT: return nodes == 1 ? "scheduling constraints not satisfied on 1 node" : "scheduling constraints not satisfied on " + std::to_string(nodes) + " nodes";
    case F_PLATFORM: return nodes == 1 ? "unsupported platform on 1 node" : "unsupported platform on " + std::to_string(nodes) + " nodes";
    case F_HOSTPORT: return nodes == 1 ? "host-mode port already in use on 1 node" : "host-mode port already in use on " + std::to_string(nodes) + " nodes";
    case F_MAXREPLICAS: return "max replicas per node limit exceed";
    case F_VOLUMES: return nodes == 1 ? "cannot fulfill requested CSI volume mounts on 1 node" : "cannot fulfill requested CSI volume mounts on " + std::to_string(nodes) + " nodes";   // filter.go:434-441
    }
    return "";
}

// Explain, pipeline.go:84-103. sort.Sort(sort.Reverse(byFailures)) on ≤12 elements is Go's
// insertion sort: for i=1..n-1 { for j=i; j>0 && less(j, j-1); j-- { swap } } with
// less(a,b) = count[b] < count[a]  ⇒ stable, descending.
std::string Pipeline::explain() const {
    int order[F_COUNT];
    for (int i = 0; i < F_COUNT; ++i) order[i] = i;
    for (int i = 1; i < F_COUNT; ++i)
        for (int j = i; j > 0 && checklist[order[j - 1]].failure_count < checklist[order[j]].failure_count; --j)
            std::swap(order[j], order[j - 1]);
    std::string out;
    for (int k = 0; k < F_COUNT; ++k) {
        const Entry& e = checklist[order[k]];
        if (e.failure_count > 0) {
            if (!out.empty()) out += "; ";
            out += filter_explain(order[k], e.failure_count);
        }
    }
    return out;
}

// ============================================================================
// topology.go:23-47, volumes.go:19-316
// ============================================================================

// IsInTopology: `top` (a node's topology for one plugin; nil = has_top false) lies within `accessible` when one of its topologies has
// every (subdomain, segment) pair in `top` as well. Anything missing = it fits.
bool is_in_topology(bool has_top, const std::map<std::string, std::string>& top, const std::vector<Topology>& accessible) {
    if (!has_top || accessible.empty()) return true;
    for (const Topology& topology : accessible) {
        bool all = true;
        for (const auto& kv : topology.segments) {
            auto it = top.find(kv.first);
            if ((it == top.end() ? std::string() : it->second) != kv.second) { all = false; break; }   // (a missing map key reads as "")
        }
        if (all) return true;
    }
    return false;
}

void VolumeSet::add_or_update(const VolumePtr& v) {
    auto it = volumes_.find(v->id);
    if (it == volumes_.end()) {
        Info info;
        info.volume = v;
        info.store = v;
        info.order = next_order_++;
        volumes_.emplace(v->id, std::move(info));
    } else it->second.store = v;   // (this harness has no store: the event's object is what store.GetVolume would return from now on)
    // else: volumes.go:68-72 says `info.volume = v` — on `info`, a COPY of the map's value (vs.volumes is map[string]volumeInfo, structs
    // by value): the assignment is lost, the set keeps the volume object of the FIRST call, and that is what checkVolume reads from then
    // on (availability, access mode, driver, accessible topology). Restated as it behaves, not as its comment intends.
    std::vector<std::string>& set = by_group_[v->group];
    if (std::find(set.begin(), set.end(), v->id) == set.end()) set.push_back(v->id);
    by_name_[v->name] = v->id;
}

void VolumeSet::remove(const std::string& id) {
    auto it = volumes_.find(id);
    if (it == volumes_.end()) return;
    std::vector<std::string>& set = by_group_[it->second.volume->group];
    set.erase(std::remove(set.begin(), set.end(), id), set.end());
    by_name_.erase(it->second.volume->name);
    volumes_.erase(it);
}

bool VolumeSet::choose_task_volumes(const Task& task, const NodeInfo& node, std::vector<VolumeAttachment>* out, std::string* err, std::vector<VolumeAttachment>* prefix) {
    out->clear();
    if (prefix) prefix->clear();
    if (!task.has_container) return true;
    bool ok = true;
    for (const Mount& m : task.mounts) {
        if (m.type != MountTypeCluster) continue;
        const std::string candidate = is_available_on_node(m, node);
        if (candidate.empty()) {
            if (err) *err = "cannot find volume to satisfy mount with source " + m.source;
            ok = false;
            break;
        }
        reserve(candidate, task.id, node.node->id, m.read_only);   // (so that the task's next mount sees this one; released below)
        out->push_back(VolumeAttachment{candidate, m.source, m.target});
    }
    for (const VolumeAttachment& va : *out) release(va.id, task.id);   // the deferred release: the caller reserves for good
    if (!ok) {
        if (prefix) *prefix = *out;   // (harness only: what had been chosen in front of the failing mount — the engine double replays it)
        out->clear();   // (the reference returns nil attachments with the error)
    }
    return ok;
}

void VolumeSet::reserve_task_volumes(const Task& task) {
    for (const VolumeAttachment& va : task.volumes)
        for (const Mount& m : task.mounts)
            if (m.source == va.source && m.target == va.target) reserve(va.id, task.id, task.node_id, m.read_only);
}

void VolumeSet::reserve(const std::string& volume_id, const std::string& task_id, const std::string& node_id, bool read_only) {
    auto it = volumes_.find(volume_id);
    if (it == volumes_.end()) return;
    it->second.tasks[task_id] = Usage{node_id, read_only};
    it->second.nodes[node_id] += 1;
}

void VolumeSet::release(const std::string& volume_id, const std::string& task_id) {
    auto it = volumes_.find(volume_id);
    if (it == volumes_.end()) return;
    auto ut = it->second.tasks.find(task_id);
    if (ut == it->second.tasks.end()) return;
    int& c = it->second.nodes[ut->second.node_id];
    if (c > 0) c -= 1;
    it->second.tasks.erase(ut);
}

std::vector<std::pair<std::string, std::vector<std::string>>> VolumeSet::free_volumes() {
    std::vector<std::pair<std::string, std::vector<std::string>>> out;
    for (auto& kv : volumes_) {
        Info& info = kv.second;
        if (!info.store) continue;
        std::vector<std::string> changed;
        for (PublishStatus& st : info.store->publish_status) {
            auto n = info.nodes.find(st.node_id);
            if ((n == info.nodes.end() || n->second == 0) && st.state == VolumePublished) {   // volumes.go:200-203
                st.state = VolumePendingNodeUnpublish;
                changed.push_back(st.node_id);
            }
        }
        if (!changed.empty()) out.emplace_back(kv.first, std::move(changed));
    }
    return out;
}

std::string VolumeSet::is_available_on_node(const Mount& mount, const NodeInfo& node) const {
    static const std::string prefix = "group:";
    if (mount.source.compare(0, prefix.size(), prefix) == 0) {
        auto g = by_group_.find(mount.source.substr(prefix.size()));
        if (g == by_group_.end()) return "";
        for (const std::string& id : g->second)
            if (check_volume(id, node, mount.read_only)) return id;
        return "";
    }
    auto n = by_name_.find(mount.source);
    if (n == by_name_.end() || !check_volume(n->second, node, mount.read_only)) return "";
    return n->second;
}

bool VolumeSet::check_volume(const std::string& id, const NodeInfo& node, bool read_only) const {
    auto it = volumes_.find(id);
    if (it == volumes_.end()) return false;   // (the reference would dereference a nil volume here; ids come from byGroup / byName, which only hold volumes of the set)
    const Info& vi = it->second;
    if (vi.volume->availability != VolumeAvailabilityActive) return false;
    bool has_top = false;
    const std::map<std::string, std::string>* top = nullptr;
    static const std::map<std::string, std::string> none;
    if (node.node->has_description)
        for (const Node::CSIInfo& c : node.node->csi)
            if (c.plugin_name == vi.volume->driver_name) {
                has_top = c.has_topology;
                top = &c.segments;
                break;
            }
    if (vi.volume->scope == VolumeScopeSingleNode)
        for (const auto& kv : vi.tasks)
            if (kv.second.node_id != node.node->id) return false;
    switch (vi.volume->sharing) {
    case VolumeSharingNone:
        if (!vi.tasks.empty()) return false;
        break;
    case VolumeSharingOneWriter:
        if (!read_only)
            for (const auto& kv : vi.tasks)
                if (!kv.second.read_only) return false;
        break;
    case VolumeSharingReadOnly:
        if (!read_only) return false;
        break;
    default: break;
    }
    return is_in_topology(has_top, top ? *top : none, vi.volume->accessible);
}

// ============================================================================
// container/heap over nodeMaxHeap (nodeheap.go:3-31; Go stdlib container/heap)
// ============================================================================

static bool heap_less(const NodeMaxHeap& h, int i, int j) {   // reversed: max-heap (nodeheap.go:17-20)
    return h.less_func(h.nodes[size_t(j)], h.nodes[size_t(i)]);
}
static void heap_swap(NodeMaxHeap& h, int i, int j) { std::swap(h.nodes[size_t(i)], h.nodes[size_t(j)]); }
static void heap_up(NodeMaxHeap& h, int j) {
    for (;;) {
        int i = (j - 1) / 2;   // parent
        if (i == j || !heap_less(h, j, i)) break;
        heap_swap(h, i, j);
        j = i;
    }
}
static bool heap_down(NodeMaxHeap& h, int i0, int n) {
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1;
        int j2 = j1 + 1;
        if (j2 < n && heap_less(h, j2, j1)) j = j2;
        if (!heap_less(h, j, i)) break;
        heap_swap(h, i, j);
        i = j;
    }
    return i > i0;
}
static void heap_push(NodeMaxHeap& h, const NodeInfo& x) {
    h.nodes.push_back(x);   // Push: append + length++ (nodeheap.go:22-25)
    h.length++;
    heap_up(h, h.length - 1);
}
static void heap_pop(NodeMaxHeap& h) {
    int n = h.length - 1;
    heap_swap(h, 0, n);
    heap_down(h, 0, n);
    h.length--;   // Pop only shrinks length (nodeheap.go:27-31)
}
static void heap_fix(NodeMaxHeap& h, int i) {
    if (!heap_down(h, i, h.length)) heap_up(h, i);
}
static void heap_init(NodeMaxHeap& h) {
    int n = h.length;
    for (int i = n / 2 - 1; i >= 0; --i) heap_down(h, i, n);
}

DecisionTree* DecisionTree::child(const std::string& v) {
    for (auto& kv : next)
        if (kv.first == v) return kv.second.get();
    next.emplace_back(v, std::make_unique<DecisionTree>());
    return next.back().second.get();
}

// orderedNodes, decision_tree.go:24-52
std::vector<NodeInfo>& DecisionTree::ordered_nodes(const std::function<bool(const NodeInfo&)>& meets) {
    if (heap.length != int(heap.nodes.size())) {
        for (size_t i = 0; i < heap.nodes.size();) {
            if (meets(heap.nodes[i])) ++i;
            else {
                heap.nodes[i] = heap.nodes.back();
                heap.nodes.pop_back();
            }
        }
        heap.length = int(heap.nodes.size());
        heap_init(heap);
    }
    while (heap.length > 0) heap_pop(heap);
    return heap.nodes;
}

// ============================================================================
// manager/scheduler/nodeset.go
// ============================================================================

void Scheduler::ns_add_or_update(const NodeInfo& ni) {   // addOrUpdateNode nodeset.go:33-35
    auto it = slot_of_.find(ni.node->id);
    if (it == slot_of_.end()) {
        // the canonical scan order (Go's map order is unspecified) is the SLOT order; a node that is new to the set takes the lowest
        // slot a removed node left behind, else a new one at the end — the rule the engine's node index follows
        if (!free_slots_.empty()) {
            const size_t sl = *free_slots_.begin();
            free_slots_.erase(free_slots_.begin());
            slot_of_[ni.node->id] = sl;
            slots_[sl] = Slot{true, ni};
        } else {
            slot_of_[ni.node->id] = slots_.size();
            slots_.push_back(Slot{true, ni});
        }
    } else {
        slots_[it->second].present = true;
        slots_[it->second].info = ni;
    }
}
void Scheduler::ns_update(const NodeInfo& ni) {   // updateNode nodeset.go:39-44
    auto it = slot_of_.find(ni.node->id);
    if (it != slot_of_.end() && slots_[it->second].present) slots_[it->second].info = ni;
}
void Scheduler::add_or_update_node_info(const NodeInfo& ni) { ns_add_or_update(ni); }
bool Scheduler::node_info(const std::string& id, NodeInfo* out) const {   // nodeInfo nodeset.go:23-29
    auto it = slot_of_.find(id);
    if (it == slot_of_.end() || !slots_[it->second].present) return false;
    *out = slots_[it->second].info;
    return true;
}
size_t Scheduler::node_count() const {
    size_t n = 0;
    for (auto& s : slots_) n += s.present;
    return n;
}
std::vector<std::string> Scheduler::node_ids() const {
    std::vector<std::string> out;
    for (auto& s : slots_)
        if (s.present) out.push_back(s.info.node->id);
    return out;
}
void Scheduler::delete_node(const std::string& id) {   // remove nodeset.go:46-48 (EventDeleteNode, scheduler.go:198-199)
    auto it = slot_of_.find(id);
    if (it != slot_of_.end()) {
        slots_[it->second].present = false;
        slots_[it->second].info = NodeInfo();
        free_slots_.insert(it->second);   // delete(ns.nodes, nodeID): the entry is gone, its place in the canonical order is free
        slot_of_.erase(it);
    }
}

// tree, nodeset.go:50-124 — node iteration in canonical (ascending index) order.
DecisionTree Scheduler::tree(const std::string& service_id, const std::vector<Preference>& prefs, int max_assignments,
                             const std::function<bool(const NodeInfo&)>& meets, const NodeLess& less) {
    DecisionTree root;
    if (max_assignments == 0) return root;
    const uint32_t svc_key = service_key(service_id);
    for (const Slot& slot : slots_) {
        if (!slot.present) continue;
        const NodeInfo& node = slot.info;   // Go copies the struct (plain memcpy); a reference is the cost-fair equivalent
        DecisionTree* tree = &root;
        for (const Preference& pref : prefs) {
            if (!pref.is_spread) continue;
            const std::string& d = pref.descriptor;
            std::string value;
            if (has_prefix_fold(d, kNodeLabelPrefix)) {
                if (!node.node->labels_nil) {
                    auto it = node.node->labels.find(d.substr(sizeof(kNodeLabelPrefix) - 1));
                    if (it != node.node->labels.end()) value = it->second;
                }
            } else if (has_prefix_fold(d, kEngineLabelPrefix)) {
                if (node.node->has_description && node.node->has_engine && !node.node->engine_labels_nil) {
                    auto it = node.node->engine_labels.find(d.substr(sizeof(kEngineLabelPrefix) - 1));
                    if (it != node.node->engine_labels.end()) value = it->second;
                }
            } else {
                continue;
            }
            if (node.by_service) tree->tasks += node.svc_count_key(svc_key);
            tree->has_next = true;
            tree = tree->child(value);
        }
        if (node.by_service) tree->tasks += node.svc_count_key(svc_key);
        if (!tree->heap.less_func) tree->heap.less_func = less;
        if (tree->heap.length < max_assignments) {
            if (meets(node)) heap_push(tree->heap, node);
        } else if (less(node, tree->heap.nodes[0])) {
            if (meets(node)) {
                tree->heap.nodes[0] = node;
                heap_fix(tree->heap, 0);
            }
        }
    }
    return root;
}

// ============================================================================
// manager/scheduler/scheduler.go
// ============================================================================

void Scheduler::OrderedTasks::put(const std::string& id, const TaskPtr& t) {
    auto it = pos.find(id);
    if (it != pos.end()) { items[it->second].second = t; return; }
    pos[id] = items.size();
    items.emplace_back(id, t);
}
void Scheduler::OrderedTasks::erase(const std::string& id) {
    auto it = pos.find(id);
    if (it == pos.end()) return;
    items[it->second].second = nullptr;
    items[it->second].first.clear();
    pos.erase(it);
}
TaskPtr Scheduler::OrderedTasks::get(const std::string& id) const {
    auto it = pos.find(id);
    return it == pos.end() ? nullptr : items[it->second].second;
}
void Scheduler::OrderedTasks::compact() {
    if (pos.size() == items.size()) return;
    std::vector<std::pair<std::string, TaskPtr>> keep;
    keep.reserve(pos.size());
    pos.clear();
    for (auto& kv : items)
        if (kv.second) {
            pos[kv.first] = keep.size();
            keep.push_back(std::move(kv));
        }
    items.swap(keep);
}

// createTask, scheduler.go:254-283
bool Scheduler::create_task(const TaskPtr& t) {
    if (t->state < TaskStatePending || t->state > TaskStateRunning) return false;
    all_tasks_[t->id] = t;
    if (t->node_id.empty()) {
        enqueue(t);
        return true;
    }
    if (t->state == TaskStatePending) {
        preassigned_[t->id] = true;
        pending_preassigned_.put(t->id, t);
        return false;
    }
    NodeInfo ni;
    if (node_info(t->node_id, &ni) && ni.add_task(t)) ns_update(ni);
    return false;
}

// setupTasksList, scheduler.go:88-124: what Run() does with every task found in the store when it starts. Unlike the
// createTask event it also ignores tasks that are still PENDING although their desired state is already past
// COMPLETED (:98-103). Assigned tasks end up in the node's NodeInfo (buildNodeSet -> newNodeInfo == addTask).
bool Scheduler::setup_task(const TaskPtr& t) {
    if (t->state < TaskStatePending || t->state > TaskStateRunning) return false;
    if (t->state == TaskStatePending && t->desired_state > TaskStateCompleted) return false;
    all_tasks_[t->id] = t;
    if (t->node_id.empty()) {
        enqueue(t);
        return true;
    }
    if (t->state == TaskStatePending) {
        preassigned_[t->id] = true;
        pending_preassigned_.put(t->id, t);
        return false;
    }
    volumes_.reserve_task_volumes(*t);   // scheduler.go:115-116
    NodeInfo ni;
    if (node_info(t->node_id, &ni) && ni.add_task(t)) ns_update(ni);
    return false;
}

// updateTask, scheduler.go:285-349
bool Scheduler::update_task(const TaskPtr& t) {
    if (t->state < TaskStatePending) return false;
    TaskPtr old_task;
    auto it = all_tasks_.find(t->id);
    if (it != all_tasks_.end()) old_task = it->second;
    if (t->state > TaskStateRunning) {
        if (!old_task) return false;
        if (t->state != old_task->state && (t->state == TaskStateFailed || t->state == TaskStateRejected)) {
            if (!preassigned_.count(t->id)) {
                NodeInfo ni;
                if (node_info(t->node_id, &ni)) {
                    ni.task_failed(now, *t);
                    ns_update(ni);
                }
            }
        }
        delete_task(*old_task);
        return true;
    }
    if (t->node_id.empty()) {
        if (old_task) delete_task(*old_task);
        all_tasks_[t->id] = t;
        enqueue(t);
        return true;
    }
    if (t->state == TaskStatePending) {
        if (old_task) delete_task(*old_task);
        preassigned_[t->id] = true;
        all_tasks_[t->id] = t;
        pending_preassigned_.put(t->id, t);
        return false;
    }
    all_tasks_[t->id] = t;
    NodeInfo ni;
    if (node_info(t->node_id, &ni) && ni.add_task(t)) ns_update(ni);
    return false;
}

// deleteTask, scheduler.go:351-366
bool Scheduler::delete_task(const Task& t) {
    all_tasks_.erase(t.id);
    preassigned_.erase(t.id);
    pending_preassigned_.erase(t.id);
    for (const VolumeAttachment& va : t.volumes) volumes_.release(va.id, t.id);   // scheduler.go:355-358
    NodeInfo ni;
    if (node_info(t.node_id, &ni) && ni.remove_task(t)) {
        ns_update(ni);
        return true;
    }
    return false;
}
bool Scheduler::delete_task_event(const TaskPtr& t) { return delete_task(*t); }

// EventUpdateVolume, scheduler.go:200-213 (and the volumes of the store at start, :70-81): only volumes the plugin has created
void Scheduler::update_volume(const VolumePtr& v) {
    if (v->has_volume_info && !v->volume_id.empty()) volumes_.add_or_update(v);
}

// createOrUpdateNode, scheduler.go:368-396
void Scheduler::create_or_update_node(const NodePtr& n) {
    NodeInfo ni;
    bool found = node_info(n->id, &ni);
    auto resources = std::make_shared<Resources>();
    if (n->has_description && n->has_resources) {
        *resources = n->resources;   // Copy()
        if (found) {
            for (auto& kv : *ni.tasks) {
                Resources r = task_reservations(*kv.second);
                resources->memory_bytes -= r.memory_bytes;
                resources->nano_cpus -= r.nano_cpus;
                generic_consume(&resources->generic, kv.second->assigned_generic);
            }
        }
    }
    if (!found) {
        ni = new_node_info(n, {}, *resources, now);
    } else {
        ni.node = n;
        ni.available = resources;
    }
    ns_add_or_update(ni);
}

void Scheduler::put_decision(std::vector<Decision>& decisions, std::unordered_map<std::string, size_t>& decided,
                             const std::string& id, Decision d) {
    auto it = decided.find(id);
    if (it != decided.end()) { decisions[it->second] = std::move(d); return; }
    decided[id] = decisions.size();
    decisions.push_back(std::move(d));
}

// taskFitNode, scheduler.go:646-690
TaskPtr Scheduler::task_fit_node(const TaskPtr& t, const std::string& node_id) {
    NodeInfo ni;
    if (!node_info(node_id, &ni)) return nullptr;
    auto new_t = std::make_shared<Task>(*t);
    pipeline.set_task(t.get());
    if (!pipeline.process(ni)) {
        new_t->err = pipeline.explain();
        all_tasks_[t->id] = new_t;
        return new_t;
    }
    // scheduler.go:663-677: the attachments for the task on this node (chosen, NOT reserved: the reference reserves only in
    // scheduleNTasksOnNodes and at start-up)
    {
        std::vector<VolumeAttachment> attachments;
        std::string verr;
        if (!volumes_.choose_task_volumes(*t, ni, &attachments, &verr, &new_t->chosen_prefix)) {
            new_t->err = verr;
            all_tasks_[t->id] = new_t;
            return new_t;
        }
        new_t->volumes = attachments;
    }
    new_t->state = TaskStateAssigned;
    new_t->message = "scheduler confirmed task can run on preassigned node";
    new_t->err.clear();
    all_tasks_[t->id] = new_t;
    if (ni.add_task(new_t)) ns_update(ni);
    return new_t;
}

// processPreassignedTasks, scheduler.go:398-426 (store commit always succeeds here)
std::vector<Decision> Scheduler::process_preassigned() {
    std::vector<Decision> decisions;
    pending_preassigned_.compact();
    auto items = pending_preassigned_.items;   // iterate a snapshot: the loop below erases entries
    for (auto& kv : items) {
        if (!kv.second) continue;
        TaskPtr new_t = task_fit_node(kv.second, kv.second->node_id);
        if (!new_t) continue;
        decisions.push_back(Decision{kv.second, new_t});
    }
    for (const Decision& d : decisions)
        if (d.new_task->state == TaskStateAssigned) pending_preassigned_.erase(d.old_task->id);
    return decisions;
}

// tick, scheduler.go:429-488
std::vector<Decision> Scheduler::tick() {
    struct GroupKey {
        std::string service_id; uint64_t spec_version;
        bool operator<(const GroupKey& o) const { return service_id != o.service_id ? service_id < o.service_id : spec_version < o.spec_version; }
    };
    std::vector<std::unique_ptr<OrderedTasks>> groups;   // first-seen order
    std::map<GroupKey, size_t> group_index;
    std::vector<TaskPtr> one_off;
    std::vector<Decision> decisions;
    std::unordered_map<std::string, size_t> decided;

    unassigned_.compact();
    auto items = unassigned_.items;
    for (auto& kv : items) {
        const std::string& task_id = kv.first;
        TaskPtr t = kv.second;
        if (!t || !t->node_id.empty()) {
            unassigned_.erase(task_id);
            continue;
        }
        if (t->has_spec_version) {
            GroupKey key{t->service_id, t->spec_version};
            auto it = group_index.find(key);
            if (it == group_index.end()) {
                group_index[key] = groups.size();
                groups.push_back(std::make_unique<OrderedTasks>());
                it = group_index.find(key);
            }
            groups[it->second]->put(task_id, t);
        } else {
            one_off.push_back(t);
        }
        unassigned_.erase(task_id);
    }
    for (auto& g : groups) schedule_task_group(*g, decisions, decided);
    for (const TaskPtr& t : one_off) {
        OrderedTasks g;
        g.put(t->id, t);
        schedule_task_group(g, decisions, decided);
    }
    // applySchedulingDecisions (scheduler.go:490-643): the oracle has no store; every
    // decision commits, so the `failed` rollback at 472-487 never runs.
    return decisions;
}

// scheduleTaskGroup, scheduler.go:694-748
void Scheduler::schedule_task_group(OrderedTasks& group, std::vector<Decision>& decisions,
                                    std::unordered_map<std::string, size_t>& decided) {
    TaskPtr t;
    for (auto& kv : group.items)
        if (kv.second) { t = kv.second; break; }
    pipeline.set_task(t.get());
    int64_t now_captured = now;
    const Task* tp = t.get();
    const VersionedService vs_key{tp->service_id, tp->has_spec_version ? tp->spec_version : 0};
    const uint32_t svc_key = task_service_key(*tp);
    NodeLess node_less = [this, now_captured, svc_key, vs_key](const NodeInfo& a, const NodeInfo& b) {
        ++nodeless_calls;
        int64_t fa = a.count_recent_failures_key(now_captured, vs_key);
        int64_t fb = b.count_recent_failures_key(now_captured, vs_key);
        if (fa >= kMaxFailures || fb >= kMaxFailures) {
            if (fa > fb) return false;
            if (fb > fa) return true;
        }
        int64_t sa = a.svc_count_key(svc_key), sb = b.svc_count_key(svc_key);
        if (sa < sb) return true;
        if (sa > sb) return false;
        return a.active_tasks_count < b.active_tasks_count;
    };
    std::vector<Preference> prefs;
    if (t->has_placement) prefs = t->preferences;
    int k = int(group.live());
    DecisionTree tree = this->tree(t->service_id, prefs, k, [this](const NodeInfo& n) { return pipeline.process(n); }, node_less);
    schedule_n_on_subtree(k, group, &tree, decisions, decided, node_less);
    if (group.live() != 0) no_suitable_node(group, decisions, decided);
}

// scheduleNTasksOnSubtree, scheduler.go:772-825
int Scheduler::schedule_n_on_subtree(int n, OrderedTasks& group, DecisionTree* tree, std::vector<Decision>& decisions,
                                     std::unordered_map<std::string, size_t>& decided, const NodeLess& less) {
    if (!tree->has_next) {
        std::vector<NodeInfo>& nodes = tree->ordered_nodes([this](const NodeInfo& ni) { return pipeline.process(ni); });
        if (nodes.empty()) return 0;
        return schedule_n_on_nodes(n, group, nodes, decisions, decided, less);
    }
    int tasks_scheduled = 0;
    int64_t tasks_in_usable = tree->tasks;
    std::map<DecisionTree*, bool> no_room;
    bool converging = true;
    while (tasks_scheduled != n && no_room.size() != tree->next.size() && converging) {
        int64_t branches = int64_t(tree->next.size()) - int64_t(no_room.size());
        int64_t desired = (tasks_in_usable + n - tasks_scheduled) / branches;
        int64_t remainder = (tasks_in_usable + n - tasks_scheduled) % branches;
        converging = false;
        for (auto& kv : tree->next) {
            DecisionTree* subtree = kv.second.get();
            if (no_room.count(subtree)) continue;
            int64_t subtree_tasks = subtree->tasks;
            if (subtree_tasks < desired || (subtree_tasks == desired && remainder > 0)) {
                converging = true;
                int64_t to_assign = desired - subtree_tasks;
                if (remainder > 0) to_assign++;
                int res = schedule_n_on_subtree(int(to_assign), group, subtree, decisions, decided, less);
                if (res < to_assign) {
                    no_room[subtree] = true;
                    tasks_in_usable -= subtree_tasks;
                } else if (remainder > 0) {
                    remainder--;
                }
                tasks_scheduled += res;
            }
        }
    }
    return tasks_scheduled;
}

// scheduleNTasksOnNodes, scheduler.go:844-924
int Scheduler::schedule_n_on_nodes(int n, OrderedTasks& group, std::vector<NodeInfo>& nodes, std::vector<Decision>& decisions,
                                   std::unordered_map<std::string, size_t>& decided, const NodeLess& less) {
    int tasks_scheduled = 0;
    std::map<int, bool> failed_constraints;
    int node_iter = 0;
    int node_count = int(nodes.size());
    // `for taskID, t := range taskGroup` with deletes of the current key during iteration.
    for (size_t gi = 0; gi < group.items.size(); ++gi) {
        if (!group.items[gi].second) continue;
        std::string task_id = group.items[gi].first;
        TaskPtr t = group.items[gi].second;
        if (decided.count(task_id)) continue;   // scheduler.go:852

        NodeInfo* node = &nodes[size_t(node_iter % node_count)];
        auto new_t = std::make_shared<Task>(*t);
        {   // scheduler.go:857-874: choose the volumes on this node (an error is logged, the task is assigned without them), then reserve
            std::vector<VolumeAttachment> attachments;
            volumes_.choose_task_volumes(*t, *node, &attachments, nullptr, &new_t->chosen_prefix);
            new_t->volumes = attachments;
        }
        new_t->node_id = node->node->id;
        volumes_.reserve_task_volumes(*new_t);
        new_t->state = TaskStateAssigned;
        new_t->message = "scheduler assigned task to node";
        new_t->err.clear();
        all_tasks_[t->id] = new_t;

        NodeInfo ni;
        bool have = node_info(node->node->id, &ni);
        if (have && ni.add_task(new_t)) {
            ns_update(ni);
            nodes[size_t(node_iter % node_count)] = ni;
        }
        put_decision(decisions, decided, task_id, Decision{t, new_t});
        group.erase(task_id);
        tasks_scheduled++;
        if (tasks_scheduled == n) return tasks_scheduled;

        if (node_iter + 1 < node_count) {
            NodeInfo next_node = nodes[size_t((node_iter + 1) % node_count)];
            if (less(next_node, ni)) node_iter++;
        } else {
            node_iter++;
        }
        int orig = node_iter;
        while (failed_constraints[node_iter % node_count] || !pipeline.process(nodes[size_t(node_iter % node_count)])) {
            failed_constraints[node_iter % node_count] = true;
            node_iter++;
            if (node_iter - orig == node_count) return tasks_scheduled;
        }
    }
    return tasks_scheduled;
}

// noSuitableNode, scheduler.go:928-971
void Scheduler::no_suitable_node(OrderedTasks& group, std::vector<Decision>& decisions,
                                 std::unordered_map<std::string, size_t>& decided) {
    std::string explanation = pipeline.explain();
    for (auto& kv : group.items) {
        TaskPtr t = kv.second;
        if (!t) continue;
        auto sit = services_.find(t->service_id);
        if (sit == services_.end()) continue;   // service == nil: task is dropped from the scheduler
        auto new_t = std::make_shared<Task>(*t);
        const ServiceRec& svc = sit->second;
        if (svc.has_spec_version && new_t->has_spec_version && svc.spec_version > new_t->spec_version) {
            if (t->state == TaskStatePending && t->desired_state >= TaskStateShutdown) {
                new_t->state = TaskStateShutdown;
                new_t->err.clear();
            }
        } else {
            new_t->err = explanation.empty() ? "no suitable node" : "no suitable node (" + explanation + ")";
            enqueue(new_t);
        }
        all_tasks_[t->id] = new_t;
        put_decision(decisions, decided, t->id, Decision{t, new_t});
    }
}

// ---------------------------------------------------------------------------------------------
// api/genericresource/validate.go:54-85 — HasResource
// ---------------------------------------------------------------------------------------------
bool generic_has_resource(const GenericResource& res, const GenericList& resources) {
    for (const GenericResource& r : resources) {
        if (res.kind != r.kind) continue;                       // :56-58
        if (!r.named) {                                         // DiscreteResourceSpec, :61-70
            if (res.named) return false;
            if (res.ivalue > r.ivalue) return false;
            return true;
        }
        if (!res.named) return false;                           // NamedResourceSpec, :71-80
        if (res.svalue != r.svalue) continue;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// manager/orchestrator/constraintenforcer/constraint_enforcer.go:65-196 — rejectNoncompliantTasks
// ---------------------------------------------------------------------------------------------
std::vector<std::string> enforce_node(const Node& node, const std::vector<TaskPtr>& tasks,
                                      const std::map<std::string, ServicePlacement>& service_placement) {
    std::vector<std::string> removed;
    if (node.availability != NodeAvailabilityActive) return removed;   // :70-72: drain = orchestrator's job, pause = hands off
    Resources available;                                               // :101-106
    if (node.has_description && node.has_resources) available = node.resources;
    GenericList fake_store;
    for (const TaskPtr& tp : tasks) {
        const Task& t = *tp;
        if (t.desired_state < TaskStateAssigned || t.desired_state > TaskStateCompleted) continue;   // :118-120
        if (t.state >= TaskStateCompleted) continue;                                                  // :124-126
        // placement: the service's CURRENT spec if the service exists, else the task's own (:152-161)
        bool has_placement = t.has_placement;
        const std::vector<std::string>* cons = &t.constraints;
        auto it = service_placement.find(t.service_id);
        if (it != service_placement.end()) {
            has_placement = it->second.has_placement;
            cons = &it->second.constraints;
        }
        if (has_placement && !cons->empty()) {                                                        // :162-168
            std::vector<Constraint> cs;
            if (!constraint_parse(*cons, &cs, nullptr)) cs.clear();   // `constraints, _ := constraint.Parse(...)`: nil on error
            if (!node_matches(cs, node)) {
                removed.push_back(t.id);
                continue;
            }
        }
        if (t.has_reservations) {                                                                     // :172-184
            if (t.reservations.memory_bytes > available.memory_bytes) { removed.push_back(t.id); continue; }
            if (t.reservations.nano_cpus > available.nano_cpus) { removed.push_back(t.id); continue; }
            available.memory_bytes -= t.reservations.memory_bytes;
            available.nano_cpus -= t.reservations.nano_cpus;
        }
        if (!t.assigned_generic.empty()) {                                                            // :188-199
            bool gone = false;
            for (const GenericResource& ta : t.assigned_generic)
                if (!generic_has_resource(ta, available.generic)) { gone = true; break; }
            if (gone) {
                removed.push_back(t.id);
                break;   // `break loop`: the whole task loop ends here (:193)
            }
            fake_store.insert(fake_store.end(), t.assigned_generic.begin(), t.assigned_generic.end());   // ClaimResources,
            generic_consume(&available.generic, t.assigned_generic);                                    // resource_management.go:36-40
        }
    }
    return removed;
}

}  // namespace orc
