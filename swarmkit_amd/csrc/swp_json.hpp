// swp_json.hpp — the small JSON document model the host-side scheduler mirror (swp_sched.cpp) uses for api.Node /
// api.Task / api.Service documents. In swarmkit these are protobuf-generated Go structs; across the test-bed boundary
// they travel as JSON with the Go field names, so that the parity tests read like the reference's struct literals.
// Objects keep insertion order (a Go struct's field order / Python dict order); sub-documents are shared by reference
// (immutable once parsed), so copying a task document to change its Status costs one small vector.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace swp {
namespace json {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind : uint8_t { Null, Bool, Int, Real, Str, Arr, Obj };
    Kind kind = Null;
    union {   // the scalar of a Bool / Int / Real (read only under its kind): one word, not three — a member is 104 bytes instead of 120
        bool b;
        int64_t i = 0;
        double d;
    };
    std::string s;
    std::shared_ptr<std::vector<Value>> a;
    std::shared_ptr<std::vector<Member>> o;

    Value() = default;
    static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
    static Value integer(int64_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value real(double v) { Value x; x.kind = Real; x.d = v; return x; }
    static Value str(std::string v) { Value x; x.kind = Str; x.s = std::move(v); return x; }
    static Value array() { Value x; x.kind = Arr; x.a = std::make_shared<std::vector<Value>>(); return x; }
    static Value object() { Value x; x.kind = Obj; x.o = std::make_shared<std::vector<Member>>(); return x; }

    bool is_null() const { return kind == Null; }
    bool is_obj() const { return kind == Obj; }
    bool is_arr() const { return kind == Arr; }
    bool is_str() const { return kind == Str; }
    bool is_num() const { return kind == Int || kind == Real; }

    // member lookup; nullptr when this is not an object, the key is absent, or the member is null
    // (a nil pointer / absent field and JSON null are the same thing for every reader on this path).
    // Member names are literals at every call site but two: the length is a compile-time constant, so a probe is one size
    // comparison and — on the one member whose length fits — a fixed-size memcmp the compiler turns into word compares
    // (round 5: `std::string == const char*` was a strlen + compare per member per look-up, 67 ms of a 100k-task tick).
    template <size_t N> static bool key_is(const std::string& k, const char (&lit)[N]) { return k.size() == N - 1 && std::memcmp(k.data(), lit, N - 1) == 0; }
    template <size_t N> const Value* get(const char (&key)[N]) const {
        if (kind != Obj) return nullptr;
        for (const Member& m : *o)
            if (key_is(m.first, key)) return m.second.kind == Null ? nullptr : &m.second;
        return nullptr;
    }
    const Value* get_key(const std::string& key) const {   // a member named at run time (a node id, a service id)
        if (kind != Obj) return nullptr;
        for (const Member& m : *o)
            if (m.first == key) return m.second.kind == Null ? nullptr : &m.second;
        return nullptr;
    }
    // a copy whose member vector is private (sub-documents stay shared): `dict(t)` in the Python twin
    Value shallow_copy() const {
        Value x = *this;
        if (kind == Obj) x.o = std::make_shared<std::vector<Member>>(*o);
        if (kind == Arr) x.a = std::make_shared<std::vector<Value>>(*a);
        return x;
    }
    void set(const std::string& key, Value v) {   // only on objects this code has just created / shallow-copied
        for (Member& m : *o)
            if (m.first == key) { m.second = std::move(v); return; }
        o->emplace_back(key, std::move(v));
    }
    void push(Value v) { a->push_back(std::move(v)); }
    size_t size() const { return kind == Arr ? a->size() : kind == Obj ? o->size() : 0; }
};

// nested lookup: at(doc, "Spec", "Resources", "Reservations") — nullptr as soon as a level is nil
inline const Value* at(const Value* d) { return d; }
template <size_t N, class... Rest> inline const Value* at(const Value* d, const char (&key)[N], const Rest&... rest) {
    return d == nullptr ? nullptr : at(d->get(key), rest...);
}
inline int64_t as_i64(const Value* v, int64_t def = 0) {
    if (v == nullptr) return def;
    switch (v->kind) {
        case Value::Int: return v->i;
        case Value::Real:   // (a real beyond int64 saturates: the conversion itself would be undefined)
            if (!(v->d == v->d)) return def;
            if (v->d >= 9223372036854775807.0) return INT64_MAX;
            if (v->d <= -9223372036854775808.0) return INT64_MIN;
            return (int64_t)v->d;
        case Value::Bool: return v->b ? 1 : 0;
        default: return def;
    }
}
inline const std::string& as_str(const Value* v) {
    static const std::string empty;
    return (v != nullptr && v->kind == Value::Str) ? v->s : empty;
}
// Python truthiness of `d.get(key)`: absent / null / "" / 0 / [] / {} are false
inline bool truthy(const Value* v) {
    if (v == nullptr) return false;
    switch (v->kind) {
        case Value::Null: return false;
        case Value::Bool: return v->b;
        case Value::Int: return v->i != 0;
        case Value::Real: return v->d != 0.0;
        case Value::Str: return !v->s.empty();
        case Value::Arr: return !v->a->empty();
        case Value::Obj: return !v->o->empty();
    }
    return false;
}

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Parser {
  public:
    Parser(const char* p, size_t n) : p_(p), e_(p + n) {}
    // NOT re-entrant on one thread: every Parser of a thread shares the scratch stacks below, so a parse started while another one is
    // under way on the same thread (nothing does that: a document is parsed in one go, no callbacks) would clear the outer one's open
    // members. The flag turns such a use into an error instead of a corrupted document.
    Value parse_document() {
        struct Busy {
            bool& b;
            explicit Busy(bool& x) : b(x) { if (b) throw ParseError("json: nested parse on one thread"); b = true; }
            ~Busy() { b = false; }
        } busy(scratch().busy);
        members_.clear();   // (left behind by a parse that failed half-way)
        elements_.clear();
        Value v = value(0);
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

  private:
    const char* p_;
    const char* e_;
    // the members / elements of the objects and arrays that are still open, innermost last; per thread, so that the room they have
    // grown to serves the next document (an event is one small document: two allocations less per event)
    struct Scratch { std::vector<Member> members; std::vector<Value> elements; bool busy = false; };
    static Scratch& scratch() { static thread_local Scratch s; return s; }
    std::vector<Member>& members_ = scratch().members;   // (bound once per parser: a thread_local is a function call per access)
    std::vector<Value>& elements_ = scratch().elements;
    [[noreturn]] void fail(const char* what) { throw ParseError(std::string("json: ") + what); }
    void ws() {
        while (p_ != e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool lit(const char* w) {
        size_t n = std::strlen(w);
        if ((size_t)(e_ - p_) >= n && std::memcmp(p_, w, n) == 0) { p_ += n; return true; }
        return false;
    }
    static void utf8(std::string& out, uint32_t cp) {
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
    // length of the well-formed UTF-8 sequence at p (RFC 3629 byte ranges: no overlong forms, no surrogates, nothing beyond U+10FFFF); 0: none
    static size_t utf8_seq(const unsigned char* p, const unsigned char* e) {
        const unsigned char c = *p;
        if (c < 0x80) return 1;
        size_t n;
        if (c >= 0xC2 && c <= 0xDF) n = 1;
        else if (c >= 0xE0 && c <= 0xEF) n = 2;
        else if (c >= 0xF0 && c <= 0xF4) n = 3;
        else return 0;
        if ((size_t)(e - p) < n + 1) return 0;
        if (c == 0xE0 && p[1] < 0xA0) return 0;                    // overlong
        if (c == 0xED && p[1] > 0x9F) return 0;                    // U+D800..DFFF
        if (c == 0xF0 && p[1] < 0x90) return 0;
        if (c == 0xF4 && p[1] > 0x8F) return 0;                    // beyond U+10FFFF
        for (size_t k = 1; k <= n; ++k)
            if ((p[k] & 0xC0) != 0x80) return 0;
        return n + 1;
    }
    static bool utf8_ok(const std::string& s) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
        const unsigned char* e = p + s.size();
        while (p != e) {
            const size_t n = utf8_seq(p, e);
            if (n == 0) return false;
            p += n;
        }
        return true;
    }
    // A string that is not UTF-8 is coerced the way Go's encoding/json does it (decode.go, unquote: utf8.DecodeRune — an ill-formed
    // sequence is ONE byte long and reads as U+FFFD), so that an event written by something other than Go's marshaller is scheduled, not
    // lost, and what goes out in the decisions is JSON again. (Until round 6 such a document was refused: ADVICE r5.)
    static void utf8_coerce(std::string& s) {
        if (utf8_ok(s)) return;
        std::string out;
        out.reserve(s.size() + 8);
        const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
        const unsigned char* e = p + s.size();
        while (p != e) {
            const size_t n = utf8_seq(p, e);
            if (n == 0) { out.append("\xEF\xBF\xBD"); ++p; }
            else { out.append(reinterpret_cast<const char*>(p), n); p += n; }
        }
        s.swap(out);
    }
    uint32_t hex4() {
        if (e_ - p_ < 4) fail("short \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        ++p_;   // opening quote
        // a name, an id, a label: nothing escaped — found by one scan, copied by one assignment
        const char* q = p_;
        unsigned char high = 0;
        while (q != e_ && *q != '"' && *q != '\\') high |= (unsigned char)*q++;
        if (q == e_) fail("unterminated string");
        std::string out(p_, q);
        p_ = q;
        if (*p_ == '"') {
            ++p_;
            if (high & 0x80u) utf8_coerce(out);   // (what goes in comes out again: the decisions must stay JSON)
            return out;
        }
        for (;;) {
            if (p_ == e_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') {
                utf8_coerce(out);
                return out;
            }
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ == e_) fail("unterminated escape");
            char x = *p_++;
            switch (x) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        const char* save = p_;
                        p_ += 2;
                        uint32_t lo = hex4();
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else p_ = save;
                    }
                    if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;   // half of a surrogate pair on its own: U+FFFD, as Go's encoding/json decodes it
                    utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
    }
    Value number() {
        const char* s = p_;
        bool real = false;
        if (p_ != e_ && (*p_ == '-' || *p_ == '+')) ++p_;
        while (p_ != e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) {
            if (*p_ == '.' || *p_ == 'e' || *p_ == 'E') real = true;
            ++p_;
        }
        if (!real && p_ - s >= 1 && p_ - s <= 18 && *s != '+') {   // the usual case: an integer of at most 17 digits — no overflow to look for
            const char* q = s;
            const bool neg = *q == '-';
            if (neg) ++q;
            if (q != p_) {
                int64_t v = 0;
                bool digits = true;
                for (; q != p_; ++q) {
                    digits = digits && *q >= '0' && *q <= '9';
                    v = v * 10 + (*q - '0');
                }
                if (digits) return Value::integer(neg ? -v : v);
            }
        }
        std::string t(s, p_);
        if (t.empty() || t == "-") fail("bad number");
        if (!real) {
            errno = 0;
            char* end = nullptr;
            long long v = std::strtoll(t.c_str(), &end, 10);
            if (errno == 0 && end != nullptr && *end == 0) return Value::integer((int64_t)v);
            // uint64 values above INT64_MAX (MaxReplicas is a uint64): keep the bit pattern
            errno = 0;
            unsigned long long u = std::strtoull(t.c_str(), &end, 10);
            if (errno == 0 && end != nullptr && *end == 0) return Value::integer((int64_t)u);
        }
        return Value::real(std::strtod(t.c_str(), nullptr));
    }
    Value value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p_ == e_) fail("unexpected end");
        char c = *p_;
        if (c == '{') {
            ++p_;
            ws();
            if (p_ != e_ && *p_ == '}') { ++p_; return Value::object(); }
            // The members are gathered on the parser's own stack (nested objects use the part above) and moved into a vector of exactly
            // their number when the object closes: a document's member vectors were grown one by one before (four allocations and seven
            // moved members for an object of five).
            const size_t base = members_.size();
            for (;;) {
                ws();
                if (p_ == e_ || *p_ != '"') fail("expected a member name");
                std::string k = string();
                ws();
                if (p_ == e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                Value m = value(depth + 1);
                size_t dup = base;   // a repeated key keeps its first position, last value (Python dict)
                while (dup < members_.size() && members_[dup].first != k) ++dup;
                if (dup < members_.size()) members_[dup].second = std::move(m);
                else members_.emplace_back(std::move(k), std::move(m));
                ws();
                if (p_ != e_ && *p_ == ',') { ++p_; continue; }
                if (p_ != e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
            Value v;
            v.kind = Value::Obj;
            v.o = std::make_shared<std::vector<Member>>(std::make_move_iterator(members_.begin() + (std::ptrdiff_t)base), std::make_move_iterator(members_.end()));
            members_.resize(base);
            return v;
        }
        if (c == '[') {
            ++p_;
            ws();
            if (p_ != e_ && *p_ == ']') { ++p_; return Value::array(); }
            const size_t base = elements_.size();
            for (;;) {
                Value x = value(depth + 1);
                elements_.push_back(std::move(x));
                ws();
                if (p_ != e_ && *p_ == ',') { ++p_; continue; }
                if (p_ != e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
            Value v;
            v.kind = Value::Arr;
            v.a = std::make_shared<std::vector<Value>>(std::make_move_iterator(elements_.begin() + (std::ptrdiff_t)base), std::make_move_iterator(elements_.end()));
            elements_.resize(base);
            return v;
        }
        if (c == '"') return Value::str(string());
        if (lit("true")) return Value::boolean(true);
        if (lit("false")) return Value::boolean(false);
        if (lit("null")) return Value();
        if (c == '-' || (c >= '0' && c <= '9')) return number();
        fail("unexpected character");
    }
};

inline Value parse(const char* text, size_t len) { return Parser(text, len).parse_document(); }
inline Value parse(const std::string& text) { return parse(text.data(), text.size()); }

inline void dump_string(std::string& out, const std::string& s) {
    out.push_back('"');
    bool plain = true;   // ids, names, messages: nothing to escape — one append
    for (unsigned char c : s) plain = plain && c >= 0x20 && c != '"' && c != '\\';
    if (plain) {
        out += s;
        out.push_back('"');
        return;
    }
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof buf, "\\u%04x", c);
                    out += buf;
                } else out.push_back((char)c);   // UTF-8 passes through
        }
    }
    out.push_back('"');
}
inline void dump(std::string& out, const Value& v) {
    switch (v.kind) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Int: out += std::to_string(v.i); break;
        case Value::Real: {
            char buf[40];
            std::snprintf(buf, sizeof buf, "%.17g", v.d);
            out += buf;
            break;
        }
        case Value::Str: dump_string(out, v.s); break;
        case Value::Arr: {
            out.push_back('[');
            bool first = true;
            for (const Value& x : *v.a) {
                if (!first) out.push_back(',');
                first = false;
                dump(out, x);
            }
            out.push_back(']');
            break;
        }
        case Value::Obj: {
            out.push_back('{');
            bool first = true;
            for (const Member& m : *v.o) {
                if (!first) out.push_back(',');
                first = false;
                dump_string(out, m.first);
                out.push_back(':');
                dump(out, m.second);
            }
            out.push_back('}');
            break;
        }
    }
}
inline std::string dump(const Value& v) {
    std::string out;
    dump(out, v);
    return out;
}

}   // namespace json
}   // namespace swp
