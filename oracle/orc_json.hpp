// orc_json.hpp — minimal JSON reader/writer for the ORACLE's test-facing C API.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md). Integers are parsed exactly
// as int64/uint64 (NanoCPUs / MemoryBytes are int64 in api/types.proto:68-77;
// a double would silently round them).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orcjson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Int, Str, Arr, Obj } kind = Null;
    bool b = false;
    int64_t i = 0;      // integers only; reals are rejected (none on this path)
    uint64_t u = 0;     // same value as unsigned when non-negative
    std::string s;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;   // insertion-ordered

    bool is_null() const { return kind == Null; }
    const Value* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    // present and not JSON null
    const Value* obj_or_null(const char* key) const {
        const Value* v = get(key);
        if (!v || v->is_null()) return nullptr;
        return v;
    }
    int64_t int_or(const char* key, int64_t dflt) const {
        const Value* v = get(key);
        if (!v || v->is_null()) return dflt;
        if (v->kind == Bool) return v->b ? 1 : 0;
        if (v->kind != Int) throw std::runtime_error(std::string("json: field ") + key + " is not an integer");
        return v->i;
    }
    uint64_t uint_or(const char* key, uint64_t dflt) const {
        const Value* v = get(key);
        if (!v || v->is_null()) return dflt;
        if (v->kind != Int) throw std::runtime_error(std::string("json: field ") + key + " is not an integer");
        return v->u;
    }
    std::string str_or(const char* key, const std::string& dflt) const {
        const Value* v = get(key);
        if (!v || v->is_null()) return dflt;
        if (v->kind != Str) throw std::runtime_error(std::string("json: field ") + key + " is not a string");
        return v->s;
    }
};

class Parser {
  public:
    explicit Parser(const char* text) : p_(text) {}
    ValuePtr parse() {
        ValuePtr v = value();
        ws();
        if (*p_) fail("trailing characters");
        return v;
    }

  private:
    const char* p_;
    [[noreturn]] void fail(const char* why) { throw std::runtime_error(std::string("json: ") + why); }
    void ws() {
        while (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r') ++p_;
    }
    ValuePtr value() {
        ws();
        auto v = std::make_shared<Value>();
        switch (*p_) {
        case '{': {
            ++p_;
            v->kind = Value::Obj;
            ws();
            if (*p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                if (*p_ != '"') fail("expected object key");
                std::string k = str();
                ws();
                if (*p_ != ':') fail("expected ':'");
                ++p_;
                v->obj.emplace_back(std::move(k), value());
                ws();
                if (*p_ == ',') { ++p_; continue; }
                if (*p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
            return v;
        }
        case '[': {
            ++p_;
            v->kind = Value::Arr;
            ws();
            if (*p_ == ']') { ++p_; return v; }
            for (;;) {
                v->arr.push_back(value());
                ws();
                if (*p_ == ',') { ++p_; continue; }
                if (*p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
            return v;
        }
        case '"':
            v->kind = Value::Str;
            v->s = str();
            return v;
        case 't':
            lit("true");
            v->kind = Value::Bool;
            v->b = true;
            return v;
        case 'f':
            lit("false");
            v->kind = Value::Bool;
            return v;
        case 'n':
            lit("null");
            return v;
        default:
            break;
        }
        // integer
        bool neg = false;
        const char* q = p_;
        if (*q == '-') { neg = true; ++q; }
        if (*q < '0' || *q > '9') fail("unexpected character");
        uint64_t acc = 0;
        while (*q >= '0' && *q <= '9') {
            acc = acc * 10 + uint64_t(*q - '0');
            ++q;
        }
        if (*q == '.' || *q == 'e' || *q == 'E') fail("real numbers are not part of this schema");
        p_ = q;
        v->kind = Value::Int;
        if (neg) {
            v->i = -int64_t(acc);
            v->u = uint64_t(v->i);
        } else {
            v->u = acc;
            v->i = int64_t(acc);
        }
        return v;
    }
    void lit(const char* w) {
        for (const char* c = w; *c; ++c, ++p_)
            if (*p_ != *c) fail("bad literal");
    }
    static void put_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back(char(cp));
        else if (cp < 0x800) {
            out.push_back(char(0xC0 | (cp >> 6)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back(char(0xE0 | (cp >> 12)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        } else {
            out.push_back(char(0xF0 | (cp >> 18)));
            out.push_back(char(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t hex4() {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k, ++p_) {
            char c = *p_;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
            else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string str() {
        ++p_;   // opening quote
        std::string out;
        while (*p_ && *p_ != '"') {
            if (*p_ == '\\') {
                ++p_;
                switch (*p_) {
                case 'n': out.push_back('\n'); ++p_; break;
                case 't': out.push_back('\t'); ++p_; break;
                case 'r': out.push_back('\r'); ++p_; break;
                case 'b': out.push_back('\b'); ++p_; break;
                case 'f': out.push_back('\f'); ++p_; break;
                case '/': out.push_back('/'); ++p_; break;
                case '\\': out.push_back('\\'); ++p_; break;
                case '"': out.push_back('"'); ++p_; break;
                case 'u': {
                    ++p_;
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        uint32_t lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
                }
            } else {
                out.push_back(*p_++);
            }
        }
        if (*p_ != '"') fail("unterminated string");
        ++p_;
        return out;
    }
};

inline ValuePtr parse(const char* text) { return Parser(text).parse(); }

inline void escape_into(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\t': out += "\\t"; break;
        case '\r': out += "\\r"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            } else out.push_back(char(c));
        }
    }
    out.push_back('"');
}

}  // namespace orcjson
