// json.hpp -- small JSON value + parser + pretty printer for the CLI hosts.
// Replaces the reference's nlohmann::json use (e.g. server/src/changeCipherDomain.cpp:53-55,123:
// `inFile >> inputJson`, `outFile << std::setw(2) << outputJson`).  Object keys keep sorted order like
// nlohmann's default std::map-backed objects, and the printer matches its 2-space `setw(2)` layout.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mkh {

class Json {
public:
    enum Kind { Null, Bool, Int, Real, Str, Arr, Obj };
    Kind kind = Null;
    bool b = false;
    long long i = 0;
    unsigned long long u = 0;  // integers above LLONG_MAX (OpenFHE moduli in cereal JSON)
    bool is_unsigned = false;
    double d = 0;
    std::string s;
    std::vector<Json> a;
    std::map<std::string, Json> o;

    Json() = default;
    static Json object() { Json j; j.kind = Obj; return j; }
    static Json array() { Json j; j.kind = Arr; return j; }
    Json(const char *v) : kind(Str), s(v) {}
    Json(const std::string &v) : kind(Str), s(v) {}
    Json(std::string &&v) : kind(Str), s(std::move(v)) {}
    Json(double v) : kind(Real), d(v) {}
    Json(long long v) : kind(Int), i(v) {}
    Json(int v) : kind(Int), i(v) {}
    Json(unsigned long long v) : kind(Int), i((long long)v), u(v), is_unsigned(true) {}
    Json(size_t v, int) : kind(Int), i((long long)v) {}

    bool is_object() const { return kind == Obj; }
    bool is_array() const { return kind == Arr; }
    bool is_string() const { return kind == Str; }
    bool is_number() const { return kind == Int || kind == Real; }
    bool contains(const std::string &k) const { return kind == Obj && o.count(k); }

    Json &operator[](const std::string &k) {
        if (kind == Null) kind = Obj;
        if (kind != Obj) throw std::runtime_error("json: not an object");
        return o[k];
    }
    const Json &at(const std::string &k) const {
        if (kind != Obj) throw std::runtime_error("json: not an object (key '" + k + "')");
        auto it = o.find(k);
        if (it == o.end()) throw std::runtime_error("json: missing key '" + k + "'");
        return it->second;
    }
    const Json &at(size_t idx) const {
        if (kind != Arr || idx >= a.size()) throw std::runtime_error("json: bad array index");
        return a[idx];
    }
    void push_back(Json v) {
        if (kind == Null) kind = Arr;
        if (kind != Arr) throw std::runtime_error("json: not an array");
        a.push_back(std::move(v));
    }
    size_t size() const { return kind == Arr ? a.size() : kind == Obj ? o.size() : 0; }

    double as_double() const {
        if (kind == Real) return d;
        if (kind == Int) return is_unsigned ? (double)u : (double)i;
        throw std::runtime_error("json: not a number");
    }
    long long as_int() const {
        if (kind == Int) return i;
        if (kind == Real && d == std::floor(d)) return (long long)d;
        throw std::runtime_error("json: not an integer");
    }
    unsigned long long as_u64() const {
        if (kind == Int) return is_unsigned ? u : (unsigned long long)i;
        throw std::runtime_error("json: not an integer");
    }
    const std::string &as_string() const {
        if (kind != Str) throw std::runtime_error("json: not a string");
        return s;
    }
    bool operator==(const Json &r) const {
        if (kind != r.kind) return is_number() && r.is_number() && as_double() == r.as_double();
        switch (kind) {
            case Null: return true;
            case Bool: return b == r.b;
            case Int: return i == r.i && u == r.u;
            case Real: return d == r.d;
            case Str: return s == r.s;
            case Arr: return a == r.a;
            case Obj: return o == r.o;
        }
        return false;
    }
    bool operator!=(const Json &r) const { return !(*this == r); }

    // ---- parsing
    static Json parse(const std::string &text) {
        Parser p{text.data(), text.data() + text.size()};
        Json v = p.value();
        p.ws();
        if (p.cur != p.end) throw std::runtime_error("json: trailing characters");
        return v;
    }
    static Json parse_file(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::string buf;
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        size_t got = buf.empty() ? 0 : std::fread(&buf[0], 1, buf.size(), f);
        std::fclose(f);
        buf.resize(got);
        return parse(buf);
    }

    // ---- printing (indent < 0: compact)
    void dump(std::string &out, int indent = 2, int depth = 0) const {
        auto nl = [&](int dep) {
            if (indent < 0) return;
            out.push_back('\n');
            out.append((size_t)(dep * indent), ' ');
        };
        switch (kind) {
            case Null: out += "null"; break;
            case Bool: out += b ? "true" : "false"; break;
            case Int: out += is_unsigned ? std::to_string(u) : std::to_string(i); break;
            case Real: {
                if (!std::isfinite(d)) { out += "null"; break; }
                char buf[40];
                std::snprintf(buf, sizeof buf, "%.17g", d);
                // shortest representation that round-trips (what nlohmann's dump produces for doubles)
                for (int prec = 1; prec <= 17; ++prec) {
                    char tmp[40];
                    std::snprintf(tmp, sizeof tmp, "%.*g", prec, d);
                    if (std::strtod(tmp, nullptr) == d) { std::strcpy(buf, tmp); break; }
                }
                out += buf;
                if (!std::strpbrk(buf, ".eEn")) out += ".0";
                break;
            }
            case Str: quote(out, s); break;
            case Arr:
                if (a.empty()) { out += "[]"; break; }
                out.push_back('[');
                for (size_t k = 0; k < a.size(); ++k) {
                    nl(depth + 1);
                    a[k].dump(out, indent, depth + 1);
                    if (k + 1 < a.size()) out.push_back(',');
                }
                nl(depth);
                out.push_back(']');
                break;
            case Obj: {
                if (o.empty()) { out += "{}"; break; }
                out.push_back('{');
                size_t k = 0;
                for (const auto &kv : o) {
                    nl(depth + 1);
                    quote(out, kv.first);
                    out += indent < 0 ? ":" : ": ";
                    kv.second.dump(out, indent, depth + 1);
                    if (++k < o.size()) out.push_back(',');
                }
                nl(depth);
                out.push_back('}');
                break;
            }
        }
    }
    void write_file(const std::string &path, int indent = 2) const {
        std::string out;
        dump(out, indent);
        out.push_back('\n');
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        std::fwrite(out.data(), 1, out.size(), f);
        std::fclose(f);
    }

private:
    static void quote(std::string &out, const std::string &v) {
        out.push_back('"');
        for (unsigned char c : v) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                default:
                    if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out += b; }
                    else out.push_back((char)c);
            }
        }
        out.push_back('"');
    }
    struct Parser {
        const char *cur, *end;
        int depth = 0;                          // nesting of the value being parsed
        static constexpr int MAX_DEPTH = 64;    // OpenFHE's cereal documents nest about 12 deep; input is untrusted
        struct Nest {
            Parser &p;
            explicit Nest(Parser &q) : p(q) {
                if (++p.depth > MAX_DEPTH) p.fail("nesting too deep");
            }
            ~Nest() { --p.depth; }
        };
        void ws() { while (cur < end && (*cur == ' ' || *cur == '\n' || *cur == '\t' || *cur == '\r')) ++cur; }
        [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("json parse error: ") + m); }
        Json value() {
            ws();
            if (cur >= end) fail("unexpected end");
            switch (*cur) {
                case '{': return object();
                case '[': return array();
                case '"': return Json(string());
                case 't': lit("true"); { Json j; j.kind = Bool; j.b = true; return j; }
                case 'f': lit("false"); { Json j; j.kind = Bool; j.b = false; return j; }
                case 'n': lit("null"); return Json();
                default: return number();
            }
        }
        void lit(const char *w) {
            size_t n = std::strlen(w);
            if ((size_t)(end - cur) < n || std::memcmp(cur, w, n)) fail("bad literal");
            cur += n;
        }
        Json object() {
            Nest nest(*this);
            Json j = Json::object();
            ++cur;
            ws();
            if (cur < end && *cur == '}') { ++cur; return j; }
            for (;;) {
                ws();
                if (cur >= end || *cur != '"') fail("expected key");
                std::string k = string();
                ws();
                if (cur >= end || *cur != ':') fail("expected ':'");
                ++cur;
                j.o[std::move(k)] = value();
                ws();
                if (cur < end && *cur == ',') { ++cur; continue; }
                if (cur < end && *cur == '}') { ++cur; return j; }
                fail("expected ',' or '}'");
            }
        }
        Json array() {
            Nest nest(*this);
            Json j = Json::array();
            ++cur;
            ws();
            if (cur < end && *cur == ']') { ++cur; return j; }
            for (;;) {
                j.a.push_back(value());
                ws();
                if (cur < end && *cur == ',') { ++cur; continue; }
                if (cur < end && *cur == ']') { ++cur; return j; }
                fail("expected ',' or ']'");
            }
        }
        std::string string() {
            ++cur;
            const char *start = cur;
            while (cur < end && *cur != '"' && *cur != '\\') ++cur;  // fast path: no escapes (base64 blobs)
            std::string out(start, cur);
            while (cur < end && *cur != '"') {
                if (*cur == '\\') {
                    if (++cur >= end) fail("bad escape");
                    switch (*cur) {
                        case 'n': out.push_back('\n'); break;
                        case 't': out.push_back('\t'); break;
                        case 'r': out.push_back('\r'); break;
                        case 'b': out.push_back('\b'); break;
                        case 'f': out.push_back('\f'); break;
                        case 'u': {
                            if (end - cur < 5) fail("bad \\u");
                            unsigned cp = (unsigned)std::strtoul(std::string(cur + 1, cur + 5).c_str(), nullptr, 16);
                            cur += 4;
                            if (cp < 0x80) out.push_back((char)cp);
                            else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                            else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                            break;
                        }
                        default: out.push_back(*cur);
                    }
                    ++cur;
                } else {
                    out.push_back(*cur++);
                }
            }
            if (cur >= end) fail("unterminated string");
            ++cur;
            return out;
        }
        Json number() {
            const char *start = cur;
            bool real = false;
            if (cur < end && (*cur == '-' || *cur == '+')) ++cur;
            while (cur < end && ((*cur >= '0' && *cur <= '9') || *cur == '.' || *cur == 'e' || *cur == 'E' || *cur == '-' || *cur == '+')) {
                if (*cur == '.' || *cur == 'e' || *cur == 'E') real = true;
                ++cur;
            }
            if (cur == start) fail("unexpected character");
            std::string tok(start, cur);
            Json j;
            if (real) { j.kind = Real; j.d = std::strtod(tok.c_str(), nullptr); return j; }
            j.kind = Int;
            if (tok[0] == '-') { j.i = std::strtoll(tok.c_str(), nullptr, 10); }
            else { j.u = std::strtoull(tok.c_str(), nullptr, 10); j.i = (long long)j.u; j.is_unsigned = j.u > 9223372036854775807ull; }
            return j;
        }
    };
};

}  // namespace mkh
