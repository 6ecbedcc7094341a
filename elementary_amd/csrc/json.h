// json.h — minimal JSON value + recursive-descent parser for instruction batches.
//
// Plays the role of elem::js::Value / parseJSON (runtime/elem/Value.h:37-197, JSON.h:17-156) at
// the C-ABI boundary: every number becomes a double (JSON.h:91-104), objects keep insertion order.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace elemhip {

struct Value {
    enum Type : uint8_t { Undefined, Null, Bool, Number, String, Array, Object };
    Type type = Undefined;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool isNumber() const { return type == Number; }
    bool isString() const { return type == String; }
    bool isBool() const { return type == Bool; }
    bool isArray() const { return type == Array; }
    bool isObject() const { return type == Object; }

    static Value number(double d) { Value v; v.type = Number; v.num = d; return v; }
    static Value boolean(bool x) { Value v; v.type = Bool; v.b = x; return v; }
    static Value string(std::string s) { Value v; v.type = String; v.str = std::move(s); return v; }

    const Value* find(const char* key) const {
        for (auto const& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
public:
    JsonParser(const char* s, size_t n) : p(s), end(s + n) {}

    bool parse(Value& out) {
        skip();
        if (!value(out, 0)) return false;
        skip();
        return p == end;
    }

private:
    const char* p;
    const char* end;

    void skip() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }

    bool lit(const char* w) {
        size_t n = std::strlen(w);
        if ((size_t)(end - p) < n || std::memcmp(p, w, n) != 0) return false;
        p += n;
        return true;
    }

    static void utf8(std::string& s, uint32_t cp) {
        if (cp < 0x80) s += (char)cp;
        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
        else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    }

    bool hex4(uint32_t& v) {
        if (end - p < 4) return false;
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return false;
        }
        return true;
    }

    bool string(std::string& s) {
        if (p >= end || *p != '"') return false;
        ++p;
        while (p < end) {
            char c = *p++;
            if (c == '"') return true;
            if (c == '\\') {
                if (p >= end) return false;
                char e = *p++;
                switch (e) {
                    case '"': s += '"'; break;
                    case '\\': s += '\\'; break;
                    case '/': s += '/'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'n': s += '\n'; break;
                    case 'r': s += '\r'; break;
                    case 't': s += '\t'; break;
                    case 'u': {
                        uint32_t cp;
                        if (!hex4(cp)) return false;
                        if (cp >= 0xD800 && cp <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2;
                            uint32_t lo;
                            if (!hex4(lo)) return false;
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(s, cp);
                        break;
                    }
                    default: return false;
                }
            } else {
                s += c;
            }
        }
        return false;
    }

    bool value(Value& v, int depth) {
        if (depth > 64 || p >= end) return false;
        char c = *p;
        if (c == '[') {
            ++p;
            v.type = Value::Array;
            skip();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                v.arr.emplace_back();
                skip();
                if (!value(v.arr.back(), depth + 1)) return false;
                skip();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return false;
            }
        }
        if (c == '{') {
            ++p;
            v.type = Value::Object;
            skip();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                skip();
                std::string k;
                if (!string(k)) return false;
                skip();
                if (p >= end || *p != ':') return false;
                ++p;
                skip();
                v.obj.emplace_back(std::move(k), Value());
                if (!value(v.obj.back().second, depth + 1)) return false;
                skip();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return false;
            }
        }
        if (c == '"') { v.type = Value::String; return string(v.str); }
        if (c == 't') { v.type = Value::Bool; v.b = true; return lit("true"); }
        if (c == 'f') { v.type = Value::Bool; v.b = false; return lit("false"); }
        if (c == 'n') { v.type = Value::Null; return lit("null"); }
        // number
        const char* s = p;
        if (p < end && *p == '-') ++p;
        bool digits = false;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
            if (*p >= '0' && *p <= '9') digits = true;
            ++p;
        }
        if (!digits) return false;
        std::string tmp(s, p);
        char* e = nullptr;
        v.num = std::strtod(tmp.c_str(), &e);
        if (e == tmp.c_str()) return false;
        v.type = Value::Number;
        return true;
    }
};

// Value -> JSON text (the inverse of JsonParser; numbers with 17 significant digits round-trip every double)
inline void toJson(const Value& v, std::string& out) {
    switch (v.type) {
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Number: {
            if (!std::isfinite(v.num)) { out += "null"; break; }
            char b[40]; std::snprintf(b, sizeof b, "%.17g", v.num); out += b; break;
        }
        case Value::String: {
            out += '"';
            for (unsigned char ch : v.str) {
                if (ch == '"' || ch == '\\') { out += '\\'; out += (char)ch; }
                else if (ch < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", ch); out += b; }
                else out += (char)ch;
            }
            out += '"';
            break;
        }
        case Value::Array: {
            out += '[';
            for (size_t i = 0; i < v.arr.size(); ++i) { if (i) out += ','; toJson(v.arr[i], out); }
            out += ']';
            break;
        }
        case Value::Object: {
            out += '{';
            for (size_t i = 0; i < v.obj.size(); ++i) { if (i) out += ','; toJson(Value::string(v.obj[i].first), out); out += ':'; toJson(v.obj[i].second, out); }
            out += '}';
            break;
        }
        default: out += "null"; break;
    }
}

} // namespace elemhip
