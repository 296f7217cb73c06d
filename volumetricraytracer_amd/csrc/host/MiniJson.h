/* MiniJson.h — a small JSON reader (objects, arrays, strings, numbers, true/false/null) for the
 * glTF manifest and the texture-library file.  The reference uses rapidjson + the Microsoft glTF
 * SDK (Voxelizer/Private/Voxelizer.cpp:21-24), neither of which is available here. */
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace minijson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::map<std::string, ValuePtr> obj;

    bool IsObject() const { return kind == Object; }
    bool IsArray() const { return kind == Array; }
    bool IsNumber() const { return kind == Number; }
    bool IsString() const { return kind == String; }
    bool Has(const std::string& k) const { return kind == Object && obj.find(k) != obj.end(); }
    const Value& operator[](const std::string& k) const {
        static const Value null_value;
        auto it = obj.find(k);
        return (kind == Object && it != obj.end()) ? *it->second : null_value;
    }
    const Value& operator[](size_t i) const {
        static const Value null_value;
        return (kind == Array && i < arr.size()) ? *arr[i] : null_value;
    }
    size_t Size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    double GetDouble(double def = 0.0) const { return kind == Number ? num : def; }
    float GetFloat(float def = 0.f) const { return kind == Number ? (float)num : def; }
    long long GetInt(long long def = -1) const { return kind == Number ? (long long)num : def; }
    std::string GetString(const std::string& def = "") const { return kind == String ? str : def; }
};

class Parser {
public:
    explicit Parser(const std::string& text) : s(text) {}
    ValuePtr Parse() {
        ValuePtr v = value(0);
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t p = 0;
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("JSON: ") + what + " at offset " + std::to_string(p)); }
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
    }
    ValuePtr value(int depth) {
        if (depth > 128) fail("nesting too deep");
        ws();
        if (p >= s.size()) fail("unexpected end");
        auto v = std::make_shared<Value>();
        char c = s[p];
        if (c == '{') {
            v->kind = Value::Object;
            p++;
            ws();
            if (p < s.size() && s[p] == '}') { p++; return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p >= s.size() || s[p] != ':') fail("expected ':'");
                p++;
                v->obj[k] = value(depth + 1);
                ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == '}') { p++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = Value::Array;
            p++;
            ws();
            if (p < s.size() && s[p] == ']') { p++; return v; }
            for (;;) {
                v->arr.push_back(value(depth + 1));
                ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == ']') { p++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->str = string();
        } else if (s.compare(p, 4, "true") == 0) {
            v->kind = Value::Bool; v->b = true; p += 4;
        } else if (s.compare(p, 5, "false") == 0) {
            v->kind = Value::Bool; v->b = false; p += 5;
        } else if (s.compare(p, 4, "null") == 0) {
            p += 4;
        } else {
            size_t end = p;
            while (end < s.size() && (isdigit((unsigned char)s[end]) || s[end] == '-' || s[end] == '+' || s[end] == '.' || s[end] == 'e' || s[end] == 'E')) end++;
            if (end == p) fail("unexpected character");
            v->kind = Value::Number;
            try { v->num = std::stod(s.substr(p, end - p)); } catch (...) { fail("bad number"); }
            p = end;
        }
        return v;
    }
    std::string string() {
        if (p >= s.size() || s[p] != '"') fail("expected string");
        p++;
        std::string out;
        while (p < s.size() && s[p] != '"') {
            char c = s[p++];
            if (c == '\\') {
                if (p >= s.size()) fail("bad escape");
                char e = s[p++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (p + 4 > s.size()) fail("bad \\u escape");
                        unsigned cp = (unsigned)std::stoul(s.substr(p, 4), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        if (p >= s.size()) fail("unterminated string");
        p++;
        return out;
    }
};

inline ValuePtr Parse(const std::string& text) { return Parser(text).Parse(); }

}  // namespace minijson
