// Minimal JSON reader for the take / scene / material snippets the su_* C ABI accepts
// (reference: std.json.parseFromSlice at src/capi/capi.zig:199-238,341-349).
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace zyg {

struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool        b = false;
    double      num = 0.0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;  // keeps document order like std.json's iterator

    const JsonValue* get(const char* key) const {
        if (Object != type) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    float  f32(float def = 0.f) const { return Number == type ? float(num) : def; }
    float  memberF32(const char* key, float def) const {
        const JsonValue* v = get(key);
        return v ? v->f32(def) : def;
    }
    uint32_t memberU32(const char* key, uint32_t def) const {
        const JsonValue* v = get(key);
        return (v && Number == v->type) ? uint32_t(v->num) : def;
    }
    double memberF64(const char* key, double def) const {
        const JsonValue* v = get(key);
        return (v && Number == v->type) ? v->num : def;
    }
    bool memberBool(const char* key, bool def) const {
        const JsonValue* v = get(key);
        return (v && Bool == v->type) ? v->b : def;
    }
};

class JsonParser {
  public:
    static bool parse(const char* text, JsonValue& out) {
        JsonParser p(text);
        p.ws();
        if (!p.value(out)) return false;
        p.ws();
        return '\0' == *p.s_;
    }

  private:
    explicit JsonParser(const char* s) : s_(s) {}
    const char* s_;

    void ws() {
        while (' ' == *s_ || '\n' == *s_ || '\t' == *s_ || '\r' == *s_) ++s_;
    }
    bool lit(const char* l) {
        const char* p = s_;
        while (*l) {
            if (*p++ != *l++) return false;
        }
        s_ = p;
        return true;
    }
    bool string(std::string& out) {
        if ('"' != *s_) return false;
        ++s_;
        out.clear();
        while (*s_ && '"' != *s_) {
            if ('\\' == *s_) {
                ++s_;
                switch (*s_) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        unsigned cp = 0;
                        for (int i = 0; i < 4; ++i) {
                            ++s_;
                            if (!*s_) return false;
                            cp = cp * 16 + unsigned(std::strtol(std::string(1, *s_).c_str(), nullptr, 16));
                        }
                        out.push_back(cp < 128 ? char(cp) : '?');
                        break;
                    }
                    case '\0': return false;
                    default: out.push_back(*s_); break;
                }
                ++s_;
            } else {
                out.push_back(*s_++);
            }
        }
        if ('"' != *s_) return false;
        ++s_;
        return true;
    }
    bool value(JsonValue& v) {
        ws();
        if ('{' == *s_) {
            ++s_;
            v.type = JsonValue::Object;
            ws();
            if ('}' == *s_) {
                ++s_;
                return true;
            }
            for (;;) {
                ws();
                std::string key;
                if (!string(key)) return false;
                ws();
                if (':' != *s_) return false;
                ++s_;
                JsonValue child;
                if (!value(child)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (',' == *s_) {
                    ++s_;
                    continue;
                }
                if ('}' == *s_) {
                    ++s_;
                    return true;
                }
                return false;
            }
        }
        if ('[' == *s_) {
            ++s_;
            v.type = JsonValue::Array;
            ws();
            if (']' == *s_) {
                ++s_;
                return true;
            }
            for (;;) {
                JsonValue child;
                if (!value(child)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (',' == *s_) {
                    ++s_;
                    continue;
                }
                if (']' == *s_) {
                    ++s_;
                    return true;
                }
                return false;
            }
        }
        if ('"' == *s_) {
            v.type = JsonValue::String;
            return string(v.str);
        }
        if (lit("true")) {
            v.type = JsonValue::Bool;
            v.b = true;
            return true;
        }
        if (lit("false")) {
            v.type = JsonValue::Bool;
            v.b = false;
            return true;
        }
        if (lit("null")) {
            v.type = JsonValue::Null;
            return true;
        }
        char*  end = nullptr;
        double d = std::strtod(s_, &end);
        if (end == s_) return false;
        s_ = end;
        v.type = JsonValue::Number;
        v.num = d;
        return true;
    }
};

}  // namespace zyg
