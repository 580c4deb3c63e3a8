#include "config.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ie {
namespace {

struct Parser {
    const std::string& s;
    size_t p = 0;
    int depth = 0;
    explicit Parser(const std::string& text) : s(text) {}

    [[noreturn]] void fail(const std::string& what) const {
        throw std::runtime_error("config.json parse error at byte " + std::to_string(p) + ": " + what);
    }
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { ++p; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail(std::string("expected '") + c + "'"); }

    static void utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out += char(cp);
        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
        else { out += char(0xF0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3F)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (p + 4 > s.size()) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = s[p++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
            else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        ws();
        if (p >= s.size() || s[p] != '"') fail("expected a string");
        ++p;
        std::string out;
        for (;;) {
            if (p >= s.size()) fail("unterminated string");
            const char c = s[p++];
            if (c == '"') return out;
            if (static_cast<unsigned char>(c) < 0x20) fail("control character in string");
            if (c != '\\') { out += c; continue; }
            if (p >= s.size()) fail("unterminated escape");
            const char e = s[p++];
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF && p + 1 < s.size() && s[p] == '\\' && s[p + 1] == 'u') {
                        p += 2;
                        const uint32_t lo = hex4();
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else fail("unpaired surrogate");
                    }
                    utf8(out, cp);
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    JsonValue number() {
        const size_t b = p;
        if (p < s.size() && s[p] == '-') ++p;
        if (p >= s.size() || !std::isdigit(static_cast<unsigned char>(s[p]))) fail("bad number");
        if (s[p] == '0') ++p;
        else while (p < s.size() && std::isdigit(static_cast<unsigned char>(s[p]))) ++p;
        if (p < s.size() && s[p] == '.') {
            ++p;
            if (p >= s.size() || !std::isdigit(static_cast<unsigned char>(s[p]))) fail("bad fraction");
            while (p < s.size() && std::isdigit(static_cast<unsigned char>(s[p]))) ++p;
        }
        if (p < s.size() && (s[p] == 'e' || s[p] == 'E')) {
            ++p;
            if (p < s.size() && (s[p] == '+' || s[p] == '-')) ++p;
            if (p >= s.size() || !std::isdigit(static_cast<unsigned char>(s[p]))) fail("bad exponent");
            while (p < s.size() && std::isdigit(static_cast<unsigned char>(s[p]))) ++p;
        }
        JsonValue v;
        v.type = JsonValue::Number;
        v.num = std::strtod(s.substr(b, p - b).c_str(), nullptr);
        return v;
    }
    JsonValue value() {
        if (++depth > 64) fail("nesting too deep");
        ws();
        if (p >= s.size()) fail("unexpected end of input");
        JsonValue v;
        const char c = s[p];
        if (c == '{') {
            ++p;
            v.type = JsonValue::Object;
            if (!eat('}')) {
                do {
                    std::string k = string();
                    expect(':');
                    v.obj.emplace_back(std::move(k), value());
                } while (eat(','));
                expect('}');
            }
        } else if (c == '[') {
            ++p;
            v.type = JsonValue::Array;
            if (!eat(']')) {
                do v.arr.push_back(value()); while (eat(','));
                expect(']');
            }
        } else if (c == '"') {
            v.type = JsonValue::String;
            v.str = string();
        } else if (s.compare(p, 4, "true") == 0) { p += 4; v.type = JsonValue::Bool; v.b = true; }
        else if (s.compare(p, 5, "false") == 0) { p += 5; v.type = JsonValue::Bool; v.b = false; }
        else if (s.compare(p, 4, "null") == 0) { p += 4; v.type = JsonValue::Null; }
        else v = number();
        --depth;
        return v;
    }
};

// Numbers are doubles in JSON: a value outside the target integer's range (1e30, NaN) would be undefined behaviour in the cast; it is
// clamped to +-2^31 (dims, batch sizes and counts are small; the callers clamp further to what they accept).
int64_t clamp_ll(double x) {
    if (!(x == x)) return 0;
    const double lim = 2147483648.0;
    return int64_t(std::llround(x > lim ? lim : (x < -lim ? -lim : x)));
}
std::vector<int64_t> int_list(const JsonValue* v) {
    std::vector<int64_t> out;
    if (v && v->type == JsonValue::Array)
        for (const auto& e : v->arr)
            if (e.type == JsonValue::Number && out.size() < 64) out.push_back(clamp_ll(e.num));
    return out;
}
std::string str_of(const JsonValue* v) { return v && v->type == JsonValue::String ? v->str : std::string(); }
int int_of(const JsonValue* v, int def) {
    if (!v || v->type != JsonValue::Number) return def;
    const int64_t x = clamp_ll(v->num);
    return int(x > 2147483647 ? 2147483647 : x);
}

std::vector<IoConfig> io_list(const JsonValue* v) {
    std::vector<IoConfig> out;
    if (!v || v->type != JsonValue::Array) return out;
    for (const auto& e : v->arr) {
        if (e.type != JsonValue::Object) continue;
        IoConfig io;
        io.name = str_of(e.find("name"));
        io.data_type = str_of(e.find("data_type"));
        io.label_filename = str_of(e.find("label_filename"));
        io.dims = int_list(e.find("dims"));
        io.shape = int_list(e.find("shape"));
        out.push_back(std::move(io));
    }
    return out;
}

}  // namespace

const JsonValue* JsonValue::find(const std::string& key) const {
    const JsonValue* hit = nullptr;
    if (type == Object)
        for (const auto& kv : obj) if (kv.first == key) hit = &kv.second;
    return hit;
}

JsonValue ParseJson(const std::string& text) {
    Parser ps(text);
    JsonValue v = ps.value();
    ps.ws();
    if (ps.p != text.size()) ps.fail("trailing characters after the document");
    return v;
}

EngineConfig ParseEngineConfig(const std::string& json_text) {
    const JsonValue doc = ParseJson(json_text);
    if (doc.type != JsonValue::Object) throw std::runtime_error("config.json parse error: the document must be an object");
    EngineConfig c;
    c.present = true;
    c.name = str_of(doc.find("name"));
    c.version = str_of(doc.find("version"));
    c.platform = str_of(doc.find("platform"));
    c.inputs = io_list(doc.find("inputs"));
    c.outputs = io_list(doc.find("outputs"));
    c.precision = str_of(doc.find("precision"));
    for (auto& ch : c.precision) ch = char(std::tolower(static_cast<unsigned char>(ch)));
    c.gpus = int_of(doc.find("gpus"), 0);
    if (const JsonValue* v = doc.find("uint8_scale"); v && v->type == JsonValue::Number) { c.uint8_scale = float(v->num); c.has_u8 = true; }
    if (const JsonValue* v = doc.find("uint8_bias"); v && v->type == JsonValue::Number) { c.uint8_bias = float(v->num); c.has_u8 = true; }
    if (const JsonValue* v = doc.find("dynamic_batching"); v && v->type == JsonValue::Bool) c.dynamic_batching = v->b;
    c.max_batch_size = int_of(doc.find("max_batch_size"), 0);
    c.batch_window_us = int_of(doc.find("batch_window_us"), -1);
    c.instance_count = int_of(doc.find("instance_count"), 0);
    c.tune_batches = int_list(doc.find("tune_batches"));
    if (const JsonValue* v = doc.find("fp32_split"); v && v->type == JsonValue::Bool) c.fp32_split = v->b;
    return c;
}

EngineConfig LoadEngineConfig(const std::string& model_dir) {
    std::ifstream f(model_dir + "/config.json", std::ios::binary);
    if (!f) return EngineConfig();
    std::stringstream ss;
    ss << f.rdbuf();
    return ParseEngineConfig(ss.str());
}

}  // namespace ie
