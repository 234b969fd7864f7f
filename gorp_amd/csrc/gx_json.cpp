// gx_json.cpp -- the small JSON reader / writer behind `append` (definition language) and result materialisation.
//
// The reference parses `append` values with jackson-jr (JSON.std.anyFrom, core/DefinitionReader.java:602-640) and has no
// writer of its own; here a value is validated and re-serialised canonically -- compact, key order kept (LinkedHashMap),
// a repeated key keeps its first position and its last value, numbers / true / false / null verbatim.
#include "gx_dsl.hpp"

#include <cstring>
#include <stdexcept>

namespace gx {
namespace dsl {

namespace {
struct Json {
    const std::string& s;
    size_t at = 0;
    explicit Json(const std::string& text) : s(text) {}
    [[noreturn]] void bad(const std::string& m) { throw std::runtime_error(m + " at offset " + std::to_string(at)); }
    void ws() { while (at < s.size() && (s[at] == ' ' || s[at] == '\t' || s[at] == '\n' || s[at] == '\r')) ++at; }
    std::string value() {
        ws();
        if (at >= s.size()) bad("Unexpected end-of-input");
        char c = s[at];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') return string();
        if (c == 't' || c == 'f' || c == 'n') {
            for (const char* w : {"true", "false", "null"})
                if (s.compare(at, strlen(w), w) == 0) { at += strlen(w); return w; }
            bad("Unrecognized token");
        }
        if (c == '-' || (c >= '0' && c <= '9')) {
            size_t b = at;
            if (s[at] == '-') ++at;
            if (at >= s.size() || !(s[at] >= '0' && s[at] <= '9')) bad("Invalid number");
            while (at < s.size() && s[at] >= '0' && s[at] <= '9') ++at;
            if (at < s.size() && s[at] == '.') { ++at; while (at < s.size() && s[at] >= '0' && s[at] <= '9') ++at; }
            if (at < s.size() && (s[at] == 'e' || s[at] == 'E')) {
                ++at;
                if (at < s.size() && (s[at] == '+' || s[at] == '-')) ++at;
                while (at < s.size() && s[at] >= '0' && s[at] <= '9') ++at;
            }
            return s.substr(b, at - b);
        }
        bad(std::string("Unexpected character ('") + c + "')");
    }
    std::string string() {
        size_t b = at++;
        while (at < s.size() && s[at] != '"') { if (s[at] == '\\') ++at; ++at; }
        if (at >= s.size()) bad("Unexpected end-of-input in a String value");
        ++at;
        return s.substr(b, at - b);
    }
    std::string object() {
        ++at;
        std::vector<std::pair<std::string, std::string>> kv;
        ws();
        if (at < s.size() && s[at] == '}') { ++at; return "{}"; }
        for (;;) {
            ws();
            if (at >= s.size() || s[at] != '"') bad("was expecting double-quote to start field name");
            std::string k = string();
            ws();
            if (at >= s.size() || s[at] != ':') bad("was expecting a colon to separate field name and value");
            ++at;
            std::string v = value();
            bool replaced = false;
            for (auto& e : kv) if (e.first == k) { e.second = v; replaced = true; }
            if (!replaced) kv.push_back({k, v});
            ws();
            if (at < s.size() && s[at] == ',') { ++at; continue; }
            if (at < s.size() && s[at] == '}') { ++at; break; }
            bad("was expecting comma to separate Object entries");
        }
        std::string out = "{";
        for (size_t i = 0; i < kv.size(); ++i) out += (i ? "," : "") + kv[i].first + ":" + kv[i].second;
        return out + "}";
    }
    std::string array() {
        ++at;
        std::string out = "[";
        ws();
        if (at < s.size() && s[at] == ']') { ++at; return "[]"; }
        for (bool first = true;; first = false) {
            out += (first ? "" : ",") + value();
            ws();
            if (at < s.size() && s[at] == ',') { ++at; continue; }
            if (at < s.size() && s[at] == ']') { ++at; break; }
            bad("was expecting comma to separate Array entries");
        }
        return out + "]";
    }
};

}  // namespace

// merge b's entries into a (Map.putAll on LinkedHashMaps); both canonical object texts
std::string json_merge(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    std::string joined = a.substr(0, a.size() - 1) + (a.size() > 2 && b.size() > 2 ? "," : "") + b.substr(1);
    Json j(joined);
    return j.object();
}

// ---------------------------------------------------------------------------
// JSON helpers for result materialisation
// ---------------------------------------------------------------------------
std::string json_quote(const std::string& s) {
    static const char* HEX = "0123456789ABCDEF";
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\t': o += "\\t"; break;
        case '\n': o += "\\n"; break;
        case '\f': o += "\\f"; break;
        case '\r': o += "\\r"; break;
        default:
            if (c < 0x20) { o += "\\u00"; o += HEX[c >> 4]; o += HEX[c & 15]; }
            else o += static_cast<char>(c);
        }
    }
    return o + "\"";
}

namespace {
void append_utf8(std::string& o, uint32_t cp) {
    if (cp < 0x80) o += static_cast<char>(cp);
    else if (cp < 0x800) { o += static_cast<char>(0xC0 | (cp >> 6)); o += static_cast<char>(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) {
        o += static_cast<char>(0xE0 | (cp >> 12)); o += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); o += static_cast<char>(0x80 | (cp & 0x3F));
    } else {
        o += static_cast<char>(0xF0 | (cp >> 18)); o += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
        o += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); o += static_cast<char>(0x80 | (cp & 0x3F));
    }
}
// raw = a JSON string literal including its quotes
std::string json_unquote(const std::string& raw) {
    std::string o;
    auto hex4 = [&](size_t at) {
        uint32_t v = 0;
        for (size_t q = at; q < at + 4 && q < raw.size(); ++q) {
            const char c = raw[q];
            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
        }
        return v;
    };
    for (size_t i = 1; i + 1 < raw.size(); ++i) {
        char c = raw[i];
        if (c != '\\') { o += c; continue; }
        c = raw[++i];
        switch (c) {
        case 'b': o += '\b'; break;
        case 'f': o += '\f'; break;
        case 'n': o += '\n'; break;
        case 'r': o += '\r'; break;
        case 't': o += '\t'; break;
        case 'u': {
            uint32_t cp = hex4(i + 1);
            i += 4;
            if (cp >= 0xD800 && cp <= 0xDBFF && i + 6 < raw.size() && raw[i + 1] == '\\' && raw[i + 2] == 'u') {
                const uint32_t lo = hex4(i + 3);
                if (lo >= 0xDC00 && lo <= 0xDFFF) { cp = 0x10000 + ((cp & 0x3FF) << 10) + (lo & 0x3FF); i += 6; }
            }
            append_utf8(o, cp);
            break;
        }
        default: o += c;  // \" \\ \/
        }
    }
    return o;
}
}  // namespace

std::string canonical_json_object(const std::string& text) {
    try {
        Json j(text);
        j.ws();
        if (j.at >= text.size() || text[j.at] != '{') throw std::runtime_error("not a JSON object");
        std::string canon = j.value();
        j.ws();
        if (j.at != text.size()) throw std::runtime_error("trailing content after the JSON object");
        return canon;
    } catch (std::runtime_error& e) {
        if (dynamic_cast<GxError*>(&e)) throw;
        throw GxError(GX_E_ARG, std::string("invalid JSON object: ") + e.what());
    }
}

std::vector<std::pair<std::string, std::string>> json_object_entries(const std::string& canon) {
    std::vector<std::pair<std::string, std::string>> out;
    if (canon.size() <= 2) return out;
    Json j(canon);
    ++j.at;  // '{'
    for (;;) {
        j.ws();
        const std::string k = j.string();
        j.ws();
        ++j.at;  // ':'
        const std::string v = j.value();
        out.push_back({json_unquote(k), v});
        j.ws();
        if (j.at < canon.size() && canon[j.at] == ',') { ++j.at; continue; }
        break;
    }
    return out;
}

// An `append` value as written in a definition: a JSON object, or a bare "key": value list.  Returns the canonical object
// text; *not_object receives the canonical text of a value that parsed but is no object.  Throws std::runtime_error with
// the parser's message on malformed JSON.
std::string parse_append_value(std::string raw, std::string* not_object) {
    size_t b = raw.find_first_not_of(" \t\r\n\f\v"), e = raw.find_last_not_of(" \t\r\n\f\v");
    raw = (b == std::string::npos) ? "" : raw.substr(b, e - b + 1);
    not_object->clear();
    if (raw.empty()) return "";
    if (raw[0] == '"') raw = "{" + raw + "}";
    Json j(raw);
    j.ws();
    const bool is_object = j.at < raw.size() && raw[j.at] == '{';
    std::string canon = j.value();
    j.ws();
    if (j.at != raw.size()) j.bad("Unexpected trailing content");
    if (!is_object) { *not_object = canon; return ""; }
    return canon;
}

}  // namespace dsl
}  // namespace gx
