// gx_dsl.cpp -- native front-end for Gorp's definition language (.grp files).
//
// This is the row SURVEY.md section 8(f) marks "next #1": the step BEFORE the hot path, which turns a
// definition text into the flattened extractions Gorp.construct consumes.  It lets a .grp file drive the GPU
// engine with no JVM.  It mirrors, function for function:
//   core/io/InputLineReader.java:69-150      physical -> logical lines (comments, backslash continuation)
//   core/io/InputLine.java:101-119           (row, column) of an offset in a joined line
//   core/util/TokenHelper.java               keyword / name / inline-pattern tokenising
//   core/DefinitionReader.java:126-181       keyword dispatch and the three tokenising passes
//   core/DefinitionReader.java:189-294,303-392,394-526,528-640   per-declaration parsing, template contents,
//                                            template references and parameters, extraction blocks, append JSON
//   core/model/CookedDefinitions.java:57-132,144-242,255-453      pattern / template / extraction resolution
// Error texts follow the reference's so that its own error tests can be replayed as fixtures.
#include "gx_dsl.hpp"

#include <map>
#include <set>

namespace gx {
namespace dsl {
namespace {

const char* KNOWN_KEYWORDS = "(pattern, template, extract)";
const char* EXTRACTOR_PROPERTIES = "(template, append)";

// ---------------------------------------------------------------------------
// Logical input lines (InputLine / InputLineReader)
// ---------------------------------------------------------------------------
struct Line {
    std::string source_ref;
    int start_row = 0;
    ustr text;
    std::vector<int> joins;  // offsets where a continuation segment begins

    std::string desc(int col) const {  // InputLine.constructDesc
        int row = start_row, column = col, base = 0;
        for (int off : joins) {
            if (column < off) break;
            ++row;
            base = off;
        }
        column -= base;
        return "[" + source_ref + " (" + std::to_string(row) + "," + std::to_string(column + 1) + ")]";
    }
};
typedef std::shared_ptr<Line> LineP;

[[noreturn]] void report(const LineP& line, int offset, const std::string& msg) {
    const std::string d = line ? line->desc(offset) : std::string("N/A");
    throw GxError(GX_E_DEFINITION, "(" + d + "): " + msg);
}

struct LineReader {
    std::vector<ustr> physical;
    size_t next = 0;
    int row = 0;
    std::string source_ref;

    LineReader(const ustr& text, const std::string& ref) : source_ref(ref) {
        // BufferedReader.readLine(): \n, \r or \r\n end a line; a trailing terminator does not start another
        ustr cur;
        bool any = false;
        for (size_t i = 0; i < text.size(); ++i) {
            char16_t c = text[i];
            if (c == u'\n' || c == u'\r') {
                physical.push_back(cur);
                cur.clear();
                any = false;
                if (c == u'\r' && i + 1 < text.size() && text[i + 1] == u'\n') ++i;
            } else { cur.push_back(c); any = true; }
        }
        if (any) physical.push_back(cur);
    }

    static bool empty_or_comment(const ustr& s) {
        for (char16_t c : s) {
            if (c <= 0x20) continue;
            return c == u'#';
        }
        return true;
    }
    [[noreturn]] void io_error(const std::string& msg) {
        throw GxError(GX_E_DEFINITION, "(" + source_ref + ", row " + std::to_string(row) + "): " + msg);
    }

    LineP next_line() {
        ustr s;
        for (;;) {
            if (next >= physical.size()) return nullptr;
            s = physical[next++];
            ++row;
            if (!empty_or_comment(s)) break;
        }
        LineP line(new Line());
        line->source_ref = source_ref;
        line->start_row = row;
        if (s.empty() || s.back() != u'\\') { line->text = s; return line; }
        s.pop_back();
        line->text = s;
        for (;;) {
            // continuation lines are taken verbatim: no comment / blank skipping
            if (next >= physical.size()) io_error("Unexpected end-of-input when expecting line continuation'");
            ustr seg = physical[next++];
            ++row;
            const bool more = !seg.empty() && seg.back() == u'\\';
            if (more) seg.pop_back();
            line->joins.push_back(static_cast<int>(line->text.size()));
            line->text += seg;
            if (!more) return line;
        }
    }
};

// ---------------------------------------------------------------------------
// Definition pieces (core/model/DefPiece and subclasses)
// ---------------------------------------------------------------------------
struct DefPiece;
typedef std::shared_ptr<DefPiece> DP;
struct DefPiece {
    enum Kind { TEXT, PATTERN, PATTERN_REF, TEMPLATE_REF, TEMPLATE_PARAM, EXTRACTOR_PARAM, EXTRACTOR } kind;
    LineP src;
    int off = 0;
    ustr text;             // literal text / pattern / referenced name / extractor name
    int position = -1;     // TEMPLATE_PARAM / EXTRACTOR_PARAM / positional EXTRACTOR
    bool has_params = false;  // TEMPLATE_REF: parameter list present (TemplateReference.takesParameters)
    std::vector<DP> parts;    // EXTRACTOR contents / TEMPLATE_REF parameters
};

DP make(DefPiece::Kind k, const LineP& src, int off, const ustr& text) {
    DP p(new DefPiece());
    p->kind = k; p->src = src; p->off = off; p->text = text;
    return p;
}

// anything pieces can be appended to (UncookedDefinition, ExtractorExpression, TemplateReference, CookedTemplate)
struct Container {
    ustr name;
    std::vector<DP>* parts;
    DefPiece* as_ref = nullptr;  // appending to a TemplateReference creates its parameter list
    void append(const DP& p) {
        if (as_ref) as_ref->has_params = true;
        parts->push_back(p);
    }
};

struct ParamCollector {  // core/model/ParameterCollector.java
    std::string types;
    int count = 0;
    void add(const LineP& src, int src_off, int pos, char type) {
        --pos;
        if (pos >= count) {
            if (pos >= static_cast<int>(types.size())) types.resize(pos + 1, '\0');
            count = pos + 1;
        } else {
            char old = types[pos];
            if (old != type && old != '\0')
                report(src, src_off, std::string("Inconsistent references to parameter ") + std::to_string(pos + 1) + ": " + old + " vs " + type);
        }
        types[pos] = type;
    }
    std::string declarations() const { return types.substr(0, count); }
};

struct Uncooked {  // UncookedDefinition
    LineP src;
    ustr name;
    bool has_params = false;
    ParamCollector params;
    int def_start = 0;
    std::vector<DP> parts;
};
typedef std::shared_ptr<Uncooked> UP;

struct UncookedExtraction {
    LineP src;
    ustr name;
    UP tmpl;
    std::string append_json;  // canonical JSON object text, or empty
};

template <typename V> struct OrderedMap {  // LinkedHashMap: a re-put keeps the key's original slot
    std::vector<ustr> keys;
    std::map<ustr, V> vals;
    bool put(const ustr& k, const V& v) {  // true when the key already existed
        auto it = vals.find(k);
        if (it != vals.end()) { it->second = v; return true; }
        keys.push_back(k);
        vals.emplace(k, v);
        return false;
    }
    const V* find(const ustr& k) const { auto it = vals.find(k); return it == vals.end() ? nullptr : &it->second; }
};

// ---------------------------------------------------------------------------
// TokenHelper
// ---------------------------------------------------------------------------
bool is_ws(char16_t c) { return c <= u' '; }
bool is_digit(char16_t c) { return c >= u'0' && c <= u'9'; }
bool is_word(char16_t c) { return is_digit(c) || (c >= u'a' && c <= u'z') || (c >= u'A' && c <= u'Z') || c == u'_'; }
// Character.isJavaIdentifierStart / Part: exact on ASCII; above it, letters (approximated as >= U+00A0
// except the Latin-1 symbols) and, for Part, also the ignorable control ranges
bool is_ident_start(char16_t c) {
    if (c < 0x80) return (c >= u'a' && c <= u'z') || (c >= u'A' && c <= u'Z') || c == u'_' || c == u'$';
    if (c < 0xA0) return false;
    if (c < 0x100) return c == 0xA2 || c == 0xA3 || c == 0xA4 || c == 0xA5 || c == 0xAA || c == 0xB5 || c == 0xBA ||
                          (c >= 0xC0 && c != 0xD7 && c != 0xF7);
    return true;
}
bool is_ident_part(char16_t c) {
    if (is_ident_start(c) || is_digit(c)) return true;
    return c <= 0x08 || (c >= 0x0E && c <= 0x1B) || (c >= 0x7F && c <= 0x9F) || c == 0xAD;
}
bool is_name_char(char16_t c) { return c == u'-' || is_ident_part(c); }

std::string u8(const ustr& s) { return u16_to_utf8(s); }

std::string char_desc(char16_t c) {  // TokenHelper.charDesc
    char buf[32];
    if (c < 0x20 || (c >= 0x7F && c <= 0x9F)) { snprintf(buf, sizeof buf, "code 0x%04x", static_cast<int>(c)); return buf; }
    snprintf(buf, sizeof buf, " (code 0x%04x)", static_cast<int>(c));
    return "'" + u8(ustr(1, c)) + "'" + buf;
}

struct Tok { ustr match; bool has_match = true; int rest = 0; };

// "\\s*(\\w*)\\s*(.*)": always matches a line without terminators
Tok find_keyword(const ustr& s) {
    size_t i = 0;
    auto re_space = [](char16_t c) { return c == u' ' || (c >= 9 && c <= 13); };
    while (i < s.size() && re_space(s[i])) ++i;
    size_t b = i;
    while (i < s.size() && is_word(s[i])) ++i;
    Tok t;
    t.match = s.substr(b, i - b);
    while (i < s.size() && re_space(s[i])) ++i;
    t.rest = static_cast<int>(i);
    return t;
}

int find_type_marker(char16_t marker, const ustr& s, int ix) {
    for (int end = static_cast<int>(s.size()); ix < end; ++ix) {
        if (s[ix] == marker) return ix;
        if (is_ws(s[ix])) break;
    }
    return -1;
}
int skip_empty_parens(const ustr& s, int ix) {
    if (ix + 1 < static_cast<int>(s.size()) && s[ix] == u'(' && s[ix + 1] == u')') return ix + 2;
    return -1;
}
int skip_space(const ustr& s, int ix) {
    while (ix < static_cast<int>(s.size()) && is_ws(s[ix])) ++ix;
    return ix;
}
int match_remaining(const ustr& s, int ix, char16_t want) {
    bool found = false;
    const int end = static_cast<int>(s.size());
    while (ix < end) {
        char16_t c = s[ix++];
        if (c == want) {
            if (found) break;
            found = true;
        } else if (!is_ws(c)) break;
    }
    return found ? ix : -1;
}

Tok parse_name(const char* type, const LineP& line, const ustr& s, int ix, bool allow_numbers) {
    const int end = static_cast<int>(s.size());
    if (ix >= end) report(line, end, std::string("Missing ") + type + " name");
    Tok t;
    char16_t c = s[ix];
    if (c == u'"' || c == u'\'') {
        ++ix;
        size_t q = s.find(c, ix);
        if (q == ustr::npos)
            report(line, end, std::string("Missing closing quote ('") + static_cast<char>(c) + "') for " + type + " name");
        t.match = s.substr(ix, q - ix);
        t.rest = static_cast<int>(q) + 1;
    } else if (!is_ident_start(c)) {
        t.has_match = false;  // Java: name stays null
        if (is_digit(c)) {
            if (!allow_numbers)
                report(line, ix, std::string("Invalid variable reference instead of ") + type +
                                     " name: can not use variable references here (missing parenthesis after template name?)");
            int b = ix;
            while (ix < end && is_digit(s[ix])) ++ix;
            t.match = s.substr(b, ix - b);
            t.has_match = true;
        }
        t.rest = ix;
    } else {
        int b = ix;
        while (++ix < end && is_name_char(s[ix])) {}
        t.match = s.substr(b, ix - b);
        t.rest = ix;
    }
    return t;
}

Tok parse_name_and_skip_space(const char* type, const LineP& line, const ustr& s, int ix) {
    Tok t = parse_name(type, line, s, ix, false);
    int rest = t.rest;
    const int end = static_cast<int>(s.size());
    if (rest >= end) return t;
    if (!is_ws(s[rest]))
        report(line, rest, std::string("Missing space character after ") + type + " name '" + (t.has_match ? u8(t.match) : "null") + "'");
    while (++rest < end && is_ws(s[rest])) {}
    t.rest = rest;
    return t;
}

Tok parse_inline_pattern(const LineP& line, const ustr& s, int start) {
    const int end = static_cast<int>(s.size());
    int nesting = 1, i = start;
    while (i < end) {
        char16_t c = s[i++];
        if (c == u'\\') { ++i; continue; }
        if (c == u'{') ++nesting;
        else if (c == u'}' && --nesting == 0) {
            Tok t;
            t.match = s.substr(start, i - 1 - start);
            t.rest = i;
            return t;
        }
    }
    report(line, start, "Missing closing '{' for inline pattern");
}

int parse_if_non_negative_number(const ustr& s) {
    if (s.empty()) return -1;
    long v = 0;
    for (char16_t c : s) {
        if (!is_digit(c)) return -1;
        v = v * 10 + (c - u'0');
        if (v > 2000000000L) return 2000000000;
    }
    return static_cast<int>(v);
}

// ---------------------------------------------------------------------------
// Minimal JSON reader for `append` (the reference uses jackson-jr JSON.std.anyFrom): validates the text and
// re-serialises it canonically, preserving key order (LinkedHashMap) and the int / double / bool / null types.
// ---------------------------------------------------------------------------
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

// merge b's entries into a (Map.putAll on LinkedHashMaps); both canonical object texts
std::string json_merge(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    std::string joined = a.substr(0, a.size() - 1) + (a.size() > 2 && b.size() > 2 ? "," : "") + b.substr(1);
    Json j(joined);
    return j.object();
}

// ---------------------------------------------------------------------------
// DefinitionReader
// ---------------------------------------------------------------------------
struct Reader {
    LineReader lines;
    OrderedMap<UP> patterns, templates;
    OrderedMap<std::shared_ptr<UncookedExtraction>> extractions;

    Reader(const ustr& text, const std::string& ref) : lines(text, ref) {}

    // ---- readUncooked (DefinitionReader.java:126-181) ----
    void read_uncooked() {
        LineP line;
        while ((line = lines.next_line())) {
            Tok kw = find_keyword(line->text);
            const std::string k = u8(kw.match);
            if (k == "pattern") read_pattern(line, kw.rest);
            else if (k == "template") read_template(line, kw.rest);
            else if (k == "extract") read_extraction(line, kw.rest);
            else report(line, 0, "Unrecognized keyword \"" + k + "\" encountered; expected one of " + KNOWN_KEYWORDS);
        }
        for (auto& name : patterns.keys) tokenize_pattern(*patterns.find(name));
        for (auto& name : templates.keys) {
            UP t = *templates.find(name);
            Container c{t->name, &t->parts};
            tokenize_template_contents(t->src, t->def_start, c, -1, "template '" + u8(t->name) + "' definition", t->has_params ? &t->params : nullptr);
        }
        for (auto& name : extractions.keys) {
            UP t = (*extractions.find(name))->tmpl;
            Container c{t->name, &t->parts};
            tokenize_template_contents(t->src, t->def_start, c, 0, "extraction template for '" + u8(t->name) + "'", nullptr);
        }
    }

    void read_pattern(const LineP& line, int offset) {
        const ustr& s = line->text;
        int ix = find_type_marker(u'%', s, offset);
        if (ix < 0) report(line, offset, "Pattern name must be prefixed with '%'");
        offset = ix + 1;
        Tok p = parse_name_and_skip_space("pattern", line, s, offset);
        UP u(new Uncooked());
        u->src = line; u->name = p.match; u->def_start = p.rest;
        if (patterns.put(p.match, u)) report(line, offset, "Duplicate pattern definition for name '" + u8(p.match) + "'");
    }

    void tokenize_pattern(const UP& unp) {
        const LineP& line = unp->src;
        const ustr& s = line->text;
        const int end = static_cast<int>(s.size());
        const int offset = unp->def_start;
        size_t pct = s.find(u'%', offset);
        if (pct == ustr::npos) {
            unp->parts.push_back(make(DefPiece::PATTERN, line, offset, s.substr(std::min<size_t>(offset, s.size()))));
            return;
        }
        int ix = static_cast<int>(pct);
        ustr sb;
        if (ix > 0) sb = s.substr(offset, ix - offset);
        int literal_start = offset;
        while (ix < end) {
            char16_t c = s[ix++];
            if (c != u'%') { sb.push_back(c); continue; }
            if (ix == end) report(line, ix, "Orphan '%' at end of pattern '" + u8(unp->name) + "' definition");
            c = s[ix];
            if (c == u'%') { sb.push_back(c); ++ix; continue; }
            Tok ref = parse_name("pattern", line, s, ix, false);
            if (!sb.empty()) { unp->parts.push_back(make(DefPiece::PATTERN, line, literal_start, sb)); sb.clear(); }
            unp->parts.push_back(make(DefPiece::PATTERN_REF, line, ix, ref.match));
            ix = ref.rest;
            literal_start = offset;
        }
        if (!sb.empty()) unp->parts.push_back(make(DefPiece::PATTERN, line, literal_start, sb));
    }

    void read_template(const LineP& line, int start) {
        const ustr& s = line->text;
        int ix = find_type_marker(u'@', s, start);
        if (ix < 0) report(line, start, "Template name must be prefixed with '@'");
        ix += 1;
        Tok p = parse_name("template", line, s, ix, false);
        const int name_off = ix;
        ix = p.rest;
        bool has_params = false;
        int ix2 = skip_empty_parens(s, ix);
        if (ix2 > ix) { ix = ix2; has_params = true; }
        ix2 = skip_space(s, ix);
        if (ix == ix2) report(line, ix, "Missing space character after template name '" + u8(p.match) + "'");
        UP u(new Uncooked());
        u->src = line; u->name = p.match; u->has_params = has_params; u->def_start = ix2;
        if (templates.put(p.match, u)) report(line, name_off, "Duplicate template definition for name '" + u8(p.match) + "'");
    }

    // ---- _tokenizeTemplateContents (DefinitionReader.java:303-392) ----
    int tokenize_template_contents(const LineP& line, int ix, Container& container, int paren_count, const std::string& desc,
                                   ParamCollector* vars) {
        const ustr& s = line->text;
        const int end = static_cast<int>(s.size());
        ustr sb;
        int literal_start = ix;
        while (ix < end) {
            char16_t c = s[ix++];
            if (c == u'%' || c == u'@' || c == u'$') {
                if (ix == end) report(line, ix, std::string("Orphan '") + static_cast<char>(c) + "' at end of " + desc);
                char16_t d = s[ix];
                if (c == d) { sb.push_back(c); ++ix; continue; }  // doubling escapes the sigil
                if (!sb.empty()) { container.append(make(DefPiece::TEXT, line, literal_start, sb)); sb.clear(); }
                if (c == u'%') {
                    Tok p;
                    if (d == u'{') {
                        ++ix;
                        p = parse_inline_pattern(line, s, ix);
                        container.append(make(DefPiece::PATTERN, line, ix, p.match));
                    } else {
                        p = parse_name("pattern", line, s, ix, false);
                        container.append(make(DefPiece::PATTERN_REF, line, ix, p.match));
                    }
                    ix = p.rest;
                } else if (c == u'@') {
                    ix = tokenize_template_reference(line, ix, desc, vars, container);
                } else {
                    Tok p = parse_name("extractor", line, s, ix, vars != nullptr);
                    ix = p.rest;
                    DP extr;
                    int pos;
                    if (vars && (pos = parse_if_non_negative_number(p.match)) >= 0) {
                        if (pos < 1 || pos > 999999)
                            report(line, ix, "Invalid extractor name parameter " + std::to_string(pos) + " in " + desc);
                        vars->add(line, ix, pos, '$');
                        extr = make(DefPiece::EXTRACTOR, line, ix, utf8_to_u16(std::to_string(pos).c_str()));
                        extr->position = pos;
                    } else extr = make(DefPiece::EXTRACTOR, line, ix, p.match);
                    container.append(extr);
                    ix = tokenize_inline_extractor(line, ix, vars, extr);
                }
                literal_start = ix;
                continue;
            }
            if (paren_count > 0) {
                if (c == u'(') ++paren_count;
                else if (c == u')' && --paren_count == 0) break;
            }
            sb.push_back(c);
        }
        if (!sb.empty()) container.append(make(DefPiece::TEXT, line, literal_start, sb));
        if (paren_count > 0) report(line, ix, "Missing closing parenthesis at end of " + desc);
        return ix;
    }

    int tokenize_inline_extractor(const LineP& line, int ix, ParamCollector* vars, const DP& extr) {
        const ustr& s = line->text;
        if (ix >= static_cast<int>(s.size()) || s[ix] != u'(')
            report(line, ix, "Invalid declaration for extractor '" + u8(extr->text) + "': missing opening parenthesis");
        ++ix;
        Container c{extr->text, &extr->parts};
        return tokenize_template_contents(line, ix, c, 1, "extractor '" + u8(extr->text) + "' expression", vars);
    }

    int tokenize_template_reference(const LineP& line, int ix, const std::string& desc, ParamCollector* vars, Container& container) {
        const ustr& s = line->text;
        Tok p = parse_name("template parameter", line, s, ix, vars != nullptr);
        ix = p.rest;
        int pos;
        if (vars && (pos = parse_if_non_negative_number(p.match)) >= 0) {
            if (pos < 1 || pos > 999999) report(line, ix, "Invalid template parameter " + std::to_string(pos) + " in " + desc);
            vars->add(line, ix, pos, '@');
            DP r = make(DefPiece::TEMPLATE_PARAM, line, ix, utf8_to_u16(std::to_string(pos).c_str()));
            r->position = pos;
            container.append(r);
        } else {
            const UP* target = templates.find(p.match);
            if (!target) report(line, ix, "Referencing non-existing template '@" + (p.has_match ? u8(p.match) : "null") + "' from '" + desc + "'");
            DP ref = make(DefPiece::TEMPLATE_REF, line, ix, p.match);
            container.append(ref);
            if ((*target)->has_params) ix = tokenize_parameterized_template(line, ix, desc, vars, ref);
        }
        return ix;
    }

    int tokenize_parameterized_template(const LineP& line, int ix, const std::string& desc, ParamCollector* vars, const DP& ref) {
        const ustr& s = line->text;
        const int end = static_cast<int>(s.size());
        const std::string rname = u8(ref->text);
        if (ix >= end || s[ix] != u'(') report(line, ix, "Missing parameter list for template reference '@" + rname + "'");
        ++ix;
        Container c{ref->text, &ref->parts, ref.get()};
        for (int param = 1; ix < end; ++param) {
            char16_t ch = s[ix++];
            if (ch == u')') return ix;
            if (param > 1) {
                if (ch != u',')
                    report(line, ix, "Unexpected character " + char_desc(ch) + " in template parameter list for '@" + rname +
                                         "': expected either ',' or ')')'");
                if (ix >= end) break;
                ch = s[ix++];
            }
            if (ch == u'@') ix = tokenize_template_reference(line, ix, desc, vars, c);
            else if (ch == u'$') ix = tokenize_extractor_parameter(line, ix, desc, vars, c);
            else
                report(line, ix, "Unexpected character " + char_desc(ch) + " in template parameter list for '@" + rname +
                                     "': expected either type marker '@' or closing ')'");
        }
        report(line, ix, "Unexpected end of line within parameter list for template '@" + rname + "'");
    }

    int tokenize_extractor_parameter(const LineP& line, int ix, const std::string& desc, ParamCollector* vars, Container& container) {
        const ustr& s = line->text;
        Tok p = parse_name("extractor parameter", line, s, ix, vars != nullptr);
        ix = p.rest;
        int pos;
        if (vars && (pos = parse_if_non_negative_number(p.match)) >= 0) {
            if (pos < 1 || pos > 999999) report(line, ix, "Invalid extractor parameter " + std::to_string(pos) + " in " + desc);
            vars->add(line, ix, pos, '$');
            DP r = make(DefPiece::EXTRACTOR_PARAM, line, ix, utf8_to_u16(std::to_string(pos).c_str()));
            r->position = pos;
            container.append(r);
        } else container.append(make(DefPiece::EXTRACTOR, line, ix, p.match));
        return ix;
    }

    // ---- _readExtractionDefinition (DefinitionReader.java:528-594) ----
    void read_extraction(LineP line, int offset) {
        Tok p = parse_name_and_skip_space("extraction", line, line->text, offset);
        const ustr name = p.match;
        const std::string name8 = p.has_match ? u8(name) : "null";
        int ix = match_remaining(line->text, p.rest, u'{');
        if (ix != static_cast<int>(line->text.size()))
            report(line, p.rest, "Unexpected content for extraction '" + name8 + "': expected only opening '{'");
        UP tmpl;
        std::string append;
        for (;;) {
            line = lines.next_line();
            if (!line) lines.io_error("Unexpected end-of-input in extraction '" + name8 + "' definition");
            const ustr& s = line->text;
            ix = match_remaining(s, 0, u'}');
            if (ix >= 0) {
                if (ix >= static_cast<int>(s.size())) break;
                report(line, p.rest, "Unexpected content after closing '}' for extraction '" + name8 + "'");
            }
            ix = skip_space(s, 0);
            p = parse_name_and_skip_space("extraction", line, s, ix);
            ix = p.rest;
            const std::string prop = p.has_match ? u8(p.match) : "null";
            if (prop == "template") {
                if (tmpl) report(line, ix, "More than one 'template' specified for '" + name8 + "'");
                tmpl.reset(new Uncooked());
                tmpl->src = line; tmpl->def_start = ix;
            } else if (prop == "append") {
                append = read_append(line, ix, u8(s.substr(std::min<size_t>(ix, s.size()))), append);
            } else
                report(line, ix, "Unrecognized extraction property \"" + prop + "\" encountered; expected one of " + EXTRACTOR_PROPERTIES);
        }
        if (!tmpl) report(line, ix, "Missing 'template' for extraction '" + name8 + "'");
        std::shared_ptr<UncookedExtraction> x(new UncookedExtraction());
        x->src = line; x->name = name; x->tmpl = tmpl; x->append_json = append;
        extractions.put(name, x);  // a duplicate name silently replaces the earlier one, in its slot
    }

    std::string read_append(const LineP& line, int offset, std::string raw, const std::string& old) {
        size_t b = raw.find_first_not_of(" \t\r\n\f\v"), e = raw.find_last_not_of(" \t\r\n\f\v");
        raw = (b == std::string::npos) ? "" : raw.substr(b, e - b + 1);
        if (raw.empty()) return old;
        if (raw[0] != '{' && raw[0] == '"') raw = "{" + raw + "}";
        std::string canon;
        try {
            Json j(raw);
            j.ws();
            const bool is_object = j.at < raw.size() && raw[j.at] == '{';
            canon = j.value();
            j.ws();
            if (j.at != raw.size()) j.bad("Unexpected trailing content");
            if (!is_object)
                report(line, offset, "Invalid 'append' value: must be JSON Object, or sequence of key/value pairs; was parsed as " + canon);
        } catch (std::runtime_error& e2) {
            if (dynamic_cast<GxError*>(&e2)) throw;
            report(line, offset, std::string("Invalid JSON content to 'append': ") + e2.what());
        }
        return json_merge(old, canon);
    }
};

// ---------------------------------------------------------------------------
// CookedDefinitions
// ---------------------------------------------------------------------------
struct CookedTemplate {
    ustr name;
    LineP src;
    bool has_params = false;
    std::string param_types;
    std::vector<DP> parts;
};
typedef std::shared_ptr<CookedTemplate> CT;

struct Resolver {
    Reader& R;
    std::map<ustr, DP> patterns;      // resolved: one PATTERN piece each
    std::map<ustr, CT> templates;
    explicit Resolver(Reader& r) : R(r) {}

    static std::string stack_desc(const char* marker, const std::vector<ustr>* stack, const ustr& last) {
        if (!stack) return "";
        std::string s = "(";
        for (auto& n : *stack) s += marker + u8(n) + "->";
        return s + marker + u8(last) + ")";
    }
    [[noreturn]] static void piece_error(const DP& p, const std::string& msg) { report(p->src, p->off, msg); }

    // ---- patterns (CookedDefinitions.java:57-132) ----
    void resolve_patterns() {
        for (auto& name : R.patterns.keys) {
            if (patterns.count(name)) continue;
            patterns[name] = resolve_pattern(name, *R.patterns.find(name), nullptr);
        }
    }
    DP resolve_pattern(const ustr& name, const UP& def, std::vector<ustr>* stack) {
        std::vector<ustr> local;
        if (def->parts.size() == 1) {
            const DP& piece = def->parts[0];
            if (piece->kind == DefPiece::PATTERN) return piece;
            return resolve_pattern_ref(name, piece, stack ? *stack : local);
        }
        ustr sb;
        for (auto& piece : def->parts) {
            DP lit = piece->kind == DefPiece::PATTERN ? piece : resolve_pattern_ref(name, piece, stack ? *stack : local);
            sb += lit->text;
        }
        return make(DefPiece::PATTERN, def->src, def->parts.empty() ? 0 : def->parts[0]->off, sb);
    }
    DP resolve_pattern_ref(const ustr& from, const DP& ref, std::vector<ustr>& stack) {
        const ustr& to = ref->text;
        auto it = patterns.find(to);
        if (it != patterns.end()) return it->second;
        stack.push_back(from);
        if (std::find(stack.begin(), stack.end(), to) != stack.end())
            piece_error(ref, "Cyclic pattern reference to '%" + u8(to) + "' " + stack_desc("%", &stack, to));
        const UP* raw = R.patterns.find(to);
        if (!raw) piece_error(ref, "Referencing non-existing pattern '%" + u8(to) + "' " + stack_desc("%", &stack, to));
        DP p = resolve_pattern(to, *raw, &stack);
        patterns[to] = p;
        stack.pop_back();
        return p;
    }

    // ---- templates (CookedDefinitions.java:144-242) ----
    static CT construct(const UP& u) {
        CT t(new CookedTemplate());
        t->name = u->name; t->src = u->src; t->has_params = u->has_params;
        if (u->has_params) t->param_types = u->params.declarations();
        return t;
    }
    void resolve_templates() {
        for (auto& name : R.templates.keys) {
            if (templates.count(name)) continue;
            const UP& u = *R.templates.find(name);
            CT t = construct(u);
            resolve_template_contents(u->name, u->parts, t->name, t->parts, nullptr, name);
            templates[name] = t;
        }
    }
    void resolve_template_contents(const ustr& name, const std::vector<DP>& todo, const ustr& result_name, std::vector<DP>& result,
                                   std::vector<ustr>* stack, const ustr& top) {
        std::vector<ustr> local;
        for (const DP& def : todo) {
            switch (def->kind) {
            case DefPiece::TEXT: case DefPiece::PATTERN: result.push_back(def); break;
            case DefPiece::PATTERN_REF: {
                auto it = patterns.find(def->text);
                if (it == patterns.end())
                    piece_error(def, "Referencing non-existing pattern '%" + u8(def->text) + "' from template '" + u8(top) + "' " +
                                         stack_desc("@", stack, result_name));
                result.push_back(it->second);
                break;
            }
            case DefPiece::TEMPLATE_REF:
                if (def->has_params) result.push_back(def);
                else {
                    CT t = resolve_template_ref(name, def, stack ? *stack : local, top);
                    for (auto& p : t->parts) result.push_back(p);
                }
                break;
            case DefPiece::EXTRACTOR: {
                DP resolved = make(DefPiece::EXTRACTOR, def->src, def->off, def->text);
                resolved->position = def->position;
                resolve_template_contents(name, def->parts, resolved->text, resolved->parts, stack ? stack : &local, top);
                result.push_back(resolved);
                break;
            }
            case DefPiece::TEMPLATE_PARAM: result.push_back(def); break;
            default:
                report(def->src, 0, "Internal error: unexpected definition type when resolving template definition '" + u8(top) + "'");
            }
        }
    }
    CT resolve_template_ref(const ustr& from, const DP& ref, std::vector<ustr>& stack, const ustr& top) {
        const ustr& to = ref->text;
        auto it = templates.find(to);
        if (it != templates.end()) return it->second;
        stack.push_back(from);
        if (std::find(stack.begin(), stack.end(), to) != stack.end())
            piece_error(ref, "Cyclic template reference to '%" + u8(to) + "' " + stack_desc("@", &stack, to));
        const UP* raw = R.templates.find(to);
        if (!raw) piece_error(ref, "Referencing non-existing template '%" + u8(to) + "' " + stack_desc("@", &stack, to));
        CT result = construct(*raw);
        // the reference resolves the (still empty) cooked parts here, so a forward-referenced template
        // contributes nothing at this point; kept as is
        std::vector<DP> none;
        resolve_template_contents(result->name, none, result->name, result->parts, &stack, top);
        templates[to] = result;
        stack.pop_back();
        return result;
    }

    // ---- extractions (CookedDefinitions.java:255-453) ----
    struct Bindings { std::string types; std::vector<DP> bound; DP get(int ix) const { return (ix < 1 || ix > static_cast<int>(bound.size())) ? nullptr : bound[ix - 1]; } };

    std::vector<Extraction> resolve_extractions() {
        std::vector<Extraction> out;
        for (auto& key : R.extractions.keys) {
            const auto& raw = *R.extractions.find(key);
            const UP& rt = raw->tmpl;
            std::vector<DP> tparts;
            resolve_template_contents(rt->name, rt->parts, rt->name, tparts, nullptr, rt->name);
            Extraction x;
            x.name = u8(raw->name);
            x.append_json = raw->append_json;
            std::vector<DP> parts;
            std::set<ustr> seen;
            std::vector<ustr> names;
            resolve_parts(tparts, parts, seen, names, nullptr, /*top_level=*/true);
            for (auto& n : names) x.extractor_names.push_back(u8(n));
            for (auto& p : parts) x.pieces.push_back(convert(p));
            out.push_back(std::move(x));
        }
        return out;
    }

    static Piece convert(const DP& p) {
        Piece q;
        q.kind = p->kind == DefPiece::TEXT ? Piece::TEXT : (p->kind == DefPiece::PATTERN ? Piece::PATTERN : Piece::EXTRACTOR);
        q.text = u8(p->text);
        for (auto& k : p->parts) q.kids.push_back(convert(k));
        return q;
    }

    bool resolve_literal(const DP& part, std::vector<DP>& parts) {
        if (part->kind == DefPiece::TEXT || part->kind == DefPiece::PATTERN) { parts.push_back(part); return true; }
        if (part->kind == DefPiece::PATTERN_REF) {
            auto it = patterns.find(part->text);
            if (it == patterns.end())
                throw GxError(GX_E_DEFINITION, "Internal error: non-existing pattern '%" + u8(part->text) + "', should have been caught earlier");
            parts.push_back(it->second);
            return true;
        }
        return false;
    }

    bool resolve_extractor(const DP& part, std::vector<DP>& parts, std::set<ustr>& seen, std::vector<ustr>& names, const Bindings* b) {
        if (part->kind != DefPiece::EXTRACTOR) return false;
        DP extr = part;
        if (extr->position >= 0) {
            DP p = b ? b->get(extr->position) : nullptr;
            if (!p || p->kind != DefPiece::EXTRACTOR)
                piece_error(part, "Internal error: unexpected extractor parameter (expecting ExtractorExpression)");
            if (p->position >= 0)
                piece_error(part, "Internal error: positional extractor parameter (" + std::to_string(extr->position) +
                                      ") resolves to another positional (" + std::to_string(p->position) + ")");
            DP renamed = make(DefPiece::EXTRACTOR, extr->src, extr->off, p->text);
            renamed->parts = extr->parts;
            extr = renamed;
        }
        if (!seen.insert(extr->text).second) piece_error(part, "Duplicate extractor name ($" + u8(extr->text) + ")");
        names.push_back(extr->text);
        DP out = make(DefPiece::EXTRACTOR, extr->src, extr->off, extr->text);
        resolve_parts(extr->parts, out->parts, seen, names, b, false);
        parts.push_back(out);
        return true;
    }

    // _resolveExtraction (top level) and _resolveExtractionParts (below) in one routine
    void resolve_parts(const std::vector<DP>& input, std::vector<DP>& result, std::set<ustr>& seen, std::vector<ustr>& names,
                       const Bindings* b, bool top_level) {
        for (DP part : input) {
            if (part->kind == DefPiece::TEMPLATE_PARAM) {
                if (top_level)
                    piece_error(part, "Internal error: should not encounter template parameter #" + std::to_string(part->position));
                if (!b) piece_error(part, "Invalid parameter variable reference @" + std::to_string(part->position) + "; template takes no parameters");
                DP param = b->get(part->position);
                if (!param)
                    piece_error(part, "Invalid parameter variable reference @" + std::to_string(part->position) + "; template takes " +
                                          std::to_string(b->types.size()) + " parameters");
                part = param;
            }
            if (resolve_literal(part, result) || resolve_extractor(part, result, seen, names, b)) continue;
            if (part->kind == DefPiece::TEMPLATE_REF) { resolve_template_ref_from_extraction(part, result, seen, names, b); continue; }
            piece_error(part, "Internal error: unrecognized DefPiece");
        }
    }

    void resolve_template_ref_from_extraction(const DP& ref, std::vector<DP>& result, std::set<ustr>& seen, std::vector<ustr>& names,
                                              const Bindings* incoming) {
        auto it = templates.find(ref->text);
        if (it == templates.end()) report(ref->src, ref->off, "Internal error: reference to unknown template '@" + u8(ref->text) + "'");
        const CT& t = it->second;
        Bindings bindings;
        const Bindings* use = nullptr;
        if (t->has_params) {
            const size_t pcount = t->param_types.size();
            if (ref->parts.size() != pcount)
                report(ref->src, ref->off, "Parameter mismatch: template '@" + u8(ref->text) + "' expects " + std::to_string(pcount) +
                                               " parameters; " + std::to_string(ref->parts.size()) + " passed");
            bindings.types = t->param_types;
            size_t i = 0;
            for (const DP& piece : ref->parts) {
                const char exp = t->param_types[i++];
                const bool ok = exp == '@' ? piece->kind == DefPiece::TEMPLATE_REF
                                           : (exp == '$' ? piece->kind == DefPiece::EXTRACTOR : false);
                if (exp != '@' && exp != '$')
                    throw GxError(GX_E_DEFINITION, "Internal error: unrecognized template parameter type");
                if (!ok)
                    report(ref->src, ref->off, std::string("Parameter mismatch: template '@") + u8(ref->text) + "' expects type '" + exp + "' parameter");
                bindings.bound.push_back(resolve_parameters(piece, incoming));
            }
            use = &bindings;
        }
        resolve_parts(t->parts, result, seen, names, use, false);
    }

    DP resolve_parameters(const DP& piece, const Bindings* b) {
        if (piece->kind == DefPiece::TEMPLATE_PARAM || piece->kind == DefPiece::EXTRACTOR_PARAM) {
            DP v = b ? b->get(piece->position) : nullptr;
            if (!v)
                piece_error(piece, "Invalid parameter variable reference @" + std::to_string(piece->position) + "; template has " +
                                       std::to_string(b ? b->types.size() : 0) + " parameters");
            return v;
        }
        if (piece->kind == DefPiece::TEMPLATE_REF) {
            if (!piece->has_params) return piece;
            DP n = make(DefPiece::TEMPLATE_REF, piece->src, piece->off, piece->text);
            n->has_params = true;
            for (auto& p : piece->parts) n->parts.push_back(resolve_parameters(p, b));
            return n;
        }
        if (piece->kind == DefPiece::EXTRACTOR) {
            DP n = make(DefPiece::EXTRACTOR, piece->src, piece->off, piece->text);
            n->position = piece->position;
            for (auto& p : piece->parts) n->parts.push_back(resolve_parameters(p, b));
            return n;
        }
        piece_error(piece, "Internal error: unexpected template parameter type");
    }
};

}  // namespace

std::vector<Extraction> read_definition(const std::string& utf8_text, const std::string& source_ref) {
    Reader reader(utf8_to_u16(utf8_text.c_str()), source_ref);
    reader.read_uncooked();
    if (reader.extractions.keys.empty()) throw GxError(GX_E_DEFINITION, "(N/A): No extraction definitions found from definition");
    Resolver res(reader);
    res.resolve_patterns();
    res.resolve_templates();
    return res.resolve_extractions();
}

void build_regex_strings(const Extraction& x, std::string& automaton_rx, std::string& jdk_rx) {
    ustr a, j;
    std::function<void(const Piece&)> walk = [&](const Piece& p) {
        switch (p.kind) {
        case Piece::PATTERN: {
            const ustr t = utf8_to_u16(p.text.c_str());
            try {
                a += massage_regexp_for_automaton(t);
                j += massage_regexp_for_jdk(t);
            } catch (GxError& e) {
                throw GxError(GX_E_DEFINITION, std::string("Invalid pattern definition, problem (java.lang.IllegalArgumentException): ") + e.what());
            }
            break;
        }
        case Piece::TEXT: {
            const ustr q = quote_literal_as_regexp(utf8_to_u16(p.text.c_str()));
            a += q;
            j += q;
            break;
        }
        case Piece::EXTRACTOR:
            a += u'('; j += u'(';
            for (auto& k : p.kids) walk(k);
            a += u')'; j += u')';
            break;
        }
    };
    for (auto& p : x.pieces) walk(p);
    automaton_rx = u16_to_utf8(a);
    jdk_rx = u16_to_utf8(j);
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

// ---------------------------------------------------------------------------
// JSON views (test support)
// ---------------------------------------------------------------------------
namespace {
std::string jstr(const std::string& s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o += static_cast<char>(c);
        }
    }
    return o + "\"";
}
const char* kind_name(DefPiece::Kind k) {
    switch (k) {
    case DefPiece::TEXT: return "LiteralText";
    case DefPiece::PATTERN: return "LiteralPattern";
    case DefPiece::PATTERN_REF: return "PatternReference";
    case DefPiece::TEMPLATE_REF: return "TemplateReference";
    case DefPiece::TEMPLATE_PARAM: return "TemplateParameterReference";
    case DefPiece::EXTRACTOR_PARAM: return "ExtractorParameterReference";
    case DefPiece::EXTRACTOR: return "ExtractorExpression";
    }
    return "?";
}
std::string dump_piece(const DP& p) {
    std::string o = "{\"class\":" + jstr(kind_name(p->kind)) + ",\"text\":" + jstr(u8(p->text));
    if (p->position >= 0) o += ",\"position\":" + std::to_string(p->position);
    if (p->kind == DefPiece::TEMPLATE_REF) o += std::string(",\"takesParameters\":") + (p->has_params ? "true" : "false");
    if (!p->parts.empty() || p->kind == DefPiece::EXTRACTOR) {
        o += ",\"parts\":[";
        for (size_t i = 0; i < p->parts.size(); ++i) o += (i ? "," : "") + dump_piece(p->parts[i]);
        o += "]";
    }
    return o + "}";
}
std::string dump_parts(const std::vector<DP>& v) {
    std::string o = "[";
    for (size_t i = 0; i < v.size(); ++i) o += (i ? "," : "") + dump_piece(v[i]);
    return o + "]";
}
std::string dump_flat_piece(const Piece& p) {
    static const char* names[] = {"text", "pattern", "extractor"};
    std::string o = std::string("[") + jstr(names[p.kind]) + "," + jstr(p.text);
    if (p.kind == Piece::EXTRACTOR) {
        o += ",[";
        for (size_t i = 0; i < p.kids.size(); ++i) o += (i ? "," : "") + dump_flat_piece(p.kids[i]);
        o += "]";
    }
    return o + "]";
}
}  // namespace

std::string dump_json(const std::string& utf8_text, const std::string& source_ref, const std::string& stage) {
    if (stage == "lines") {  // InputLineReader alone: logical lines with their start rows
        LineReader lr(utf8_to_u16(utf8_text.c_str()), source_ref);
        std::string o = "{\"lines\":[";
        bool first = true;
        while (LineP l = lr.next_line()) {
            o += (first ? "" : ",") + std::string("{\"row\":") + std::to_string(l->start_row) + ",\"rows\":" +
                 std::to_string(1 + l->joins.size()) + ",\"contents\":" + jstr(u8(l->text)) + "}";
            first = false;
        }
        return o + "]}";
    }
    Reader reader(utf8_to_u16(utf8_text.c_str()), source_ref);
    reader.read_uncooked();
    std::string o = "{";
    if (stage == "uncooked") {
        o += "\"patterns\":{";
        for (size_t i = 0; i < reader.patterns.keys.size(); ++i) {
            const UP& u = *reader.patterns.find(reader.patterns.keys[i]);
            o += (i ? "," : "") + jstr(u8(u->name)) + ":" + dump_parts(u->parts);
        }
        o += "},\"templates\":{";
        for (size_t i = 0; i < reader.templates.keys.size(); ++i) {
            const UP& u = *reader.templates.find(reader.templates.keys[i]);
            o += (i ? "," : "") + jstr(u8(u->name)) + ":{\"hasParameters\":" + (u->has_params ? "true" : "false") +
                 ",\"parameterTypes\":" + jstr(u->has_params ? std::string(u->params.declarations().c_str()) : "") +
                 ",\"parts\":" + dump_parts(u->parts) + "}";
        }
        o += "},\"extractions\":{";
        for (size_t i = 0; i < reader.extractions.keys.size(); ++i) {
            const auto& x = *reader.extractions.find(reader.extractions.keys[i]);
            o += (i ? "," : "") + jstr(u8(x->name)) + ":{\"template\":" + dump_parts(x->tmpl->parts) + ",\"append\":" +
                 (x->append_json.empty() ? "null" : x->append_json) + "}";
        }
        return o + "}}";
    }
    Resolver res(reader);
    res.resolve_patterns();
    if (stage == "cooked") {
        res.resolve_templates();
        o += "\"patterns\":{";
        bool first = true;
        for (auto& name : reader.patterns.keys) {
            o += (first ? "" : ",") + jstr(u8(name)) + ":" + jstr(u8(res.patterns[name]->text));
            first = false;
        }
        o += "},\"templates\":{";
        first = true;
        for (auto& name : reader.templates.keys) {
            o += (first ? "" : ",") + jstr(u8(name)) + ":" + dump_parts(res.templates[name]->parts);
            first = false;
        }
        return o + "}}";
    }
    if (reader.extractions.keys.empty()) throw GxError(GX_E_DEFINITION, "(N/A): No extraction definitions found from definition");
    res.resolve_templates();
    std::vector<Extraction> xs = res.resolve_extractions();
    o += "\"extractions\":[";
    for (size_t i = 0; i < xs.size(); ++i) {
        const Extraction& x = xs[i];
        std::string a, j;
        build_regex_strings(x, a, j);
        o += (i ? "," : "") + std::string("{\"name\":") + jstr(x.name) + ",\"pieces\":[";
        for (size_t k = 0; k < x.pieces.size(); ++k) o += (k ? "," : "") + dump_flat_piece(x.pieces[k]);
        o += "],\"extractor_names\":[";
        for (size_t k = 0; k < x.extractor_names.size(); ++k) o += (k ? "," : "") + jstr(x.extractor_names[k]);
        o += "],\"append\":" + (x.append_json.empty() ? std::string("null") : x.append_json) + ",\"automaton_rx\":" + jstr(a) +
             ",\"jdk_rx\":" + jstr(j) + "}";
    }
    return o + "]}";
}

}  // namespace dsl
}  // namespace gx
