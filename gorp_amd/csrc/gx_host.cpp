// gx_host.cpp -- definition-time string rewriting that fixes the two regex
// dialects (host side of the path; runs once per definition).
//
// Mirrors core/util/RegexHelper.java: quoteLiteralAsRegexp (:20-70),
// massageRegexpForAutomaton (:79-182) with _appendCharClass (:184-201),
// massageRegexpForJDK (:210-237).  Gorp._buildExtractor (core/Gorp.java:94-129)
// concatenates their outputs into the strings gx_create_from_patterns takes.
#include "gx_common.hpp"

namespace gx {

ustr quote_literal_as_regexp(const ustr& text) {
    static const ustr special = u"()[]\\{}|*?+$^<>\"&";
    ustr out;
    size_t i = 0;
    while (i < text.size()) {
        const char16_t c = text[i++];
        if (c == u' ' || c == u'\t') {
            // a blank swallows the rest of the run (anything <= ' ') and becomes "one or more blanks"
            while (i < text.size() && text[i] <= u' ') ++i;
            out += u"[ \t]+";
        } else if (c == u'.') {
            out += u"\\.";
        } else {
            if (special.find(c) != ustr::npos) out += u'\\';
            out += c;
        }
    }
    return out;
}

namespace {

// Character.isAlphabetic(c) || Character.isDigit(c) for the cases that can matter:
// exact on ASCII; Latin-1 letters and everything above U+00FF are treated as alphabetic.
bool alnum_like(char16_t c) {
    if (c < 0x80) return (c >= u'0' && c <= u'9') || (c >= u'A' && c <= u'Z') || (c >= u'a' && c <= u'z');
    if (c == 0xAA || c == 0xB5 || c == 0xBA) return true;
    if (c >= 0xC0 && c <= 0xFF) return c != 0xD7 && c != 0xF7;
    return c > 0xFF;
}

struct NamedClass { char16_t letter; const char16_t* members; };
const NamedClass kClasses[] = {
    {u'd', u"0-9"},
    {u's', u" \b\f\n\r\t"},
    {u'w', u"a-zA-Z_0-9"},
};

}  // namespace

ustr massage_regexp_for_automaton(const ustr& pattern) {
    if (pattern.find(u'\\') == ustr::npos) return pattern;
    ustr out;
    int depth = 0;  // bracket nesting as the reference counts it (no escape awareness for '[' / ']')
    for (size_t i = 0; i < pattern.size();) {
        const char16_t c = pattern[i++];
        if (c == u'[') { ++depth; out += c; continue; }
        if (c == u']') { --depth; out += c; continue; }
        if (c != u'\\' || i >= pattern.size()) { out += c; continue; }
        const bool right_after_open = depth > 0 && pattern[i - 2] == u'[';
        char16_t d = pattern[i++];
        // named classes, lower case = members, upper case = negated
        const NamedClass* named = nullptr;
        bool negated = false;
        for (const NamedClass& nc : kClasses) {
            if (d == nc.letter) { named = &nc; break; }
            if (d == nc.letter - 32) { named = &nc; negated = true; break; }
        }
        if (named) {
            if (depth == 0) {
                out += u'[';
                if (negated) out += u'^';
                out += named->members;
                out += u']';
            } else {
                if (negated && !right_after_open) {
                    std::string m = "Can not use negated character class \\";
                    m += static_cast<char>(d);
                    m += " within character class in position other than first (Automaton limitation)";
                    throw GxError(GX_E_REGEX_SYNTAX, m);
                }
                if (negated) out += u'^';
                out += named->members;
            }
            continue;
        }
        switch (d) {
        case u'\\': break;
        case u'b': d = u'\b'; break;
        case u'f': d = u'\f'; break;
        case u'n': d = u'\n'; break;
        case u'r': d = u'\r'; break;
        case u't': d = u'\t'; break;
        default:
            if (alnum_like(d)) {
                throw GxError(GX_E_REGEX_SYNTAX,
                              "Unrecognized backslash escape '\\" + u16_to_utf8(ustr(1, d)) +
                                  "; can only escape backslash (\\\\), use known control-codes (\\n, \\r, \\t),"
                                  " escape non-alphanumeric (\\$, \\(, ...) or refer to a 'well-known' character class"
                                  " (\\s, \\S, \\d, \\D, \\w, \\W)");
            }
        }
        out += c;
        out += d;
    }
    return out;
}

ustr massage_regexp_for_jdk(const ustr& pattern) {
    ustr out;
    for (size_t i = 0; i < pattern.size(); ++i) {
        const char16_t c = pattern[i];
        if (c == u'\\') {
            out += c;
            if (i + 1 < pattern.size()) out += pattern[++i];
        } else if (c == u'(') {
            out += u"(?:";  // pattern-internal groups must not capture: only extractors do
        } else out += c;
    }
    return out;
}

}  // namespace gx
