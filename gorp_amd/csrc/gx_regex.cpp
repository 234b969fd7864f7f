// gx_regex.cpp -- the two regex front-ends of the gorp_amd table compiler.
//
// Gorp hands every extraction to two engines (core/Gorp.java:58-79,94-129):
//   * an "automaton" string for dk.brics.automaton.RegExp with flags NONE
//     (core/autom/PolyMatcher.java:58,76) -> decides WHICH extraction matches;
//   * a "JDK" string for java.util.regex.Pattern.compile
//     (core/jdkre/JDKRegexpExtractionCooker.java:23) -> yields the capture groups.
// Both are parsed here into one AST type so that the DFA and the tagged-DFA
// builders share the rest of the pipeline.  Dialect differences that matter
//  live entirely in this file.
#include "gx_common.hpp"

namespace gx {

// ----------------------------------------------------------------------------
// UTF-8 <-> UTF-16 (patterns arrive as UTF-8 over the C ABI; Java sees UTF-16)
// ----------------------------------------------------------------------------
ustr utf8_to_u16(const char* s) {
    ustr out;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
    while (*p) {
        uint32_t cp = 0;
        int extra = 0;
        if (*p < 0x80) { cp = *p; }
        else if ((*p & 0xE0) == 0xC0) { cp = *p & 0x1F; extra = 1; }
        else if ((*p & 0xF0) == 0xE0) { cp = *p & 0x0F; extra = 2; }
        else if ((*p & 0xF8) == 0xF0) { cp = *p & 0x07; extra = 3; }
        else throw GxError(GX_E_ARG, "pattern is not valid UTF-8");
        ++p;
        for (int i = 0; i < extra; ++i, ++p) {
            if ((*p & 0xC0) != 0x80) throw GxError(GX_E_ARG, "pattern is not valid UTF-8");
            cp = (cp << 6) | (*p & 0x3F);
        }
        if (cp > 0xFFFF) {
            cp -= 0x10000;
            out.push_back(static_cast<char16_t>(0xD800 | (cp >> 10)));
            out.push_back(static_cast<char16_t>(0xDC00 | (cp & 0x3FF)));
        } else out.push_back(static_cast<char16_t>(cp));
    }
    return out;
}

std::string u16_to_utf8(const ustr& s) {
    std::string out;
    for (size_t i = 0; i < s.size(); ++i) {
        uint32_t cp = s[i];
        if (cp >= 0xD800 && cp <= 0xDBFF && i + 1 < s.size() && s[i + 1] >= 0xDC00 && s[i + 1] <= 0xDFFF) {
            cp = 0x10000 + ((cp & 0x3FF) << 10) + (s[i + 1] & 0x3FF);
            ++i;
        }
        if (cp < 0x80) out += static_cast<char>(cp);
        else if (cp < 0x800) { out += static_cast<char>(0xC0 | (cp >> 6)); out += static_cast<char>(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) {
            out += static_cast<char>(0xE0 | (cp >> 12));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        } else {
            out += static_cast<char>(0xF0 | (cp >> 18));
            out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        }
    }
    return out;
}

namespace {

AstP node(Ast::Kind k) { AstP n(new Ast()); n->kind = k; return n; }
AstP set_node(CharSet s) { s.canon(); AstP n = node(Ast::SET); n->set = std::move(s); return n; }
AstP lit_node(int c) { return set_node(CharSet::single(c)); }

AstP cat2(AstP a, AstP b) {
    AstP n = node(Ast::CAT);
    n->kids.push_back(std::move(a));
    n->kids.push_back(std::move(b));
    return n;
}

AstP rep_node(AstP body, int mn, int mx, bool greedy) {
    AstP n = node(Ast::REP);
    n->min = mn; n->max = mx; n->greedy = greedy;
    n->kids.push_back(std::move(body));
    return n;
}

const int REPEAT_CAP = 100000;

// ============================================================================
// Automaton dialect
// ============================================================================
// Grammar of dk.brics.automaton.RegExp with syntax flags NONE, as published:
//   union  := concat ( '|' union )?
//   concat := repeat ( concat )?            -- continues while next is not ')' or '|'
//   repeat := class ( '?' | '*' | '+' | '{n}' | '{n,}' | '{n,m}' )*
//   class  := '[' '^'? item+ ']' | simple
//   simple := '.' | '"' chars '"' | '()' | '(' union ')' | char
//   char   := '\'? any
// A consequence worth spelling out: every concat starts by parsing one atom
// unconditionally, so a metacharacter in that position is a literal.
class AutomatonParser {
public:
    explicit AutomatonParser(const ustr& s) : src_(s) {}

    AstP run() {
        if (src_.empty()) return node(Ast::EMPTY);
        AstP e = alternation();
        if (at_ < src_.size()) fail("end-of-string expected at position " + std::to_string(at_));
        return e;
    }

private:
    const ustr& src_;
    size_t at_ = 0;

    [[noreturn]] void fail(const std::string& m) { throw GxError(GX_E_REGEX_SYNTAX, m); }
    bool done() const { return at_ >= src_.size(); }
    bool looking_at(char16_t c) const { return !done() && src_[at_] == c; }
    bool eat(char16_t c) { if (looking_at(c)) { ++at_; return true; } return false; }
    bool digit_here() const { return !done() && src_[at_] >= u'0' && src_[at_] <= u'9'; }
    char16_t take() {
        if (done()) fail("unexpected end-of-string");
        return src_[at_++];
    }
    char16_t take_char() { eat(u'\\'); return take(); }

    AstP alternation() {
        AstP left = sequence();
        if (!eat(u'|')) return left;
        AstP alt = node(Ast::ALT);
        alt->kids.push_back(std::move(left));
        alt->kids.push_back(alternation());
        return alt;
    }

    AstP sequence() {
        AstP left = quantified();
        if (!done() && !looking_at(u')') && !looking_at(u'|')) return cat2(std::move(left), sequence());
        return left;
    }

    int number() {
        size_t b = at_;
        long v = 0;
        while (digit_here()) {
            v = v * 10 + (src_[at_++] - u'0');
            if (v > REPEAT_CAP) throw GxError(GX_E_LIMIT, "repeat count too large");
        }
        if (b == at_) fail("integer expected at position " + std::to_string(at_));
        return static_cast<int>(v);
    }

    AstP quantified() {
        AstP e = bracket_or_simple();
        for (;;) {
            if (eat(u'?')) e = rep_node(std::move(e), 0, 1, true);
            else if (eat(u'*')) e = rep_node(std::move(e), 0, -1, true);
            else if (eat(u'+')) e = rep_node(std::move(e), 1, -1, true);
            else if (eat(u'{')) {
                int n = number();
                int m = n;
                if (eat(u',')) m = digit_here() ? number() : -1;
                if (!eat(u'}')) fail("expected '}' at position " + std::to_string(at_));
                if (m >= 0 && n > m) e = node(Ast::FAIL);  // Automaton.repeat(min,max): empty language
                else e = rep_node(std::move(e), n, m, true);
            } else return e;
        }
    }

    AstP bracket_or_simple() {
        if (!eat(u'[')) return simple();
        bool neg = eat(u'^');
        CharSet cs;
        do {
            char16_t lo = take_char();
            if (eat(u'-')) {
                if (looking_at(u']')) { cs.add(lo, lo); cs.add(u'-', u'-'); }
                else { char16_t hi = take_char(); cs.add(lo, hi); }  // lo > hi adds nothing
            } else cs.add(lo, lo);
        } while (!done() && !looking_at(u']'));
        cs.canon();
        if (neg) cs = cs.negated();
        if (!eat(u']')) fail("expected ']' at position " + std::to_string(at_));
        if (cs.empty()) return node(Ast::FAIL);
        return set_node(cs);
    }

    AstP simple() {
        if (eat(u'.')) return set_node(CharSet::all());
        if (eat(u'"')) {
            size_t b = at_;
            while (!done() && !looking_at(u'"')) ++at_;
            if (!eat(u'"')) fail("expected '\"' at position " + std::to_string(at_));
            AstP seq = node(Ast::CAT);
            for (size_t i = b; i + 1 < at_; ++i) seq->kids.push_back(lit_node(src_[i]));
            if (seq->kids.empty()) return node(Ast::EMPTY);
            return seq;
        }
        if (eat(u'(')) {
            if (eat(u')')) return node(Ast::EMPTY);
            AstP inner = alternation();
            if (!eat(u')')) fail("expected ')' at position " + std::to_string(at_));
            AstP g = node(Ast::GROUP);
            g->cap = 0;
            g->kids.push_back(std::move(inner));
            return g;
        }
        return lit_node(take_char());
    }
};

// ============================================================================
// JDK dialect
// ============================================================================
// java.util.regex.Pattern with no flags, restricted to what Gorp documents as
// supported (README.md:197-224): literals, escaped punctuation, \d\D\s\S\w\W,
// control escapes, '.', bracket classes, (?:..), capturing groups, '|',
// greedy and reluctant quantifiers.  Anything else that Pattern would accept
// is refused with GX_E_UNSUPPORTED_CONSTRUCT rather than approximated.
class JdkParser {
public:
    explicit JdkParser(const ustr& s) : src_(s) {}
    int groups = 0;

    AstP run() {
        AstP e = alternation();
        if (at_ < src_.size()) syntax(src_[at_] == u')' ? "Unmatched closing ')'" : "unexpected character");
        return e;
    }

private:
    const ustr& src_;
    size_t at_ = 0;

    [[noreturn]] void syntax(const std::string& m) {
        throw GxError(GX_E_REGEX_SYNTAX, m + " near index " + std::to_string(at_));
    }
    [[noreturn]] void unsupported(const std::string& m) {
        throw GxError(GX_E_UNSUPPORTED_CONSTRUCT, "unsupported java.util.regex construct: " + m + " near index " + std::to_string(at_));
    }
    bool done() const { return at_ >= src_.size(); }
    int cur() const { return done() ? -1 : src_[at_]; }
    int ahead(size_t k) const { return at_ + k < src_.size() ? src_[at_ + k] : -1; }
    static bool is_digit(int c) { return c >= '0' && c <= '9'; }
    static int hex(int c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    static CharSet digits() { return CharSet::range('0', '9'); }
    static CharSet word() { CharSet s; s.add('a', 'z'); s.add('A', 'Z'); s.add('_', '_'); s.add('0', '9'); s.canon(); return s; }
    static CharSet space() { CharSet s; s.add(' ', ' '); s.add('\t', '\r'); s.canon(); return s; }  // [ \t\n\x0B\f\r]
    static CharSet dot() {
        CharSet s; s.add('\n', '\n'); s.add('\r', '\r'); s.add(0x85, 0x85); s.add(0x2028, 0x2029); s.canon();
        return s.negated();
    }

    AstP alternation() {
        AstP first = sequence();
        if (cur() != '|') return first;
        AstP alt = node(Ast::ALT);
        alt->kids.push_back(std::move(first));
        while (cur() == '|') { ++at_; alt->kids.push_back(sequence()); }
        return alt;
    }

    AstP sequence() {
        AstP seq = node(Ast::CAT);
        while (!done() && cur() != '|' && cur() != ')') seq->kids.push_back(quantified(atom()));
        if (seq->kids.empty()) return node(Ast::EMPTY);
        if (seq->kids.size() == 1) return std::move(seq->kids[0]);
        return seq;
    }

    int number() {
        long v = 0;
        if (!is_digit(cur())) syntax("Illegal repetition");
        while (is_digit(cur())) {
            v = v * 10 + (src_[at_++] - u'0');
            if (v > REPEAT_CAP) throw GxError(GX_E_LIMIT, "repeat count too large");
        }
        return static_cast<int>(v);
    }

    // Can the sub-expression match the empty string / does it hold a capturing group?
    static bool nullable(const Ast& n) {
        switch (n.kind) {
        case Ast::EMPTY: return true;
        case Ast::FAIL: case Ast::SET: return false;
        case Ast::CAT: for (auto& k : n.kids) if (!nullable(*k)) return false; return true;
        case Ast::ALT: for (auto& k : n.kids) if (nullable(*k)) return true; return false;
        case Ast::REP: return n.min == 0 || nullable(*n.kids[0]);
        case Ast::GROUP: return nullable(*n.kids[0]);
        }
        return false;
    }
    static bool captures(const Ast& n) {
        if (n.kind == Ast::GROUP && n.cap > 0) return true;
        for (auto& k : n.kids) if (captures(*k)) return true;
        return false;
    }

    // One quantifier per atom, as java.util.regex has it: Pattern.sequence() meets a second '*', '+' or '?' as a
    // "Dangling meta character" (PatternSyntaxException), and a second '{' would quantify an empty literal -- refused
    // rather than imitated.  A loop around a capturing body that can match the empty string is refused too: there
    // java.util.regex lets an empty last iteration move the group (Loop / GroupTail: "(a*)*" on "aaa" leaves group 1
    // at (3,3)), a rule the tagged automaton does not implement; Gorp's own extractor groups are never quantified.
    AstP quantified(AstP a) {
        for (int count = 0;; ++count) {
            int mn, mx;
            int c = cur();
            if (count == 1) {
                if (c == '?' || c == '*' || c == '+') syntax(std::string("Dangling meta character '") + static_cast<char>(c) + "'");
                if (c == '{') unsupported("a quantifier applied to a quantifier");
                return a;
            }
            if (c == '?') { ++at_; mn = 0; mx = 1; }
            else if (c == '*') { ++at_; mn = 0; mx = -1; }
            else if (c == '+') { ++at_; mn = 1; mx = -1; }
            else if (c == '{') {
                ++at_;
                mn = number();
                mx = mn;
                if (cur() == ',') {
                    ++at_;
                    if (cur() == '}') mx = -1;
                    else { mx = number(); if (mx < mn) syntax("Illegal repetition range"); }
                }
                if (cur() != '}') syntax("Unclosed counted closure");
                ++at_;
            } else return a;
            bool greedy = true;
            if (cur() == '?') { ++at_; greedy = false; }
            else if (cur() == '+') unsupported("possessive quantifier");
            if (mx != 1 && mx != 0 && captures(*a) && nullable(*a))
                unsupported("a repeated capturing group that can match the empty string");
            a = rep_node(std::move(a), mn, mx, greedy);
        }
    }

    // Text after a backslash.  Returns true when it denotes a class (in cls),
    // false when it denotes one code unit (in ch).
    bool escape(CharSet& cls, int& ch) {
        if (done()) syntax("trailing backslash");
        int c = src_[at_++];
        switch (c) {
        case 'd': cls = digits(); return true;
        case 'D': cls = digits().negated(); return true;
        case 's': cls = space(); return true;
        case 'S': cls = space().negated(); return true;
        case 'w': cls = word(); return true;
        case 'W': cls = word().negated(); return true;
        case 't': ch = '\t'; return false;
        case 'n': ch = '\n'; return false;
        case 'r': ch = '\r'; return false;
        case 'f': ch = '\f'; return false;
        case 'a': ch = 0x07; return false;
        case 'e': ch = 0x1B; return false;
        case '0': {
            int v = 0, k = 0;
            while (k < 3 && cur() >= '0' && cur() <= '7' && v * 8 + (cur() - '0') <= 0377) { v = v * 8 + (cur() - '0'); ++at_; ++k; }
            if (k == 0) syntax("Illegal octal escape sequence");
            ch = v; return false;
        }
        case 'x': {
            if (cur() == '{') unsupported("\\x{...}");
            int h1 = hex(cur()), h2 = hex(ahead(1));
            if (h1 < 0 || h2 < 0) syntax("Illegal hexadecimal escape sequence");
            at_ += 2; ch = h1 * 16 + h2; return false;
        }
        case 'u': {
            int v = 0;
            for (int i = 0; i < 4; ++i) {
                int h = hex(ahead(i));
                if (h < 0) syntax("Illegal Unicode escape sequence");
                v = v * 16 + h;
            }
            at_ += 4; ch = v; return false;
        }
        case 'c':
            if (done()) syntax("Illegal control escape sequence");
            ch = src_[at_++] ^ 64; return false;
        default:
            if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '1' && c <= '9')) {
                --at_;
                unsupported(std::string("escape \\") + static_cast<char>(c));
            }
            ch = c; return false;
        }
    }

    AstP atom() {
        int c = cur();
        switch (c) {
        case '(': {
            ++at_;
            AstP g = node(Ast::GROUP);
            if (cur() == '?') {
                if (ahead(1) != ':') unsupported("special group (?...)");
                at_ += 2;
                g->cap = 0;
            } else g->cap = ++groups;
            g->kids.push_back(alternation());
            if (cur() != ')') syntax("Unclosed group");
            ++at_;
            return g;
        }
        case '[': ++at_; return bracket();
        case '.': ++at_; return set_node(dot());
        case '^': unsupported("anchor '^'");
        case '$': unsupported("anchor '$'");
        case '*': case '+': case '?': syntax(std::string("Dangling meta character '") + static_cast<char>(c) + "'");
        case '{': syntax("Illegal repetition");
        case '\\': {
            ++at_;
            CharSet cls; int ch = 0;
            if (escape(cls, ch)) return set_node(cls);
            return lit_node(ch);
        }
        default: ++at_; return lit_node(c);
        }
    }

    AstP bracket() {
        bool neg = false;
        if (cur() == '^') { neg = true; ++at_; }
        CharSet cs;
        for (bool first = true;; first = false) {
            if (done()) syntax("Unclosed character class");
            int c = cur();
            if (c == ']' && !first) { ++at_; break; }
            if (c == '[') unsupported("nested character class");
            if (c == '&' && ahead(1) == '&') unsupported("character class intersection");
            int lo;
            ++at_;
            if (c == '\\') {
                CharSet cls; int ch = 0;
                if (escape(cls, ch)) { cs.add(cls); continue; }
                lo = ch;
            } else lo = c;
            if (cur() == '-' && ahead(1) != -1 && ahead(1) != ']' && ahead(1) != '[') {
                ++at_;
                int d = src_[at_++];
                int hi = d;
                if (d == '\\') {
                    CharSet cls; int ch = 0;
                    if (escape(cls, ch)) syntax("Illegal character range");
                    hi = ch;
                }
                if (hi < lo) syntax("Illegal character range");
                cs.add(lo, hi);
            } else cs.add(lo, lo);
        }
        cs.canon();
        if (neg) cs = cs.negated();
        if (cs.empty()) return node(Ast::FAIL);
        return set_node(cs);
    }
};

}  // namespace

Parsed parse_automaton_dialect(const ustr& src) {
    AutomatonParser p(src);
    Parsed out;
    out.root = p.run();
    out.ngroups = 0;
    return out;
}

Parsed parse_jdk_dialect(const ustr& src) {
    JdkParser p(src);
    Parsed out;
    out.root = p.run();
    out.ngroups = p.groups;
    return out;
}

}  // namespace gx
