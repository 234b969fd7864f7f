// =============================================================================
// gorp_oracle.cpp -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.
//
// CPU restatement of the salesforce/gorp match-and-extract hot path, used as
// the parity oracle for the HIP implementation in gorp_amd/csrc and as the
// "cpu_baseline" leg of bench.py.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product
// (gorp_amd/) never links, imports or calls it.
//
// Parity pinning: the reference is Java and no JVM / jar exists in this image,
// so the reference itself cannot be run here (oracle/_ref is unbuildable).
// This restatement is pinned by the reference's own known-answer tests,
// transcribed as data under tests/golden/ (see tests/test_oracle_golden.py).
//
// What is restated (paths relative to /root/reference,
//   core/ = gorp-core/src/main/java/com/salesforce/gorp/):
//   * core/util/RegexHelper.java:20-70    quoteLiteralAsRegexp
//   * core/util/RegexHelper.java:79-201   massageRegexpForAutomaton/_appendCharClass
//   * core/util/RegexHelper.java:210-237  massageRegexpForJDK
//   * core/autom/PolyMatcher.java:72-84   createAutomaton (RegExp(ptn,NONE) -> minimize)
//   * core/autom/Automata.java:45-55      alphabet(points)
//   * core/autom/Automata.java:57-124     construct (product BFS)
//   * core/autom/Automata.java:133-139    step / accept
//   * core/autom/Automata.java:150-166    pointsUnion
//   * core/autom/PolyState.java:46-77     isNull / step / toAcceptValues
//   * core/autom/PolyMatcher.java:123-133 match (hot loop #1)
//   * core/jdkre/JDKRegexpCookedExtraction.java:36-59  match/_constructMatch (hot loop #2)
//   * core/Gorp.java:159-186              extract decision tree
// Third-party arithmetic that is NOT under /root/reference and is restated
// from its published behaviour:
//   * dk.brics.automaton:automaton:1.11-8 (gorp-core/pom.xml:26-30): RegExp
//     grammar with flags NONE, toAutomaton, minimize (unique minimal trimmed
//     DFA), getStartPoints, State.step.
//   * java.util.regex (JDK 7/8 level, pom.xml:37-38): Pattern.compile with no
//     flags, Matcher.matches(), group(i) -- greedy/lazy backtracking with
//     leftmost-alternative priority.
// =============================================================================
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

typedef std::u16string ustr;
typedef std::pair<int, int> Ival;          // inclusive [lo, hi] of UTF-16 code units
typedef std::vector<Ival> IvalSet;         // sorted, disjoint, non-adjacent

static const int CMAX = 0xFFFF;

struct OracleError : std::runtime_error {
    explicit OracleError(const std::string& m) : std::runtime_error(m) {}
};

// ---------------------------------------------------------------------------
// UTF-8 <-> UTF-16 helpers (API strings are UTF-8; Java strings are UTF-16)
// ---------------------------------------------------------------------------
static ustr utf8_to_utf16(const char* s) {
    ustr out;
    const unsigned char* p = (const unsigned char*)s;
    while (*p) {
        uint32_t cp;
        if (*p < 0x80) cp = *p++;
        else if ((*p >> 5) == 6) { cp = (*p & 0x1F) << 6 | (p[1] & 0x3F); p += 2; }
        else if ((*p >> 4) == 14) { cp = (*p & 0x0F) << 12 | (p[1] & 0x3F) << 6 | (p[2] & 0x3F); p += 3; }
        else { cp = (*p & 0x07) << 18 | (p[1] & 0x3F) << 12 | (p[2] & 0x3F) << 6 | (p[3] & 0x3F); p += 4; }
        if (cp >= 0x10000) {
            cp -= 0x10000;
            out.push_back((char16_t)(0xD800 + (cp >> 10)));
            out.push_back((char16_t)(0xDC00 + (cp & 0x3FF)));
        } else out.push_back((char16_t)cp);
    }
    return out;
}

static std::string utf16_to_utf8(const ustr& s) {
    std::string out;
    for (size_t i = 0; i < s.size(); ++i) {
        uint32_t cp = s[i];
        if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < s.size() && s[i + 1] >= 0xDC00 && s[i + 1] < 0xE000) {
            cp = 0x10000 + ((cp - 0xD800) << 10) + (s[i + 1] - 0xDC00);
            ++i;
        }
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | cp >> 6)); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { out.push_back((char)(0xE0 | cp >> 12)); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else { out.push_back((char)(0xF0 | cp >> 18)); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    return out;
}

// ---------------------------------------------------------------------------
// Interval-set helpers
// ---------------------------------------------------------------------------
static IvalSet normalize(IvalSet v) {
    std::sort(v.begin(), v.end());
    IvalSet out;
    for (auto& iv : v) {
        if (iv.first > iv.second) continue;
        if (!out.empty() && iv.first <= out.back().second + 1)
            out.back().second = std::max(out.back().second, iv.second);
        else out.push_back(iv);
    }
    return out;
}

static IvalSet complement(const IvalSet& v) {
    IvalSet out;
    int next = 0;
    for (auto& iv : v) {
        if (iv.first > next) out.push_back(Ival(next, iv.first - 1));
        next = iv.second + 1;
    }
    if (next <= CMAX) out.push_back(Ival(next, CMAX));
    return out;
}

static bool set_has(const IvalSet& v, int c) {
    // binary search
    int lo = 0, hi = (int)v.size() - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (c < v[mid].first) hi = mid - 1;
        else if (c > v[mid].second) lo = mid + 1;
        else return true;
    }
    return false;
}

// ===========================================================================
// Part 0: RegexHelper restatement (core/util/RegexHelper.java)
// ===========================================================================
static const char16_t* CC_d = u"0-9";                 // RegexHelper.java:11
static const char16_t* CC_s = u" \b\f\n\r\t";        // RegexHelper.java:12
static const char16_t* CC_w = u"a-zA-Z_0-9";          // RegexHelper.java:13

// RegexHelper.java:20-70
static ustr quote_literal_as_regexp(const ustr& text) {
    ustr sb;
    const size_t end = text.size();
    for (size_t i = 0; i < end;) {
        char16_t c = text[i++];
        switch (c) {
        case u' ':
        case u'\t':
            while (i < end && text[i] <= u' ') ++i;
            sb += u"[ \t]+";
            break;
        case u'.':
            sb += u"\\.";
            break;
        case u'(': case u')': case u'[': case u']': case u'\\': case u'{': case u'}':
        case u'|': case u'*': case u'?': case u'+': case u'$': case u'^':
        case u'<': case u'>': case u'"': case u'&':
            sb.push_back(u'\\');
            sb.push_back(c);
            break;
        default:
            sb.push_back(c);
        }
    }
    return sb;
}

// Character.isAlphabetic(d) || Character.isDigit(d) -- RegexHelper.java:171.
// Exact for ASCII; for non-ASCII code units we approximate "alphabetic" with
// the Latin-1 letters (the escape would be nonsense in either engine anyway).
static bool java_is_alnum(char16_t d) {
    if (d < 0x80) return (d >= u'0' && d <= u'9') || (d >= u'a' && d <= u'z') || (d >= u'A' && d <= u'Z');
    if (d == 0xAA || d == 0xB5 || d == 0xBA) return true;
    if (d >= 0xC0 && d <= 0xFF && d != 0xD7 && d != 0xF7) return true;
    return d > 0xFF;  // conservative: treat other BMP chars as alphabetic
}

// RegexHelper.java:184-201
static void append_char_class(ustr& sb, char16_t charClass, bool hadBracket, int bracketNesting, const ustr& chars) {
    if (bracketNesting == 0) {
        sb.push_back(u'[');
        sb += chars;
        sb.push_back(u']');
        return;
    }
    if (!chars.empty() && chars[0] == u'^' && !hadBracket) {
        std::string m = "Can not use negated character class \\";
        m += (char)charClass;
        m += " within character class in position other than first (Automaton limitation)";
        throw OracleError(m);
    }
    sb += chars;
}

// RegexHelper.java:79-182
static ustr massage_regexp_for_automaton(const ustr& pattern) {
    ustr sb;
    if (pattern.find(u'\\') == ustr::npos) return pattern;
    const size_t end = pattern.size();
    int bracketLevels = 0;
    for (size_t i = 0; i < end;) {
        char16_t c = pattern[i++];
        if (c == u'[') { sb.push_back(c); ++bracketLevels; continue; }
        if (c == u']') { sb.push_back(c); --bracketLevels; continue; }
        if (c != u'\\' || i >= end) { sb.push_back(c); continue; }
        bool hadBracket = (bracketLevels > 0) && (pattern[i - 2] == u'[');
        char16_t d = pattern[i++];
        switch (d) {
        case u'\\': break;
        case u'b': d = u'\b'; break;
        case u'f': d = u'\f'; break;
        case u'n': d = u'\n'; break;
        case u'r': d = u'\r'; break;
        case u't': d = u'\t'; break;
        case u'd': append_char_class(sb, d, hadBracket, bracketLevels, CC_d); continue;
        case u'D': append_char_class(sb, d, hadBracket, bracketLevels, ustr(u"^") + CC_d); continue;
        case u's': append_char_class(sb, d, hadBracket, bracketLevels, CC_s); continue;
        case u'S': append_char_class(sb, d, hadBracket, bracketLevels, ustr(u"^") + CC_s); continue;
        case u'w': append_char_class(sb, d, hadBracket, bracketLevels, CC_w); continue;
        case u'W': append_char_class(sb, d, hadBracket, bracketLevels, ustr(u"^") + CC_w); continue;
        default:
            if (java_is_alnum(d)) {
                std::string m = "Unrecognized backslash escape '\\";
                m += utf16_to_utf8(ustr(1, d));
                m += "; can only escape backslash (\\\\), use known control-codes (\\n, \\r, \\t),"
                     " escape non-alphanumeric (\\$, \\(, ...) or refer to a 'well-known' character class"
                     " (\\s, \\S, \\d, \\D, \\w, \\W)";
                throw OracleError(m);
            }
        }
        sb.push_back(c);
        sb.push_back(d);
    }
    return sb;
}

// RegexHelper.java:210-237
static ustr massage_regexp_for_jdk(const ustr& pattern) {
    ustr sb;
    const size_t end = pattern.size();
    for (size_t i = 0; i < end;) {
        char16_t c = pattern[i++];
        if (c == u'\\') {
            sb.push_back(c);
            if (i < end) sb.push_back(pattern[i++]);
        } else if (c == u'(') {
            sb += u"(?:";
        } else sb.push_back(c);
    }
    return sb;
}

// ===========================================================================
// Part 1: dk.brics.automaton RegExp (flags = NONE) -> minimal trimmed DFA
//   [third-party restatement; call site core/autom/PolyMatcher.java:76-77]
// ===========================================================================
enum BKind { B_UNION, B_CONCAT, B_OPT, B_REPEAT, B_SET, B_STRING };

struct BNode {
    BKind kind;
    std::unique_ptr<BNode> a, b;
    int min = 0, max = -1;   // B_REPEAT: max == -1 => unbounded
    IvalSet set;             // B_SET
    ustr str;                // B_STRING
};
typedef std::unique_ptr<BNode> BP;

// Recursive-descent parser following the published grammar of
// dk.brics.automaton.RegExp (1.11-8) with every optional syntax flag off:
//   union  := inter ('|' union)?        [inter == concat: INTERSECTION off]
//   concat := repeat concat?            (stops at ')' or '|')
//   repeat := compl ('?'|'*'|'+'|'{n}'|'{n,}'|'{n,m}')*   [compl == charclass: COMPLEMENT off]
//   cclass := '[' '^'? classes ']' | simple
//   simple := '.' | '"' ... '"' | '(' ')' | '(' union ')' | charexp
//   charexp:= '\'? anychar
struct BricsParser {
    const ustr& b;
    size_t pos = 0;
    explicit BricsParser(const ustr& s) : b(s) {}

    bool more() const { return pos < b.size(); }
    bool peek(const char* s) const {
        if (!more()) return false;
        char16_t c = b[pos];
        for (const char* p = s; *p; ++p) if ((char16_t)(unsigned char)*p == c) return true;
        return false;
    }
    bool match(char16_t c) {
        if (pos >= b.size()) return false;
        if (b[pos] == c) { pos++; return true; }
        return false;
    }
    char16_t next() {
        if (!more()) throw OracleError("unexpected end-of-string");
        return b[pos++];
    }

    static BP mk(BKind k) { BP n(new BNode()); n->kind = k; return n; }
    static BP mkset(IvalSet s) { BP n = mk(B_SET); n->set = normalize(std::move(s)); return n; }
    static BP mk2(BKind k, BP a, BP b) { BP n = mk(k); n->a = std::move(a); n->b = std::move(b); return n; }

    BP parse() {
        if (b.empty()) { BP n = mk(B_STRING); return n; }
        BP e = parseUnionExp();
        if (pos < b.size())
            throw OracleError("end-of-string expected at position " + std::to_string(pos));
        return e;
    }
    BP parseUnionExp() {
        BP e = parseConcatExp();   // parseInterExp with INTERSECTION disabled
        if (match(u'|')) e = mk2(B_UNION, std::move(e), parseUnionExp());
        return e;
    }
    BP parseConcatExp() {
        BP e = parseRepeatExp();
        if (more() && !peek(")|")) e = mk2(B_CONCAT, std::move(e), parseConcatExp());
        return e;
    }
    BP parseRepeatExp() {
        BP e = parseCharClassExp();  // parseComplExp with COMPLEMENT disabled
        while (peek("?*+{")) {
            if (match(u'?')) { BP n = mk(B_OPT); n->a = std::move(e); e = std::move(n); }
            else if (match(u'*')) { BP n = mk(B_REPEAT); n->a = std::move(e); n->min = 0; n->max = -1; e = std::move(n); }
            else if (match(u'+')) { BP n = mk(B_REPEAT); n->a = std::move(e); n->min = 1; n->max = -1; e = std::move(n); }
            else if (match(u'{')) {
                size_t start = pos;
                while (peek("0123456789")) next();
                if (start == pos) throw OracleError("integer expected at position " + std::to_string(pos));
                int n = parse_int(start, pos);
                int m = -1;
                if (match(u',')) {
                    start = pos;
                    while (peek("0123456789")) next();
                    if (start != pos) m = parse_int(start, pos);
                } else m = n;
                if (!match(u'}')) throw OracleError("expected '}' at position " + std::to_string(pos));
                BP r = mk(B_REPEAT);
                r->a = std::move(e); r->min = n; r->max = m;  // m == -1: makeRepeat(e, n)
                e = std::move(r);
            }
        }
        return e;
    }
    int parse_int(size_t s, size_t e) {
        long v = 0;
        for (size_t i = s; i < e; ++i) { v = v * 10 + (b[i] - u'0'); if (v > 100000) throw OracleError("repeat count too large"); }
        return (int)v;
    }
    BP parseCharClassExp() {
        if (match(u'[')) {
            bool negate = false;
            if (match(u'^')) negate = true;
            IvalSet s = parseCharClasses();
            if (negate) s = complement(normalize(s));
            if (!match(u']')) throw OracleError("expected ']' at position " + std::to_string(pos));
            return mkset(s);
        }
        return parseSimpleExp();
    }
    IvalSet parseCharClasses() {
        IvalSet s;
        parseCharClass(s);
        while (more() && !peek("]")) parseCharClass(s);
        return s;
    }
    void parseCharClass(IvalSet& s) {
        char16_t c = parseCharExp();
        if (match(u'-')) {
            if (peek("]")) { s.push_back(Ival(c, c)); s.push_back(Ival(u'-', u'-')); }
            else {
                char16_t d = parseCharExp();
                // BasicAutomata.makeCharRange: empty language if min > max
                if (c <= d) s.push_back(Ival(c, d));
            }
        } else s.push_back(Ival(c, c));
    }
    BP parseSimpleExp() {
        if (match(u'.')) return mkset(IvalSet{Ival(0, CMAX)});
        if (match(u'"')) {
            size_t start = pos;
            while (more() && !peek("\"")) next();
            if (!match(u'"')) throw OracleError("expected '\"' at position " + std::to_string(pos));
            BP n = mk(B_STRING);
            n->str = b.substr(start, pos - 1 - start);
            return n;
        }
        if (match(u'(')) {
            if (match(u')')) return mk(B_STRING);
            BP e = parseUnionExp();
            if (!match(u')')) throw OracleError("expected ')' at position " + std::to_string(pos));
            return e;
        }
        char16_t c = parseCharExp();
        return mkset(IvalSet{Ival(c, c)});
    }
    char16_t parseCharExp() {
        match(u'\\');
        return next();
    }
};

// --- Thompson NFA over interval-labelled transitions ------------------------
struct Nfa {
    struct Tr { int lo, hi, to; };
    struct St { std::vector<Tr> tr; std::vector<int> eps; };
    std::vector<St> st;
    int add() { st.emplace_back(); return (int)st.size() - 1; }
};
struct Frag { int s, e; };

static const size_t NFA_LIMIT = 2000000;

static Frag nfa_build(Nfa& n, const BNode* x) {
    if (n.st.size() > NFA_LIMIT) throw OracleError("automaton too large");
    switch (x->kind) {
    case B_SET: {
        int s = n.add(), e = n.add();
        for (auto& iv : x->set) n.st[s].tr.push_back({iv.first, iv.second, e});
        return {s, e};
    }
    case B_STRING: {
        int s = n.add(), cur = s;
        for (char16_t c : x->str) { int t = n.add(); n.st[cur].tr.push_back({c, c, t}); cur = t; }
        return {s, cur};
    }
    case B_CONCAT: {
        Frag a = nfa_build(n, x->a.get());
        Frag b = nfa_build(n, x->b.get());
        n.st[a.e].eps.push_back(b.s);
        return {a.s, b.e};
    }
    case B_UNION: {
        Frag a = nfa_build(n, x->a.get());
        Frag b = nfa_build(n, x->b.get());
        int s = n.add(), e = n.add();
        n.st[s].eps.push_back(a.s); n.st[s].eps.push_back(b.s);
        n.st[a.e].eps.push_back(e); n.st[b.e].eps.push_back(e);
        return {s, e};
    }
    case B_OPT: {
        Frag a = nfa_build(n, x->a.get());
        int s = n.add(), e = n.add();
        n.st[s].eps.push_back(a.s); n.st[s].eps.push_back(e);
        n.st[a.e].eps.push_back(e);
        return {s, e};
    }
    case B_REPEAT: {
        // Automaton.repeat(min) = a^min a* ; repeat(min,max) = a^min (a?)^(max-min), empty if min > max
        int s = n.add(), cur = s;
        if (x->max >= 0 && x->min > x->max) { int e = n.add(); return {s, e}; }  // empty language
        for (int i = 0; i < x->min; ++i) {
            Frag a = nfa_build(n, x->a.get());
            n.st[cur].eps.push_back(a.s);
            cur = a.e;
        }
        if (x->max < 0) {
            Frag a = nfa_build(n, x->a.get());
            int e = n.add();
            n.st[cur].eps.push_back(a.s); n.st[cur].eps.push_back(e);
            n.st[a.e].eps.push_back(a.s); n.st[a.e].eps.push_back(e);
            return {s, e};
        }
        int e = n.add();
        for (int i = x->min; i < x->max; ++i) {
            Frag a = nfa_build(n, x->a.get());
            n.st[cur].eps.push_back(a.s);
            n.st[cur].eps.push_back(e);
            cur = a.e;
        }
        n.st[cur].eps.push_back(e);
        return {s, e};
    }
    }
    throw OracleError("internal: bad node");
}

// --- Deterministic automaton with interval transitions (brics State/Transition)
struct Dfa {
    struct Tr { int lo, hi, to; };
    struct St { bool accept = false; std::vector<Tr> tr; };  // tr sorted by lo, reduced
    std::vector<St> st;
    int initial = 0;
    // State.step(char): linear scan of transitions, null (-1) if none
    int step(int s, int c) const {
        for (auto& t : st[s].tr) if (t.lo <= c && c <= t.hi) return t.to;
        return -1;
    }
};

static void eps_closure(const Nfa& n, std::vector<int>& set, std::vector<char>& mark) {
    std::vector<int> stack(set);
    for (int s : set) mark[s] = 1;
    while (!stack.empty()) {
        int s = stack.back(); stack.pop_back();
        for (int t : n.st[s].eps) if (!mark[t]) { mark[t] = 1; set.push_back(t); stack.push_back(t); }
    }
    for (int s : set) mark[s] = 0;
    std::sort(set.begin(), set.end());
}

// RegExp.toAutomaton() + Automaton.minimize(): the unique minimal DFA of the
// language with dead states removed (brics minimize ends with
// removeDeadTransitions + reduce).
static Dfa regex_to_min_dfa(const ustr& pattern) {
    BricsParser parser(pattern);
    BP root = parser.parse();
    Nfa nfa;
    Frag f = nfa_build(nfa, root.get());
    const int nfaFinal = f.e;

    // atomic intervals
    std::vector<int> pts{0};
    for (auto& s : nfa.st) for (auto& t : s.tr) { pts.push_back(t.lo); if (t.hi < CMAX) pts.push_back(t.hi + 1); }
    std::sort(pts.begin(), pts.end());
    pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
    const int K = (int)pts.size();

    // subset construction
    std::map<std::vector<int>, int> index;
    std::vector<std::vector<int>> subsets;
    std::vector<std::vector<int>> trans;  // [state][k] -> state or -1
    std::vector<char> acc;
    std::vector<char> mark(nfa.st.size(), 0);
    {
        std::vector<int> init{f.s};
        eps_closure(nfa, init, mark);
        index[init] = 0; subsets.push_back(init);
    }
    for (size_t i = 0; i < subsets.size(); ++i) {
        if (subsets.size() > 200000) throw OracleError("automaton too large");
        std::vector<int> cur = subsets[i];
        acc.push_back(std::binary_search(cur.begin(), cur.end(), nfaFinal));
        std::vector<int> row(K, -1);
        for (int k = 0; k < K; ++k) {
            int c = pts[k];
            std::vector<int> tgt;
            for (int s : cur) for (auto& t : nfa.st[s].tr) if (t.lo <= c && c <= t.hi) tgt.push_back(t.to);
            if (tgt.empty()) continue;
            std::sort(tgt.begin(), tgt.end());
            tgt.erase(std::unique(tgt.begin(), tgt.end()), tgt.end());
            eps_closure(nfa, tgt, mark);
            auto it = index.find(tgt);
            int id;
            if (it == index.end()) { id = (int)subsets.size(); index[tgt] = id; subsets.push_back(tgt); }
            else id = it->second;
            row[k] = id;
        }
        trans.push_back(row);
    }
    const int N = (int)subsets.size();

    // liveness (can reach accept)
    std::vector<char> live(N, 0);
    {
        std::vector<std::vector<int>> rev(N);
        for (int s = 0; s < N; ++s) for (int k = 0; k < K; ++k) if (trans[s][k] >= 0) rev[trans[s][k]].push_back(s);
        std::vector<int> stack;
        for (int s = 0; s < N; ++s) if (acc[s]) { live[s] = 1; stack.push_back(s); }
        while (!stack.empty()) {
            int s = stack.back(); stack.pop_back();
            for (int p : rev[s]) if (!live[p]) { live[p] = 1; stack.push_back(p); }
        }
    }
    // Moore partition refinement; class 0 = dead (incl. implicit sink)
    std::vector<int> cls(N);
    bool anyAcc = false, anyNon = false;
    for (int s = 0; s < N; ++s) {
        cls[s] = !live[s] ? 0 : (acc[s] ? 2 : 1);
        if (cls[s] == 2) anyAcc = true;
        if (cls[s] == 1) anyNon = true;
    }
    // number of classes actually in use (dead class always counted); refinement
    // is monotone, so an unchanged count means an unchanged partition
    int ncls = 1 + (anyAcc ? 1 : 0) + (anyNon ? 1 : 0);
    for (;;) {
        std::map<std::vector<int>, int> sigidx;
        std::vector<int> ncl(N);
        int cnt = 1;  // keep 0 for dead
        for (int s = 0; s < N; ++s) {
            if (cls[s] == 0) { ncl[s] = 0; continue; }
            std::vector<int> sig(K + 1);
            sig[0] = cls[s];
            for (int k = 0; k < K; ++k) sig[k + 1] = trans[s][k] < 0 ? 0 : cls[trans[s][k]];
            auto it = sigidx.find(sig);
            if (it == sigidx.end()) { sigidx[sig] = cnt; ncl[s] = cnt++; }
            else ncl[s] = it->second;
        }
        bool same = (cnt == ncls);
        cls.swap(ncl);
        ncls = cnt;
        if (same) break;
    }
    // build minimal DFA over live classes; keep the initial state even when dead
    Dfa d;
    std::vector<int> clsToState(ncls, -1);
    std::vector<int> rep;
    auto get_state = [&](int c, int r) {
        if (clsToState[c] < 0) { clsToState[c] = (int)rep.size(); rep.push_back(r); }
        return clsToState[c];
    };
    if (cls[0] == 0) {  // empty language: single non-accepting state, no transitions
        d.st.emplace_back();
        d.initial = 0;
        return d;
    }
    get_state(cls[0], 0);
    for (size_t i = 0; i < rep.size(); ++i) {
        int r = rep[i];
        Dfa::St st;
        st.accept = acc[r];
        for (int k = 0; k < K; ++k) {
            int t = trans[r][k];
            if (t < 0 || cls[t] == 0) continue;
            int to = get_state(cls[t], t);
            int lo = pts[k], hi = (k + 1 < K) ? pts[k + 1] - 1 : CMAX;
            // Automaton.reduce(): merge adjacent ranges with the same destination
            if (!st.tr.empty() && st.tr.back().to == to && st.tr.back().hi + 1 == lo) st.tr.back().hi = hi;
            else st.tr.push_back({lo, hi, to});
        }
        if (d.st.size() <= i) d.st.resize(i + 1);
        d.st[i] = st;
    }
    d.st.resize(rep.size());
    d.initial = 0;
    return d;
}

// Automaton.getStartPoints(): {0} U {t.min} U {t.max+1 | t.max < 0xFFFF}, sorted
static std::vector<int> start_points(const Dfa& d) {
    std::set<int> p;
    p.insert(0);
    for (auto& s : d.st) for (auto& t : s.tr) { p.insert(t.lo); if (t.hi < CMAX) p.insert(t.hi + 1); }
    return std::vector<int>(p.begin(), p.end());
}

// ===========================================================================
// Part 2: Automata.construct / step / accept  (core/autom/Automata.java)
// ===========================================================================
struct Automata {
    std::vector<std::vector<int>> accept;  // _accept
    int stride = 0;                        // _stride
    std::vector<int32_t> transitions;      // _transitions
    std::vector<int32_t> alphabet;         // _alphabet (65536 entries)
    std::vector<int> points;
    int inputRegexpCount = 0;
    std::vector<int> componentStates;      // minimal-DFA size per regex (diagnostics)

    int nbStates() const { return stride ? (int)(transitions.size() / stride) : 0; }
    // Automata.java:133-135
    inline int step(int state, int c) const { return transitions[state * stride + alphabet[c]]; }
};

// Automata.java:45-55
static std::vector<int32_t> alphabet_of(const std::vector<int>& points) {
    const int size = 65536;
    std::vector<int32_t> alphabet(size);
    for (int i = 0, j = 0; j < size; ++j) {
        if (i + 1 < (int)points.size() && j == points[i + 1]) i++;
        alphabet[j] = i;
    }
    return alphabet;
}

// PolyState: tuple of per-regex State (null = -1).  Stored sparsely as sorted
// (component, state) pairs of the non-null entries -- same equality relation
// as Arrays.equals on the dense tuple (PolyState.java:79-90).
typedef std::vector<std::pair<int, int>> PolyState;
struct PolyHash {
    size_t operator()(const PolyState& p) const {
        size_t h = 1469598103934665603ull;
        for (auto& e : p) { h ^= (size_t)e.first * 0x9E3779B97F4A7C15ull + e.second; h *= 1099511628211ull; }
        return h;
    }
};

// Automata.java:57-124
static Automata automata_construct(const std::vector<Dfa>& automata) {
    Automata A;
    A.inputRegexpCount = (int)automata.size();
    // pointsUnion (Automata.java:150-166)
    std::set<int> pset;
    for (auto& a : automata) for (int p : start_points(a)) pset.insert(p);
    A.points.assign(pset.begin(), pset.end());
    const int plen = (int)A.points.size();
    for (auto& a : automata) A.componentStates.push_back((int)a.st.size());

    std::vector<PolyState> queue;  // FIFO by index
    std::unordered_map<PolyState, int, PolyHash> multiStateIndex;
    PolyState init;
    for (int c = 0; c < (int)automata.size(); ++c) init.push_back({c, automata[c].initial});
    multiStateIndex[init] = 0;
    queue.push_back(init);
    std::vector<int32_t> trans;
    for (size_t head = 0; head < queue.size(); ++head) {
        if (queue.size() > 4000000) throw OracleError("product automaton too large");
        PolyState visiting = queue[head];
        for (int c = 0; c < plen; ++c) {
            const int point = A.points[c];
            PolyState dest;  // PolyState.step (PolyState.java:55-62)
            for (auto& e : visiting) {
                int t = automata[e.first].step(e.second, point);
                if (t >= 0) dest.push_back({e.first, t});
            }
            if (dest.empty()) { trans.push_back(-1); continue; }  // isNull (PolyState.java:46-53)
            auto it = multiStateIndex.find(dest);
            int id;
            if (it == multiStateIndex.end()) {
                id = (int)multiStateIndex.size();
                multiStateIndex[dest] = id;
                queue.push_back(dest);
            } else id = it->second;
            trans.push_back(id);
        }
    }
    A.stride = plen;
    A.transitions = trans;
    A.accept.resize(queue.size());
    for (size_t id = 0; id < queue.size(); ++id) {
        // PolyState.toAcceptValues (PolyState.java:64-77): ascending component indexes
        for (auto& e : queue[id]) if (automata[e.first].st[e.second].accept) A.accept[id].push_back(e.first);
    }
    A.alphabet = alphabet_of(A.points);
    return A;
}

// PolyMatcher.match (PolyMatcher.java:123-133).  Returns state reached or -1.
template <typename CH>
static inline int poly_walk(const Automata& A, const CH* s, int l) {
    int p = 0;
    for (int i = 0; i < l; ++i) {
        p = A.step(p, (int)s[i]);
        if (p == -1) return -1;
    }
    return p;
}

// ===========================================================================
// Part 3: java.util.regex restatement (subset reachable from Gorp, Appendix A.2)
//   [third-party restatement; call sites core/jdkre/JDKRegexpExtractionCooker.java:23,
//    core/jdkre/JDKRegexpCookedExtraction.java:36-59]
// ===========================================================================
enum JOp { J_CHAR, J_SPLIT, J_JMP, J_SAVE, J_MATCH, J_BOL, J_EOL };
struct JInst { JOp op; int x = 0, y = 0; };  // CHAR: x = set index; SPLIT: prefer x then y; SAVE: x = slot

struct JSet {
    IvalSet iv;
    uint64_t lat[4] = {0, 0, 0, 0};  // bitmap for code units < 256
    void finish() {
        iv = normalize(iv);
        for (int c = 0; c < 256; ++c) if (set_has(iv, c)) lat[c >> 6] |= 1ull << (c & 63);
    }
    inline bool has(int c) const {
        if (c < 256) return (lat[c >> 6] >> (c & 63)) & 1;
        return set_has(iv, c);
    }
};

struct JProg {
    std::vector<JInst> code;
    std::vector<JSet> sets;
    int ngroups = 0;
};

enum JKind { N_ALT, N_CAT, N_REP, N_GROUP, N_SET, N_EMPTY, N_BOL, N_EOL };
struct JNode {
    JKind kind;
    std::vector<std::unique_ptr<JNode>> kids;
    int min = 0, max = -1; bool lazy = false;  // N_REP
    int cap = -1;                                // N_GROUP: capture index (1-based) or -1
    IvalSet set;                                 // N_SET
};
typedef std::unique_ptr<JNode> JP;

static const IvalSet JS_d{Ival('0', '9')};
static const IvalSet JS_w{Ival('0', '9'), Ival('A', 'Z'), Ival('_', '_'), Ival('a', 'z')};
static const IvalSet JS_s{Ival(9, 13), Ival(' ', ' ')};  // [ \t\n\x0B\f\r]

struct JdkParser {
    const ustr& p;
    size_t pos = 0;
    int ngroups = 0;
    explicit JdkParser(const ustr& s) : p(s) {}
    bool more() const { return pos < p.size(); }
    int peekc() const { return more() ? p[pos] : -1; }

    [[noreturn]] void unsupported(const std::string& what) {
        throw OracleError("unsupported java.util.regex construct: " + what + " near index " + std::to_string(pos));
    }
    [[noreturn]] void syntax(const std::string& what) {
        throw OracleError("PatternSyntaxException: " + what + " near index " + std::to_string(pos));
    }
    static JP mk(JKind k) { JP n(new JNode()); n->kind = k; return n; }

    JP parse() {
        JP e = expr();
        if (more()) {
            if (p[pos] == u')') syntax("Unmatched closing ')'");
            syntax("Unexpected internal error");
        }
        return e;
    }
    JP expr() {
        JP alt = mk(N_ALT);
        alt->kids.push_back(sequence());
        while (peekc() == u'|') { pos++; alt->kids.push_back(sequence()); }
        if (alt->kids.size() == 1) return std::move(alt->kids[0]);
        return alt;
    }
    JP sequence() {
        JP cat = mk(N_CAT);
        while (more() && p[pos] != u'|' && p[pos] != u')') {
            JP a = atom();
            a = closure(std::move(a));
            cat->kids.push_back(std::move(a));
        }
        return cat;
    }
    // java.util.regex takes ONE quantifier per atom: Pattern.sequence() throws "Dangling meta character" on a second
    // '*', '+' or '?' (a second '{' would quantify an empty literal: refused here, not imitated).  A loop around a
    // capturing body that can match the empty string is refused as well: java.util.regex's Loop / GroupTail let an
    // empty last iteration move the group ("(a*)*" on "aaa": group 1 = (3,3)), which this restatement does not
    // model; Gorp's extractor groups are never quantified (core/Gorp.java:94-129).
    static bool nullable(const JNode& n) {
        switch (n.kind) {
        case N_EMPTY: case N_BOL: case N_EOL: return true;
        case N_SET: return false;
        case N_CAT: for (auto& k : n.kids) if (!nullable(*k)) return false; return true;
        case N_ALT: for (auto& k : n.kids) if (nullable(*k)) return true; return false;
        case N_REP: return n.min == 0 || nullable(*n.kids[0]);
        case N_GROUP: return nullable(*n.kids[0]);
        }
        return false;
    }
    static bool captures(const JNode& n) {
        if (n.kind == N_GROUP && n.cap > 0) return true;
        for (auto& k : n.kids) if (captures(*k)) return true;
        return false;
    }
    JP closure(JP a) {
        for (int count = 0;; ++count) {
            int c = peekc();
            int mn, mx;
            if (count == 1) {
                if (c == u'?' || c == u'*' || c == u'+') syntax(std::string("Dangling meta character '") + (char)c + "'");
                if (c == u'{') unsupported("a quantifier applied to a quantifier");
                return a;
            }
            if (c == u'?') { pos++; mn = 0; mx = 1; }
            else if (c == u'*') { pos++; mn = 0; mx = -1; }
            else if (c == u'+') { pos++; mn = 1; mx = -1; }
            else if (c == u'{') {
                size_t save = pos;
                pos++;
                if (!(more() && p[pos] >= u'0' && p[pos] <= u'9')) { pos = save; syntax("Illegal repetition"); }
                long n = 0;
                while (more() && p[pos] >= u'0' && p[pos] <= u'9') { n = n * 10 + (p[pos++] - u'0'); if (n > 100000) syntax("Illegal repetition range"); }
                mn = (int)n; mx = mn;
                if (peekc() == u',') {
                    pos++;
                    mx = -1;
                    if (peekc() != u'}') {
                        if (!(more() && p[pos] >= u'0' && p[pos] <= u'9')) syntax("Illegal repetition");
                        long m = 0;
                        while (more() && p[pos] >= u'0' && p[pos] <= u'9') { m = m * 10 + (p[pos++] - u'0'); if (m > 100000) syntax("Illegal repetition range"); }
                        mx = (int)m;
                        if (mx < mn) syntax("Illegal repetition range");
                    }
                }
                if (peekc() != u'}') syntax("Unclosed counted closure");
                pos++;
            } else return a;
            bool lazy = false;
            if (peekc() == u'?') { pos++; lazy = true; }
            else if (peekc() == u'+') unsupported("possessive quantifier");
            if (mx != 1 && mx != 0 && captures(*a) && nullable(*a)) unsupported("a repeated capturing group that can match the empty string");
            JP r = mk(N_REP);
            r->min = mn; r->max = mx; r->lazy = lazy;
            r->kids.push_back(std::move(a));
            a = std::move(r);
        }
    }
    static JP mkset(IvalSet s) { JP n = mk(N_SET); n->set = normalize(std::move(s)); return n; }

    int hexval(int c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    // Parses what follows a backslash. Returns true and fills 'set' for class
    // escapes, else returns the single code unit in 'ch'.
    bool escape(bool inClass, IvalSet& set, int& ch) {
        if (!more()) syntax("Unexpected internal error (trailing backslash)");
        char16_t c = p[pos++];
        switch (c) {
        case u'd': set = JS_d; return true;
        case u'D': set = complement(JS_d); return true;
        case u's': set = JS_s; return true;
        case u'S': set = complement(JS_s); return true;
        case u'w': set = JS_w; return true;
        case u'W': set = complement(JS_w); return true;
        case u't': ch = 9; return false;
        case u'n': ch = 10; return false;
        case u'r': ch = 13; return false;
        case u'f': ch = 12; return false;
        case u'a': ch = 7; return false;
        case u'e': ch = 27; return false;
        case u'0': {
            int n = 0, cnt = 0;
            while (cnt < 3 && more() && p[pos] >= u'0' && p[pos] <= u'7') {
                int v = n * 8 + (p[pos] - u'0');
                if (v > 0377) break;
                n = v; pos++; cnt++;
            }
            if (cnt == 0) syntax("Illegal octal escape sequence");
            ch = n; return false;
        }
        case u'x': {
            if (peekc() == u'{') unsupported("\\x{...}");
            if (pos + 1 < p.size() && hexval(p[pos]) >= 0 && hexval(p[pos + 1]) >= 0) {
                ch = hexval(p[pos]) * 16 + hexval(p[pos + 1]); pos += 2; return false;
            }
            syntax("Illegal hexadecimal escape sequence");
        }
        case u'u': {
            if (pos + 3 < p.size()) {
                int v = 0; bool ok = true;
                for (int i = 0; i < 4; ++i) { int h = hexval(p[pos + i]); if (h < 0) ok = false; v = v * 16 + h; }
                if (ok) { pos += 4; ch = v; return false; }
            }
            syntax("Illegal Unicode escape sequence");
        }
        case u'c': {
            if (!more()) syntax("Illegal control escape sequence");
            ch = p[pos++] ^ 64; return false;
        }
        default:
            if ((c >= u'a' && c <= u'z') || (c >= u'A' && c <= u'Z') || (c >= u'1' && c <= u'9'))
                unsupported(std::string("escape \\") + (char)c);
            (void)inClass;
            ch = c; return false;  // escaped non-alphanumeric: literal
        }
    }
    JP atom() {
        char16_t c = p[pos];
        switch (c) {
        case u'(': {
            pos++;
            JP g = mk(N_GROUP);
            if (peekc() == u'?') {
                if (pos + 1 < p.size() && p[pos + 1] == u':') { pos += 2; g->cap = -1; }
                else unsupported("special group (?...");
            } else g->cap = ++ngroups;
            g->kids.push_back(expr());
            if (peekc() != u')') syntax("Unclosed group");
            pos++;
            return g;
        }
        case u'[': pos++; return clazz();
        case u'.': pos++; return mkset(complement(normalize(IvalSet{Ival(10, 10), Ival(13, 13), Ival(0x85, 0x85), Ival(0x2028, 0x2029)})));
        case u'^': pos++; return mk(N_BOL);
        case u'$': pos++; return mk(N_EOL);
        case u'*': case u'+': case u'?': syntax(std::string("Dangling meta character '") + (char)c + "'");
        case u'{': syntax("Illegal repetition");
        case u'\\': {
            pos++;
            IvalSet s; int ch = 0;
            if (escape(false, s, ch)) return mkset(s);
            return mkset(IvalSet{Ival(ch, ch)});
        }
        default:
            pos++;
            return mkset(IvalSet{Ival(c, c)});
        }
    }
    // after '[' consumed
    JP clazz() {
        bool negate = false;
        if (peekc() == u'^') { negate = true; pos++; }
        IvalSet acc;
        bool first = true;
        for (;;) {
            if (!more()) syntax("Unclosed character class");
            char16_t c = p[pos];
            if (c == u']' && !first) { pos++; break; }
            first = false;
            if (c == u'[') unsupported("nested character class");
            if (c == u'&' && pos + 1 < p.size() && p[pos + 1] == u'&') unsupported("character class intersection");
            int lo;
            if (c == u'\\') {
                pos++;
                IvalSet s; int ch = 0;
                if (escape(true, s, ch)) { acc.insert(acc.end(), s.begin(), s.end()); continue; }
                lo = ch;
            } else { pos++; lo = c; }
            // range?
            if (peekc() == u'-' && pos + 1 < p.size() && p[pos + 1] != u']') {
                if (p[pos + 1] == u'[') { acc.push_back(Ival(lo, lo)); continue; }
                pos++;  // consume '-'
                int hi;
                char16_t d = p[pos];
                if (d == u'\\') {
                    pos++;
                    IvalSet s; int ch = 0;
                    if (escape(true, s, ch)) syntax("Illegal character range");
                    hi = ch;
                } else { pos++; hi = d; }
                if (hi < lo) syntax("Illegal character range");
                acc.push_back(Ival(lo, hi));
            } else acc.push_back(Ival(lo, lo));
        }
        IvalSet s = normalize(acc);
        if (negate) s = complement(s);
        return mkset(s);
    }
};

static const size_t JPROG_LIMIT = 1000000;

struct JCompiler {
    JProg& P;
    explicit JCompiler(JProg& p) : P(p) {}
    int emit(JOp op, int x = 0, int y = 0) {
        if (P.code.size() > JPROG_LIMIT) throw OracleError("regex program too large");
        P.code.push_back({op, x, y});
        return (int)P.code.size() - 1;
    }
    void gen(const JNode* n) {
        switch (n->kind) {
        case N_EMPTY: break;
        case N_SET: { JSet s; s.iv = n->set; s.finish(); P.sets.push_back(s); emit(J_CHAR, (int)P.sets.size() - 1); break; }
        case N_BOL: emit(J_BOL); break;
        case N_EOL: emit(J_EOL); break;
        case N_CAT: for (auto& k : n->kids) gen(k.get()); break;
        case N_ALT: {
            std::vector<int> jmps;
            for (size_t i = 0; i < n->kids.size(); ++i) {
                if (i + 1 < n->kids.size()) {
                    int sp = emit(J_SPLIT);
                    P.code[sp].x = sp + 1;
                    gen(n->kids[i].get());
                    jmps.push_back(emit(J_JMP));
                    P.code[sp].y = (int)P.code.size();
                } else gen(n->kids[i].get());
            }
            for (int j : jmps) P.code[j].x = (int)P.code.size();
            break;
        }
        case N_GROUP:
            if (n->cap > 0) emit(J_SAVE, 2 * n->cap);
            gen(n->kids[0].get());
            if (n->cap > 0) emit(J_SAVE, 2 * n->cap + 1);
            break;
        case N_REP: {
            const JNode* body = n->kids[0].get();
            for (int i = 0; i < n->min; ++i) gen(body);
            if (n->max < 0) {
                // L1: split L2, L3 ; L2: body ; jmp L1 ; L3:
                int sp = emit(J_SPLIT);
                gen(body);
                emit(J_JMP, sp);
                int after = (int)P.code.size();
                if (n->lazy) { P.code[sp].x = after; P.code[sp].y = sp + 1; }
                else { P.code[sp].x = sp + 1; P.code[sp].y = after; }
            } else {
                std::vector<int> sps;
                for (int i = n->min; i < n->max; ++i) {
                    int sp = emit(J_SPLIT);
                    sps.push_back(sp);
                    gen(body);
                }
                int after = (int)P.code.size();
                for (int sp : sps) {
                    if (n->lazy) { P.code[sp].x = after; P.code[sp].y = sp + 1; }
                    else { P.code[sp].x = sp + 1; P.code[sp].y = after; }
                }
            }
            break;
        }
        }
    }
};

static JProg jdk_compile(const ustr& pattern) {
    JdkParser parser(pattern);
    JP root = parser.parse();
    JProg P;
    P.ngroups = parser.ngroups;
    JCompiler c(P);
    c.gen(root.get());
    c.emit(J_MATCH);
    return P;
}

// Matcher.matches(): whole-region anchored backtracking search.  Exploration
// order = java.util.regex's (left alternative first; greedy loops try one more
// iteration before leaving).  A (pc,pos) visited set prunes re-exploration of
// states that already failed (success from (pc,pos) does not depend on capture
// contents because back-references are outside the subset), so the first
// accepting path found is the same one plain backtracking finds.
struct JMatcher {
    const JProg& P;
    std::vector<uint64_t> visited;
    struct Frame { int pc, pos, slot, old; };
    std::vector<Frame> stack;
    std::vector<int> caps;
    explicit JMatcher(const JProg& p) : P(p) {}

    static bool is_line_term(int c) { return c == 10 || c == 13 || c == 0x85 || c == 0x2028 || c == 0x2029; }

    template <typename CH>
    bool matches(const CH* s, int len, int* out /* 2*ngroups: begin,end or -1,-1 */) {
        const int nprog = (int)P.code.size();
        size_t bits = (size_t)nprog * (len + 1);
        visited.assign((bits + 63) / 64, 0);
        caps.assign(2 * (P.ngroups + 1), -1);
        stack.clear();
        stack.push_back({0, 0, -1, 0});
        while (!stack.empty()) {
            Frame f = stack.back(); stack.pop_back();
            if (f.slot >= 0) { caps[f.slot] = f.old; continue; }
            int pc = f.pc, pos = f.pos;
            for (;;) {
                size_t bit = (size_t)pc * (len + 1) + pos;
                if (visited[bit >> 6] >> (bit & 63) & 1) break;
                visited[bit >> 6] |= 1ull << (bit & 63);
                const JInst& in = P.code[pc];
                bool fail = false;
                switch (in.op) {
                case J_CHAR:
                    if (pos < len && P.sets[in.x].has((int)s[pos])) { pc++; pos++; } else fail = true;
                    break;
                case J_SPLIT:
                    stack.push_back({in.y, pos, -1, 0});
                    pc = in.x;
                    break;
                case J_JMP: pc = in.x; break;
                case J_SAVE:
                    stack.push_back({0, 0, in.x, caps[in.x]});
                    caps[in.x] = pos;
                    pc++;
                    break;
                case J_BOL: if (pos == 0) pc++; else fail = true; break;
                case J_EOL: {
                    // Pattern.Dollar (non-multiline): at end, or before the final line terminator
                    bool ok = false;
                    if (pos == len) ok = true;
                    else if (pos == len - 1 && is_line_term((int)s[pos])) {
                        // "\r\n": '$' does not match between \r and \n
                        if (!((int)s[pos] == 10 && pos > 0 && (int)s[pos - 1] == 13)) ok = true;
                    } else if (pos == len - 2 && (int)s[pos] == 13 && (int)s[pos + 1] == 10) ok = true;
                    if (ok) pc++; else fail = true;
                    break;
                }
                case J_MATCH:
                    if (pos == len) {
                        for (int g = 1; g <= P.ngroups; ++g) {
                            int b = caps[2 * g], e = caps[2 * g + 1];
                            if (b < 0 || e < 0) { b = -1; e = -1; }
                            out[2 * (g - 1)] = b; out[2 * (g - 1) + 1] = e;
                        }
                        return true;
                    }
                    fail = true;
                    break;
                }
                if (fail) break;
            }
        }
        return false;
    }
};

// ===========================================================================
// Part 4: Gorp object + extract driver (core/Gorp.java:159-186)
// ===========================================================================
struct Gorp {
    Automata automata;
    std::vector<JProg> regexps;
    int maxGroups = 0;
};

template <typename CH>
static void gorp_extract(const Gorp& g, JMatcher* matchers /* one per extraction, thread-private */,
                         const CH* s, int len, int32_t* match_id, int32_t* caps /* 2*maxGroups */) {
    for (int i = 0; i < 2 * g.maxGroups; ++i) caps[i] = -1;
    int p = poly_walk(g.automata, s, len);                    // _matcher.match(input)
    if (p < 0 || g.automata.accept[p].empty()) { *match_id = -1; return; }  // matchIndexes.length == 0 -> null
    int matchIndex = g.automata.accept[p][0];                 // matchIndexes[0]
    if (matchers[matchIndex].matches(s, len, caps)) { *match_id = matchIndex; return; }
    for (int i = 0; i < 2 * g.maxGroups; ++i) caps[i] = -1;
    *match_id = -2 - matchIndex;                              // ExtractionException (Gorp.java:173-177)
}

}  // namespace orc

// ===========================================================================
// C ABI (ctypes-friendly).  All strings UTF-8.  Return 0 on success.
// ===========================================================================
using namespace orc;

static void set_err(char* err, int errlen, const std::string& m) {
    if (!err || errlen <= 0) return;
    snprintf(err, (size_t)errlen, "%s", m.c_str());
}

static int copy_out(const ustr& s, char* out, int cap) {
    std::string u = utf16_to_utf8(s);
    if ((int)u.size() + 1 > cap) return -(int)u.size() - 1;
    memcpy(out, u.c_str(), u.size() + 1);
    return (int)u.size();
}

extern "C" {

int orc_quote_literal(const char* in, char* out, int cap) {
    return copy_out(quote_literal_as_regexp(utf8_to_utf16(in)), out, cap);
}
int orc_massage_automaton(const char* in, char* out, int cap, char* err, int errlen) {
    try { return copy_out(massage_regexp_for_automaton(utf8_to_utf16(in)), out, cap); }
    catch (std::exception& e) { set_err(err, errlen, e.what()); return -1000000; }
}
int orc_massage_jdk(const char* in, char* out, int cap) {
    return copy_out(massage_regexp_for_jdk(utf8_to_utf16(in)), out, cap);
}

// jdk_rx may be NULL => PolyMatcher only (PolyMatcher.create, PolyMatcher.java:64-70)
int orc_create(const char* const* autom_rx, const char* const* jdk_rx, int n, void** out, char* err, int errlen) {
    try {
        std::unique_ptr<Gorp> g(new Gorp());
        std::vector<Dfa> dfas;
        for (int i = 0; i < n; ++i) {
            ustr ptn = utf8_to_utf16(autom_rx[i]);
            try { dfas.push_back(regex_to_min_dfa(ptn)); }
            catch (OracleError& e) {
                // PolyMatcher.java:79-81
                throw OracleError(std::string("Invalid regexp, ") + e.what() + ", source: " + autom_rx[i]);
            }
        }
        g->automata = automata_construct(dfas);
        if (jdk_rx) {
            for (int i = 0; i < n; ++i) {
                g->regexps.push_back(jdk_compile(utf8_to_utf16(jdk_rx[i])));
                g->maxGroups = std::max(g->maxGroups, g->regexps.back().ngroups);
            }
        }
        *out = g.release();
        return 0;
    } catch (std::exception& e) { set_err(err, errlen, e.what()); return 1; }
}
void orc_destroy(void* h) { delete (Gorp*)h; }

int orc_num_states(void* h) { return ((Gorp*)h)->automata.nbStates(); }
int orc_num_points(void* h) { return ((Gorp*)h)->automata.stride; }
int orc_num_extractions(void* h) { return ((Gorp*)h)->automata.inputRegexpCount; }
int orc_component_states(void* h, int k) { return ((Gorp*)h)->automata.componentStates[k]; }
int orc_max_groups(void* h) { return ((Gorp*)h)->maxGroups; }
int orc_num_groups(void* h, int k) { Gorp* g = (Gorp*)h; return k < (int)g->regexps.size() ? g->regexps[k].ngroups : 0; }
void orc_points(void* h, int32_t* out) { Gorp* g = (Gorp*)h; for (size_t i = 0; i < g->automata.points.size(); ++i) out[i] = g->automata.points[i]; }
void orc_transitions(void* h, int32_t* out) { Gorp* g = (Gorp*)h; memcpy(out, g->automata.transitions.data(), g->automata.transitions.size() * 4); }
// accept(state): writes up to cap indexes, returns count
int orc_accept(void* h, int state, int32_t* out, int cap) {
    Gorp* g = (Gorp*)h;
    auto& a = g->automata.accept[state];
    for (int i = 0; i < (int)a.size() && i < cap; ++i) out[i] = a[i];
    return (int)a.size();
}

// PolyMatcher.match on UTF-16 code units; returns number of matching indexes
int orc_match_utf16(void* h, const uint16_t* s, int len, int32_t* out, int cap) {
    Gorp* g = (Gorp*)h;
    int p = poly_walk(g->automata, s, len);
    if (p < 0) return 0;
    return orc_accept(h, p, out, cap);
}
int orc_match_bytes(void* h, const uint8_t* s, int len, int32_t* out, int cap) {
    Gorp* g = (Gorp*)h;
    int p = poly_walk(g->automata, s, len);
    if (p < 0) return 0;
    return orc_accept(h, p, out, cap);
}

// Gorp.extract on one line. match_id: k >= 0 | -1 (null) | -2-k (ExtractionException)
void orc_extract_utf16(void* h, const uint16_t* s, int len, int32_t* match_id, int32_t* caps) {
    Gorp* g = (Gorp*)h;
    std::vector<JMatcher> ms; ms.reserve(g->regexps.size());
    for (auto& p : g->regexps) ms.emplace_back(p);
    gorp_extract(*g, ms.data(), s, len, match_id, caps);
}
void orc_extract_bytes(void* h, const uint8_t* s, int len, int32_t* match_id, int32_t* caps) {
    Gorp* g = (Gorp*)h;
    std::vector<JMatcher> ms; ms.reserve(g->regexps.size());
    for (auto& p : g->regexps) ms.emplace_back(p);
    gorp_extract(*g, ms.data(), s, len, match_id, caps);
}

// Batch over a CSR byte buffer (bytes are Latin-1 code units), T threads over
// contiguous line ranges.  offsets has n+1 entries of width off_bytes (4 or 8).
// caps is dense: n * 2*maxGroups.  mode: 0 = full extract, 1 = DFA match only
// (match_id = accept[0] or -1).
void orc_extract_batch(void* h, const uint8_t* bytes, const void* offsets, int off_bytes, uint64_t n,
                       int32_t* match_id, int32_t* caps, int nthreads, int mode) {
    Gorp* g = (Gorp*)h;
    auto off = [&](uint64_t i) -> uint64_t {
        return off_bytes == 8 ? ((const uint64_t*)offsets)[i] : ((const uint32_t*)offsets)[i];
    };
    auto work = [&](uint64_t lo, uint64_t hi) {
        std::vector<JMatcher> ms; ms.reserve(g->regexps.size());
        for (auto& p : g->regexps) ms.emplace_back(p);
        std::vector<int32_t> dummy(2 * g->maxGroups + 2);
        for (uint64_t i = lo; i < hi; ++i) {
            const uint8_t* s = bytes + off(i);
            int len = (int)(off(i + 1) - off(i));
            if (mode == 1) {
                int p = poly_walk(g->automata, s, len);
                match_id[i] = (p < 0 || g->automata.accept[p].empty()) ? -1 : g->automata.accept[p][0];
            } else {
                gorp_extract(*g, ms.data(), s, len, &match_id[i], caps ? caps + i * 2 * g->maxGroups : dummy.data());
            }
        }
    };
    if (nthreads <= 1) { work(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        uint64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        th.emplace_back(work, lo, hi);
    }
    for (auto& t : th) t.join();
}

// Stand-alone java.util.regex restatement: compile + matches() on one input
// (used to cross-check the backtracker against Python's re in the tests).
int orc_jdk_matches_utf16(const char* rx, const uint16_t* s, int len, int32_t* caps, int cap_groups, int* ngroups, char* err, int errlen) {
    try {
        JProg P = jdk_compile(utf8_to_utf16(rx));
        *ngroups = P.ngroups;
        std::vector<int32_t> tmp(2 * P.ngroups + 2, -1);
        JMatcher m(P);
        bool ok = m.matches(s, len, tmp.data());
        for (int i = 0; i < 2 * P.ngroups && i < 2 * cap_groups; ++i) caps[i] = ok ? tmp[i] : -1;
        return ok ? 1 : 0;
    } catch (std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

}  // extern "C"
