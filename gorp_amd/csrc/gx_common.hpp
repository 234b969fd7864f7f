// gx_common.hpp -- shared definitions for the gorp_amd table compiler (host side).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gorp_hip.h"

namespace gx {

struct GxError : std::runtime_error {
    int code;
    GxError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

typedef std::u16string ustr;

// Sorted, disjoint, non-adjacent inclusive ranges of UTF-16 code units.
struct CharSet {
    std::vector<std::pair<int, int>> iv;

    static CharSet single(int c) { CharSet s; s.iv.push_back({c, c}); return s; }
    static CharSet range(int lo, int hi) { CharSet s; if (lo <= hi) s.iv.push_back({lo, hi}); return s; }
    static CharSet all() { return range(0, 0xFFFF); }
    bool empty() const { return iv.empty(); }
    void add(int lo, int hi) { if (lo <= hi) iv.push_back({lo, hi}); }
    void add(const CharSet& o) { iv.insert(iv.end(), o.iv.begin(), o.iv.end()); }
    void canon() {
        std::sort(iv.begin(), iv.end());
        std::vector<std::pair<int, int>> out;
        for (auto& r : iv) {
            if (!out.empty() && r.first <= out.back().second + 1) out.back().second = std::max(out.back().second, r.second);
            else out.push_back(r);
        }
        iv.swap(out);
    }
    CharSet negated() const {  // requires canon()
        CharSet o;
        int next = 0;
        for (auto& r : iv) {
            if (r.first > next) o.iv.push_back({next, r.first - 1});
            next = r.second + 1;
        }
        if (next <= 0xFFFF) o.iv.push_back({next, 0xFFFF});
        return o;
    }
    bool has(int c) const {
        for (auto& r : iv) if (r.first <= c && c <= r.second) return true;
        return false;
    }
};

// Regex AST shared by both dialects.
struct Ast {
    enum Kind { EMPTY, FAIL, SET, CAT, ALT, REP, GROUP } kind = EMPTY;
    std::vector<std::unique_ptr<Ast>> kids;
    CharSet set;             // SET
    int min = 0, max = -1;   // REP (max < 0: unbounded)
    bool greedy = true;      // REP
    int cap = 0;             // GROUP: 1-based capture index, 0 = non-capturing
};
typedef std::unique_ptr<Ast> AstP;

struct Parsed {
    AstP root;
    int ngroups = 0;
};

// Automaton dialect: dk.brics.automaton RegExp with every optional operator off
// (what PolyMatcher feeds it, core/autom/PolyMatcher.java:58,76).  Groups never capture.
Parsed parse_automaton_dialect(const ustr& src);
// JDK dialect: the java.util.regex subset Gorp documents (README.md:197-224).
Parsed parse_jdk_dialect(const ustr& src);

// RegexHelper mirror (gx_host.cpp)
ustr quote_literal_as_regexp(const ustr& text);
ustr massage_regexp_for_automaton(const ustr& pattern);
ustr massage_regexp_for_jdk(const ustr& pattern);

ustr utf8_to_u16(const char* s);
std::string u16_to_utf8(const ustr& s);

}  // namespace gx
