// gx_compile.hpp -- host-side table compiler: regex strings -> device tables.
#pragma once
#include "gx_common.hpp"

namespace gx {

// Special 16-bit operand codes used in capture programs and final-tag lists.
enum : uint16_t { GX_SRC_POS = 0xFFFF, GX_SRC_NIL = 0xFFFE };

// One extraction's capture automaton: a tagged DFA (leftmost-greedy priority,
// one symbol of lookahead) over the shared character classes.  It reproduces
// Matcher.matches() + group(1..n) of core/jdkre/JDKRegexpCookedExtraction.java:36-59
// in a single forward pass without backtracking.
struct RuleTables {
    int n_groups = 0;
    int n_states = 0;   // including the absorbing dead state
    int n_regs = 0;
    int dead = 0;
    std::vector<uint32_t> trans;  // [n_states * ncls]: next state | (op-list id << 16)
    std::vector<int32_t> fin;     // [n_states]: offset into Tables::fin_tags of 2*n_groups entries, or -1
    // The automaton would be too large ahead of time (many independent ambiguities: 2^n register patterns where java.util.regex
    // backtracks at run time): the extraction keeps its PROGRAM instead (Tables::pike_*) and the kernels run that -- a Pike VM,
    // thread lists in priority order, linear in line x program.  trans / fin are then a one-state placeholder.
    bool pike = false;
};

struct Tables {
    int n_rules = 0;
    int ncls = 0;
    int max_groups = 0;
    bool has_capture = false;

    // code unit -> class.  Bytes (< 256) through cls256; the rest through the
    // sorted breakpoint list (class of c = hi_cls[last i with hi_lo[i] <= c]).
    uint8_t cls256[256];
    std::vector<uint16_t> hi_lo;
    std::vector<uint16_t> hi_cls;

    // Match automaton: the product of all extractions' automaton-dialect DFAs
    // (Automata.construct, core/autom/Automata.java:57-124), minimised with the
    // full accept set as the distinguishing label, made total with one
    // absorbing dead state (the reference's -1).
    int m_states = 0;
    int m_dead = 0;
    std::vector<uint32_t> m_next;          // [m_states * ncls]
    std::vector<int32_t> m_accept_first;   // [m_states]: accept(p)[0] or -1
    std::vector<uint32_t> m_accept_off;    // [m_states + 1]
    std::vector<int32_t> m_accept_list;    // ascending extraction indexes per state

    std::vector<RuleTables> rules;
    // Fused automaton: all extractions' capture automata joined in priority order, in product with the
    // match automaton -- one pass yields the winning extraction AND its captures.  fin[s] >= 0: offset of
    // a final-tag record that starts with the extraction index; -1: no match (null); -2-k: the match
    // automaton chose k but k's regex rejects (ExtractionException).  Absent (union_ok false) only when
    // it exceeds a size limit.
    bool union_ok = false;
    RuleTables uni;
    std::vector<uint32_t> ops_off;   // [n_oplists + 1]; list 0 is empty
    std::vector<uint16_t> ops;       // (dst, src) pairs, executed in order
    std::vector<uint16_t> fin_tags;  // register id | GX_SRC_POS (= line length) | GX_SRC_NIL

    // Programs of the extractions whose capture automaton is not built ahead of time (RuleTables::pike).  Empty when there is none.
    // An instruction is two words: op | x << 8, y (op: 0 CHAR x = set, 1 SPLIT x preferred over y, 2 JMP x, 3 TAG x, 4 MATCH, 5 FAIL;
    // targets are relative to the extraction's first instruction); a set is eight words, one bit per character class.
    std::vector<uint32_t> pike_off;    // [n_rules + 1], in instructions; equal neighbours: the extraction has its automaton
    std::vector<uint32_t> pike_code;
    std::vector<uint32_t> pike_sets;
    bool has_pike() const { return !pike_code.empty(); }

    int class_of(int c) const {
        if (c < 256) return cls256[c];
        size_t lo = 0, hi = hi_lo.size();
        while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (hi_lo[mid] <= c) lo = mid; else hi = mid; }
        return hi_cls[lo];
    }
};

// jdk == nullptr: match automaton only.
Tables compile_tables(const std::vector<ustr>& automaton_rx, const std::vector<ustr>* jdk_rx);

std::vector<uint8_t> pack_blob(const Tables& t);
Tables unpack_blob(const void* data, size_t size);

}  // namespace gx
