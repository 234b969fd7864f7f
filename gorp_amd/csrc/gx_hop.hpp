// gx_hop.hpp -- host side of the HOP tier: the fused automaton as "hop records" (run interval + literal chain) over a
// dense-row backstop, for definitions whose dense rows do not fit LDS (64 extractions and more).
//
// What it replaces is still one step per char, Automata.step (core/autom/Automata.java:133-135) folded over the line by
// PolyMatcher.match (core/autom/PolyMatcher.java:123-133) with the capture scan fused in: the walk below reaches the
// same state with the same capture registers, but consumes a RUN of bytes (a field: \S+, \d+, \w+) and a CHAIN of up to
// eight bytes (the literal text between two fields: " key=") per iteration instead of one byte.
//
// Everything is in CLASS space: the kernel maps a tile's bytes to character-class ids while it stages them (one lookup
// per byte and tile), classes are ordered so that the sets the automaton uses are id intervals (order_classes), and
// every test of the walk is an interval test on four or eight class ids at once.
//
// Per state (24 bytes, `HopRec`):
//   run    [run_lo, run_hi]: classes on which the state loops to itself with no capture program.  The walk skips up to
//          16 such bytes with one SWAR test.
//   chain  up to 8 elements (lo, span): "the next klen bytes lie in these class intervals" is a PROVEN shortcut -- the
//          host follows the state's one plausible exit (and its successor's, ...) through the dense rows and records
//          where that path ends (target) and the (at most two) capture programs on it with their offsets.  When the bytes
//          do not match, the walk takes ONE exact step through the state's dense row instead (global memory / L2), so the
//          result never depends on which chains exist: a chain is a fact about the dense rows.  The branching nodes of the
//          literal trie (no chain, several plausible exits) are the states every well-formed line steps through exactly:
//          the first of them keep a copy of their dense row in LDS (their record's target field holds its address / 4).
// States are renumbered "hot first" (breadth-first over chain targets and plausible exits from the start state): the
// records of the first n_hot states are copied into LDS by every workgroup, the others are read from global memory.
#pragma once
#include <array>
#include <map>

#include "gx_compile.hpp"

namespace gx {

typedef std::array<uint64_t, 4> ClassSet;  // one bit per class (<= 256)

// An order of the classes in which as many as possible of the weighted sets are contiguous (greedy partition refinement,
// heaviest sets first).  Returns new_id[class].
std::vector<int> order_classes(const std::map<ClassSet, uint64_t>& weight, int ncls);

constexpr uint32_t HOP_REC_BYTES = 24;
constexpr uint32_t HOP_AT = 272;          // LDS address of the hot records (behind the u8[256] class map)
constexpr uint32_t HOP_CHAIN = 8;         // elements per chain

// what a workgroup copies into LDS: [0, 256): byte -> class id; [HOP_AT, ...): hop records of the states [0, n_hot); their
// info words as int16 (final record offset / 16, or -1 / -2-k) at info_lds; the dense rows of the first hot branching states
// (no chain, several plausible exits; compact: u16 successors, u8 columns); the final records, when they are small, at
// fin_lds (0: read them from the global image)
struct HopLds {
    std::vector<uint8_t> bytes;
    uint32_t n_hot = 0, info_lds = 0, fin_lds = 0, n_lds_rows = 0;
    uint32_t sets_lds = 0;         // the loop sets (gx_hop.cpp): 8 bytes per entry, entry 0 = none
};

struct HopImage {
    bool ok = false;
    uint32_t refused = 0;          // why there is no image: 1 no fused automaton (or no captures), 2 capture programs other than one
                                   // "register := position" per step, 3 beyond a limit of the tier (states, registers, final records)
    HopLds full;                   // as many reachable states as the hot budget holds (the tile kernel)
    HopLds small;                  // the slice kernel's: fewer records, more waves -- the same as `full` when that holds them all
    std::vector<uint8_t> global;   // dense rows u32[n_states][ncls + 1] | hop records of ALL states | final records | final records by state
    uint32_t ncls = 0, row_bytes = 0, n_states = 0, n_hot = 0;
    uint32_t hops_off = 0, fin_off = 0;  // byte offsets in `global`
    uint32_t fin_state_off = 0, fin_state_rec = 0;   // ... of the final records BY STATE (0: none), bytes per record
    uint32_t start = 0, dead = 0;        // state indexes (renumbered)
    uint32_t n_regs = 0;
    uint32_t col_unset = 0;              // byte offset (from a wave's dummy column) of the "unset" column: registers, then "length", then this
    bool match_automaton = false;        // built from the match automaton (PolyMatcher.match batches): an info word is the first
                                         // accepting extraction or -1, there are no programs and no final records
    // diagnostics (gx_stat)
    uint32_t n_reachable_hot = 0, n_chains = 0, n_runs = 0, n_loop_sets = 0;
};

// hot_budget_bytes: LDS bytes the hot records may take.  Returns false (out.ok stays false) when the definition is
// outside the tier's limits: no fused automaton, general capture programs, more than 127 classes, 65 536 states, 254 registers.
// match_automaton: the tables of the match automaton alone (match-only batches) instead of the fused automaton's.
bool build_hop_image(const Tables& T, bool match_automaton, uint32_t hot_budget_bytes, uint32_t small_budget_bytes, HopImage& out);

}  // namespace gx
