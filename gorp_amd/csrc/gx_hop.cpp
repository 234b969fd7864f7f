// gx_hop.cpp -- builds the HOP tier's tables from the fused automaton (see gx_hop.hpp for what they are and why every
// shortcut in them is a fact about the dense rows).  Host code, once per definition.
#include "gx_hop.hpp"

#include <algorithm>
#include <deque>

namespace gx {

namespace {
inline bool has(const ClassSet& s, int c) { return (s[c >> 6] >> (c & 63)) & 1ull; }
inline void put(ClassSet& s, int c) { s[c >> 6] |= 1ull << (c & 63); }
}  // namespace

std::vector<int> order_classes(const std::map<ClassSet, uint64_t>& weight, int ncls) {
    // ordered partition of the classes, refined set by set: a set that is a union of whole blocks plus parts of the two
    // blocks at its ends can be made contiguous by splitting those two blocks; a set with a hole in the middle stays split
    std::vector<std::vector<int>> blocks(1);
    for (int c = 0; c < ncls; ++c) blocks[0].push_back(c);
    std::vector<std::pair<uint64_t, ClassSet>> by_weight;
    for (auto& w : weight) {
        int size = 0;
        for (int c = 0; c < ncls; ++c) size += has(w.first, c) ? 1 : 0;
        if (size > 1 && size < ncls) by_weight.push_back({w.second * static_cast<uint64_t>(size), w.first});
    }
    std::sort(by_weight.begin(), by_weight.end(), [](const std::pair<uint64_t, ClassSet>& a, const std::pair<uint64_t, ClassSet>& b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    for (auto& ws : by_weight) {
        const ClassSet& S = ws.second;
        int first = -1, last = -1;
        bool ok = true;
        std::vector<int> inside(blocks.size());
        for (size_t b = 0; b < blocks.size(); ++b) {
            int in = 0;
            for (int c : blocks[b]) in += has(S, c) ? 1 : 0;
            inside[b] = in;
            if (in) { if (first < 0) first = static_cast<int>(b); last = static_cast<int>(b); }
        }
        for (int b = first + 1; b < last && ok; ++b) if (inside[b] != static_cast<int>(blocks[b].size())) ok = false;  // a hole in the middle
        if (!ok || first < 0) continue;
        auto split = [&](int b, bool inside_last) {  // block b -> (outside, inside) or (inside, outside)
            std::vector<int> in, out;
            for (int c : blocks[b]) (has(S, c) ? in : out).push_back(c);
            if (in.empty() || out.empty()) return 0;
            blocks[b] = inside_last ? out : in;
            blocks.insert(blocks.begin() + b + 1, inside_last ? in : out);
            return 1;
        };
        if (first == last) split(first, false);
        else {
            split(last, false);          // the inside part first, next to the run
            split(first, true);          // the inside part last
        }
    }
    std::vector<int> new_id(ncls);
    int id = 0;
    for (auto& b : blocks) for (int c : b) new_id[c] = id++;
    return new_id;
}

bool build_hop_image(const Tables& T, bool match_automaton, uint32_t hot_budget_bytes, uint32_t small_budget_bytes, HopImage& out) {
    out = HopImage{};
    if (!match_automaton && (!T.union_ok || !T.has_capture)) { out.refused = 1; return false; }
    const RuleTables& U = T.uni;   // (the fused automaton; not looked at for the match automaton)
    const int ncls = T.ncls;
    const size_t S = static_cast<size_t>(match_automaton ? T.m_states : U.n_states);
    const uint32_t n_regs = match_automaton ? 0u : static_cast<uint32_t>(U.n_regs);
    if (ncls < 1 || ncls > 256 || S < 2 || S > 65536u || n_regs > 253u) { out.refused = 3; return false; }
    const uint32_t dead = static_cast<uint32_t>(match_automaton ? T.m_dead : U.dead);

    // entry(s, c) = successor | register column << 16 (0: no program); every program must be one "register := position"
    std::vector<uint32_t> ent(S * ncls);
    for (size_t i = 0; i < ent.size(); ++i) {
        if (match_automaton) { ent[i] = T.m_next[i]; continue; }
        const uint32_t w = U.trans[i], op = w >> 16;
        uint32_t col = 0;
        if (op) {
            const uint32_t b = T.ops_off[op], e = T.ops_off[op + 1];
            if (e - b != 1 || T.ops[2 * b + 1] != GX_SRC_POS || T.ops[2 * b] >= 254u) { out.refused = 2; return false; }
            col = T.ops[2 * b] + 1u;
        }
        ent[i] = (w & 0xFFFFu) | (col << 16);
    }

    // Everything the walk tests is a BYTE interval (bytes below 0x80 only: a byte with its top bit set always takes the exact
    // step).  The byte set of a class set; how likely text is to take a byte: printable ones (and the tab)
    struct ByteSet { uint64_t w[2] = {0, 0}; };
    auto bytes_of = [&](const ClassSet& cs) {
        ByteSet m;
        for (int bt = 0; bt < 128; ++bt) if (has(cs, T.cls256[bt])) m.w[bt >> 6] |= 1ull << (bt & 63);
        return m;
    };
    auto in_set = [](const ByteSet& m, int bt) { return (m.w[bt >> 6] >> (bt & 63)) & 1ull; };
    auto is_printable = [](int bt) { return (bt >= 0x20 && bt < 0x7F) || bt == 0x09; };
    auto set_weight = [&](const ClassSet& cs) { const ByteSet m = bytes_of(cs); long p = 0; for (int bt = 0; bt < 128; ++bt) if (in_set(m, bt) && is_printable(bt)) ++p; return p; };
    // ... and which of a set's intervals to keep when it has several (\w is four: digits, upper case, '_', lower case): log text is
    // lower case before it is digits before it is upper case
    auto byte_worth = [&](int bt) { return (bt >= 'a' && bt <= 'z') ? 4 : (bt >= '0' && bt <= '9') ? 3 : (bt >= 'A' && bt <= 'Z') ? 2 : bt == ' ' ? 2 : is_printable(bt) ? 1 : 0; };   // (a blank is a space before it is a tab)

    // the groups of every state: entry -> classes (the dead successor is not a group)
    struct Group { uint32_t entry; ClassSet set; long weight; };
    std::vector<std::vector<Group>> groups(S);
    for (size_t s = 0; s < S; ++s) {
        std::map<uint32_t, ClassSet> by_entry;
        for (int c = 0; c < ncls; ++c) {
            const uint32_t e = ent[s * ncls + c];
            if ((e & 0xFFFFu) == dead) continue;
            put(by_entry[e], c);
        }
        for (auto& g : by_entry) groups[s].push_back(Group{g.first, g.second, set_weight(g.second)});
    }

    // the heaviest interval of byte values inside a class set's bytes (ties: the longer one)
    struct Range { int lo = 0, hi = -1; long weight = -1; };
    auto best_range = [&](const ClassSet& cs) {
        const ByteSet m = bytes_of(cs);
        Range best;
        for (int i = 0; i < 128; ++i) {
            if (!in_set(m, i)) continue;
            int j = i;
            long w = 0;
            while (j < 128 && in_set(m, j)) { w += byte_worth(j); ++j; }
            if (w > best.weight || (w == best.weight && j - i > best.hi - best.lo + 1)) best = Range{i, j - 1, w};
            i = j;
        }
        return best;
    };

    // per state: the run (plain self-loop) and the one plausible exit, if there is exactly one
    struct Exit { bool any = false; Range r; uint32_t entry = 0; ClassSet set{}; };
    std::vector<Range> run(S);
    std::vector<long> run_weight(S, 0);
    // LOOP SETS.  A state's run is ONE interval of the bytes it loops on; the others (\\w: digits, upper case, '_' beside lower
    // case; [^,]: what lies below the comma) cost text that uses them an exact step per byte -- `JohnDoe42` in a \\w+ field was six
    // round trips through the dense rows in global memory (config 3 with mixed-case values: 4.7 ms against 0.79).  So the
    // record names the state's loop set: up to four intervals (the heaviest ones, the run among them) in a table of 8-byte
    // entries in LDS, and the walk gives a lane whose run ended and whose chain did not apply a SECOND CHANCE -- the window
    // tested against the union of those intervals -- before it takes the exact step (gx_hop_dev.hpp).  A fact about the dense
    // rows like the run itself (checked below), taken only where the fast tests have failed: text that stays inside the run
    // intervals never gets there.
    std::vector<std::array<uint8_t, 8>> set_table(1, std::array<uint8_t, 8>{0, 0, 0, 0, 0x80, 0x80, 0x80, 0x80});   // entry 0: no interval
    std::map<std::array<uint8_t, 8>, uint32_t> set_index;
    std::vector<uint32_t> set_of(S, 0);
    auto loop_set = [&](const ClassSet& cs) -> uint32_t {
        const ByteSet m = bytes_of(cs);
        std::vector<Range> rs;
        for (int i = 0; i < 128; ++i) {
            if (!in_set(m, i)) continue;
            int j = i;
            long w = 0;
            while (j < 128 && in_set(m, j)) { w += byte_worth(j); ++j; }
            rs.push_back(Range{i, j - 1, w});
            i = j;
        }
        if (rs.size() < 2) return 0u;   // (one interval: the run is all there is)
        std::stable_sort(rs.begin(), rs.end(), [](const Range& a, const Range& b) { return a.weight != b.weight ? a.weight > b.weight : (a.hi - a.lo) > (b.hi - b.lo); });
        if (rs.size() > 4) rs.resize(4);
        std::sort(rs.begin(), rs.end(), [](const Range& a, const Range& b) { return a.lo < b.lo; });
        std::array<uint8_t, 8> key{0, 0, 0, 0, 0x80, 0x80, 0x80, 0x80};
        for (size_t q = 0; q < rs.size(); ++q) { key[q] = static_cast<uint8_t>(rs[q].lo); key[4 + q] = static_cast<uint8_t>(0x7F - rs[q].hi); }
        auto it = set_index.find(key);
        if (it != set_index.end()) return it->second;
        if (set_table.size() >= 256) return 0u;   // (the record has a byte for the index)
        set_table.push_back(key);
        set_index[key] = static_cast<uint32_t>(set_table.size() - 1);
        return static_cast<uint32_t>(set_table.size() - 1);
    };
    std::vector<Exit> exit_of(S);
    std::vector<std::vector<uint32_t>> plausible_targets(S);
    for (size_t s = 0; s < S; ++s) {
        int n_plausible = 0;
        const Group* only = nullptr;
        for (auto& g : groups[s]) {
            if (g.entry == static_cast<uint32_t>(s)) {  // the plain self-loop: same state, no program
                run[s] = best_range(g.set);
                run_weight[s] = g.weight;
                if (s != dead) set_of[s] = loop_set(g.set);
                continue;
            }
            if (g.weight > 0) { ++n_plausible; only = &g; plausible_targets[s].push_back(g.entry & 0xFFFFu); }
        }
        if (n_plausible == 1) {
            exit_of[s].any = true;
            exit_of[s].r = best_range(only->set);
            exit_of[s].entry = only->entry;
            exit_of[s].set = only->set;
        }
    }

    // chains
    // (at most ONE element of a chain is a proper interval -- the "tail": typically the first byte of the next field; the others are
    // single bytes, which the walk compares four at a time with v_msad_u8)
    struct Chain { int klen = 0, tail = -1; uint8_t lo[HOP_CHAIN], span[HOP_CHAIN]; uint32_t target = 0, col[2] = {0, 0}, off[2] = {0, 0}, tail_set = 0; };
    std::vector<Chain> chain(S);
    for (size_t s = 0; s < S; ++s) {
        if (s == dead) continue;
        Chain& ch = chain[s];
        size_t cur = s;
        int nops = 0;
        for (uint32_t k = 0; k < HOP_CHAIN; ++k) {
            // a chain does not run through a field state (a state with a wide plausible self-loop): it ends there, and the
            // next iteration skips the field's bytes as a run.  A blank-run state ([ \t]+ with one blank in the line) it passes.
            if (k > 0 && run_weight[cur] > 4) break;
            const Exit& x = exit_of[cur];
            if (!x.any || x.r.hi < x.r.lo) break;
            const bool ranged = x.r.hi > x.r.lo || x.r.lo == 0;   // (a literal NUL cannot be told from "no byte here": it goes as an interval)
            if (ranged && ch.tail >= 0) break;
            const uint32_t col = x.entry >> 16;
            if (col) {
                if (nops == 2) break;
                ch.col[nops] = col;
                ch.off[nops] = k;
                ++nops;
            }
            if (ranged) {
                ch.tail = static_cast<int>(k);
                // (the tail is ONE interval of the exit's bytes -- lower case for a \\w field; a value that begins with a digit or an upper-case
                // letter failed the whole chain and took its literal bytes as exact steps, one by one.  The exit's other intervals, as a loop
                // set: the walk looks there when the chain's single bytes matched and only the tail did not.)
                ch.tail_set = loop_set(x.set);
            }
            ch.lo[k] = static_cast<uint8_t>(x.r.lo);
            ch.span[k] = static_cast<uint8_t>(x.r.hi - x.r.lo);
            ch.klen = static_cast<int>(k) + 1;
            cur = x.entry & 0xFFFFu;
        }
        ch.target = static_cast<uint32_t>(cur);
    }

    // hot order: breadth-first from the start state over what the walk lands on when lines look like the definition --
    // a chain's target, or (no chain) the targets of the plausible exits
    std::vector<uint32_t> order;
    std::vector<char> seen(S, 0);
    std::deque<uint32_t> queue;
    auto visit = [&](uint32_t s) { if (!seen[s]) { seen[s] = 1; queue.push_back(s); } };
    visit(0);
    while (!queue.empty()) {
        const uint32_t s = queue.front();
        queue.pop_front();
        order.push_back(s);
        if (chain[s].klen > 0) visit(chain[s].target);
        else for (uint32_t t : plausible_targets[s]) visit(t);
    }
    out.n_reachable_hot = static_cast<uint32_t>(order.size());
    for (size_t s = 0; s < S; ++s) if (!seen[s]) order.push_back(static_cast<uint32_t>(s));
    std::vector<uint32_t> perm(S);
    for (size_t i = 0; i < S; ++i) perm[order[i]] = static_cast<uint32_t>(i);
    out.n_hot = std::min<uint32_t>(out.n_reachable_hot, hot_budget_bytes / HOP_REC_BYTES);
    if (out.n_hot == 0) out.n_hot = 1;

    // final records, as gx_walk.hpp's line_result reads them (the layout build_tile_image gives the other tiers):
    // u16 [begin tag, end tag] x max_groups padded to four groups, then the extraction; record 0 = nothing set
    // Round 5: EVERY tag names a column of the wave's register block -- a register's, "the line's length" (which the lane writes there
    // before it reads its result) or "unset" -- so a group's two values are two reads and one test on the tag (the selects on "tag 0:
    // unset, tag 1: the length" were half of the results' instructions).  A group with one end unset has both unset.
    // ("the length" is the dummy column itself -- write-only while the line is walked, free afterwards: no room of its own; "unset"
    // is a column that does not exist: the test is on the tag, what the read returns is not looked at)
    const uint16_t col_len = 0u, col_unset = 0xFF80u;
    out.col_unset = col_unset;
    const size_t tag_slots = 8 * static_cast<size_t>((T.max_groups + 3) / 4), rec_len = tag_slots + 8;
    std::vector<uint16_t> fin_rec(rec_len, 0);
    for (size_t q = 0; q < tag_slots; ++q) fin_rec[q] = col_unset;
    fin_rec[tag_slots] = 0xFFFFu;
    std::map<int32_t, uint32_t> rec_of;
    auto fin_record = [&](int32_t f) -> uint32_t {
        auto it = rec_of.find(f);
        if (it != rec_of.end()) return it->second;
        const int32_t k = static_cast<int32_t>(T.fin_tags[f]);
        const size_t at = fin_rec.size();
        fin_rec.resize(at + rec_len, 0);
        for (size_t q = 0; q < tag_slots; ++q) fin_rec[at + q] = col_unset;
        for (int g = 0; g < T.rules[k].n_groups; ++g) {
            const uint16_t vb = T.fin_tags[f + 1 + 2 * g], ve = T.fin_tags[f + 2 + 2 * g];
            if (vb == GX_SRC_NIL || ve == GX_SRC_NIL) continue;   // (Matcher.group() is null: both ends)
            fin_rec[at + 2 * g] = vb == GX_SRC_POS ? col_len : static_cast<uint16_t>((vb + 1u) * 128u);
            fin_rec[at + 2 * g + 1] = ve == GX_SRC_POS ? col_len : static_cast<uint16_t>((ve + 1u) * 128u);
        }
        fin_rec[at + tag_slots] = static_cast<uint16_t>(k);
        rec_of[f] = static_cast<uint32_t>(at * 2);
        return static_cast<uint32_t>(at * 2);
    };

    // dense rows (the exact step): u32[S][ncls + 1], successor | column << 16 by NEW state index and class id; the last
    // column is the state's info word (byte offset of its final record, or -1 / -2-k)
    const uint32_t cols = static_cast<uint32_t>(ncls) + 1u;
    std::vector<uint32_t> rows(S * cols, 0);
    for (size_t s = 0; s < S; ++s) {
        uint32_t* row = &rows[static_cast<size_t>(perm[s]) * cols];
        for (int c = 0; c < ncls; ++c) {
            const uint32_t e = ent[s * ncls + c];
            row[c] = perm[e & 0xFFFFu] | (e & 0xFFFF0000u);
        }
        // info word: the match automaton's first accepting extraction (or -1); the fused automaton's final record (or -1 / -2-k)
        if (match_automaton) row[ncls] = static_cast<uint32_t>(T.m_accept_first[s]);
        else row[ncls] = U.fin[s] >= 0 ? fin_record(U.fin[s]) : static_cast<uint32_t>(U.fin[s]);
    }
    if (fin_rec.size() * 2 > 0x7FFFu * 16u || T.n_rules > 32000) { out.refused = 3; return false; }  // (a hot state's info word is an int16: offset / 16, or -2-k)

    // hop records
    std::vector<uint32_t> hops(S * (HOP_REC_BYTES / 4), 0);
    for (size_t s = 0; s < S; ++s) {
        uint32_t* r = &hops[static_cast<size_t>(perm[s]) * (HOP_REC_BYTES / 4)];
        const Chain& ch = chain[s];
        uint32_t run_lo = 0, run_k = 0x80;  // none: every byte fails the test
        if (run[s].hi >= run[s].lo && s != dead) { run_lo = static_cast<uint32_t>(run[s].lo); run_k = 0x7Fu - static_cast<uint32_t>(run[s].hi); ++out.n_runs; }
        // w0: run_lo | run_k << 8 | klen << 16 | loop set << 24 ; w1: target | off1 << 16 | off2 << 24 ; w2: column1 * 128 | column2 * 128 << 16 ;
        // w3: tail position (a v_perm selector: 0 .. 7) | tail_lo << 8 | tail_span << 16 | the tail's byte set << 24 (0: the interval is all) ; w4, w5: the chain's single bytes, 0 = no byte to compare
        // (fields sit where SDWA operands can take them: gx_hop_dev.hpp)
        r[0] = run_lo | run_k << 8 | static_cast<uint32_t>(ch.klen) << 16 | set_of[s] << 24;   // (byte 3: the state's loop set, 0 = none)
        r[1] = (ch.klen ? perm[ch.target] : 0u) | ch.off[0] << 16 | ch.off[1] << 24;   // (no chain: the field is the LDS row's address / 4, below)
        r[2] = (ch.col[0] << 7) | (ch.col[1] << 7) << 16;
        uint8_t lits[HOP_CHAIN];
        for (uint32_t k = 0; k < HOP_CHAIN; ++k) lits[k] = (static_cast<int>(k) < ch.klen && static_cast<int>(k) != ch.tail) ? ch.lo[k] : 0;
        if (ch.tail >= 0) r[3] = static_cast<uint32_t>(ch.tail) | static_cast<uint32_t>(ch.lo[ch.tail]) << 8 | static_cast<uint32_t>(ch.span[ch.tail]) << 16 | ch.tail_set << 24;
        else r[3] = 0u | 0u << 8 | 0xFFu << 16;           // no tail: byte 0, any value
        if (ch.klen == 0) r[3] = 0u | 2u << 8 | 0u << 16, lits[0] = 0x01;   // no chain: never matches (byte 0 would have to be 1 and 2 at once); an exact step follows
        else ++out.n_chains;
        memcpy(&r[4], lits, 8);
    }

    // self-check against the dense rows: every byte of a run loops with no program; every byte sequence a chain accepts
    // leads where the chain says, with the chain's programs at the chain's offsets and no others
    for (size_t s = 0; s < S; ++s) {
        const uint32_t* r = &hops[s * (HOP_REC_BYTES / 4)];
        const uint32_t run_lo = r[0] & 0xFFu, run_k = (r[0] >> 8) & 0xFFu;
        if (run_k != 0x80u)
            for (uint32_t bt = run_lo; bt <= 0x7Fu - run_k; ++bt)
                if (rows[s * cols + T.cls256[bt]] != static_cast<uint32_t>(s)) throw GxError(GX_E_ARG, "internal: hop tier run does not match the dense rows");
        {
            const std::array<uint8_t, 8>& ls = set_table[r[0] >> 24];
            for (int q = 0; q < 4; ++q)
                if (ls[4 + q] != 0x80u)
                    for (uint32_t bt = ls[q]; bt <= 0x7Fu - ls[4 + q]; ++bt)
                        if (rows[s * cols + T.cls256[bt]] != static_cast<uint32_t>(s)) throw GxError(GX_E_ARG, "internal: hop tier loop set does not match the dense rows");
        }
        const uint32_t klen = (r[0] >> 16) & 0xFFu;
        const uint8_t* lits = reinterpret_cast<const uint8_t*>(&r[4]);
        const uint32_t tail_pos = r[3] & 0xFFu, tail_lo = (r[3] >> 8) & 0xFFu, tail_span = (r[3] >> 16) & 0xFFu;
        uint32_t cur = static_cast<uint32_t>(s), nops = 0;
        const uint32_t want_col[2] = {(r[2] & 0xFFFFu) >> 7, (r[2] >> 16) >> 7}, want_off[2] = {(r[1] >> 16) & 0xFFu, r[1] >> 24};
        if (klen > HOP_CHAIN || tail_pos >= HOP_CHAIN) throw GxError(GX_E_ARG, "internal: hop tier chain length");
        for (uint32_t k = 0; k < klen; ++k) {
            // what the walk accepts at position k: the single byte, or -- at the tail position -- the interval (with a single byte too: both)
            uint32_t lo = lits[k], hi = lits[k];
            if (lits[k] == 0) {
                if (k != tail_pos) throw GxError(GX_E_ARG, "internal: hop tier chain element without a test");
                lo = tail_lo; hi = tail_lo + tail_span;
                // (the tail's other intervals lead where its interval leads)
                const std::array<uint8_t, 8>& ts = set_table[r[3] >> 24];
                for (int q = 0; q < 4; ++q)
                    if (ts[4 + q] != 0x80u)
                        for (uint32_t bt = ts[q]; bt <= 0x7Fu - ts[4 + q]; ++bt)
                            if (rows[cur * cols + T.cls256[bt]] != rows[cur * cols + T.cls256[lo]]) throw GxError(GX_E_ARG, "internal: hop tier chain tail set is not one group");
            } else if (k == tail_pos && (lits[k] < tail_lo || lits[k] > tail_lo + tail_span)) lo = 1, hi = 0;   // (accepts nothing: fine)
            if (hi > 0x7Fu) throw GxError(GX_E_ARG, "internal: hop tier chain element out of range");
            if (hi < lo) continue;
            const uint32_t e0 = rows[cur * cols + T.cls256[lo]];
            for (uint32_t bt = lo; bt <= hi; ++bt) if (rows[cur * cols + T.cls256[bt]] != e0) throw GxError(GX_E_ARG, "internal: hop tier chain element is not one group");
            if (e0 >> 16) {
                if (nops >= 2 || want_col[nops] != (e0 >> 16) || want_off[nops] != k) throw GxError(GX_E_ARG, "internal: hop tier chain programs");
                ++nops;
            }
            cur = e0 & 0xFFFFu;
        }
        if (klen && tail_pos >= klen && !(tail_lo == 0 && tail_span == 0xFFu)) throw GxError(GX_E_ARG, "internal: hop tier chain tail beyond the chain");
        if (klen && (cur != (r[1] & 0xFFFFu) || (nops < 2 && want_col[nops] != 0))) throw GxError(GX_E_ARG, "internal: hop tier chain target");
    }

    // images
    out.ncls = static_cast<uint32_t>(ncls);
    out.row_bytes = cols * 4u;
    out.n_states = static_cast<uint32_t>(S);
    out.start = perm[0];
    out.dead = perm[dead];
    out.n_regs = n_regs;
    out.match_automaton = match_automaton;
    out.n_loop_sets = static_cast<uint32_t>(set_table.size() - 1);
    const uint8_t* hb = reinterpret_cast<const uint8_t*>(hops.data());
    // One LDS image per hot budget (the state order does not depend on it, so the global image serves both).  The records
    // in the global image keep a zero row address: a state that is hot under one budget is read from there under the other.
    auto make_lds = [&](uint32_t n_hot) {
        HopLds L;
        L.n_hot = n_hot;
        L.bytes.assign(HOP_AT, 0);
        for (int b = 0; b < 256; ++b) L.bytes[b] = T.cls256[b];   // (the exact step's class id)
        L.bytes.insert(L.bytes.end(), hb, hb + static_cast<size_t>(n_hot) * HOP_REC_BYTES);
        while (L.bytes.size() % 16) L.bytes.push_back(0);
        L.info_lds = static_cast<uint32_t>(L.bytes.size());
        for (uint32_t s = 0; s < n_hot; ++s) {
            const int32_t info = static_cast<int32_t>(rows[static_cast<size_t>(s) * cols + ncls]);
            const int16_t v = static_cast<int16_t>(info >= 0 && !match_automaton ? info / 16 : info);   // (an extraction index as it is)
            L.bytes.push_back(static_cast<uint8_t>(v & 0xFF));
            L.bytes.push_back(static_cast<uint8_t>((v >> 8) & 0xFF));
        }
        while (L.bytes.size() % 16) L.bytes.push_back(0);
        // the dense rows of the first hot branching states: an exact step there is an LDS read
        // (compact: u16 successors [ncls, padded to a dword], then u8 register columns [ncls, padded])
        const uint32_t row_budget = 8192u, succ_bytes = (2u * ncls + 3u) & ~3u, row_lds_bytes = succ_bytes + ((ncls + 3u) & ~3u);
        for (uint32_t s = 0; s < n_hot && (L.n_lds_rows + 1) * row_lds_bytes <= row_budget; ++s) {
            const uint32_t orig = order[s];
            if (chain[orig].klen != 0 || plausible_targets[orig].size() < 2) continue;
            const uint32_t at = static_cast<uint32_t>(L.bytes.size());
            if ((at >> 2) > 0xFFFFu) break;
            L.bytes.resize(at + row_lds_bytes, 0);
            for (int cc = 0; cc < ncls; ++cc) {
                const uint32_t e = rows[static_cast<size_t>(s) * cols + cc];
                L.bytes[at + 2 * cc] = static_cast<uint8_t>(e & 0xFFu);
                L.bytes[at + 2 * cc + 1] = static_cast<uint8_t>((e >> 8) & 0xFFu);
                L.bytes[at + succ_bytes + cc] = static_cast<uint8_t>(e >> 16);
            }
            uint32_t w1;   // the record's copy in THIS image gets the row's address / 4 in its (unused) target field
            memcpy(&w1, &L.bytes[HOP_AT + static_cast<size_t>(s) * HOP_REC_BYTES + 4], 4);
            w1 |= at >> 2;
            memcpy(&L.bytes[HOP_AT + static_cast<size_t>(s) * HOP_REC_BYTES + 4], &w1, 4);
            ++L.n_lds_rows;
        }
        while (L.bytes.size() % 16) L.bytes.push_back(0);
        if (!match_automaton && fin_rec.size() * 2 <= 8192u) {  // the final records beside them: a line's result then needs no global read
            L.fin_lds = static_cast<uint32_t>(L.bytes.size());
            const uint8_t* fl = reinterpret_cast<const uint8_t*>(fin_rec.data());
            L.bytes.insert(L.bytes.end(), fl, fl + fin_rec.size() * 2);
            while (L.bytes.size() % 16) L.bytes.push_back(0);
        }
        // the loop sets: u8 lo[4], u8 k[4] per entry (k = 0x7F - hi; 0x80: no interval), entry 0 = none
        L.sets_lds = static_cast<uint32_t>(L.bytes.size());
        for (auto& e : set_table) L.bytes.insert(L.bytes.end(), e.begin(), e.end());
        while (L.bytes.size() % 16) L.bytes.push_back(0);
        return L;
    };
    out.full = make_lds(out.n_hot);
    // the slice kernel's image: when not every reachable state fits anyway, fewer records and more waves (measured on
    // BASELINE configs[4], 2 M lines: 48 KB of records 1.57 ms, 24 KB 1.39, 12 KB 1.26, 6 KB 1.27, 3 KB 1.28)
    out.small = out.n_hot < out.n_reachable_hot ? make_lds(std::min<uint32_t>(out.n_hot, small_budget_bytes / HOP_REC_BYTES)) : out.full;
    const uint8_t* rb = reinterpret_cast<const uint8_t*>(rows.data());
    out.global.assign(rb, rb + rows.size() * 4);
    while (out.global.size() % 16) out.global.push_back(0);
    out.hops_off = static_cast<uint32_t>(out.global.size());
    out.global.insert(out.global.end(), hb, hb + hops.size() * 4);
    while (out.global.size() % 16) out.global.push_back(0);
    out.fin_off = static_cast<uint32_t>(out.global.size());
    const uint8_t* fr = reinterpret_cast<const uint8_t*>(fin_rec.data());
    out.global.insert(out.global.end(), fr, fr + fin_rec.size() * 2);
    while (out.global.size() % 16) out.global.push_back(0);
    // ... and BY STATE, for the hop slice kernel: a finished line's row is one read of a record its state names -- no info word out
    // of the dense rows first, no second round trip for the record behind it (configs[4]: 512 extractions' final records are 32 KB,
    // the info words of its 5 101 reachable states 10 KB: neither fits LDS beside twelve waves).  The record of a state that accepts
    // nothing holds no tags and its info (-1 / -2-k) where the extraction's index stands.
    out.fin_state_rec = static_cast<uint32_t>(rec_len * 2);
    out.fin_state_off = 0;
    if (!match_automaton && S * rec_len * 2 <= (64ull << 20)) {
        out.fin_state_off = static_cast<uint32_t>(out.global.size());
        std::vector<uint16_t> by_state(S * rec_len, 0);
        for (size_t s2 = 0; s2 < S; ++s2) for (size_t q = 0; q < tag_slots; ++q) by_state[s2 * rec_len + q] = col_unset;
        for (size_t s = 0; s < S; ++s) {
            const int32_t info = static_cast<int32_t>(rows[s * cols + ncls]);
            if (info >= 0) memcpy(&by_state[s * rec_len], &fin_rec[static_cast<size_t>(info) / 2], rec_len * 2);
            else by_state[s * rec_len + tag_slots] = static_cast<uint16_t>(static_cast<int16_t>(info));
        }
        const uint8_t* bs = reinterpret_cast<const uint8_t*>(by_state.data());
        out.global.insert(out.global.end(), bs, bs + by_state.size() * 2);
        while (out.global.size() % 16) out.global.push_back(0);
    }
    if (out.global.size() > 0xFFFFFFF0ull) { out.refused = 3; return false; }
    out.ok = true;
    return true;
}

}  // namespace gx
