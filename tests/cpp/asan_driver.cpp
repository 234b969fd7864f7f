// ASan/UBSan driver for the host-side compiler (no HIP): definitions -> regex strings -> tables -> blob -> tables.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "gx_common.hpp"
#include "gx_compile.hpp"
#include "gx_dsl.hpp"
#include "gx_hop.hpp"
using namespace gx;

// ---- the hop tier's tables (gx_hop.hpp) walked on the host exactly as gx_hop_dev.hpp walks them, against the dense fused
// automaton: same extraction, same group spans, on lines sampled from the automaton and damaged copies of them ----
namespace {
struct Outcome { int id; std::vector<int> caps; bool operator==(const Outcome& o) const { return id == o.id && caps == o.caps; } };

Outcome dense_outcome(const Tables& T, const std::vector<uint8_t>& line) {
    const RuleTables& U = T.uni;
    std::vector<int> regs(static_cast<size_t>(U.n_regs) + 1, -1);
    uint32_t s = 0;
    for (size_t p = 0; p < line.size(); ++p) {
        const uint32_t w = U.trans[static_cast<size_t>(s) * T.ncls + T.cls256[line[p]]];
        s = w & 0xFFFFu;
        const uint32_t op = w >> 16;
        for (uint32_t j = T.ops_off[op]; op && j < T.ops_off[op + 1]; ++j)
            regs[T.ops[2 * j]] = T.ops[2 * j + 1] == GX_SRC_POS ? static_cast<int>(p) : regs[T.ops[2 * j + 1]];
    }
    Outcome o{U.fin[s] < 0 ? U.fin[s] : 0, {}};
    if (U.fin[s] < 0) return o;
    const int f = U.fin[s], k = T.fin_tags[f];
    o.id = k;
    for (int g = 0; g < T.rules[k].n_groups; ++g) {
        int v[2];
        for (int e = 0; e < 2; ++e) {
            const uint16_t t = T.fin_tags[f + 1 + 2 * g + e];
            v[e] = t == GX_SRC_POS ? static_cast<int>(line.size()) : t == GX_SRC_NIL ? -1 : regs[t];
        }
        if (v[0] < 0 || v[1] < 0) v[0] = v[1] = -1;
        o.caps.push_back(v[0]); o.caps.push_back(v[1]);
    }
    return o;
}

static size_t second_chances_total = 0;
Outcome hop_outcome(const Tables& T, const HopImage& H, const std::vector<uint8_t>& line, size_t* iterations, size_t* second_chances = &second_chances_total) {
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(H.global.data());
    const uint32_t cols = H.row_bytes / 4;
    const uint8_t* hops = H.global.data() + H.hops_off;
    std::vector<uint8_t> cls(line.size() + 32, 0);   // (the staged bytes, as they are; H.full.bytes[b] = the class id of byte b)
    for (size_t i = 0; i < line.size(); ++i) cls[i] = line[i];
    std::vector<int> col(static_cast<size_t>(H.n_regs) + 2, -1);   // register columns (0 = the dummy)
    uint32_t s = H.start;
    size_t p = 0;
    const size_t e = line.size();
    while (p < e) {
        ++*iterations;
        uint32_t r[6];
        memcpy(r, hops + static_cast<size_t>(s) * HOP_REC_BYTES, HOP_REC_BYTES);
        const uint32_t run_lo = r[0] & 0xFFu, run_k = (r[0] >> 8) & 0xFFu, klen = (r[0] >> 16) & 0xFFu;
        size_t n = 0;
        while (n < 16 && p + n < e && run_k != 0x80u && cls[p + n] < 0x80u && cls[p + n] >= run_lo && cls[p + n] <= 0x7Fu - run_k) ++n;
        const size_t q = p + n;
        if (n == 16 || q >= e) { p = q; continue; }
        // (w1: target | off1 << 16 | off2 << 24; w2: column1 * 128 | column2 * 128 << 16; w3: tail pos | lo << 8 | span << 16; w4, w5: single bytes)
        const uint8_t* lits = reinterpret_cast<const uint8_t*>(&r[4]);
        bool ok = q + klen <= e;
        for (int j = 0; j < 8 && ok; ++j) ok = lits[j] == 0 || cls[q + j] == lits[j];
        {
            const uint32_t tb = cls[q + (r[3] & 0xFFu)], lo = (r[3] >> 8) & 0xFFu, span = (r[3] >> 16) & 0xFFu;
            ok = ok && tb >= lo && tb - lo <= span;
        }
        if (ok) {
            col[(r[2] & 0xFFFFu) >> 7] = static_cast<int>(q + ((r[1] >> 16) & 0xFFu));
            col[(r[2] >> 16) >> 7] = static_cast<int>(q + (r[1] >> 24));
            p = q + klen;
            s = r[1] & 0xFFFFu;
        } else {
            // the chain again with the tail byte against the union of the tail's byte set
            {
                const uint8_t* ts = H.full.bytes.data() + H.full.sets_lds + 8u * (r[3] >> 24);
                bool lits_ok = q + klen <= e;
                for (int j = 0; j < 8 && lits_ok; ++j) lits_ok = lits[j] == 0 || cls[q + j] == lits[j];
                const uint8_t tb = cls[q + (r[3] & 0xFFu)];
                bool in = false;
                for (int j = 0; j < 4; ++j) in = in || (ts[4 + j] != 0x80u && tb < 0x80u && tb >= ts[j] && tb <= 0x7Fu - ts[4 + j]);
                if (lits_ok && in && (r[3] >> 24) != 0) {
                    col[(r[2] & 0xFFFFu) >> 7] = static_cast<int>(q + ((r[1] >> 16) & 0xFFu));
                    col[(r[2] >> 16) >> 7] = static_cast<int>(q + (r[1] >> 24));
                    p = q + klen;
                    s = r[1] & 0xFFFFu;
                    ++*second_chances;
                    continue;
                }
            }
            // the second chance: the window at p against the union of the state's loop set (H.full.bytes at sets_lds: u8 lo[4], u8 k[4] per entry)
            const uint8_t* ls = H.full.bytes.data() + H.full.sets_lds + 8u * (r[0] >> 24);
            size_t nu = 0;
            auto in_union = [&](uint8_t b) { for (int j = 0; j < 4; ++j) if (ls[4 + j] != 0x80u && b < 0x80u && b >= ls[j] && b <= 0x7Fu - ls[4 + j]) return true; return false; };
            while (nu < 16 && p + nu < e && in_union(cls[p + nu])) ++nu;
            if (nu > n) { p = p + nu; ++*second_chances; continue; }
            const uint32_t x = rows[static_cast<size_t>(s) * cols + H.full.bytes[cls[q]]];
            col[x >> 16] = static_cast<int>(q);
            s = x & 0xFFFFu;
            p = s == H.dead ? e : q + 1;
        }
    }
    const int32_t info = static_cast<int32_t>(rows[static_cast<size_t>(s) * cols + H.ncls]);
    Outcome o{info < 0 ? info : 0, {}};
    if (H.match_automaton) { o.id = info; return o; }   // (the first accepting extraction, or -1)
    if (info < 0) return o;
    const uint16_t* rec = reinterpret_cast<const uint16_t*>(H.global.data() + H.fin_off + info);
    const size_t tag_slots = 8 * static_cast<size_t>((T.max_groups + 3) / 4);
    o.id = static_cast<int16_t>(rec[tag_slots]);
    for (int g = 0; g < T.rules[o.id].n_groups; ++g) {
        int v[2];
        for (int e2 = 0; e2 < 2; ++e2) {
            // (every tag names a column: a register's, "the length" = the dummy column 0, "unset" = H.col_unset)
            const uint16_t t = rec[2 * g + e2];
            v[e2] = t == H.col_unset ? -1 : t == 0 ? static_cast<int>(line.size()) : col[t / 128];
        }
        if (rec[2 * g] == H.col_unset || rec[2 * g + 1] == H.col_unset) v[0] = v[1] = -1;
        o.caps.push_back(v[0]); o.caps.push_back(v[1]);
    }
    return o;
}

// PolyMatcher.match on the dense match automaton: the first accepting extraction of the final state, or -1
Outcome dense_match(const Tables& T, const std::vector<uint8_t>& line) {
    uint32_t s = 0;
    for (uint8_t b : line) s = T.m_next[static_cast<size_t>(s) * T.ncls + T.cls256[b]];
    return Outcome{T.m_accept_first[s], {}};
}

// returns the number of lines checked (0: the definition is outside the hop tier's limits); throws on a difference
size_t check_hop_tier(const Tables& T, uint64_t seed) {
    size_t checked = 0, iterations = 0, bytes = 0;
    for (uint32_t budget : {48u * 1024u, 3u * HOP_REC_BYTES}) {   // (a tiny LDS budget: another state order)
        HopImage H, M;
        if (!build_hop_image(T, false, budget, 2u * HOP_REC_BYTES, H)) return 0;
        if (!build_hop_image(T, true, budget, 2u * HOP_REC_BYTES, M)) return 0;   // the match automaton alone
        if (H.small.bytes.size() > H.full.bytes.size() || memcmp(H.small.bytes.data(), H.full.bytes.data(), 256) != 0) throw std::runtime_error("hop tier: the two LDS images disagree");
        uint64_t rng = seed * 0x9E3779B97F4A7C15ull + budget;
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        std::vector<std::vector<uint8_t>> of_class(T.ncls);
        for (int b = 0; b < 256; ++b) of_class[T.cls256[b]].push_back(static_cast<uint8_t>(b));
        for (int l = 0; l < 200; ++l) {
            // a walk through the live transitions of the dense automaton, then (two lines in three) some damage
            std::vector<uint8_t> line;
            uint32_t s = 0;
            const size_t want = next() % 150;
            while (line.size() < want) {
                std::vector<int> live;
                for (int c = 0; c < T.ncls; ++c) if ((T.uni.trans[static_cast<size_t>(s) * T.ncls + c] & 0xFFFFu) != static_cast<uint32_t>(T.uni.dead)) live.push_back(c);
                if (live.empty()) break;
                const int c = live[next() % live.size()];
                // (printable bytes first: they are what the chains are built for)
                uint8_t b = of_class[c][next() % of_class[c].size()];
                for (int tries = 0; tries < 8 && !(b >= 0x20 && b < 0x7F); ++tries) b = of_class[c][next() % of_class[c].size()];
                line.push_back(b);
                s = T.uni.trans[static_cast<size_t>(s) * T.ncls + c] & 0xFFFFu;
            }
            for (int d = static_cast<int>(next() % 3); d > 0 && !line.empty(); --d) {
                const size_t at = next() % line.size();
                switch (next() % 3) {
                case 0: line[at] = static_cast<uint8_t>(next()); break;
                case 1: line.erase(line.begin() + at); break;
                default: line.insert(line.begin() + at, line[next() % line.size()]);
                }
            }
            if (!(dense_outcome(T, line) == hop_outcome(T, H, line, &iterations)))
                throw std::runtime_error("hop tier and dense automaton disagree on a line of " + std::to_string(line.size()) + " bytes");
            if (!(dense_match(T, line) == hop_outcome(T, M, line, &iterations)))
                throw std::runtime_error("hop tier and dense match automaton disagree on a line of " + std::to_string(line.size()) + " bytes");
            ++checked;
            bytes += line.size();
        }
    }
    (void)iterations; (void)bytes;
    return checked;
}
}  // namespace
int main(int argc, char** argv) {
    int ok = 0, bad = 0, mutated_ok = 0, mutated_bad = 0;
    size_t hop_lines = 0, hop_defs = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream f(argv[a]);
        std::stringstream ss; ss << f.rdbuf();
        const std::string text = ss.str();
        try {
            auto xs = dsl::read_definition(text, argv[a]);
            std::vector<ustr> au, jd;
            for (auto& x : xs) { std::string p, q; dsl::build_regex_strings(x, p, q); au.push_back(utf8_to_u16(p.c_str())); jd.push_back(utf8_to_u16(q.c_str())); }
            Tables T = compile_tables(au, &jd);
            auto blob = pack_blob(T);
            Tables U = unpack_blob(blob.data(), blob.size());
            (void)U;
            const size_t hop_checked = check_hop_tier(T, static_cast<uint64_t>(a));
            hop_lines += hop_checked;
            hop_defs += hop_checked ? 1 : 0;
            // damaged blobs (truncated, bytes overwritten): refused with an error or accepted, never walked out of bounds --
            // an accepted one is walked over every state and class the way the host-side matchers do
            uint64_t rng = 0x9E3779B97F4A7C15ull ^ blob.size();
            auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            for (int m = 0; m < 60; ++m) {
                std::vector<uint8_t> b2 = blob;
                if (m % 3 == 0) b2.resize(next() % (b2.size() + 1));
                else for (int q = 0; q < 1 + m % 4 && !b2.empty(); ++q) b2[next() % b2.size()] = static_cast<uint8_t>(next());
                try {
                    Tables V = unpack_blob(b2.data(), b2.size());
                    size_t sink = 0;
                    for (int s = 0; s < V.m_states; ++s) {
                        for (int c = 0; c < V.ncls; ++c) sink += V.m_accept_first[V.m_next[static_cast<size_t>(s) * V.ncls + c]] + 2;
                        for (uint32_t i = V.m_accept_off[s]; i < V.m_accept_off[s + 1]; ++i) sink += V.m_accept_list[i];
                    }
                    auto walk = [&](const RuleTables& r) {
                        for (uint32_t w : r.trans) {
                            sink += r.fin[w & 0xFFFFu] + 3;
                            for (uint32_t j = V.ops_off[w >> 16]; (w >> 16) && j < V.ops_off[(w >> 16) + 1]; ++j) sink += V.ops[2 * j] + V.ops[2 * j + 1];
                        }
                    };
                    for (auto& r : V.rules) walk(r);
                    if (V.union_ok) walk(V.uni);
                    if (sink == 1) printf(" ");
                    ++mutated_ok;
                } catch (GxError&) { ++mutated_bad; }
            }
            (void)dsl::dump_json(text, argv[a], "flattened");
            ++ok;
        } catch (GxError& e) { ++bad; }
    }
    printf("asan driver: %d compiled, %d rejected; damaged blobs: %d accepted, %d refused\n", ok, bad, mutated_ok, mutated_bad);
    printf("hop tier: %zu definitions, %zu lines agree with the dense automaton (%zu second chances on loop sets)\n", hop_defs, hop_lines, second_chances_total);
    return 0;
}
