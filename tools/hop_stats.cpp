// Developer tool (host only, no HIP): the hop tier's tables for a definition given as regex pairs, walked over sample lines the
// way gx_hop_dev.hpp walks them, counting what the walk does: iterations, runs, chains taken, exact steps per line.
//   g++ -O2 -std=c++17 -Igorp_amd/csrc -Iinclude tools/hop_stats.cpp gorp_amd/csrc/gx_{compile,regex,host,hop}.cpp -o /tmp/hop_stats
//   /tmp/hop_stats rules.txt lines.txt      (rules.txt: records `automaton regex, byte 0x01, jdk regex, byte 0x02`; tools/hop_stats.py writes both)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "gx_common.hpp"
#include "gx_compile.hpp"
#include "gx_hop.hpp"
using namespace gx;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::vector<ustr> au, jd;
    {
        std::ifstream f(argv[1], std::ios::binary);
        std::string ln;
        while (std::getline(f, ln, '\x02')) {   // (a regexp may hold tabs and line feeds: records end with byte 0x02)
            const size_t t = ln.find('\x01');
            if (t == std::string::npos) continue;
            au.push_back(utf8_to_u16(ln.substr(0, t).c_str()));
            jd.push_back(utf8_to_u16(ln.substr(t + 1).c_str()));
        }
    }
    Tables T = compile_tables(au, &jd);
    for (int pass = 0; pass < 2; ++pass) {
        HopImage H;
        if (!build_hop_image(T, pass == 1, 48u * 1024u, 12u * 1024u, H)) { printf("no hop image\n"); return 1; }
        printf("%s: %u states, %u reachable, %u hot, %u chains, %u runs, %u rows in LDS\n", pass ? "match automaton" : "fused automaton", H.n_states,
               H.n_reachable_hot, H.full.n_hot, H.n_chains, H.n_runs, H.full.n_lds_rows);
        printf("  the tile kernel's LDS image: %zu bytes = class map %u | %u records %u | info words %u | %u branching rows of %u classes %u | final records %u | loop sets %zu; %u register columns\n",
               H.full.bytes.size(), HOP_AT, H.full.n_hot, H.full.n_hot * HOP_REC_BYTES, (H.full.n_hot * 2u + 15u) & ~15u, H.full.n_lds_rows, H.ncls,
               H.full.fin_lds ? H.full.fin_lds - (H.full.info_lds + ((H.full.n_hot * 2u + 15u) & ~15u)) : 0u, H.full.fin_lds ? H.full.sets_lds - H.full.fin_lds : 0u,
               H.full.bytes.size() - H.full.sets_lds, H.n_regs);
        const uint32_t* rows = reinterpret_cast<const uint32_t*>(H.global.data());
        const uint32_t cols = H.row_bytes / 4;
        const uint8_t* hops = H.global.data() + H.hops_off;
        std::ifstream f(argv[2]);
        std::string ln;
        size_t by_byte[256] = {0}, failed_chain = 0, no_chain = 0;
        size_t cold_small = 0, cold_full = 0, cold_switch = 0, next_is_plus1 = 0, cold_moves = 0, in_block4 = 0; uint32_t block = 0xFFFFFFFFu;
        size_t lines = 0, bytes = 0, iters = 0, chains = 0, exacts = 0, exact_cold = 0, run_full = 0, run_bytes = 0, chain_bytes = 0;
        std::vector<std::vector<uint16_t>> traces;
        std::vector<std::vector<uint8_t>> kinds;   // per iteration: 0 a run that fills its window / ends the line, 1 chain, 2 exact step, 3 second chance (loop set / tail set)
        std::vector<std::vector<uint16_t>> st_at, run_at;   // per iteration: the state the lane is in, the bytes its run covers (where the chain's window is read)
        size_t second = 0;
        while (std::getline(f, ln)) {
            traces.emplace_back();
            kinds.emplace_back();
            st_at.emplace_back();
            run_at.emplace_back();
            std::vector<uint8_t> b(ln.begin(), ln.end());
            b.resize(b.size() + 32, 0);
            const size_t e = ln.size();
            uint32_t s = H.start;
            size_t p = 0;
            while (p < e) {
                struct Rec { std::vector<uint16_t>& t; size_t& p; size_t p0; ~Rec() { t.push_back(static_cast<uint16_t>(p - p0)); } } rec{traces.back(), p, p};
                ++iters;
                if (s >= H.small.n_hot) { ++cold_small; if ((s >> 2) != block) { ++cold_switch; block = s >> 2; } }
                if (s >= H.full.n_hot) ++cold_full;
                const uint32_t s_before = s;
                uint32_t r[6];
                memcpy(r, hops + static_cast<size_t>(s) * HOP_REC_BYTES, HOP_REC_BYTES);
                // (a hot state's record as the kernels read it: the LDS image's copy names the state's dense row in LDS, if it has one)
                if (s < H.full.n_hot) memcpy(r, H.full.bytes.data() + HOP_AT + static_cast<size_t>(s) * HOP_REC_BYTES, HOP_REC_BYTES);
                const uint32_t run_lo = r[0] & 0xFFu, run_k = (r[0] >> 8) & 0xFFu, klen = (r[0] >> 16) & 0xFFu;
                size_t n = 0;
                while (n < 16 && p + n < e && run_k != 0x80u && b[p + n] < 0x80u && b[p + n] >= run_lo && b[p + n] <= 0x7Fu - run_k) ++n;
                run_bytes += n;
                const size_t q = p + n;
                st_at.back().push_back(static_cast<uint16_t>(s));
                run_at.back().push_back(static_cast<uint16_t>(n));
                if (n == 16 || q >= e) { p = q; ++run_full; kinds.back().push_back(0); continue; }
                struct Moved { uint32_t from; uint32_t* to; size_t *moves, *plus1, *b4; uint32_t nh; ~Moved() { if (from >= nh && *to != from) { ++*moves; if (*to == from + 1) ++*plus1; if (*to > from && *to < from + 4) ++*b4; } } } moved{s_before, &s, &cold_moves, &next_is_plus1, &in_block4, H.small.n_hot};
                const uint8_t* lits = reinterpret_cast<const uint8_t*>(&r[4]);
                bool ok = q + klen <= e;
                for (int j = 0; j < 8 && ok; ++j) ok = lits[j] == 0 || b[q + j] == lits[j];
                {
                    const uint32_t tb = b[q + (r[3] & 0xFFu)], lo = (r[3] >> 8) & 0xFFu, span = (r[3] >> 16) & 0xFFu;
                    ok = ok && tb >= lo && tb - lo <= span;
                }
                if (ok) { p = q + klen; s = r[1] & 0xFFFFu; ++chains; chain_bytes += klen; kinds.back().push_back(1); }
                else {
                    // the second chances (gx_hop_dev.hpp): the chain with its tail byte in another interval of the tail's set; the window
                    // against the union of the state's loop set
                    {
                        const uint8_t* sets = H.full.bytes.data() + H.full.sets_lds;
                        const uint8_t* ts = sets + 8u * (r[3] >> 24);
                        bool lits_ok = q + klen <= e;
                        for (int j = 0; j < 8 && lits_ok; ++j) lits_ok = lits[j] == 0 || b[q + j] == lits[j];
                        const uint8_t tbyte = b[q + (r[3] & 0xFFu)];
                        bool in = false;
                        for (int j = 0; j < 4; ++j) in = in || (ts[4 + j] != 0x80u && tbyte < 0x80u && tbyte >= ts[j] && tbyte <= 0x7Fu - ts[4 + j]);
                        if (lits_ok && in && (r[3] >> 24) != 0) { p = q + klen; s = r[1] & 0xFFFFu; ++second; kinds.back().push_back(3); continue; }
                        const uint8_t* ls = sets + 8u * (r[0] >> 24);
                        size_t nu = 0;
                        auto in_union = [&](uint8_t x) { for (int j = 0; j < 4; ++j) if (ls[4 + j] != 0x80u && x < 0x80u && x >= ls[j] && x <= 0x7Fu - ls[4 + j]) return true; return false; };
                        while (nu < 16 && p + nu < e && in_union(b[p + nu])) ++nu;
                        if (nu > n) { p = p + nu; ++second; kinds.back().push_back(3); continue; }
                    }
                    kinds.back().push_back(2);
                    ++by_byte[b[q]];
                    if (klen) ++failed_chain; else ++no_chain;
                    const uint32_t x = rows[static_cast<size_t>(s) * cols + H.full.bytes[b[q]]];
                    if (klen != 0 || (r[1] & 0xFFFFu) == 0) ++exact_cold;
                    s = x & 0xFFFFu;
                    p = s == H.dead ? e : q + 1;
                    ++exacts;
                }
            }
            ++lines; bytes += e;
        }
        printf("  %zu lines, %.1f bytes/line; per line: %.1f iterations (%.1f chains of %.1f bytes, %.1f exact steps of which %.1f through a global row, %.1f whole-window runs), %.1f bytes in runs\n",
               lines, double(bytes) / lines, double(iters) / lines, double(chains) / lines, chains ? double(chain_bytes) / chains : 0.0, double(exacts) / lines,
               double(exact_cold) / lines, double(run_full) / lines, double(run_bytes) / lines);
        printf("  iterations at states whose record is not in LDS: %.1f per line with %u hot records, %.1f with %u; block-of-4 switches %.1f per line; moves out of such a state %.1f per line, %.1f to state + 1, %.1f to state + 1..3\n",
               double(cold_small) / lines, H.small.n_hot, double(cold_full) / lines, H.full.n_hot, double(cold_switch) / lines, double(cold_moves) / lines, double(next_is_plus1) / lines, double(in_block4) / lines);
        {
            // the tile kernel: a wave walks 64 consecutive lines until the last of them is through
            size_t tiles = 0, sum_max = 0, sum_all = 0;
            for (size_t t0 = 0; t0 + 64 <= traces.size(); t0 += 64) {
                size_t mx = 0;
                for (size_t q = 0; q < 64; ++q) { mx = std::max(mx, traces[t0 + q].size()); sum_all += traces[t0 + q].size(); }
                sum_max += mx; ++tiles;
            }
            if (tiles) printf("  tiles of 64 consecutive lines: %.1f iterations until the last lane is through, %.1f per lane on average (lanes busy %.0f %%)\n",
                              double(sum_max) / tiles, double(sum_all) / (64.0 * tiles), 100.0 * sum_all / (64.0 * sum_max));
            // in how many of a tile's iterations does SOME lane leave the common path (an exact step or a second chance: the wave takes that branch)
            size_t off_iters = 0;
            for (size_t t0 = 0; t0 + 64 <= kinds.size(); t0 += 64) {
                size_t mx = 0;
                for (size_t q = 0; q < 64; ++q) mx = std::max(mx, kinds[t0 + q].size());
                for (size_t k = 0; k < mx; ++k) {
                    bool off = false;
                    for (size_t q = 0; q < 64 && !off; ++q) off = k < kinds[t0 + q].size() && kinds[t0 + q][k] >= 2;
                    off_iters += off ? 1 : 0;
                }
            }
            if (tiles) printf("  ... of which %.1f have a lane that takes an exact step or a second chance (%.2f second chances per line)\n", double(off_iters) / tiles, double(second) / lines);
            // ---- LDS array cycles of the walk's reads under the bank rules of MI355X_MICROARCH.md (section LDS): only lanes of one group
            // conflict, equal addresses broadcast, every further address on a busy bank is one more cycle.  The tile's lines lie one behind
            // the other in the staging area; a lane that is through keeps reading where it stopped (the kernel does not mask it). ----
            auto dword_access = [](const uint32_t* a, int first, int count, uint32_t banks) {   // one dword per lane, lanes [first, first + count)
                uint32_t seen[64][8]; int n_seen[64] = {0};
                int worst = 1;
                for (int l = first; l < first + count; ++l) {
                    const uint32_t b = (a[l] >> 2) % banks;
                    bool dup = false;
                    for (int k = 0; k < n_seen[b]; ++k) dup = dup || seen[b][k] == (a[l] >> 2);
                    if (!dup && n_seen[b] < 8) seen[b][n_seen[b]++] = a[l] >> 2;
                    worst = std::max(worst, n_seen[b]);
                }
                return worst;
            };
            auto b32 = [&](const uint32_t* a) { return dword_access(a, 0, 32, 32) + dword_access(a, 32, 32, 32); };   // ds_read_b32: 2 x 32 lanes
            auto wide = [&](const uint32_t* a, int dwords, int group, uint32_t banks) {   // a lane reads `dwords` consecutive dwords; groups of `group` contiguous lanes
                int cycles = 0;
                for (int g0 = 0; g0 < 64; g0 += group) {
                    uint32_t seen[64][16]; int n_seen[64] = {0};
                    int worst = 1;
                    for (int l = g0; l < g0 + group; ++l)
                        for (int d = 0; d < dwords; ++d) {
                            const uint32_t w = (a[l] >> 2) + d, b = w % banks;
                            bool dup = false;
                            for (int k = 0; k < n_seen[b]; ++k) dup = dup || seen[b][k] == w;
                            if (!dup && n_seen[b] < 16) seen[b][n_seen[b]++] = w;
                            worst = std::max(worst, n_seen[b]);
                        }
                    cycles += worst;
                }
                return cycles;
            };
            size_t c_win = 0, c_chain = 0, c_rec = 0, n_it = 0, alt_win = 0, alt_chain = 0, alt_rec = 0, alt_rec32 = 0;
            for (size_t t0 = 0; t0 + 64 <= traces.size(); t0 += 64) {
                size_t mx = 0, base[64], pos[64] = {0};
                size_t acc = 0;
                for (size_t q = 0; q < 64; ++q) { mx = std::max(mx, traces[t0 + q].size()); base[q] = acc; for (auto v : traces[t0 + q]) acc += v; }
                uint32_t st[64]; for (size_t q = 0; q < 64; ++q) st[q] = H.start;
                uint32_t run[64] = {0};
                for (size_t k = 0; k < mx; ++k, ++n_it) {
                    uint32_t aw[64], ac[64], ar[64];
                    for (size_t q = 0; q < 64; ++q) {
                        if (k < traces[t0 + q].size()) { st[q] = st_at[t0 + q][k]; run[q] = run_at[t0 + q][k]; }
                        aw[q] = static_cast<uint32_t>(base[q] + pos[q]) & ~3u;
                        ac[q] = static_cast<uint32_t>(base[q] + pos[q] + (k < traces[t0 + q].size() ? run[q] : 0u)) & ~3u;
                        ar[q] = HOP_AT + std::min<uint32_t>(st[q], H.full.n_hot - 1u) * HOP_REC_BYTES;
                    }
                    uint32_t t[64];
                    for (int d = 0; d < 5; ++d) { for (int q = 0; q < 64; ++q) t[q] = aw[q] + 4u * d; c_win += b32(t); }      // ds_read2_b32 x 2 + ds_read_b32
                    for (int d = 0; d < 3; ++d) { for (int q = 0; q < 64; ++q) t[q] = ac[q] + 4u * d; c_chain += b32(t); }    // ds_read2_b32 + ds_read_b32
                    c_rec += wide(ar, 2, 16, 32);                                                                             // ds_read2_b64: two accesses of 4 x 16 lanes
                    for (int q = 0; q < 64; ++q) t[q] = ar[q] + 8u;
                    c_rec += wide(t, 2, 16, 32);
                    for (int q = 0; q < 64; ++q) t[q] = ar[q] + 16u;
                    c_rec += wide(t, 2, 32, 64);                                                                              // ds_read_b64: 2 x 32 lanes, 64 banks
                    // alternatives: the windows as 8-byte aligned ds_read_b64 (three / two of them), the record as three ds_read_b64, the
                    // record in 32 bytes (two ds_read_b128: 4 x 16 lanes, 64 banks)
                    for (int d = 0; d < 3; ++d) { for (int q = 0; q < 64; ++q) t[q] = (aw[q] & ~7u) + 8u * d; alt_win += wide(t, 2, 32, 64); }
                    for (int d = 0; d < 2; ++d) { for (int q = 0; q < 64; ++q) t[q] = (ac[q] & ~7u) + 8u * d; alt_chain += wide(t, 2, 32, 64); }
                    for (int d = 0; d < 3; ++d) { for (int q = 0; q < 64; ++q) t[q] = ar[q] + 8u * d; alt_rec += wide(t, 2, 32, 64); }
                    for (int d = 0; d < 2; ++d) {
                        for (int q = 0; q < 64; ++q) t[q] = HOP_AT + std::min<uint32_t>(st[q], H.full.n_hot - 1u) * 32u + 16u * d;
                        // (ds_read_b128's four groups are not contiguous lanes; contiguous ones model the same load)
                        alt_rec32 += wide(t, 4, 16, 64);
                    }
                    for (size_t q = 0; q < 64; ++q) if (k < traces[t0 + q].size()) pos[q] += traces[t0 + q][k];
                }
            }
            if (n_it) printf("  LDS array cycles per wave iteration under the bank rules: the run's window %.1f (10 without conflicts), the chain's window %.1f (6), the record %.1f (10)\n",
                             double(c_win) / n_it, double(c_chain) / n_it, double(c_rec) / n_it);
            if (n_it) printf("  ... alternatives: the run's window as three aligned ds_read_b64 %.1f, the chain's as two %.1f, the record as three ds_read_b64 %.1f, in 32 bytes as two ds_read_b128 %.1f\n",
                             double(alt_win) / n_it, double(alt_chain) / n_it, double(alt_rec) / n_it, double(alt_rec32) / n_it);
        }
        if (pass == 0) {
            // ---- the hop slice kernel's rounds, replayed: 64 lanes, pieces of 128 bytes from a lane's own position, a lane comes back 24 bytes
            // before the end of its piece; service (results + new lines) when 16 lanes are idle.  leave_at: the walk of a round ends
            // when that many lanes have nothing to walk (64: when all have: the kernel as it is). ----
            for (int leave_at : {64, 48, 32, 24, 16, 8, 4, 1}) {
                const size_t L = traces.size();
                size_t next = 0, rounds = 0, walk_iters = 0, lane_iters = 0, services = 0, staged_bytes = 0;
                struct Lane { bool has = false; size_t line = 0, it = 0, pos = 0, len = 0; } lanes[64];
                auto line_len = [&](size_t i) { size_t t = 0; for (auto v : traces[i]) t += v; return t; };
                for (;;) {
                    int idle = 0; bool any_walking = false;
                    for (auto& l : lanes) { const bool fin = l.has && l.it >= traces[l.line].size(); if (fin || !l.has) ++idle; else any_walking = true; }
                    const bool service = idle >= (leave_at < 16 ? leave_at : 16) || !any_walking;
                    if (service) {
                        ++services;
                        for (auto& l : lanes) if (l.has && l.it >= traces[l.line].size()) l.has = false;
                        for (auto& l : lanes) if (!l.has && next < L * 8) { l.has = true; l.line = next % L; l.it = 0; l.pos = 0; l.len = line_len(l.line); ++next; }
                    }
                    bool any = false; for (auto& l : lanes) any = any || l.has;
                    if (!any) break;
                    ++rounds;
                    size_t lim[64];
                    for (int q = 0; q < 64; ++q) { auto& l = lanes[q]; const size_t left = l.has ? l.len - l.pos : 0; lim[q] = left <= 128 ? l.pos + left : l.pos + 128 - 24; if (l.has) staged_bytes += left < 128 ? left : 128; }
                    for (;;) {
                        int starving = 0, walking = 0;   // starving: lanes that a new round would give something to walk
                        for (int q = 0; q < 64; ++q) {
                            auto& l = lanes[q];
                            const bool unfinished = l.has && l.it < traces[l.line].size();
                            if (unfinished && l.pos < lim[q]) ++walking;
                            else if (unfinished || next < L * 8) ++starving;
                        }
                        if (walking == 0 || starving >= leave_at) break;
                        ++walk_iters;
                        for (int q = 0; q < 64; ++q) { auto& l = lanes[q]; if (l.has && l.it < traces[l.line].size() && l.pos < lim[q]) { l.pos += traces[l.line][l.it++]; ++lane_iters; } }
                    }
                }
                const double lines_done = double(next);
                printf("  leave the walk when %2d lanes have nothing to walk: %.2f rounds and %.1f wave iterations per 64 lines-worth (lane utilisation %.0f %%), %.2f services; staged %.0f bytes per line; cost at 12 K + 1.15 K per iteration: %.0f K cycles per 64 lines, at 7 K: %.0f K\n",
                       leave_at, rounds / (lines_done / 64), walk_iters / (lines_done / 64), 100.0 * lane_iters / (64.0 * walk_iters), services / (lines_done / 64), staged_bytes / lines_done,
                       (rounds * 12.0 + walk_iters * 1.15) / (lines_done / 64), (rounds * 7.0 + walk_iters * 1.15) / (lines_done / 64));
            }
        }
        printf("  exact steps: %.1f per line in states with a chain that did not apply, %.1f in states without one; by byte:", double(failed_chain) / lines, double(no_chain) / lines);
        for (int bt = 0; bt < 256; ++bt) if (by_byte[bt] * 20 > lines) printf(" %c:%.1f", bt >= 0x20 && bt < 0x7F ? bt : '.', double(by_byte[bt]) / lines);
        printf("\n");
    }
    return 0;
}
