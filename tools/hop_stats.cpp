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
        const uint32_t* rows = reinterpret_cast<const uint32_t*>(H.global.data());
        const uint32_t cols = H.row_bytes / 4;
        const uint8_t* hops = H.global.data() + H.hops_off;
        std::ifstream f(argv[2]);
        std::string ln;
        size_t by_byte[256] = {0}, failed_chain = 0, no_chain = 0;
        size_t lines = 0, bytes = 0, iters = 0, chains = 0, exacts = 0, exact_cold = 0, run_full = 0, run_bytes = 0, chain_bytes = 0;
        while (std::getline(f, ln)) {
            std::vector<uint8_t> b(ln.begin(), ln.end());
            b.resize(b.size() + 32, 0);
            const size_t e = ln.size();
            uint32_t s = H.start;
            size_t p = 0;
            while (p < e) {
                ++iters;
                uint32_t r[6];
                memcpy(r, hops + static_cast<size_t>(s) * HOP_REC_BYTES, HOP_REC_BYTES);
                const uint32_t run_lo = r[0] & 0xFFu, run_k = (r[0] >> 8) & 0xFFu, klen = (r[0] >> 16) & 0xFFu;
                size_t n = 0;
                while (n < 16 && p + n < e && run_k != 0x80u && b[p + n] < 0x80u && b[p + n] >= run_lo && b[p + n] <= 0x7Fu - run_k) ++n;
                run_bytes += n;
                const size_t q = p + n;
                if (n == 16 || q >= e) { p = q; ++run_full; continue; }
                const uint8_t* lits = reinterpret_cast<const uint8_t*>(&r[4]);
                bool ok = q + klen <= e;
                for (int j = 0; j < 8 && ok; ++j) ok = lits[j] == 0 || b[q + j] == lits[j];
                {
                    const uint32_t tb = b[q + (r[3] & 0xFFu)], lo = (r[3] >> 8) & 0xFFu, span = (r[3] >> 16) & 0xFFu;
                    ok = ok && tb >= lo && tb - lo <= span;
                }
                if (ok) { p = q + klen; s = r[1] & 0xFFFFu; ++chains; chain_bytes += klen; }
                else {
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
        printf("  exact steps: %.1f per line in states with a chain that did not apply, %.1f in states without one; by byte:", double(failed_chain) / lines, double(no_chain) / lines);
        for (int bt = 0; bt < 256; ++bt) if (by_byte[bt] * 20 > lines) printf(" %c:%.1f", bt >= 0x20 && bt < 0x7F ? bt : '.', double(by_byte[bt]) / lines);
        printf("\n");
    }
    return 0;
}
