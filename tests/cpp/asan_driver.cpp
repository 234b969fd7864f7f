// ASan/UBSan driver for the host-side compiler (no HIP): definitions -> regex strings -> tables -> blob -> tables.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "gx_common.hpp"
#include "gx_compile.hpp"
#include "gx_dsl.hpp"
using namespace gx;
int main(int argc, char** argv) {
    int ok = 0, bad = 0, mutated_ok = 0, mutated_bad = 0;
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
    return 0;
}
