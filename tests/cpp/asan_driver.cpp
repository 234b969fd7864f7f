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
    int ok = 0, bad = 0;
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
            (void)dsl::dump_json(text, argv[a], "flattened");
            ++ok;
        } catch (GxError& e) { ++bad; }
    }
    printf("asan driver: %d compiled, %d rejected\n", ok, bad);
    return 0;
}
