// Developer tool: what ONE gx_extract_one_utf16 costs a C caller (a JNI shim is one): no interpreter between the calls.
// Built and run by tools/bench_single_line.py (argv[1]: the definition file, argv[2]: create flags).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "gorp_hip.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1]);
    std::stringstream ss;
    ss << f.rdbuf();
    const uint32_t flags = static_cast<uint32_t>(std::strtoul(argv[2], nullptr, 10));
    gx_handle* h = nullptr;
    if (gx_create_from_definition(ss.str().c_str(), "definition", flags, &h) != GX_OK) { std::printf("create: %s\n", gx_last_error()); return 1; }
    const std::string lines[2] = {"[123456789]: GET 12ms /index.html?x=1&y=2", "[123456789]: GET 12ms /" + std::string(170, 'a')};
    for (const std::string& line : lines) {
        std::vector<uint16_t> u(line.begin(), line.end());
        int32_t mid = 0, caps[64];
        for (int i = 0; i < 200; ++i) gx_extract_one_utf16(h, u.data(), static_cast<uint32_t>(u.size()), &mid, caps);
        const int reps = 20000;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) gx_extract_one_utf16(h, u.data(), static_cast<uint32_t>(u.size()), &mid, caps);
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("  C caller, %3zu characters: %5.2f us per call (match_id %d, last offset %d)\n", line.size(),
                    std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, mid, caps[7]);
    }
    gx_destroy(h);
    return 0;
}
