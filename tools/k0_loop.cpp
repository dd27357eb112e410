// Calls of acx_find on short host haystacks in a loop, without Python: what one call costs at the C ABI (the reference's
// benchmark loop, /root/reference/benchmarks/test_comparison.py:113-124, "short" dataset: ten patterns, 75-byte haystacks).
// build: g++ -O2 -std=c++17 -Iinclude tools/k0_loop.cpp -o tools/k0_loop.bin -Lahocorasick_rs_amd -lacx_hip -Wl,-rpath,$PWD/ahocorasick_rs_amd
// usage: k0_loop.bin [calls] [haystack bytes] [matches: 0 | 1]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "acx.h"

int main(int argc, char **argv) {
    const int calls = argc > 1 ? std::atoi(argv[1]) : 100000;
    const size_t bytes = argc > 2 ? (size_t)std::atoll(argv[2]) : 75;
    const bool matches = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const char *pats[] = {"abc", "hello", "world", "aardvark", "fish", "what", "arbitrarymonkey", "birds", "host7", "host76"};
    std::string blob;
    std::vector<uint64_t> off{0};
    for (const char *p : pats) { blob += p; off.push_back(blob.size()); }
    acx_automaton_t *a = nullptr;
    if (acx_build((const uint8_t *)blob.data(), off.data(), 10, 0, 0, &a)) { std::fprintf(stderr, "%s\n", acx_last_error()); return 1; }
    std::vector<std::string> hays;
    for (int i = 0; i < 64; i++) {
        std::string h = matches ? "arbitrarymonkey says hello to fish host76, 0.123 my friend, but why??? " + std::to_string(i)
                                : "nothing of the kind is said here, 0.123 my friend, and no reason??? " + std::to_string(i);
        while (h.size() < bytes) h += " and so on and so forth";
        h.resize(bytes);
        hays.push_back(h);
    }
    uint64_t total = 0;
    auto run = [&](int n) {
        for (int i = 0; i < n; i++) {
            acx_match_t *m = nullptr;
            uint64_t nm = 0;
            const std::string &h = hays[i & 63];
            if (acx_find(a, (const uint8_t *)h.data(), h.size(), 0, 0, &m, &nm)) { std::fprintf(stderr, "%s\n", acx_last_error()); std::exit(1); }
            total += nm;
            acx_free_matches(m);
        }
    };
    run(1000);
    const auto t0 = std::chrono::steady_clock::now();
    run(calls);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    uint64_t st[ACX_PATH_STATS];
    acx_path_stats(a, st, 0);
    std::printf("%d calls of %zu bytes: %.2f us per call, %llu matches, k0 %llu, resident launches %llu\n", calls, bytes, us / calls,
                (unsigned long long)total, (unsigned long long)st[7], (unsigned long long)st[10]);
    acx_free_automaton(a);
    return 0;
}
