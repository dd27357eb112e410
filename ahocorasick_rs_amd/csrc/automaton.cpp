// automaton.cpp -- see automaton.hpp.  Host compiler: patterns -> BFS-numbered
// Standard Aho-Corasick DFA + output lists + prefilter bitmaps.
//
// Reference behaviour reproduced (semantics only, /root/reference/src/lib.rs):
//   * empty patterns are an error (204-208, 386-389) -> ACX_EEMPTY
//   * zero patterns is legal and matches nothing
//   * duplicate patterns are legal; all copies are reported (overlapping) and
//     ties resolve to the lowest pattern index
#include "automaton.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "../../include/acx.h"

namespace acx {
// ACX_BUILD_TIMES=1: the host compiler's phases on stderr
struct PhaseClock {
    bool on = std::getenv("ACX_BUILD_TIMES") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    const char *name = "start";
    void next(const char *n) {
        if (on) {
            auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "acx build: %-42s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(now - t).count());
            t = now;
        }
        name = n;
    }
};
#define ACX_PHASE(N) phase_clock.next(N)

static unsigned build_threads() {
    unsigned t = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char *e = std::getenv("ACX_BUILD_THREADS")) t = (unsigned)std::max(1, std::min(64, std::atoi(e)));
    return t;
}
// fn(lo, hi, thread index) over [0, n) cut into one contiguous range per thread (small n: one call)
template <typename F> static void parallel_ranges(uint64_t n, unsigned threads, F fn) {
    if (n < 65536 || threads <= 1) { fn((uint64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    const uint64_t step = (n + threads - 1) / threads;
    unsigned k = 0;
    for (uint64_t a0 = 0; a0 < n; a0 += step, k++) th.emplace_back(fn, a0, std::min(n, a0 + step), k);
    for (auto &t : th) t.join();
}

namespace {

// open-addressing map (parent << 8 | byte) -> child, for trie construction
struct EdgeMap {
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    uint64_t mask = 0, used = 0;
    static uint64_t mixh(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    }
    void init(uint64_t cap_pow2) {
        keys.assign(cap_pow2, ~0ULL);
        vals.assign(cap_pow2, 0);
        mask = cap_pow2 - 1; used = 0;
    }
    void rehash() {
        std::vector<uint64_t> ok; ok.swap(keys);
        std::vector<uint32_t> ov; ov.swap(vals);
        init((mask + 1) * 2);
        for (size_t i = 0; i < ok.size(); i++)
            if (ok[i] != ~0ULL) insert(ok[i], ov[i]);
    }
    uint32_t find(uint64_t k) const {
        uint64_t i = mixh(k) & mask;
        while (keys[i] != ~0ULL) {
            if (keys[i] == k) return vals[i];
            i = (i + 1) & mask;
        }
        return NONE;
    }
    void insert(uint64_t k, uint32_t v) {
        if ((used + 1) * 10 > (mask + 1) * 7) rehash();
        uint64_t i = mixh(k) & mask;
        while (keys[i] != ~0ULL) i = (i + 1) & mask;
        keys[i] = k; vals[i] = v; used++;
    }
};

} // namespace

std::string compile(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                    int match_kind, Automaton &A, int &code, uint64_t dense_limit) {
    PhaseClock phase_clock;
    code = ACX_OK;
    if (match_kind < 0 || match_kind > 2) { code = ACX_EINVAL; return "unknown match kind"; }
    if (n > (1ull << 24)) { code = ACX_ETOOBIG; return "more than 2^24 patterns"; }
    A = Automaton();
    A.match_kind = match_kind;
    A.n_patterns = n;
    uint64_t total = n ? offsets[n] - offsets[0] : 0;
    A.plen.resize(n);
    uint64_t minl = ~0ull, maxl = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) { code = ACX_EINVAL; return "offsets not monotone"; }
        uint64_t L = offsets[i + 1] - offsets[i];
        if (L == 0) { code = ACX_EEMPTY; return "empty pattern"; }
        if (L > 0x3FFFFFFFull) { code = ACX_ETOOBIG; return "pattern longer than 2^30 bytes"; }
        A.plen[i] = (uint32_t)L;
        minl = std::min(minl, L); maxl = std::max(maxl, L);
    }
    A.min_len = n ? (uint32_t)minl : 0;
    A.max_len = (uint32_t)maxl;
    A.blob.assign(blob + (n ? offsets[0] : 0), blob + (n ? offsets[0] : 0) + total);
    A.offsets.resize(n + 1);
    for (uint64_t i = 0; i <= n; i++) A.offsets[i] = n ? offsets[i] - offsets[0] : 0;
    const uint8_t *pb = A.blob.data();

    ACX_PHASE("byte classes");
    // ---- byte classes
    {
        bool bound[256] = {false};
        for (uint64_t i = 0; i < total; i++) {
            uint8_t b = pb[i];
            if (b > 0) bound[b - 1] = true;
            bound[b] = true;
        }
        uint32_t c = 0;
        for (int b = 0; b < 256; b++) {
            A.classes[b] = (uint8_t)c;
            if (bound[b] && b != 255) c++;
        }
        A.n_classes = (uint32_t)A.classes[255] + 1;
        A.stride = 1; A.stride2 = 0;
        while (A.stride < A.n_classes) { A.stride <<= 1; A.stride2++; }
    }

    ACX_PHASE("trie: sort the patterns");
    // ---- trie + BFS numbering in one go, from the patterns in lexicographic order.  The final numbering
    // -- BFS, the children of a state by ascending byte -- lists the states of one depth in the
    // lexicographic order of their paths, which is the order a walk over the SORTED patterns meets them
    // in: a new state at depth d is simply the next id of its depth.  (Until round 3 the trie was built in
    // creation order through a hash map of edges and renumbered by a queue: 1.9 of the 3.8 s of a 10^6-
    // pattern set; now a sort, two linear passes over the pattern bytes, no hash map.)
    const unsigned hw_threads = build_threads();
    std::vector<uint32_t> sorted(n);
    {
        // bucket by first byte, the buckets sorted side by side
        uint64_t bcnt[257] = {0};
        for (uint64_t i = 0; i < n; i++) bcnt[pb[A.offsets[i]] + 1]++;
        for (int b = 0; b < 256; b++) bcnt[b + 1] += bcnt[b];
        {
            uint64_t fill[256];
            for (int b = 0; b < 256; b++) fill[b] = bcnt[b];
            for (uint64_t i = 0; i < n; i++) sorted[fill[pb[A.offsets[i]]]++] = (uint32_t)i;
        }
        auto less = [&](uint32_t a, uint32_t b) {
            const uint64_t la = A.offsets[a + 1] - A.offsets[a], lb = A.offsets[b + 1] - A.offsets[b];
            const int c = std::memcmp(pb + A.offsets[a], pb + A.offsets[b], (size_t)std::min(la, lb));
            if (c) return c < 0;
            if (la != lb) return la < lb;
            return a < b;
        };
        std::atomic<int> next_bucket{0};
        auto work = [&]() {
            for (int b; (b = next_bucket.fetch_add(1)) < 256;)
                if (bcnt[b + 1] - bcnt[b] > 1) std::sort(sorted.begin() + bcnt[b], sorted.begin() + bcnt[b + 1], less);
        };
        if (n < 50000 || hw_threads == 1) {
            work();
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < hw_threads; t++) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
    }
    ACX_PHASE("trie: prefixes, states per depth, ids");
    // pass 1: the longest common prefix with the pattern in front, the states of every depth
    // (memory: 4 B per unit of max_len for the differences, + 4 B per thread when the pass is cut into
    // ranges -- which it is only for many patterns whose histograms are small; a single 1 GiB pattern costs
    // 4 GiB here, not the 24 B x max_len x threads it did)
    std::vector<uint32_t> lcp(n, 0);
    std::vector<int32_t> diff((size_t)A.max_len + 3, 0); // states c + 1 .. L of a pattern are new: +1 at c + 1, -1 at L + 1
    {
        const size_t slots = (size_t)A.max_len + 3;
        const unsigned ranges = (n >= 65536 && hw_threads > 1 && slots * hw_threads * 4 <= ((size_t)64 << 20)) ? hw_threads : 1;
        std::vector<std::vector<int32_t>> part(ranges > 1 ? ranges : 0, std::vector<int32_t>(slots, 0));
        parallel_ranges(n, ranges, [&](uint64_t k0, uint64_t k1, unsigned t) {
            int32_t *pd = ranges > 1 ? part[t].data() : diff.data();
            for (uint64_t k = k0; k < k1; k++) {
                const uint32_t i = sorted[k];
                const uint8_t *x = pb + A.offsets[i];
                const uint64_t L = A.offsets[i + 1] - A.offsets[i];
                uint64_t c = 0;
                if (k) {
                    const uint32_t j = sorted[k - 1];
                    const uint8_t *y = pb + A.offsets[j];
                    const uint64_t Lj = A.offsets[j + 1] - A.offsets[j], m = std::min(L, Lj);
                    while (c < m && x[c] == y[c]) c++;
                }
                lcp[k] = (uint32_t)c;
                pd[c + 1]++;
                pd[L + 1]--;
            }
        });
        for (auto &pd : part)
            for (size_t d = 0; d < pd.size(); d++) diff[d] += pd[d];
    }
    uint64_t n_nodes64 = 0;
    A.level_start.assign((size_t)A.max_len + 2, 0);
    {
        int64_t run = 0; // states of depth d
        for (size_t d = 0; d <= (size_t)A.max_len; d++) {
            run = d == 0 ? 1 : (d == 1 ? (int64_t)diff[1] : run + diff[d]);
            A.level_start[d] = (uint32_t)std::min<uint64_t>(n_nodes64, ID_MASK);
            n_nodes64 += (uint64_t)run;
        }
    }
    diff = {};
    if (n_nodes64 >= ID_MASK) { code = ACX_ETOOBIG; return "more than 2^30 states"; }
    const uint32_t n_nodes = (uint32_t)n_nodes64;
    A.level_start[(size_t)A.max_len + 1] = n_nodes;
    // pass 2: ids, edge bytes, first children, the state every pattern ends in
    std::vector<uint32_t> first_child(n_nodes + 1, NONE);
    std::vector<uint8_t> in_byte(n_nodes, 0);
    std::vector<uint32_t> term_node(n);
    {
        std::vector<uint32_t> next_id(A.level_start.begin(), A.level_start.end() - 1);
        std::vector<uint32_t> path((size_t)A.max_len + 1, 0);
        for (uint64_t k = 0; k < n; k++) {
            const uint32_t i = sorted[k];
            const uint8_t *x = pb + A.offsets[i];
            const uint64_t L = A.offsets[i + 1] - A.offsets[i];
            for (uint64_t d = (uint64_t)lcp[k] + 1; d <= L; d++) {
                const uint32_t id = next_id[d]++;
                in_byte[id] = x[d - 1];
                if (first_child[path[d - 1]] == NONE) first_child[path[d - 1]] = id;
                path[d] = id;
            }
            term_node[i] = path[L];
        }
        first_child[n_nodes] = n_nodes;
        for (uint32_t s2 = n_nodes; s2-- > 0;)
            if (first_child[s2] == NONE) first_child[s2] = first_child[s2 + 1]; // (no children: an empty range)
    }
    sorted = {}; lcp = {};
    // Leftmost kinds: of several identical patterns only the first can ever be reported (the
    // reference's tie-break: lowest index), so the later ones stay out of every table -- a set with
    // the same short pattern thousands of times would otherwise multiply the occurrences the device
    // enumerates before it resolves them.  (Standard keeps them: an overlapping search reports all.)
    std::vector<uint8_t> dup(n, 0);
    if (match_kind != ACX_MATCH_STANDARD) {
        std::vector<uint8_t> seen(n_nodes, 0);
        for (uint64_t i = 0; i < n; i++) {
            if (seen[term_node[i]]) dup[i] = 1;
            seen[term_node[i]] = 1;
        }
    }
    A.n_states = n_nodes;
    // (no limit on n_states * stride here: a dense table is built only below ACX_DENSE_LIMIT, the
    // compressed form -- 13 B per state -- serves every automaton of up to 2^30 states)
    // own lists (stable: pattern id order)
    A.own_off.assign((size_t)n_nodes + 1, 0);
    for (uint64_t i = 0; i < n; i++) if (!dup[i]) A.own_off[term_node[i] + 1]++;
    for (uint32_t s = 0; s < n_nodes; s++) A.own_off[s + 1] += A.own_off[s];
    A.own_pid.assign(n, 0); // (n entries whatever is filed: the C ABI's view has this size)
    {
        std::vector<uint32_t> fill(A.own_off.begin(), A.own_off.end() - 1);
        for (uint64_t i = 0; i < n; i++) if (!dup[i]) A.own_pid[fill[term_node[i]]++] = (uint32_t)i;
    }
    A.own1.assign(n_nodes, OWN1_NONE);
    for (uint32_t s = 0; s < n_nodes; s++) {
        uint32_t c = A.own_off[s + 1] - A.own_off[s];
        if (c == 1) A.own1[s] = A.own_pid[A.own_off[s]];
        else if (c > 1) A.own1[s] = OWN1_MANY;
    }
    term_node = {};

    ACX_PHASE("failure links (classic construction on t");
    // ---- failure links (classic construction on the trie: children of s are the consecutive BFS
    // ids [first_child[s], first_child[s + 1]), bytes ascending)
    auto child_of = [&](uint32_t s, uint8_t b) -> uint32_t {
        uint32_t lo = first_child[s], hi = first_child[s + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (in_byte[mid] < b) lo = mid + 1; else hi = mid;
        }
        return lo < first_child[s + 1] && in_byte[lo] == b ? lo : 0u;
    };
    std::vector<uint32_t> fail(n_nodes, 0);
    // level by level: the links of a level's children only need the links of the levels above
    // (a level of a large automaton is cut into ranges for up to 16 threads)
    auto links_of = [&](uint32_t lo_s, uint32_t hi_s) {
        for (uint32_t s = lo_s; s < hi_s; s++) {
            for (uint32_t c = first_child[s]; c < first_child[s + 1]; c++) {
                const uint8_t b = in_byte[c];
                uint32_t f = fail[s];
                for (;;) {
                    const uint32_t x = child_of(f, b);
                    if (x) { fail[c] = x; break; }
                    if (f == 0) { fail[c] = 0; break; }
                    f = fail[f];
                }
            }
        }
    };
    for (size_t d = 1; d + 1 < A.level_start.size(); d++) { // (the root's children keep 0)
        const uint32_t lo_s = A.level_start[d], hi_s = A.level_start[d + 1];
        if (hi_s - lo_s < 65536 || hw_threads == 1) { links_of(lo_s, hi_s); continue; }
        std::vector<std::thread> th;
        const uint32_t step = (hi_s - lo_s + hw_threads - 1) / hw_threads;
        for (uint32_t a0 = lo_s; a0 < hi_s; a0 += step) th.emplace_back(links_of, a0, std::min(hi_s, a0 + step));
        for (auto &t : th) t.join();
    }
    ACX_PHASE("dictionary suffix links and output flags");
    // ---- dictionary suffix links and output flags
    A.dlink.assign(n_nodes, NONE);
    std::vector<uint32_t> flags(n_nodes, 0);
    for (uint32_t s = 1; s < n_nodes; s++) {
        uint32_t f = fail[s];
        bool f_own = A.own_off[f + 1] > A.own_off[f];
        A.dlink[s] = f_own ? f : A.dlink[f];
        bool own = A.own_off[s + 1] > A.own_off[s];
        if (own) flags[s] |= FLAG_OWN | FLAG_OUT;
        if (A.dlink[s] != NONE) flags[s] |= FLAG_OUT;
    }
    ACX_PHASE("the compressed form (always): trie edges");
    // ---- the compressed form (always): trie edges + failure links -- "NFA" rows of a few bytes per
    // state.  K0's anchored walk and the failure-link walk (K1a on automata without a dense table)
    // run on it; K1b needs neither form.
    A.root_next.assign(256, 0);
    for (uint32_t c = first_child[0]; c < first_child[1]; c++) A.root_next[in_byte[c]] = c;
    A.sflags.resize(n_nodes);
    for (uint32_t s = 0; s < n_nodes; s++) A.sflags[s] = (uint8_t)(flags[s] >> 30);
    ACX_PHASE("the dense form: only while it is worth i");
    // ---- the dense form: only while it is worth its memory (the crate's own rule of thumb: a DFA
    // for small sets, an NFA beyond -- /root/reference/README.md:173-177).  One row per state, built
    // level by level: a row is its failure state's row (one level up at least) + its own edges, so
    // the states of one level are independent -- host threads share them.
    const uint32_t S = A.stride;
    {
        const char *env = std::getenv("ACX_DENSE_LIMIT"); // bytes; tests force the compressed form with 0
        const unsigned __int128 limit = env           ? (unsigned __int128)std::strtoull(env, nullptr, 10)
                                        : dense_limit ? (unsigned __int128)dense_limit
                                                      : ((unsigned __int128)256 << 20);
        A.dense = (unsigned __int128)n_nodes * S * 4 <= limit;
    }
    if (A.dense) {
        A.table.assign((size_t)n_nodes * S, 0);
        auto fill = [&](uint32_t s) {
            uint32_t *row = A.table.data() + (size_t)s * S;
            if (s != 0) std::memcpy(row, A.table.data() + (size_t)fail[s] * S, sizeof(uint32_t) * S);
            for (uint32_t c = first_child[s]; c < first_child[s + 1]; c++) row[A.classes[in_byte[c]]] = c;
        };
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        for (size_t d = 0; d + 1 < A.level_start.size(); d++) {
            const uint32_t lo = A.level_start[d], hi = A.level_start[d + 1];
            if (hi <= lo) continue;
            const unsigned T = (uint64_t)(hi - lo) * S < (1u << 18) ? 1 : hw; // small levels: not worth a thread
            if (T == 1) { for (uint32_t s2 = lo; s2 < hi; s2++) fill(s2); continue; }
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    const uint32_t a = lo + (uint64_t)(hi - lo) * t / T, b = lo + (uint64_t)(hi - lo) * (t + 1) / T;
                    for (uint32_t s2 = a; s2 < b; s2++) fill(s2);
                });
            for (auto &x : th) x.join();
        }
        // the flags of the TARGET state ride on every entry
        auto tag = [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) A.table[i] |= flags[A.table[i]]; };
        const size_t total_e = A.table.size();
        if (total_e < (1u << 20)) tag(0, total_e);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < hw; t++) th.emplace_back(tag, total_e * t / hw, total_e * (t + 1) / hw);
            for (auto &x : th) x.join();
        }
    }
    A.first_child = first_child;
    A.in_byte = in_byte;
    A.fail = fail;

    ACX_PHASE("tie-break rank (len desc, pid asc)");
    // ---- tie-break rank (len desc, pid asc)
    {
        // (a counting sort over the lengths: longest first, pattern ids ascending within a length)
        A.rank.resize(n);
        if (A.max_len <= (1u << 20)) {
            std::vector<uint64_t> at((size_t)A.max_len + 2, 0);
            for (uint64_t i = 0; i < n; i++) at[A.plen[i]]++;
            uint64_t run = 0;
            for (size_t L = (size_t)A.max_len + 1; L-- > 0;) { const uint64_t c = at[L]; at[L] = run; run += c; }
            for (uint64_t i = 0; i < n; i++) A.rank[i] = (uint32_t)at[A.plen[i]]++;
        } else {
            std::vector<uint32_t> idx(n);
            std::iota(idx.begin(), idx.end(), 0u);
            std::stable_sort(idx.begin(), idx.end(),
                             [&](uint32_t a, uint32_t b) { return A.plen[a] > A.plen[b]; });
            for (uint32_t r = 0; r < n; r++) A.rank[idx[r]] = r;
        }
    }

    ACX_PHASE("K1b prefilter tables");
    // ---- K1b prefilter tables
    A.filter_q = 0; A.filter_q2 = 0;
    A.n_short = 0; A.short_min_len = 0; A.long_min_len = 0;
    A.short_xy.clear(); A.short_codes.clear();
    if (n > 0) {
        // ---- the split (automaton.hpp): patterns of 1 and 2 bytes go to the side test, the prefilter tables
        // are built from the others (ACX_NO_SHORT_SPLIT: measurements / the round-3 behaviour -- one set, Q from
        // the shortest pattern whatever it is)
        static const bool no_split = std::getenv("ACX_NO_SHORT_SPLIT") != nullptr;
        const bool split = !no_split && A.min_len <= SHORT_MAX_LEN;
        auto is_short = [&](uint64_t i) { return split && A.plen[i] <= SHORT_MAX_LEN; };
        uint32_t long_min = split ? 0xFFFFFFFFu : A.min_len;
        if (split)
            for (uint64_t i = 0; i < n; i++)
                if (!is_short(i)) long_min = std::min(long_min, A.plen[i]);
        if (long_min == 0xFFFFFFFFu) long_min = FILTER_MAX_Q; // no long pattern at all: empty tables of the usual shape
        A.long_min_len = long_min;
        const uint32_t Q = std::min<uint32_t>(FILTER_MAX_Q, long_min), g = Q - 1;
        const uint32_t Q2 = std::min<uint32_t>(FILTER2_MAX_Q, long_min);
        A.filter_q = Q; A.filter_q2 = Q2;
        const uint32_t gmask = g == 4 ? 0xFFFFFFFFu : ((1u << (8 * g)) - 1);
        // ---- anchors (automaton.hpp): where every long pattern is filed.  How often a Q-gram will turn up in
        // haystacks is estimated from the pattern set itself -- a first-order model of its bytes, P(b0) x prod
        // P(b(k) | b(k-1)): the continuation bytes of a multi-byte character, "ttp:/" behind "h" are certain, a
        // letter behind a letter is 1 in 26 -- and a pattern leaves its beginning for the offset with the
        // rarest gram when that is at least SHIFT_GAIN times rarer (the first offset within a factor of two of
        // the rarest: heads stay short).  Random sets have no such offsets and stay as they are.  The suffix
        // keeps at least long_min bytes, so Q and Q2 are what they were.
        // ACX_NO_ANCHORS: every pattern under its beginning (measurements / the round-3 tables).
        A.shift.assign(n, 0);
        A.max_shift = 0;
        if (std::getenv("ACX_NO_ANCHORS") == nullptr && long_min >= Q && Q >= 3) {
            std::vector<uint32_t> uni(256, 1), big((size_t)65536, 0), row(256, 256); // (+1 smoothing per cell)
            uint64_t total_b = 256;
            for (uint64_t i = 0; i < n; i++) {
                if (is_short(i)) continue;
                const uint8_t *pp = pb + A.offsets[i];
                for (uint32_t k = 0; k < A.plen[i]; k++) {
                    uni[pp[k]]++; total_b++;
                    if (k) { big[(size_t)pp[k - 1] * 256 + pp[k]]++; row[pp[k - 1]]++; }
                }
            }
            std::vector<float> lu(256), lb((size_t)65536);
            for (uint32_t b = 0; b < 256; b++) lu[b] = std::log2((float)uni[b] / (float)total_b);
            for (uint32_t a = 0; a < 256; a++)
                for (uint32_t b = 0; b < 256; b++) lb[(size_t)a * 256 + b] = std::log2((float)(big[(size_t)a * 256 + b] + 1) / (float)row[a]);
            auto gram_bits = [&](const uint8_t *x) { // -log2 of the gram's estimated probability
                float v = -lu[x[0]];
                for (uint32_t k = 1; k < Q; k++) v -= lb[(size_t)x[k - 1] * 256 + x[k]];
                return v;
            };
            const char *gain_env = std::getenv("ACX_SHIFT_GAIN"); // measurements: how much rarer an offset must be
            const float gain_bits = std::log2(gain_env ? std::max(1.0f, (float)std::atof(gain_env)) : (float)SHIFT_GAIN);
            for (uint64_t i = 0; i < n; i++) {
                if (is_short(i)) continue;
                const uint8_t *pp = pb + A.offsets[i];
                const uint32_t dmax = std::min<uint32_t>(SHIFT_MAX, A.plen[i] - long_min);
                if (!dmax) continue;
                const float b0 = gram_bits(pp);
                float best = b0;
                for (uint32_t d = 1; d <= dmax; d++) best = std::max(best, gram_bits(pp + d));
                if (best < b0 + gain_bits) continue;
                uint32_t pick = 0;
                for (uint32_t d = 1; d <= dmax; d++)
                    if (gram_bits(pp + d) >= best - 1.0f) { pick = d; break; }
                A.shift[i] = (uint8_t)pick;
                A.max_shift = std::max(A.max_shift, pick);
            }
        }
        // (from here on a long pattern IS its anchored suffix for every table of the prefilter)
        auto abytes = [&](uint64_t i) { return pb + A.offsets[i] + A.shift[i]; };
        auto alen = [&](uint64_t i) { return A.plen[i] - A.shift[i]; };
        A.filterA.assign(FILTER_WORDS, 0);
        for (uint64_t i = 0; i < n; i++) {
            if (is_short(i)) continue;
            const uint8_t *pp = abytes(i);
            uint32_t wx = (uint32_t)gram_of(pp + 1, g) & gmask; // p[1..1+g)
            uint32_t wy = (uint32_t)gram_of(pp, g) & gmask;     // p[0..g)
            A.filterA[2 * filter_entry(filter_hash(wx))] |= filter_bit(pp[0]) | filter_bit(wx);
            A.filterA[2 * filter_entry(filter_hash(wy)) + 1] |= filter_bit(pp[Q - 1]);
            A.filterA[2 * filter_entry(filter_hash(wy))] |= filter_bit(wy);
        }
        uint64_t set = 0;
        for (uint32_t e = 0; e < (1u << FILTER_ENTRIES_LOG2); e++)
            set += __builtin_popcount(A.filterA[2 * e]);
        A.filter_density = (double)set / (double)(32u << FILTER_ENTRIES_LOG2);

        // ---- prefix table (K1b level 2): exact keys of VARIABLE length in one open-addressing table.
        // A *group* is the set of patterns that share their first Q2 bytes (Q2 = the set-wide
        // minimum, at most 8).  Its keys are the first Lg bytes of its patterns, Lg = min over the
        // group of min(len, 8): a haystack position becomes a prefix hit only if it agrees with some
        // pattern on min(len, 8) bytes of the SHORTEST pattern of its group -- not merely on the
        // set-wide minimum (a set that mixes "xyzzy" with patterns starting with a 4-byte UTF-8
        // character would otherwise turn every occurrence of that character into a hit).
        //   * a group with ONE key: the key is filed under the hash of its first Q2 bytes -- all the
        //     lookup knows before it has seen an entry: ONE gather settles a position (the common case);
        //   * a group with several keys: a REDIRECT entry {the Q2 bytes, next = Lg} is filed there
        //     instead, and the keys under the hash of their own Lg bytes: two dependent gathers.
        // Entry = {key lo, key hi, meta, code}: meta = key length K | next << 4 | filter of displaced keys << 8,
        // 0xFFFFFFFF = empty; next = 0: code = the only pattern with this key, or HIT_LIST | index
        // into blist; next = N > K: look the first N bytes up (salt N).
        std::vector<uint64_t> g1(n);              // first Q2 bytes of every pattern
        for (uint64_t i = 0; i < n; i++) g1[i] = is_short(i) ? 0 : gram_of(abytes(i), Q2);
        std::vector<uint32_t> by_g1; // (without the identical later copies of a pattern: leftmost kinds)
        by_g1.reserve(n);
        for (uint64_t i = 0; i < n; i++) if (!dup[i] && !is_short(i)) by_g1.push_back((uint32_t)i);
        // (sorted by (first Q2 bytes, pattern id); every group then in place by (key bytes, pattern id): the
        // patterns of a key are a stretch of by_g1 in id order -- no vector per group or per key)
        {   // (pairs sorted in place: no lookup of g1[] per comparison)
            struct GI { uint64_t g; uint32_t id; };
            std::vector<GI> gi(by_g1.size());
            for (size_t k = 0; k < by_g1.size(); k++) gi[k] = GI{g1[by_g1[k]], by_g1[k]};
            std::sort(gi.begin(), gi.end(), [](const GI &a, const GI &b) { return a.g != b.g ? a.g < b.g : a.id < b.id; });
            for (size_t k = 0; k < by_g1.size(); k++) by_g1[k] = gi[k].id;
        }
        const size_t n_filed = by_g1.size();
        struct Key { uint64_t gram; uint32_t K, next, salt; uint32_t pid0, npid; }; // patterns: by_g1[pid0 .. pid0 + npid)
        std::vector<Key> keys;
        keys.reserve(n_filed + n_filed / 8);
        for (size_t b = 0; b < n_filed;) {
            size_t e = b;
            uint32_t Lg = FILTER2_MAX_Q;
            while (e < n_filed && g1[by_g1[e]] == g1[by_g1[b]]) {
                Lg = std::min<uint32_t>(Lg, std::min<uint32_t>(alen(by_g1[e]), FILTER2_MAX_Q));
                e++;
            }
            if (e - b > 1)
                std::sort(by_g1.begin() + b, by_g1.begin() + e, [&](uint32_t x, uint32_t y) {
                    const uint64_t gx = gram_of(abytes(x), Lg), gy = gram_of(abytes(y), Lg);
                    return gx != gy ? gx < gy : x < y;
                });
            const size_t first_key = keys.size();
            for (size_t j = b; j < e;) {
                const uint64_t g2 = gram_of(abytes(by_g1[j]), Lg);
                Key k{g2, Lg, 0, Q2, (uint32_t)j, 0};
                while (j < e && gram_of(abytes(by_g1[j]), Lg) == g2) { k.npid++; j++; }
                keys.push_back(k);
            }
            if (keys.size() - first_key > 1) { // several keys: redirect from the Q2 bytes to the keys' own hash
                for (size_t k = first_key; k < keys.size(); k++) keys[k].salt = Lg;
                keys.push_back(Key{g1[by_g1[b]], Q2, Lg, Q2, 0, 0});
            }
            b = e;
        }
        A.n_prefix_keys = (uint32_t)keys.size();
        uint32_t lg = 4;
        // Load.  A group whose entry is not in its home slot turns every haystack position that starts
        // with its first Q2 bytes into a HIT_RETRY (a dependent lookup in k_tile_main; most of them
        // then fail on the full key): linear probing displaces ~14 % of the keys at load 1/4, ~7 % at
        // 1/8.  But a table beyond ~1 MiB starts missing the L2 under K1b's level-2 gathers.  Measured
        // on MI355X (1 GiB), 1/4 -> 1/8:
        //   10^4 patterns (1 -> 2 MiB): k_tile_main 54 -> 51 us, but K1b's FETCH_SIZE +14 % (traffic
        //     1.53x -> 1.73x of the algorithmic bytes): not worth it;
        //   10^5 patterns (8 -> 16 MiB; the bitmap in front of the table keeps the level-1 false
        //     positives away from it): k_tile_main 100 -> 77 us, K1b unchanged: worth it;
        //   1/16 on that set: K1b +7 % (the table leaves the MALL more often), k_tile_main -9 us.
        //   10^4 patterns over a-z + 2/3/4-byte characters (str API; every occurrence of a 4-byte
        //     character is a 5-byte prefix of ~12 patterns: 3.6 M hits per GiB, a third of them retries
        //     of displaced groups): k_tile_main 221 -> 189 us (1/8), 183 (1/16), 171 (1/32, a 5 MiB
        //     table), K1b unchanged: worth it.
        // So: 1/8 for the sets that run with the bitmap (saturated level-1 table); for the sets where
        // more than 5 % of the patterns have a UTF-8 lead byte of a multi-byte character among their
        // first Q2 bytes (few characters per key: many true prefix hits) 1/32 up to 65 536 keys
        // (a table of at most 32 MiB), 1/8 beyond; 1/4 otherwise.
        uint64_t multibyte = 0;
        for (uint64_t i = 0; i < n; i++)
            for (uint32_t k = 0; k < Q2 && !is_short(i); k++)
                if (abytes(i)[k] >= 0xC0) { multibyte++; break; }
        const char *inv_env = std::getenv("ACX_PTAB_INV_LOAD"); // measurements: slots per key
        const size_t inv_load = inv_env ? (size_t)std::max(2, std::atoi(inv_env))
                                        : A.max_shift                                        ? 8 // (anchored sets: the hot beginnings are gone)
                                        : 20 * multibyte > n && keys.size() <= 65536         ? 32
                                          : (A.filter_q == 5 && A.filter_density > 0.2) || 20 * multibyte > n ? 8
                                                                                                : 4;
        while ((1u << lg) < inv_load * keys.size()) lg++;
        A.ptab_log2 = lg;
        A.ptab.assign((size_t)4 << lg, 0);
        for (size_t e = 0; e < ((size_t)1 << lg); e++) A.ptab[4 * e + 2] = PREFIX_EMPTY;
        const uint32_t pmask = (1u << lg) - 1;
        A.pinfo.assign((size_t)4 * n, 0);
        // (the 12 bytes that follow the first Q2 of the ANCHORED suffix; the length is the whole pattern's)
        for (uint64_t i = 0; i < n; i++) {
            const uint8_t *pp = abytes(i);
            const uint32_t L = A.plen[i], La = alen(i);
            uint8_t tail[12] = {0};
            for (uint32_t k = 0; k < 12 && Q2 + k < La; k++) tail[k] = pp[Q2 + k];
            A.pinfo[4 * i] = A.rank[i] | (std::min<uint32_t>(L, 255) << 24);
            std::memcpy(&A.pinfo[4 * i + 1], tail, 12);
        }
        A.phead.clear();
        if (A.max_shift) { // the bytes in front of the anchor (at most SHIFT_MAX = 12 of them are ever compared)
            A.phead.assign((size_t)4 * n, 0);
            for (uint64_t i = 0; i < n; i++)
                std::memcpy(&A.phead[4 * i], pb + A.offsets[i], std::min<uint32_t>(A.plen[i], 12));
        }
        A.blist.clear();
        auto hash_of = [&](uint64_t gram, uint32_t salt) { // salt = the number of key bytes hashed
            const uint64_t m = salt >= 8 ? ~0ull : ((1ull << (8 * salt)) - 1);
            return prefix_home_hash(gram & m, salt);
        };
        std::vector<uint32_t> hash_at((size_t)1 << lg, 0); // hash of the entry stored in each slot
        A.rbloom.assign(REDIRECT_BLOOM_WORDS, 0);
        A.pbits.assign(((size_t)1 << (lg + PREFIX_BITMAP_LOG2)) / 32, 0);
        // Redirect entries first: every haystack position that starts like ANY pattern of the group
        // looks the entry up, so it must sit in its home slot (a displaced one would turn all of
        // them into HIT_RETRY traffic); then the single keys of the other groups, then the keys
        // behind the redirects.
        for (int pass = 0; pass < 3; pass++)
        for (Key &k : keys) {
            if ((k.next ? 0 : (k.salt == Q2 ? 1 : 2)) != pass) continue;
            uint32_t code = 0;
            if (k.next == 0) { // (the patterns of a key are in id order)
                auto coded = [&](uint32_t pid) { return pid | ((uint32_t)A.shift[pid] << CODE_SHIFT_SHIFT); };
                if (k.npid == 1) {
                    code = coded(by_g1[k.pid0]);
                } else {
                    code = 0x80000000u | (uint32_t)A.blist.size();
                    A.blist.push_back(k.npid);
                    for (uint32_t q = 0; q < k.npid; q++) A.blist.push_back(coded(by_g1[k.pid0 + q]));
                }
            }
            const uint32_t h = hash_of(k.gram, k.salt);
            uint32_t idx = prefix_slot(h, lg);
            while (A.ptab[4 * (size_t)idx + 2] != PREFIX_EMPTY) idx = (idx + 1) & pmask;
            uint32_t *en = &A.ptab[4 * (size_t)idx];
            en[0] = (uint32_t)k.gram; en[1] = (uint32_t)(k.gram >> 32);
            en[2] = k.K | (k.next << 4);
            en[3] = code;
            hash_at[idx] = h;
            if (k.salt != Q2) { // a key behind a redirect entry
                const uint32_t bit = redirect_bloom_bit(h);
                A.rbloom[bit >> 5] |= 1u << (bit & 31);
            } else { // filed under its first Q2 bytes: what a lookup starts from
                const uint32_t bit = prefix_bitmap_bit(h, lg);
                A.pbits[bit >> 5] |= 1u << (bit & 31);
            }
        }
        // ---- the short patterns' side tables (automaton.hpp): exact codes + the {X, Y} pair table
        if (split) {
            A.short_xy.assign(SHORT_XY_WORDS, 0);
            A.short_codes.assign(SHORT_CODES, SHORT_NONE);
            A.short_min_len = SHORT_MAX_LEN;
            // the patterns of one short key, in id order, become its code (a list when there are several:
            // duplicates -- the leftmost kinds keep the first only, like every other table)
            std::vector<std::vector<uint32_t>> of_key; // (filled sparsely: index into it from short_codes while building)
            std::vector<uint32_t> key_of;
            for (uint64_t i = 0; i < n; i++) {
                if (!is_short(i)) continue;
                A.n_short++;
                A.short_min_len = std::min(A.short_min_len, A.plen[i]);
                if (dup[i]) continue;
                const uint8_t *pp = pb + A.offsets[i];
                const uint32_t key = A.plen[i] == 1 ? pp[0] : 256u + (pp[0] | ((uint32_t)pp[1] << 8));
                if (A.short_codes[key] == SHORT_NONE) { A.short_codes[key] = (uint32_t)of_key.size(); of_key.emplace_back(); key_of.push_back(key); }
                of_key[A.short_codes[key]].push_back((uint32_t)i);
                if (A.plen[i] == 1) {
                    for (uint32_t m = 0; m < 256; m++) A.short_xy[2 * m] |= filter_bit(pp[0]); // even position: any middle byte
                    A.short_xy[2 * (uint32_t)pp[0] + 1] = 0xFFFFFFFFu;                          // odd position: it IS the middle byte
                } else {
                    A.short_xy[2 * (uint32_t)pp[1]] |= filter_bit(pp[0]);     // even position: the middle byte is its second
                    A.short_xy[2 * (uint32_t)pp[0] + 1] |= filter_bit(pp[1]); // odd position: the middle byte is its first
                }
            }
            for (size_t k = 0; k < of_key.size(); k++) {
                const std::vector<uint32_t> &v = of_key[k];
                uint32_t code = v[0];
                if (v.size() > 1) {
                    code = 0x80000000u | (uint32_t)A.blist.size();
                    A.blist.push_back((uint32_t)v.size());
                    A.blist.insert(A.blist.end(), v.begin(), v.end());
                }
                A.short_codes[key_of[k]] = code;
            }
        }
        // the filter of displaced keys on every home slot: without the lookup's own bit a home slot
        // holding a different key proves absence (one probe)
        for (size_t e = 0; e < ((size_t)1 << lg); e++) {
            if (A.ptab[4 * e + 2] == PREFIX_EMPTY) continue;
            const uint32_t hm = prefix_slot(hash_at[e], lg);
            if (hm != e) A.ptab[4 * (size_t)hm + 2] |= prefix_more_bit(hash_at[e]);
        }
    }
    ACX_PHASE("walk tables");
    build_walk_tables(A);
    ACX_PHASE("end");
    return std::string();
}


// K1a's failureless form (automata of at most 32 byte classes): trie records for every state
// (walk_grec), the depth-3 node by class triple (walk_t3r), and level 1 of the scan: by the symbols
// (low five bits) of three bytes, the symbols a fourth byte can have on a trie path (walk_t3b).
void build_walk_tables(Automaton &A) {
    A.walk_t3b.clear(); A.walk_t3r.clear(); A.walk_grec.clear();
    if (A.n_classes > 32 || A.n_patterns == 0) return;
    std::vector<uint32_t> &t3b = A.walk_t3b, &t3r = A.walk_t3r, &grec = A.walk_grec;
    std::vector<uint32_t> tail_pid;
    std::vector<uint8_t> tail_len;
    const uint32_t NS = A.n_states;
    const unsigned threads = build_threads();
    grec.assign((size_t)4 * NS, 0);
    parallel_ranges(NS, threads, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t s2 = lo; s2 < hi; s2++) {
            uint32_t bm = 0;
            for (uint32_t c = A.first_child[s2]; c < A.first_child[s2 + 1]; c++) bm |= 1u << A.classes[A.in_byte[c]];
            grec[4 * (size_t)s2] = bm;
            grec[4 * (size_t)s2 + 1] = A.first_child[s2] | ((A.sflags[s2] & 1u) ? GREC_OWN : 0u);
            grec[4 * (size_t)s2 + 2] = A.own1[s2];
        }
    });
    // tails: a node without a pattern of its own whose subtree is one chain of at most 8 edges that
    // ends in a leaf with exactly one pattern (a node's child lies one level down: bottom-up, level by level)
    {
        std::vector<uint8_t> tl(NS, 0xFF);
        std::vector<uint32_t> tp(NS, 0);
        for (size_t d = A.level_start.size() - 1; d-- > 1;) {
            const uint32_t l0 = A.level_start[d], l1 = A.level_start[d + 1];
            parallel_ranges(l1 - l0, threads, [&](uint64_t lo, uint64_t hi, unsigned) {
                for (uint64_t s2 = l0 + lo; s2 < l0 + hi; s2++) {
                    const uint32_t c0 = A.first_child[s2], nc = A.first_child[s2 + 1] - c0;
                    const bool own = (A.sflags[s2] & 1u) != 0;
                    if (nc == 0 && own && A.own1[s2] != OWN1_MANY) { tl[s2] = 0; tp[s2] = A.own1[s2]; }
                    else if (nc == 1 && !own && tl[c0] < 8) { tl[s2] = (uint8_t)(tl[c0] + 1); tp[s2] = tp[c0]; }
                }
            });
        }
        tail_len.swap(tl); tail_pid.swap(tp);
    }
    // (a used byte is alone in its class, so the children of a node have distinct classes, ascending
    // like their bytes: child = first child + the set bits below the class)
    auto child = [&](uint32_t s2, uint32_t c) -> uint32_t {
        const uint32_t bm = grec[4 * (size_t)s2];
        if (!((bm >> c) & 1u)) return 0;
        return A.first_child[s2] + (uint32_t)__builtin_popcount(bm & ((1u << c) - 1u));
    };
    t3r.assign(2 * 32768, 0);
    for (uint32_t c0 = 0; c0 < A.n_classes; c0++)
        for (uint32_t c1 = 0; c1 < A.n_classes; c1++)
            for (uint32_t c2 = 0; c2 < A.n_classes; c2++) {
                const uint32_t idx = (c0 * A.n_classes + c1) * A.n_classes + c2;
                const uint32_t n1 = child(0, c0), n2 = n1 ? child(n1, c1) : 0, n3 = n2 ? child(n2, c2) : 0;
                const bool ends = (n1 && (A.sflags[n1] & 1u)) || (n2 && (A.sflags[n2] & 1u)) || (n3 && (A.sflags[n3] & 1u));
                t3r[2 * (size_t)idx] = n3 ? grec[4 * (size_t)n3] : 0u;
                t3r[2 * (size_t)idx + 1] = (n3 ? A.first_child[n3] : 0u) | (ends ? T3R_SHORT : 0u);
            }
    // level 1 of the scan works on symbols (the low five bits of a byte), not classes: no class
    // lookup per byte, and automata with any class map share one kernel.  Every trie path of depth
    // 3 ORs the symbols of its node's children into the entry of its symbol triple; a pattern that
    // ends on the way makes every entry below it pass.
    t3b.assign(K1A_T3B_WORDS, 0);
    auto entry = [&](uint32_t s0, uint32_t s1, uint32_t s2) -> uint32_t & { return t3b[((s0 << 5) | s1) * 33 + s2]; };
    for (uint32_t e1 = A.first_child[0]; e1 < A.first_child[1]; e1++) {
        const uint32_t s0 = A.in_byte[e1] & 31u;
        if (A.sflags[e1] & 1u) {
            for (uint32_t s1 = 0; s1 < 32; s1++) for (uint32_t s2 = 0; s2 < 32; s2++) entry(s0, s1, s2) = ~0u;
            continue;
        }
        for (uint32_t e2 = A.first_child[e1]; e2 < A.first_child[e1 + 1]; e2++) {
            const uint32_t s1 = A.in_byte[e2] & 31u;
            if (A.sflags[e2] & 1u) {
                for (uint32_t s2 = 0; s2 < 32; s2++) entry(s0, s1, s2) = ~0u;
                continue;
            }
            for (uint32_t e3 = A.first_child[e2]; e3 < A.first_child[e2 + 1]; e3++) {
                uint32_t bm = 0;
                if (A.sflags[e3] & 1u) bm = ~0u;
                for (uint32_t e4 = A.first_child[e3]; e4 < A.first_child[e3 + 1]; e4++) bm |= 1u << (A.in_byte[e4] & 31u);
                entry(s0, s1, A.in_byte[e3] & 31u) |= bm;
            }
        }
    }
    // (last: t3r above was filled from the plain records) the tail nodes' records
    parallel_ranges(NS, threads, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t s2 = std::max<uint64_t>(lo, 1); s2 < hi; s2++) {
            if (tail_len[s2] == 0xFF) continue;
            uint32_t by[2] = {0, 0};
            for (uint32_t k = 0, n = (uint32_t)s2; k < tail_len[s2]; k++) {
                n = A.first_child[n];
                by[k >> 2] |= (uint32_t)A.in_byte[n] << (8 * (k & 3));
            }
            grec[4 * (size_t)s2] = by[0];
            grec[4 * (size_t)s2 + 1] = GREC_TAIL | ((uint32_t)tail_len[s2] << 24);
            grec[4 * (size_t)s2 + 2] = tail_pid[s2];
            grec[4 * (size_t)s2 + 3] = by[1];
        }
    });
}

} // namespace acx
