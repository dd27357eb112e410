// automaton.hpp -- host-side automaton compiler of the MI355X-native matcher.
//
// Replaces the *construction* half of the hot path: the reference hands its
// patterns to `AhoCorasickBuilder::build` (/root/reference/src/lib.rs:186-215,
// 401-406; crate aho-corasick 1.1.4).  This is NOT that crate's layout: the
// device only ever runs ONE automaton -- the Standard (all-occurrence)
// Aho-Corasick DFA -- and the match kind (Standard / LeftmostFirst /
// LeftmostLongest / overlapping) is applied afterwards by the resolve kernels
// (kernels.hip).  What is built here, all in BFS numbering so that shallow (hot)
// states have the lowest ids and can be staged into LDS as one prefix:
//
//   * byte -> class map (every byte used by a pattern is its own class, runs
//     of unused bytes share one), stride = next_pow2(classes)
//   * dense state-major transition table  u32[n_states * stride], entry =
//     target id | HAS_OUT<<31 | HAS_OWN<<30
//   * own-terminal pattern lists (CSR) + dictionary-suffix links, so that all
//     patterns ending at a state are enumerated without flattening the lists
//   * level_start[d] = first BFS id of depth d (depth test for anchored walks)
//   * per-pattern length and tie-break rank (len desc, pid asc)
//   * q-gram prefilter bitmaps for the K1b kernel
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace acx {

constexpr uint32_t FLAG_OUT = 0x80000000u; // target state reports >= 1 pattern
constexpr uint32_t FLAG_OWN = 0x40000000u; // target state ends a pattern itself
constexpr uint32_t ID_MASK = 0x3FFFFFFFu;
constexpr uint32_t NONE = 0xFFFFFFFFu;

// K1b prefilter geometry: ONE bitmap of 2^20 bits (128 KiB) resident in LDS.
constexpr uint32_t FILTER_BITS_LOG2 = 20;
constexpr uint32_t FILTER_WORDS = (1u << FILTER_BITS_LOG2) / 32;
constexpr uint32_t FILTER_MAX_Q = 6;
constexpr uint32_t HASH_K1 = 0x9E3779u; // 24-bit odd multipliers (v_mul_u32_u24)
constexpr uint32_t HASH_K2 = 0x85EBCBu;

// The hash both the host (bitmap construction) and K1b (lookup) use, over the
// first q (1..6) bytes p[0..q) of a pattern / of the haystack window:
//   lo = up to three bytes p[0..3)   (24 bits)   * K1
//   hi = the remaining bytes p[3..q) (<= 24 bits) * K2      (mod 2^32)
// Bits 12..28 of h address a byte of the bitmap, bits 29..31 the bit in it.
static inline uint32_t gram_hash(const uint8_t *p, uint32_t q) {
    uint32_t lo = 0, hi = 0;
    for (uint32_t k = 0; k < q && k < 3; k++) lo |= (uint32_t)p[k] << (8 * k);
    for (uint32_t k = 3; k < q; k++) hi |= (uint32_t)p[k] << (8 * (k - 3));
    return lo * HASH_K1 + hi * HASH_K2;
}
static inline uint32_t gram_byte(uint32_t h) { return (h >> 12) & 0x1FFFFu; }
static inline uint32_t gram_bit(uint32_t h) { return h >> 29; }

struct Automaton {
    int match_kind = 0;
    uint64_t n_patterns = 0;
    uint32_t min_len = 0, max_len = 0;
    uint32_t n_classes = 1, stride = 1, stride2 = 0;
    uint32_t n_states = 1;
    uint8_t classes[256];
    std::vector<uint32_t> table;       // n_states * stride
    std::vector<uint32_t> own_off;     // n_states + 1
    std::vector<uint32_t> own_pid;     // patterns ending exactly at the state, id order
    std::vector<uint32_t> dlink;       // nearest proper suffix state with own patterns, or NONE
    std::vector<uint32_t> level_start; // max_len + 2 entries
    std::vector<uint32_t> plen;        // n_patterns
    std::vector<uint32_t> rank;        // n_patterns: rank in (len desc, pid asc)
    // prefilter
    uint32_t filter_q = 0;             // gram length (1..6), 0 = no patterns
    std::vector<uint32_t> filterA;     // FILTER_WORDS (little-endian bytes of the bitmap)
    double filter_density = 0.0;       // fraction of bits set
    // pattern bytes (kept for the synthetic text generator)
    std::vector<uint8_t> blob;
    std::vector<uint64_t> offsets;
};

// Returns empty string on success, otherwise an error message; `code` receives
// an ACX_E* value.
std::string compile(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                    int match_kind, Automaton &out, int &code);

} // namespace acx
