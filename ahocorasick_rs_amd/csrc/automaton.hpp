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
constexpr uint32_t OWN1_NONE = 0xFFFFFFFFu; // own1[s]: no pattern ends exactly at s
constexpr uint32_t OWN1_MANY = 0xFFFFFFFEu; // own1[s]: several do (duplicates): use the CSR list

// ---- K1b prefilter geometry and hashes (shared by the host compiler and the kernel)
//
// Level 1 (LDS, 128 KiB): 2^14 entries of two 32-bit words {X, Y}.  With
// Q = min(5, shortest pattern) and g = Q - 1, every pattern prefix p[0..Q) sets
// a two-bit signature in
//     X[e(p[1..1+g))]: bits  p[0] & 31    and  p[1] & 31   (the gram's first byte)
//     Y[e(p[0..g))]:   bits  p[Q-1] & 31  and  p[0] & 31   (the gram's first byte)
// (e = entry index of the gram; Q = 1 degenerates to a single bit), so that the
// haystack positions j (even) and j+1 share ONE 8-byte LDS read: both index
// the table with the g-gram at j+1; position j tests X with byte j, position
// j+1 tests Y with byte j+Q.  A position survives when BOTH signature bits are
// set: true Q-byte prefix hits plus ~0.1 % collisions.
// Level 2 (HBM, L2-resident): exact open-addressing table of pattern prefixes of variable length:
// per group of patterns sharing their first Q2 = min(8, shortest pattern) bytes, the first
// min(8, shortest pattern of the group) bytes; hashed by the Q2 bytes.
constexpr uint32_t FILTER_ENTRIES_LOG2 = 14;
constexpr uint32_t FILTER_WORDS = 2u << FILTER_ENTRIES_LOG2; // u32 words of the X|Y table
constexpr uint32_t FILTER_MAX_Q = 5;
constexpr uint32_t FILTER2_MAX_Q = 8;
constexpr uint32_t HASH_K1 = 0x9E3779u; // 24-bit odd multiplier (v_mad_u32_u24)

#if defined(__HIPCC__)
#define ACX_HD __host__ __device__
#else
#define ACX_HD
#endif

// 32-bit hash of a g-gram W (little-endian, masked to g bytes)
ACX_HD static inline uint32_t filter_hash(uint32_t W) { return (W & 0xFFFFFFu) * HASH_K1 + W; }
ACX_HD static inline uint32_t filter_entry(uint32_t H) { return H >> (32 - FILTER_ENTRIES_LOG2); }
// Signature bits of a table row {X, Y} selected by the hash of a gram W.  A pattern whose bytes
// 1..g are W (it could start one position before the gram) sets bit (p[0] & 31) in X; a pattern
// whose bytes 0..g-1 are W sets bit (p[Q-1] & 31) in Y; BOTH set bit (W & 31) in X -- the gate
// the kernel shares between the two tests of a pair of positions:
//     g = X >> W;   position j: (X >> byte_j) & g;   position j + 1: (Y >> byte_{j+Q}) & g
// (the hardware uses the low 5 bits of a shift amount).  One gate instead of one per word costs
// nothing in selectivity: Y gets sparser by what X gets denser.
ACX_HD static inline uint32_t filter_bit(uint32_t v) { return 1u << (v & 31); }
// 32-bit hash of a Q2-gram (little-endian in a u64, masked to Q2 bytes): two multiplicative
// hashes xor-ed; the table index is taken from the TOP bits (prefix_slot), where a
// multiplicative hash is well mixed.  (K1b evaluates this for every level-1 survivor.)
ACX_HD static inline uint32_t gram_hash2(uint64_t gram) {
    return ((uint32_t)gram * 0x9E3779B1u) ^ ((uint32_t)(gram >> 32) * 0x85EBCA6Bu);
}
// prefix-table entry, word 2 (meta): key length K (bits 3:0, 1..8) | next << 4 (0: final entry;
// N > K: redirect -- look the first N bytes up) | a 16-bit filter of the keys that have this slot
// as their home but live further along the probe sequence (bits 23:8: bit prefix_more_bit(h) of
// every such key's hash h); PREFIX_EMPTY marks a free slot.  A lookup that finds another key in
// its home slot goes on only if its own bit is set -- otherwise the key is not in the table.
constexpr uint32_t PREFIX_EMPTY = 0xFFFFFFFFu;
// Bloom filter (one bit per key, in LDS) of the keys behind redirect entries: a survivor whose group
// redirects is looked up only if the bit of its own N bytes is set (hash = prefix_home_hash(.., N))
constexpr uint32_t REDIRECT_BLOOM_WORDS = 1024; // 32 Kbit
ACX_HD static inline uint32_t redirect_bloom_bit(uint32_t key_hash) { return (key_hash >> 3) & (REDIRECT_BLOOM_WORDS * 32 - 1); }
ACX_HD static inline uint32_t prefix_more_index(uint32_t home_hash) { return (home_hash >> 11) & 15u; }
ACX_HD static inline uint32_t prefix_more_bit(uint32_t home_hash) { return 1u << (8 + prefix_more_index(home_hash)); }
// home-slot hash of the first `salt` bytes of a key (little-endian in a u64, masked): salt = Q2,
// the set-wide minimum key length -- all a lookup knows before it has seen an entry -- or the
// key length a redirect entry names
ACX_HD static inline uint32_t prefix_home_hash(uint64_t gram, uint32_t salt) {
    return gram_hash2(gram) + salt * 0x9E3779B1u;
}
// slot of a Q2-gram in the prefix table (2^log2 entries)
ACX_HD static inline uint32_t prefix_slot(uint32_t h2, uint32_t log2) {
    return log2 ? h2 >> (32 - log2) : 0;
}
// Bitmap in front of the prefix table (8 bits per slot, keyed like the home slot: the next finer
// bits of the same hash): the first Q2 bytes of every group.  A table that has outgrown the L2
// (~10^5 patterns) costs a 128-byte line from the MALL / HBM per lookup; the bitmap is an eighth
// of a byte per slot bit, stays in the L2 and turns away the level-1 false positives (K1b, BIG).
constexpr uint32_t PREFIX_BITMAP_LOG2 = 3;
ACX_HD static inline uint32_t prefix_bitmap_bit(uint32_t h2, uint32_t log2) {
    return h2 >> (32 - log2 - PREFIX_BITMAP_LOG2);
}
static inline uint64_t gram_of(const uint8_t *p, uint32_t q) {
    uint64_t g = 0;
    for (uint32_t k = 0; k < q; k++) g |= (uint64_t)p[k] << (8 * k);
    return g;
}

// ---- K1b, anchors (round 4).  The prefilter can only be as selective as the bytes it is keyed by.  When many
// patterns START alike -- a multi-byte UTF-8 character ("🤦..." opens 345 of cfg5's 10^4 patterns), "http://",
// a common word stem -- their first Q bytes occur in every haystack at the rate of that beginning, level 1
// cannot reject a true prefix, and every occurrence costs the whole level-2 pipeline (cfg5, round 3: 53 M
// survivors per GiB, 5 % of all positions, K1b 2.7x slower than on cfg2).  So a pattern is filed under the
// bytes at its ANCHOR, an offset d = shift[i] in [0, SHIFT_MAX] chosen by the compiler: the offset whose
// Q-gram a first-order model of the pattern set's own bytes calls the rarest, when that is at least SHIFT_GAIN
// times rarer than the beginning (automaton.cpp).  Level 1 and the prefix keys are built from the anchored
// suffix p[d..]; a prefix-table code carries the shift (pattern id | d << 24); a hit at position i means
// "pattern id may START at i - d", and the verification compares the d bytes in front of the hit (phead: the
// pattern's first 12 bytes) as well as the rest.
constexpr uint32_t SHIFT_MAX = 12;
constexpr uint32_t SHIFT_GAIN = 16;     // how much rarer (estimated) an offset's gram must be for a pattern to move there
constexpr uint32_t CODE_PID_MASK = 0x00FFFFFFu, CODE_SHIFT_SHIFT = 24, CODE_SHIFT_MASK = 15u;

// ---- K1b, short patterns (1 and 2 bytes).  A q-gram prefilter keyed by the set's SHORTEST pattern
// degenerates when that is 1 or 2 bytes long (Q = 1: every occurrence of a byte is a survivor), and
// until round 4 such sets left K1b altogether.  Now the set is split: the LONG patterns (>= 3 bytes)
// build the level-1 / level-2 tables above with Q, Q2 taken from THEIR shortest; the SHORT ones are
// found by a side test of the same shape as level 1 -- positions j, j+1 share ONE 8-byte LDS read of
// short_xy[middle byte b(j+1)] = {X, Y}:
//     position j   survives iff bit (b(j)   & 31) of X:  a 2-byte pattern (b(j), b(j+1)) or a 1-byte pattern b(j)
//     position j+1 survives iff bit (b(j+2) & 31) of Y:  a 2-byte pattern (b(j+1), b(j+2)) or a 1-byte pattern b(j+1)
// (a superset where bytes alias in their low five bits) -- and settled exactly by short_codes: [b0] the
// code of the 1-byte pattern b0, [256 + (b0 | b1 << 8)] of the 2-byte pattern (b0, b1); a code is a
// pattern id, 0x80000000 | index into blist (duplicates), or SHORT_NONE.
constexpr uint32_t SHORT_MAX_LEN = 2;
constexpr uint32_t SHORT_XY_WORDS = 512;
constexpr uint32_t SHORT_CODES = 256 + 65536;
constexpr uint32_t SHORT_NONE = 0xFFFFFFFFu;

// K1a's failureless walk: the records of the trie
// entry of the symbols (s0, s1, s2) in walk_t3b: word ((s0 << 5 | s1) * 33 + s2) -- the odd stride puts the
// sum of two symbols into the LDS bank (with a stride of 32 every position in front of a space met in ONE bank)
constexpr uint32_t K1A_T3B_WORDS = 1024 * 33;
constexpr uint32_t T3R_SHORT = 0x80000000u; // a pattern ends within the first three levels: walk from the root
constexpr uint32_t GREC_OWN = 0x80000000u;  // a pattern ends exactly at this node
// a node whose subtree is the rest of ONE pattern (no branch, no other pattern end on the way, at most 8
// bytes): its record is {bytes 0..3, GREC_TAIL | n << 24, the pattern, bytes 4..7} -- compared, not walked
constexpr uint32_t GREC_TAIL = 0x40000000u;

struct Automaton {
    int match_kind = 0;
    uint64_t n_patterns = 0;
    uint32_t min_len = 0, max_len = 0;
    uint32_t n_classes = 1, stride = 1, stride2 = 0;
    uint32_t n_states = 1;
    uint8_t classes[256];
    bool dense = true;                 // the dense table exists (n_states * stride * 4 <= ACX_DENSE_LIMIT, 256 MiB)
    std::vector<uint32_t> table;       // n_states * stride (empty when !dense)
    // compressed form (always): trie edges + failure links
    std::vector<uint32_t> first_child; // n_states + 1: children of s = BFS ids [first_child[s], first_child[s + 1])
    std::vector<uint8_t> in_byte;      // n_states: the byte on the edge INTO the state (children sorted by it)
    std::vector<uint32_t> fail;        // n_states: failure link
    std::vector<uint8_t> sflags;       // n_states: bit 1 = reports something (OUT), bit 0 = ends a pattern itself (OWN)
    std::vector<uint32_t> root_next;   // 256: the root's child for every byte, or 0
    std::vector<uint32_t> own_off;     // n_states + 1
    std::vector<uint32_t> own_pid;     // patterns ending exactly at the state, id order
    std::vector<uint32_t> own1;        // n_states: single own pattern / OWN1_NONE / OWN1_MANY
    std::vector<uint32_t> dlink;       // nearest proper suffix state with own patterns, or NONE
    std::vector<uint32_t> level_start; // max_len + 2 entries
    std::vector<uint32_t> plen;        // n_patterns
    std::vector<uint32_t> rank;        // n_patterns: rank in (len desc, pid asc)
    // prefilter
    uint32_t filter_q = 0;             // level-1 prefix length Q (3..5; 1..2 only with ACX_NO_SHORT_SPLIT), 0 = no patterns
    uint32_t filter_q2 = 0;            // level-2 prefix length Q2 (3..8)
    uint32_t max_shift = 0;            // largest anchor offset in use (0: every pattern is filed under its beginning)
    std::vector<uint8_t> shift;        // n_patterns: the anchor offset d of every pattern (automaton.hpp, anchors)
    std::vector<uint32_t> phead;       // n_patterns x 4 (empty when max_shift == 0): {the pattern's first 12 bytes, 0}
    uint32_t long_min_len = 0;         // shortest LONG pattern (what Q / Q2 are taken from); no long pattern: 5
    uint32_t n_short = 0;              // patterns of at most SHORT_MAX_LEN bytes: K1b's side test (0: none, the
                                       // tables below are empty)
    uint32_t short_min_len = 0;        // the shortest of them (1 or 2)
    std::vector<uint32_t> short_xy;    // SHORT_XY_WORDS: {X, Y} by middle byte
    std::vector<uint32_t> short_codes; // SHORT_CODES
    std::vector<uint32_t> filterA;     // FILTER_WORDS: interleaved {X, Y}
    double filter_density = 0.0;       // fraction of X bits set
    // prefix table (K1b level 2): open addressing, 2^ptab_log2 entries of 4 u32:
    //   {key lo, key hi, K | next << 4 | filter of displaced keys << 8 (0xFFFFFFFF = empty),
    //    the only pattern with this key, or 0x80000000 | index into blist}
    // keys have variable length K: the first min(8, shortest pattern of the group) bytes of the
    // patterns of a group (= the patterns sharing their first Q2 bytes); a group's single key is
    // filed under the hash of those Q2 bytes, several keys behind a redirect entry (automaton.cpp)
    std::vector<uint32_t> blist;       // {count, pid, pid, ...} per key shared by several patterns
    std::vector<uint32_t> rbloom;      // REDIRECT_BLOOM_WORDS: Bloom filter of the keys behind redirect entries
    uint32_t n_prefix_keys = 0;        // entries in use
    // per pattern, 4 u32: {rank | min(len, 255) << 24, the 12 bytes that follow its first Q2 bytes}
    // -- everything the walk kernel needs to settle a short candidate with ONE 16-byte load
    std::vector<uint32_t> pinfo;
    std::vector<uint32_t> ptab;
    uint32_t ptab_log2 = 0;
    std::vector<uint32_t> pbits;       // 2^(ptab_log2 + 3) bits: prefix_bitmap_bit of every group's first Q2 bytes
    // K1a's failureless walk (n_classes <= 32, else empty): see build_walk_tables()
    std::vector<uint32_t> walk_t3b;    // K1A_T3B_WORDS: entry of the symbols (s0, s1, s2) = word ((s0 << 5 | s1) * 33 + s2)
    std::vector<uint32_t> walk_t3r;    // n_classes^3 x {children bitmap, first child | T3R_SHORT}, stride n_classes
    std::vector<uint32_t> walk_grec;   // n_states x 4: trie record or tail record
    // pattern bytes (kept for the synthetic text generator)
    std::vector<uint8_t> blob;
    std::vector<uint64_t> offsets;
};

// Returns empty string on success, otherwise an error message; `code` receives
// an ACX_E* value.
// `dense_limit`: the dense transition table is kept when it is at most this many bytes
// (0: ACX_DENSE_LIMIT, default 256 MiB); the compressed form is always built.
std::string compile(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                    int match_kind, Automaton &out, int &code, uint64_t dense_limit = 0);
void build_walk_tables(Automaton &A); // (called by compile)

} // namespace acx
