// device_types.hpp -- structs passed by value to the HIP kernels.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace acx {

// Device-resident automaton (all pointers are device pointers).
struct DevAutomaton {
    const uint32_t *table;       // n_states << stride2 entries: id | FLAG_OUT | FLAG_OWN; NULL: the automaton
                                 // is kept in its compressed form only (trie edges + failure links below)
    const uint32_t *first_child; // n_states + 1: children of s = BFS ids [first_child[s], first_child[s + 1])
    const uint8_t *in_byte;      // n_states: byte on the edge into the state (children ascending)
    const uint32_t *fail;        // n_states: failure link
    const uint8_t *sflags;       // n_states: bit 1 = OUT, bit 0 = OWN
    const uint32_t *root_next;   // 256: the root's child per byte, or 0
    const uint16_t *hot16;       // hot_rows << stride2 entries, compact copy for LDS
    const uint8_t *classes;      // 256
    const uint32_t *own_off;     // n_states + 1
    const uint32_t *own_pid;
    const uint32_t *own1;        // n_states: the state's only own pattern, OWN1_NONE or OWN1_MANY
    const uint32_t *dlink;       // n_states
    const uint32_t *level_start; // max_len + 2
    const uint32_t *plen;        // n_patterns
    const uint32_t *pchars;      // n_patterns: UTF-8 lead (non-continuation) bytes of the pattern = its code points
    const uint32_t *rank;        // n_patterns
    const uint32_t *by_rank;     // n_patterns: the pattern of a rank (inverse of rank)
    const uint32_t *filterA;     // FILTER_WORDS: level-1 {X, Y} table of the K1b prefilter
    const uint32_t *ptab;        // prefix table: 4 u32 per entry (gram lo, hi, state|flags, pid or list)
    const uint32_t *blist;       // candidate lists of prefixes shared by several patterns
    const uint32_t *rbloom;      // REDIRECT_BLOOM_WORDS: Bloom filter of the keys behind redirect entries
    const uint32_t *pbits;       // 2^(ptab_log2 + 3) bits: the groups' first Q2 bytes (prefix_bitmap_bit)
    const uint4 *pinfo;          // per pattern {rank | min(len,255) << 24, 12 bytes after the first Q2}
    const uint8_t *pat_blob;     // pattern bytes (generator only)
    const uint64_t *pat_off;     // n_patterns + 1
    uint64_t n_patterns;
    uint32_t n_states;
    uint32_t stride2;
    uint32_t min_len, max_len;
    uint32_t hot_rows;           // rows present in hot16
    uint32_t filter_q;           // level-1 prefix length, 0 (no patterns) .. 5
    uint32_t filter_q2;          // level-2/3 prefix length, .. 8
    uint32_t rank_bits;          // bits of the tie-break field of an occurrence key
    uint32_t ptab_log2;
    uint32_t filter_big;         // the level-1 table is saturated: K1b puts every position to both tests
    // K1b, short patterns (automaton.hpp): the side test's pair table and exact codes; short_min_len = 0: the
    // set has none.  k1b_min_len: the shortest pattern the prefilter tables were built from
    const uint32_t *short_xy;    // SHORT_XY_WORDS (copied to LDS)
    const uint32_t *short_codes; // SHORT_CODES
    uint32_t short_min_len, k1b_min_len;
    // K1b, anchors (automaton.hpp): a prefix-table code is pattern id | shift << 24 -- the pattern may START
    // shift bytes in front of the hit; phead = every pattern's first 12 bytes (null when max_shift == 0)
    const uint4 *phead;
    uint32_t max_shift;
    // K1a, automata of at most 65 535 states: the whole DFA as u16, rows of n_classes entries
    // (no padding: 63 277 states x 28 classes x 2 B = 3.4 MiB fits one XCD's 4 MiB L2), in an
    // order of its own: states that report nothing first (BFS order), the reporting ones after
    // them -- "target >= walk_plain" is the whole output test.  The first rows are K1a's LDS tile.
    const uint16_t *table16;     // n_states * n_classes (null: the automaton is too large)
    const uint32_t *walk_bfs;    // walk order id -> BFS id (the emit path works in BFS ids)
    uint32_t n_classes;
    uint32_t walk_plain;         // states that report nothing
    // K1a, failureless form (automata of at most 32 byte classes): every haystack position walks the
    // trie -- the goto function only, no failure links: an occurrence is found from its own start.  The
    // first three levels are ONE LDS lookup by the class triple ((c0 * n_classes + c1) * n_classes + c2:
    // at most 32^3 entries, 128 KiB), the levels below are records in HBM (L2-resident).  null: the
    // automaton has more classes.
    const uint32_t *t3b;         // level 1 of the scan, K1A_T3B_WORDS words (LDS): by the SYMBOLS (low five bits) of
                                 // three bytes, the symbols a fourth byte can have on a trie path of depth 4 (bit =
                                 // symbol); ~0: a pattern of <= 3 bytes ends on the path.  A superset test (bytes
                                 // that share their low five bits alias); the records below are exact
    const uint2 *t3r;            // {children bitmap, first child | T3R_SHORT} by class triple: the depth-3 node's record
    const uint4 *grec;           // n_states x {children bitmap, first child | GREC_OWN, own1, 0}: trie records
                                 // (or a GREC_TAIL record: automaton.hpp)
};
// (K1A_T3B_WORDS, T3R_SHORT, GREC_OWN, GREC_TAIL: automaton.hpp)

// How the byte stream is cut into haystacks.
struct Segments {
    const uint64_t *offsets; // device, n_hay + 1 (ragged) or null
    uint64_t n_hay;          // >= 1
    uint64_t uniform_len;    // > 0: every haystack has this length
};

// Output of the scan kernels.
//   occurrence record  {key lo, key hi, pid, pattern length}                    (16 B)
//   prefix-hit record  two quads: {pos lo, pos hi, code, aux} {16 haystack bytes at pos}
// Occurrence keys:
//   key_mode 0 (Standard / overlapping): key = end   << rank_bits | rank(pid)
//   key_mode 1 (LeftmostFirst):          key = start << rank_bits | pid
//   key_mode 2 (LeftmostLongest):        key = start << rank_bits | rank(pid)
// (rank_bits = bits needed for n_patterns - 1.)  Sorting by key ascending yields
// exactly the order each match kind consumes.
//
// Hit-slot mode (sparse output, the default): the stream is cut into 4 KiB *tiles* of the
// 16-byte-aligned index space (index = stream position + lead, lead = address & 15).  Every
// tile owns HIT_SLOTS hit records and a count.  K1b: the wave that scans a tile is the only
// producer of its slots (no atomics, the count is a plain store).  K1a: verified occurrences
// take arrival ranks with one global atomic on the tile's count (an address shared by the
// handful of occurrences of one 4 KiB stretch only).  More hits than slots: *abort_flag, the
// host redoes the call in region mode.
// Region mode (dense output): no contended global atomic either (one HBM word saturates at
// ~88 atomics/us on MI355X): every wave (K1b) / workgroup (walk kernels) owns the region
// [i * region_cap, (i + 1) * region_cap) and keeps its cursor in a register / in LDS; its final
// count goes to block_counts[i] (it keeps counting past region_cap so that the host can size a
// retry exactly).
constexpr uint32_t TILE_BITS = 12;    // tile = 4 KiB of index space
constexpr uint32_t HIT_SLOTS = 64;    // hit records per tile (HBM is 288 GB: half a byte of workspace per haystack byte)
#ifndef ACX_GROUP_TILES
#define ACX_GROUP_TILES 64
#endif
constexpr uint32_t GROUP_TILES = ACX_GROUP_TILES;  // tiles per workgroup of k_tile_main (64: 256 KiB)
#ifndef ACX_GROUP_MAX
#define ACX_GROUP_MAX 1024
#endif
constexpr uint32_t GROUP_MAX = ACX_GROUP_MAX;  // reported matches per group
// The WIDE form of the sparse path's post stage (round 6): stages of STAGE_SLOTS_WIDE occurrences per 4 KiB bucket and stretches
// of GROUP_MAX_WIDE matches per group -- inputs with a match every 100 - 500 bytes stay on the sparse kernels (until round 5 a
// group of more than GROUP_MAX matches, or a bucket of more than 24 occurrences, left it: from one match per 256 bytes on the
// whole call went to the dense path, 2.3x slower).  A context takes the wide form after a call whose groups mostly gave up.
constexpr uint32_t GROUP_MAX_WIDE = 4 * GROUP_MAX;
constexpr uint32_t MAX_LOOKBACK = 4;  // context tiles k_tile_main can stage in front of a group
// Where a tile's hit count lives: the counts of the tiles ONE K1b wave scans (tile, tile + nw,
// tile + 2 nw, ...) are contiguous, so that the wave writes them 16 at a time with one store
// instead of one store per iteration (nw = waves of the launch, iters = tiles per wave; K1a,
// whose counts are atomic arrival counters: nw = 1, the identity).
__host__ __device__ inline uint64_t hcnt_index(uint64_t tile, uint32_t nw, uint32_t iters) {
    return nw <= 1 ? tile : (tile % nw) * iters + tile / nw;
}
// Sparse path, per-call control words: ONE 64-byte block per call, two blocks used by the calls in turn (k_tile_write
// leaves the other one clear for the next call).  Sink::abort_flag / the kernels' abort_flag parameter point at the
// block's first word; the others are reached from there (u32 indexes):
//   [CTL_ABORT]     the call cannot be finished on the sparse path at all (K1a's slots overflowed, ...)
//   [CTL_OVF_LOST]  an overflow list was too small: hits were lost (the host grows the lists and repeats the call)
//   [CTL_HOT_COUNT] groups k_tile_main left to the HOT pipeline (a staged tile with overflow hits, a full bucket,
//                   more than GROUP_MAX matches, an uncertifiable chain): their ids are in the hot list
//   [CTL_OVF_CAP]   records ONE overflow list holds;  [CTL_OVF_RECS] (u64) the lists, back to back;  [CTL_HOT_LIST]
//                   (u64) n_groups ids;  [CTL_OVF_COUNTS] (u64) the lists' fill counters
// K1b files prefix hits beyond a tile's HIT_SLOTS in OVF_LISTS overflow lists, by tile index (nothing is lost).  Not ONE
// list: a counter takes ~16 returning atomics per microsecond whoever asks (measured, round 5: a haystack that is dense
// everywhere -- 400 000 pushes -- kept the scan busy for 24 ms on one counter, and 1 % of hot groups doubled its time),
// so the counters are OVF_LISTS words on cache lines of their own; the tiles of a dense stretch are consecutive and
// spread over all of them.
// One dense region no longer costs the whole call the dense path: the sparse kernels finish every other group, the
// hot groups (+ their context tiles) go through the tile-ordered dense machinery, k_tile_write splices both by the
// groups' counts (reference behaviour: the cost per byte does not depend on where the matches are, src/lib.rs:59).
constexpr uint32_t CTL_ABORT = 0, CTL_OVF_LOST = 1, CTL_HOT_COUNT = 2, CTL_OVF_CAP = 4, CTL_OVF_RECS = 6, CTL_HOT_LIST = 8, CTL_OVF_COUNTS = 10;
constexpr uint32_t OVF_LISTS = 256, OVF_COUNT_STRIDE = 16; // lists; u32 words from one list's counter to the next (64 bytes)
constexpr uint32_t CTL_WORDS = 16;         // u32 words per block
constexpr uint32_t HOT_BIT = 0x80000000u;  // TileSpace::btot[g]: the group is the hot pipeline's (low bits: its matches)
struct Sink {
    uint4 *recs;            // region mode: region_cap * quads uint4 per region
    uint64_t *block_counts; // region mode: one per region
    uint64_t region_cap;    // records per region
    int key_mode;
    uint4 *hslots;          // hit-slot mode: n_tiles * HIT_SLOTS records of two quads
    uint32_t *hcnt;         // hit-slot mode: hits of every tile, at hcnt_index(tile, cnt_nw, cnt_iters)
    uint32_t *abort_flag;   // hit-slot mode: the call's control block (above); [0] set when the slots cannot hold the output
    uint32_t lead;          // index = stream position + lead
    uint32_t cnt_nw, cnt_iters;
};

// Dense outputs, tile-ordered (round 4; kernels.hip: k_dense_verify / k_dense_main): occurrences are filed by
// the 4 KiB tile of their KEY position -- DT_SLOTS words per tile (one occurrence per 8 bytes: denser inputs
// take the radix-sort path) -- and a workgroup orders and resolves DT_GROUP tiles in LDS.
constexpr uint32_t DT_SLOTS = 512;
#ifndef ACX_DT_GROUP
#define ACX_DT_GROUP 4
#endif
constexpr uint32_t DT_GROUP = ACX_DT_GROUP;
constexpr uint32_t DT_GMAX = DT_GROUP * DT_SLOTS; // reported occurrences per group
constexpr uint32_t HOT_SUB = GROUP_TILES / DT_GROUP; // dense groups of one (hot) group of the sparse path
static_assert(GROUP_TILES % DT_GROUP == 0, "a hot group is a whole number of dense groups");
struct DenseTiles {
    uint64_t *words;   // (n_tiles + 1) * DT_SLOTS: [rel : 12 | tie : rank_bits | length], rel = key index & 4095
    uint32_t *counts;  // n_tiles + 1 (atomic arrival counters; cleared before every call)
    uint32_t n_tiles;  // key tiles: tiles of the stream + 1 (an occurrence may END at the very end)
};

// Storage of the sparse path, sized by the number of tiles / groups of the stream.
struct TileSpace {
    uint4 *hslots;      // n_tiles * HIT_SLOTS * 2
    uint32_t *hcnt;     // n_tiles (+ slack), indexed by hcnt_index(tile, cnt_nw, cnt_iters)
    uint32_t cnt_nw, cnt_iters;
    uint4 *trecs;       // groups * gmax: the REPORTED occurrences of a group, in order
    uint32_t gmax;      // records a group's stretch of trecs holds: GROUP_MAX, or GROUP_MAX_WIDE (the wide form of k_tile_main)
    uint32_t w8;        // trecs holds ONE 64-bit word per reported occurrence -- [key position : 44 | tie : rank_bits | length : 20 -
                        // rank_bits] -- not a 16-byte record: the narrow-word form of k_tile_main (tile_words_narrow, kernels.hpp)
    uint32_t *btot;     // reported occurrences of each group
    uint64_t *sgw;      // 2 sets (used by the calls in turn) of 2 * sg_cap words: per supergroup of 64
                        // groups, the sum of their counts / of their statistics (kernels.hip)
    uint32_t sg_cap;
    uint32_t n_tiles, n_groups;
};

} // namespace acx
