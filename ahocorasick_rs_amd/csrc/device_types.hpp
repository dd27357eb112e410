// device_types.hpp -- structs passed by value to the HIP kernels.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace acx {

// Device-resident automaton (all pointers are device pointers).
struct DevAutomaton {
    const uint32_t *table;       // n_states << stride2 entries: id | FLAG_OUT | FLAG_OWN
    const uint16_t *hot16;       // hot_rows << stride2 entries, compact copy for LDS
    const uint8_t *classes;      // 256
    const uint32_t *own_off;     // n_states + 1
    const uint32_t *own_pid;
    const uint32_t *own1;        // n_states: the state's only own pattern, OWN1_NONE or OWN1_MANY
    const uint32_t *dlink;       // n_states
    const uint32_t *level_start; // max_len + 2
    const uint32_t *plen;        // n_patterns
    const uint32_t *rank;        // n_patterns
    const uint32_t *filterA;     // FILTER_WORDS: level-1 {X, Y} table of the K1b prefilter
    const uint32_t *ptab;        // prefix table: 4 u32 per entry (gram lo, hi, state|flags, pid or list)
    const uint32_t *blist;       // candidate lists of prefixes shared by several patterns
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
};

// How the byte stream is cut into haystacks.
struct Segments {
    const uint64_t *offsets; // device, n_hay + 1 (ragged) or null
    uint64_t n_hay;          // >= 1
    uint64_t uniform_len;    // > 0: every haystack has this length
};

// Sink of the scan kernels: 16-byte records in per-workgroup regions.
//   occurrence record  {key lo, key hi, pid, pattern length}
//   prefix-hit record  two quads: {pos lo, pos hi, code, 0} {16 haystack bytes at pos}
// Occurrence keys:
//   key_mode 0 (Standard / overlapping): key = end   << rank_bits | rank(pid)
//   key_mode 1 (LeftmostFirst):          key = start << rank_bits | pid
//   key_mode 2 (LeftmostLongest):        key = start << rank_bits | rank(pid)
// (rank_bits = bits needed for n_patterns - 1.)  Sorting by key ascending yields
// exactly the order each match kind consumes.
// Region mode (dense output, prefix hits): slot allocation uses NO contended global
// atomic (one HBM word saturates at ~88 atomics/us on MI355X): every workgroup owns the
// region [blockIdx * region_cap, (blockIdx + 1) * region_cap) and hands out slots from
// a counter in LDS; its final count goes to block_counts[blockIdx] (it keeps
// counting past region_cap so that the host can size a retry exactly).
// Slot mode (sparse output, slots != null): an occurrence goes straight to slot
// `arrival rank` of its 4 KiB-of-position bucket (bucket_cnt is the rank counter: an
// address is shared by the few occurrences of one bucket only); a full bucket sets
// *abort_flag and the host redoes the call in region mode.
constexpr uint32_t BUCKET_SLOTS = 32;  // occurrence slots per bucket
constexpr uint32_t BUCKET_BITS = 12;   // bucket = 4 KiB of stream position
constexpr uint32_t TILE_BUCKETS = 64;  // buckets per workgroup of the tile kernels (K2b)
constexpr uint32_t TILE_MAX = 1024;    // occurrences per tile (held in LDS)
struct Sink {
    uint4 *recs;            // region_cap * quads uint4 per region
    uint32_t *bucket_cnt;   // slot mode: per-bucket arrival counters
    uint64_t *block_counts; // gridDim.x entries
    uint64_t region_cap;    // records per region
    uint32_t bucket_shift;  // bucket = key >> bucket_shift
    int key_mode;
    uint4 *slots;           // slot mode: n_buckets * BUCKET_SLOTS records {key lo, key hi, pid, len}
    uint32_t *abort_flag;   // slot mode: set when the sparse path cannot hold the output
};

// Storage of the sparse path, sized by the number of buckets / tiles of the stream.
struct TileSpace {
    uint4 *slots;       // n_buckets * BUCKET_SLOTS (filled by the scan's emission)
    uint32_t *bcnt;     // n_buckets + 1 arrival counters
    uint4 *trecs;       // tiles * TILE_MAX: the occurrences of a tile, sorted {key lo, key hi, pid, len}
    uint8_t *syncf;     // tiles * TILE_MAX: occurrence is a sync point of the greedy
    uint8_t *accf;      // tiles * TILE_MAX: occurrence is reported
    uint32_t *tile_n;   // occurrences of each tile
    uint32_t *btot;     // reported occurrences of each tile
    uint32_t *bbase;    // exclusive prefix of btot
    uint32_t n_buckets, n_tiles;
};

} // namespace acx
