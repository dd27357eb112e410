// kernels.hpp -- launch wrappers of the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/acx.h"
#include "device_types.hpp"

namespace acx {

// ---- K1: scan kernels
// K1a: chunked DFA walk, class map + hot rows in LDS.  Emits verified occurrences: into the hit
// slots of the tile their start lies in (K.hslots != null; K.hcnt must be zero beforehand) or
// into per-workgroup regions (dense path).
uint32_t dfa_walk_grid(const DevAutomaton &A, uint64_t len, int n_cus);
// Ad: device-resident copy of A (read by the cold, out-of-line emit paths)
hipError_t launch_dfa_walk(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G,
                           const Sink &K, const uint8_t *d_hay, uint64_t len, uint32_t grid,
                           size_t max_lds, hipStream_t st);
// K1a, failureless form (automata of at most 32 byte classes; kernels.hip): k1a_scan settles the first
// four levels of the walk of every position in LDS, takes the survivors one level on in a software
// pipeline of gathers and leaves the walks that go on as 32-byte items in per-wave regions; k1a_walk
// finishes them and emits the occurrences into K.  Hit slots (K.hslots != null, walk_grid arbitrary): the
// scan settles the occurrences below tail nodes itself and writes every tile's count (layout: K.cnt_nw
// = scan_grid * 16 waves), the walk appends with atomics.  Regions (dense output: walk_grid = scan_grid *
// 16 = the number of occurrence regions, one per block of the walk): everything goes through the walk.
// work: pfac_workspace_words() u64 words; counts: scan_grid * 16 u64 words.  More items than the
// regions hold: *K.abort_flag = 1 (both modes).
bool pfac_available(const DevAutomaton &A, size_t max_lds);
uint32_t pfac_scan_grid(const uint8_t *d_hay, uint64_t len, int n_cus);
uint64_t pfac_workspace_words(uint64_t len, uint32_t scan_grid, bool dense);
hipError_t launch_pfac(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G, const Sink &K,
                       const uint8_t *d_hay, uint64_t len, uint32_t scan_grid, uint64_t *work, uint64_t *counts,
                       uint32_t walk_grid, bool dense, hipStream_t st);
// K1b: LDS q-gram prefilter + exact prefix table.  Emits prefix hits: into the hit slots of
// their tile (K.hslots != null; every tile's count is written) or into per-wave regions
// (prefilter_hit_regions(grid) of them, dense path).
uint32_t prefilter_grid(const uint8_t *d_hay, uint64_t len, int n_cus);
uint64_t prefilter_tiles(const uint8_t *d_hay, uint64_t len); // 4 KiB tiles of the aligned index space
uint32_t prefilter_hit_regions(uint32_t grid);
// cp_sub != null (sparse mode, 16-byte aligned d_hay only): the scan also writes the lead-byte
// counts of every 16 bytes it streams (64 per 1 KiB block; 4 KiB-tile granularity: the array
// holds 256 * prefilter_tiles() bytes) -- block_totals() then replaces count_lead_bytes()
hipError_t launch_prefilter(const DevAutomaton &A, const Sink &K, const uint8_t *d_hay, uint64_t len,
                            uint32_t grid, hipStream_t st, hipEvent_t ev_start = nullptr,
                            hipEvent_t ev_stop = nullptr, uint8_t *cp_sub = nullptr);
// sink bookkeeping (dense path): summary[0] = total kept, summary[1] = max count of a region
hipError_t sink_summary(const uint64_t *block_counts, uint32_t grid, uint64_t region_cap,
                        const uint64_t *hit_counts, uint32_t hit_grid, uint64_t hit_cap,
                        uint64_t *summary, uint64_t *offsets, hipStream_t st);
hipError_t sink_compact(const uint4 *recs, const uint64_t *offsets, uint32_t grid,
                        uint64_t region_cap, uint64_t *keys_out, uint32_t *pids_out, hipStream_t st);
// dense path: K1b's prefix hits (per-wave regions, `hit_grid` of them) -> verified occurrences in
// the occurrence sink (`occ_grid` = walk_hits_grid(hit_grid) regions, one per workgroup)
uint32_t walk_hits_grid(uint32_t hit_regions);
hipError_t launch_walk_hits(const DevAutomaton &A, const Segments &G, const Sink &hits, uint32_t hit_grid,
                            const Sink &occ, uint32_t occ_grid, const uint8_t *d_hay, uint64_t len,
                            hipStream_t st);
size_t prefilter_lds_bytes(); // static LDS of K1b
// rows of the hot16 table K1a can stage for this automaton and LDS size
uint32_t dfa_walk_hot_rows(uint32_t n_states, uint32_t stride2, size_t max_lds);

// ---- post-processing
size_t sort_temp_bytes(uint64_t n);
hipError_t sort_occurrences(void *temp, size_t temp_bytes, const uint64_t *keys_in,
                            uint64_t *keys_out, const uint32_t *pids_in, uint32_t *pids_out,
                            uint64_t n, int end_bit, hipStream_t st);
// K0: a whole call in one workgroup (haystacks of at most SMALL_MAX_LEN bytes, at most
// SMALL_MAX_OCC occurrences, one haystack).  hay / out / res may be pinned host memory.
// out: SMALL_MAX_OCC records; res[0] = matches written, res[1] != 0: too dense, nothing written.
constexpr uint32_t SMALL_MAX_LEN = 16384, SMALL_MAX_OCC = 1024;
// ... up to SMALL_PF_MAX_LEN bytes for automata K0's prefilter mode takes (small_prefilter_ok: K1b's tables exist, no pattern
// of 1 or 2 bytes): one workgroup, one launch, whatever the automaton's size
constexpr uint32_t SMALL_PF_MAX_LEN = 65536;
bool small_prefilter_ok(const DevAutomaton &A);
// seq != 0 (the host polls): res is the call's RESULT LINE -- 64 aligned bytes of coherent pinned host memory, written by one
// store instruction: [0] seq, [1] matches | too dense << 32 | hash of out[] << 33, [2 .. 6] the first K0_LINE_MATCHES matches packed as
// pattern | start << 32 | (end - 1) << 48, [7] seq ^ k0_line_check(words 1 .. 6) -- and out[] takes the matches beyond those, packed the same way, first.
// seq == 0: out[] = acx_match_t records, res[0] / res[1] as above (device memory, read behind a stream synchronisation)
#define ACX_K0_LINE_MATCHES 5
constexpr uint32_t K0_LINE_WORDS = 8;
// Round 6: up to K0_RESULT_LINES lines, one store instruction (sixteen lanes): the lines behind the first are [0] seq,
// [1 .. 6] the next six matches, [7] seq ^ k0_line_check(words 1 .. 6) each -- 23 matches travel without a second write to
// host memory and the release between the two (a sixth match cost a call ~7 us); out[] takes the matches beyond those.
constexpr uint32_t K0_RESULT_LINES = 4, K0_MORE_MATCHES = 6, K0_LINES_MATCHES = ACX_K0_LINE_MATCHES + (K0_RESULT_LINES - 1) * K0_MORE_MATCHES;
__host__ __device__ inline uint32_t k0_result_lines(uint64_t matches) { // lines that carry the first min(matches, K0_LINES_MATCHES) matches
    const uint64_t m = matches < K0_LINES_MATCHES ? matches : K0_LINES_MATCHES;
    return m <= ACX_K0_LINE_MATCHES ? 1u : 1u + (uint32_t)((m - ACX_K0_LINE_MATCHES + K0_MORE_MATCHES - 1) / K0_MORE_MATCHES);
}
// the line's last word = seq ^ k0_line_check(words 1 .. 6): the host takes the line when word 0 carries the call's number AND
// the last word agrees with the six in the middle as it read them -- whatever order the line's four 16-byte pieces arrive in,
// a line with a stale or half-written middle is not accepted (it is polled again)
// The matches beyond the line's (packed, in out[]) are covered too: word 1 of the line carries, above the count, a hash of
// them (k0_rest_mix per entry, XORed): the host reads out[] behind the line and takes it when the hash agrees -- the line and
// out[] are separate writes to host memory, and nothing orders their arrival (found in round 5 by a stress run of eight threads
// on one handle: 1 call in ~30 000 read the PREVIOUS call's entries in out[]; until then consecutive calls with more than five
// matches happened to carry the same ones in every test).
__host__ __device__ inline uint32_t k0_rest_mix(uint64_t entry, uint32_t k, uint64_t seq) {
    uint64_t x = entry ^ ((uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull) ^ (seq * 0xD6E8FEB86659FD93ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return (uint32_t)x & 0x7FFFFFFFu;
}
constexpr uint32_t K0_REST_HASH_SHIFT = 33; // word 1: matches (32 bits) | too dense (bit 32) | hash of out[] (31 bits)
__host__ __device__ inline uint64_t k0_line_check(const uint64_t *mid /* words 1 .. 6 */) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 6; i++) {
        h ^= mid[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
    }
    return h ^ (h >> 32);
}
hipError_t launch_small(const DevAutomaton &A, const uint8_t *hay, uint32_t len, int key_mode, bool overlapping,
                        bool codepoints, acx_match_t *out, uint64_t *res, uint64_t seq, hipStream_t st, bool direct_ok = true);
// (direct_ok = false: not MODE 2 -- it compares against the patterns themselves, every copy of a string, whatever view of the
// tables A is: an overlapping search over a set with copies expects one occurrence per string, acx_api.cpp expand_copies)
// how K0 finds the occurrences of a call of `len` bytes (kernels.hip: 0 the walk, 1 the tables in LDS, 2 direct comparison,
// 3 the prefilter); -1: K0 does not take the call
int small_mode(const DevAutomaton &A, uint32_t len, bool direct_ok);
// The RESIDENT K0 (kernels.hip, k0_resident; A: the automaton's description in HBM): launched once per context in one `mode`
// and one `overlapping`, fed through a mailbox in coherent pinned host memory: [0] the word below, [1] the check of the
// haystack's first bytes (k0_hay_check), the haystack K0_MAILBOX_HAY bytes behind the mailbox's start, zero-padded to
// 16 bytes.  The host writes the haystack, the check, then the word (one aligned store).  seq: the number of the last call
// before the launch (the first call the kernel takes carries seq + 1 in its word); *status = epoch when the kernel has
// left (told to, idle, or at the end of its life: ticks of the 100 MHz clock).
constexpr uint64_t K0_MAILBOX_QUIT = 1ull << 31, K0_MAILBOX_CP = 1ull << 30, K0_MAILBOX_LEN_MASK = (1ull << 30) - 1;
constexpr uint32_t K0_MAILBOX_HAY = 16;
constexpr uint32_t K0_MAILBOX_INLINE = 1008; // bytes of the haystack that travel with the poll (63 lanes x 16 bytes)
inline uint64_t k0_mailbox_word(uint64_t seq, uint32_t len, bool codepoints, bool quit) {
    return (seq << 32) | (quit ? K0_MAILBOX_QUIT : 0) | (codepoints ? K0_MAILBOX_CP : 0) | len;
}
// the check: XOR over the 16-byte pieces j of the first min(len rounded up to 16, K0_MAILBOX_INLINE) bytes
__host__ __device__ inline uint64_t k0_hay_mix(uint64_t lo, uint64_t hi, uint32_t j, uint64_t key) {
    uint64_t x = (lo ^ key) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 31;
    x = (x ^ hi ^ ((uint64_t)(j + 1) * 0xC2B2AE3D27D4EB4Full)) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    return x;
}
inline uint64_t k0_hay_check(const uint8_t *hay, uint32_t len, uint64_t seq, uint64_t secret) { // (the last piece zero-padded)
    const uint32_t covered = len < K0_MAILBOX_INLINE ? (len + 15) & ~15u : K0_MAILBOX_INLINE;
    uint64_t h = 0;
    for (uint32_t j = 0; 16 * j < covered; j++) {
        uint64_t v[2] = {0, 0};
        __builtin_memcpy(v, hay + 16 * j, 16 * j + 16 <= len ? 16 : len - 16 * j);
        h ^= k0_hay_mix(v[0], v[1], j, seq ^ secret);
    }
    return h;
}
hipError_t launch_resident(const DevAutomaton *A, int mode, const uint64_t *mailbox, int key_mode, bool overlapping,
                           acx_match_t *out, uint64_t *res, uint64_t *status, uint64_t epoch, uint64_t seq,
                           uint64_t idle_ticks, uint64_t life_ticks, uint64_t secret, uint32_t delay, hipStream_t st);
// sparse path: k_tile_main (verify the hits, order, match kind) -> k_tile_write
// (final records in out[], capacity n_groups * GROUP_MAX).  The launch geometry depends on the
// number of tiles only, so no host round trip is needed before them.  The first group of the write
// kernel publishes the call's totals as ONE 64-byte line of pinned host memory, one store instruction -- host_out:
// {[0] seq, [1] matches, [2] occurrences, [3] prefix hits, [4] why (1 aborted, 2 an overflow list was too small) | hot
// groups << 8, [5] overflow hits | the fullest overflow list << 32, [6] 0, [7] seq ^ k0_line_check([1 .. 6])}; the host may
// poll it while the write kernel still runs and takes the line when its check word agrees (round 5; until then eight
// separate words, a fence, then seq) -- and clears *next_flag for the next call.  seq: the context's call counter
// (its parity selects the set of supergroup words, TileSpace::sgw).  Aborted: the output did not fit
// the slots, out[] and the totals are meaningless.  seg_counts != null (batch, byte offsets):
// offsets are made local to the match's haystack (G) and the per-haystack counts are accumulated
// into seg_counts (cleared by k_tile_main).  Automata with tile_lookback(max_len) > MAX_LOOKBACK
// cannot take this path.  cp_blockpre != null (one haystack): the write kernel turns the byte
// offsets into code-point indexes on the way out (prefix of the 1 KiB blocks + the counts of the 16-byte
// chunks + what k_tile_main carried over of the start's own chunk: no haystack access).
// abort_flag / next_flag: the control blocks of this call and of the next one (device_types.hpp).  hot_ok (K1b's prefix
// hits): groups the sparse kernels cannot finish are listed for the HOT pipeline instead of aborting the call; the write
// kernel then -- when its line announces hot groups -- writes nothing:
// the caller runs hot_verify_main + hot_write (below) and waits for the second publication.  hot_counts (may be null): the
// hot pipeline's bucket counters (DenseTiles::counts, hot_tiles of them), cleared by the write kernel when it announces hot groups.
uint32_t tile_lookback(uint32_t max_len);
// the sparse path's post stage runs with narrow staged words and 8-byte reported occurrences for this automaton / index kind
// (the caller sets TileSpace::w8 accordingly before tile_post)
bool tile_words_narrow(const DevAutomaton &A, bool codepoints);
hipError_t tile_post(const DevAutomaton &A, int key_mode, bool overlapping, const TileSpace &T, uint32_t lead,
                     const uint8_t *d_hay, uint64_t len, acx_match_t *out, uint64_t *summary,
                     uint32_t *abort_flag, uint32_t *next_flag, uint64_t *host_out, uint64_t seq,
                     const Segments &G, uint64_t *seg_counts, const uint64_t *cp_blockpre,
                     const uint8_t *cp_sub, hipEvent_t before_write, bool hot_ok, uint32_t *hot_counts, uint32_t hot_tiles,
                     hipStream_t st);
// HOT pipeline: a dense stretch of the input costs the groups it lies in, not the call (reference behaviour: the cost
// per byte does not depend on where the matches are, /root/reference/src/lib.rs:59).
//   hot_verify_main  k_hot_verify: the hits of the hot groups' staged tiles (their slots + the overflow lists of the call's
//                    control block ctl; ovf_max = the fullest list's fill) -> occurrence
//                    words in the buckets of their key tiles (D.counts must be zero); k_dense_main over the hot groups'
//                    dense groups (HOT_SUB each): records in TD.trecs, counts in TD.btot AND credited to the hot group in
//                    S.btot / the supergroup words of the call's set.  *hot_abort != 0: a bucket overflowed or a chain left
//                    its context (the caller redoes the call on the radix-sort form of the dense path)
//   hot_totals       host_out: one 64-byte line (seq, payload, check): [1] = the call's matches (for a caller that sizes the output exactly)
//   hot_write        the hot groups' records, then the sparse path's write kernel again: every group placed with all
//                    counts in; host_out: the totals' line again ([4] = *hot_abort != 0), seq = pub
hipError_t hot_verify_main(const DevAutomaton &A, int key_mode, bool overlapping, const Segments &G, const TileSpace &S,
                           const uint32_t *hot_list, uint32_t n_hot, const uint32_t *ctl, uint32_t ovf_max, const DenseTiles &D,
                           const TileSpace &TD, uint32_t lead, const uint8_t *d_hay, uint64_t len, uint32_t *hot_abort,
                           uint64_t seq, uint32_t spec_bound, hipStream_t st);
hipError_t hot_totals(const TileSpace &S, uint64_t seq, uint64_t *host_out, uint64_t pub, hipStream_t st);
hipError_t hot_write(const DevAutomaton &A, int key_mode, const TileSpace &S, const TileSpace &TD, const uint32_t *hot_list,
                     uint32_t n_hot, uint32_t lead, const uint8_t *d_hay, acx_match_t *out, uint64_t *summary,
                     const uint32_t *abort_flag, const uint32_t *hot_abort, uint64_t *host_out, uint64_t seq, uint64_t pub,
                     const Segments &G, uint64_t *seg_counts, const uint64_t *cp_blockpre, const uint8_t *cp_sub, uint32_t spec_bound,
                     hipStream_t st);
// (spec_bound != 0, both: a SPECULATIVE launch, queued right behind tile_post before the host knows whether the call has hot
// groups -- grids for spec_bound of them, their number read from the control block on the device: kernels.hip, hot_groups_here)
// dense outputs, tile-ordered (kernels.hip): K1b's prefix hits (per-wave regions) -> occurrence words in the
// bucket of their key tile (D.counts must be zero) -> per group of DT_GROUP tiles: sort + match kind in LDS, the
// reported occurrences in T.trecs (DT_GMAX per group), T.btot, the supergroup words of set 0 (must be zero); summary[8]
// = matches, [9] = occurrences; *abort_flag != 0: a bucket overflowed or a chain left its context (the caller
// takes the radix-sort path).  dense_tiles_write: the sparse path's write kernel over T (out: capacity = matches).
hipError_t dense_tiles_verify(const DevAutomaton &A, const Segments &G, const Sink &hits, uint32_t hit_grid,
                              const DenseTiles &D, int key_mode, uint32_t lead, const uint8_t *d_hay, uint64_t len,
                              uint32_t *abort_flag, hipStream_t st);
// (compact: the kernel's narrow-stage form, sixteen groups per CU; the abort flag reads 2 when a group did not fit: again, full)
hipError_t dense_tiles_main(const DevAutomaton &A, int key_mode, bool overlapping, const DenseTiles &D, const TileSpace &T,
                            uint32_t lead, uint32_t *abort_flag, uint64_t *summary, bool compact, hipStream_t st);
hipError_t dense_tiles_write(const DevAutomaton &A, int key_mode, const TileSpace &T, const uint8_t *d_hay, acx_match_t *out,
                             uint64_t *summary, const uint32_t *zero_flag, uint64_t *host_out, uint32_t lead, const Segments &G,
                             uint64_t *seg_counts, const uint64_t *cp_blockpre, const uint8_t *cp_sub, hipStream_t st);
// spans from sorted (key,pid): S[i], E[i]
hipError_t make_spans(const DevAutomaton &A, int key_mode, const uint64_t *keys,
                      const uint32_t *pids, uint64_t *S, uint64_t *E, uint64_t n,
                      hipStream_t st);
size_t scan_temp_bytes(uint64_t n);
// running max of E (inclusive) -> M
hipError_t prefix_max(void *temp, size_t temp_bytes, const uint64_t *E, uint64_t *M,
                      uint64_t n, hipStream_t st);
// non-overlapping greedy: flags[i] = 1 iff occurrence i is reported
hipError_t resolve_greedy(const uint64_t *S, const uint64_t *E, const uint64_t *M,
                          uint32_t *flags, uint64_t n, hipStream_t st);
// exclusive prefix sum of flags -> idx (idx[n] = total)
hipError_t flag_offsets(void *temp, size_t temp_bytes, const uint32_t *flags,
                        uint32_t *idx, uint64_t n, hipStream_t st);
// write final matches; flags == nullptr keeps everything (overlapping)
hipError_t write_matches(const uint32_t *pids, const uint64_t *S, const uint64_t *E,
                         const uint32_t *flags, const uint32_t *idx, acx_match_t *out,
                         uint64_t n, hipStream_t st);

// ---- UTF-8 code-point fix-up (reference: get_byte_to_code_point)
// lead-byte count of every 1 KiB block -> cnt[nblocks + 1] (last = 0), of every 16 bytes of a
// block -> sub[64 * nblocks]
hipError_t count_lead_bytes(const uint8_t *d_hay, uint64_t len, uint64_t *cnt, uint8_t *sub,
                            hipStream_t st);
hipError_t block_totals(const uint8_t *sub, uint64_t *cnt, uint64_t nblocks, hipStream_t st);
hipError_t prefix_sum_u64(void *temp, size_t temp_bytes, const uint64_t *in, uint64_t *out,
                          uint64_t n, hipStream_t st);
// pre[0 .. nblocks] = the exclusive prefix of the blocks' lead-byte counts, in two launches of this file's own kernels
// (haystacks up to 4 GiB; beyond: block_totals + the library's scan).  sub != null: the counts come from the 16-byte
// stretches' counts (and are left in cnt); else cnt[0 .. nblocks) is there (count_lead_bytes).  temp: >= 32 KiB + what
// prefix_sum_u64 wants.
hipError_t block_prefix(const uint8_t *sub, uint64_t *cnt, uint64_t *pre, uint64_t nblocks, void *temp, size_t temp_bytes,
                        hipStream_t st);
hipError_t to_code_points(const uint8_t *d_hay, uint64_t len, const uint64_t *blockpre, const uint8_t *sub,
                          acx_match_t *m, uint64_t n, hipStream_t st);

// ---- a call cut into byte ranges (acx_api.cpp, run_chunked)
// cut_point: m[0 .. n) ordered so that {field + shift < limit} holds for a prefix (field: start, or end when by_end):
// out[0] = the prefix's length, out[1] = end + shift of its last element (device words; cleared here).
// copy_shifted: dst = src with start and end moved by shift.
hipError_t cut_point(const acx_match_t *m, uint64_t n, bool by_end, uint64_t shift, uint64_t limit, uint64_t *out, hipStream_t st);
hipError_t copy_shifted(acx_match_t *dst, const acx_match_t *src, uint64_t n, uint64_t shift, hipStream_t st);
// rebase_offsets: dst[i] = src[i] - base (the second part of a batch that is cut at a haystack boundary)
hipError_t rebase_offsets(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base, hipStream_t st);

// ---- copies of a pattern, overlapping searches (acx_api.cpp, expand_copies): the search reports one occurrence per string
// (lowest id); xcnt[pid] = its later copies, xoff[pid] = where their ids begin in xids[].
// copy_runs: k[i] = 1 + xcnt[m[i].pattern] (k[n] = 0), offs = exclusive prefix of k over n + 1 elements (offs[n] = records);
// expand_copies_write: out[offs[i] ..] = the copies of occurrence i, ids ascending; expand_copies_counts (batch): counts[]
// of the unexpanded result -> of the expanded one (incl: n_hay words of scratch).  temp: scan_temp_bytes(max(n, n_hay) + 1).
hipError_t copy_runs(const acx_match_t *m, uint64_t n, const uint32_t *xcnt, void *temp, size_t temp_bytes, uint64_t *k,
                     uint64_t *offs, hipStream_t st);
hipError_t expand_copies_write(const acx_match_t *m, uint64_t n, const uint64_t *offs, const uint32_t *xoff, const uint32_t *xids,
                               acx_match_t *out, uint64_t total, hipStream_t st);
hipError_t expand_copies_counts(void *temp, size_t temp_bytes, uint64_t *counts, uint64_t n_hay, uint64_t *incl, const uint64_t *offs,
                                hipStream_t st);

// ---- batch: make offsets local to each haystack, count matches per haystack.
// base_cp != nullptr: subtract the code-point index of the haystack start
// (computed from blockpre) instead of the byte offset.
hipError_t localize(const Segments &G, const uint8_t *d_hay, uint64_t len,
                    const uint64_t *blockpre, const uint8_t *sub, int codepoints, acx_match_t *m, uint64_t n,
                    uint64_t *counts, hipStream_t st);

// ---- synthetic haystacks
hipError_t generate(const DevAutomaton &A, uint8_t *dst, uint64_t len, int kind,
                    uint64_t seed, uint64_t stream_offset, hipStream_t st);

} // namespace acx
