/*
 * acx.h -- C ABI of the MI355X-native Aho-Corasick matcher (libacx_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path this project replaces:
 * the match loop behind `find_matches_as_indexes` of G-Research/ahocorasick_rs.
 * Every entry point cites the reference interface it replaces
 * (paths relative to the reference repository).  Plain pointers and sizes
 * only; no torch / Python types.  A Rust/PyO3 host would bind these with an
 * `extern "C"` block; this repository's host shim is the
 * C++ CPython extension ahocorasick_rs_amd/csrc/pymodule.cpp.
 *
 * All functions return ACX_OK (0) or a negative ACX_E* code; the message of
 * the last failure on the calling thread is available from acx_last_error().
 * There is NO CPU fallback: without a usable HIP device every matching call
 * fails with ACX_EDEVICE.
 */
#ifndef ACX_H
#define ACX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 9: acx_path_stats gained [10], [11] (round 6: launches of a context's resident K0; mid-size host haystacks read in place);
 * 8: acx_path_stats gained [9] (round 6: calls repeated with the wide form of the sparse path's post stage);
 * 7: acx_path_stats gained [8] (byte ranges of calls that were cut); K0's result line carries end - 1 and a hash of the
 *    matches beside it (round 5);
 * 6: acx_path_stats added (round 5);
 * 5: acx_host_tables_t grew (short-pattern tables), acx_device_synchronize_on, acx_comm_* added (round 4);
 * 4: acx_host_tables_t grew (walk_t3b / walk_t3r / walk_grec, round 3);
 * 3: acx_replicate / acx_find_batch_multi / acx_shard_range / acx_automaton_device added (round 3);
 * 2: acx_prefix_slot gained `salt`, acx_host_tables_t grew (round 2).  A binding built against another
 * header must refuse to load: compare acx_version() with the ACX_VERSION it was compiled with. */
#define ACX_VERSION 9

/* status codes */
#define ACX_OK 0
#define ACX_EINVAL (-1)      /* bad argument                                  */
#define ACX_EEMPTY (-2)      /* empty pattern (src/lib.rs:204-208, 386-389)   */
#define ACX_EOVERLAP (-3)    /* overlapping search on a non-Standard automaton:
                                the crate's MatchError -> ValueError,
                                src/lib.rs:36-39, 52-54                        */
#define ACX_ENOMEM (-4)
#define ACX_EDEVICE (-5)     /* HIP runtime / no device / kernel failure      */
#define ACX_ETOOBIG (-6)     /* automaton or haystack exceeds an encoding limit: 2^24 patterns, 2^30
                                states, a haystack stream of 2^38 bytes.  A ONE-haystack call that
                                would enumerate 2^32 occurrences or more in one pass (the width of the
                                device's indexes) is cut into byte ranges that are searched one after
                                the other (round 5; acx_path_stats [8]) -- the error remains for a
                                BATCH that does (cut it at a haystack boundary) and for a range of
                                ~2 * max pattern length bytes that still does                     */

/* enum PyMatchKind, src/lib.rs:92-108 */
#define ACX_MATCH_STANDARD 0
#define ACX_MATCH_LEFTMOST_FIRST 1
#define ACX_MATCH_LEFTMOST_LONGEST 2

/* enum Implementation (+ None), src/lib.rs:111-128.  A hint only: every value
 * yields identical results (tests/test_ac.py:23-31) and NO value selects a slower
 * scan kernel (the reference recommends the contiguous NFA as the default trade-off,
 * README.md:173-177).  It decides how large a dense transition table is kept: at most
 * ACX_DENSE_LIMIT bytes (default 256 MiB; DFA: 16 GiB); beyond that the automaton stays
 * in its compressed form (trie edges + failure links) and the walking kernels step
 * that -- the reference's "DFA for small sets, NFA beyond".  The scan kernel is picked
 * from the pattern statistics (acx_info.kernel; override: acx_set_kernel / ACX_KERNEL). */
#define ACX_IMPL_AUTO (-1)
#define ACX_IMPL_NONCONTIGUOUS_NFA 0
#define ACX_IMPL_CONTIGUOUS_NFA 1
#define ACX_IMPL_DFA 2

/* scan kernels (acx_info.kernel, acx_set_kernel).  Haystacks of at most 16 KiB are
 * answered by K0 -- one workgroup does the whole call (the occurrences by direct comparison for a
 * handful of short patterns, else by an anchored walk from every position, over tables staged in LDS
 * when they are small; sort, resolve, output; one launch, the result -- one 64-byte line that carries
 * the call's number at both ends -- polled from pinned memory) -- unless a scan
 * kernel was chosen explicitly with acx_set_kernel / ACX_KERNEL or the output is too dense for it.
 * Since round 4 the library's own choice is the prefilter for EVERY pattern set (patterns of 1 and
 * 2 bytes through its side test); the DFA walk runs when it is asked for. */
#define ACX_KERNEL_AUTO 0
#define ACX_KERNEL_DFA_WALK 1   /* K1a: the DFA walk -- failureless form, first four levels in LDS;
                                 * chunked walk for automata of more than 32 byte classes */
#define ACX_KERNEL_PREFILTER 2  /* K1b: LDS q-gram prefilter + exact prefix keys + verification */

/* One match: the tuple `(u64, usize, usize)` of src/lib.rs:234, 427. */
typedef struct acx_match {
    uint64_t pattern; /* index into the patterns iterable                     */
    uint64_t start;   /* byte offset, or code-point index when codepoints != 0 */
    uint64_t end;     /* exclusive                                            */
} acx_match_t;

typedef struct acx_automaton acx_automaton_t; /* owns host + device tables    */
typedef struct acx_result acx_result_t;       /* owns device-resident results */

typedef struct acx_info {
    uint64_t n_patterns;
    uint64_t n_states;
    uint32_t n_classes;   /* byte equivalence classes                          */
    uint32_t stride;      /* row length of the dense table (pow2 >= n_classes) */
    uint32_t min_pattern_len;
    uint32_t max_pattern_len;
    uint64_t table_bytes; /* dense DFA in HBM                                  */
    uint32_t lds_hot_rows;/* rows staged in LDS by K1a's chunked walk          */
    int32_t kernel;       /* ACX_KERNEL_* actually selected                    */
    int32_t match_kind;
    int32_t device;       /* HIP device ordinal the tables live on             */
    uint32_t filter_q;    /* q-gram length of the K1b prefilter (0 = none)     */
} acx_info_t;

typedef struct acx_profile {
    double scan_ms;        /* accumulated HIP-event time of the scan kernel (K1) */
    uint64_t scan_launches;
    double post_ms;        /* kernels after the scan (collected only when ACX_PROFILE_POST is
                              set: it costs every call a wait for the previous one) */
    uint64_t scan_bytes;   /* haystack bytes scanned by those launches          */
    uint64_t raw_occurrences; /* occurrences emitted by K1 before resolution    */
    uint64_t prefix_hits;     /* K1b: prefix hits handed to the walk kernel      */
    uint64_t small_calls;     /* calls answered by K0 (whole call in one workgroup;
                                 counted whether or not profiling is enabled)     */
} acx_profile_t;

/* ---- process-wide ---- */
int acx_version(void);
const char *acx_last_error(void);
int acx_device_count(int *n);
int acx_set_device(int ordinal); /* device for subsequent acx_build on this thread */

/* ---- construction: replaces AhoCorasickBuilder::new().kind(..).match_kind(..)
 * .build(patterns) at src/lib.rs:186-215 (str) and 401-406 (bytes).
 * `blob` is the concatenation of the pattern bytes (UTF-8 for str patterns),
 * `offsets[n_patterns + 1]` delimits them.  Patterns are copied. */
int acx_build(const uint8_t *blob, const uint64_t *offsets, uint64_t n_patterns,
              int match_kind, int implementation, acx_automaton_t **out);
void acx_free_automaton(acx_automaton_t *a);
int acx_automaton_info(const acx_automaton_t *a, acx_info_t *out);
int acx_set_kernel(acx_automaton_t *a, int kernel); /* override the selection  */

/* ---- host-only compilation (no device needed): the compiled tables exactly
 * as they are uploaded to HBM.  For sizing an automaton before committing
 * device memory, for tooling, and for the CPU-side tests of the compiler.
 * This is NOT a matching path. ---- */
typedef struct acx_host_automaton acx_host_automaton_t;
typedef struct acx_host_tables {
    uint64_t n_patterns, n_states;
    uint32_t n_classes, stride, min_pattern_len, max_pattern_len;
    const uint8_t *classes;       /* 256: byte -> class                              */
    const uint32_t *table;        /* n_states * stride: id | OUT<<31 | OWN<<30; NULL when the dense form
                                     is not kept (dense == 0: n_states * stride * 4 > ACX_DENSE_LIMIT,
                                     default 256 MiB -- the reference's own "DFA for small sets, NFA
                                     beyond", README.md:173-177)                        */
    const uint32_t *own_off;      /* n_states + 1                                    */
    const uint32_t *own_pid;      /* patterns ending exactly at a state, id order    */
    const uint32_t *dlink;        /* dictionary-suffix link or 0xFFFFFFFF            */
    const uint32_t *level_start;  /* max_pattern_len + 2: first BFS id of each depth */
    const uint32_t *pattern_len;  /* n_patterns                                      */
    const uint32_t *rank;         /* n_patterns: rank in (len desc, id asc)          */
    const uint32_t *filter_xy;    /* K1b level 1: 2^filter_entries_log2 x {X, Y} signature words
                                     (bit layout: csrc/automaton.hpp, filter_bit)      */
    const uint32_t *prefix_table; /* K1b level 2: 2^prefix_table_log2 x {key lo, key hi, meta, code}:
                                     meta = key length K (1..8) | next << 4 | MORE << 8 (0xFFFFFFFF =
                                     empty; MORE: 16-bit filter of the keys with this home slot that sit
                                     further along the probe sequence: bit ((hash >> 11) & 15) of each).  next = 0: code = the only pattern with this key, or
                                     0x80000000 | index into prefix_lists; next = N: redirect -- look the
                                     first N bytes up.  A key = the first min(8, shortest pattern of its
                                     group) bytes of a pattern, a group = the patterns sharing their
                                     first filter_q2 bytes; a group's single key sits at the hash of those
                                     bytes, several keys behind a redirect entry (csrc/automaton.cpp)   */
    const uint32_t *prefix_lists; /* {count, pattern id, ...} per key shared by several patterns   */
    uint32_t filter_q, filter_q2; /* prefix lengths used by level 1 / level 2 (first-level keys)   */
    uint32_t filter_entries_log2, prefix_table_log2;
    double filter_density;        /* fraction of X bits set                            */
    uint32_t n_prefix_keys;       /* entries of prefix_table in use                    */
    uint32_t n_prefix_lists;      /* u32 words of prefix_lists                         */
    const uint32_t *prefix_bitmap;/* 2^(prefix_table_log2 + 3) bits: bit (hash >> (29 - prefix_table_log2)) of
                                     the first filter_q2 bytes of every group (hash: acx_prefix_slot(gram,
                                     filter_q2, 32)); K1b asks it before the table when the table is large */
    /* the compressed form (always present): trie edges + failure links               */
    uint32_t dense;               /* 1: `table` exists                                 */
    const uint32_t *first_child;  /* n_states + 1: children of s = ids [first_child[s], first_child[s+1]) */
    const uint8_t *in_byte;       /* n_states: byte on the edge into the state (children ascending)     */
    const uint32_t *fail;         /* n_states: failure link                                             */
    const uint8_t *state_flags;   /* n_states: bit 1 = reports something, bit 0 = ends a pattern itself */
    /* K1a's failureless walk (n_classes <= 32, else NULL; layouts: csrc/automaton.hpp)                 */
    const uint32_t *walk_t3b;     /* 33 792 words: by the symbols (low five bits) s0, s1, s2 of three bytes,
                                     word ((s0 << 5 | s1) * 33 + s2): the symbols a fourth byte can have on a
                                     trie path of depth 4; ~0: a pattern of <= 3 bytes ends on the path       */
    const uint32_t *walk_t3r;     /* n_classes^3 x {children bitmap by class, first child | SHORT << 31}: the
                                     depth-3 node of a class triple ((c0 * n_classes + c1) * n_classes + c2) */
    const uint32_t *walk_grec;    /* n_states x 4: {children bitmap, first child | OWN << 31, own pattern, 0}
                                     or a tail {bytes 0-3, 1 << 30 | n << 24, pattern, bytes 4-7}: the rest
                                     of the only pattern below the node, n <= 8 bytes                         */
    /* K1b, short patterns (round 4; layouts: csrc/automaton.hpp).  Patterns of 1 and 2 bytes are kept out of
       the prefilter tables above (filter_q / filter_q2 are taken from the shortest of the OTHER patterns,
       long_min_len) and found by a side test instead                                                       */
    uint32_t long_min_len;        /* shortest pattern of 3 bytes or more (5 when there is none)             */
    uint32_t n_short;             /* patterns of 1 or 2 bytes (0: the two tables below are NULL)             */
    uint32_t short_min_len;       /* the shortest of them                                                    */
    const uint32_t *short_xy;     /* 256 x {X, Y} by middle byte b(j+1): bit (b(j) & 31) of X -- a short pattern
                                     may start at j; bit (b(j+2) & 31) of Y -- one may start at j+1 (superset)  */
    const uint32_t *short_codes;  /* [b0]: the 1-byte pattern b0; [256 + (b0 | b1 << 8)]: the 2-byte pattern
                                     (b0, b1): pattern id, 0x80000000 | index into prefix_lists, or 0xFFFFFFFF  */
    /* K1b, anchors (round 4; csrc/automaton.hpp): a pattern is filed under the bytes at offset
       pattern_shift[i] (0 .. 12; crowded beginnings move away from theirs); filter_xy, prefix_table keys and
       the 12 tail bytes of a pattern's info are taken from that anchored suffix; a code in prefix_table /
       prefix_lists is pattern id | shift << 24: "the pattern may start `shift` bytes in front of the hit"   */
    uint32_t max_shift;           /* largest shift in use (0: none, pattern_head is NULL)                    */
    const uint8_t *pattern_shift; /* n_patterns                                                              */
    const uint32_t *pattern_head; /* n_patterns x 4: the pattern's first 12 bytes (what lies in front of the anchor) */
} acx_host_tables_t;
int acx_compile_host(const uint8_t *blob, const uint64_t *offsets, uint64_t n_patterns,
                     int match_kind, acx_host_automaton_t **out);
int acx_host_tables(const acx_host_automaton_t *h, acx_host_tables_t *out);
uint32_t acx_filter_hash(uint32_t gram);   /* level-1 hash of a little-endian (Q-1)-gram   */
uint32_t acx_prefix_slot(uint64_t gram, uint32_t salt, uint32_t log2); /* home slot of the `salt` low
                                                                         bytes of gram */
void acx_free_host(acx_host_automaton_t *h);

/* ---- concurrency: every function taking an acx_automaton_t may be called from several
 * threads on ONE handle at the same time (the reference lets threads search one object
 * concurrently: methods take a shared reference and release the GIL, src/lib.rs:238, 261,
 * 433, 438).  Calls run side by side on separate HIP streams (up to ACX_MAX_CONCURRENCY,
 * default 4, per handle); acx_free_automaton must not race with them. */

/* ---- the hot path, host-memory form.  Replaces get_matches + collect:
 * src/lib.rs:42-68 with consumers 229-249 (str: codepoints = 1 applies the
 * get_byte_to_code_point fix-up of 73-88 on the device) and 422-434 (bytes).
 * `hay` is borrowed for the call.  `*out` is library-owned (acx_free_matches).
 * Order and content are bit-exact with the reference iterator.
 * Large haystacks are staged through pinned chunks by several host threads, each chunk's DMA
 * under the next chunk's copy; large results are returned in pinned host memory.
 * Short haystacks (<= 16 KiB; round 6): a loop of calls is answered by a RESIDENT workgroup per calling context that stays on
 * the device between the calls and is fed through pinned host memory (a poll on either side instead of a kernel launch:
 * ~5 us per call instead of ~12; the reference's benchmark loop, benchmarks/test_comparison.py:113-124).  It leaves
 * 200 us after the last call and at most 1 ms after its launch (ACX_RESIDENT_IDLE_US / ACX_RESIDENT_LIFE_US), and at once
 * when its context is used for anything else or the handle is freed: a device-wide synchronisation elsewhere in the
 * process waits that long at most.  ACX_NO_RESIDENT=1: a launch per call.  Haystacks up to 1 MiB are copied into
 * pinned host memory by the calling thread and read there by the scan (ACX_INPLACE_MAX, bytes; 0: never). */
int acx_find(acx_automaton_t *a, const uint8_t *hay, uint64_t len,
             int overlapping, int codepoints, acx_match_t **out, uint64_t *n_out);
void acx_free_matches(acx_match_t *m);

/* Precondition of codepoints = 1 (all entry points): haystack AND patterns are valid UTF-8 -- what
 * the reference's str API guarantees by construction (src/lib.rs:147-160, 232).  The device converts
 * a match's start through the lead-byte counts and takes its end as start + the pattern's own code
 * points; for a pattern that ends inside a character the reference's table would hold usize::MAX
 * (src/lib.rs:73-88) -- that input cannot come from a str and is not supported here. */

/* ---- batched host form (new API; parity definition:
 * batch(hs)[i] == find(hs[i])).  `hay` is the concatenation
 * of n_hay haystacks delimited by offsets[n_hay + 1]; counts[n_hay] receives
 * the number of matches of each haystack; matches are grouped by haystack in
 * order, offsets LOCAL to each haystack. */
int acx_find_batch(acx_automaton_t *a, const uint8_t *hay, const uint64_t *offsets,
                   uint64_t n_hay, int overlapping, int codepoints,
                   acx_match_t **out, uint64_t *n_out, uint64_t *counts);

/* ---- one process, several devices (north_star: "a batched find_matches_as_indexes over many
 * haystacks shards naturally across the 8 GPUs"; the reference's shape is ONE call from ONE process,
 * benchmarks/test_comparison.py:113-124).  acx_replicate compiles the automaton of `a` again on
 * another device (same patterns, match kind, implementation hint).  acx_find_batch_multi cuts the
 * batch into n_handles contiguous ranges of haystacks (acx_shard_range: sizes differ by at most one),
 * runs acx_find_batch on every handle from its own host thread and concatenates: the result is
 * identical to acx_find_batch on one handle.  No device-to-device traffic; the shards' match counts
 * are combined on the host (the multi-PROCESS form all-gathers them over RCCL instead:
 * ahocorasick_rs_amd/distributed.py). */
int acx_replicate(const acx_automaton_t *a, int device, acx_automaton_t **out);
int acx_automaton_device(const acx_automaton_t *a);
void acx_shard_range(uint64_t n_items, int shard, int n_shards, uint64_t *lo, uint64_t *hi);
int acx_find_batch_multi(acx_automaton_t *const *handles, int n_handles, const uint8_t *hay,
                         const uint64_t *offsets, uint64_t n_hay, int overlapping, int codepoints,
                         acx_match_t **out, uint64_t *n_out, uint64_t *counts);

/* ---- one process PER device: the count exchange over RCCL (north_star: "RCCL over xGMI only to gather
 * per-shard match counts").  The search itself needs no communicator: every rank scans its own range of
 * the batch (acx_shard_range) with its own replica of the automaton.  What the ranks exchange is one u64
 * each -- their match counts -- from which acx_output_offsets gives every rank its place in the global
 * output.  acx_comm_init_all: one process holding n devices (ncclCommInitAll; local_counts has n entries, one
 * per device in the order given).  acx_comm_unique_id + acx_comm_init_rank: one process per device
 * (ncclCommInitRank; the 128-byte id is created on one rank and carried to the others by the host's own
 * means -- a file, a socket, MPI); local_counts has one entry.  all_counts receives `world` entries on
 * every caller.  librccl is loaded on first use.  (The torch.distributed form of the same exchange:
 * ahocorasick_rs_amd/distributed.py.) */
#define ACX_COMM_ID_BYTES 128
typedef struct acx_comm acx_comm_t;
int acx_comm_init_all(const int *devices, int n, acx_comm_t **out);
int acx_comm_unique_id(uint8_t id[ACX_COMM_ID_BYTES]);
int acx_comm_init_rank(const uint8_t id[ACX_COMM_ID_BYTES], int world, int rank, int device, acx_comm_t **out);
int acx_comm_world(const acx_comm_t *c);
int acx_comm_local_ranks(const acx_comm_t *c);
int acx_comm_allgather_counts(acx_comm_t *c, const uint64_t *local_counts, uint64_t *all_counts);
void acx_comm_free(acx_comm_t *c);
/* offsets[r] = the matches of the ranks in front of r, offsets[world] = their total (exclusive prefix) */
void acx_output_offsets(const uint64_t *counts, int world, uint64_t *offsets);

/* ---- device-resident form (what bench.py times).  d_hay is a device pointer
 * to `len` bytes on the automaton's device.  Batches: either d_offsets
 * (device, n_hay + 1 u64, ragged) or uniform_len > 0 (n_hay * uniform_len ==
 * len) or neither (one haystack).  Results stay in HBM inside *out.
 * The call returns as soon as the number of matches is known (acx_result_count is valid
 * at once); the kernels that write the records may still be running on the library's
 * stream -- every accessor of the records waits for them, d_hay (and d_offsets) must stay
 * valid until one of them or acx_free_result has returned. */
int acx_find_device(acx_automaton_t *a, const void *d_hay, uint64_t len,
                    const uint64_t *d_offsets, uint64_t n_hay, uint64_t uniform_len,
                    int overlapping, int codepoints, acx_result_t **out);
uint64_t acx_result_count(const acx_result_t *r);
const acx_match_t *acx_result_device_matches(const acx_result_t *r);
const uint64_t *acx_result_device_counts(const acx_result_t *r); /* per haystack, or NULL */
int acx_result_copy(const acx_result_t *r, acx_match_t *host_out);
int acx_result_copy_counts(const acx_result_t *r, uint64_t *host_counts);
void acx_free_result(acx_result_t *r);

/* ---- measurement hooks (HIP events on the library's stream) ---- */
/* on = 0: off; 1: every call carries the event pair around its scan kernel; N > 1: every N-th call of
 * a context does (the pair costs the dispatch ~6 us: a sampled measurement leaves the other calls
 * alone).  The totals of acx_profile_read cover the measured calls only. */
int acx_profile_enable(acx_automaton_t *a, int on);
int acx_profile_read(acx_automaton_t *a, acx_profile_t *out, int reset);
/* Which way the handle's calls went (always counted, no events involved): out[0] calls finished by the sparse kernels
 * alone, [1] calls that also ran the hot pipeline (a dense stretch of the input costs the groups it lies in, not the
 * call: reference behaviour /root/reference/src/lib.rs:59), [2] hot groups in all, [3] prefix hits beyond their tiles'
 * slots in all, [4] calls on the tile-ordered dense path, [5] calls on its radix-sort form, [6] calls that were redone
 * with a larger overflow list, [7] calls K0 answered, [8] byte ranges searched for calls that were cut (more than 2^32
 * occurrences in one pass: the pieces count in [0 .. 7] as well), [9] calls that were repeated with the WIDE form of the
 * sparse path's post stage (a match every 100 - 500 bytes: the context keeps that form while its inputs are like that),
 * [10] launches of a context's RESIDENT K0 (acx_find on short haystacks: one workgroup stays on the device between the
 * calls of a loop and is fed through pinned host memory -- [7] counts the calls, [10] the launches they cost), [11] calls
 * of acx_find whose haystack (beyond K0's sizes, up to 1 MiB) the scan read IN PLACE from pinned host memory instead of
 * a copy in HBM.  reset != 0 clears the counters. */
#define ACX_PATH_STATS 12
int acx_path_stats(acx_automaton_t *a, uint64_t out[ACX_PATH_STATS], int reset);

/* ---- device memory helpers so that a host without torch can stage data ---- */
int acx_device_alloc(void **d_ptr, uint64_t bytes);
int acx_device_free(void *d_ptr);
int acx_device_upload(void *d_dst, const void *h_src, uint64_t bytes);
int acx_device_download(void *h_dst, const void *d_src, uint64_t bytes);
int acx_device_synchronize(void);
int acx_device_synchronize_on(int device); /* the same on a given device (the calling thread's current device is left alone) */

/* ---- seeded synthetic haystacks generated in HBM (bench / tests).  Bit-exact
 * twins of tests/gen.py gen_uniform (kind 0, alphabet a-z) and gen_textlike
 * (kind 1, patterns of `a` planted every 1024 B). ---- */
int acx_generate_haystack(acx_automaton_t *a, void *d_dst, uint64_t len, int kind,
                          uint64_t seed, uint64_t stream_offset);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H */
