/*
 * oracle/ac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the algorithm the reference executes for its
 * hot path.  The reference (`/root/reference/src/lib.rs`) delegates all
 * matching to the third-party crate `aho-corasick` 1.1.4 (Cargo.lock:6-7),
 * whose source is NOT under /root/reference; this file restates that crate's
 * published algorithm (noncontiguous NFA construction, DFA construction,
 * `try_find_fwd`, `try_find_overlapping_fwd`, `FindIter`) and the reference's
 * own glue (`get_matches` src/lib.rs:42-68, `get_byte_to_code_point`
 * src/lib.rs:73-88).
 *
 * Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
 * may load this library.  The product (ahocorasick_rs_amd/) never does.
 *
 * Pinning: checked against every known-answer vector in the reference's
 * tests/README (tests/golden/reference_vectors.json), against the genuine
 * crate 1.1.4 embedded in `tokenizers` 0.22.2 (LeftmostLongest fixtures,
 * tests/golden/ll_tokenizers_*.json) and against Python `re` alternation
 * (LeftmostFirst fixtures).  See tests/test_oracle_golden.py.
 */
#ifndef AC_ORACLE_H
#define AC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct aco aco_t;

/* match_kind: 0 Standard, 1 LeftmostFirst, 2 LeftmostLongest
 *   (reference enum PyMatchKind, src/lib.rs:92-108)
 * kind: 0 NoncontiguousNFA (fail links followed at search time), 2 DFA
 *   (dense, byte-classed, premultiplied).  ContiguousNFA (1) is served by the
 *   noncontiguous walker -- all kinds give identical results
 *   (reference enum Implementation, src/lib.rs:111-128).
 * Patterns: concatenated bytes `blob`, `off[n+1]` offsets.  Empty patterns are
 * rejected by the reference before the builder (src/lib.rs:204-208,386-389);
 * here they return NULL. */
aco_t *aco_build(const uint8_t *blob, const uint64_t *off, uint64_t n,
                 int match_kind, int kind);
void aco_free(aco_t *a);

uint64_t aco_num_states(const aco_t *a);
uint64_t aco_num_classes(const aco_t *a);
uint64_t aco_max_pattern_len(const aco_t *a);
uint64_t aco_min_pattern_len(const aco_t *a);

/* Non-overlapping iteration == crate `try_find_iter` (called at
 * src/lib.rs:59).  Writes up to `cap` (pattern,start,end) u64 triples to
 * `out` and returns the TOTAL number of matches (call again with a larger
 * buffer if > cap). */
int64_t aco_find_iter(const aco_t *a, const uint8_t *hay, uint64_t len,
                      uint64_t *out, uint64_t cap);

/* Overlapping iteration == crate `try_find_overlapping_iter` (called at
 * src/lib.rs:53).  Returns -1 (the crate's MatchError::UnsupportedOverlapping,
 * mapped to ValueError at src/lib.rs:36-39) unless match_kind == Standard. */
int64_t aco_find_overlapping_iter(const aco_t *a, const uint8_t *hay,
                                  uint64_t len, uint64_t *out, uint64_t cap);

/* Count-only variants (no output buffer traffic) used by the CPU baseline. */
int64_t aco_count_iter(const aco_t *a, const uint8_t *hay, uint64_t len);

/* src/lib.rs:73-88: out has len+1 entries; UINT64_MAX at non-boundaries. */
void aco_byte_to_code_point(const uint8_t *hay, uint64_t len, uint64_t *out);

/* Full str entry point (src/lib.rs:229-249): matches over UTF-8 bytes mapped
 * through byte_to_code_point.  overlapping as above. */
int64_t aco_find_str(const aco_t *a, const uint8_t *utf8, uint64_t len,
                     int overlapping, uint64_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
