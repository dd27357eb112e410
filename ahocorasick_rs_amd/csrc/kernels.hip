// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the MI355X-native
// Aho-Corasick matcher.  This file replaces the reference's match loop: the
// crate iterators drained at /root/reference/src/lib.rs:229-249 and 422-434
// (`try_find_iter` / `try_find_overlapping_iter`, called at src/lib.rs:59, 53).
//
// Pipeline (all on the device, one stream):
//   K1   scan
//        K1b prefilter   position-parallel: coalesced 16-B loads, a q-gram signature table
//                        in LDS says whether a pattern can start here; survivors are ballot-
//                        compacted into a per-wave LDS queue and settled against the exact
//                        prefix table (HBM, L2-resident).  Output: *prefix hits* in per-tile
//                        hit slots (the wave that owns a 4 KiB tile is their only producer)
//        K1a dfa_walk    one lane walks one chunk of the stream through the dense DFA; class
//                        map + hot (shallow, BFS-first) rows live in LDS, cold rows come from
//                        the HBM table.  Output: verified occurrences, same hit slots
//   K2   k_tile_main     one workgroup per 64 tiles: verifies the hits against the pattern
//                        bytes, orders the occurrences, applies the match kind (Standard /
//                        LeftmostFirst / LeftmostLongest, overlapping or not) exactly as the
//                        reference iterators would; k_tile_write places and writes them
//                        into the final (pattern, start, end) records.  Forms (round 6): narrow
//                        staged words (32 bits, 128 threads, sixteen groups per CU) for sets
//                        like the headline's, wide words otherwise; the WIDE FORM (64
//                        occurrences per bucket) for inputs with a match every 100-500 bytes
//        dense stretches the groups the sparse kernels cannot finish: the hot pipeline (k_hot_verify ->
//                        k_dense_main -> k_tile_write<HOT>); inputs dense everywhere: K1b in region
//                        mode -> k_dense_verify -> k_dense_main (compact or full) -> k_tile_write;
//                        the path of last resort: k_walk_hits -> rocPRIM radix sort -> k_resolve
//   K0   small haystacks: the whole call in one workgroup
//   K3   UTF-8 byte offset -> code-point index (get_byte_to_code_point,
//        src/lib.rs:73-88) by per-KiB lead-byte counts + prefix sum
//
// This is integer, HBM-bound work: no MFMA anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include <hip/hip_ext.h>

#include "automaton.hpp"
#include "kernels.hpp"

namespace acx {

// ---------------------------------------------------------------------------
// common device helpers
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streaming (non-temporal) 16-byte load: the haystack is read exactly once, so
// keep it from evicting the DFA table out of L2
__device__ __forceinline__ u32x4 load16_stream(const uint8_t *p) {
    return __builtin_nontemporal_load((const u32x4 *)p);
}

// per-workgroup view of the occurrence sink
//   region mode (dense output): region base + LDS slot counter
//   hit-slot mode (sparse output, K1a): verified occurrences go to the hit slots of the tile
//   their START lies in, as records k_tile_main takes without verification
struct BlockSink {
    uint4 *recs;
    unsigned long long *lcount; // LDS (64-bit: a region's count never wraps, the host tests the sum for ACX_ETOOBIG)
    uint64_t region_cap;
    int key_mode;
    uint4 *hslots;
    uint32_t *hcnt;
    uint32_t *abort_flag;
    uint32_t lead;
    uint32_t cnt_nw, cnt_iters;
};

// region_cap == 0 (second pass of the dense path): the regions lie at the exclusive prefix of the
// first pass's counts, stored behind the counts: block_counts[gridDim.x + b]
__device__ __forceinline__ BlockSink block_sink(const Sink &K, unsigned long long *lcount) {
    uint64_t base = (uint64_t)blockIdx.x * K.region_cap, cap = K.region_cap;
    if (K.region_cap == 0 && K.block_counts && !K.hslots) {
        const uint64_t *rb = K.block_counts + gridDim.x;
        base = rb[blockIdx.x];
        cap = rb[blockIdx.x + 1] - base;
    }
    return BlockSink{K.recs + base, lcount, cap, K.key_mode,
                     K.hslots, K.hcnt, K.abort_flag, K.lead, K.cnt_nw, K.cnt_iters};
}

// region mode: store one occurrence (ONE 16-byte store) into the next slot of the region
// (records carry the pattern's length so that nothing downstream has to gather it again)
__device__ __forceinline__ void emit_key(const BlockSink &K, uint64_t key, uint32_t pid, uint32_t plen) {
    const uint64_t slot = atomicAdd(K.lcount, 1ull); // LDS atomic
    if (slot < K.region_cap) K.recs[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, plen);
}

// Wave-aggregated variant for code where many lanes emit together (walk kernel): one LDS
// atomic per wave reserves the slots.
__device__ __forceinline__ void emit_key_agg(const BlockSink &K, bool ok, uint64_t key, uint32_t pid,
                                             uint32_t plen) {
    const unsigned long long fm = __ballot(ok);
    if (!fm) return;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below_me = (1ull << lane) - 1;
    const uint32_t leader = (uint32_t)__builtin_ctzll(fm);
    unsigned long long sbase = 0;
    if (lane == leader) sbase = atomicAdd(K.lcount, (unsigned long long)__popcll(fm));
    sbase = __shfl(sbase, leader);
    if (ok) {
        const uint64_t slot = sbase + (uint32_t)__popcll(fm & below_me);
        if (slot < K.region_cap) K.recs[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, plen);
    }
}

// A prefix hit / occurrence record in the hit slots is two quads:
//   {position lo, position hi, code, aux} {16 haystack bytes at the position}
// code = the id of the only pattern with that prefix, or HIT_LIST | index into blist
// ({count, pid, ...}); HIT_RETRY: the window has to be looked up in the prefix table first;
// HIT_VERIFIED | pid: an occurrence found by the DFA walk (aux = pattern length, no second
// quad) -- nothing left to verify.
constexpr uint32_t HIT_LIST = 0x80000000u;
constexpr uint32_t HIT_VERIFIED = 0x40000000u;
constexpr uint32_t HIT_NONE = 0xFFFFFFFFu;
constexpr uint32_t HIT_RETRY = 0xFFFFFFFEu; // K1b found another key in the home slot (MORE set): look the window up

// The emit paths are cold and out of line; they read the automaton through a
// pointer to its device-resident copy so that the kernels never have to spill
// their by-value kernel arguments to scratch for them.
__device__ __forceinline__ void emit_one(const DevAutomaton *A, const BlockSink &K, uint32_t pid,
                                         uint64_t end) {
    uint64_t key;
    const uint32_t plen = A->plen[pid];
    if (K.hslots) { // sparse output: arrival rank inside the tile of the occurrence's start
        const uint64_t start = end - plen;
        const uint64_t tile = (start + K.lead) >> TILE_BITS;
        const uint32_t r = atomicAdd(&K.hcnt[hcnt_index(tile, K.cnt_nw, K.cnt_iters)], 1u);
        if (r < HIT_SLOTS)
            K.hslots[(tile * HIT_SLOTS + r) * 2] = make_uint4((uint32_t)start, (uint32_t)(start >> 32),
                                                              HIT_VERIFIED | pid, plen);
        else
            *K.abort_flag = 1;
        return;
    }
    if (K.key_mode == 0) {
        key = (end << A->rank_bits) | A->rank[pid];
    } else {
        uint64_t start = end - plen;
        key = (start << A->rank_bits) | (K.key_mode == 1 ? pid : A->rank[pid]);
    }
    emit_key(K, key, pid, plen);
}

// every pattern that ends at state s (own patterns, then the dictionary-suffix
// chain: progressively shorter suffixes)
// (the sink view lives in LDS and travels as a pointer: by value it was a 592-byte scratch frame
// in front of every call site of this cold path)
__device__ __noinline__ void emit_state(const DevAutomaton *A, const BlockSink *Kp, uint32_t s,
                                        uint64_t end) {
    const BlockSink K = *Kp;
    for (uint32_t t = s; t != NONE; t = A->dlink[t]) {
        uint32_t b = A->own_off[t], e = A->own_off[t + 1];
        for (uint32_t k = b; k < e; k++) emit_one(A, K, A->own_pid[k], end);
    }
}

// first index i in [0, n] with off[i] > x
__device__ __forceinline__ uint64_t upper_bound_u64(const uint64_t *off, uint64_t n,
                                                    uint64_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (off[mid] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// end (exclusive) of the haystack containing stream position x (x < len)
__device__ __forceinline__ uint64_t segment_end(const Segments &G, uint64_t len, uint64_t x) {
    if (G.uniform_len) return (x / G.uniform_len + 1) * G.uniform_len;
    if (G.offsets) return G.offsets[upper_bound_u64(G.offsets, G.n_hay + 1, x)];
    return len;
}

// [*lo, *hi) = the haystack containing stream position x (x < len)
__device__ __forceinline__ void segment_bounds(const Segments &G, uint64_t len, uint64_t x, uint64_t *lo, uint64_t *hi) {
    if (G.uniform_len) { *lo = (x / G.uniform_len) * G.uniform_len; *hi = *lo + G.uniform_len; }
    else if (G.offsets) { const uint64_t h = upper_bound_u64(G.offsets, G.n_hay + 1, x); *lo = G.offsets[h - 1]; *hi = G.offsets[h]; }
    else { *lo = 0; *hi = len; }
}

// ---------------------------------------------------------------------------
// K1a: chunked DFA walk
// ---------------------------------------------------------------------------
// LDS layout: [0,256) byte classes; [256,272) slot counter; then hot rows as
// u16 entries:
//   0xFFFF           -> target not representable, read the HBM table
//   id | flags<<14   -> target id < 0x3FFF, flags = (OUT, OWN)
constexpr size_t K1A_LDS_HEADER = 272;
struct WalkCtx {
    const uint8_t *lcls;
    const uint16_t *lrows;
    uint32_t hot_rows;
};

// the trie child of s on byte b (0: none): the root by table, the others by their short sorted list
__device__ __forceinline__ uint32_t trie_child(const DevAutomaton &A, uint32_t s, uint32_t b) {
    if (s == 0) return A.root_next[b];
    for (uint32_t c = A.first_child[s], e = A.first_child[s + 1]; c < e; c++) {
        const uint32_t x = A.in_byte[c];
        if (x == b) return c;
        if (x > b) break;
    }
    return 0;
}

// one step of the compressed automaton (trie edges + failure links): the classic Aho-Corasick
// transition, for automata whose dense table is not kept.  Returns id | OUT << 31 | OWN << 30.
__device__ __forceinline__ uint32_t nfa_step(const DevAutomaton &A, uint32_t s, uint32_t b) {
    for (;;) {
        const uint32_t c = trie_child(A, s, b);
        if (c) { s = c; break; }
        if (s == 0) break;
        s = A.fail[s];
    }
    return s | ((uint32_t)A.sflags[s] << 30);
}

__device__ __forceinline__ uint32_t dfa_step(const DevAutomaton &A, const WalkCtx &W,
                                             uint32_t s, uint32_t byte) {
    if (!A.table) return nfa_step(A, s, byte); // (uniform: the automaton has no dense table)
    uint32_t c = W.lcls[byte];
    if (s < W.hot_rows) {
        uint32_t v = W.lrows[(s << A.stride2) + c];
        if (v != 0xFFFFu) return (v & 0x3FFFu) | ((v & 0xC000u) << 16);
    }
    return A.table[((size_t)s << A.stride2) + c];
}

template <bool EMIT>
__device__ __forceinline__ uint32_t walk_span(const DevAutomaton &A, const DevAutomaton *Ad,
                                              const WalkCtx &W, const BlockSink *K,
                                              const uint8_t *hay, uint64_t pos, uint64_t lim,
                                              uint32_t s) {
#define ACX_STEP(BYTE, POS)                                            \
    {                                                                  \
        uint32_t e_ = dfa_step(A, W, s, (BYTE));                       \
        s = e_ & ID_MASK;                                              \
        if (EMIT && (e_ & FLAG_OUT)) emit_state(Ad, K, s, (POS) + 1);  \
    }
    while (pos < lim && ((uintptr_t)(hay + pos) & 15)) {
        ACX_STEP(hay[pos], pos);
        pos++;
    }
    while (pos + 16 <= lim) {
        u32x4 v = load16_stream(hay + pos);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t b = (w[k >> 2] >> ((k & 3) * 8)) & 0xFF;
            ACX_STEP(b, pos + k);
        }
        pos += 16;
    }
    while (pos < lim) {
        ACX_STEP(hay[pos], pos);
        pos++;
    }
#undef ACX_STEP
    return s;
}

__global__ __launch_bounds__(1024) void k1a_dfa_walk(DevAutomaton A, const DevAutomaton *Ad,
                                                     Segments G, Sink GK,
                                                     const uint8_t *__restrict__ hay,
                                                     uint64_t len, uint32_t chunk,
                                                     uint32_t lds_rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lcls = smem;
    unsigned long long *lcount = (unsigned long long *)(smem + 256);
    uint16_t *lrows = (uint16_t *)(smem + K1A_LDS_HEADER);
    __shared__ BlockSink sK;
    if (threadIdx.x == 0) { *lcount = 0; sK = block_sink(GK, lcount); }
    const BlockSink *K = &sK;
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x)
        ((uint32_t *)lcls)[i] = ((const uint32_t *)A.classes)[i];
    {
        uint32_t nwords = ((lds_rows << A.stride2) + 1) >> 1; // hot16 is padded to 16 B
        const uint32_t *src = (const uint32_t *)A.hot16;
        uint32_t *dst = (uint32_t *)lrows;
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    WalkCtx W{lcls, lrows, lds_rows};
    const bool seg = G.uniform_len != 0 || G.offsets != nullptr;
    const uint64_t nchunks = (len + chunk - 1) / chunk;
    const uint64_t warm = A.max_len ? A.max_len - 1 : 0;
    for (uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks;
         ch += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c0 = ch * chunk;
        uint64_t c1 = c0 + chunk < len ? c0 + chunk : len;
        uint64_t pos = c0 > warm ? c0 - warm : 0;
        uint64_t nb = ~0ull; // next haystack boundary strictly after pos
        uint64_t h = 0;
        if (seg) {
            if (G.uniform_len) {
                nb = (pos / G.uniform_len + 1) * G.uniform_len;
            } else {
                h = upper_bound_u64(G.offsets, G.n_hay + 1, pos); // offsets[h] > pos
                nb = G.offsets[h];
            }
        }
        uint32_t s = 0;
        while (pos < c1) {
            uint64_t lim = c1 < nb ? c1 : nb;
            bool emit = pos >= c0;
            if (!emit && c0 < lim) lim = c0;
            s = emit ? walk_span<true>(A, Ad, W, K, hay, pos, lim, s)
                     : walk_span<false>(A, Ad, W, K, hay, pos, lim, s);
            pos = lim;
            if (pos == nb) { // a new haystack starts here: fresh start state
                s = 0;
                if (G.uniform_len) {
                    nb += G.uniform_len;
                } else {
                    do { h++; nb = (h <= G.n_hay) ? G.offsets[h] : ~0ull; } while (nb == pos);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && GK.block_counts) GK.block_counts[blockIdx.x] = *lcount;
}

// ---------------------------------------------------------------------------
// K1a, compact form (automata of at most 65 535 states, one haystack): the DFA as u16 with rows
// of n_classes entries -- 3.4 MiB for the headline automaton: it stays in one XCD's L2 -- and
// NCH independent chains per lane: a lane walks NCH adjacent chunks in lock step, so NCH dependent
// lookups are in flight per lane instead of one.  Every chain starts from the root `warm` (>=
// max_len - 1, a multiple of 16) bytes before its chunk and reports the matches that END inside the
// chunk.  The haystack is read as aligned 16-byte blocks of the index space (index = position +
// lead).  LDS: class map + the first rows of the table (walk order: states that report nothing
// first, shallow ones first among them).
constexpr int K1A_CHAINS = 4;
__global__ __launch_bounds__(1024) void k1a_walk16(DevAutomaton A, const DevAutomaton *Ad, Sink GK,
                                                   const uint8_t *__restrict__ base, uint64_t total,
                                                   uint32_t lead, uint32_t chunk, uint32_t warm,
                                                   uint32_t lds_rows, unsigned long long *lds_hits) {
    constexpr int NCH = K1A_CHAINS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lcls = smem;
    unsigned long long *lcount = (unsigned long long *)(smem + 256);
    uint16_t *lrows = (uint16_t *)(smem + K1A_LDS_HEADER);
    __shared__ BlockSink sK;
    if (threadIdx.x == 0) { *lcount = 0; sK = block_sink(GK, lcount); }
    const BlockSink *K = &sK;
    const uint32_t NC = A.n_classes, plain = A.walk_plain;
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) ((uint32_t *)lcls)[i] = ((const uint32_t *)A.classes)[i];
    {
        const uint32_t nwords = (lds_rows * NC + 1) >> 1; // table16 is padded
        const uint32_t *src = (const uint32_t *)A.table16;
        uint32_t *dst = (uint32_t *)lrows;
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint64_t nchunks = (total + chunk - 1) / chunk, ngroups = (nchunks + NCH - 1) / NCH;
    const uint64_t total16 = (total + 15) & ~15ull;
    uint32_t hot = 0, all = 0; // statistics: transitions served by LDS / all (lds_hits != null)
    // (scalar variables, not arrays: the chains must live in registers)
    // a lane reads its chains in 32-byte pieces (two quads): with 16-byte pieces every 64-byte
    // line was fetched four times -- 4096 chains per CU keep 256 KiB of partly consumed lines
    // alive, more than L1 and the XCD's share of L2 hold (measured: 5.5 GB fetched per GiB)
#define K1A_LOAD(k, BLK, BLK2)                                                                  \
    {                                                                                           \
        const int64_t idx_ = first + (int64_t)(k) * chunk + i;                                  \
        BLK = u32x4{0, 0, 0, 0}; BLK2 = u32x4{0, 0, 0, 0};                                      \
        if (idx_ >= 0 && (uint64_t)idx_ < total16) BLK = *(const u32x4 *)(base + idx_);         \
        if (idx_ + 16 >= 0 && (uint64_t)(idx_ + 16) < total16) BLK2 = *(const u32x4 *)(base + idx_ + 16); \
    }
#define K1A_BYTE(BLK, j) ((((j) < 4 ? BLK.x : (j) < 8 ? BLK.y : (j) < 12 ? BLK.z : BLK.w) >> (8 * ((j) & 3))) & 0xFF)
    // one byte of one chain, everything checked (blocks that touch the ends of the stream)
#define K1A_STEP(k, BLK, S, j)                                                                  \
    {                                                                                           \
        const int64_t idx_ = first + (int64_t)(k) * chunk + i + (j);                            \
        if (idx_ >= (int64_t)lead && (uint64_t)idx_ < total) { /* outside the stream the chain rests */ \
            const uint32_t e_ = S * NC + lcls[K1A_BYTE(BLK, j)];                                \
            uint32_t t_;                                                                        \
            if (S < lds_rows) { t_ = lrows[e_]; hot++; }                                        \
            else t_ = A.table16[e_];                                                            \
            all++;                                                                              \
            S = t_;                                                                             \
            if (emit && t_ >= plain) emit_state(Ad, K, A.walk_bfs[t_], (uint64_t)idx_ + 1 - lead); \
        }                                                                                       \
    }
    // one byte of ALL chains (blocks inside the stream): the four class reads, then the four row
    // reads (LDS for hot states, an exec-masked gather of the L2-resident table for cold ones) are
    // issued back to back -- four dependent chains in flight per lane, not one
#define K1A_STEP4(j)                                                                            \
    {                                                                                           \
        const uint32_t c0_ = lcls[K1A_BYTE(b0, j)], c1_ = lcls[K1A_BYTE(b1, j)];                \
        const uint32_t c2_ = lcls[K1A_BYTE(b2, j)], c3_ = lcls[K1A_BYTE(b3, j)];                \
        const uint32_t e0_ = s0 * NC + c0_, e1_ = s1 * NC + c1_, e2_ = s2 * NC + c2_, e3_ = s3 * NC + c3_; \
        const bool h0_ = s0 < lds_rows, h1_ = s1 < lds_rows, h2_ = s2 < lds_rows, h3_ = s3 < lds_rows; \
        const uint32_t l0_ = lrows[h0_ ? e0_ : 0], l1_ = lrows[h1_ ? e1_ : 0];                  \
        const uint32_t l2_ = lrows[h2_ ? e2_ : 0], l3_ = lrows[h3_ ? e3_ : 0];                  \
        uint32_t g0_ = 0, g1_ = 0, g2_ = 0, g3_ = 0;                                            \
        if (!h0_) g0_ = A.table16[e0_];                                                         \
        if (!h1_) g1_ = A.table16[e1_];                                                         \
        if (!h2_) g2_ = A.table16[e2_];                                                         \
        if (!h3_) g3_ = A.table16[e3_];                                                         \
        s0 = h0_ ? l0_ : g0_; s1 = h1_ ? l1_ : g1_; s2 = h2_ ? l2_ : g2_; s3 = h3_ ? l3_ : g3_; \
        hot += (uint32_t)h0_ + h1_ + h2_ + h3_; all += 4;                                       \
        if (emit && (s0 >= plain || s1 >= plain || s2 >= plain || s3 >= plain)) {               \
            const uint64_t p_ = (uint64_t)(first + i + (j)) + 1 - lead;                         \
            if (s0 >= plain) emit_state(Ad, K, A.walk_bfs[s0], p_);                             \
            if (s1 >= plain) emit_state(Ad, K, A.walk_bfs[s1], p_ + chunk);                     \
            if (s2 >= plain) emit_state(Ad, K, A.walk_bfs[s2], p_ + 2ull * chunk);              \
            if (s3 >= plain) emit_state(Ad, K, A.walk_bfs[s3], p_ + 3ull * chunk);              \
        }                                                                                       \
    }
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const int64_t first = (int64_t)(g * NCH * chunk) - (int64_t)warm; // index of chain 0's first byte
        for (uint32_t i0 = 0; i0 < warm + chunk; i0 += 32) {
            u32x4 b0, b1, b2, b3, n0, n1, n2, n3;
            uint32_t i = i0;
            K1A_LOAD(0, b0, n0) K1A_LOAD(1, b1, n1) K1A_LOAD(2, b2, n2) K1A_LOAD(3, b3, n3)
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if (half) { i = i0 + 16; b0 = n0; b1 = n1; b2 = n2; b3 = n3; }
                if (i >= warm + chunk) break; // (the walk length is a multiple of 16, not of 32)
                const bool emit = i >= warm; // (warm is a multiple of 16: a whole block is warm-up or not)
                const int64_t lo = first + i, hi = first + 3 * (int64_t)chunk + i + 16; // first / one past the last index touched
                if (lo >= (int64_t)lead && hi <= (int64_t)total) {
#pragma unroll
                    for (int j = 0; j < 16; j++) K1A_STEP4(j)
                } else if (hi > (int64_t)lead && lo < (int64_t)total) {
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        K1A_STEP(0, b0, s0, j) K1A_STEP(1, b1, s1, j) K1A_STEP(2, b2, s2, j) K1A_STEP(3, b3, s3, j)
                    }
                }
            }
        }
    }
#undef K1A_LOAD
#undef K1A_BYTE
#undef K1A_STEP
#undef K1A_STEP4
    if (lds_hits) { // (statistics launches only)
        for (int o = 32; o > 0; o >>= 1) { hot += __shfl_down(hot, o); all += __shfl_down(all, o); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&lds_hits[0], (unsigned long long)hot); atomicAdd(&lds_hits[1], (unsigned long long)all); }
    }
    __syncthreads();
    if (threadIdx.x == 0 && GK.block_counts) GK.block_counts[blockIdx.x] = *lcount;
}



// chunk of a chain: enough chains to fill the chip, warm-up overhead <= 1/8, a multiple of 16
static uint32_t walk16_chunk(uint32_t warm, uint64_t total, int n_cus) {
    uint64_t c = total / ((uint64_t)n_cus * 1024 * K1A_CHAINS);
    c = (c + 15) & ~15ull;
    if (c < 256) c = 256;
    if (c < (uint64_t)warm * 8) c = (uint64_t)warm * 8;
    if (c > (1u << 20)) c = 1u << 20;
    return (uint32_t)c;
}

uint32_t dfa_walk_hot_rows(uint32_t n_states, uint32_t stride2, size_t max_lds) {
    size_t budget = max_lds > 4096 + K1A_LDS_HEADER ? max_lds - 4096 - K1A_LDS_HEADER : 0;
    size_t rows = budget / ((size_t)2 << stride2);
    if (rows > n_states) rows = n_states;
    if (rows > 0x3FFF) rows = 0x3FFF;
    return (uint32_t)rows;
}

static uint32_t pick_chunk(uint32_t max_len, uint64_t len) {
    // warm-up overhead (max_len - 1) / chunk <= 1/8, at least 256 B, multiple of 64
    uint64_t c = 256;
    uint64_t need = (uint64_t)(max_len ? max_len - 1 : 0) * 8;
    if (c < need) c = (need + 63) & ~63ull;
    if (c > (1ull << 30)) c = 1ull << 30;
    (void)len;
    return (uint32_t)c;
}

uint32_t dfa_walk_grid(const DevAutomaton &A, uint64_t len, int n_cus) {
    if (A.table16) return (uint32_t)n_cus; // compact form: persistent, one workgroup per CU (region mode: n_cus regions)
    uint32_t chunk = pick_chunk(A.max_len, len);
    uint64_t nchunks = (len + chunk - 1) / chunk;
    uint64_t blocks = (nchunks + 1023) / 1024;
    if (blocks > (uint64_t)n_cus) blocks = n_cus;
    return blocks ? (uint32_t)blocks : 1;
}

hipError_t launch_dfa_walk(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G,
                           const Sink &K, const uint8_t *d_hay, uint64_t len, uint32_t grid,
                           size_t max_lds, hipStream_t st) {
    if (len == 0) return hipSuccess;
    const bool segmented = G.uniform_len != 0 || G.offsets != nullptr;
    if (A.table16 && !segmented) { // compact form
        static std::mutex mu16;
        static bool attr16[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(mu16);
            if (dev < 0 || dev >= 64 || !attr16[dev]) {
                hipError_t e = hipFuncSetAttribute((const void *)k1a_walk16,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(max_lds - 2048));
                if (e != hipSuccess) return e;
                if (dev >= 0 && dev < 64) attr16[dev] = true;
            }
        }
        const uint32_t NC = A.n_classes;
        size_t budget = max_lds > 4096 + K1A_LDS_HEADER ? max_lds - 4096 - K1A_LDS_HEADER : 0;
        uint32_t rows16 = (uint32_t)std::min<size_t>(budget / ((size_t)2 * NC), A.n_states);
        const size_t lds16 = K1A_LDS_HEADER + (((size_t)rows16 * NC * 2 + 15) / 16) * 16 + 16;
        const uint32_t lead = (uint32_t)((uintptr_t)d_hay & 15);
        const uint64_t total = lead + len;
        const uint32_t warm = ((A.max_len ? A.max_len - 1 : 0) + 15u) & ~15u;
        const uint32_t chunk16 = walk16_chunk(warm, total, (int)grid);
        static unsigned long long *stats = nullptr; // ACX_WALK_STATS=1: LDS hit fraction of the walk
        static const bool want_stats = std::getenv("ACX_WALK_STATS") != nullptr;
        if (want_stats && !stats) { (void)hipMalloc((void **)&stats, 16); }
        if (want_stats) (void)hipMemsetAsync(stats, 0, 16, st);
        hipLaunchKernelGGL(k1a_walk16, dim3(grid), dim3(1024), lds16, st, A, Ad, K, d_hay - lead, total, lead,
                           chunk16, warm, rows16, want_stats ? stats : nullptr);
        if (want_stats) {
            unsigned long long h[2] = {0, 0};
            (void)hipMemcpyAsync(h, stats, 16, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            std::fprintf(stderr, "acx: k1a_walk16 LDS rows %u of %u states, transitions from LDS %.4f (%llu of %llu)\n",
                         rows16, A.n_states, h[1] ? (double)h[0] / (double)h[1] : 0.0, h[0], h[1]);
        }
        return hipGetLastError();
    }
    uint32_t chunk = pick_chunk(A.max_len, len);
    uint32_t rows = A.hot_rows;
    uint32_t cap_rows = dfa_walk_hot_rows(A.n_states, A.stride2, max_lds);
    if (rows > cap_rows) rows = cap_rows;
    size_t lds = K1A_LDS_HEADER + (((size_t)rows << A.stride2) * 2 + 15) / 16 * 16;
    uint64_t blocks = grid;
    { // more than 64 KiB of dynamic LDS needs the attribute -- per DEVICE (the function is loaded per device)
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void *)k1a_dfa_walk,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(max_lds - 2048));
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(k1a_dfa_walk, dim3((uint32_t)blocks), dim3(1024), lds, st, A, Ad, G, K,
                       d_hay, len, chunk, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1b: LDS prefix prefilter -> exact prefix table
// ---------------------------------------------------------------------------
// Geometry: 1024-thread workgroups (16 waves), one per CU, persistent.  A wave
// owns tiles of K1B_ROWS rows; one row = 64 lanes x 16 B = 1 KiB read by ONE
// coalesced global_load_dwordx4 per lane; the wave's next tile is prefetched
// into registers while the current one is processed.  Lane l tests the 16
// positions that start inside its 16 bytes; the bytes it needs beyond them (up
// to 8) come from lane l+1 by cross-lane shuffle.  Levels (hashes: automaton.hpp):
//   L1  {X,Y} signature table in LDS (128 KiB): positions j, j+1 share ONE
//       ds_read_b64 addressed by the (Q-1)-gram at j+1 and test a two-bit
//       signature each; ~8 VALU + 0.5 LDS reads per haystack byte.  Survivors
//       (true Q-byte prefix hits + ~0.4 % collisions) are ballot-compacted into
//       the wave's queue Q1 (tile-relative u16 offsets).
//   L2  exact probe of the prefix table (HBM, L2-resident; keys of variable length -- the
//       first min(8, shortest pattern of the group) bytes -- filed under their first Q2
//       bytes, automaton.cpp), software-pipelined over tiles so that no wave waits on it:
//       tile t's survivors re-read their 16-byte window (stage A -> B, top of iteration
//       t+1), fetch their home slot (B -> C, t+2), compare (C, t+3).  A home slot holding
//       another key ends the search unless the slot's filter of displaced keys has the window's
//       bit (then the hit travels as HIT_RETRY and k_tile_main / k_walk_hits look it up); a
//       group with several keys answers with a redirect to the keys' own hash: those gathers
//       are waited for in place (rare unless many patterns share their first Q2 bytes).  A tile with more survivors
//       than Q1 holds (dense filters: 10^5 patterns) moves the pipeline on in the middle of
//       its compaction, a batch of 64 at a time (three batches in flight).
//   Output: prefix hits (position, candidate code, 16 haystack bytes).  Sparse mode
//       (SLOTS): into the hit slots of the hit's tile -- the wave is their only producer,
//       the count is a plain store; dense mode: appended to the wave's region.  The kernel
//       never walks the DFA and never compares pattern tails (k_tile_main / k_walk_hits do).
// All LDS is ONE static object with the L1 table at offset 0.
constexpr int K1B_ROWS = 4;
static_assert(K1B_ROWS * 1024 == (1 << TILE_BITS), "a K1b tile is a tile of the hit slots");
static_assert((DT_GROUP << TILE_BITS) <= (1u << 16), "group-relative key positions of the dense path's words");
constexpr uint32_t K1B_Q1CAP = 64;  // level-1 survivors settled per round (1 per lane)
constexpr uint32_t K1B_HB = 40;     // prefix hits a wave collects in LDS before one burst store
// STAGED (round 5; BIG sets whose level-1 table passes nearly every position): the survivors' windows never come from HBM.  Every survivor used to re-read the 8 bytes at its
// position one iteration later -- 13.5 M gathers per GiB on 10^5 patterns, lines that had left the XCD's L2 by then: 1.7 GB
// of the kernel's 4.2 GB of traffic, 180 of its 700 us (measured: the same kernel with those loads pointed at one hot
// line, profiles/r05/exp_cfg4_windows_*).  The bytes are in the wave's registers when level 1 flags them, but at a
// per-lane position: the row (1 KiB + the 8 bytes behind it) is staged in LDS, the compaction runs row by row, and a
// survivor's lane reads its window from the stage -- three ALIGNED dwords and two v_alignbyte (an unaligned
// ds_read_b64 works but is slow: tools/ubench_lds_align.hip, profiles/r05/exp_byte_table_*) -- into a second queue
// beside the offsets.  Room: the hit buffer shrinks to K1B_HB_BIG records, the redirect keys' Bloom filter is read from
// global memory (it is touched by the rare redirect entries only).
constexpr uint32_t K1B_HB_BIG = 4;
constexpr uint32_t K1B_STAGE_BYTES = 1024 + 16; // a row + the 8 bytes behind it (+ padding to 16)
struct K1bLds {
    uint32_t xy[FILTER_WORDS];
    uint16_t q1[16][K1B_Q1CAP];
    union {
        struct {
            uint4 hb[16][K1B_HB][2];
            uint32_t rbloom[REDIRECT_BLOOM_WORDS]; // Bloom filter of the keys behind redirect entries
        } n;
        struct {
            uint4 hb[16][K1B_HB_BIG][2];
            uint8_t stage[16][K1B_STAGE_BYTES] __attribute__((aligned(16))); // the row under compaction
            uint64_t q1w[16][K1B_Q1CAP];                                      // the queued survivors' windows
        } b;
    } u;
    uint32_t cb[16][16]; // sparse mode: hit counts of the wave's last tiles, stored 16 at a time
    uint32_t sxy[SHORT_XY_WORDS];          // SH: the short patterns' {X, Y} pair table by middle byte
};
static_assert(sizeof(K1bLds) <= 160 * 1024, "K1b LDS image exceeds 160 KiB");

__device__ __forceinline__ uint32_t hash_mul24(uint32_t a, uint32_t k) {
    return __umul24(a, k); // v_mul_u32_u24: uses bits [23:0] of each operand
}

// 8 bytes of the stream at position p (zero beyond the end)
__device__ __forceinline__ uint64_t load_window(const uint8_t *__restrict__ stream, uint64_t len,
                                                uint64_t p) {
    uint64_t w = 0;
    if (p + 8 <= len) {
        __builtin_memcpy(&w, stream + p, 8);
    } else {
        for (uint32_t k = 0; k < 8 && p + k < len; k++) w |= (uint64_t)stream[p + k] << (8 * k);
    }
    return w;
}

// 16 bytes of the stream at position p (zero beyond the end): ONE unaligned 16-byte load when
// they all exist (everywhere but in the last bytes of the stream)
__device__ __forceinline__ void load_window16(const uint8_t *__restrict__ stream, uint64_t len, uint64_t p,
                                              uint64_t *w0, uint64_t *w1) {
    if (p + 16 <= len) {
        u32x4 v;
        __builtin_memcpy(&v, stream + p, 16);
        *w0 = ((uint64_t)v.y << 32) | v.x;
        *w1 = ((uint64_t)v.w << 32) | v.z;
    } else {
        *w0 = load_window(stream, len, p);
        *w1 = load_window(stream, len, p + 8);
    }
}

__device__ __forceinline__ uint64_t low_bytes(uint64_t w, uint32_t n) { // the first n (1..8) bytes of w
    return n >= 8 ? w : (w & ((1ull << (8 * n)) - 1));
}

// Does the window (8 haystack bytes, little-endian) start with the entry's key?  (e.z != PREFIX_EMPTY)
__device__ __forceinline__ bool entry_matches(const uint4 e, uint64_t w0) {
    const uint32_t sh = (0u - (e.z << 3)) & 56u; // 64 - 8 * key length (1..8), mod 64
    return (((((uint64_t)e.y << 32) | e.x) ^ w0) << sh) == 0;
}

// One probe sequence of the prefix table, from the home slot of the window's first `salt` bytes:
// the code of the first FINAL entry (next = 0) whose key the window w0 starts with, or HIT_NONE.
// redirect != null (first-level walk): a matching REDIRECT entry ends the walk with *redirect = its
// `next` (the key length to look up instead).
__device__ __forceinline__ uint32_t prefix_walk(const uint32_t *__restrict__ ptab, uint32_t log2, uint32_t salt,
                                                uint64_t w0, uint32_t *redirect) {
    const uint32_t mask = (1u << log2) - 1;
    const uint32_t h = prefix_home_hash(low_bytes(w0, salt), salt);
    uint32_t idx = prefix_slot(h, log2);
    for (bool home = true;; home = false) {
        const uint4 e = *(const uint4 *)(ptab + (size_t)idx * 4);
        if (e.z == PREFIX_EMPTY) return HIT_NONE;
        if (entry_matches(e, w0)) {
            const uint32_t next = (e.z >> 4) & 15u;
            if (!next) return e.w;
            if (redirect) { *redirect = next; return HIT_NONE; }
        }
        if (home && !(e.z & prefix_more_bit(h))) return HIT_NONE; // no key with this hash lives elsewhere
        idx = (idx + 1) & mask;
    }
}

// The candidate code of the patterns whose prefix-table key the window w0 starts with, or
// HIT_NONE (automaton.cpp: a group's single key, or a redirect to the group's keys).
__device__ __forceinline__ uint32_t prefix_code(const uint32_t *__restrict__ ptab, uint32_t log2, uint32_t q2,
                                                uint64_t w0) {
    uint32_t next = 0;
    const uint32_t code = prefix_walk(ptab, log2, q2, w0, &next);
    return next ? prefix_walk(ptab, log2, next, w0, nullptr) : code;
}

// L3: does the pattern of `code` (pattern id | anchor shift << 24, automaton.hpp) occur with its ANCHOR at stream
// position p -- i.e. does it start at p - shift?  The first Q2 bytes at the anchor are known to match (the prefix
// table said so), the rest is compared with the haystack: ONE 16-byte load (pinfo: rank, length and the 12 bytes
// after the anchor's first Q2) settles a pattern of up to Q2 + 12 bytes behind its anchor against the window
// (w0, w1) = the 16 haystack bytes at p that travel with the hit -- independent loads, no dependent DFA walk, no
// second touch of the (by now cold) haystack.  A shifted pattern also has its head compared: its first `shift`
// bytes (phead) with the haystack bytes in front of the hit -- two more independent loads.  Returns the pattern's
// length (0: no occurrence); *rk = its tie-break rank, *start = p - shift.  room = bytes from p to the end of
// its haystack, back = bytes from the start of its haystack to p.
// (ANCH = false: the automaton files every pattern under its beginning -- the instantiation without anchors is
// the round-3 code, register for register)
template <bool ANCH>
__device__ __forceinline__ uint32_t verify_candidate(const DevAutomaton &A, const uint8_t *__restrict__ stream,
                                                     uint64_t len, uint64_t p, uint32_t code, uint64_t w0,
                                                     uint64_t w1, uint64_t room, uint64_t back, uint32_t *rk,
                                                     uint64_t *start, uint64_t *sw0 = nullptr, uint64_t *sw1 = nullptr) {
    const uint32_t q = A.filter_q2;
    const uint32_t pid = code & CODE_PID_MASK;
    const uint32_t sh = ANCH ? (code >> CODE_SHIFT_SHIFT) & CODE_SHIFT_MASK : 0u;
    const uint4 pi = A.pinfo[pid];
    // the bytes in front of the anchor: the pattern's and the haystack's, requested together with the info
    uint4 hd = make_uint4(0, 0, 0, 0);
    uint64_t hw0 = 0, hw1 = 0;
    if (ANCH && sh != 0 && sh <= back) {
        hd = A.phead[pid];
        load_window16(stream, len, p - sh, &hw0, &hw1);
    }
    // patterns that can reach beyond the carried window: where the pattern's bytes lie, requested
    // together with its info (uniform branch; the bytes themselves together with the haystack's)
    // (the carried window holds the haystack's first 16 bytes, pinfo the pattern's first filter_q2 + 12: a
    // pattern that is longer than EITHER has bytes to compare in place.  Round 2 tested the window only --
    // and let 14..16-byte patterns behind a short prefix through; the first fix of round 3 tested pinfo
    // only -- and let 17..20-byte patterns behind an 8-byte prefix through: found by tools/gpu_fuzz.py)
    const bool far = A.max_len > (A.filter_q2 + 12 < 16 ? A.filter_q2 + 12 : 16);
    const uint64_t po = far ? A.pat_off[pid] + sh : 0; // (the anchored suffix's bytes)
    *rk = pi.x & 0xFFFFFFu;
    uint32_t Lw = pi.x >> 24;
    if (Lw == 255) Lw = A.plen[pid];
    const uint32_t L = Lw - sh; // bytes from the anchor on
    *start = p - sh;
    if (sw0) { *sw0 = sh ? hw0 : w0; *sw1 = sh ? hw1 : w1; } // (the 16 haystack bytes at the START of the occurrence)
    bool ok = L <= room && sh <= back;
    if (ok && L > q) {
        // haystack bytes q.. from the carried window (16 - q of them), pattern bytes from pinfo
        const uint32_t have = 16 - q;                  // carried bytes beyond the prefix
        const uint32_t need = L - q < 12 ? L - q : 12; // bytes checkable against pinfo
        uint64_t h0 = q < 8 ? (q ? (w0 >> (8 * q)) | (w1 << (64 - 8 * q)) : w0) : (w1 >> (8 * (q - 8)));
        uint64_t h1 = q < 8 ? (q ? (w1 >> (8 * q)) : w1) : 0;
        uint64_t p0 = ((uint64_t)pi.z << 32) | pi.y, p1 = pi.w;
        uint32_t n0 = need < have ? need : have;       // bytes compared from the carried window
        uint64_t m0 = n0 >= 8 ? ~0ull : ((1ull << (8 * n0)) - 1);
        uint64_t m1 = n0 > 8 ? ((1ull << (8 * (n0 - 8))) - 1) : 0;
        ok = (((h0 ^ p0) & m0) | ((h1 ^ p1) & m1)) == 0;
        // whatever lies beyond the carried window / pinfo (long patterns): compare in place, four
        // 8-byte pieces of haystack and pattern per round, their loads in flight together
        for (uint32_t d = q + n0; far && ok && d < L; d += 32) {
            uint64_t a[4], b[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                a[k] = b[k] = 0;
                if (d + 8 * k < L) {
                    a[k] = load_window(stream, len, p + d + 8 * k);
                    __builtin_memcpy(&b[k], A.pat_blob + po + d + 8 * k, 8); // pat_blob is padded by 16 bytes
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (d + 8 * k >= L) continue;
                const uint32_t nbytes = L - (d + 8 * k) < 8 ? L - (d + 8 * k) : 8;
                const uint64_t m = nbytes >= 8 ? ~0ull : ((1ull << (8 * nbytes)) - 1);
                ok = ok && ((a[k] ^ b[k]) & m) == 0;
            }
        }
    }
    if (ANCH && sh) { // the head (sh <= SHIFT_MAX = 12 bytes)
        const uint64_t m0 = sh >= 8 ? ~0ull : ((1ull << (8 * sh)) - 1);
        const uint64_t m1 = sh > 8 ? ((1ull << (8 * (sh - 8))) - 1) : 0;
        ok = ok && ((((hw0 ^ (((uint64_t)hd.y << 32) | hd.x)) & m0) | ((hw1 ^ (uint64_t)hd.z) & m1)) == 0);
    }
    return ok ? Lw : 0;
}

__device__ __forceinline__ uint64_t occurrence_key(int key_mode, uint32_t rank_bits, uint64_t p, uint32_t L,
                                                   uint32_t pid, uint32_t rk) {
    return key_mode == 0   ? ((p + L) << rank_bits) | rk
           : key_mode == 1 ? (p << rank_bits) | pid
                           : (p << rank_bits) | rk;
}

// Dense path: one thread per prefix hit of K1b (region mode).  Hits live in the per-wave regions
// of the scan's sink (H); occurrences go to the occurrence sink (GK, one region per workgroup).
template <bool ANCH>
__global__ __launch_bounds__(256) void k_walk_hits(DevAutomaton A, Segments G, Sink H, uint32_t h_grid,
                                                   Sink GK, const uint8_t *__restrict__ stream,
                                                   uint64_t len) {
    __shared__ unsigned long long lcount;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    const BlockSink K = block_sink(GK, &lcount);
    for (uint32_t b = blockIdx.x; b < h_grid; b += gridDim.x) {
        uint64_t n = H.block_counts[b];
        if (n > H.region_cap) n = H.region_cap; // hits were dropped: the host sees the count and redoes the call
        const uint4 *rec = H.recs + (uint64_t)b * H.region_cap * 2;
        for (uint64_t i = threadIdx.x; i < n; i += 256) {
            const uint4 h = rec[2 * i], w = rec[2 * i + 1];
            const uint64_t p = ((uint64_t)h.y << 32) | h.x;
            const uint64_t w0 = ((uint64_t)w.y << 32) | w.x, w1 = ((uint64_t)w.w << 32) | w.z;
            uint64_t seg_lo, seg_hi;
            segment_bounds(G, len, p, &seg_lo, &seg_hi);
            const uint64_t room = seg_hi - p, back = p - seg_lo;
            uint32_t code = h.z;
            if (code == HIT_RETRY) code = prefix_code(A.ptab, A.ptab_log2, A.filter_q2, w0);
            const bool list = code != HIT_NONE && (code & HIT_LIST) != 0;
            const uint32_t li = code & ~HIT_LIST;
            const uint32_t nc = code == HIT_NONE ? 0 : list ? A.blist[li] : 1;
            for (uint32_t k = 0; k < nc; k++) {
                const uint32_t cand = list ? A.blist[li + 1 + k] : code; // pattern id | anchor shift << 24
                const uint32_t pid = cand & CODE_PID_MASK;
                uint32_t rk;
                uint64_t ps;
                const uint32_t L = verify_candidate<ANCH>(A, stream, len, p, cand, w0, w1, room, back, &rk, &ps);
                emit_key_agg(K, L != 0, occurrence_key(K.key_mode, A.rank_bits, ps, L, pid, rk), pid, L);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) GK.block_counts[blockIdx.x] = lcount;
}

__device__ __forceinline__ uint32_t lead_bytes_in_word(uint32_t w) {
    // continuation byte: bit7 = 1 and bit6 = 0
    uint32_t cont = w & 0x80808080u & ((~w) << 1);
    return 4 - __popc(cont);
}

// what K1b needs of the automaton (the full struct would sit in ~50 SGPRs for the whole kernel)
struct K1bTables {
    const uint32_t *filterA;
    const uint32_t *ptab;
    const uint32_t *rbloom;
    const uint32_t *pbits; // BIG: the bitmap in front of the prefix table
    uint32_t ptab_log2, filter_q2, min_len; // min_len: the shortest pattern IN THESE TABLES (DevAutomaton::k1b_min_len)
    uint8_t *cp_sub; // CP: lead (non-continuation) bytes of every 64 bytes of the stream (K3's sub counts)
    const uint32_t *short_xy, *short_codes; // SH: the side test for patterns of 1 and 2 bytes (automaton.hpp)
    uint32_t short_min;                     // SH: the shortest of them
};

// CP (str API, sparse mode, lead == 0): the scan also counts the UTF-8 lead bytes of every 64-byte
// stretch it streams (the code-point fix-up then needs no second pass over the haystack).
// BIG (Q = 5 only; pattern sets that saturate the level-1 table, ~10^5 patterns): EVERY position is
// put to both tests -- as the position after the gram in front of it (Y) and as the position in front
// of the gram behind it (X): 17 LDS reads per lane-row instead of 8, twice the level-1 VALU, but
// the false positives multiply (0.18 x 0.07 instead of their mean) and the level-2 gathers, which
// bound the kernel on such sets, go down by a factor of five.
// SH (pattern sets with patterns of 1 or 2 bytes; they are NOT in the level-1 / level-2 tables): every pair
// of positions is also put to the short patterns' pair table (one more 8-byte LDS read per pair, ~6 VALU); its
// survivors travel through the same queue and pipeline with a flag (bit 15 of the offset) and are settled
// against the exact codes (two 4-byte gathers: the 1-byte and the 2-byte pattern that may start there)
// instead of the prefix table.
template <int Q, bool SLOTS, bool CP, int BIGV, bool SH>
__global__ __launch_bounds__(1024) void k1b_prefilter(K1bTables A, Sink GK,
                                                      const uint8_t *__restrict__ hay,
                                                      uint64_t len, uint64_t lead) {
    // `hay` is 16-byte aligned; the first `lead` bytes (< 16) precede the real stream and are
    // never candidates.  Stream position = index - lead.
    __shared__ __attribute__((aligned(16))) K1bLds L;
    // BIGV: 0 -- the pair test; 1 (BIG) -- every position put to both tests, the survivors' windows gathered; 2 (BIG, STAGED) --
    // ... their windows captured from the row staged in LDS (K1bLds): sets whose table passes (nearly) every position
    constexpr bool BIG = BIGV != 0, STAGED = BIGV == 2;
    constexpr uint32_t FLOG = FILTER_ENTRIES_LOG2;
    constexpr uint32_t HB = STAGED ? K1B_HB_BIG : K1B_HB;
    // the wave index is wave-uniform: say so, and the tile index, its byte offset, the
    // interior test and most of the prefetch address arithmetic move from VALU to SALU
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint16_t *q1 = L.q1[wave];
    // region mode: every WAVE owns a region of the hit sink and keeps its cursor in an SGPR --
    // no atomic, no cross-lane traffic on the push path
    const uint32_t region = blockIdx.x * 16 + wave;
    uint4 *const hrec = SLOTS ? GK.hslots : GK.recs + (uint64_t)region * GK.region_cap * 2;
    const uint32_t hcap = SLOTS ? 0u : (uint32_t)(GK.region_cap < 0xFFFFFFFFull ? GK.region_cap : 0xFFFFFFFFull);
    uint32_t hcur = 0;
    {
        const uint4 *src = (const uint4 *)A.filterA;
        uint4 *dst = (uint4 *)L.xy;
        for (uint32_t i = threadIdx.x; i < sizeof(L.xy) / 16; i += blockDim.x) dst[i] = src[i];
        if (!STAGED) for (uint32_t i = threadIdx.x; i < REDIRECT_BLOOM_WORDS; i += blockDim.x) L.u.n.rbloom[i] = A.rbloom[i];
    }
    if (SH && threadIdx.x < SHORT_XY_WORDS) L.sxy[threadIdx.x] = A.short_xy[threadIdx.x];
    __syncthreads();

    constexpr int GB = Q - 1; // bytes of the shared gram
    constexpr uint32_t GMASK = GB >= 4 ? 0xFFFFFFFFu : ((1u << (8 * GB)) - 1u);
    const uint8_t *stream = hay + lead;
    const uint64_t total = lead + len;          // bytes addressable from `hay`
    const uint64_t total16 = (total + 15) & ~15ull;
    const uint64_t last_start = total >= A.min_len ? total - A.min_len : 0; // last index a pattern can start at
    const bool any_start = total >= lead + A.min_len;
    const uint64_t tile_bytes = (uint64_t)K1B_ROWS * 1024;
    const uint64_t ntiles = (total + tile_bytes - 1) / tile_bytes; // every tile gets its hit count
    const uint64_t gw = (uint64_t)blockIdx.x * 16 + wave;
    const uint64_t nw = (uint64_t)gridDim.x * 16;
    const uint32_t q2len = A.filter_q2;
    const uint64_t q2mask = q2len >= 8 ? ~0ull : ((1ull << (8 * q2len)) - 1);
    const uint32_t q2salt = q2len * 0x9E3779B1u; // prefix_key_hash(gram, q2len) = gram_hash2(gram) + q2salt
    const uint32_t ptab_log2 = A.ptab_log2;
    uint32_t q1c = 0; // wave-uniform queue fill
    constexpr uint32_t OFFMASK = SH ? 0x0FFFu : 0xFFFFu; // a queued offset (12 bits); SH: bit 15 = a survivor of the side test
    constexpr uint32_t SHFLAG = 0x8000u;

    // Prefix hits leave the wave THROUGH an LDS buffer that is stored in bursts of up to K1B_HB
    // records: on this architecture stores count in vmcnt like loads, so a store issued every
    // iteration makes the `s_waitcnt vmcnt(0)` in front of the next tile wait for HBM write latency
    // every iteration; a burst every few iterations does not (measured: 1-2 % of the kernel).
    // Word 3 of a record's first quad carries its destination slot (sparse mode).
    uint4 (*const hb)[2] = STAGED ? L.u.b.hb[wave] : L.u.n.hb[wave];
    uint32_t hbn = 0; // wave-uniform fill of the buffer
    auto hit_flush = [&]() __attribute__((always_inline)) {
        if (lane < hbn) {
            uint4 r0 = hb[lane][0];
            const uint4 r1 = hb[lane][1];
            if (SLOTS) {
                const uint32_t dst = r0.w;
                r0.w = 0;
                if (dst != 0xFFFFFFFFu) { hrec[2 * (uint64_t)dst] = r0; hrec[2 * (uint64_t)dst + 1] = r1; }
            } else {
                const uint32_t s = hcur + lane;
                if (s < hcap) { hrec[2 * (uint64_t)s] = r0; hrec[2 * (uint64_t)s + 1] = r1; }
            }
        }
        if (!SLOTS) hcur += hbn; // keeps counting past the capacity
        hbn = 0;
    };
    // the lanes with found == true push (p, code, 16 window bytes); sparse mode: into the slots
    // cnt, cnt + 1, ... of `tile` (cnt is wave-uniform).  Returns the number of hits pushed.
    auto hit_push = [&](bool found, uint64_t p, uint32_t code, uint64_t w0, uint64_t w1, uint64_t tile,
                        uint32_t cnt) __attribute__((always_inline)) -> uint32_t {
        const unsigned long long fm = __ballot(found);
        if (!fm) return 0;
        const uint32_t np = (uint32_t)__popcll(fm);
        const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0));
        uint32_t dst = 0;
        bool keep = found;
        if (SLOTS) {
            const uint32_t slot = cnt + rk;
            keep = found && slot < HIT_SLOTS;
            dst = (uint32_t)tile * HIT_SLOTS + slot;
        }
        // (sparse mode: a hit beyond the slots still takes its place in the buffer; its slot word says "nowhere")
        const uint4 r0 = make_uint4((uint32_t)p, (uint32_t)(p >> 32), code, keep ? dst : 0xFFFFFFFFu);
        const uint4 r1 = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
        if (SLOTS) {
            // more hits than the tile holds (a dense stretch of the input): those go to the call's overflow list -- the
            // tile's count says "overfull", k_tile_main leaves its group to the hot pipeline (device_types.hpp: control
            // block).  ONE atomic per wave and push, on the counter of the tile's list
            const unsigned long long om = __ballot(found && !keep);
            if (om) {
                uint32_t *ctl = GK.abort_flag;
                const uint32_t list = (uint32_t)tile & (OVF_LISTS - 1); // (wave-uniform: the hits of a push are one tile's)
                uint32_t *cnt = *(uint32_t *const *)(ctl + CTL_OVF_COUNTS) + list * OVF_COUNT_STRIDE;
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(om)) base = atomicAdd(cnt, (uint32_t)__popcll(om));
                base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(om));
                const uint32_t i = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0));
                const uint32_t cap = ctl[CTL_OVF_CAP];
                if (found && !keep) {
                    if (i < cap) {
                        uint4 *o = *(uint4 *const *)(ctl + CTL_OVF_RECS) + 2 * ((uint64_t)list * cap + i);
                        o[0] = make_uint4(r0.x, r0.y, r0.z, 0);
                        o[1] = r1;
                    } else {
                        ctl[CTL_OVF_LOST] = 1;
                    }
                }
            }
        }
        if (np > HB) { // more hits at once than the buffer holds: straight to HBM
            if (SLOTS) {
                if (keep) { hrec[2 * (uint64_t)dst] = make_uint4(r0.x, r0.y, r0.z, 0); hrec[2 * (uint64_t)dst + 1] = r1; }
            } else {
                hit_flush();
                if (found && hcur + rk < hcap) { hrec[2 * (uint64_t)(hcur + rk)] = r0; hrec[2 * (uint64_t)(hcur + rk) + 1] = r1; }
                hcur += np;
            }
            return np;
        }
        if (hbn + np > HB) hit_flush();
        if (found) { hb[hbn + rk][0] = r0; hb[hbn + rk][1] = r1; }
        hbn += np;
        return np;
    };
    // Tile loads are UNCONDITIONAL (addresses clamped to the last 16-byte block of
    // the stream) so that exactly five loads are in flight per prefetch: garbage
    // read for out-of-range blocks only ever feeds positions that are masked off.
    // nxtL = the 8 bytes that follow the tile (lane 63's look-ahead), read by all
    // lanes from one address.
    const uint64_t last_block = total16 - 16;
    // (ONE set of registers for the tile: the prefetch of the wave's next tile is issued when level 1
    // has finished with the rows and lands in the same registers -- round 2 kept a second set and copied)
    u32x4 nxt0, nxt1, nxt2, nxt3; // the tile's rows ("nxt": loaded one iteration ahead)
    uint2 nxtL;
    // PLAIN loads, not non-temporal ones: the survivors' windows are re-read one iteration later
    // and should still be in the XCD's L2 (measured: nt tile loads cost the kernel 15-20 %).
#define K1B_LOAD16(PTR) (*(const u32x4 *)(PTR))
#define K1B_ISSUE_ROW(DST, TILE, R)                                                              \
    {                                                                                            \
        uint64_t off_ = (TILE) * tile_bytes + (uint64_t)(R) * 1024 + lane * 16;                  \
        DST = K1B_LOAD16(hay + (off_ < last_block ? off_ : last_block));                         \
    }
    // The tile index is wave-uniform (SGPRs): a tile that lies wholly inside the stream -- all
    // but the last one -- is addressed as scalar base + lane * 16 + immediate row offset, no
    // VALU address arithmetic; both branches issue the same five loads.
#define K1B_ISSUE_TILE(TILE)                                                                     \
    {                                                                                            \
        const uint64_t tb_ = (TILE) * tile_bytes;                                                \
        if (tb_ + tile_bytes <= last_block) {                                                    \
            const uint8_t *tp_ = hay + tb_ + lane * 16;                                          \
            nxt0 = K1B_LOAD16(tp_); nxt1 = K1B_LOAD16(tp_ + 1024);                               \
            nxt2 = K1B_LOAD16(tp_ + 2048); nxt3 = K1B_LOAD16(tp_ + 3072);                        \
            nxtL = *(const uint2 *)(hay + tb_ + tile_bytes);                                     \
        } else {                                                                                 \
            K1B_ISSUE_ROW(nxt0, TILE, 0) K1B_ISSUE_ROW(nxt1, TILE, 1) K1B_ISSUE_ROW(nxt2, TILE, 2) \
            K1B_ISSUE_ROW(nxt3, TILE, 3)                                                         \
            uint64_t off_ = tb_ + tile_bytes;                                                    \
            nxtL = *(const uint2 *)(hay + (off_ < last_block ? off_ : last_block));              \
        }                                                                                        \
    }
    K1B_ISSUE_TILE(gw)

    // ---- level-2 pipeline (one entry per lane per stage).  A *batch* is up to 64 survivors of one
    // tile.  Q1 collects the survivors of the tile under compaction (stage A); advance() moves every
    // batch one stage on: the windows of the batch in Q1 are requested (A -> B), the windows
    // requested by the previous advance are hashed and their home slots requested (B -> C), the
    // slots requested by the advance before are compared and the hits pushed (C).  It runs once at
    // the top of every iteration, with the tile's remaining survivors (its LAST batch): each gather
    // then has a whole tile of level-1 work to land in, no wave waits.  A tile with more survivors
    // than Q1 holds (dense filters: 10^5 patterns, multi-byte prefixes) advances the pipeline in the
    // middle of its compaction as often as it takes: three batches are in flight, one gather
    // latency is waited for per advance.  Batches reach stage C in tile order, the batches of one
    // tile back to back, so ONE running count (cntC) numbers a tile's hit slots.
    uint32_t stB = 0, stC = 0;     // stage holds: 0 nothing, 1 a batch, 2 the last batch of its tile
    uint32_t tileB = 0, tileC = 0; // the tile of the batch (tiles < 2^32: 16 TiB of haystack)
    uint32_t nB = 0, nC = 0;       // survivors in the batch
    uint32_t cntC = 0;             // sparse mode: hits of tileC pushed so far
    uint32_t kC = 0;               // tiles of this wave that have left stage C
    uint64_t winB = 0, winB1 = 0, winC = 0, winC1 = 0;
    uint32_t offB = 0, offC = 0;
    uint4 entC = make_uint4(0, 0, PREFIX_EMPTY, 0);
    // BIG: one more stage between B and C.  The prefix table of such a set has outgrown the L2
    // (measured on 10^5 patterns: 54 M L2 misses per GiB, the kernel bound by them); the windows are
    // first put to the bitmap of the groups' first Q2 bytes (L2-resident), and only the ones it
    // passes (~1 in 5) fetch their slot.  Stage M holds the batch whose bitmap words are in flight.
    uint32_t stM = 0, tileM = 0, nM = 0, offM = 0, bitM = 0;
    uint64_t winM = 0;
    bool liveC = true; // BIG: the lane's survivor of the batch in C passed the bitmap
    auto advance = [&](uint32_t tileQ, uint32_t stQ) __attribute__((always_inline)) {
        // ---- stage C: compare the slots with their windows
        if (stC) {
            if (nC) {
                const bool shC = SH && lane < nC && (offC & SHFLAG) != 0; // the side test's survivor: entC = {code of byte 0, code of bytes 0-1}
                const bool act = lane < nC && (!BIG || liveC) && !shC && entC.z != PREFIX_EMPTY;
                const bool same = act && entry_matches(entC, winC);
                uint32_t code = entC.w;
                // a group with several keys: its home slot redirects to the keys' own hash.  These
                // dependent gathers are waited for in place (rare unless many patterns share their
                // first Q2 bytes; the other waves of the SIMD cover them)
                uint32_t next = same ? (entC.z >> 4) & 15u : 0u;
                if (next) { // (most such positions end here: the LDS Bloom filter of those keys says no)
                    const uint32_t bit = redirect_bloom_bit(prefix_home_hash(low_bytes(winC, next), next));
                    const uint32_t bw = STAGED ? A.rbloom[bit >> 5] : L.u.n.rbloom[bit >> 5]; // (STAGED: no room in LDS; these entries are rare)
                    if (!((bw >> (bit & 31)) & 1u)) { next = 0; code = HIT_NONE; }
                }
                if (next) code = prefix_walk(A.ptab, ptab_log2, next, winC, nullptr);
                // a home slot holding another key proves absence unless the slot's filter of
                // displaced keys has the window's bit: then the hit travels as HIT_RETRY and
                // k_tile_main / k_walk_hits look the window up (no dependent gathers here: walking
                // the probe sequence in place was measured at +15 % of the kernel on the headline set)
                // (the four bits of the window's hash that select the filter bit travel in the offset's spare bits)
                const bool retry = act && !same && ((entC.z >> (8 + (BIG ? prefix_more_index(gram_hash2(winC & q2mask) + q2salt) : offC >> 16))) & 1u);
                if constexpr (!SH) {
                    cntC += hit_push((same && code != HIT_NONE) || retry, (uint64_t)tileC * tile_bytes + (offC & 0xFFFFu) - lead,
                                     same ? code : HIT_RETRY, winC, winC1, tileC, cntC);
                } else {
                    const uint64_t posC = (uint64_t)tileC * tile_bytes + (offC & OFFMASK) - lead;
                    bool found = (same && code != HIT_NONE) || retry;
                    uint32_t hcode = same ? code : HIT_RETRY;
                    if (shC) { found = entC.x != SHORT_NONE; hcode = entC.x; }
                    cntC += hit_push(found, posC, hcode, winC, winC1, tileC, cntC);
                    // (a 1-byte AND a 2-byte pattern at one position: the second hit)
                    cntC += hit_push(shC && entC.y != SHORT_NONE, posC, entC.y, winC, winC1, tileC, cntC);
                }
            }
            if (stC == 2) { // the tile is complete
                if (SLOTS) { // its count: through LDS, 16 tiles of the wave per store
                    if (lane == 0) L.cb[wave][kC & 15] = cntC <= HIT_SLOTS ? cntC : HIT_SLOTS + 1; // (HIT_SLOTS + 1: overfull)
                    if ((kC & 15) == 15) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 16) GK.hcnt[gw * GK.cnt_iters + (kC - 15) + lane] = L.cb[wave][lane];
                        __builtin_amdgcn_wave_barrier();
                    }
                    kC++;
                }
                cntC = 0;
            }
        }
        if (BIG) {
            // ---- stage M -> C: the windows the bitmap passes fetch their home slots and their second half
            if (nM) {
                const bool shM = SH && (offM & SHFLAG) != 0;
                liveC = lane < nM && (shM || ((bitM >> (prefix_bitmap_bit(gram_hash2(winM & q2mask) + q2salt, ptab_log2) & 31)) & 1u));
                if (liveC) {
                    if (shM) {
                        entC.x = A.short_codes[(uint32_t)winM & 0xFFu];
                        entC.y = A.short_codes[256u + ((uint32_t)winM & 0xFFFFu)];
                    } else {
                        entC = *(const uint4 *)(A.ptab + (size_t)prefix_slot(gram_hash2(winM & q2mask) + q2salt, ptab_log2) * 4);
                    }
                    winC1 = load_window(stream, len, (uint64_t)tileM * tile_bytes + (offM & OFFMASK) - lead + 8);
                }
                offC = offM; winC = winM;
            }
            nC = nM; stC = stM; tileC = tileM;
            if constexpr (STAGED) {
                // ---- stage A -> M: the queued survivors' windows (captured from the staged row at their compaction) are
                // hashed and their bitmap words requested -- nothing of the haystack is read again
                if (q1c) {
                    if (lane < q1c) {
                        offM = q1[lane];
                        winM = L.u.b.q1w[wave][lane];
                        if (!(SH && (offM & SHFLAG)))
                            bitM = A.pbits[prefix_bitmap_bit(gram_hash2(winM & q2mask) + q2salt, ptab_log2) >> 5];
                    }
                }
                nM = q1c; stM = stQ; tileM = tileQ;
            } else {
                // ---- stage B -> M: hash the windows, fetch their bitmap words
                if (nB) {
                    if (lane < nB && !(SH && (offB & SHFLAG)))
                        bitM = A.pbits[prefix_bitmap_bit(gram_hash2(winB & q2mask) + q2salt, ptab_log2) >> 5];
                    offM = offB; winM = winB;
                }
                nM = nB; stM = stB; tileM = tileB;
            }
        } else {
            // ---- stage B -> C: hash the windows, fetch their home slots
            if (nB) {
                const uint32_t hB = gram_hash2(winB & q2mask) + q2salt;
                if (lane < nB) {
                    if (SH && (offB & SHFLAG)) { // the side test's survivor: the exact codes instead of the prefix table
                        entC.x = A.short_codes[(uint32_t)winB & 0xFFu];
                        entC.y = A.short_codes[256u + ((uint32_t)winB & 0xFFFFu)];
                    } else {
                        entC = *(const uint4 *)(A.ptab + (size_t)prefix_slot(hB, ptab_log2) * 4);
                    }
                }
                offC = offB | (prefix_more_index(hB) << 16); winC = winB; winC1 = winB1;
            }
            nC = nB; stC = stB; tileC = tileB;
        }
        if (BIG) {
            // ---- stage A -> B: fetch the first 8 bytes of the queued survivors' windows
            if (!STAGED && q1c) {
                if (lane < q1c) {
                    offB = q1[lane];
                    winB = load_window(stream, len, (uint64_t)tileQ * tile_bytes + (offB & OFFMASK) - lead);
                }
            }
        } else {
            // ---- stage A -> B: fetch the 16-byte windows of the queued survivors
            if (q1c) {
                // (wave-uniform: every window of a tile that ends 16 bytes inside the stream is one unaligned load)
                const bool inside = ((uint64_t)tileQ + 1) * tile_bytes + 16 <= total;
                if (lane < q1c) {
                    offB = q1[lane];
                    const uint64_t p_ = (uint64_t)tileQ * tile_bytes + (offB & OFFMASK) - lead;
                    if (inside) {
                        u32x4 w_;
                        __builtin_memcpy(&w_, stream + p_, 16);
                        winB = ((uint64_t)w_.y << 32) | w_.x;
                        winB1 = ((uint64_t)w_.w << 32) | w_.z;
                    } else {
                        load_window16(stream, len, p_, &winB, &winB1);
                    }
                }
            }
        }
        if (!STAGED) { nB = q1c; stB = stQ; tileB = tileQ; }
        q1c = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // (measured in round 5, same-box pairs: requesting the windows of a tile's last batch at the END of the tile's own
    // iteration -- half the distance between a tile's load and the re-read of its survivors' windows -- changes neither
    // the L2 misses (TCC_MISS 13.40 M -> 12.93 M per GiB on cfg2: the XCD's 4 MiB do not hold a tile for one iteration of
    // its 512 waves either) nor the kernel's time: the kernel is bound by its VALU instructions, 91 % busy)
    // (measured in round 5, same-box triples -- the kernel before / with / the same text without: batches of the survivors of
    // TWO consecutive tiles of a wave, the pipeline moving every second iteration -- the ablation builds put a quarter of
    // the kernel, 72 us on T, on level 2, whose instructions are per batch -- brought 1 % on T (285 against 289 us), -1 %
    // on U, +5 % on the mixed-length set, and cost cfg5 5 % and cfg4 9 % (their tiles fill a batch by themselves); a
    // first, more general version ran 9 % SLOWER than the kernel it replaced: three more wave-uniform words across
    // level 1 were v_readlanes in the loop, and the compiler parked the freshly requested windows in other registers
    // behind an s_waitcnt vmcnt(0) at the end of every advance().  Out of the tree; profiles/r05/exp_pair_*.)
    constexpr uint64_t DRAIN = BIG && !STAGED ? 4 : 3; // extra iterations that empty the pipeline (A -> B -> C; BIG: A -> B -> M -> C; STAGED: A -> M -> C)

    for (uint64_t tile = gw; tile < ntiles + DRAIN * nw; tile += nw) {
        // Everything loaded during the previous iteration (the tile prefetch and the level-2
        // windows / slots) is consumed from here on.  Passing the tile through an empty asm makes
        // the compiler wait for those loads HERE, not with a vmcnt(0) somewhere in the middle of level 1.
        asm volatile("" : "+v"(nxt0), "+v"(nxt1), "+v"(nxt2), "+v"(nxt3), "+v"(nxtL.x), "+v"(nxtL.y));
        // the previous tile's remaining survivors: its last batch (possibly empty)
        advance((uint32_t)(tile - nw), tile >= gw + nw && tile - nw < ntiles ? 2u : 0u);
        if (tile >= ntiles) continue;

        // ---- level 1 on this tile
        const uint64_t tbase = tile * tile_bytes;
        uint32_t mrow0 = 0, mrow1 = 0, mrow2 = 0, mrow3 = 0;
        // a register whose LOW byte is byte k of the lane's 24-byte view: an odd window,
        // a dword, or a dword shifted by 16 (only bits [4:0] are consumed)
#define K1B_SCHED_BARRIER __builtin_amdgcn_sched_barrier(0);
#define K1B_BYTE_REG(k) (((k) & 1) ? w_[(k)] : (((k) & 2) ? d_[(k) >> 2] >> 16 : d_[(k) >> 2]))
#define K1B_ROW(RI, VR, RX, RY, MROW)                                                            \
        {                                                                                        \
            /* look-ahead dwords: lane l+1's first two dwords by DPP wave_shl:1 (one VALU op   */\
            /* each, no LDS); lane 63 takes them from the next row's lane 0 (scalar)          */\
            uint32_t nx_ = __builtin_amdgcn_update_dpp(0u, VR.x, 0x130, 0xf, 0xf, true);         \
            uint32_t ny_ = __builtin_amdgcn_update_dpp(0u, VR.y, 0x130, 0xf, 0xf, true);         \
            const uint32_t rx_ = (RX), ry_ = (RY); /* evaluated by ALL lanes (readfirstlane) */  \
            uint32_t d4_ = lane == 63 ? rx_ : nx_, d5_ = lane == 63 ? ry_ : ny_;                 \
            uint32_t d_[6] = {VR.x, VR.y, VR.z, VR.w, d4_, d5_};                                 \
            /* w_[j] (odd j): the 4 bytes starting at byte j (4-gram of pair j-1; its low     */\
            /* byte is also the Y signature byte of pair j-5)                                 */\
            uint32_t w_[20];                                                                     \
            _Pragma("unroll") for (int j = 1; j < 20; j += 2)                                    \
                w_[j] = __builtin_amdgcn_alignbyte(d_[(j >> 2) + 1], d_[j >> 2], j & 3);         \
            uint32_t m_ = 0, mg_ = 0;                                                            \
            /* (round 5: the mask that aligns the address to the 8-byte entries is one instruction */\
            /* per pair, 7 % of the kernel's VALU, and the LDS does read 8 bytes at any address --  */\
            /* tools/ubench_lds_align.hip; a table addressed by byte, rows overlapping, filters as   */\
            /* well (simulated: 0.55 % against 0.53 % survivors on T) -- but an unaligned ds_read_b64 */\
            /* is SLOW: K1b 0.278 -> 0.92 ms, same box, profiles/r05/exp_byte_table_*: the mask stays) */\
            /* all eight table reads of the row are issued before the first test (the scheduler    */\
            /* otherwise keeps ONE read in flight: ds_read, s_waitcnt lgkmcnt(0), test, next read; */\
            /* measured: -2.6 us of 282 -- the other waves covered most of that latency already)    */\
            uint2 e8_[8];                                                                        \
            _Pragma("unroll") for (int j = 0; j < 16; j += 2) {                                  \
                const uint32_t W_ = w_[j + 1] & GMASK;                                           \
                const uint32_t H_ = hash_mul24(W_, HASH_K1) + W_;                                \
                e8_[j >> 1] = *(const uint2 *)((const uint8_t *)L.xy +                           \
                    ((H_ >> (32 - FLOG - 3)) & (((8u << FLOG) - 1) & ~7u)));                     \
            }                                                                                    \
            K1B_SCHED_BARRIER                                                                    \
            _Pragma("unroll") for (int j = 0; j < 16; j += 2) {                                  \
                const uint32_t W_ = w_[j + 1] & GMASK;                                           \
                const uint2 e_ = e8_[j >> 1];                                                    \
                /* byte j: low byte of d_[j / 4] (j % 4 == 0) or of d_ >> 16 (j % 4 == 2) */      \
                const uint32_t bx_ = K1B_BYTE_REG(j);                                            \
                const uint32_t by_ = K1B_BYTE_REG(j + Q);                                        \
                /* the gate both tests share, as 0 / ~0 (v_bfe_i32): its two low bits go into  */\
                /* a mask of their own, one AND per row instead of one per test                */\
                const uint32_t g_ = (uint32_t)__builtin_amdgcn_sbfe((int)e_.x, W_, 1);           \
                mg_ = __builtin_amdgcn_alignbit(g_, mg_, 2);                                     \
                m_ = __builtin_amdgcn_alignbit(e_.x >> (bx_ & 31), m_, 1); /* position j:   X, byte j   */ \
                m_ = __builtin_amdgcn_alignbit(e_.y >> (by_ & 31), m_, 1); /* position j+1: Y, byte j+Q */ \
            }                                                                                    \
            m_ = (m_ & mg_) >> 16;                                                               \
            if (!interior) { /* wave-uniform: a scalar branch */                                 \
                const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;                  \
                uint32_t keep_ = 0;                                                              \
                _Pragma("unroll") for (int j = 0; j < 16; j++)                                   \
                    if (any_start && p0_ + j >= lead && p0_ + j <= last_start) keep_ |= 1u << j; \
                m_ &= keep_;                                                                     \
            }                                                                                    \
            MROW = m_;                                                                           \
        }
        // BIG: position j passes iff the gram at j + 1 vouches for byte j in front of it (X) AND the
        // gram at j vouches for byte j + 4 behind it (Y); one table row per gram j = 0 .. 16
#define K1B_ROW_BIG(RI, VR, RX, RY, MROW)                                                        \
        {                                                                                        \
            uint32_t nx_ = __builtin_amdgcn_update_dpp(0u, VR.x, 0x130, 0xf, 0xf, true);         \
            uint32_t ny_ = __builtin_amdgcn_update_dpp(0u, VR.y, 0x130, 0xf, 0xf, true);         \
            const uint32_t rx_ = (RX), ry_ = (RY); /* evaluated by ALL lanes (readfirstlane) */  \
            uint32_t d4_ = lane == 63 ? rx_ : nx_, d5_ = lane == 63 ? ry_ : ny_;                 \
            uint32_t d_[6] = {VR.x, VR.y, VR.z, VR.w, d4_, d5_};                                 \
            uint32_t w_[21]; /* w_[j]: the 4 bytes starting at byte j */                         \
            _Pragma("unroll") for (int j = 0; j < 21; j++)                                       \
                w_[j] = (j & 3) ? __builtin_amdgcn_alignbyte(d_[(j >> 2) + 1], d_[j >> 2], j & 3) : d_[j >> 2]; \
            uint2 e_[17];                                                                        \
            _Pragma("unroll") for (int j = 0; j < 17; j++) {                                     \
                const uint32_t H_ = hash_mul24(w_[j], HASH_K1) + w_[j];                          \
                e_[j] = *(const uint2 *)((const uint8_t *)L.xy +                                 \
                    ((H_ >> (32 - FILTER_ENTRIES_LOG2 - 3)) & ((FILTER_WORDS * 4 - 1) & ~7u)));  \
            }                                                                                    \
            uint32_t m_ = 0;                                                                     \
            _Pragma("unroll") for (int j = 0; j < 16; j++) {                                     \
                const uint32_t tx_ = (e_[j + 1].x >> (w_[j] & 31)) & (e_[j + 1].x >> (w_[j + 1] & 31)); \
                const uint32_t ty_ = (e_[j].y >> (w_[j + 4] & 31)) & (e_[j].x >> (w_[j] & 31));  \
                m_ = __builtin_amdgcn_alignbit(tx_ & ty_, m_, 1);                                \
            }                                                                                    \
            m_ >>= 16;                                                                           \
            if (!interior) { /* wave-uniform: a scalar branch */                                 \
                const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;                  \
                uint32_t keep_ = 0;                                                              \
                _Pragma("unroll") for (int j = 0; j < 16; j++)                                   \
                    if (any_start && p0_ + j >= lead && p0_ + j <= last_start) keep_ |= 1u << j; \
                m_ &= keep_;                                                                     \
            }                                                                                    \
            MROW = m_;                                                                           \
        }
        // every position of an interior tile is a legal start: no per-row masking
        const bool interior = tbase >= lead && tbase + tile_bytes <= last_start;
        if (BIG && Q == 5) {
            K1B_ROW_BIG(0, nxt0, __builtin_amdgcn_readfirstlane(nxt1.x), __builtin_amdgcn_readfirstlane(nxt1.y), mrow0)
            K1B_ROW_BIG(1, nxt1, __builtin_amdgcn_readfirstlane(nxt2.x), __builtin_amdgcn_readfirstlane(nxt2.y), mrow1)
            K1B_ROW_BIG(2, nxt2, __builtin_amdgcn_readfirstlane(nxt3.x), __builtin_amdgcn_readfirstlane(nxt3.y), mrow2)
            K1B_ROW_BIG(3, nxt3, nxtL.x, nxtL.y, mrow3)
        } else {
            K1B_ROW(0, nxt0, __builtin_amdgcn_readfirstlane(nxt1.x), __builtin_amdgcn_readfirstlane(nxt1.y), mrow0)
            K1B_ROW(1, nxt1, __builtin_amdgcn_readfirstlane(nxt2.x), __builtin_amdgcn_readfirstlane(nxt2.y), mrow1)
            K1B_ROW(2, nxt2, __builtin_amdgcn_readfirstlane(nxt3.x), __builtin_amdgcn_readfirstlane(nxt3.y), mrow2)
            K1B_ROW(3, nxt3, nxtL.x, nxtL.y, mrow3)
        }
        // SH: the side test for patterns of 1 and 2 bytes.  Positions j, j+1 share the read of sxy[byte j+1]:
        // bit (byte j) of X, bit (byte j+2) of Y (the shifts take the low five bits of their operand by themselves)
        uint32_t srow0 = 0, srow1 = 0, srow2 = 0, srow3 = 0;
#define K1B_ROW_SHORT(RI, VR, RX, MROW)                                                          \
        {                                                                                        \
            const uint32_t nx_ = __builtin_amdgcn_update_dpp(0u, VR.x, 0x130, 0xf, 0xf, true);   \
            const uint32_t rx_ = (RX);                                                           \
            const uint32_t d_[5] = {VR.x, VR.y, VR.z, VR.w, lane == 63 ? rx_ : nx_};             \
            uint2 e8_[8];                                                                        \
            _Pragma("unroll") for (int j = 0; j < 16; j += 2) {                                  \
                const uint32_t mid_ = (d_[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu;         \
                e8_[j >> 1] = *(const uint2 *)((const uint8_t *)L.sxy + mid_ * 8);               \
            }                                                                                    \
            uint32_t m_ = 0;                                                                     \
            _Pragma("unroll") for (int j = 0; j < 16; j += 2) {                                  \
                const uint32_t bx_ = d_[j >> 2] >> (8 * (j & 3));                                \
                const uint32_t by_ = d_[(j + 2) >> 2] >> (8 * ((j + 2) & 3));                    \
                m_ = __builtin_amdgcn_alignbit(e8_[j >> 1].x >> (bx_ & 31), m_, 1);              \
                m_ = __builtin_amdgcn_alignbit(e8_[j >> 1].y >> (by_ & 31), m_, 1);              \
            }                                                                                    \
            m_ >>= 16;                                                                           \
            if (!interior) { /* wave-uniform: a scalar branch */                                 \
                const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;                  \
                uint32_t keep_ = 0;                                                              \
                _Pragma("unroll") for (int j = 0; j < 16; j++)                                   \
                    if (p0_ + j >= lead && p0_ + j + A.short_min <= total) keep_ |= 1u << j;     \
                m_ &= keep_;                                                                     \
            }                                                                                    \
            MROW = m_;                                                                           \
        }
        if (SH) {
            K1B_ROW_SHORT(0, nxt0, __builtin_amdgcn_readfirstlane(nxt1.x), srow0)
            K1B_ROW_SHORT(1, nxt1, __builtin_amdgcn_readfirstlane(nxt2.x), srow1)
            K1B_ROW_SHORT(2, nxt2, __builtin_amdgcn_readfirstlane(nxt3.x), srow2)
            K1B_ROW_SHORT(3, nxt3, nxtL.x, srow3)
        }
#undef K1B_ROW_SHORT
        if (CP) { // lead bytes of the lane's 16 bytes of every row: one count byte per lane, 64 per row (coalesced)
#define K1B_LEADS(RI, VR)                                                                        \
            {                                                                                    \
                uint32_t c_ = lead_bytes_in_word(VR.x) + lead_bytes_in_word(VR.y) +              \
                              lead_bytes_in_word(VR.z) + lead_bytes_in_word(VR.w);               \
                if (!interior) { /* bytes beyond the end of the stream do not count */            \
                    const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;              \
                    if (p0_ + 16 > total) {                                                      \
                        const uint32_t w_[4] = {VR.x, VR.y, VR.z, VR.w};                         \
                        c_ = 0;                                                                  \
                        for (uint32_t k_ = 0; k_ < 16; k_++)                                     \
                            if (p0_ + k_ < total && ((w_[k_ >> 2] >> (8 * (k_ & 3))) & 0xC0) != 0x80) c_++; \
                    }                                                                            \
                }                                                                                \
                A.cp_sub[(tile * 4 + (RI)) * 64 + lane] = (uint8_t)c_;                           \
            }
            K1B_LEADS(0, nxt0) K1B_LEADS(1, nxt1) K1B_LEADS(2, nxt2) K1B_LEADS(3, nxt3)
#undef K1B_LEADS
        }
        if constexpr (STAGED && Q == 5) {
            // ---- STAGED: row by row -- the row is staged in LDS (1 KiB + the 8 bytes behind it), the wave's NEXT tile's row is
            // requested into the registers the row has just left, and the row's survivors are ballot-compacted into Q1
            // with their windows, which their own lanes read from the stage (three aligned dwords, two v_alignbyte: the
            // K1bLds comment).  Nothing of the haystack is read a second time.
            uint8_t *const stg = L.u.b.stage[wave];
            uint64_t *const q1w = L.u.b.q1w[wave];
            const uint64_t ntb_ = (tile + nw) * tile_bytes;
            const bool nfast_ = ntb_ + tile_bytes <= last_block; // (wave-uniform: the next tile lies wholly inside the stream)
            const uint8_t *const ntp_ = hay + ntb_ + lane * 16;
            // (two loops: the first one -- the one that runs -- never moves the pipeline: with advance() inside it the whole
            // pipeline state is loop-carried, and the compiler copied it from register to register in EVERY round, behind
            // an s_waitcnt vmcnt(0) that waited for the row requested a moment before: K1b 0.70 -> 0.92 ms.  A row whose
            // survivors do not fit Q1 leaves the first loop and finishes in the second, which may move the pipeline.)
            auto take = [&](uint32_t &m, uint32_t r, uint32_t flag, const bool has, const unsigned long long act,
                            const uint32_t np) __attribute__((always_inline)) {
                const uint32_t j = (uint32_t)__builtin_ctz(m | 0x10000u);
                m &= m - 1;
                const uint32_t slot = q1c + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                // (only the lanes with a survivor touch the stage: the lanes' 16-byte stride puts every eighth lane on
                // the same LDS banks)
                if (has) {
                    const uint32_t a = lane * 16 + j; // the survivor's byte in the staged row
                    const uint32_t *sp = (const uint32_t *)(stg + (a & ~3u));
                    const uint32_t d0 = sp[0], d1 = sp[1], d2 = sp[2];
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
                    q1[slot] = (uint16_t)(((r << 10) + a) | flag); // offset in the tile
                    q1w[slot] = ((uint64_t)hi << 32) | lo;
                }
                q1c += np;
            };
            auto compact_row = [&](uint32_t m, uint32_t r, uint32_t flag) __attribute__((always_inline)) {
                bool full = false;
                while (true) {
                    const bool has = m != 0;
                    const unsigned long long act = __ballot(has);
                    if (!act) break;
                    const uint32_t np = __popcll(act);
                    if (q1c + np > K1B_Q1CAP) { full = true; break; }
                    take(m, r, flag, has, act, np);
                }
                if (full) { // (dense survivors: full batches move on in the middle of the row)
                    while (true) {
                        const bool has = m != 0;
                        const unsigned long long act = __ballot(has);
                        if (!act) break;
                        const uint32_t np = __popcll(act);
                        if (q1c + np > K1B_Q1CAP) advance((uint32_t)tile, 1u);
                        take(m, r, flag, has, act, np);
                    }
                }
            };
#define K1B_STAGE_ROW(R, VR, RX, RY, MROW, SROW)                                                  \
            {                                                                                        \
                const uint32_t rx_ = (RX), ry_ = (RY); /* the 8 bytes behind the row (wave-uniform) */ \
                __builtin_amdgcn_wave_barrier();                                                     \
                *(u32x4 *)(stg + lane * 16) = VR;                                                    \
                if (lane == 0) *(uint2 *)(stg + 1024) = make_uint2(rx_, ry_);                        \
                __builtin_amdgcn_wave_barrier();                                                     \
                if (nfast_) VR = K1B_LOAD16(ntp_ + (R) * 1024);                                      \
                else K1B_ISSUE_ROW(VR, tile + nw, R)                                                 \
                compact_row(MROW, R, 0u);                                                            \
                if (SH) compact_row(SROW, R, SHFLAG); /* the side test's survivors, flagged */       \
            }
            {
                // (the look-ahead of row r = the first 8 bytes of row r + 1 of THIS tile: read before that row's
                // registers are given to the next tile -- row r + 1 is staged after row r)
                K1B_STAGE_ROW(0, nxt0, __builtin_amdgcn_readfirstlane(nxt1.x), __builtin_amdgcn_readfirstlane(nxt1.y), mrow0, srow0)
                K1B_STAGE_ROW(1, nxt1, __builtin_amdgcn_readfirstlane(nxt2.x), __builtin_amdgcn_readfirstlane(nxt2.y), mrow1, srow1)
                K1B_STAGE_ROW(2, nxt2, __builtin_amdgcn_readfirstlane(nxt3.x), __builtin_amdgcn_readfirstlane(nxt3.y), mrow2, srow2)
                const uint32_t lx_ = nxtL.x, ly_ = nxtL.y;
                {
                    const uint64_t off_ = ntb_ + tile_bytes;
                    nxtL = *(const uint2 *)(hay + (nfast_ || off_ < last_block ? off_ : last_block));
                }
                K1B_STAGE_ROW(3, nxt3, lx_, ly_, mrow3, srow3)
            }
#undef K1B_STAGE_ROW
        } else {
        // Prefetch of the wave's next tile, issued LATE: the compaction below, level 2 at the top
        // of the next iteration and the three other waves of the SIMD cover its latency.  Measured
        // (round 1, T): issued before row 0: 310 us; after row 1: 299; after row 2: 293; here: 291;
        // no prefetch at all (loads at the top of the tile's own iteration): 308.
        K1B_ISSUE_TILE(tile + nw)
        // ---- ballot-compact the survivors of the tile into Q1, one per lane per round
        // (one 64-bit mask per lane; the round is branch-free: the lowest set bit by two v_ffbl, lanes
        // without a survivor compute along and do not store)
        if constexpr (!SH) {
            uint64_t m64 = (uint64_t)(mrow0 | (mrow1 << 16)) | ((uint64_t)(mrow2 | (mrow3 << 16)) << 32);
            while (true) {
                const bool has = m64 != 0;
                const unsigned long long act = __ballot(has);
                if (!act) break;
                const uint32_t np = __popcll(act);
                if (q1c + np > K1B_Q1CAP) advance((uint32_t)tile, 1u); // dense survivors: a full batch moves on now
                const uint32_t pos = (uint32_t)__builtin_ctzll(m64 | (1ull << 63));
                m64 &= m64 - 1;
                const uint32_t slot = q1c + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                if (has) q1[slot] = (uint16_t)(((pos >> 4) << 10) + lane * 16 + (pos & 15)); // offset in the tile
                q1c += np;
            }
        } else {
            auto compact = [&](uint64_t m64, uint32_t flag) __attribute__((always_inline)) {
                while (true) {
                    const bool has = m64 != 0;
                    const unsigned long long act = __ballot(has);
                    if (!act) break;
                    const uint32_t np = __popcll(act);
                    if (q1c + np > K1B_Q1CAP) advance((uint32_t)tile, 1u); // dense survivors: a full batch moves on now
                    const uint32_t pos = (uint32_t)__builtin_ctzll(m64 | (1ull << 63));
                    m64 &= m64 - 1;
                    const uint32_t slot = q1c + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                    if (has) q1[slot] = (uint16_t)((((pos >> 4) << 10) + lane * 16 + (pos & 15)) | flag); // offset in the tile
                    q1c += np;
                }
            };
            compact((uint64_t)(mrow0 | (mrow1 << 16)) | ((uint64_t)(mrow2 | (mrow3 << 16)) << 32), 0u);
            // SH: the side test's survivors, flagged (a position may be queued twice: once for each kind of table)
            compact((uint64_t)(srow0 | (srow1 << 16)) | ((uint64_t)(srow2 | (srow3 << 16)) << 32), SHFLAG);
        }
        }
        __builtin_amdgcn_wave_barrier();
        // Q1 now holds this tile's (remaining) survivors: stage A
    }
    hit_flush();
    if (SLOTS && (kC & 15)) { // the counts of the wave's last tiles
        __builtin_amdgcn_wave_barrier();
        if (lane < (kC & 15)) GK.hcnt[gw * GK.cnt_iters + (kC & ~15u) + lane] = L.cb[wave][lane];
    }
    if (!SLOTS && lane == 0) GK.block_counts[region] = hcur;
#undef K1B_LOAD16
#undef K1B_ISSUE_ROW
#undef K1B_ISSUE_TILE
#undef K1B_ROW
#undef K1B_ROW_BIG
#undef K1B_BYTE_REG
}

size_t prefilter_lds_bytes() { return sizeof(K1bLds); }

uint32_t prefilter_grid(const uint8_t *d_hay, uint64_t len, int n_cus) {
    uint64_t ntiles = prefilter_tiles(d_hay, len);
    uint64_t blocks = (ntiles + 15) / 16;
    if (blocks > (uint64_t)n_cus) blocks = n_cus;
    return blocks ? (uint32_t)blocks : 1;
}

uint32_t walk_hits_grid(uint32_t hit_regions) { return hit_regions < 4096 ? hit_regions : 4096; }

hipError_t launch_walk_hits(const DevAutomaton &A, const Segments &G, const Sink &hits, uint32_t hit_grid,
                            const Sink &occ, uint32_t occ_grid, const uint8_t *d_hay, uint64_t len,
                            hipStream_t st) {
    if (A.max_shift) hipLaunchKernelGGL(k_walk_hits<true>, dim3(occ_grid), dim3(256), 0, st, A, G, hits, hit_grid, occ, d_hay, len);
    else hipLaunchKernelGGL(k_walk_hits<false>, dim3(occ_grid), dim3(256), 0, st, A, G, hits, hit_grid, occ, d_hay, len);
    return hipGetLastError();
}

uint32_t prefilter_hit_regions(uint32_t grid) { return grid * 16; } // one per wave

uint64_t prefilter_tiles(const uint8_t *d_hay, uint64_t len) {
    const uint64_t total = ((uintptr_t)d_hay & 15) + len;
    return (total + (uint64_t)K1B_ROWS * 1024 - 1) / ((uint64_t)K1B_ROWS * 1024);
}

// K.hslots != null: sparse mode (hit slots + counts); else region mode (per-wave regions)
hipError_t launch_prefilter(const DevAutomaton &A, const Sink &K, const uint8_t *d_hay, uint64_t len,
                            uint32_t grid, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop,
                            uint8_t *cp_sub) {
    if (len == 0 || A.filter_q == 0) return hipSuccess;
    uint64_t lead = (uintptr_t)d_hay & 15;
    const uint8_t *base = d_hay - lead;
    dim3 g(grid), b(1024);
    const K1bTables T{A.filterA, A.ptab, A.rbloom, A.pbits, A.ptab_log2, A.filter_q2, A.k1b_min_len, cp_sub,
                      A.short_xy, A.short_codes, A.short_min_len};
    const bool sh = A.short_min_len != 0; // the set has patterns of 1 or 2 bytes: the side test runs too
    if (cp_sub && (lead != 0 || !K.hslots)) return hipErrorInvalidValue;
    // the events (measurement only) ride on the dispatch itself: no barrier packets, no gaps
#define ACX_K1B_LAUNCH(Q, S, C, B, H)                                                                      \
    hipExtLaunchKernelGGL((k1b_prefilter<Q, S, C, B, H>), g, b, 0, st, ev_start, ev_stop, 0, T, K, base, len, lead)
#define ACX_K1B(Q, B, H)                                                                                   \
    if (cp_sub) ACX_K1B_LAUNCH(Q, true, true, B, H);                                                       \
    else if (K.hslots) ACX_K1B_LAUNCH(Q, true, false, B, H);                                               \
    else ACX_K1B_LAUNCH(Q, false, false, B, H)
#define ACX_K1B_SH(Q, B)                                                                                   \
    if (sh) { ACX_K1B(Q, B, true); } else { ACX_K1B(Q, B, false); }
    switch (A.filter_q) {
    // (Q = 1, 2: only with ACX_NO_SHORT_SPLIT -- the split keeps such patterns out of these tables)
    case 1: ACX_K1B(1, 0, false); break;
    case 2: ACX_K1B(2, 0, false); break;
    case 3: ACX_K1B_SH(3, 0) break;
    case 4: ACX_K1B_SH(4, 0) break;
    default:
        // (saturated level-1 table: both tests for every position; a table that passes nearly every position -- 10^6 patterns --:
        // ... and the survivors' windows from the staged row, not from HBM)
        if (A.filter_big >= 2) { ACX_K1B_SH(5, 2) }
        else if (A.filter_big) { ACX_K1B_SH(5, 1) }
        else { ACX_K1B_SH(5, 0) }
        break;
    }
#undef ACX_K1B_SH
#undef ACX_K1B_LAUNCH
#undef ACX_K1B
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1a, failureless form: every position walks the trie (automata of at most 32 byte classes)
// ---------------------------------------------------------------------------
// The goto function of the Aho-Corasick automaton, walked from EVERY haystack position (no failure
// links: an occurrence is found by the walk that starts at its first byte -- the form of the DFA walk
// that has no dependent chain across positions).  Two kernels:
//   k1a_scan   streams the haystack exactly like K1b (one coalesced 16-byte load per lane and row, the
//              wave's next tile prefetched into the registers level 1 has finished with) and settles the
//              first FOUR levels of every walk in LDS: the SYMBOLS (low five bits) of bytes j .. j+2
//              index t3b (132 KiB: which symbols a fourth byte can have on a trie path through such a
//              triple, 0 when there is none), bit symbol(byte j+3) says whether the walk can reach depth
//              4 -- exact for an alphabet whose bytes differ in their low five bits (a-z + space), a
//              superset otherwise (the stages below work on the exact byte classes).  ~2 % of the
//              positions of a 10^4-pattern set over text pass; they are ballot-compacted into a per-wave
//              queue and move through a software pipeline, one stage per step like K1b's level 2 (every
//              gather has a tile of level-1 work to land in): window gather (8 bytes, L2) -> the depth-3
//              node's record by the class triple (t3r) -> the depth-4 node's record (grec) -> does the
//              walk go on beyond depth 4, or does a pattern end there?  ~5 % of the survivors: they leave
//              as 32-byte items {position, node, depth, the window} in per-wave regions.
//   k1a_walk   one thread per item: the levels below from the trie records (grec, 16 B per state, L2) and
//              the window; every pattern that ends on the way is an occurrence (start, pattern, length):
//              into the hit slots of its tile (sparse output) or the occurrence regions (dense output),
//              exactly like the chunked walk.
// Patterns of at most 3 bytes: t3b holds ~0 for every triple with such a pattern on its path, and the
// walk of such a position starts from the root (k1a_walk).  The last positions of a haystack (fewer than
// 4 bytes left) survive unconditionally.
struct K1aLds {
    uint32_t t3b[K1A_T3B_WORDS];
    uint8_t cls[256]; // class << 2
    uint16_t q1[16][64];
    uint32_t cb[16][16]; // hit counts of the wave's last tiles, stored 16 at a time
};
static_assert(sizeof(K1aLds) <= 160 * 1024, "K1a LDS image exceeds 160 KiB");
constexpr uint32_t ITEM_ROOT = 0x80000000u; // item.w: walk from the root (a short pattern, or the end of the haystack)

struct DeepSink {
    uint4 *recs;       // regions * cap items of two quads: {position lo, position hi, node, depth | ITEM_ROOT}
                       // {window lo, hi, the node's record: children bitmap, first child | GREC_OWN}
    uint64_t *counts;  // one per region (keeps counting past cap)
    uint64_t cap;
    uint32_t regions;
};

__global__ __launch_bounds__(1024) void k1a_scan(const uint32_t *__restrict__ t3b, const uint8_t *__restrict__ classes,
                                                  const uint2 *__restrict__ t3r, const uint4 *__restrict__ grec,
                                                  DeepSink D, Sink GK, Segments G, const uint8_t *__restrict__ hay,
                                                  uint64_t len, uint64_t lead, uint32_t min_len, uint32_t NC) {
    __shared__ __attribute__((aligned(16))) K1aLds L;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint16_t *q1 = L.q1[wave];
    {
        const uint4 *src = (const uint4 *)t3b;
        uint4 *dst = (uint4 *)L.t3b;
        for (uint32_t i = threadIdx.x; i < K1A_T3B_WORDS / 4; i += blockDim.x) dst[i] = src[i];
        if (threadIdx.x < 256) L.cls[threadIdx.x] = (uint8_t)(classes[threadIdx.x] << 2);
    }
    __syncthreads();
    const uint8_t *stream = hay + lead;
    const uint64_t total = lead + len, total16 = (total + 15) & ~15ull;
    const uint64_t last_start = total >= min_len ? total - min_len : 0; // last index a pattern can start at
    const bool any_start = total >= lead + min_len;
    const uint64_t tile_bytes = 4096;
    const uint64_t ntiles = (total + tile_bytes - 1) / tile_bytes;
    const uint64_t gw = (uint64_t)blockIdx.x * 16 + wave, nw = (uint64_t)gridDim.x * 16;
    const uint32_t region = blockIdx.x * 16 + wave;
    uint4 *const drec = D.recs + (uint64_t)region * D.cap * 2;
    const uint32_t dcap = (uint32_t)(D.cap < 0xFFFFFFFFull ? D.cap : 0xFFFFFFFFull);
    uint32_t dcur = 0, q1c = 0;
    const uint64_t last_block = total16 - 16;
    u32x4 v0, v1, v2, v3; // the tile's rows: loaded one iteration ahead, into the registers level 1 is done with
    uint32_t vL;
#define K1A_ISSUE_ROW(DST, TILE, R)                                                              \
    {                                                                                            \
        uint64_t off_ = (TILE) * tile_bytes + (uint64_t)(R) * 1024 + lane * 16;                  \
        DST = *(const u32x4 *)(hay + (off_ < last_block ? off_ : last_block));                   \
    }
#define K1A_ISSUE_TILE(TILE)                                                                     \
    {                                                                                            \
        const uint64_t tb_ = (TILE) * tile_bytes;                                                \
        if (tb_ + tile_bytes <= last_block) {                                                    \
            const uint8_t *tp_ = hay + tb_ + lane * 16;                                          \
            v0 = *(const u32x4 *)(tp_); v1 = *(const u32x4 *)(tp_ + 1024);                       \
            v2 = *(const u32x4 *)(tp_ + 2048); v3 = *(const u32x4 *)(tp_ + 3072);               \
            vL = *(const uint32_t *)(hay + tb_ + tile_bytes);                                    \
        } else {                                                                                 \
            K1A_ISSUE_ROW(v0, TILE, 0) K1A_ISSUE_ROW(v1, TILE, 1) K1A_ISSUE_ROW(v2, TILE, 2)     \
            K1A_ISSUE_ROW(v3, TILE, 3)                                                           \
            uint64_t off_ = tb_ + tile_bytes;                                                    \
            vL = *(const uint32_t *)(hay + (off_ < last_block ? off_ : last_block));             \
        }                                                                                        \
    }
    K1A_ISSUE_TILE(gw)
    // ---- survivors: Q1 (offsets of the tile under compaction) -> W (windows in flight) -> T (depth-3
    // records in flight) -> G (depth-4 records in flight) -> hits / items.  One batch of up to 64 per
    // stage; a stage holds: 0 nothing, 1 a batch, 2 the last batch of its tile (as in K1b: batches reach
    // stage G in tile order, the batches of a tile back to back, ONE running count numbers its hit slots).
    // Stage G settles what it can by itself: the record of a node whose subtree is ONE pattern's remaining
    // bytes (GREC_TAIL: up to 8 of them, compared with the 16-byte window) -- on a set of random patterns
    // practically every node of depth 4 -- is an occurrence or nothing, written straight into the hit
    // slots of the tile (this wave is their only producer so far; k1a_walk appends with atomics later).
    const bool slots = GK.hslots != nullptr; // sparse output (else: everything becomes an item of the walk)
    uint32_t stW = 0, stT = 0, stG = 0, tileW = 0, tileT = 0, tileG = 0;
    uint32_t nW = 0, nT = 0, nG = 0;
    uint32_t cntG = 0, kG = 0; // hits of tileG pushed so far; tiles of this wave that have left stage G
    uint64_t posW = 0, winW = 0, winW1 = 0, posT = 0, winT = 0, winT1 = 0, posG = 0, winG = 0, winG1 = 0;
    uint2 eT = make_uint2(0, 0);
    uint32_t c34T = 0, nodeG = 0, c4G = 0;
    uint4 rG = make_uint4(0, 0, 0, 0);
    bool rootG = false, liveG = false;
    auto advance = [&](uint32_t tileQ, uint32_t stQ) __attribute__((always_inline)) {
        // ---- stage G: the depth-4 node's record has landed
        if (stG) {
            if (nG) {
                const bool in = lane < nG;
                const bool tail = in && !rootG && liveG && (rG.y & GREC_TAIL);
                // a tail: the pattern's remaining bytes against haystack bytes 4 .. 4 + n - 1 of the window
                const uint32_t tn = (rG.y >> 24) & 15u;
                const uint64_t tb = ((uint64_t)rG.w << 32) | rG.x;
                const uint64_t hb = (winG >> 32) | (winG1 << 32);
                const uint32_t sh = (0u - (tn << 3)) & 63u;
                bool hit = tail && (tn == 0 || ((tb ^ hb) << sh) == 0);
                if (hit) hit = 4 + tn <= segment_end(G, len, posG) - posG; // (rare lanes: the end of this haystack)
                // (dense output -- no hit slots: the occurrence goes the walk's way, into its region)
                const unsigned long long hm = slots ? __ballot(hit) : 0ull;
                if (hm) {
                    const uint32_t slot = cntG + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0));
                    if (hit) {
                        if (slot < HIT_SLOTS)
                            GK.hslots[((uint64_t)tileG * HIT_SLOTS + slot) * 2] =
                                make_uint4((uint32_t)posG, (uint32_t)(posG >> 32), HIT_VERIFIED | rG.z, 4 + tn);
                        else
                            *GK.abort_flag = 1; // more hits than the tile holds: dense input
                    }
                    cntG += (uint32_t)__popcll(hm);
                }
                // everything else the walk looks at: from the root (short patterns, the end of the
                // stream), or from this node when a pattern ends here or the next byte has a child
                const bool go = in && (rootG || (!slots && hit) ||
                                       (liveG && !(rG.y & GREC_TAIL) && ((rG.y & GREC_OWN) || ((rG.x >> c4G) & 1u))));
                const unsigned long long fm = __ballot(go);
                if (fm) {
                    const uint32_t slot = dcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0));
                    if (go && slot < dcap) {
                        drec[2 * (uint64_t)slot] = make_uint4((uint32_t)posG, (uint32_t)(posG >> 32), rootG ? 0u : nodeG,
                                                              rootG ? ITEM_ROOT : 4u);
                        drec[2 * (uint64_t)slot + 1] = make_uint4((uint32_t)winG, (uint32_t)(winG >> 32), rG.x, rG.y);
                    }
                    dcur += (uint32_t)__popcll(fm); // keeps counting past the capacity
                }
            }
            if (slots && stG == 2) { // the tile is complete: its count, through LDS, 16 tiles of the wave per store
                if (lane == 0) L.cb[wave][kG & 15] = cntG < HIT_SLOTS ? cntG : HIT_SLOTS;
                if ((kG & 15) == 15) {
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 16) GK.hcnt[gw * GK.cnt_iters + (kG - 15) + lane] = L.cb[wave][lane];
                    __builtin_amdgcn_wave_barrier();
                }
                kG++;
                cntG = 0;
            }
        }
        // ---- stage T -> G: the depth-3 record by the class triple has landed: the depth-4 node, its record
        if (nT) {
            const uint32_t c3 = c34T & 31u;
            rootG = (eT.y & T3R_SHORT) != 0 || posT + 5 > len; // (a short pattern on the path, or the stream ends: from the root)
            liveG = !rootG && ((eT.x >> c3) & 1u);
            nodeG = (eT.y & ID_MASK) + __popc(eT.x & ((1u << c3) - 1u));
            if (lane < nT && liveG) rG = grec[nodeG];
            posG = posT; winG = winT; winG1 = winT1; c4G = c34T >> 5;
        }
        nG = nT; stG = stT; tileG = tileT;
        // ---- stage W -> T: the window has landed: classes of bytes 0 .. 4, the depth-3 record
        if (nW) {
            const uint32_t lo = (uint32_t)winW, hi = (uint32_t)(winW >> 32);
            const uint32_t c0 = L.cls[lo & 0xFF] >> 2, c1 = L.cls[(lo >> 8) & 0xFF] >> 2, c2 = L.cls[(lo >> 16) & 0xFF] >> 2;
            c34T = (L.cls[lo >> 24] >> 2) | ((uint32_t)(L.cls[hi & 0xFF] >> 2) << 5);
            if (lane < nW) eT = t3r[(c0 * NC + c1) * NC + c2];
            posT = posW; winT = winW; winT1 = winW1;
        }
        nT = nW; stT = stW; tileT = tileW;
        // ---- stage Q -> W: the 16-byte windows of the queued survivors
        if (q1c) {
            // (wave-uniform: every window of a tile that ends 16 bytes inside the stream is one unaligned load)
            const bool inside = ((uint64_t)tileQ + 1) * tile_bytes + 16 <= total;
            if (lane < q1c) {
                posW = (uint64_t)tileQ * tile_bytes + q1[lane] - lead;
                if (inside) {
                    u32x4 w_;
                    __builtin_memcpy(&w_, stream + posW, 16);
                    winW = ((uint64_t)w_.y << 32) | w_.x;
                    winW1 = ((uint64_t)w_.w << 32) | w_.z;
                } else {
                    load_window16(stream, len, posW, &winW, &winW1);
                }
            }
        }
        nW = q1c; stW = stQ; tileW = tileQ;
        q1c = 0;
        __builtin_amdgcn_wave_barrier();
    };
    for (uint64_t tile = gw; tile < ntiles + 4 * nw; tile += nw) {
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(vL));
        // the previous tile's remaining survivors: its last batch (possibly empty)
        advance((uint32_t)(tile - nw), tile >= gw + nw && tile - nw < ntiles ? 2u : 0u);
        if (tile >= ntiles) continue;
        const uint64_t tbase = tile * tile_bytes;
        // every position of an interior tile is a legal start with at least 4 bytes behind it
        const bool interior = tbase >= lead && tbase + tile_bytes + 4 <= last_start;
        uint32_t mrow0 = 0, mrow1 = 0, mrow2 = 0, mrow3 = 0;
        // one row: 16 positions per lane; the walk of position j needs bytes j .. j+3: three bytes of
        // the lane behind by DPP (lane 63: the next row's first bytes, scalar).  s2 = symbol << 2 (four
        // of them per AND: (dword << 2) & 0x7C7C7C7C), y2 = (s2[k] << 5) | s2[k+1]; the entry of position
        // j lies at byte y2[j] * 33 + s2[j+2] (one v_mad_u32_u24), the bit is byte j+3's low five bits
        // (the shift takes them by itself)
#define K1A_BYTE(R, k) ((R[(k) >> 2] >> (8 * ((k) & 3))) & 0xFFu)
#define K1A_ROW(RI, VR, RX, MROW)                                                                \
        {                                                                                        \
            const uint32_t nx_ = __builtin_amdgcn_update_dpp(0u, VR.x, 0x130, 0xf, 0xf, true);   \
            const uint32_t rx_ = (RX);                                                           \
            const uint32_t d_[5] = {VR.x, VR.y, VR.z, VR.w, lane == 63 ? rx_ : nx_};             \
            uint32_t t_[5];                                                                      \
            _Pragma("unroll") for (int k = 0; k < 5; k++) t_[k] = (d_[k] << 2) & 0x7C7C7C7Cu;    \
            uint32_t s2_[19], y2_[17];                                                           \
            _Pragma("unroll") for (int k = 0; k < 19; k++) s2_[k] = K1A_BYTE(t_, k);             \
            _Pragma("unroll") for (int k = 0; k < 17; k++) y2_[k] = (s2_[k] << 5) | s2_[k + 1];  \
            uint32_t m_ = 0;                                                                     \
            _Pragma("unroll") for (int j = 0; j < 16; j++) {                                     \
                const uint32_t a_ = __umul24(y2_[j], 33u) + s2_[j + 2];                          \
                const uint32_t bm_ = *(const uint32_t *)((const uint8_t *)L.t3b + a_);           \
                m_ = __builtin_amdgcn_alignbit(bm_ >> (K1A_BYTE(d_, j + 3) & 31u), m_, 1);       \
            }                                                                                    \
            m_ >>= 16;                                                                           \
            if (!interior) { /* wave-uniform: a scalar branch */                                 \
                const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;                  \
                uint32_t keep_ = 0, force_ = 0;                                                  \
                _Pragma("unroll") for (int j = 0; j < 16; j++) {                                 \
                    if (any_start && p0_ + j >= lead && p0_ + j <= last_start) keep_ |= 1u << j; \
                    if (p0_ + j + 4 > total) force_ |= 1u << j; /* bytes behind the end took part in the test */ \
                }                                                                                \
                m_ = (m_ | force_) & keep_;                                                      \
            }                                                                                    \
            MROW = m_;                                                                           \
        }
        K1A_ROW(0, v0, __builtin_amdgcn_readfirstlane(v1.x), mrow0)
        K1A_ROW(1, v1, __builtin_amdgcn_readfirstlane(v2.x), mrow1)
        K1A_ROW(2, v2, __builtin_amdgcn_readfirstlane(v3.x), mrow2)
        K1A_ROW(3, v3, vL, mrow3)
        K1A_ISSUE_TILE(tile + nw)
        uint64_t m64 = (uint64_t)(mrow0 | (mrow1 << 16)) | ((uint64_t)(mrow2 | (mrow3 << 16)) << 32);
        while (true) { // (branch-free rounds, as in K1b)
            const bool has = m64 != 0;
            const unsigned long long act = __ballot(has);
            if (!act) break;
            const uint32_t np = __popcll(act);
            if (q1c + np > 64) advance((uint32_t)tile, 1u);
            const uint32_t pos = (uint32_t)__builtin_ctzll(m64 | (1ull << 63));
            m64 &= m64 - 1;
            const uint32_t slot = q1c + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
            if (has) q1[slot] = (uint16_t)(((pos >> 4) << 10) + lane * 16 + (pos & 15));
            q1c += np;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (slots && (kG & 15)) { // the counts of the wave's last tiles
        __builtin_amdgcn_wave_barrier();
        if (lane < (kG & 15)) GK.hcnt[gw * GK.cnt_iters + (kG & ~15u) + lane] = L.cb[wave][lane];
    }
    if (lane == 0) D.counts[region] = dcur;
#undef K1A_ISSUE_ROW
#undef K1A_ISSUE_TILE
#undef K1A_BYTE
#undef K1A_ROW
}

// the walks that go on: trie records + the window (+ haystack bytes beyond it), every pattern that ends
// on the way is an occurrence
__global__ __launch_bounds__(256) void k1a_walk(DevAutomaton A, const DevAutomaton *Ad, Segments G, DeepSink D,
                                                Sink GK, const uint8_t *__restrict__ stream, uint64_t len) {
    __shared__ unsigned long long lcount;
    __shared__ BlockSink sK;
    __shared__ uint8_t cls[256];
    if (threadIdx.x == 0) { lcount = 0; sK = block_sink(GK, &lcount); }
    cls[threadIdx.x] = A.classes[threadIdx.x];
    __syncthreads();
    const BlockSink *K = &sK;
    for (uint32_t b = blockIdx.x; b < D.regions; b += gridDim.x) {
        uint64_t n = D.counts[b];
        if (n > D.cap) { // items were dropped: the caller redoes the call another way
            n = D.cap;
            if (threadIdx.x == 0 && GK.abort_flag) *GK.abort_flag = 1;
        }
        const uint4 *recs = D.recs + (uint64_t)b * D.cap * 2;
        for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint4 it = recs[2 * i], wq = recs[2 * i + 1];
            const uint64_t pos = ((uint64_t)it.y << 32) | it.x;
            const uint64_t room = segment_end(G, len, pos) - pos;
            uint32_t node = it.z, d = it.w & ~ITEM_ROOT;
            if (d > room) continue; // (the levels the scan settled reached beyond the end of this haystack)
            uint64_t w = ((uint64_t)wq.y << 32) | wq.x; // haystack bytes pos + wd .. pos + wd + 7
            uint32_t wd = 0;
            bool first = !(it.w & ITEM_ROOT);
            for (;;) {
                // (the record of an item's first node travelled with it -- unless a pattern ends there or it is a
                // tail: the pattern's id / the tail's second half did not)
                uint4 r = make_uint4(wq.z, wq.w, 0u, 0u);
                if (!first || (wq.w & (GREC_OWN | GREC_TAIL))) r = A.grec[node];
                first = false;
                if (r.y & GREC_TAIL) { // ONE pattern's remaining bytes below this node: compare them and stop
                    const uint32_t tn = (r.y >> 24) & 15u;
                    bool same = d + tn <= room;
                    for (uint32_t k = 0; same && k < tn; k++) {
                        if (d + k - wd >= 8) { w = load_window(stream, len, pos + d + k); wd = d + k; }
                        const uint32_t hb = (uint32_t)(w >> (8 * (d + k - wd))) & 0xFFu;
                        const uint32_t tb = ((k < 4 ? r.x : r.w) >> (8 * (k & 3))) & 0xFFu;
                        same = hb == tb;
                    }
                    if (same) {
                        if (K->hslots) {
                            const uint64_t tile = (pos + K->lead) >> TILE_BITS;
                            const uint32_t slot = atomicAdd(&K->hcnt[hcnt_index(tile, K->cnt_nw, K->cnt_iters)], 1u);
                            if (slot < HIT_SLOTS)
                                K->hslots[(tile * HIT_SLOTS + slot) * 2] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), HIT_VERIFIED | r.z, d + tn);
                            else
                                *K->abort_flag = 1;
                        } else {
                            emit_one(Ad, *K, r.z, pos + d + tn);
                        }
                    }
                    break;
                }
                if (r.y & GREC_OWN) {
                    if (r.z == OWN1_MANY) {
                        for (uint32_t k = A.own_off[node]; k < A.own_off[node + 1]; k++) emit_one(Ad, *K, A.own_pid[k], pos + d);
                    } else if (K->hslots) { // (the pattern's length is the depth: nothing to gather)
                        const uint64_t tile = (pos + K->lead) >> TILE_BITS;
                        const uint32_t slot = atomicAdd(&K->hcnt[hcnt_index(tile, K->cnt_nw, K->cnt_iters)], 1u);
                        if (slot < HIT_SLOTS)
                            K->hslots[(tile * HIT_SLOTS + slot) * 2] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), HIT_VERIFIED | r.z, d);
                        else
                            *K->abort_flag = 1;
                    } else {
                        emit_one(Ad, *K, r.z, pos + d);
                    }
                }
                if (d >= room) break;
                if (d - wd >= 8) { w = load_window(stream, len, pos + d); wd = d; }
                const uint32_t c = cls[(w >> (8 * (d - wd))) & 0xFF];
                if (!((r.x >> c) & 1u)) break;
                node = (r.y & ID_MASK) + __popc(r.x & ((1u << c) - 1u));
                d++;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && GK.block_counts) GK.block_counts[blockIdx.x] = lcount;
}

// (k1a_scan's LDS image is static: on a device with less LDS the launch would fail outright -- such a
// device walks in chunks)
bool pfac_available(const DevAutomaton &A, size_t max_lds) { return A.t3b != nullptr && max_lds >= sizeof(K1aLds); }
uint32_t pfac_scan_grid(const uint8_t *d_hay, uint64_t len, int n_cus) {
    const uint64_t total = ((uintptr_t)d_hay & 15) + len;
    uint64_t blocks = ((total + 4095) / 4096 + 15) / 16;
    if (blocks > (uint64_t)n_cus) blocks = n_cus;
    return blocks ? (uint32_t)blocks : 1;
}
// items the scan may leave before the call gives up: 1 per 64 haystack bytes (32 B each: half a byte of
// workspace per haystack byte; the headline input leaves 1 per 500); dense output: 1 per 16 bytes (a
// one-byte pattern that every 26th letter matches is 1 per 26): the u64 words they take
static uint64_t pfac_region_cap(uint64_t len, uint32_t scan_grid, bool dense) {
    return len / (dense ? 16 : 64) / ((uint64_t)scan_grid * 16) + 1024;
}
uint64_t pfac_workspace_words(uint64_t len, uint32_t scan_grid, bool dense) {
    return (uint64_t)scan_grid * 16 * pfac_region_cap(len, scan_grid, dense) * 4;
}
hipError_t launch_pfac(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G, const Sink &K,
                       const uint8_t *d_hay, uint64_t len, uint32_t scan_grid, uint64_t *work, uint64_t *counts,
                       uint32_t walk_grid, bool dense, hipStream_t st) {
    if (len == 0) return hipSuccess;
    const uint64_t lead = (uintptr_t)d_hay & 15;
    const DeepSink D{(uint4 *)work, counts, pfac_region_cap(len, scan_grid, dense), scan_grid * 16};
    hipLaunchKernelGGL(k1a_scan, dim3(scan_grid), dim3(1024), 0, st, A.t3b, A.classes, A.t3r, A.grec, D, K, G,
                       d_hay - lead, len, lead, A.min_len, A.n_classes);
    hipLaunchKernelGGL(k1a_walk, dim3(walk_grid), dim3(256), 0, st, A, Ad, G, D, K, d_hay, len);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sink bookkeeping: totals + compaction of the per-workgroup regions
// ---------------------------------------------------------------------------
// One workgroup per sink (blockIdx 0: occurrences, 1: prefix hits):
// summary[2 b] = total records kept, summary[2 b + 1] = max count of a region,
// offsets[r] = exclusive prefix of min(count, region_cap), offsets[grid] = total (sink 0 only).
struct SinkView { const uint64_t *block_counts; uint32_t grid; uint64_t region_cap; };
__global__ __launch_bounds__(1024) void k_sink_summary(SinkView occ, SinkView hit, uint64_t *summary,
                                                       uint64_t *offsets) {
    // thread t owns the regions [t * per, (t + 1) * per)
    using scan_t = rocprim::block_scan<uint64_t, 1024>;
    __shared__ typename scan_t::storage_type scan_tmp;
    __shared__ uint64_t red[16];
    const SinkView V = blockIdx.x == 0 ? occ : hit;
    if (blockIdx.x != 0) offsets = nullptr;
    summary += 2 * blockIdx.x;
    const uint32_t per = (V.grid + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    uint64_t mine = 0, mx = 0;
    for (uint32_t b = b0; b < b0 + per && b < V.grid; b++) {
        uint64_t c = V.block_counts[b];
        mx = c > mx ? c : mx;
        mine += c < V.region_cap ? c : V.region_cap;
    }
    uint64_t excl = 0;
    scan_t().exclusive_scan(mine, excl, (uint64_t)0, scan_tmp);
    uint64_t run = excl;
    for (uint32_t b = b0; b < b0 + per && b < V.grid; b++) {
        if (offsets) offsets[b] = run;
        uint64_t c = V.block_counts[b];
        run += c < V.region_cap ? c : V.region_cap;
    }
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t other = __shfl_down(mx, o);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 1023) { if (offsets) offsets[V.grid] = run; summary[0] = run; }
    if (threadIdx.x == 0) {
        uint64_t m = 0;
        for (int i = 0; i < 16; i++) m = red[i] > m ? red[i] : m;
        summary[1] = m;
    }
}

// AoS regions -> dense SoA (key, pid) arrays (input of the radix sort)
__global__ __launch_bounds__(256) void k_sink_compact(const uint4 *recs, const uint64_t *offsets,
                                                      uint64_t region_cap, uint64_t *keys_out,
                                                      uint32_t *pids_out) {
    uint64_t o0 = offsets[blockIdx.x], n = offsets[blockIdx.x + 1] - o0;
    const uint4 *r = recs + (region_cap ? (uint64_t)blockIdx.x * region_cap : o0); // (0: exact regions, contiguous)
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint4 v = r[i];
        keys_out[o0 + i] = ((uint64_t)v.y << 32) | v.x;
        pids_out[o0 + i] = v.z;
    }
}

// hit_counts may be null (K1a: no prefix-hit sink): summary[2..3] are then left alone
hipError_t sink_summary(const uint64_t *block_counts, uint32_t grid, uint64_t region_cap,
                        const uint64_t *hit_counts, uint32_t hit_grid, uint64_t hit_cap,
                        uint64_t *summary, uint64_t *offsets, hipStream_t st) {
    hipLaunchKernelGGL(k_sink_summary, dim3(hit_counts ? 2 : 1), dim3(1024), 0, st,
                       SinkView{block_counts, grid, region_cap}, SinkView{hit_counts, hit_grid, hit_cap},
                       summary, offsets);
    return hipGetLastError();
}

hipError_t sink_compact(const uint4 *recs, const uint64_t *offsets, uint32_t grid,
                        uint64_t region_cap, uint64_t *keys_out, uint32_t *pids_out, hipStream_t st) {
    hipLaunchKernelGGL(k_sink_compact, dim3(grid), dim3(256), 0, st, recs, offsets, region_cap,
                       keys_out, pids_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K2: sort + resolve
// ---------------------------------------------------------------------------
size_t sort_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr,
                                    (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

hipError_t sort_occurrences(void *temp, size_t temp_bytes, const uint64_t *keys_in,
                            uint64_t *keys_out, const uint32_t *pids_in, uint32_t *pids_out,
                            uint64_t n, int end_bit, hipStream_t st) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, pids_in, pids_out,
                                     (size_t)n, 0, (unsigned)end_bit, st);
}

__global__ void k_make_spans(DevAutomaton A, int key_mode, const uint64_t *keys,
                             const uint32_t *pids, uint64_t *S, uint64_t *E, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = keys[i] >> A.rank_bits;
    uint64_t l = A.plen[pids[i]];
    if (key_mode == 0) { E[i] = x; S[i] = x - l; }
    else { S[i] = x; E[i] = x + l; }
}

hipError_t make_spans(const DevAutomaton &A, int key_mode, const uint64_t *keys,
                      const uint32_t *pids, uint64_t *S, uint64_t *E, uint64_t n,
                      hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_make_spans, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, A,
                       key_mode, keys, pids, S, E, n);
    return hipGetLastError();
}

size_t scan_temp_bytes(uint64_t n) {
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::inclusive_scan(nullptr, a, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (size_t)n, rocprim::maximum<uint64_t>(), (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  0u, (size_t)n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, c, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(),
                                  (hipStream_t)0);
    size_t d = 0;
    (void)rocprim::inclusive_scan(nullptr, d, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (size_t)n + 1, rocprim::plus<uint64_t>(), (hipStream_t)0);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > d ? m : d;
}

hipError_t prefix_max(void *temp, size_t temp_bytes, const uint64_t *E, uint64_t *M,
                      uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    return rocprim::inclusive_scan(temp, temp_bytes, E, M, (size_t)n,
                                   rocprim::maximum<uint64_t>(), st);
}

// Occurrence i is a "sync point" when every earlier occurrence ends at or
// before its start (M[i-1] <= S[i]): whatever the greedy did before, i is
// reported.  Between sync points the greedy chain is walked sequentially by
// the thread that owns the sync point (chains are short unless matches pile
// up on one another).
__global__ void k_resolve(const uint64_t *S, const uint64_t *E, const uint64_t *M,
                          uint32_t *flags, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool sync = i == 0 || M[i - 1] <= S[i];
    if (!sync) return;
    flags[i] = 1;
    uint64_t pos = E[i];
    for (uint64_t j = i + 1; j < n; j++) {
        uint64_t sj = S[j];
        if (M[j - 1] <= sj) break; // next sync point: its owner takes over
        bool take = sj >= pos;
        flags[j] = take ? 1u : 0u;
        if (take) pos = E[j];
    }
}

hipError_t resolve_greedy(const uint64_t *S, const uint64_t *E, const uint64_t *M,
                          uint32_t *flags, uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, S, E, M,
                       flags, n);
    return hipGetLastError();
}

hipError_t flag_offsets(void *temp, size_t temp_bytes, const uint32_t *flags, uint32_t *idx,
                        uint64_t n, hipStream_t st) {
    // flags has n + 1 entries (flags[n] = 0) so that idx[n] = total
    return rocprim::exclusive_scan(temp, temp_bytes, flags, idx, 0u, (size_t)n + 1,
                                   rocprim::plus<uint32_t>(), st);
}

__global__ void k_write_matches(const uint32_t *pids, const uint64_t *S, const uint64_t *E,
                                const uint32_t *flags, const uint32_t *idx, acx_match_t *out,
                                uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t o = i;
    if (flags) {
        if (!flags[i]) return;
        o = idx[i];
    }
    out[o].pattern = pids[i];
    out[o].start = S[i];
    out[o].end = E[i];
}

hipError_t write_matches(const uint32_t *pids, const uint64_t *S, const uint64_t *E,
                         const uint32_t *flags, const uint32_t *idx, acx_match_t *out,
                         uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_write_matches, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                       pids, S, E, flags, idx, out, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K2: sparse path -- verification, order, match kind, compaction
// ---------------------------------------------------------------------------
// The scan left its hits in per-tile slots (Sink / TileSpace in device_types.hpp).  The launch
// geometry of everything that follows depends only on the number of tiles (known to the host),
// never on the number of hits (known only to the device), so the whole post stage is queued
// behind the scan without a host round trip.
//   k_tile_main   one workgroup per GROUP of 64 tiles (256 KiB of index space): verifies the
//                 hits of its tiles (+ `lookback` context tiles in front) against the pattern
//                 bytes, drops the occurrences into LDS buckets of 4 KiB of KEY position,
//                 orders every bucket, applies the match kind and writes the group's REPORTED
//                 occurrences, in order, to its stretch of trecs
//   k_tile_write  the groups' output offsets (sums of supergroup words), the call's totals to
//                 pinned host memory, final (pattern, start, end) records, coalesced through LDS
//
// Match kind without communication between groups.  Sorted by key, the non-overlapping greedy
// accepts occurrence i iff start_i >= end of the last accepted one.  Occurrence i is a *sync
// point* when every earlier occurrence ends at or before its start: it is accepted whatever
// happened before, and the greedy from there on is independent of the past.  A group stages
// the occurrences of its context tiles too; an occurrence whose start lies at or beyond
// W = (first staged index + max_len - 1) has all the occurrences that could overlap it staged
// (nothing that starts before the staged tiles reaches it), so its sync-point test over the
// staged occurrences alone is exact ("certified").  Every output bucket re-derives the greedy
// from the nearest certified sync point at or before it -- chains are 1-2 long unless matches pile
// up.  If there is none (a chain of overlapping occurrences longer than the context: periodic
// patterns on periodic text), or a bucket overflows, the group raises the abort flag and the
// host redoes the call on the dense path, whose resolve is global.
// lead (non-continuation) bytes among the bytes of w selected by `valid` (0x80 per byte kept)
__device__ __forceinline__ uint32_t lead_in_word(uint64_t w, uint64_t valid) {
    const uint64_t HI = 0x8080808080808080ull;
    const uint64_t cont = w & ((~w) << 1) & HI; // continuation: bit7 = 1, bit6 = 0
    return __popcll(valid & HI & ~cont);
}

// str API: how many of the lead bytes of the 16-byte chunk that holds a match's start lie AT OR BEHIND the start
// (0 .. 16) -- k_tile_main has the 16 haystack bytes at the start in registers (the hit's window, or the head it
// compared), so the write kernel turns the offset into a code point from the counts alone and never touches the
// haystack again.  CP_UNKNOWN: no window came with the occurrence (the DFA walk's hits).
constexpr uint32_t CP_UNKNOWN = 31, CP_BITS = 5;

#ifndef ACX_MAIN_THREADS
#define ACX_MAIN_THREADS 256
#endif
constexpr uint32_t MAIN_THREADS = ACX_MAIN_THREADS;
#ifndef ACX_MAIN_THREADS_W32
#define ACX_MAIN_THREADS_W32 128 // threads of the instantiation with narrow staged words (sixteen groups per CU)
#endif
#ifndef ACX_MAIN_THREADS_CP
#define ACX_MAIN_THREADS_CP 128  // ... of the str API's (code points; round 6, same-box pairs: cfg5 0.5404 -> 0.5283 ms; cfg4's plain
#endif                           // wide-word form loses 1.6 % with 128 threads and keeps 256)
// How many groups a CU works on at once is what this latency-bound kernel lives on.  LDS (14.7 KiB) and VGPRs
// (63) allow 8 workgroups of four waves per CU, but 256-thread workgroups are admitted up to
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) per CU (MI355X guide): the 106 SGPRs the compiler takes by itself
// admit 6, 80 admit 8 -- the few values that do not fit ride in VGPR lanes.
#ifndef ACX_MAIN_SGPR_LIMIT
#define ACX_MAIN_SGPR_LIMIT 80
#endif
#if ACX_MAIN_SGPR_LIMIT > 0
#define ACX_MAIN_SGPR __attribute__((amdgpu_num_sgpr(ACX_MAIN_SGPR_LIMIT)))
#else
#define ACX_MAIN_SGPR
#endif
constexpr uint32_t STAGE_BUCKETS = GROUP_TILES + MAX_LOOKBACK;
// occurrences a bucket can stage (LDS per group decides how many groups a CU works on at once)
#ifndef ACX_STAGE_SLOTS
#define ACX_STAGE_SLOTS 24
#endif
constexpr uint32_t STAGE_SLOTS = ACX_STAGE_SLOTS;
static_assert(GROUP_TILES <= 64, "one wave owns the output buckets of a group");
static_assert(STAGE_SLOTS <= 32, "sync / accept flags of a bucket are one 32-bit mask");
constexpr uint32_t STAGE_SLOTS_WIDE = 64; // the wide form (device_types.hpp: GROUP_MAX_WIDE): the flags are 64-bit masks
// (narrow words leave room: 28 slots keep sixteen 128-thread groups on a CU -- 9.3 KiB each -- and a bucket of a haystack with a
// match every 512 bytes, 12 occurrences on average, overflows 200 times less often than with 24)
constexpr uint32_t STAGE_SLOTS_W32 = 28;
// A staged occurrence is ONE 64-bit word (the group's LDS footprint decides how many groups a CU
// works on at once, and the kernel is bound by the latency of its gathers, not by anything it
// computes): [ rel : 19 | tie : rank_bits | length : 45 - rank_bits ], rel = index of the key
// position relative to the first staged tile.  Words compare like the occurrence keys they stand
// for.  (The sparse path takes automata whose patterns are shorter than the context tiles --
// tile_lookback -- so a length always fits.)
constexpr uint32_t REL_BITS = 19;
static_assert(((GROUP_TILES + MAX_LOOKBACK) << TILE_BITS) <= (1u << REL_BITS), "group-relative key positions");
static_assert((MAX_LOOKBACK << TILE_BITS) < (1u << (64 - REL_BITS - 24)), "pattern lengths of the sparse path");
// The output offset of a group = the matches of the groups in front of it.  Groups add their count
// (and their statistics: occurrences << 32 | prefix hits) into the words of their SUPERGROUP of 64
// groups -- relaxed atomics, nobody waits: the kernel boundary orders them before k_tile_write --
// so that a group of k_tile_write sums n_groups / 64 supergroup words + at most 63 group counts.
// Two sets of supergroup words are used by the calls in turn (seq & 1): k_tile_write leaves the
// other set clear for the next call.
constexpr uint32_t SUPER = 64;

__device__ __forceinline__ uint64_t rec_key(const uint4 v) { return ((uint64_t)v.y << 32) | v.x; }

// span of an occurrence record {key lo, key hi, tie (rank or pattern id), pattern length}
__device__ __forceinline__ void span_of(uint32_t rank_bits, int key_mode, uint4 v, uint64_t *s, uint64_t *e) {
    const uint64_t x = rec_key(v) >> rank_bits;
    if (key_mode == 0) { *e = x; *s = x - v.w; }
    else { *s = x; *e = x + v.w; }
}

// span of a staged occurrence, relative to the first staged tile (a start may lie in front of it)
// (W32 -- the narrow form of a staged word, round 6: [ key position & 4095 : 12 | tie : rank_bits | length : 20 - rank_bits ] in 32
// bits; the bucket b the word lies in supplies the rest of the position)
constexpr uint32_t W32_FIELD = 20; // bits of tie + length in a narrow staged word
template <bool CP, bool W32>
__device__ __forceinline__ void staged_span(uint32_t rank_bits, int key_mode, uint64_t r, uint32_t b, int32_t *s, int32_t *e) {
    int32_t rel, L;
    if constexpr (W32) {
        rel = (int32_t)((b << TILE_BITS) | ((uint32_t)r >> W32_FIELD));
        L = (int32_t)((uint32_t)r & ((1u << (W32_FIELD - rank_bits)) - 1));
    } else {
        const uint32_t lb = 64 - REL_BITS - rank_bits - (CP ? CP_BITS : 0); // (CP: the carried count sits above the length)
        rel = (int32_t)(r >> (64 - REL_BITS));
        L = (int32_t)(r & ((1ull << lb) - 1));
    }
    if (key_mode == 0) { *e = rel; *s = rel - L; }
    else { *s = rel; *e = rel + L; }
}

// W32 (round 6): staged words of 32 bits -- pattern sets whose tie-break field and lengths fit W32_FIELD bits together (10^4
// patterns of up to 63 bytes), byte offsets.  The stage is what decides how many groups a CU works on at once, and this
// kernel lives on groups in flight: with narrow words a group takes 8 KiB of LDS, and with NT = 128 threads (two waves)
// sixteen groups fit a CU instead of eight -- all 4 096 groups of a 1 GiB haystack are resident at once.
// SLOTS: occurrences a bucket stages -- SLOTS, or SLOTS_WIDE (the wide form: inputs with a match every 100 - 500 bytes).
template <bool ANCH, bool CP, bool W32, uint32_t NT, uint32_t SLOTS>
__global__ __launch_bounds__(NT, NT == 128 ? (W32 ? 8 : 5) : SLOTS > 32 ? 4 : 6) ACX_MAIN_SGPR void k_tile_main(DevAutomaton A, Segments G, int key_mode,
                                                            int overlapping, TileSpace T, uint32_t lookback,
                                                            uint32_t lead, const uint8_t *__restrict__ stream,
                                                            uint64_t len, uint32_t *abort_flag, uint64_t seq,
                                                            uint64_t *seg_counts, uint64_t n_seg, int hot_ok) {
    // abort_flag: the call's control block (device_types.hpp).  hot_ok (K1b's hits): a group that cannot be finished here
    // -- a staged tile with more hits than its slots, a full bucket, more than GROUP_MAX matches, a chain that no
    // certified sync point cuts -- is handed to the hot pipeline (its id in the hot list, T.btot[g] = HOT_BIT) and costs
    // the CALL nothing; without it such a group raises the abort flag and the call is redone on the dense path.
    // (batch: the per-haystack counts k_tile_write adds to -- cleared here, no memset in front of the scan)
    if (seg_counts)
        for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n_seg; i += (uint64_t)gridDim.x * NT)
            seg_counts[i] = 0;
    // (rows padded to an odd number of words: lane t works on row t, and a power-of-two row stride would
    // put all 64 lanes on the same LDS banks -- measured: 67 % of this kernel's LDS cycles were conflicts)
    static_assert(!(W32 && CP), "a narrow staged word has no room for the carried lead-byte count");
    using SW = typename std::conditional<W32, uint32_t, uint64_t>::type;
    __shared__ SW st[STAGE_BUCKETS][SLOTS + 1]; // staged occurrences by bucket of key position
    __shared__ uint32_t bn[STAGE_BUCKETS];         // occurrences per bucket
    __shared__ int32_t bmax[STAGE_BUCKETS];        // largest end per bucket
    __shared__ uint32_t hoff[STAGE_BUCKETS + 2];   // exclusive prefix of the tiles' hit counts
    using MK = typename std::conditional<(SLOTS > 32), uint64_t, uint32_t>::type; // one flag per slot of a bucket
    __shared__ MK synm[STAGE_BUCKETS];             // per bucket: sync points
    __shared__ uint32_t fail, stop, tail_base;
    const uint32_t t = threadIdx.x, g = blockIdx.x;
    const uint32_t tile0 = g * GROUP_TILES;
    const uint32_t first = tile0 >= lookback ? tile0 - lookback : 0; // first staged tile
    const uint32_t lb = tile0 - first;                               // context tiles in front
    // anchors (automaton.hpp): a hit lies up to SHIFT_MAX bytes BEHIND the start of its occurrence, so an
    // occurrence that starts in the group's last bytes has its hit in the first tile of the next group: that
    // tile's hits are read too (the occurrences that start beyond the group are dropped like the context's)
    const uint32_t la = ANCH ? 1u : 0u;
    const uint32_t nb = GROUP_TILES + lb + la;                       // tiles whose hits are read
    // ---- hits of the staged tiles: exclusive prefix of the counts (wave 0: 64 tiles, wave 1: the
    // few beyond -- nb <= 68)
    uint32_t c = 0;
    bool overfull = false; // the tile's hits did not fit its slots (K1b: the rest is in the overflow list)
    if (t < nb && first + t < T.n_tiles) {
        c = T.hcnt[hcnt_index(first + t, T.cnt_nw, T.cnt_iters)];
        overfull = c > HIT_SLOTS;
        c = c < HIT_SLOTS ? c : HIT_SLOTS;
    }
    if (t < STAGE_BUCKETS) { bn[t] = 0; bmax[t] = 0; }
    if (t == 0) { fail = 0; stop = *abort_flag; }
    // the group gives up: all threads get here together
    auto give_up = [&]() {
        if (t == 0) {
            if (hot_ok) {
                T.btot[g] = HOT_BIT;
                const uint32_t i = atomicAdd(abort_flag + CTL_HOT_COUNT, 1u);
                (*(uint32_t *const *)(abort_flag + CTL_HOT_LIST))[i] = g; // (the list holds n_groups ids: a group enters once)
            } else {
                *abort_flag = 1;
            }
        }
    };
    uint32_t incl = c;
    if (t < 128) { // (lanes beyond nb carry zeros)
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)(t & 63) >= o) incl += v;
        }
        if (t == 63) tail_base = incl;
    }
    __syncthreads();
    if (t >= 64 && t < 128) incl += tail_base;
    if (t <= nb) hoff[t] = incl - c;
    if (overfull) fail = 1;
    __syncthreads();
    const uint32_t H = hoff[nb];
    if (stop) return;
    if (fail) { give_up(); return; }
    if (H == 0) { // nothing staged at all (sparse inputs): the group reports nothing
        if (t == 0) T.btot[g] = 0;
        return;
    }
    // index space: index = stream position + lead; a tile / bucket is 4 KiB of it
    const uint64_t idx_lo = (uint64_t)tile0 << TILE_BITS, idx_hi = idx_lo + ((uint64_t)GROUP_TILES << TILE_BITS);
    // occurrences whose key index is below `complete` may have unseen company: not staged.
    // occurrences that start at or beyond `wlow` can be certified as sync points
    const uint64_t margin = A.max_len ? A.max_len - 1 : 0;
    const uint64_t first_idx = (uint64_t)first << TILE_BITS;
    const uint64_t complete = first == 0 ? 0 : first_idx + (key_mode == 0 ? margin : 0);
    const int32_t wlow = first == 0 ? 0 : (int32_t)margin; // relative to first_idx
    const uint32_t rank_bits = A.rank_bits, len_bits = 64 - REL_BITS - rank_bits;
    // ---- verify: one thread per hit
    for (uint32_t h = t; h < H; h += NT) {
        uint32_t lo = 0, hi = nb; // tile j = the last one with hoff[j] <= h
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (hoff[mid] <= h) lo = mid; else hi = mid;
        }
        const uint4 *rec = T.hslots + ((uint64_t)(first + lo) * HIT_SLOTS + (h - hoff[lo])) * 2;
        const uint4 r0 = rec[0];
        const uint4 w = rec[1]; // (unconditional: in flight together with the first half)
        const uint64_t p = ((uint64_t)r0.y << 32) | r0.x;
        uint64_t w0 = 0, w1 = 0, room = 0, back = 0;
        uint32_t nc = 1, li = 0, code = r0.z;
        const bool verified = code != HIT_RETRY && (code & HIT_VERIFIED) != 0;
        bool list = false;
        if (!verified) {
            w0 = ((uint64_t)w.y << 32) | w.x; w1 = ((uint64_t)w.w << 32) | w.z;
            if (ANCH) { // the haystack's start matters only to patterns filed behind their beginning
                uint64_t seg_lo, seg_hi;
                segment_bounds(G, len, p, &seg_lo, &seg_hi);
                room = seg_hi - p; back = p - seg_lo;
            } else {
                room = segment_end(G, len, p) - p;
            }
            if (code == HIT_RETRY) code = prefix_code(A.ptab, A.ptab_log2, A.filter_q2, w0);
            if (code == HIT_NONE) nc = 0;
            else if (code & HIT_LIST) { list = true; li = code & ~HIT_LIST; nc = A.blist[li]; }
        }
        // (cin -- CP only: lead bytes of the start's 16-byte chunk at or behind the start, or CP_UNKNOWN)
        auto stage = [&](uint64_t ps, uint32_t pid, uint32_t L, uint32_t rk, uint32_t cin) { // ps: where the occurrence starts
            const uint64_t kidx = (key_mode == 0 ? ps + L : ps) + lead;
            if (kidx < complete || kidx >= idx_hi) return; // another group's (or nobody's) business
            const uint32_t rel = (uint32_t)(kidx - first_idx);
            const uint32_t b = rel >> TILE_BITS;
            const uint32_t r = atomicAdd(&bn[b], 1u);
            if (r < SLOTS) {
                if constexpr (W32) {
                    st[b][r] = ((rel & ((1u << TILE_BITS) - 1)) << W32_FIELD) | ((key_mode == 1 ? pid : rk) << (W32_FIELD - rank_bits)) | L;
                } else {
                    uint64_t word = ((((uint64_t)rel << rank_bits) | (key_mode == 1 ? pid : rk)) << len_bits) | L;
                    if constexpr (CP) word |= (uint64_t)cin << (len_bits - CP_BITS);
                    st[b][r] = word;
                }
            } else fail = 1;
        };
        // lead bytes among the first (16 - (ps & 15)) bytes of the window (x0, x1) at ps, not beyond the stream's end
        auto leads_in_chunk = [&](uint64_t ps, uint64_t x0, uint64_t x1) -> uint32_t {
            uint32_t n_in = 16 - (uint32_t)(ps & 15);
            if (len - ps < n_in) n_in = (uint32_t)(len - ps);
            const uint64_t v0 = n_in >= 8 ? ~0ull : (n_in ? ~0ull >> (8 * (8 - n_in)) : 0);
            const uint64_t v1 = n_in > 8 ? (n_in >= 16 ? ~0ull : ~0ull >> (8 * (16 - n_in))) : 0;
            return lead_in_word(x0, v0) + lead_in_word(x1, v1);
        };
        if (verified) {
            const uint32_t pid = code & ~HIT_VERIFIED;
            stage(p, pid, r0.w, key_mode == 1 ? 0u : A.rank[pid], CP_UNKNOWN);
        } else if (!list) {
            uint32_t rk;
            uint64_t ps = p, x0 = w0, x1 = w1;
            const uint32_t L = nc ? verify_candidate<ANCH>(A, stream, len, p, code, w0, w1, room, back, &rk, &ps,
                                                           CP && ANCH ? &x0 : nullptr, CP && ANCH ? &x1 : nullptr) : 0;
            if (L) stage(ps, code & CODE_PID_MASK, L, rk, CP ? leads_in_chunk(ps, x0, x1) : 0u);
        } else {
            // a list (patterns that share their key): two candidates at a time, their gathers in
            // flight together -- a thread with a long list otherwise holds the whole group back
            for (uint32_t k = 0; k < nc; k += 2) {
                const bool two = k + 1 < nc;
                const uint32_t pid0 = A.blist[li + 1 + k], pid1 = A.blist[li + 1 + (two ? k + 1 : k)];
                uint32_t rk0, rk1;
                uint64_t ps0, ps1, a0 = w0, a1 = w1, b0 = w0, b1 = w1;
                const uint32_t L0 = verify_candidate<ANCH>(A, stream, len, p, pid0, w0, w1, room, back, &rk0, &ps0,
                                                           CP && ANCH ? &a0 : nullptr, CP && ANCH ? &a1 : nullptr);
                const uint32_t L1 = verify_candidate<ANCH>(A, stream, len, p, pid1, w0, w1, room, back, &rk1, &ps1,
                                                           CP && ANCH ? &b0 : nullptr, CP && ANCH ? &b1 : nullptr);
                if (L0) stage(ps0, pid0 & CODE_PID_MASK, L0, rk0, CP ? leads_in_chunk(ps0, a0, a1) : 0u);
                if (two && L1) stage(ps1, pid1 & CODE_PID_MASK, L1, rk1, CP ? leads_in_chunk(ps1, b0, b1) : 0u);
            }
        }
    }
    __syncthreads();
    if (fail) { give_up(); return; }
    // ---- order every bucket (words are unique), largest end per bucket
    // (buckets, not tiles read: the look-ahead tile of an anchored set has no bucket -- with four context tiles,
    // patterns longer than 10 KiB, `t < nb` reached one row beyond the stage)
    const uint32_t nbk = GROUP_TILES + lb;
    if (t < nbk) {
        const uint32_t n = bn[t];
        int32_t mx = 0;
        for (uint32_t i = 0; i < n; i++) {
            const SW v = st[t][i];
            uint32_t j = i;
            while (j > 0 && st[t][j - 1] > v) { st[t][j] = st[t][j - 1]; j--; }
            st[t][j] = v;
            int32_t s, e;
            staged_span<CP, W32>(rank_bits, key_mode, v, t, &s, &e);
            mx = max(mx, e);
        }
        bmax[t] = mx;
    }
    __syncthreads();
    uint32_t cnt = 0; // reported occurrences of output bucket t (wave 0)
    MK accepted = 0;
    if (overlapping) {
        if (t < GROUP_TILES) { cnt = bn[lb + t]; accepted = cnt >= 8 * sizeof(MK) ? ~(MK)0 : ((MK)1 << cnt) - 1; }
    } else {
        // ---- certified sync points.  Only occurrences of the last (lookback + 1) buckets can end
        // beyond the start of one in bucket t (an occurrence spans at most max_len - 1 bytes
        // besides its key position, and lookback tiles are longer than that).
        if (t < nbk) {
            int32_t m = 0;
            for (uint32_t b = t > lookback + 1 ? t - lookback - 1 : 0; b < t; b++) m = max(m, bmax[b]);
            const uint32_t n = bn[t];
            MK sm = 0;
            for (uint32_t i = 0; i < n; i++) {
                int32_t s, e;
                staged_span<CP, W32>(rank_bits, key_mode, st[t][i], t, &s, &e);
                if (s >= wlow && m <= s) sm |= (MK)1 << i;
                m = max(m, e);
            }
            synm[t] = sm;
        }
        __syncthreads();
        // ---- greedy chains: every output bucket from the nearest certified sync point
        if (t < GROUP_TILES && bn[lb + t]) {
            const uint32_t ob = lb + t;
            int b = (int)ob, i = 0;
            bool ok = true;
            while (!((synm[b] >> i) & 1)) { // walk back to a sync point
                if (--i < 0) {
                    do { b--; } while (b >= 0 && bn[b] == 0);
                    if (b < 0) { ok = false; break; }
                    i = (int)bn[b] - 1;
                }
            }
            if (!ok) {
                fail = 1; // the chain enters from beyond the context: dense path
            } else {
                int32_t pos = INT32_MIN;
                for (;;) { // forward again, to the end of bucket ob
                    int32_t s, e;
                    staged_span<CP, W32>(rank_bits, key_mode, st[b][i], (uint32_t)b, &s, &e);
                    const bool take = s >= pos;
                    if (take) pos = e;
                    if ((uint32_t)b == ob && take) { accepted |= (MK)1 << i; cnt++; }
                    if (++i >= (int)bn[b]) {
                        if ((uint32_t)b == ob) break;
                        do { b++; } while (bn[b] == 0); // ob is not empty: terminates there at the latest
                        i = 0;
                    }
                }
            }
        }
        __syncthreads();
        if (fail) { give_up(); return; }
    }
    // ---- compact the reported occurrences of the 64 output buckets into the group's stretch
    if (t < 64) {
        uint32_t incl = cnt, occ = t < GROUP_TILES ? bn[lb + t] : 0u;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)t >= o) incl += v;
        }
        const uint32_t total = __shfl(incl, 63);
        for (int o = 32; o > 0; o >>= 1) occ += __shfl_down(occ, o);
        if (total > T.gmax) {
            give_up(); // (wave 0 is here as a whole)
        } else {
            // records {key lo, key hi, tie, length} in stream coordinates (k_tile_write maps a rank to its pattern)
            uint4 *dst = T.trecs + (uint64_t)g * T.gmax + (incl - cnt);
            uint64_t *dst8 = (uint64_t *)T.trecs + (uint64_t)g * T.gmax + (incl - cnt); // (W32: ONE word per reported occurrence)
            const uint64_t base = first_idx - lead;
            for (uint32_t k = 0; accepted; k++) {
                const uint32_t i = sizeof(MK) == 8 ? (uint32_t)__builtin_ctzll((unsigned long long)accepted) : (uint32_t)__builtin_ctz((uint32_t)accepted);
                accepted &= accepted - 1;
                uint64_t r = st[lb + t][i];
                if constexpr (W32) { // [ key position in the stream : 44 | tie | length ] (k_tile_write, T.w8)
                    dst8[k] = ((base + (((lb + t) << TILE_BITS) | ((uint32_t)r >> W32_FIELD))) << W32_FIELD) | ((uint32_t)r & ((1u << W32_FIELD) - 1));
                    continue;
                }
                const uint64_t x = r >> len_bits; // rel << rank_bits | tie
                const uint64_t key = x + (base << rank_bits);
                if constexpr (CP) { // length | carried count << 24 (k_tile_write takes it apart)
                    const uint64_t lc = r & ((1ull << len_bits) - 1); // (len_bits can exceed 32: few patterns, short ranks)
                    dst[k] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(x & ((1u << rank_bits) - 1)),
                                        (uint32_t)(lc & ((1ull << (len_bits - CP_BITS)) - 1)) | ((uint32_t)(lc >> (len_bits - CP_BITS)) << 24));
                } else {
                    dst[k] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(x & ((1u << rank_bits) - 1)),
                                        (uint32_t)(r & ((1ull << len_bits) - 1)));
                }
            }
        }
        if (t == 0 && total <= T.gmax) {
            // the group's count, and count and statistics added into the words of its supergroup
            // (k_tile_write, which starts when every group is done, places the output with them)
            const uint32_t n = total;
            T.btot[g] = n;
            uint64_t *sgw = T.sgw + (seq & 1) * 2 * (uint64_t)T.sg_cap;
            __hip_atomic_fetch_add(sgw + g / SUPER, (uint64_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sgw + T.sg_cap + g / SUPER, ((uint64_t)occ << 32) | (hoff[lb + GROUP_TILES] - hoff[lb]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// non-continuation (lead) bytes in [p, end): whole aligned 8-byte words, the bytes outside the
// range masked off (an aligned word never leaves the page its first / last byte of the range is
// in, so the words at the two ends are safe to read in full) -- no byte loops.  Four words per
// round, their loads issued together: a match span is one round, not a chain of dependent misses.
__device__ __forceinline__ uint64_t lead_bytes_between(const uint8_t *p, const uint8_t *end) {
    if (p >= end) return 0;
    const uint8_t *q = (const uint8_t *)((uintptr_t)p & ~(uintptr_t)7);
    uint64_t first = ~0ull << (8 * ((uintptr_t)p & 7)); // bytes of the first word at or after p
    uint64_t c = 0;
    for (; q < end; q += 32, first = ~0ull) {
        uint64_t w[4], valid[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t left = end - (q + 8 * k); // bytes of [q + 8k, end)
            w[k] = left > 0 ? *(const uint64_t *)(q + 8 * k) : 0;
            valid[k] = left >= 8 ? ~0ull : left > 0 ? ~0ull >> (8 * (8 - left)) : 0;
        }
        valid[0] &= first;
#pragma unroll
        for (int k = 0; k < 4; k++) c += lead_in_word(w[k], valid[k]);
    }
    return c;
}

// code-point index of byte offset x = number of non-continuation bytes in [0, x): the prefix of its 1 KiB block
// + the counts of the 16-byte chunks in front of it inside the block (64 count bytes: ONE line, byte sums by
// v_sad_u8) + the lead bytes of its own chunk in front of x: that chunk's count minus `carried` (how many of
// them lie at or behind x -- the caller had those bytes in registers), or, carried == CP_UNKNOWN, at most 15
// bytes counted in place.  Every load's address depends on x only: they are all in flight together.
__device__ __forceinline__ uint64_t code_point_of(const uint8_t *__restrict__ hay, const uint64_t *blockpre,
                                                  const uint8_t *__restrict__ sub, uint64_t x, uint32_t carried) {
    const uint64_t blk = x >> 10;
    const uint32_t i = (uint32_t)(x & 1023) >> 4; // whole 16-byte chunks before x inside the block
    const uint4 *sp = (const uint4 *)(sub + blk * 64);
    const uint4 sv[4] = {sp[0], sp[1], sp[2], sp[3]};
    const uint64_t pre = blockpre[blk];
    uint32_t tail = 0;
    if (carried == CP_UNKNOWN && (x & 15)) {
        // (uniform branch: a haystack that is not 16-byte aligned takes the word-wise count)
        if (((uintptr_t)hay & 15) == 0) {
            const uint4 v = *(const uint4 *)(hay + (x & ~15ull));
            const uint32_t n = (uint32_t)(x & 15);
            const uint64_t v0 = n >= 8 ? ~0ull : ~0ull >> (8 * (8 - n)), v1 = n > 8 ? ~0ull >> (8 * (16 - n)) : 0;
            tail = lead_in_word(((uint64_t)v.y << 32) | v.x, v0) + lead_in_word(((uint64_t)v.w << 32) | v.z, v1);
        } else {
            tail = (uint32_t)lead_bytes_between(hay + (x & ~15ull), hay + x);
        }
    }
    const uint32_t upto = carried == CP_UNKNOWN ? i : i + 1; // count bytes summed: the chunks in front (+ x's own)
    const uint32_t w[16] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w, sv[1].x, sv[1].y, sv[1].z, sv[1].w,
                            sv[2].x, sv[2].y, sv[2].z, sv[2].w, sv[3].x, sv[3].y, sv[3].z, sv[3].w};
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t d = 0; d < 16; d++) {
        const uint32_t nb = upto > 4 * d ? (upto - 4 * d < 4 ? upto - 4 * d : 4) : 0;
        const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1);
        acc = __builtin_amdgcn_sad_u8(w[d] & m, 0u, acc);
    }
    return carried == CP_UNKNOWN ? pre + acc + tail : pre + acc - carried;
}

// The groups' reported occurrences -> final records.  A workgroup takes one group: its output is
// one contiguous stretch of 24-byte records, assembled in LDS and written as a flat array of
// dwords (coalesced).  The stretch starts at the sum of the counts of the groups in front (the
// full supergroups' words + the counts of the groups in front inside the own supergroup).
// Group 0 -- the first to run -- adds up ALL supergroup words and publishes the call's totals
// to host_out (pinned, system-coherent host memory the host polls: ONE self-checking line,
// publish_line below), mirrors them in `summary`, clears the abort flag and the supergroup words the NEXT
// call will use: the host returns while the records are still being written (they are consumed
// in stream order).
// seg_counts != null (batch of haystacks, byte offsets): the records get offsets local to
// their haystack and the per-haystack counts are taken here -- one atomic per run of matches
// of the same haystack inside the stretch instead of a separate pass with one atomic per match.
constexpr uint32_t WRITE_THREADS = 256;
// matches of a group assembled in LDS at a time: the image decides how many groups a CU works on at
// once (the whole 1024-match stretch of round 2 took 28 KB: 5 groups per CU, 16.5 us; 7 KB: all 8 the
// waves allow, 13.6 us on the headline input)
constexpr uint32_t WRITE_CHUNK = 256;
// cp.blockpre != null (str API, one haystack): byte offsets -> code-point indexes on the way out.
struct CodePointTables { const uint8_t *hay; const uint64_t *blockpre; const uint8_t *sub; const uint32_t *pchars; };
// ONE aligned 64-byte line of coherent pinned host memory, written by ONE store instruction (four lanes x 16 bytes): [0] seq,
// [1 .. 6] payload, [7] seq ^ k0_line_check(payload).  The host polls word 0 and takes a COPY of the line when the last
// word agrees with the payload as read: separate writes to host memory arrive in no particular order (round 5: K0's matches
// beside its line did not, 1 call in ~30 000), so everything a poller reads is one such line -- the totals of the sparse path
// too (until round 5: eight words, a system-scope fence, then the number).
__device__ __forceinline__ void publish_line(volatile uint64_t *line, uint32_t t, uint64_t seq, const uint64_t (&mid)[6]) {
    if (t >= 4) return;
    const uint64_t last = seq ^ k0_line_check(mid);
    uint64_t w[2];
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
        const uint32_t i = 2 * t + k;
        w[k] = i == 0 ? seq : i == 7 ? last : mid[i - 1 > 5 ? 5 : i - 1];
    }
    ((ulonglong2 *)line)[t] = make_ulonglong2(w[0], w[1]);
}
struct PostOut {
    uint64_t *summary;           // device mirror of the totals
    volatile uint64_t *host_out; // pinned host memory the host polls: ONE line (publish_line): [1] matches, [2] occurrences, [3] prefix
                                 // hits, [4] why | hot groups << 8, [5] overflow hits | fullest overflow list << 32
    uint32_t *next_flag;         // the control block of the NEXT call: left clear
    uint64_t seq;                // this call's sequence number (its parity selects the set of supergroup words)
    uint32_t lead;               // DENSE: index = stream position + lead
    uint64_t pub;                // the number group 0's line carries (pass 0: seq)
    // pass 0: the call's first (normally only) write kernel -- when k_tile_main left groups to the hot pipeline it only
    // publishes {[12] hot groups, [13] overflow hits} and writes nothing; pass 1: the write kernel behind the hot
    // pipeline -- the hot groups' counts are in T.btot / the supergroup words by then, their records are written by
    // the HOT instantiation; hot_abort (pass 1 / HOT): the hot pipeline's own abort flag
    int pass;
    const uint32_t *hot_abort;
    // pass 0, when hot groups are announced: the hot pipeline's bucket counters (DenseTiles::counts, hot_tiles of them) are
    // cleared here, 64 per workgroup -- the workgroups are launched anyway and write nothing in that case; two host-side
    // memsets in the pipeline's way were 16 us of every call with a hot group (null: the host clears them)
    uint32_t *hot_counts;
    uint32_t hot_tiles;
    const uint32_t *spec_ctl; // pass 1 of a speculative launch (hot_groups_here): the call's control block, else null
    uint32_t spec_bound;
};
// Round 6 -- the hot pipeline queued AHEAD of the knowledge that it is needed (a context whose last call had hot groups): its
// kernels are launched right behind k_tile_main / the write kernel's pass 0 with grids for `bound` hot groups and read the
// number of hot groups from the call's control block themselves -- no host round trip between pass 0 and the pipeline, no
// launch latency on the critical path (one hot 64 KiB region cost the call a round trip + four launches issued behind it:
// +43 us on a 356 us step).  0 hot groups: every kernel returns at once (pass 0 has written everything); more than `bound`,
// or a call that gave up: they return too, and the host runs the pipeline the old way.  ctl == null: n is the host's figure.
__device__ __forceinline__ uint32_t hot_groups_here(const uint32_t *ctl, uint32_t bound, uint32_t n) {
    if (!ctl) return n;
    const uint32_t nh = ctl[CTL_HOT_COUNT];
    return (nh > bound || ctl[CTL_ABORT] != 0 || ctl[CTL_OVF_LOST] != 0) ? 0u : nh;
}
// HOT instantiation: one workgroup per dense group (DT_GROUP tiles) of a hot group of the sparse path
struct HotWrite {
    const uint32_t *list;  // hot groups (ids of the sparse path's groups)
    const uint64_t *trecs; // the dense groups' reported occurrences (DT_GMAX words each, k_dense_main)
    const uint32_t *btot;  // ... and their counts
    uint32_t n_dense;      // dense groups there are
    const uint32_t *spec_ctl; // speculative launch (hot_groups_here): the call's control block, else null
    uint32_t spec_bound;
};
// CPW: cp.blockpre != null (the instantiation without code points is the round-3 kernel); GMAX: records a group
// of T.trecs holds (GROUP_MAX: the sparse path; DT_GMAX: the tile-ordered dense path)
// DENSE: T.trecs holds the tile-ordered dense path's 64-bit words (k_dense_main) instead of 16-byte records
template <bool CPW, uint32_t GMAX, bool DENSE = false, bool HOT = false>
__global__ __launch_bounds__(WRITE_THREADS) void k_tile_write(uint32_t rank_bits, int key_mode,
                                                              const uint32_t *__restrict__ by_rank, TileSpace T,
                                                              acx_match_t *out, const uint32_t *abort_flag,
                                                              Segments G, uint64_t *seg_counts, CodePointTables cp,
                                                              PostOut O, HotWrite W) {
    __shared__ uint32_t img[WRITE_CHUNK * 6];
    __shared__ uint32_t hs[WRITE_CHUNK]; // haystack index of the matches, in output order
    __shared__ uint64_t red[4];
    __shared__ uint64_t s_base;
    const uint32_t t = threadIdx.x;
    // g: the group of T whose place in the output this workgroup computes; gi: the group its records are read from
    // (HOT: dense group `sub` of the hot group g)
    uint32_t g = blockIdx.x, gi = blockIdx.x, sub = 0;
    if constexpr (HOT) {
        if (blockIdx.x / HOT_SUB >= hot_groups_here(W.spec_ctl, W.spec_bound, 0xFFFFFFFFu)) return;
        g = W.list[blockIdx.x / HOT_SUB];
        sub = blockIdx.x % HOT_SUB;
        gi = g * HOT_SUB + sub;
    }
    // (the flags are stable by now: their writers completed)
    bool stop;
    uint32_t n_hot = 0, n_ovf = 0, why = 0; // why (the line's word 4): 1 the call gave up, 2 only the overflow list was too small
    if (!HOT && O.pass == 1 && O.spec_ctl && hot_groups_here(O.spec_ctl, O.spec_bound, 0) == 0) return; // (nothing for this pass to do)
    if (HOT || O.pass == 1) {
        stop = *O.hot_abort != 0;
        why = stop ? 1 : 0;
    } else {
        // sparse path: an overflow list was too small = hits were lost (the host grows the lists or takes the dense path)
        n_hot = DENSE ? 0u : abort_flag[CTL_HOT_COUNT];
        why = abort_flag[CTL_ABORT] != 0 ? 1u : (!DENSE && abort_flag[CTL_OVF_LOST] != 0) ? 2u : 0u;
        stop = why != 0;
    }
    const uint64_t *sgw = T.sgw + (O.seq & 1) * 2 * (uint64_t)T.sg_cap;
    const uint32_t n_super = (T.n_groups + SUPER - 1) / SUPER;
    if (!HOT && g == 0) { // the call's totals, as early as they can be known
        uint64_t m = 0, occ = 0, hit = 0;
        for (uint32_t k = t; k < n_super; k += WRITE_THREADS) {
            m += sgw[k];
            const uint64_t v = sgw[T.sg_cap + k];
            occ += v >> 32; hit += v & 0xFFFFFFFFu;
        }
        for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o); occ += __shfl_xor(occ, o); hit += __shfl_xor(hit, o); }
        // (three rounds through the same four words; 4 waves)
        uint64_t tot[3];
        const uint64_t part[3] = {m, occ, hit};
        for (int r = 0; r < 3; r++) {
            __syncthreads();
            if ((t & 63) == 0) red[t >> 6] = part[r];
            __syncthreads();
            tot[r] = red[0] + red[1] + red[2] + red[3];
        }
        uint64_t *other = T.sgw + ((O.seq & 1) ^ 1) * 2 * (uint64_t)T.sg_cap;
        for (uint32_t k = t; k < 2 * T.sg_cap; k += WRITE_THREADS) other[k] = 0;
        // the overflow lists' fill (pass 0 of the sparse path): total and fullest list, for the host; the next call's counters: clear
        uint32_t ovf_max = 0;
        if (!DENSE && O.pass == 0) {
            static_assert(WRITE_THREADS == OVF_LISTS, "one thread per overflow list");
            const uint32_t mine = (*(const uint32_t *const *)(abort_flag + CTL_OVF_COUNTS))[t * OVF_COUNT_STRIDE];
            uint32_t sum = mine, mx = mine;
            for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, (uint32_t)__shfl_xor(mx, o)); }
            __syncthreads();
            if ((t & 63) == 0) red[t >> 6] = ((uint64_t)mx << 32) | sum;
            __syncthreads();
            for (int k = 0; k < 4; k++) { n_ovf += (uint32_t)red[k]; ovf_max = max(ovf_max, (uint32_t)(red[k] >> 32)); }
            if (O.next_flag) (*(uint32_t *const *)(O.next_flag + CTL_OVF_COUNTS))[t * OVF_COUNT_STRIDE] = 0;
        }
        if (t == 0) {
            O.summary[0] = tot[1]; O.summary[2] = tot[2]; O.summary[4] = tot[0];
            if (O.next_flag) { O.next_flag[CTL_ABORT] = 0; O.next_flag[CTL_OVF_LOST] = 0; O.next_flag[CTL_HOT_COUNT] = 0; }
            if (!DENSE && O.pass == 0 && n_hot) { O.summary[10] = 0; O.summary[11] = 0; } // (the hot pipeline's flags: no memset in its way)
        }
        // (every thread holds the totals: lanes 0 .. 3 write the line)
        const uint64_t mid[6] = {tot[0], tot[1], tot[2], (uint64_t)why | ((uint64_t)n_hot << 8), (uint64_t)n_ovf | ((uint64_t)ovf_max << 32), 0};
        publish_line(O.host_out, t, O.pub, mid);
    }
    if (stop) return;
    if (!HOT && !DENSE && O.pass == 0 && n_hot) { // (the hot pipeline first: this kernel runs again behind it)
        if (O.hot_counts && t < GROUP_TILES) {
            const uint32_t k = g * GROUP_TILES + t;
            if (k < O.hot_tiles) O.hot_counts[k] = 0;
            if (g == T.n_groups - 1 && t == 0) // (the counters beyond the last group's tiles)
                for (uint32_t q = T.n_groups * GROUP_TILES; q < O.hot_tiles; q++) O.hot_counts[q] = 0;
        }
        return;
    }
    uint32_t n;
    if constexpr (HOT) {
        if (gi >= W.n_dense) return;
        n = W.btot[gi];
    } else {
        n = T.btot[g];
        if (!DENSE && (n & HOT_BIT)) return; // the HOT instantiation writes this group's records
    }
    if (n == 0) return;
    if (t < 64) { // the stretch's place: the groups in front
        const uint32_t sg = g / SUPER, gq = g % SUPER;
        uint64_t sum = t < gq ? (T.btot[sg * SUPER + t] & ~HOT_BIT) : 0;
        for (uint32_t k = t; k < sg; k += 64) sum += sgw[k];
        if constexpr (HOT) { // ... and the dense groups in front inside the hot group
            if (t < sub) sum += W.btot[g * HOT_SUB + t];
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (t == 0) s_base = sum;
    }
    __syncthreads();
    const uint64_t base = s_base;
    for (uint32_t c0 = 0; c0 < n; c0 += WRITE_CHUNK) {
        const uint32_t m = n - c0 < WRITE_CHUNK ? n - c0 : WRITE_CHUNK;
        for (uint32_t i = t; i < m; i += WRITE_THREADS) {
            uint4 v;
            if constexpr (DENSE) {
                const uint64_t *words = HOT ? W.trecs : (const uint64_t *)T.trecs;
                const uint64_t w_ = words[(uint64_t)gi * GMAX + c0 + i];
                const uint32_t lenb = 48 - rank_bits;
                const uint64_t pos_ = (((uint64_t)gi * DT_GROUP) << TILE_BITS) + (w_ >> 48) - O.lead; // key position in the stream
                const uint64_t tie_ = (w_ >> lenb) & ((1ull << rank_bits) - 1);
                const uint64_t key_ = (pos_ << rank_bits) | tie_;
                v = make_uint4((uint32_t)key_, (uint32_t)(key_ >> 32), (uint32_t)tie_, (uint32_t)(w_ & ((1ull << lenb) - 1)));
            } else {
                if (T.w8) { // (uniform) the narrow-word form's reported occurrences: [key position : 44 | tie | length]
                    const uint64_t w_ = ((const uint64_t *)T.trecs)[(uint64_t)g * T.gmax + c0 + i];
                    const uint32_t lb_ = W32_FIELD - rank_bits, f_ = (uint32_t)w_ & ((1u << W32_FIELD) - 1), tie_ = f_ >> lb_;
                    const uint64_t key_ = ((w_ >> W32_FIELD) << rank_bits) | tie_;
                    v = make_uint4((uint32_t)key_, (uint32_t)(key_ >> 32), tie_, f_ & ((1u << lb_) - 1));
                } else {
                    v = T.trecs[(uint64_t)g * T.gmax + c0 + i]; // (the sparse path's stretches: GROUP_MAX or GROUP_MAX_WIDE records)
                }
            }
            uint32_t carried = CP_UNKNOWN;
            // (str API: k_tile_main<.., CP> packed the count above the length; the dense path's words carry none)
            if constexpr (CPW && !DENSE) { carried = v.w >> 24; v.w &= 0xFFFFFFu; }
            uint64_t s, e;
            span_of(rank_bits, key_mode, v, &s, &e);
            if (seg_counts) {
                uint64_t h, hbase;
                if (G.uniform_len) { h = s / G.uniform_len; hbase = h * G.uniform_len; }
                else { h = upper_bound_u64(G.offsets, G.n_hay + 1, s) - 1; hbase = G.offsets[h]; }
                s -= hbase; e -= hbase;
                hs[i] = (uint32_t)h;
            }
            const uint32_t pid = key_mode == 1 ? v.z : by_rank[v.z];
            if constexpr (CPW) { // (a match is as many code points as its pattern: nothing of the span is read)
                const uint64_t cs = code_point_of(cp.hay, cp.blockpre, cp.sub, s, carried);
                e = cs + cp.pchars[pid];
                s = cs;
            }
            uint32_t *d = img + i * 6;
            d[0] = pid; d[1] = 0;
            d[2] = (uint32_t)s; d[3] = (uint32_t)(s >> 32); d[4] = (uint32_t)e; d[5] = (uint32_t)(e >> 32);
        }
        __syncthreads();
        uint32_t *flat = (uint32_t *)(out + base + c0);
        for (uint32_t k = t; k < m * 6; k += WRITE_THREADS) flat[k] = img[k];
        if (seg_counts) { // (a run of matches of one haystack cut by a chunk boundary adds twice: same sum)
            for (uint32_t c = t; c < m; c += WRITE_THREADS) {
                const uint32_t h = hs[c];
                if (c > 0 && hs[c - 1] == h) continue; // not the head of its run
                uint32_t run = 1;
                while (c + run < m && hs[c + run] == h) run++;
                atomicAdd((unsigned long long *)&seg_counts[h], (unsigned long long)run);
            }
        }
        __syncthreads();
    }
}

// narrow staged words (k_tile_main<.., W32, ..>): the tie-break field and the longest pattern's length fit W32_FIELD bits
// together, byte offsets (ACX_MAIN_WIDE=1: always the wide form -- measurements).  The groups' stretches then hold ONE 64-bit
// word per reported occurrence (TileSpace::w8): the caller sets that flag by this function.
bool tile_words_narrow(const DevAutomaton &A, bool codepoints) {
    static const bool wide_env = std::getenv("ACX_MAIN_WIDE") != nullptr;
    uint32_t lbits = 0;
    while ((1u << lbits) <= A.max_len) lbits++;
    return !codepoints && !wide_env && A.rank_bits + lbits <= W32_FIELD;
}

uint32_t tile_lookback(uint32_t max_len) {
    // context tiles in front of a group: longer than the longest pattern by at least 2 KiB
    const uint64_t need = (uint64_t)(max_len ? max_len - 1 : 0) + 2048;
    return (uint32_t)((need + (1u << TILE_BITS) - 1) >> TILE_BITS);
}

// Verify, order, resolve and compact the hits of the whole call into out[] (capacity
// n_groups * GROUP_MAX suffices).  The published line's `why` != 0 afterwards: the output did not fit the sparse
// path and out[] / the totals are meaningless.
hipError_t tile_post(const DevAutomaton &A, int key_mode, bool overlapping, const TileSpace &T, uint32_t lead,
                     const uint8_t *d_hay, uint64_t len, acx_match_t *out, uint64_t *summary,
                     uint32_t *abort_flag, uint32_t *next_flag, uint64_t *host_out, uint64_t seq,
                     const Segments &G, uint64_t *seg_counts, const uint64_t *cp_blockpre,
                     const uint8_t *cp_sub, hipEvent_t before_write, bool hot_ok, uint32_t *hot_counts, uint32_t hot_tiles,
                     hipStream_t st) {
    const int ov = overlapping ? 1 : 0;
    const uint32_t lookback = tile_lookback(A.max_len);
    if (lookback > MAX_LOOKBACK) return hipErrorInvalidValue; // (the caller keeps such automata off this path)
    const bool cpw = cp_blockpre != nullptr; // the write kernel converts to code points: the occurrences carry their chunk counts
    // narrow staged words (k_tile_main<.., W32, ..>): the tie-break field and the longest pattern's length fit W32_FIELD bits
    // together, byte offsets (ACX_MAIN_WIDE=1: always the wide form -- measurements)
    const bool w32 = tile_words_narrow(A, cpw);
    if ((T.w8 != 0) != w32) return hipErrorInvalidValue; // (the caller's TileSpace says what the stretches will hold)
    const bool wide = T.gmax > GROUP_MAX; // (the context's choice: acx_api.cpp)
#define ACX_TILE_MAIN_W(AN, CPW, W, N, SL)                                                                            \
    hipLaunchKernelGGL((k_tile_main<AN, CPW, W, N, SL>), dim3(T.n_groups), dim3(N), 0, st, A, G, key_mode, ov, T, lookback, \
                       lead, d_hay, len, abort_flag, seq, seg_counts, seg_counts ? (G.n_hay ? G.n_hay : 1) : 0, hot_ok ? 1 : 0)
#define ACX_TILE_MAIN(AN) {                                                                                           \
        if (wide) { if (cpw) ACX_TILE_MAIN_W(AN, true, false, MAIN_THREADS, STAGE_SLOTS_WIDE); else if (w32) ACX_TILE_MAIN_W(AN, false, true, MAIN_THREADS, STAGE_SLOTS_WIDE); else ACX_TILE_MAIN_W(AN, false, false, MAIN_THREADS, STAGE_SLOTS_WIDE); } \
        else if (cpw) ACX_TILE_MAIN_W(AN, true, false, ACX_MAIN_THREADS_CP, STAGE_SLOTS);                                 \
        else if (w32) ACX_TILE_MAIN_W(AN, false, true, ACX_MAIN_THREADS_W32, STAGE_SLOTS_W32);                            \
        else ACX_TILE_MAIN_W(AN, false, false, MAIN_THREADS, STAGE_SLOTS); }
    if (A.max_shift) ACX_TILE_MAIN(true) else ACX_TILE_MAIN(false)
#undef ACX_TILE_MAIN
#undef ACX_TILE_MAIN_W
    if (before_write) { // (what the write kernel needs from another stream: the code-point prefix)
        hipError_t e = hipStreamWaitEvent(st, before_write, 0);
        if (e != hipSuccess) return e;
    }
    const PostOut O{summary, (volatile uint64_t *)host_out, next_flag, seq, 0, seq, 0, nullptr, hot_counts, hot_tiles, nullptr, 0};
    const HotWrite W0{nullptr, nullptr, nullptr, 0, nullptr, 0};
    if (cpw)
        hipLaunchKernelGGL((k_tile_write<true, GROUP_MAX>), dim3(T.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode, A.by_rank,
                           T, out, abort_flag, G, seg_counts, CodePointTables{d_hay, cp_blockpre, cp_sub, A.pchars}, O, W0);
    else
        hipLaunchKernelGGL((k_tile_write<false, GROUP_MAX>), dim3(T.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode, A.by_rank,
                           T, out, abort_flag, G, seg_counts, CodePointTables{d_hay, cp_blockpre, cp_sub, A.pchars}, O, W0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dense outputs, tile-ordered (round 4)
// ---------------------------------------------------------------------------
// Until round 4 every input that did not fit the hit slots went through a global radix sort (rocPRIM, six passes
// over 16-byte pairs: 41 % of a dense step) although its occurrences are born almost in order: a hit's key lies
// within max_len bytes of the hit.  Now:
//   k_dense_verify  one thread per prefix hit of K1b's regions (as k_walk_hits): the occurrences go, as ONE 64-bit
//                   word each, into the bucket of the 4 KiB tile their KEY position lies in (an arrival counter
//                   per tile: ~130 atomics per address on the densest inputs this path takes)
//   k_dense_main    one workgroup per DT_GROUP tiles + the context tiles in front: loads their buckets into LDS,
//                   a wave sorts a bucket (bitonic: words of a bucket compare like their keys), then the staged
//                   array -- buckets in order = key order -- gets the running maximum of the ends (block scan), the
//                   certified sync points, the greedy chains from them (as k_resolve, but in LDS), and the reported
//                   occurrences of the group's own tiles are written, in order, as records k_tile_write takes
//   k_tile_write    (the sparse path's own, with DT_GMAX records per group): output offsets, totals, final records,
//                   local offsets + per-haystack counts, code points
// A bucket that overflows (more than one occurrence per 8 bytes), or a chain that enters a group from beyond its
// context: the abort flag, and the call takes the radix-sort path after all.
constexpr uint32_t DT_STAGE = DT_GROUP + MAX_LOOKBACK;
constexpr uint32_t DT_THREADS = 256;

// One wave files the occurrences of (up to) 64 prefix hits, one per lane (live: the lane has one -- h = {position lo, hi,
// code, -}, w = the 16 haystack bytes at the position): verification as k_tile_main's, every occurrence ONE 64-bit word
// [key position & 4095 : 12 | tie : rank_bits | length] in the bucket of the tile its KEY position lies in.  The lanes of
// a wave take their bucket slots together: a wave's hits mostly share ONE bucket, so one atomic per distinct tile and step
// instead of 64 on the same address (measured before: 6.3 ms for 33 M occurrences, the atomics serialised).  All 64 lanes
// must call (wave-uniform loops).
template <bool ANCH>
__device__ __forceinline__ void dense_file_hits(const DevAutomaton &A, const Segments &G, const DenseTiles &D, int key_mode,
                                                uint32_t lead, const uint8_t *__restrict__ stream, uint64_t len,
                                                uint32_t *abort_flag, bool live, const uint4 h, const uint4 w) {
    const uint32_t rank_bits = A.rank_bits, len_bits = 52 - rank_bits;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below_me = (1ull << lane) - 1;
    uint64_t p = 0, w0 = 0, w1 = 0, room = 0, back = 0;
    uint32_t code = HIT_NONE;
    if (live) {
        p = ((uint64_t)h.y << 32) | h.x;
        w0 = ((uint64_t)w.y << 32) | w.x; w1 = ((uint64_t)w.w << 32) | w.z;
        uint64_t seg_lo, seg_hi;
        segment_bounds(G, len, p, &seg_lo, &seg_hi);
        room = seg_hi - p; back = p - seg_lo;
        code = h.z;
        if (code == HIT_RETRY) code = prefix_code(A.ptab, A.ptab_log2, A.filter_q2, w0);
    }
    const bool list = code != HIT_NONE && (code & HIT_LIST) != 0;
    const uint32_t li = code & ~HIT_LIST;
    const uint32_t nc = code == HIT_NONE ? 0 : list ? A.blist[li] : 1;
    for (uint32_t k = 0; __ballot(k < nc); k++) {
        uint32_t L = 0, rk = 0, cand = 0;
        uint64_t ps = 0;
        if (k < nc) {
            cand = list ? A.blist[li + 1 + k] : code; // pattern id | anchor shift << 24
            L = verify_candidate<ANCH>(A, stream, len, p, cand, w0, w1, room, back, &rk, &ps);
        }
        const bool have = L != 0;
        const uint64_t kidx = (key_mode == 0 ? ps + L : ps) + lead;
        const uint32_t tile = (uint32_t)(kidx >> TILE_BITS);
        uint32_t slot = 0;
        unsigned long long todo = __ballot(have);
        while (todo) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
            const uint32_t t0 = __shfl(tile, leader);
            const unsigned long long same = __ballot(have && tile == t0);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&D.counts[t0], (uint32_t)__popcll(same));
            base = __shfl(base, leader);
            if (have && tile == t0) slot = base + (uint32_t)__popcll(same & below_me);
            todo &= ~same;
        }
        if (have) {
            if (slot < DT_SLOTS)
                D.words[(uint64_t)tile * DT_SLOTS + slot] =
                    (((((uint64_t)kidx & ((1u << TILE_BITS) - 1)) << rank_bits) | (key_mode == 1 ? (cand & CODE_PID_MASK) : rk)) << len_bits) | L;
            else
                *abort_flag = 1; // denser than one occurrence per 8 bytes: the radix-sort path
        }
    }
}

template <bool ANCH>
__global__ __launch_bounds__(256) void k_dense_verify(DevAutomaton A, Segments G, Sink H, uint32_t h_grid, DenseTiles D,
                                                      int key_mode, uint32_t lead, const uint8_t *__restrict__ stream,
                                                      uint64_t len, uint32_t *abort_flag) {
    for (uint32_t b = blockIdx.x; b < h_grid; b += gridDim.x) {
        uint64_t n = H.block_counts[b];
        if (n > H.region_cap) n = H.region_cap; // hits were dropped: the host sees the count and redoes the call
        const uint4 *rec = H.recs + (uint64_t)b * H.region_cap * 2;
        // (a region holds a wave's hits tile after tile)
        for (uint64_t i0 = 0; i0 < n; i0 += 256) {
            const uint64_t i = i0 + threadIdx.x;
            const bool live = i < n;
            uint4 h = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
            if (live) { h = rec[2 * i]; w = rec[2 * i + 1]; }
            dense_file_hits<ANCH>(A, G, D, key_mode, lead, stream, len, abort_flag, live, h, w);
        }
    }
}

// HOT pipeline (device_types.hpp: control block): the hits of the hot groups' staged tiles -- the tiles of the group, the
// context tiles in front, the look-ahead tile of an anchored set -- from their hit slots, and the hits beyond the slots
// from the call's overflow list, filed like the dense path's.  One wave per tile.  A tile is filed ONCE: by its own group
// when that is hot, else by the (one) hot neighbour that stages it as context (the last tiles of a group) or looks
// ahead into it (its first tile).  Workgroups beyond the hot groups' take the overflow list (its hits belong to
// overfull tiles, whose groups are always hot).
static_assert(HIT_SLOTS <= 64, "one wave reads a tile's hit slots in one step");
constexpr uint32_t HV_TILES = GROUP_TILES + MAX_LOOKBACK + 1;        // tiles a hot group stages at most
constexpr uint32_t HV_BLOCKS = (HV_TILES + 3) / 4;                   // workgroups (of four waves) per hot group
template <bool ANCH>
__global__ __launch_bounds__(256) void k_hot_verify(DevAutomaton A, Segments G, TileSpace T, const uint32_t *hot_list, uint32_t n_hot,
                                                    const uint32_t *ctl, uint32_t ovf_blocks, uint32_t lookback, DenseTiles D, int key_mode,
                                                    uint32_t lead, const uint8_t *__restrict__ stream, uint64_t len,
                                                    uint32_t *abort_flag, uint32_t spec_bound) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (a speculative launch -- spec_bound != 0 -- has its grid laid out for spec_bound hot groups and reads their number itself)
    const uint32_t n_layout = spec_bound ? spec_bound : n_hot;
    if (spec_bound) {
        n_hot = hot_groups_here(ctl, spec_bound, 0);
        if (n_hot == 0) return;
    }
    if (blockIdx.x >= n_layout * HV_BLOCKS) { // the overflow lists: ovf_blocks workgroups each
        const uint32_t q = blockIdx.x - n_layout * HV_BLOCKS, list = q / ovf_blocks, b0 = q % ovf_blocks;
        const uint32_t cap = ctl[CTL_OVF_CAP];
        uint32_t n = (*(const uint32_t *const *)(ctl + CTL_OVF_COUNTS))[list * OVF_COUNT_STRIDE];
        n = n < cap ? n : cap;
        const uint4 *ovf = *(const uint4 *const *)(ctl + CTL_OVF_RECS) + 2 * (uint64_t)list * cap;
        for (uint32_t i0 = b0 * 256; i0 < n; i0 += ovf_blocks * 256) { // (block-uniform bounds: whole waves loop together)
            const uint32_t i = i0 + threadIdx.x;
            const bool live = i < n;
            uint4 h = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
            if (live) { h = ovf[2 * (uint64_t)i]; w = ovf[2 * (uint64_t)i + 1]; }
            dense_file_hits<ANCH>(A, G, D, key_mode, lead, stream, len, abort_flag, live, h, w);
        }
        return;
    }
    if (blockIdx.x / HV_BLOCKS >= n_hot) return;
    const uint32_t g = hot_list[blockIdx.x / HV_BLOCKS];
    const uint32_t tile0 = g * GROUP_TILES;
    const uint32_t first = tile0 >= lookback ? tile0 - lookback : 0;
    const uint32_t nb = GROUP_TILES + (tile0 - first) + (ANCH ? 1u : 0u);
    const uint32_t j = (blockIdx.x % HV_BLOCKS) * 4 + wave; // (wave-uniform)
    if (j >= nb) return;
    const uint32_t tile = first + j;
    if (tile >= T.n_tiles) return;
    const uint32_t owner = tile / GROUP_TILES;
    if (owner != g && (T.btot[owner] & HOT_BIT)) return; // a hot group files its own tiles
    uint32_t c = T.hcnt[hcnt_index(tile, T.cnt_nw, T.cnt_iters)];
    c = c < HIT_SLOTS ? c : HIT_SLOTS;
    if (c == 0) return;
    const bool live = lane < c;
    uint4 h = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
    if (live) {
        const uint4 *rec = T.hslots + ((uint64_t)tile * HIT_SLOTS + lane) * 2;
        h = rec[0]; w = rec[1];
    }
    dense_file_hits<ANCH>(A, G, D, key_mode, lead, stream, len, abort_flag, live, h, w);
}

// One wave sorts P = 64 * EPL words (ascending; the words are unique, the padding ~0): lane l holds the elements
// l * EPL .. l * EPL + EPL - 1 in registers; partners inside a lane swap in place, partners in other lanes come by
// ds_bpermute (two per 64-bit word).  (The first version of k_dense_main sorted in LDS: read, compare, write back,
// 36 dependent passes for 256 words -- 12 us per bucket.)
template <int EPL>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&v)[EPL], uint32_t lane) {
    constexpr uint32_t P = 64 * EPL;
#pragma unroll
    for (uint32_t k = 2; k <= P; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= (uint32_t)EPL) {
                const uint32_t lj = j / EPL;
                const bool lower = (lane & lj) == 0;
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const uint32_t lo = __shfl_xor((uint32_t)v[e], lj), hi = __shfl_xor((uint32_t)(v[e] >> 32), lj);
                    const uint64_t other = ((uint64_t)hi << 32) | lo;
                    const bool up = ((lane * EPL + e) & k) == 0;
                    const uint64_t mn = v[e] < other ? v[e] : other, mx = v[e] < other ? other : v[e];
                    v[e] = (up == lower) ? mn : mx;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    if (e & j) continue;
                    const bool up = ((lane * EPL + e) & k) == 0;
                    const uint64_t a = v[e], c = v[e | j];
                    const uint64_t mn = a < c ? a : c, mx = a < c ? c : a;
                    v[e] = up ? mn : mx;
                    v[e | j] = up ? mx : mn;
                }
            }
        }
    }
}
template <int EPL>
__device__ __forceinline__ void sort_bucket(uint64_t *bucket, const uint64_t *src, uint32_t n, uint32_t lane) {
    uint64_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) { const uint32_t i = lane * EPL + e; v[e] = i < n ? src[i] : ~0ull; }
    wave_bitonic_sort<EPL>(v, lane);
#pragma unroll
    for (int e = 0; e < EPL; e++) if (lane * EPL + e < n) bucket[lane * EPL + e] = v[e]; // (the padding sorts behind the words: not stored)
}

// LDS: static part below + dynamic: arr[nb_max][DT_SLOTS] (the staged buckets), syn[], acc[] (one byte per staged
// occurrence, by virtual index = position in the staged array without its gaps), nb_max = DT_GROUP + lookback --
// sized by the automaton's lookback so that sets with short patterns get 5 workgroups per CU, not 3
template <uint32_t NT>
struct DtLds {
    using scan_max = rocprim::block_scan<int32_t, NT>;
    using scan_sum = rocprim::block_scan<uint32_t, NT>;
    uint32_t cnt[DT_STAGE], voff[DT_STAGE + 1];
    union { typename scan_max::storage_type mx; typename scan_sum::storage_type sum; } scan;
    uint32_t first_sync, total, stop;
};
static size_t dense_main_lds(uint32_t lookback) { return (size_t)(DT_GROUP + lookback) * DT_SLOTS * (8 + 2); }
// COMPACT (round 6): the staged buckets lie back to back (a group's stage is as long as its occurrences, not DT_SLOTS words per
// bucket): DT_COMPACT_WORDS words -- 10 KiB of LDS instead of 26 -- and NT = 128 threads, so that sixteen groups fit a CU
// instead of six: the kernel is a chain of latencies per group (buckets -> sort -> scans -> chains), it lives on groups in
// flight.  A group with more staged occurrences sets the abort flag to DT_ABORT_COMPACT: the host repeats the kernel in its
// full form (inputs denser than ~200 occurrences per tile: the context then keeps to the full form).
constexpr uint32_t DT_COMPACT_WORDS = 1024, DT_ABORT_COMPACT = 2;
static size_t dense_main_lds_compact() { return (size_t)DT_COMPACT_WORDS * (8 + 2); }

// HOT pipeline (hot.list != null): the workgroups take the dense groups of the hot groups of the sparse path (HOT_SUB each),
// and a group's count is credited to ITS hot group in the sparse path's TileSpace (hot.S: btot keeps its HOT_BIT, the
// supergroup words of the call's set) -- the sparse path's write kernel then places every group of the call.
struct HotMain {
    const uint32_t *list; // hot groups, null: every dense group of the stream (the dense path proper)
    TileSpace S;          // the sparse path's groups
    uint64_t seq;         // the call's sequence number (its set of supergroup words)
    const uint32_t *spec_ctl; // speculative launch (hot_groups_here): the call's control block, else null
    uint32_t spec_bound;
};
template <uint32_t NT, bool COMPACT>
__global__ __launch_bounds__(NT) void k_dense_main(uint32_t rank_bits, uint32_t max_len, int key_mode, int overlapping,
                                                   DenseTiles D, TileSpace T, uint32_t lookback, uint32_t lead,
                                                   uint32_t *abort_flag, HotMain hot) {
    __shared__ DtLds<NT> L;
    using dt_scan_max = typename DtLds<NT>::scan_max;
    using dt_scan_sum = typename DtLds<NT>::scan_sum;
    extern __shared__ __attribute__((aligned(16))) uint8_t dt_dyn[];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (hot.list && blockIdx.x / HOT_SUB >= hot_groups_here(hot.spec_ctl, hot.spec_bound, 0xFFFFFFFFu)) return;
    const uint32_t gs = hot.list ? hot.list[blockIdx.x / HOT_SUB] : 0u;
    const uint32_t g = hot.list ? gs * HOT_SUB + blockIdx.x % HOT_SUB : blockIdx.x;
    if (g >= T.n_groups) return; // (hot: the last group of the sparse path may reach beyond the stream's tiles)
    const uint32_t nb_max = DT_GROUP + lookback;
    // (the stage: bucket b's words from word row(b) on -- DT_SLOTS apart, or, COMPACT, back to back: then a virtual index IS a word's place)
    uint64_t *const arr = (uint64_t *)dt_dyn;
    const uint32_t cap_words = COMPACT ? DT_COMPACT_WORDS : nb_max * DT_SLOTS;
    uint8_t *const syn = dt_dyn + (size_t)cap_words * 8, *const acc = syn + cap_words;
    const uint32_t tile0 = g * DT_GROUP;
    const uint32_t first = tile0 >= lookback ? tile0 - lookback : 0; // first staged tile
    const uint32_t lb = tile0 - first, nb = DT_GROUP + lb;
    const uint32_t len_bits = 52 - rank_bits;
    if (t < DT_STAGE) {
        uint32_t c = 0;
        if (t < nb && first + t < D.n_tiles) c = D.counts[first + t];
        L.cnt[t] = c < DT_SLOTS ? c : DT_SLOTS; // (overfull: the producer raised the abort flag)
    }
    if (t == 0) L.first_sync = 0xFFFFFFFFu;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < DT_STAGE; b++) { L.voff[b] = run; run += b < nb ? L.cnt[b] : 0; }
        L.voff[DT_STAGE] = run;
        // (other workgroups of this launch may raise the flag while this one runs: ONE read per workgroup, so that all its
        // waves take the same way past the barriers below)
        L.stop = *abort_flag;
    }
    __syncthreads();
    const uint32_t N = L.voff[DT_STAGE], out0 = L.voff[lb]; // staged occurrences; the first one of the group's own tiles
    if (N == out0 || L.stop) { // nothing to report (or the call is lost already)
        if (t == 0) T.btot[g] = 0;
        return;
    }
    if (COMPACT && N > DT_COMPACT_WORDS) { // (uniform) the full form takes the call
        if (t == 0) atomicCAS(abort_flag, 0u, DT_ABORT_COMPACT);
        return;
    }
    auto row = [&](uint32_t b) -> uint64_t * { return arr + (COMPACT ? L.voff[b] : b * DT_SLOTS); };
    // ---- load + sort: a wave takes the buckets wave, wave + 4 (words are unique and compare like their keys); the
    // sort runs in registers (wave_bitonic_sort), sized by the bucket's fill
    for (uint32_t b = wave; b < nb; b += NT / 64) {
        const uint32_t n = L.cnt[b];
        const uint64_t *src = D.words + (uint64_t)(first + b) * DT_SLOTS;
        if (n == 0) continue;
        if (n <= 64) sort_bucket<1>(row(b), src, n, lane);
        else if (n <= 128) sort_bucket<2>(row(b), src, n, lane);
        else if (n <= 256) sort_bucket<4>(row(b), src, n, lane);
        else sort_bucket<8>(row(b), src, n, lane);
    }
    __syncthreads();
    // virtual index v -> its bucket and word; spans relative to the first staged tile (position = bucket << 12 | rel)
    uint32_t vo[DT_STAGE]; // (the buckets' first virtual indexes, in registers: locate() runs for every element and pass)
#pragma unroll
    for (uint32_t q = 0; q < DT_STAGE; q++) vo[q] = q < nb ? L.voff[q] : 0xFFFFFFFFu;
    auto locate = [&](uint32_t v, uint32_t *b_) -> uint64_t {
        uint32_t b = 0, off = 0;
#pragma unroll
        for (uint32_t q = 1; q < DT_STAGE; q++) { const bool in = v >= vo[q]; b += in ? 1u : 0u; off = in ? vo[q] : off; }
        *b_ = b;
        return COMPACT ? arr[v] : arr[b * DT_SLOTS + (v - off)];
    };
    auto span = [&](uint32_t b, uint64_t w, int32_t *s_, int32_t *e_) {
        const int32_t rel = (int32_t)((b << TILE_BITS) | (uint32_t)(w >> 52)), Ln = (int32_t)(w & ((1ull << len_bits) - 1));
        if (key_mode == 0) { *e_ = rel; *s_ = rel - Ln; } else { *s_ = rel; *e_ = rel + Ln; }
    };
    auto span_at = [&](uint32_t v, int32_t *s_, int32_t *e_) { uint32_t b; const uint64_t w = locate(v, &b); span(b, w, s_, e_); };
    const uint32_t C = (N + NT - 1) / NT; // elements per thread (<= 16)
    const uint32_t v0 = t * C, v1 = v0 + C < N ? v0 + C : N;
    if (overlapping) {
        for (uint32_t v = v0; v < v1; v++) acc[v] = 1;
    } else {
        // ---- running maximum of the ends (chunks of C, block scan of the chunk maxima), certified sync points
        const int32_t margin = max_len ? (int32_t)max_len - 1 : 0;
        const int32_t wlow = first == 0 ? 0 : margin; // occurrences that start at or beyond it have all their company staged
        int32_t mine = INT32_MIN;
        for (uint32_t v = v0; v < v1; v++) { int32_t s_, e_; span_at(v, &s_, &e_); mine = max(mine, e_); }
        int32_t m = INT32_MIN; // maximum end in front of the chunk
        dt_scan_max().exclusive_scan(mine, m, INT32_MIN, L.scan.mx, rocprim::maximum<int32_t>());
        uint32_t fs = 0xFFFFFFFFu;
        for (uint32_t v = v0; v < v1; v++) {
            int32_t s_, e_;
            span_at(v, &s_, &e_);
            const bool sy = s_ >= wlow && m <= s_;
            syn[v] = sy;
            if (sy && fs == 0xFFFFFFFFu) fs = v;
            m = max(m, e_);
        }
        if (fs != 0xFFFFFFFFu) atomicMin(&L.first_sync, fs);
        __syncthreads();
        // an occurrence of the group's own tiles in front of every certified sync point: its chain enters from
        // beyond the context (periodic patterns on periodic text): the radix-sort path resolves globally
        if (L.first_sync > out0) { if (t == 0) *abort_flag = 1; return; }
        // ---- greedy chains: every sync point walks to the next one
        for (uint32_t v = t; v < N; v += NT) {
            if (!syn[v]) continue;
            int32_t s_, pos;
            span_at(v, &s_, &pos);
            acc[v] = 1;
            for (uint32_t u = v + 1; u < N && !syn[u]; u++) {
                int32_t su, eu;
                span_at(u, &su, &eu);
                const bool take = su >= pos;
                acc[u] = take;
                if (take) pos = eu;
            }
        }
    }
    __syncthreads();
    // ---- the reported occurrences of the group's own tiles, in order, to its stretch
    uint32_t cntm = 0;
    for (uint32_t v = v0 > out0 ? v0 : out0; v < v1; v++) cntm += acc[v];
    uint32_t at = 0;
    dt_scan_sum().exclusive_scan(cntm, at, 0u, L.scan.sum);
    if (t == NT - 1) L.total = at + cntm;
    __syncthreads();
    const uint32_t total = L.total;
    // (ONE 64-bit word per reported occurrence: [key position relative to the group's first tile : 16 | tie | length];
    // k_tile_write<.., DENSE> turns it back into a span)
    uint64_t *dst = (uint64_t *)T.trecs + (uint64_t)g * DT_GMAX;
    for (uint32_t v = v0 > out0 ? v0 : out0; v < v1; v++) {
        if (!acc[v]) continue;
        uint32_t b;
        const uint64_t w = locate(v, &b);
        const uint64_t relg = ((uint64_t)(b - lb) << TILE_BITS) | (w >> 52);
        dst[at++] = (relg << 48) | (((w >> len_bits) & ((1ull << rank_bits) - 1)) << (48 - rank_bits)) | (w & ((1ull << len_bits) - 1));
    }
    if (t == 0) {
        T.btot[g] = total;
        if (hot.list) {
            atomicAdd(&hot.S.btot[gs], total); // (below HOT_BIT: a hot group reports at most HOT_SUB * DT_GMAX occurrences)
            uint64_t *sgw = hot.S.sgw + (hot.seq & 1) * 2 * (uint64_t)hot.S.sg_cap;
            __hip_atomic_fetch_add(sgw + gs / SUPER, (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sgw + hot.S.sg_cap + gs / SUPER, (uint64_t)(N - out0) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint64_t *sgw = T.sgw; // (the dense path uses set 0; k_tile_write leaves the other clear)
            __hip_atomic_fetch_add(sgw + g / SUPER, (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sgw + T.sg_cap + g / SUPER, (uint64_t)(N - out0) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the totals of the tile-ordered dense path before its write kernel: summary[8] = matches, [9] = occurrences
__global__ void k_dense_totals(TileSpace T, uint64_t *summary) {
    uint64_t m = 0, occ = 0;
    const uint32_t n_super = (T.n_groups + SUPER - 1) / SUPER;
    for (uint32_t k = threadIdx.x; k < n_super; k += 256) { m += T.sgw[k]; occ += T.sgw[T.sg_cap + k] >> 32; }
    for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o); occ += __shfl_xor(occ, o); }
    __shared__ uint64_t red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = occ; }
    __syncthreads();
    if (threadIdx.x == 0) {
        summary[8] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        summary[9] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

hipError_t dense_tiles_verify(const DevAutomaton &A, const Segments &G, const Sink &hits, uint32_t hit_grid,
                              const DenseTiles &D, int key_mode, uint32_t lead, const uint8_t *d_hay, uint64_t len,
                              uint32_t *abort_flag, hipStream_t st) {
    const uint32_t grid = walk_hits_grid(hit_grid);
    if (A.max_shift) hipLaunchKernelGGL(k_dense_verify<true>, dim3(grid), dim3(256), 0, st, A, G, hits, hit_grid, D, key_mode, lead, d_hay, len, abort_flag);
    else hipLaunchKernelGGL(k_dense_verify<false>, dim3(grid), dim3(256), 0, st, A, G, hits, hit_grid, D, key_mode, lead, d_hay, len, abort_flag);
    return hipGetLastError();
}

hipError_t dense_tiles_main(const DevAutomaton &A, int key_mode, bool overlapping, const DenseTiles &D, const TileSpace &T,
                            uint32_t lead, uint32_t *abort_flag, uint64_t *summary, bool compact, hipStream_t st) {
    const uint32_t lookback = tile_lookback(A.max_len);
    if (lookback > MAX_LOOKBACK) return hipErrorInvalidValue;
    if (compact)
        hipLaunchKernelGGL((k_dense_main<128, true>), dim3(T.n_groups), dim3(128), dense_main_lds_compact(), st, A.rank_bits, A.max_len, key_mode,
                           overlapping ? 1 : 0, D, T, lookback, lead, abort_flag, HotMain{nullptr, TileSpace{}, 0, nullptr, 0});
    else
        hipLaunchKernelGGL((k_dense_main<DT_THREADS, false>), dim3(T.n_groups), dim3(DT_THREADS), dense_main_lds(lookback), st, A.rank_bits, A.max_len, key_mode,
                           overlapping ? 1 : 0, D, T, lookback, lead, abort_flag, HotMain{nullptr, TileSpace{}, 0, nullptr, 0});
    hipLaunchKernelGGL(k_dense_totals, dim3(1), dim3(256), 0, st, T, summary);
    return hipGetLastError();
}

hipError_t dense_tiles_write(const DevAutomaton &A, int key_mode, const TileSpace &T, const uint8_t *d_hay, acx_match_t *out,
                             uint64_t *summary, const uint32_t *zero_flag, uint64_t *host_out, uint32_t lead, const Segments &G,
                             uint64_t *seg_counts, const uint64_t *cp_blockpre, const uint8_t *cp_sub, hipStream_t st) {
    const CodePointTables cp{d_hay, cp_blockpre, cp_sub, A.pchars};
    const PostOut O{summary, (volatile uint64_t *)host_out, nullptr, 0, lead, 0, 0, nullptr, nullptr, 0, nullptr, 0};
    const HotWrite W{nullptr, nullptr, nullptr, 0, nullptr, 0};
    if (cp_blockpre)
        hipLaunchKernelGGL((k_tile_write<true, DT_GMAX, true>), dim3(T.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode,
                           A.by_rank, T, out, zero_flag, G, seg_counts, cp, O, W);
    else
        hipLaunchKernelGGL((k_tile_write<false, DT_GMAX, true>), dim3(T.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode,
                           A.by_rank, T, out, zero_flag, G, seg_counts, cp, O, W);
    return hipGetLastError();
}

// ---- the HOT pipeline's launches (kernels.hpp)
hipError_t hot_verify_main(const DevAutomaton &A, int key_mode, bool overlapping, const Segments &G, const TileSpace &S,
                           const uint32_t *hot_list, uint32_t n_hot, const uint32_t *ctl, uint32_t ovf_max, const DenseTiles &D,
                           const TileSpace &TD, uint32_t lead, const uint8_t *d_hay, uint64_t len, uint32_t *hot_abort,
                           uint64_t seq, uint32_t spec_bound, hipStream_t st) {
    // spec_bound != 0: a speculative launch (hot_groups_here) -- grids for spec_bound hot groups, n_hot read on the device;
    // ovf_max is then the caller's guess (the overflow lists' workgroups stride over their list: any number of them is right)
    const uint32_t lookback = tile_lookback(A.max_len);
    const uint32_t n_grid = spec_bound ? spec_bound : n_hot;
    if (lookback > MAX_LOOKBACK || n_grid == 0) return hipErrorInvalidValue;
    // workgroups per overflow list: by the fullest one (256 hits per step and workgroup: one step each up to 32 workgroups)
    const uint32_t ovb = ovf_max ? std::min<uint32_t>((ovf_max + 255) / 256, 32u) : (spec_bound ? 1u : 0u);
    const uint32_t grid = n_grid * HV_BLOCKS + OVF_LISTS * ovb;
    if (A.max_shift)
        hipLaunchKernelGGL(k_hot_verify<true>, dim3(grid), dim3(256), 0, st, A, G, S, hot_list, n_hot, ctl, ovb, lookback, D, key_mode,
                           lead, d_hay, len, hot_abort, spec_bound);
    else
        hipLaunchKernelGGL(k_hot_verify<false>, dim3(grid), dim3(256), 0, st, A, G, S, hot_list, n_hot, ctl, ovb, lookback, D, key_mode,
                           lead, d_hay, len, hot_abort, spec_bound);
    hipLaunchKernelGGL((k_dense_main<DT_THREADS, false>), dim3(n_grid * HOT_SUB), dim3(DT_THREADS), dense_main_lds(lookback), st, A.rank_bits, A.max_len,
                       key_mode, overlapping ? 1 : 0, D, TD, lookback, lead, hot_abort, HotMain{hot_list, S, seq, spec_bound ? ctl : nullptr, spec_bound});
    return hipGetLastError();
}

// the call's matches once the hot groups' counts are in (the supergroup words of the call's set), for a caller that sizes
// the output exactly: host_out = ONE line (publish_line), [1] = matches
__global__ void k_hot_totals(TileSpace S, uint64_t seq, volatile uint64_t *host_out, uint64_t pub) {
    uint64_t m = 0;
    const uint64_t *sgw = S.sgw + (seq & 1) * 2 * (uint64_t)S.sg_cap;
    const uint32_t n_super = (S.n_groups + SUPER - 1) / SUPER;
    for (uint32_t k = threadIdx.x; k < n_super; k += 256) m += sgw[k];
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o);
    __shared__ uint64_t red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const uint64_t mid[6] = {red[0] + red[1] + red[2] + red[3], 0, 0, 0, 0, 0};
    publish_line(host_out, threadIdx.x, pub, mid);
}
hipError_t hot_totals(const TileSpace &S, uint64_t seq, uint64_t *host_out, uint64_t pub, hipStream_t st) {
    hipLaunchKernelGGL(k_hot_totals, dim3(1), dim3(256), 0, st, S, seq, (volatile uint64_t *)host_out, pub);
    return hipGetLastError();
}

// the records of the hot groups (one workgroup per dense group), then the sparse path's write kernel again (pass 1):
// it places every other group with the hot groups' counts in and publishes the totals ([5] = *hot_abort, [7] = pub)
hipError_t hot_write(const DevAutomaton &A, int key_mode, const TileSpace &S, const TileSpace &TD, const uint32_t *hot_list,
                     uint32_t n_hot, uint32_t lead, const uint8_t *d_hay, acx_match_t *out, uint64_t *summary,
                     const uint32_t *abort_flag, const uint32_t *hot_abort, uint64_t *host_out, uint64_t seq, uint64_t pub,
                     const Segments &G, uint64_t *seg_counts, const uint64_t *cp_blockpre, const uint8_t *cp_sub, uint32_t spec_bound,
                     hipStream_t st) {
    // (spec_bound != 0: a speculative launch, abort_flag = the call's control block -- hot_groups_here)
    const CodePointTables cp{d_hay, cp_blockpre, cp_sub, A.pchars};
    const uint32_t *sc = spec_bound ? abort_flag : nullptr;
    const PostOut O{summary, (volatile uint64_t *)host_out, nullptr, seq, lead, pub, 1, hot_abort, nullptr, 0, sc, spec_bound};
    const HotWrite W{hot_list, (const uint64_t *)TD.trecs, TD.btot, TD.n_groups, sc, spec_bound};
    const HotWrite W0{nullptr, nullptr, nullptr, 0, nullptr, 0};
    if (spec_bound) n_hot = spec_bound;
    if (cp_blockpre) {
        hipLaunchKernelGGL((k_tile_write<true, DT_GMAX, true, true>), dim3(n_hot * HOT_SUB), dim3(WRITE_THREADS), 0, st, A.rank_bits,
                           key_mode, A.by_rank, S, out, abort_flag, G, seg_counts, cp, O, W);
        hipLaunchKernelGGL((k_tile_write<true, GROUP_MAX>), dim3(S.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode, A.by_rank,
                           S, out, abort_flag, G, seg_counts, cp, O, W0);
    } else {
        hipLaunchKernelGGL((k_tile_write<false, DT_GMAX, true, true>), dim3(n_hot * HOT_SUB), dim3(WRITE_THREADS), 0, st, A.rank_bits,
                           key_mode, A.by_rank, S, out, abort_flag, G, seg_counts, cp, O, W);
        hipLaunchKernelGGL((k_tile_write<false, GROUP_MAX>), dim3(S.n_groups), dim3(WRITE_THREADS), 0, st, A.rank_bits, key_mode, A.by_rank,
                           S, out, abort_flag, G, seg_counts, cp, O, W0);
    }
    return hipGetLastError();
}

// lead (non-continuation) bytes among the first n (0 .. 16) bytes of a 16-byte slice
__device__ __forceinline__ uint32_t leads_in_first(const uint4 v, uint32_t n) {
    const uint64_t v0 = n >= 8 ? ~0ull : (n ? ~0ull >> (8 * (8 - n)) : 0);
    const uint64_t v1 = n > 8 ? (n >= 16 ? ~0ull : ~0ull >> (8 * (16 - n))) : 0;
    return lead_in_word(((uint64_t)v.y << 32) | v.x, v0) + lead_in_word(((uint64_t)v.w << 32) | v.z, v1);
}

// ---------------------------------------------------------------------------
// K0: the whole call in ONE workgroup, for small haystacks
// ---------------------------------------------------------------------------
// A short haystack is launch-latency bound: the general pipeline is ~10 dependent device
// operations (45-60 us) however little there is to scan.  Up to SMALL_MAX_LEN bytes, one
// workgroup does everything: stage the haystack in LDS, enumerate the occurrences by an
// ANCHORED walk from every position (one thread per start: follow trie edges only -- a
// transition that does not go one level deeper is a failure transition, BFS ids make that one
// comparison -- and report the patterns that end on the way: 2-3 dependent loads per
// position on text), rank-sort them by key in LDS, resolve the match kind, convert to code
// points, write the final records.  `hay`, `out` and `res` may live in pinned host memory
// (zero-copy): the host-memory entry point then costs one launch and one poll.
// seq == 0 (device-resident callers, read behind a stream synchronisation): out[] = acx_match_t records, res[0] =
// matches written, res[1] = 0, or 1: too many occurrences, nothing written.  seq != 0 (the host polls -- one launch and
// one PCIe read instead of a launch and a stream synchronisation, whose wake-up alone costs 5-10 us): the result is the
// 64-byte line described at k0_publish_line below.
// How the occurrences are found: MODE 0 = the walk over the tables in global memory, MODE 1 (LT) and MODE 2 (DC) below.
// LT (small automata: the dense table is at most K0_LT_ENTRIES words, at most K0_LT_IDS states and patterns, patterns
// shorter than 256 bytes): the table, the levels, own1 and the
// ranks are staged in LDS while the haystack crosses PCIe (the loads are in flight together: the staging costs no
// time of its own), and the anchored walks run at LDS latency -- a walk of depth 8 is ~1 us instead of ~5 (eight
// dependent L2 round trips), which was most of what a call on a 75-byte haystack with 4-5 matches cost beyond the launch.
constexpr uint32_t K0_LT_ENTRIES = 8192, K0_LT_IDS = 2048;
// The polled result of a K0 call (seq != 0): ONE aligned 64-byte line of coherent pinned host memory, written by ONE store
// instruction (four lanes x 16 bytes): [ seq | matches + too dense << 32 | the first K0_LINE_MATCHES matches, packed
// | seq ^ check(the six words in the middle) ].  Every separate write to host memory the kernel had to wait for before it may publish -- the records, then the
// totals behind a system-scope fence, then the number -- was a PCIe round trip of its own (measured, rocprofv3: 6.8 us
// for the kernel without matches, 8.8 with one, 12.9 with four on 75-byte haystacks); a line that carries its own
// sequence number in front and a checksum of its middle keyed with that number at the end needs none: the host takes it
// when the first word shows the call's number and the last one agrees with the middle as the host read it (no assumption
// about the order in which the line's four 16-byte pieces become visible).  A packed match:
// pattern : 32 | start : 16 | end - 1 : 16 (a K0 haystack has at most 65 536 bytes; a match is not empty).  More matches than the line holds go,
// packed, to out[] first, behind a system-scope fence.
constexpr uint32_t K0_LINE_MATCHES = ACX_K0_LINE_MATCHES;
static_assert(K0_LINE_MATCHES + 3 == K0_LINE_WORDS, "seq, totals, matches, seq");
// (tot: the call's matches, the first of them in pk[]; word1: the line's second word -- the count, flags, the hash of out[])
__device__ __forceinline__ void k0_publish_line(uint64_t *res, uint32_t t, uint64_t seq, uint64_t word1, const uint64_t *pk,
                                                uint32_t tot) {
    const uint32_t inl = tot < K0_LINES_MATCHES ? tot : K0_LINES_MATCHES, nl = k0_result_lines(tot);
    if (t >= 4 * nl) return;
    uint64_t w[2] = {0, 0};
    for (uint32_t L = 0; L < nl; L++) {
        uint64_t mid[6]; // words 1 .. 6 of line L, the same in every lane
        const uint32_t first = L ? K0_LINE_MATCHES + (L - 1) * K0_MORE_MATCHES : 0;
#pragma unroll
        for (uint32_t i = 0; i < 6; i++) {
            const uint32_t m = L ? first + i : i - 1; // (line 0: word 1 is word1, matches from word 2 on)
            mid[i] = L == 0 && i == 0 ? word1 : m < inl ? pk[m] : 0;
        }
        // (the same in every lane: in scalar registers the check's twelve 64-bit multiplications are scalar instructions)
#pragma unroll
        for (uint32_t i = 0; i < 6; i++)
            mid[i] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(mid[i] >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)mid[i]); // (readfirstlane returns an int: no sign extension)
        const uint64_t last = seq ^ k0_line_check(mid); // (the host accepts the line on word 0 AND this word: kernels.hpp)
        if ((t >> 2) == L) {
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                const uint32_t i = 2 * (t & 3) + k;
                w[k] = i == 0 ? seq : i == 7 ? last : mid[i - 1 > 5 ? 5 : i - 1];
            }
        }
    }
    ((ulonglong2 *)res)[t] = make_ulonglong2(w[0], w[1]);
}
// MODE 2, direct comparison (a handful of short patterns on a short haystack: at most K0_DC_PATTERNS patterns of at most
// 16 bytes, haystack bytes x patterns <= K0_DC_WORK): one thread per (position, pattern) compares the pattern with the
// 16-byte window at the position -- two masked 64-bit XORs -- instead of one thread per position walking the trie level
// by level.  A walk is a chain of dependent lookups as long as the match (15 for "arbitrarymonkey" in the reference's
// short benchmark), and between calls that are 10-20 us apart the shader clock idles low: a dependent LDS round trip
// costs ~0.3 us there (measured: the kernel took 6.8 us without matches, 8.8 with one 4-byte match, 12.9 with four).
// MODE 0: the automaton's tables in global memory; MODE 1 (LT above): staged in LDS.
constexpr uint32_t K0_DC_PATTERNS = 64, K0_DC_WORK = 4096;
// MODE 3 (round 5), the prefilter: haystacks of up to SMALL_PF_MAX_LEN bytes (64 KiB: a packed match carries 16-bit
// offsets) over any set K1b's tables exist for and that has no 1- or 2-byte patterns.  The occurrences are found
// the way K1b finds them -- level 1 on every pair of positions (one 8-byte read of the {X, Y} table per pair: from global
// memory here, 128 KiB, L2-resident; the haystack is in LDS), the exact prefix table for the survivors, ONE 16-byte
// pattern-info load per candidate -- instead of an anchored walk from every position, which is a chain of dependent table
// gathers per position and what a call on a large automaton cost: 16 KiB took 38 us, and beyond 16 KiB the call went to
// the three-kernel pipeline (64 KiB: 63 us).  A position costs one independent gather, whatever the automaton's size.
// tables = false (the RESIDENT kernel's calls after its first): the automaton's LDS images (classes, LT's tables, DC's
// patterns) are those of the call before.
// npre != 0 (the resident kernel): the haystack's first npre bytes (a multiple of 16, or all of it) came with the poll --
// thread 1 + j holds bytes [16 j, 16 j + 16) in `pre`.
template <int MODE>
__device__ __forceinline__ void k0_call(const DevAutomaton &A, const uint8_t *hay, uint32_t len, int key_mode,
                                        int overlapping, int codepoints, acx_match_t *out, uint64_t *res, uint64_t seq,
                                        bool tables, uint4 pre = make_uint4(0, 0, 0, 0), uint32_t npre = 0) {
    constexpr bool LT = MODE == 1, DC = MODE == 2, PF = MODE == 3;
    constexpr uint32_t MAXLEN = PF ? SMALL_PF_MAX_LEN : SMALL_MAX_LEN; // bytes the haystack's LDS image holds
    __shared__ __attribute__((aligned(16))) uint8_t sh[MAXLEN + 32];
    __shared__ uint32_t ltab[LT ? K0_LT_ENTRIES : 1], lown1[LT ? K0_LT_IDS : 1], lrank[LT ? K0_LT_IDS : 1], llevel[LT ? 258 : 1];
    __shared__ uint64_t plo[DC ? K0_DC_PATTERNS : 1], phi[DC ? K0_DC_PATTERNS : 1]; // DC: the patterns' bytes, masked to their length
    __shared__ uint32_t pl[DC ? K0_DC_PATTERNS : 1], prk[DC ? K0_DC_PATTERNS : 1];  // their lengths and ranks
    __shared__ uint8_t cls[256];
    __shared__ uint4 occ[SMALL_MAX_OCC]; // {key lo, key hi, pid, pattern length}
    __shared__ uint16_t order[SMALL_MAX_OCC];
    __shared__ uint8_t syn[SMALL_MAX_OCC], acc[SMALL_MAX_OCC];
    __shared__ uint32_t cpre[MAXLEN / 16 + 1]; // code points before every 16-byte slice
    __shared__ uint32_t nocc, s_total, rest_hash;
    // the records, assembled here and written as a flat run of dwords: `out` may be pinned HOST memory, where every
    // store instruction's every lane is a transaction of its own -- 4 matches written field by field were 12 partial
    // writes over PCIe (measured: the kernel took 12 us on 75-byte haystacks with 4 matches, ~6 without matches)
    __shared__ __attribute__((aligned(16))) uint32_t img[SMALL_MAX_OCC * 6];
    using scan_t = rocprim::block_scan<uint32_t, 1024>;
    __shared__ typename scan_t::storage_type scan_tmp;
    const uint32_t t = threadIdx.x;
    // (haystacks of at most 1 KiB -- one position per thread, at most 64 slices of 16 bytes: the prefix sums below are
    // one wave's shuffles instead of block scans, and an LT walk has its 16 class bytes in registers before it starts)
    const bool tiny = len <= 1024;
    if (t == 0) { nocc = 0; rest_hash = 0; }
    if (tables && t < 256) cls[t] = A.classes[t];
    if (npre && t >= 1 && t < 64 && 16 * (t - 1) < npre) *(uint4 *)(sh + 16 * (t - 1)) = pre;
    if ((PF || npre) && ((uintptr_t)hay & 15) == 0) { // (16 bytes per lane: the aligned block that holds the haystack's last byte is all
        // readable -- the prefilter's callers' buffers and the mailbox are; the pieces of a wave are one request each)
        for (uint32_t i = npre + 16 * t; i < len; i += 16 * 1024) *(uint4 *)(sh + i) = *(const uint4 *)(hay + i);
        if (PF) __syncthreads();
    } else {
        for (uint32_t i = npre + t; i < len; i += 1024) sh[i] = hay[i];
    }
    if (PF && t < 32) sh[len + t] = 0; // (the bytes behind the haystack are read as part of the last windows)
    if (LT && tables) {
        const uint32_t ne = A.n_states << A.stride2;
        for (uint32_t i = t; i < ne; i += 1024) ltab[i] = A.table[i];
        for (uint32_t i = t; i < A.n_states; i += 1024) lown1[i] = A.own1[i];
        for (uint32_t i = t; i < (uint32_t)A.n_patterns; i += 1024) lrank[i] = A.rank[i];
        if (t < A.max_len + 2) llevel[t] = A.level_start[t];
    }
    if (DC && tables) {
        if (t < (uint32_t)A.n_patterns) {
            const uint32_t L = A.plen[t];
            uint64_t b0, b1;
            __builtin_memcpy(&b0, A.pat_blob + A.pat_off[t], 8); // (pat_blob is padded by 16 bytes)
            __builtin_memcpy(&b1, A.pat_blob + A.pat_off[t] + 8, 8);
            plo[t] = L >= 8 ? b0 : b0 & ((1ull << (8 * L)) - 1);
            phi[t] = L > 8 ? (L >= 16 ? b1 : b1 & ((1ull << (8 * (L - 8))) - 1)) : 0;
            pl[t] = L;
            prk[t] = A.rank[t];
        }
    }
    __syncthreads();
    if constexpr (DC) {
        const uint32_t np = (uint32_t)A.n_patterns, work = len * np;
        const uint64_t *sh64 = (const uint64_t *)sh;
        for (uint32_t idx = t; idx < work; idx += 1024) {
            const uint32_t pos = idx / np, pid = idx - pos * np;
            const uint32_t L = pl[pid];
            if (pos + L > len) continue;
            const uint32_t q = pos >> 3, r = (pos & 7) * 8;
            const uint64_t a = sh64[q], b = sh64[q + 1], c = sh64[q + 2];
            const uint64_t w0 = r ? (a >> r) | (b << (64 - r)) : a, w1 = r ? (b >> r) | (c << (64 - r)) : b;
            const uint64_t m0 = L >= 8 ? ~0ull : (1ull << (8 * L)) - 1;
            const uint64_t m1 = L > 8 ? (L >= 16 ? ~0ull : (1ull << (8 * (L - 8))) - 1) : 0;
            if ((((w0 & m0) ^ plo[pid]) | ((w1 & m1) ^ phi[pid])) != 0) continue;
            const uint32_t rk = prk[pid];
            const uint64_t key = key_mode == 0   ? ((uint64_t)(pos + L) << A.rank_bits) | rk
                                 : key_mode == 1 ? ((uint64_t)pos << A.rank_bits) | pid
                                                 : ((uint64_t)pos << A.rank_bits) | rk;
            const uint32_t slot = atomicAdd(&nocc, 1u);
            if (slot < SMALL_MAX_OCC) occ[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, L);
        }
    }
    if constexpr (PF) {
        // 8 haystack bytes at any position of the LDS image (two aligned reads)
        const uint64_t *sh64 = (const uint64_t *)sh;
        auto ld64 = [&](uint32_t p_) -> uint64_t {
            const uint32_t q_ = p_ >> 3, r_ = (p_ & 7) * 8;
            const uint64_t a_ = sh64[q_], b_ = sh64[q_ + 1];
            return r_ ? (a_ >> r_) | (b_ << (64 - r_)) : a_;
        };
        const uint32_t Qf = A.filter_q, GB = Qf - 1;
        const uint32_t gmask = GB >= 4 ? 0xFFFFFFFFu : (1u << (8 * GB)) - 1u;
        const bool any = len >= A.k1b_min_len;
        const uint32_t last = any ? len - A.k1b_min_len : 0; // the last position a pattern can start at
        // an occurrence of candidate `cand` with its anchor at p?  (the verification of k_tile_main; the haystack: the LDS image)
        auto candidate = [&](uint32_t p_, uint32_t cand, uint64_t w0, uint64_t w1) {
            uint32_t rk;
            uint64_t ps;
            const uint32_t L = A.max_shift ? verify_candidate<true>(A, (const uint8_t *)sh, len, p_, cand, w0, w1, len - p_, p_, &rk, &ps)
                                           : verify_candidate<false>(A, (const uint8_t *)sh, len, p_, cand, w0, w1, len - p_, p_, &rk, &ps);
            if (!L) return;
            const uint32_t pid = cand & CODE_PID_MASK;
            const uint64_t key = key_mode == 0   ? ((ps + L) << A.rank_bits) | rk
                                 : key_mode == 1 ? (ps << A.rank_bits) | pid
                                                 : (ps << A.rank_bits) | rk;
            const uint32_t slot = atomicAdd(&nocc, 1u);
            if (slot < SMALL_MAX_OCC) occ[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, L);
        };
        auto survivor = [&](uint32_t p_) { // a level-1 survivor: the exact prefix table, then its candidates
            const uint64_t w0 = ld64(p_), w1 = ld64(p_ + 8);
            uint32_t code = prefix_code(A.ptab, A.ptab_log2, A.filter_q2, w0);
            if (code == HIT_NONE) return;
            if (code & HIT_LIST) {
                const uint32_t li = code & ~HIT_LIST, nc = A.blist[li];
                for (uint32_t k = 0; k < nc; k++) candidate(p_, A.blist[li + 1 + k], w0, w1);
            } else {
                candidate(p_, code, w0, w1);
            }
        };
        // level 1 (automaton.hpp): positions j (even) and j + 1 share the row of the gram at j + 1 -- four pairs per
        // thread and step, their rows requested together (eight: 64 KiB 39.5 -> 50.6 us per call, measured)
        const uint2 *xy = (const uint2 *)A.filterA;
        constexpr uint32_t PPS = 4; // pairs per step
        for (uint32_t j0 = 2 * PPS * t; any && j0 <= last; j0 += 2 * PPS * 1024) {
            uint64_t w[PPS];
            uint2 e[PPS];
#pragma unroll
            for (uint32_t k = 0; k < PPS; k++) {
                w[k] = ld64(j0 + 2 * k); // bytes j .. j + 7 of the pair at j = j0 + 2 k
                const uint32_t W = (uint32_t)(w[k] >> 8) & gmask;
                e[k] = xy[filter_entry(filter_hash(W))];
            }
#pragma unroll
            for (uint32_t k = 0; k < PPS; k++) {
                const uint32_t j = j0 + 2 * k;
                const uint32_t W = (uint32_t)(w[k] >> 8) & gmask;
                const uint32_t gate = (e[k].x >> (W & 31)) & 1u;
                const bool s0 = gate && ((e[k].x >> ((uint32_t)w[k] & 31)) & 1u) && j <= last;
                const bool s1 = gate && ((e[k].y >> ((uint32_t)(w[k] >> (8 * Qf)) & 31)) & 1u) && j + 1 <= last;
                if (s0) survivor(j);
                if (s1) survivor(j + 1);
            }
        }
    }
    // ---- all occurrences: anchored walk from every position
    for (uint32_t pos = t; !DC && !PF && pos < len; pos += 1024) {
        uint32_t s = 0;
        uint64_t c0 = 0, c1 = 0; // LT, tiny: the classes of the 16 bytes at pos (independent LDS reads, all in flight)
        if (LT && tiny) {
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint64_t c = pos + k < len ? cls[sh[pos + k]] : 0;
                if (k < 8) c0 |= c << (8 * k); else c1 |= c << (8 * (k - 8));
            }
        }
        for (uint32_t d = 0; pos + d < len;) {
            uint32_t id, own;
            if constexpr (LT) {
                const uint32_t cb = !tiny || d >= 16 ? cls[sh[pos + d]] : (uint32_t)((d < 8 ? c0 >> (8 * d) : c1 >> (8 * (d - 8))) & 0xFF);
                const uint32_t e = ltab[(s << A.stride2) + cb];
                id = e & ID_MASK;
                d++;
                if (id < llevel[d]) break;
                own = e & FLAG_OWN;
            } else if (A.table) {
                const uint32_t e = A.table[((size_t)s << A.stride2) + cls[sh[pos + d]]];
                id = e & ID_MASK;
                d++;
                if (id < A.level_start[d]) break; // shallower than d: a failure transition, not an edge
                own = e & FLAG_OWN;
            } else { // compressed automaton: trie edges only, which is all an anchored walk follows
                id = trie_child(A, s, sh[pos + d]);
                d++;
                if (!id) break;
                own = A.sflags[id] & 1u;
            }
            s = id;
            if (!own) continue;
            const uint32_t one = LT ? lown1[s] : A.own1[s];
            uint32_t b = 0, en = 1;
            if (one == OWN1_MANY) { b = A.own_off[s]; en = A.own_off[s + 1]; }
            for (uint32_t k = b; k < en; k++) { // the patterns that are exactly hay[pos, pos + d)
                const uint32_t pid = one == OWN1_MANY ? A.own_pid[k] : one;
                const uint32_t rk = key_mode == 1 ? 0u : (LT ? lrank[pid] : A.rank[pid]);
                const uint64_t key = key_mode == 0   ? ((uint64_t)(pos + d) << A.rank_bits) | rk
                                     : key_mode == 1 ? ((uint64_t)pos << A.rank_bits) | pid
                                                     : ((uint64_t)pos << A.rank_bits) | rk;
                const uint32_t slot = atomicAdd(&nocc, 1u);
                if (slot < SMALL_MAX_OCC) occ[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, d);
            }
        }
    }
    __syncthreads();
    const uint32_t n = nocc;
    if (n > SMALL_MAX_OCC) { // dense: the general pipeline takes the call
        if (seq) k0_publish_line(res, t, seq, 1ull << 32, nullptr, 0);
        else if (t == 0) { res[0] = 0; res[1] = 1; }
        return;
    }
    // ---- a polled call on at most 1 KiB with at most 64 occurrences (the reference's benchmark loop): WAVE 0 finishes it
    // alone, an occurrence per lane -- no barrier of the workgroup between here and the result (the general way below is
    // five of them and four passes through LDS: 2.3 of a call's 4.2 us on the device, measured in the resident kernel)
    if (seq && tiny && n <= 64) {
        if (t >= 64) return;
        const uint4 mine_occ = t < n ? occ[t] : make_uint4(~0u, ~0u, 0u, 0u);
        // rank among the keys (unique), then every occurrence to the lane of its rank
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)mine_occ.x, (int)j), jh = (uint32_t)__builtin_amdgcn_readlane((int)mine_occ.y, (int)j);
            r += (jh < mine_occ.y || (jh == mine_occ.y && jl < mine_occ.x)) ? 1u : 0u;
        }
        const uint32_t to = (t < n ? r : 63u) << 2; // (lanes without an occurrence: n < 64, lane 63 is none of the n ranks' -- or is, and is overwritten)
        uint4 v;
        v.x = (uint32_t)__builtin_amdgcn_ds_permute((int)to, (int)mine_occ.x);
        v.y = (uint32_t)__builtin_amdgcn_ds_permute((int)to, (int)mine_occ.y);
        v.z = (uint32_t)__builtin_amdgcn_ds_permute((int)to, (int)mine_occ.z);
        v.w = (uint32_t)__builtin_amdgcn_ds_permute((int)to, (int)mine_occ.w);
        const uint32_t x = (uint32_t)(((((uint64_t)v.y << 32) | v.x)) >> A.rank_bits); // (an offset of at most 1 024)
        uint32_t s = key_mode == 0 ? x - v.w : x, e = key_mode == 0 ? x : x + v.w;
        if (t >= n) s = e = 0;
        // match kind: one greedy pass over the sorted occurrences (what the sync points' chains below do piecewise)
        uint64_t taken = 0;
        if (overlapping) {
            taken = n == 64 ? ~0ull : (1ull << n) - 1;
        } else {
            uint32_t pos = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)j), ej = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)j);
                if (sj >= pos) { taken |= 1ull << j; pos = ej; }
            }
        }
        const bool mine1 = (taken >> t) & 1u;
        const uint32_t dst = __popcll(taken & ((1ull << t) - 1)), tot = __popcll(taken);
        if (codepoints) { // (src/lib.rs:73-88: lead bytes before the offset -- lane q counts slice q, the prefix by shuffles)
            const uint32_t leads = t * 16 < len ? leads_in_first(*(const uint4 *)(sh + t * 16), len - t * 16 < 16 ? len - t * 16 : 16) : 0;
            uint32_t incl = leads;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(incl, o);
                if ((int)t >= o) incl += u;
            }
            const uint32_t excl = incl - leads, all = __shfl(incl, 63);
            const uint32_t bs = __shfl(excl, (int)((s >> 4) & 63)), be = __shfl(excl, (int)((e >> 4) & 63));
            const uint32_t cs = ((s >> 4) < 64 ? bs : all) + leads_in_first(*(const uint4 *)(sh + (s & ~15u)), s & 15u);
            const uint32_t ce = ((e >> 4) < 64 ? be : all) + leads_in_first(*(const uint4 *)(sh + (e & ~15u)), e & 15u);
            s = cs; e = ce;
        }
        uint64_t *pk = (uint64_t *)img;
        if (mine1) pk[dst] = (uint64_t)v.z | ((uint64_t)s << 32) | ((uint64_t)(e - 1) << 48);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t word1 = tot;
        if (tot > K0_LINES_MATCHES) { // (uniform) out[] and the lines are separate writes to host memory: k0_rest_mix
            uint32_t hx = 0;
            if (t < tot - K0_LINES_MATCHES) {
                const uint64_t u = pk[K0_LINES_MATCHES + t];
                ((uint64_t *)out)[t] = u;
                hx = k0_rest_mix(u, t, seq);
            }
            for (int o = 32; o > 0; o >>= 1) hx ^= __shfl_xor(hx, o);
            __threadfence_system();
            word1 |= (uint64_t)hx << K0_REST_HASH_SHIFT;
        }
        k0_publish_line(res, t, seq, word1, pk, tot);
        return;
    }
    // ---- rank sort (keys are unique: position + a tie-break that is unique per pattern)
    for (uint32_t i = t; i < n; i += 1024) {
        const uint64_t ki = ((uint64_t)occ[i].y << 32) | occ[i].x;
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) r += ((((uint64_t)occ[j].y << 32) | occ[j].x) < ki) ? 1u : 0u;
        order[r] = (uint16_t)i;
    }
    __syncthreads();
#define K0_SPAN(I, S, E)                                                                          \
    {                                                                                             \
        const uint4 v_ = occ[order[(I)]];                                                         \
        const uint64_t x_ = ((((uint64_t)v_.y << 32) | v_.x) >> A.rank_bits);                     \
        if (key_mode == 0) { E = x_; S = x_ - v_.w; } else { S = x_; E = x_ + v_.w; }             \
    }
    // ---- match kind: sync points, then every sync point walks its greedy chain
    uint32_t mine = 0;
    if (overlapping) {
        mine = t < n ? 1u : 0u;
    } else {
        if (t < n) {
            uint64_t s, e, mx = 0;
            K0_SPAN(t, s, e)
            (void)e;
            for (uint32_t j = t; j > 0;) {
                j--;
                uint64_t sj, ej;
                K0_SPAN(j, sj, ej)
                mx = max(mx, ej);
                // sorted by end: the previous end is the maximum; sorted by start: nothing that
                // starts max_len or more before s can end beyond it
                if (key_mode == 0 || sj + A.max_len <= s) break;
            }
            syn[t] = mx <= s ? 1 : 0;
        }
        __syncthreads();
        if (t < n && syn[t]) {
            uint64_t s, pos;
            K0_SPAN(t, s, pos)
            acc[t] = 1;
            for (uint32_t j = t + 1; j < n && !syn[j]; j++) {
                uint64_t sj, ej;
                K0_SPAN(j, sj, ej)
                const bool take = sj >= pos;
                acc[j] = take;
                if (take) pos = ej;
            }
        }
        __syncthreads();
        mine = t < n ? acc[t] : 0u;
    }
    // ---- byte offset -> code point index (src/lib.rs:73-88): lead bytes before the offset
    // (a slice is one aligned 16-byte LDS read and two popcounts -- rounds 1-4 read it byte by byte)
    if (codepoints && PF && len > 16384) {
        // (more than 1024 slices: SPT consecutive slices per thread, a block scan of the threads' sums)
        constexpr uint32_t SPT = MAXLEN / 16 / 1024 > 0 ? MAXLEN / 16 / 1024 : 1;
        uint32_t ls[SPT], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < SPT; k++) {
            const uint32_t o = (t * SPT + k) * 16;
            ls[k] = o < len ? leads_in_first(*(const uint4 *)(sh + o), len - o < 16 ? len - o : 16) : 0;
            sum += ls[k];
        }
        uint32_t before = 0;
        scan_t().exclusive_scan(sum, before, 0u, scan_tmp);
#pragma unroll
        for (uint32_t k = 0; k < SPT; k++) { cpre[t * SPT + k] = before; before += ls[k]; }
        if (t == 1023) cpre[1024 * SPT] = before;
        __syncthreads();
    } else if (codepoints) {
        uint32_t leads = 0;
        if (t * 16 < len) leads = leads_in_first(*(const uint4 *)(sh + t * 16), len - t * 16 < 16 ? len - t * 16 : 16);
        if (tiny) {
            if (t < 64) {
                uint32_t incl = leads;
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(incl, o);
                    if ((int)t >= o) incl += v;
                }
                cpre[t] = incl - leads;
                if (t == 63) cpre[64] = incl;
            }
        } else {
            uint32_t before = 0;
            scan_t().exclusive_scan(leads, before, 0u, scan_tmp);
            cpre[t] = before;
            if (t == 1023) cpre[1024] = before + leads;
        }
        __syncthreads();
    }
    uint32_t dst = 0, total = 0;
    if (n <= 64) { // (mine != 0 only in wave 0)
        if (t < 64) {
            uint32_t incl = mine;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if ((int)t >= o) incl += v;
            }
            dst = incl - mine;
            total = __shfl(incl, 63);
        }
    } else {
        scan_t().exclusive_scan(mine, dst, 0u, scan_tmp);
        total = dst + mine; // (thread 1023's)
    }
    if (mine) {
        const uint4 v = occ[order[t]];
        uint64_t s, e;
        K0_SPAN(t, s, e)
        if (codepoints) {
            const uint32_t cs = cpre[s >> 4] + leads_in_first(*(const uint4 *)(sh + (s & ~15ull)), (uint32_t)s & 15u);
            const uint32_t ce = cpre[e >> 4] + leads_in_first(*(const uint4 *)(sh + (e & ~15ull)), (uint32_t)e & 15u);
            s = cs; e = ce;
        }
        if (seq) {
            ((uint64_t *)img)[dst] = (uint64_t)v.z | (s << 32) | ((e - 1) << 48); // (a match is not empty: end - 1 fits 16 bits up to 64 KiB)
        } else {
            uint32_t *d = img + dst * 6;
            d[0] = v.z; d[1] = 0;
            d[2] = (uint32_t)s; d[3] = (uint32_t)(s >> 32); d[4] = (uint32_t)e; d[5] = (uint32_t)(e >> 32);
        }
    }
    if (n <= 64 ? t == 63 : t == 1023) s_total = total;
    __syncthreads();
    const uint32_t tot = s_total;
    if (seq) {
        const uint64_t *pk = (const uint64_t *)img;
        uint64_t word1 = tot;
        if (tot > K0_LINES_MATCHES) { // (uniform)
            // out[] and the line are separate writes to host memory: the line says what out[] must hold (k0_rest_mix)
            uint32_t hx = 0;
            for (uint32_t k = t; k < tot - K0_LINES_MATCHES; k += 1024) {
                const uint64_t v = pk[K0_LINES_MATCHES + k];
                ((uint64_t *)out)[k] = v;
                hx ^= k0_rest_mix(v, k, seq);
            }
            for (int o = 32; o > 0; o >>= 1) hx ^= __shfl_xor(hx, o);
            if ((t & 63) == 0 && hx) atomicXor(&rest_hash, hx);
            // (the waves that wrote: a system-scope release is a write-back of the L2 per wave -- sixteen of them were most of
            // what a sixth match cost: 16 KiB with ten matches 22.7 us, 8 KiB with five 13.3, measured)
            if (t < tot - K0_LINES_MATCHES) __threadfence_system();
            __syncthreads();
            word1 |= (uint64_t)rest_hash << K0_REST_HASH_SHIFT;
        }
        k0_publish_line(res, t, seq, word1, pk, tot);
    } else {
        for (uint32_t k = t; k < tot * 6; k += 1024) ((uint32_t *)out)[k] = img[k];
        if (t == 0) *(ulonglong2 *)res = make_ulonglong2(tot, 0); // res[0] = matches, res[1] = 0: one store
    }
#undef K0_SPAN
}

template <int MODE>
__global__ __launch_bounds__(1024) void k0_small(DevAutomaton A, const uint8_t *__restrict__ hay,
                                                 uint32_t len, int key_mode, int overlapping,
                                                 int codepoints, acx_match_t *out, uint64_t *res, uint64_t seq) {
    k0_call<MODE>(A, hay, len, key_mode, overlapping, codepoints, out, res, seq, true);
}

// The RESIDENT K0 (round 6): the same workgroup, launched once and fed through a MAILBOX in coherent pinned host memory --
// word 0 = k0_mailbox_word(call number, length, code points / quit) (kernels.hpp), the haystack K0_MAILBOX_HAY bytes
// behind it -- so that a call costs a poll on either side instead of a kernel launch (the reference's own benchmark loop
// calls once per 75-byte haystack: /root/reference/benchmarks/test_comparison.py:113-124).  Thread 0 polls word 0 with
// system-scope loads (one PCIe read each); the host writes the haystack first and the word last (one aligned 8-byte store),
// the workgroup reads the haystack behind an acquire fence.  The result goes where a launched K0's goes (the line `res`
// under the call's number; `out`).  The kernel leaves when told to (K0_MAILBOX_QUIT), after idle_ticks of the 100 MHz
// clock without a call, or life_ticks after its launch whatever happens (other streams may share its hardware queue:
// nobody waits for it longer than that) -- and says so in *status (= epoch, one release store behind everything it wrote):
// the host launches the next one when a call finds the kernel gone; a call posted while the kernel was leaving is taken
// by that launch (seq: the last call before this launch; the call the kernel takes carries seq + 1).
// (one call of the resident kernel: a function of its own, the automaton's description read from HBM where it needs it --
// inlined into the kernel's loop, the description's sixty fields stay in registers around the loop and the body spills)
template <int MODE>
__device__ __noinline__ void k0_resident_call(const DevAutomaton *A, const uint8_t *hay, uint32_t len, int key_mode,
                                              int overlapping, int codepoints, acx_match_t *out, uint64_t *res, uint64_t seq,
                                              bool tables, uint4 pre, uint32_t npre) {
    // (a function's arguments arrive in vector registers: said to be uniform, the description's fields are scalar loads and
    // what follows from them scalar registers and branches -- without this MODE 3 took 124 vector registers and spilled)
    auto uni32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](uint64_t v) { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); };
    k0_call<MODE>(*(const DevAutomaton *)uni64((uint64_t)A), (const uint8_t *)uni64((uint64_t)hay), uni32(len), (int)uni32((uint32_t)key_mode),
                  (int)uni32((uint32_t)overlapping), (int)uni32((uint32_t)codepoints), (acx_match_t *)uni64((uint64_t)out),
                  (uint64_t *)uni64((uint64_t)res), uni64(seq), uni32(tables ? 1u : 0u) != 0, pre, uni32(npre));
}

// a 64-bit value that is the same in every lane, into scalar registers (readfirstlane returns an int: the low half must not
// be sign-extended over the high one)
__device__ __forceinline__ uint64_t uniform64(uint32_t lo, uint32_t hi) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(lo);
}
// 16 bytes of host memory as they are NOW (system scope: no cache of the device answers)
__device__ __forceinline__ uint4 load16_system(const void *p) {
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// The poll: wave 0 reads the mailbox's first lines in ONE instruction -- lane 0 the word and the check, lane 1 + j the
// haystack's bytes [16 j, 16 j + 16) -- so that a haystack of up to K0_MAILBOX_INLINE bytes is there when the word is
// (a dependent read of host memory is another PCIe round trip, ~1.6 us).  The lanes' reads are not ordered among
// themselves: the bytes are taken when their hash (k0_hay_mix per 16 bytes, keyed with the call's number and the
// context's secret, XORed) is the check the host wrote in front of the word; otherwise the workgroup reads the haystack
// behind an acquire fence, as a launched K0 reads it.
// WHEN to poll: a poll that leaves as soon as the result is out arrives at the host's memory just before the host has
// written the next call of a loop (the result takes as long to get there as the poll) -- measured: half of the polls
// that saw the word had caught the haystack half-written, and the word was seen on the second poll, 2.3 us after it was
// written.  The first poll after a result waits `delay` ticks: longer by K0_DELAY_UP after a call whose first poll was too
// early (the word came with the second one, or with bytes that failed the check), shorter by one tick after a call
// whose first poll was right -- it settles a little behind the host's usual answer.  (Four waves polling in their own
// rhythms, the other way to have a poll arrive close behind the word: built, measured, 8.2 -> 11.5 us per call -- the
// workgroup's barrier waits for every poll in flight, and the result's write-back queues behind them.)
constexpr uint32_t K0_DELAY_UP = 25, K0_DELAY_MAX = 500;
template <int MODE>
__global__ __launch_bounds__(1024) void k0_resident(const DevAutomaton *A, const uint64_t *mailbox, int key_mode, int overlapping,
                                                    acx_match_t *out, uint64_t *res, uint64_t *status, uint64_t epoch,
                                                    uint64_t seq, uint64_t idle_ticks, uint64_t life_ticks, uint64_t secret,
                                                    uint32_t delay) {
    __shared__ uint64_t s_cmd;
    __shared__ uint32_t s_npre;
    const uint8_t *hay = (const uint8_t *)mailbox + K0_MAILBOX_HAY;
    const uint32_t t = threadIdx.x;
    const uint64_t t_start = wall_clock64();
    uint64_t t_last = t_start;
    constexpr uint64_t LEAVE = ~0ull;
    // what the kernel did, left beside the epoch when it leaves (status[1 .. 4]: calls, calls whose bytes came with the
    // poll, ticks between a word seen and its result published, polls; [5] the delay it ended with -- the next launch's
    // first; ACX_RESIDENT_TRACE=1 prints them)
    uint64_t n_calls = 0, n_inline = 0, busy = 0, n_polls = 0;
    // lanes that poll: the word's and as many as the last call's haystack took, in whole 64-byte lines (the fewer lines a
    // poll reads, the less often it catches the host between two of them)
    uint32_t width = 64;
    for (bool first = true;; first = false) {
        uint4 pre = make_uint4(0, 0, 0, 0);
        uint64_t t_seen = 0;
        if (t < 64) {
            uint64_t w;
            uint32_t npre = 0;
            if (!first) while (wall_clock64() - t_last < delay) __builtin_amdgcn_s_sleep(1);
            for (uint32_t polls = 0;; polls++) {
                n_polls++;
                if (t < width) pre = load16_system((const uint8_t *)mailbox + 16 * t);
                w = uniform64(pre.x, pre.y);
                const uint64_t now = wall_clock64();
                // (polls: a second bound, should the clock not be what it is taken for)
                if ((w & K0_MAILBOX_QUIT) || now - t_start > life_ticks || now - t_last > idle_ticks || polls > (1u << 24)) {
                    w = LEAVE;
                    break;
                }
                if ((uint32_t)(w >> 32) != (uint32_t)(seq + 1)) continue;
                const uint32_t len = (uint32_t)(w & K0_MAILBOX_LEN_MASK);
                const uint32_t covered = len < K0_MAILBOX_INLINE ? (len + 15) & ~15u : K0_MAILBOX_INLINE;
                bool torn = false;
                if (covered <= 16 * (width - 1)) {
                    const uint64_t want = uniform64(pre.z, pre.w);
                    uint64_t h = t >= 1 && 16 * (t - 1) < covered
                                     ? k0_hay_mix(((uint64_t)pre.y << 32) | pre.x, ((uint64_t)pre.w << 32) | pre.z, t - 1, (seq + 1) ^ secret)
                                     : 0;
                    for (int o = 32; o > 0; o >>= 1) h ^= __shfl_xor(h, o);
                    if (h == want) npre = covered; else torn = true;
                }
                width = min(64u, (1 + covered / 16 + 3) & ~3u);
                if (!first) {
                    if (polls == 1 || (polls == 0 && torn)) delay = min(delay + K0_DELAY_UP, K0_DELAY_MAX);
                    else if (polls == 0 && delay) delay--;
                }
                t_seen = now;
                break;
            }
            if (t == 0) { s_cmd = w; s_npre = npre; }
        }
        __syncthreads();
        const uint64_t w = s_cmd;
        if (w == LEAVE) {
            if (t == 0) {
                status[1] = n_calls; status[2] = n_inline; status[3] = busy; status[4] = n_polls;
                status[5] = delay;
                __hip_atomic_store(status, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        const uint32_t len = (uint32_t)(w & K0_MAILBOX_LEN_MASK), npre = s_npre;
        // (bytes to read behind the word: system scope, every wave -- nothing of the last haystack in its caches)
        if (npre < len) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        seq++;
        k0_resident_call<MODE>(A, hay, len, key_mode, overlapping, (w & K0_MAILBOX_CP) ? 1 : 0, out, res, seq, first, pre, npre);
        // the result line leaves the caches NOW: a launched K0's stores do at the end of the kernel, and this kernel has no
        // end (measured, the first version: every call took the idle limit -- the line arrived when the kernel left)
        if (t < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        t_last = wall_clock64();
        n_calls++;
        n_inline += npre ? 1 : 0;
        busy += t_last - t_seen;
    }
}

int small_mode(const DevAutomaton &A, uint32_t len, bool direct_ok) {
    // (ACX_K0_NO_LDS_TABLE: measurements)
    static const bool no_lt = std::getenv("ACX_K0_NO_LDS_TABLE") != nullptr;
    static const bool no_dc = std::getenv("ACX_K0_NO_DIRECT") != nullptr;
    const bool lt = A.table && ((uint64_t)A.n_states << A.stride2) <= K0_LT_ENTRIES && A.n_states <= K0_LT_IDS &&
                    A.n_patterns <= K0_LT_IDS && A.max_len < 256 && !no_lt;
    const bool dc = A.n_patterns <= K0_DC_PATTERNS && A.min_len >= 1 && A.max_len <= 16 && A.pat_blob && A.pat_off &&
                    (uint64_t)len * A.n_patterns <= K0_DC_WORK && !no_dc && direct_ok;
    // the prefilter (MODE 3): beyond SMALL_MAX_LEN the only way; below it for automata whose tables do not fit the LDS
    // from 1 KiB on (shorter: the walk's handful of gathers is as good).  Measured again in round 6 with the resident kernel
    // (ACX_K0_PF_MIN=0: from the first byte): a haystack WITH a match gains -- 64 bytes with one match 8.9 -> 7.4 us, the walk
    // that finds it is a chain of a dozen dependent gathers --, one without loses: the reference's long dataset (1 haystack
    // in 90 holds a match) 6.1-6.3 -> 6.6-6.9 us per call.  The reference's own loop decides: 1 KiB stays.
    static const uint32_t pf_min = std::getenv("ACX_K0_PF_MIN") ? (uint32_t)std::atoi(std::getenv("ACX_K0_PF_MIN")) : 1024u;
    const bool pf = small_prefilter_ok(A) && (len > SMALL_MAX_LEN || (!dc && !lt && len > pf_min));
    if (len > SMALL_MAX_LEN && !pf) return -1;
    return pf ? 3 : dc ? 2 : lt ? 1 : 0;
}

hipError_t launch_small(const DevAutomaton &A, const uint8_t *hay, uint32_t len, int key_mode, bool overlapping,
                        bool codepoints, acx_match_t *out, uint64_t *res, uint64_t seq, hipStream_t st, bool direct_ok) {
    const int mode = small_mode(A, len, direct_ok);
    if (mode < 0) return hipErrorInvalidValue;
#define ACX_K0(M)                                                                                                      \
    hipLaunchKernelGGL(k0_small<M>, dim3(1), dim3(1024), 0, st, A, hay, len, key_mode, overlapping ? 1 : 0,            \
                       codepoints ? 1 : 0, out, res, seq)
    if (mode == 3) ACX_K0(3); else if (mode == 2) ACX_K0(2); else if (mode == 1) ACX_K0(1); else ACX_K0(0);
#undef ACX_K0
    return hipGetLastError();
}

hipError_t launch_resident(const DevAutomaton *A, int mode, const uint64_t *mailbox, int key_mode, bool overlapping,
                           acx_match_t *out, uint64_t *res, uint64_t *status, uint64_t epoch, uint64_t seq,
                           uint64_t idle_ticks, uint64_t life_ticks, uint64_t secret, uint32_t delay, hipStream_t st) {
#define ACX_K0R(M)                                                                                                     \
    hipLaunchKernelGGL(k0_resident<M>, dim3(1), dim3(1024), 0, st, A, mailbox, key_mode, overlapping ? 1 : 0, out, res, \
                       status, epoch, seq, idle_ticks, life_ticks, secret, delay)
    if (mode == 3) ACX_K0R(3); else if (mode == 2) ACX_K0R(2); else if (mode == 1) ACX_K0R(1); else if (mode == 0) ACX_K0R(0);
    else return hipErrorInvalidValue;
#undef ACX_K0R
    return hipGetLastError();
}

bool small_prefilter_ok(const DevAutomaton &A) {
    static const bool off = std::getenv("ACX_K0_NO_PREFILTER") != nullptr; // (measurements)
    return !off && A.filter_q >= 3 && A.short_min_len == 0 && A.filterA && A.ptab && A.pinfo && A.max_len < (1u << 15);
}

// ---------------------------------------------------------------------------
// K3: UTF-8 code-point indexes
// ---------------------------------------------------------------------------
// one wave per 1 KiB block: cnt[blk] = its lead bytes, sub[blk * 64 + q] = those of its q-th 16 bytes
__global__ __launch_bounds__(256) void k_count_leads(const uint8_t *__restrict__ hay,
                                                     uint64_t len, uint64_t *cnt, uint8_t *sub,
                                                     uint64_t nblocks) {
    uint64_t blk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t lane = threadIdx.x & 63;
    if (blk > nblocks) return;
    if (blk == nblocks) { if (lane == 0) cnt[blk] = 0; return; }
    uint64_t base = blk * 1024 + lane * 16;
    uint32_t c = 0;
    if (base + 16 <= len && (((uintptr_t)(hay + base)) & 15) == 0) {
        uint4 v = *(const uint4 *)(hay + base);
        c = lead_bytes_in_word(v.x) + lead_bytes_in_word(v.y) + lead_bytes_in_word(v.z) +
            lead_bytes_in_word(v.w);
    } else {
        for (uint32_t k = 0; k < 16; k++)
            if (base + k < len && (hay[base + k] & 0xC0) != 0x80) c++;
    }
    sub[blk * 64 + lane] = (uint8_t)c; // one count per 16 bytes
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane == 0) cnt[blk] = c;
}

hipError_t count_lead_bytes(const uint8_t *d_hay, uint64_t len, uint64_t *cnt, uint8_t *sub, hipStream_t st) {
    uint64_t nblocks = (len + 1023) / 1024;
    uint64_t waves = nblocks + 1;
    hipLaunchKernelGGL(k_count_leads, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, d_hay,
                       len, cnt, sub, nblocks);
    return hipGetLastError();
}

// cnt[blk] = the sum of the block's 16 sub counts (blk < nblocks), cnt[nblocks] = 0: the block
// totals when K1b has already counted the 64-byte stretches (no second pass over the haystack)
__global__ void k_block_totals(const uint8_t *__restrict__ sub, uint64_t *cnt, uint64_t nblocks) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk > nblocks) return;
    uint32_t acc = 0;
    if (blk < nblocks) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = *(const uint4 *)(sub + blk * 64 + 16 * k);
            acc = __builtin_amdgcn_sad_u8(v.x, 0u, acc); acc = __builtin_amdgcn_sad_u8(v.y, 0u, acc);
            acc = __builtin_amdgcn_sad_u8(v.z, 0u, acc); acc = __builtin_amdgcn_sad_u8(v.w, 0u, acc);
        }
    }
    cnt[blk] = acc;
}

hipError_t block_totals(const uint8_t *sub, uint64_t *cnt, uint64_t nblocks, hipStream_t st) {
    hipLaunchKernelGGL(k_block_totals, dim3((uint32_t)((nblocks + 256) / 256)), dim3(256), 0, st, sub, cnt, nblocks);
    return hipGetLastError();
}

hipError_t prefix_sum_u64(void *temp, size_t temp_bytes, const uint64_t *in, uint64_t *out,
                          uint64_t n, hipStream_t st) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, (uint64_t)0, (size_t)n,
                                   rocprim::plus<uint64_t>(), st);
}

// The code-point prefix of the 1 KiB blocks in TWO launches of this file's own kernels (round 6; until then k_block_totals +
// the library's device-wide scan: three launches, 25 + 49 + 5 us beside k_tile_main on cfg5's 1 GiB): pre[blk] = lead bytes
// in front of block blk, for blk = 0 .. nblocks (nblocks + 1 entries; cnt[nblocks] = 0).
//   k_block_partials  a workgroup per BP_BLOCKS blocks: cnt[blk] = the sum of the block's 64 sub counts (sub != null:
//                     K1b has counted the 16-byte stretches; else cnt is there already: k_count_leads), partial[wg] = the
//                     workgroup's sum
//   k_block_prefix    the same geometry: a workgroup sums the partials in front of it (at most BP_MAX_WGS of them: a
//                     haystack of up to 4 GiB; beyond, block_prefix() takes the library's scan), scans its own counts
// (workgroups of 256 threads, four blocks per thread: beside k_tile_main's 128-thread groups a workgroup of 1 024 threads
// waits for sixteen free wave slots on ONE compute unit -- the first version took 96 us for cfg5's 64 MiB of counts)
constexpr uint32_t BP_BLOCKS = 1024, BP_THREADS = 256, BP_MAX_WGS = 4096;
__global__ __launch_bounds__(BP_THREADS) void k_block_partials(const uint8_t *__restrict__ sub, uint64_t *cnt, uint64_t *partial,
                                                               uint64_t nblocks) {
    __shared__ uint32_t red[BP_THREADS / 64];
    uint32_t sum = 0; // (a block holds at most 1 024 lead bytes: a workgroup's sum fits 32 bits)
#pragma unroll
    for (uint32_t k = 0; k < BP_BLOCKS / BP_THREADS; k++) {
        const uint64_t blk = (uint64_t)blockIdx.x * BP_BLOCKS + k * BP_THREADS + threadIdx.x;
        uint32_t acc = 0;
        if (blk < nblocks) {
            if (sub) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 v = *(const uint4 *)(sub + blk * 64 + 16 * q);
                    acc = __builtin_amdgcn_sad_u8(v.x, 0u, acc); acc = __builtin_amdgcn_sad_u8(v.y, 0u, acc);
                    acc = __builtin_amdgcn_sad_u8(v.z, 0u, acc); acc = __builtin_amdgcn_sad_u8(v.w, 0u, acc);
                }
            } else {
                acc = (uint32_t)cnt[blk];
            }
        }
        if (blk <= nblocks && (sub || blk == nblocks)) cnt[blk] = acc;
        sum += acc;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tot = 0;
        for (uint32_t w = 0; w < BP_THREADS / 64; w++) tot += red[w];
        partial[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(BP_THREADS) void k_block_prefix(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ partial,
                                                             uint64_t *pre, uint64_t n) {
    using scan_t = rocprim::block_scan<uint64_t, BP_THREADS>;
    __shared__ typename scan_t::storage_type scan_tmp;
    __shared__ uint64_t red[BP_THREADS / 64];
    __shared__ uint64_t s_base;
    uint64_t mine = 0;
    for (uint32_t w = threadIdx.x; w < blockIdx.x; w += BP_THREADS) mine += partial[w];
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t b = 0;
        for (uint32_t w = 0; w < BP_THREADS / 64; w++) b += red[w];
        s_base = b;
    }
    __syncthreads();
    // four consecutive blocks per thread
    constexpr uint32_t PER = BP_BLOCKS / BP_THREADS;
    const uint64_t b0 = (uint64_t)blockIdx.x * BP_BLOCKS + (uint64_t)threadIdx.x * PER;
    uint64_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) { v[k] = b0 + k < n ? cnt[b0 + k] : 0; sum += v[k]; }
    uint64_t excl = 0;
    scan_t().exclusive_scan(sum, excl, (uint64_t)0, scan_tmp);
    uint64_t run = s_base + excl;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) { if (b0 + k < n) pre[b0 + k] = run; run += v[k]; }
}

hipError_t block_prefix(const uint8_t *sub, uint64_t *cnt, uint64_t *pre, uint64_t nblocks, void *temp, size_t temp_bytes,
                        hipStream_t st) {
    const uint64_t n = nblocks + 1, nwg = (n + BP_BLOCKS - 1) / BP_BLOCKS;
    static const bool lib = std::getenv("ACX_BLOCK_PREFIX_LIBRARY") != nullptr; // (measurements: the round-5 way)
    if (nwg > BP_MAX_WGS || temp_bytes < nwg * 8 || lib) {
        if (sub) {
            hipError_t e = block_totals(sub, cnt, nblocks, st);
            if (e != hipSuccess) return e;
        }
        return prefix_sum_u64(temp, temp_bytes, cnt, pre, n, st);
    }
    hipLaunchKernelGGL(k_block_partials, dim3((uint32_t)nwg), dim3(BP_THREADS), 0, st, sub, cnt, (uint64_t *)temp, nblocks);
    hipLaunchKernelGGL(k_block_prefix, dim3((uint32_t)nwg), dim3(BP_THREADS), 0, st, cnt, (const uint64_t *)temp, pre, n);
    return hipGetLastError();
}

// one thread per match: the start from its 1 KiB block's prefix, the end from the start
__global__ void k_to_code_points(const uint8_t *__restrict__ hay, const uint64_t *blockpre,
                                 const uint8_t *__restrict__ sub, acx_match_t *m, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = m[i].start, e = m[i].end;
    const uint64_t cs = code_point_of(hay, blockpre, sub, s, CP_UNKNOWN);
    m[i].start = cs;
    m[i].end = cs + lead_bytes_between(hay + s, hay + e);
}

hipError_t to_code_points(const uint8_t *d_hay, uint64_t len, const uint64_t *blockpre, const uint8_t *sub,
                          acx_match_t *m, uint64_t n, hipStream_t st) {
    (void)len;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_to_code_points, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                       d_hay, blockpre, sub, m, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// a call cut into byte ranges (acx_api.cpp, run_chunked): where a piece's matches are cut, and the pieces' matches
// copied into the call's result with their offsets made global
// ---------------------------------------------------------------------------
// m[0 .. n) is ordered so that {m[i].field + shift < limit} holds for a PREFIX of it (non-overlapping matches by their
// start, overlapping occurrences by their end).  out[0] = the length of that prefix, out[1] = the end (+ shift) of its last
// element (0 when it is empty).  One thread per element: the one that sees the border writes.
__global__ void k_cut_point(const acx_match_t *__restrict__ m, uint64_t n, int by_end, uint64_t shift, uint64_t limit, uint64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = (by_end ? m[i].end : m[i].start) + shift;
    if (v >= limit) return;
    if (i + 1 < n) {
        const uint64_t nx = (by_end ? m[i + 1].end : m[i + 1].start) + shift;
        if (nx < limit) return;
    }
    out[0] = i + 1;
    out[1] = m[i].end + shift;
}
hipError_t cut_point(const acx_match_t *m, uint64_t n, bool by_end, uint64_t shift, uint64_t limit, uint64_t *out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, 16, st);
    if (e != hipSuccess || !n) return e;
    hipLaunchKernelGGL(k_cut_point, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, m, n, by_end ? 1 : 0, shift, limit, out);
    return hipGetLastError();
}
// dst[0 .. n) = src[0 .. n) with start and end moved by `shift`; a thread per 64-bit word (three per match: coalesced)
__global__ void k_copy_shifted(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint64_t words, uint64_t shift) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    dst[w] = src[w] + (w % 3 ? shift : 0);
}
// dst[i] = src[i] - base: the offsets of the haystacks of a batch's second part, from that part's first byte (acx_api.cpp: a
// batch beyond 2^32 occurrences in one pass is cut at a haystack boundary)
__global__ void k_rebase_offsets(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] - base;
}
hipError_t rebase_offsets(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_rebase_offsets, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, dst, src, n, base);
    return hipGetLastError();
}
hipError_t copy_shifted(acx_match_t *dst, const acx_match_t *src, uint64_t n, uint64_t shift, hipStream_t st) {
    static_assert(sizeof(acx_match_t) == 24, "three words per match");
    for (uint64_t at = 0; at < n;) { // (launches of at most 2^30 matches: the grid's 32 bits)
        const uint64_t k = std::min<uint64_t>(n - at, 1ull << 30);
        hipLaunchKernelGGL(k_copy_shifted, dim3((uint32_t)((3 * k + 255) / 256)), dim3(256), 0, st, (uint64_t *)(dst + at),
                           (const uint64_t *)(src + at), 3 * k, shift);
        at += k;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// copies of a pattern, overlapping searches (acx_api.cpp, expand_copies): the search ran on the view without the later
// copies -- one occurrence per STRING, under the lowest id --; here every occurrence becomes the run of its string's copies
// (the reference reports them together, ids ascending: they sit in ONE state's match list in the order they were added,
// aho-corasick 1.1.4 nfa/noncontiguous.rs add_match; pinned by tests/test_gpu_round4.py and the oracle)
// ---------------------------------------------------------------------------
__global__ void k_copy_counts(const acx_match_t *__restrict__ m, uint64_t n, const uint32_t *__restrict__ xcnt, uint64_t *k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    k[i] = i < n ? 1 + (uint64_t)xcnt[m[i].pattern] : 0; // (one more element: the exclusive prefix's last one is the total)
}
hipError_t copy_runs(const acx_match_t *m, uint64_t n, const uint32_t *xcnt, void *temp, size_t temp_bytes, uint64_t *k,
                     uint64_t *offs, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_counts, dim3((uint32_t)((n + 256) / 256)), dim3(256), 0, st, m, n, xcnt, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::exclusive_scan(temp, temp_bytes, k, offs, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st);
}
// one thread per record of the output: its occurrence by binary search in the runs' offsets (strictly increasing), coalesced writes
__global__ void k_expand_copies(const acx_match_t *__restrict__ m, uint64_t n, const uint64_t *__restrict__ offs,
                                const uint32_t *__restrict__ xoff, const uint32_t *__restrict__ xids, acx_match_t *out, uint64_t total) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    uint64_t lo = 0, hi = n; // offs[lo] <= j < offs[hi]
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (offs[mid] <= j) lo = mid; else hi = mid;
    }
    const acx_match_t v = m[lo];
    const uint64_t r = j - offs[lo];
    acx_match_t o = v;
    if (r) o.pattern = xids[xoff[v.pattern] + (r - 1)];
    out[j] = o;
}
hipError_t expand_copies_write(const acx_match_t *m, uint64_t n, const uint64_t *offs, const uint32_t *xoff, const uint32_t *xids,
                               acx_match_t *out, uint64_t total, hipStream_t st) {
    if (total >= (1ull << 39)) return hipErrorInvalidValue; // (the grid's 32 bits x 256; 2^39 records = 13 TB)
    hipLaunchKernelGGL(k_expand_copies, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, m, n, offs, xoff, xids, out, total);
    return hipGetLastError();
}
// batch: the per-haystack counts of the expanded result -- the occurrences of haystack h are [C[h-1], C[h]) of the
// unexpanded one (C: inclusive prefix of its counts), their records [offs[C[h-1]], offs[C[h]])
__global__ void k_expand_counts(const uint64_t *__restrict__ incl, const uint64_t *__restrict__ offs, uint64_t *counts, uint64_t n_hay) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hay) return;
    counts[h] = offs[incl[h]] - offs[h ? incl[h - 1] : 0];
}
hipError_t expand_copies_counts(void *temp, size_t temp_bytes, uint64_t *counts, uint64_t n_hay, uint64_t *incl, const uint64_t *offs,
                                hipStream_t st) {
    if (!n_hay) return hipSuccess;
    hipError_t e = rocprim::inclusive_scan(temp, temp_bytes, counts, incl, (size_t)n_hay, rocprim::plus<uint64_t>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_expand_counts, dim3((uint32_t)((n_hay + 255) / 256)), dim3(256), 0, st, incl, offs, counts, n_hay);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// batch: local offsets + per-haystack counts
// ---------------------------------------------------------------------------
__global__ void k_localize(Segments G, const uint8_t *__restrict__ hay, uint64_t len,
                           const uint64_t *blockpre, const uint8_t *__restrict__ sub, int codepoints,
                           acx_match_t *m, uint64_t n, uint64_t *counts) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = m[i].start, e = m[i].end;
    uint64_t h, base;
    if (G.uniform_len) { h = s / G.uniform_len; base = h * G.uniform_len; }
    else { h = upper_bound_u64(G.offsets, G.n_hay + 1, s) - 1; base = G.offsets[h]; }
    (void)len;
    if (codepoints) {
        const uint64_t cs = code_point_of(hay, blockpre, sub, s, CP_UNKNOWN) - code_point_of(hay, blockpre, sub, base, CP_UNKNOWN);
        m[i].start = cs;
        m[i].end = cs + lead_bytes_between(hay + s, hay + e);
    } else {
        m[i].start = s - base;
        m[i].end = e - base;
    }
    atomicAdd((unsigned long long *)&counts[h], 1ull);
}

hipError_t localize(const Segments &G, const uint8_t *d_hay, uint64_t len,
                    const uint64_t *blockpre, const uint8_t *sub, int codepoints, acx_match_t *m, uint64_t n,
                    uint64_t *counts, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_localize, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, G, d_hay,
                       len, blockpre, sub, codepoints, m, n, counts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// synthetic haystacks (bit-exact twins of tests/gen.py)
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

__global__ void k_generate(uint8_t *dst, uint64_t len, int kind, uint64_t seed,
                           uint64_t stream_offset) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= len) return;
    uint64_t out = 0;
    uint32_t nb = len - i < 8 ? (uint32_t)(len - i) : 8;
    for (uint32_t k = 0; k < nb; k++) {
        uint64_t z = mix64(seed + (stream_offset + i + k + 1) * GOLDEN);
        uint8_t b;
        if (kind == 0) b = 97 + (uint8_t)(z % 26);
        else b = ((z & 0xFF) < 43) ? 32 : 97 + (uint8_t)((z >> 8) % 26);
        out |= (uint64_t)b << (8 * k);
    }
    if (nb == 8 && (((uintptr_t)(dst + i)) & 7) == 0) *(uint64_t *)(dst + i) = out;
    else for (uint32_t k = 0; k < nb; k++) dst[i + k] = (uint8_t)(out >> (8 * k));
}

__global__ void k_plant(DevAutomaton A, uint8_t *dst, uint64_t len, uint64_t seed,
                        uint64_t stream_offset) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // local 1 KiB block
    if ((b + 1) * 1024 > len || A.n_patterns == 0) return;
    uint64_t gb = stream_offset / 1024 + b;
    uint64_t s2 = seed ^ 0x5EEDull;
    uint64_t z0 = mix64(s2 + (2 * gb + 1) * GOLDEN), z1 = mix64(s2 + (2 * gb + 2) * GOLDEN);
    uint64_t p = z0 % A.n_patterns;
    uint64_t pl = A.pat_off[p + 1] - A.pat_off[p];
    if (pl >= 1024) return;
    uint64_t o = b * 1024 + z1 % (1024 - pl);
    for (uint64_t k = 0; k < pl; k++) dst[o + k] = A.pat_blob[A.pat_off[p] + k];
}

hipError_t generate(const DevAutomaton &A, uint8_t *dst, uint64_t len, int kind, uint64_t seed,
                    uint64_t stream_offset, hipStream_t st) {
    if (!len) return hipSuccess;
    uint64_t threads = (len + 7) / 8;
    hipLaunchKernelGGL(k_generate, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, dst,
                       len, kind, seed, stream_offset);
    if (kind == 1 && A.n_patterns) {
        uint64_t nb = len / 1024;
        if (nb)
            hipLaunchKernelGGL(k_plant, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, st, A, dst,
                               len, seed, stream_offset);
    }
    return hipGetLastError();
}

} // namespace acx
