// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the MI355X-native
// Aho-Corasick matcher.  This file replaces the reference's match loop: the
// crate iterators drained at /root/reference/src/lib.rs:229-249 and 422-434
// (`try_find_iter` / `try_find_overlapping_iter`, called at src/lib.rs:59, 53).
//
// Pipeline (all on the device, one stream):
//   K1   scan        every occurrence (pattern, end) of every pattern -> sink
//        K1a dfa_walk    one lane walks one chunk of the stream through the
//                        dense DFA; class map + hot (shallow, BFS-first) rows
//                        live in LDS, cold rows come from the HBM table
//        K1b prefilter   position-parallel: coalesced 16-B loads, a q-gram
//                        bitmap in LDS says whether a pattern can start here;
//                        survivors are ballot-compacted into a per-wave LDS
//                        queue and verified 64 at a time by an anchored walk
//                        of the HBM DFA table
//   K2   sort (rocPRIM radix sort on the 64-bit key) + resolve: applies the
//        match kind (Standard / LeftmostFirst / LeftmostLongest, overlapping
//        or not) exactly as the reference iterators would
//   K3   UTF-8 byte offset -> code-point index (get_byte_to_code_point,
//        src/lib.rs:73-88) by per-KiB lead-byte counts + prefix sum
//
// This is integer, HBM-bound work: no MFMA anywhere.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

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

// per-workgroup view of the sink: region base + LDS slot counter
struct BlockSink {
    uint4 *recs;
    uint32_t *bucket_cnt;
    uint32_t *lcount; // LDS
    uint64_t region_cap;
    uint32_t bucket_shift;
    int key_mode;
    uint4 *slots;
    uint32_t *abort_flag;
};

__device__ __forceinline__ BlockSink block_sink(const Sink &K, uint32_t *lcount, uint32_t quads = 1) {
    return BlockSink{K.recs + (uint64_t)blockIdx.x * K.region_cap * quads, K.bucket_cnt, lcount,
                     K.region_cap, K.bucket_shift, K.key_mode, K.slots, K.abort_flag};
}

// slot mode: the occurrence with arrival rank r of its bucket
__device__ __forceinline__ void store_slot(const BlockSink &K, uint32_t bucket, uint32_t r, uint64_t key,
                                           uint32_t pid, uint32_t plen) {
    if (r < BUCKET_SLOTS)
        K.slots[(uint64_t)bucket * BUCKET_SLOTS + r] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, plen);
    else
        *K.abort_flag = 1;
}

// store one occurrence (ONE 16-byte store).  Slot mode: take the occurrence's arrival
// rank inside its bucket (one global atomic, the address is shared by the ~4 occurrences
// of a 4 KiB stretch only) and store into that slot.  Region mode: next slot of the region.
// (records carry the pattern's length so that nothing downstream has to gather it again)
__device__ __forceinline__ void emit_key(const BlockSink &K, uint64_t key, uint32_t pid, uint32_t plen) {
    if (K.slots) {
        const uint32_t bucket = (uint32_t)(key >> K.bucket_shift);
        store_slot(K, bucket, atomicAdd(&K.bucket_cnt[bucket], 1u), key, pid, plen);
        return;
    }
    uint32_t slot = atomicAdd(K.lcount, 1u); // LDS atomic
    if (slot < K.region_cap) K.recs[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, plen);
}

// Wave-aggregated variant for code where many lanes emit together (walk kernel).
// Slot mode: ONE global atomic per run of adjacent emitting lanes that fall into the
// same bucket takes the bucket ranks (hits arrive grouped by 4 KiB tile, so a run is
// typically a whole tile: device-scope atomics with return are the expensive part of
// emission).  Region mode: one LDS atomic per wave reserves the slots.
__device__ __forceinline__ void emit_key_agg(const BlockSink &K, bool ok, uint64_t key, uint32_t pid,
                                             uint32_t plen) {
    const unsigned long long fm = __ballot(ok);
    if (!fm) return;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below_me = (1ull << lane) - 1;
    if (K.slots) {
        const uint32_t bucket = (uint32_t)(key >> K.bucket_shift);
        // previous emitting lane and its bucket
        const unsigned long long below = fm & below_me;
        const uint32_t prev = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
        const uint32_t pb = __shfl(bucket, prev);
        const bool head = ok && (!below || pb != bucket);
        const unsigned long long hm = __ballot(head);
        // my run = emitting lanes from my head lane up to (not including) the next head
        const unsigned long long upto_me = below_me | (1ull << lane);
        const uint32_t hl = 63u - (uint32_t)__builtin_clzll((hm & upto_me) | 1ull);
        const unsigned long long above = hl >= 63 ? 0ull : (hm & ~((2ull << hl) - 1));
        const unsigned long long run_end = above ? ((1ull << __builtin_ctzll(above)) - 1) : ~0ull;
        const unsigned long long run = fm & run_end & ~((1ull << hl) - 1);
        uint32_t base = 0;
        if (head) base = atomicAdd(&K.bucket_cnt[bucket], (uint32_t)__popcll(run));
        base = __shfl(base, hl);
        if (ok) store_slot(K, bucket, base + (uint32_t)__popcll(run & below_me), key, pid, plen);
        return;
    }
    const uint32_t leader = (uint32_t)__builtin_ctzll(fm);
    uint32_t sbase = 0;
    if (lane == leader) sbase = atomicAdd(K.lcount, (uint32_t)__popcll(fm));
    sbase = __shfl(sbase, leader);
    if (ok) {
        uint32_t slot = sbase + (uint32_t)__popcll(fm & below_me);
        if (slot < K.region_cap) K.recs[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, plen);
    }
}

// The emit paths are cold and out of line; they read the automaton through a
// pointer to its device-resident copy so that the kernels never have to spill
// their by-value kernel arguments to scratch for them.
__device__ __forceinline__ void emit_one(const DevAutomaton *A, const BlockSink K, uint32_t pid,
                                         uint64_t end) {
    uint64_t key;
    const uint32_t plen = A->plen[pid];
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
__device__ __noinline__ void emit_state(const DevAutomaton *A, const BlockSink K, uint32_t s,
                                        uint64_t end) {
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

__device__ __forceinline__ uint32_t dfa_step(const DevAutomaton &A, const WalkCtx &W,
                                             uint32_t s, uint32_t byte) {
    uint32_t c = W.lcls[byte];
    if (s < W.hot_rows) {
        uint32_t v = W.lrows[(s << A.stride2) + c];
        if (v != 0xFFFFu) return (v & 0x3FFFu) | ((v & 0xC000u) << 16);
    }
    return A.table[((size_t)s << A.stride2) + c];
}

template <bool EMIT>
__device__ __forceinline__ uint32_t walk_span(const DevAutomaton &A, const DevAutomaton *Ad,
                                              const WalkCtx &W, const BlockSink &K,
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
    uint32_t *lcount = (uint32_t *)(smem + 256);
    uint16_t *lrows = (uint16_t *)(smem + K1A_LDS_HEADER);
    if (threadIdx.x == 0) *lcount = 0;
    const BlockSink K = block_sink(GK, lcount);
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
    if (threadIdx.x == 0) GK.block_counts[blockIdx.x] = *lcount;
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
    uint32_t chunk = pick_chunk(A.max_len, len);
    uint32_t rows = A.hot_rows;
    uint32_t cap_rows = dfa_walk_hot_rows(A.n_states, A.stride2, max_lds);
    if (rows > cap_rows) rows = cap_rows;
    size_t lds = K1A_LDS_HEADER + (((size_t)rows << A.stride2) * 2 + 15) / 16 * 16;
    uint64_t blocks = grid;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k1a_dfa_walk,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)max_lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k1a_dfa_walk, dim3((uint32_t)blocks), dim3(1024), lds, st, A, Ad, G, K,
                       d_hay, len, chunk, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1b: LDS prefix prefilter -> exact prefix table -> anchored DFA walk
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
//       (true Q-byte prefix hits + ~0.1 % collisions) are ballot-compacted into
//       the wave's queue Q1 (tile-relative u16 offsets).
//   L2  exact probe of the depth-Q2 prefix table (HBM, L2-resident), software-
//       pipelined over tiles so that no wave waits on it: tile t's survivors
//       re-read their 8-byte window (phase A, tile t+1), fetch their home slot
//       (phase B, t+2), compare (phase C, t+3).  Hits (position, depth-Q2 state)
//       leave through the sink: the scan kernel never walks the DFA itself, so
//       its waves never sit in the long dependent-load chains of a walk.
//   L3  (separate kernel k_walk_hits, one thread per prefix hit): the few
//       patterns that own that prefix are settled by comparing their remaining
//       bytes with the haystack; matches go to the occurrence sink.
// All LDS is ONE static object with the L1 table at offset 0.
constexpr int K1B_ROWS = 4;
constexpr uint32_t K1B_Q1CAP = 64;  // level-1 survivors of one tile (1 per lane)
constexpr uint32_t K1B_HB = 40;     // prefix hits a wave collects in LDS before one burst store
struct K1bLds {
    uint32_t xy[FILTER_WORDS];
    uint16_t q1[16][K1B_Q1CAP];
    uint4 hb[16][K1B_HB][2];
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

// A prefix hit handed to the walk kernel is (position, code), code = word 3 of
// the prefix-table entry: the id of the only pattern with that prefix, or
// HIT_LIST | index into blist ({count, pid, ...}); HIT_RETRY when the home
// slot held another gram (the walk kernel probes the table again first).
constexpr uint32_t HIT_LIST = 0x80000000u;
constexpr uint32_t HIT_RETRY = 0xFFFFFFFFu;

// L3: verify a prefix hit at stream position p: the first Q2 bytes are known to
// match, so every candidate pattern is settled by comparing its remaining bytes
// with the haystack -- independent loads, no dependent DFA walk.  Emits every
// pattern that starts at p.
__device__ __forceinline__ void verify_hit(const DevAutomaton &A, const Segments &G,
                                           const BlockSink &K, const uint8_t *__restrict__ stream,
                                           uint64_t len, uint64_t p, uint32_t code, uint64_t w0,
                                           uint64_t w1, uint32_t ablate) {
    // (w0, w1) = the 16 haystack bytes at p, carried with the hit by K1b so that short
    // patterns are verified without touching the (by now cold) haystack again
    const uint32_t q = A.filter_q2;
    const uint64_t room = segment_end(G, len, p) - p;
    if (room < q) return; // the prefix would straddle the end of its haystack
    if (code == HIT_RETRY) {
        if (ablate & 128) return; // profiling only: drop the hits that need a second probe
        const uint64_t gram = q >= 8 ? w0 : (w0 & ((1ull << (8 * q)) - 1));
        uint32_t idx = prefix_slot(gram_hash2(gram), A.ptab_log2);
        const uint32_t mask = (1u << A.ptab_log2) - 1;
        for (;;) {
            const uint4 e = *(const uint4 *)(A.ptab + (size_t)idx * 4);
            if (e.z == PREFIX_EMPTY) return;
            if ((((uint64_t)e.y << 32) | e.x) == gram) { code = e.w; break; }
            idx = (idx + 1) & mask;
        }
    }
    const bool list = (code & HIT_LIST) != 0;
    const uint32_t li = code & ~HIT_LIST;
    const uint32_t n = list ? A.blist[li] : 1;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t pid = list ? A.blist[li + 1 + k] : code;
        // ONE 16-byte load: rank, length (if < 255) and the 12 bytes after the first q
        const uint4 pi = A.pinfo[pid];
        const uint32_t rk = pi.x & 0xFFFFFFu;
        uint32_t L = pi.x >> 24;
        if (L == 255) L = A.plen[pid];
        bool ok = L <= room;
        if (ok && L > q) {
            // haystack bytes q.. from the carried window (16 - q of them), pattern bytes from pinfo
            const uint32_t have = 16 - q;                 // carried bytes beyond the prefix
            const uint32_t need = L - q < 12 ? L - q : 12; // bytes checkable against pinfo
            uint64_t h0 = q < 8 ? (q ? (w0 >> (8 * q)) | (w1 << (64 - 8 * q)) : w0) : (w1 >> (8 * (q - 8)));
            uint64_t h1 = q < 8 ? (q ? (w1 >> (8 * q)) : w1) : 0;
            uint64_t p0 = ((uint64_t)pi.z << 32) | pi.y, p1 = pi.w;
            uint32_t n0 = need < have ? need : have;       // bytes compared from the carried window
            uint64_t m0 = n0 >= 8 ? ~0ull : ((1ull << (8 * n0)) - 1);
            uint64_t m1 = n0 > 8 ? ((1ull << (8 * (n0 - 8))) - 1) : 0;
            ok = (((h0 ^ p0) & m0) | ((h1 ^ p1) & m1)) == 0;
            // whatever lies beyond the carried window / pinfo (long patterns): compare in place
            for (uint32_t d = q + n0; ok && d < L; d += 8) {
                uint64_t a = load_window(stream, len, p + d);
                uint64_t b;
                __builtin_memcpy(&b, A.pat_blob + A.pat_off[pid] + d, 8); // pat_blob is padded by 16 bytes
                uint32_t nbytes = L - d < 8 ? L - d : 8;
                uint64_t m = nbytes >= 8 ? ~0ull : ((1ull << (8 * nbytes)) - 1);
                ok = ((a ^ b) & m) == 0;
            }
        }
        uint64_t key = K.key_mode == 0   ? ((p + L) << A.rank_bits) | rk
                       : K.key_mode == 1 ? (p << A.rank_bits) | pid
                                         : (p << A.rank_bits) | rk;
        emit_key_agg(K, ok && !(ablate & 32), key, pid, L);
    }
}

constexpr uint32_t K_WALK_SPLIT = 1; // workgroups per hit region (a region = one K1b wave's hits)

// One thread per prefix hit of K1b.  Hits live in the per-workgroup regions of
// the scan's sink (H); occurrences go to the occurrence sink (GK).
__global__ __launch_bounds__(256) void k_walk_hits(DevAutomaton A, const DevAutomaton *Ad,
                                                   Segments G, Sink H, uint32_t h_grid, uint32_t split,
                                                   Sink GK, const uint8_t *__restrict__ stream,
                                                   uint64_t len, uint32_t ablate) {
    __shared__ uint32_t lcount;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    (void)Ad;
    const BlockSink K = block_sink(GK, &lcount);
    // `split` workgroups share one hit region
    for (uint32_t b = blockIdx.x / split; b < h_grid; b += gridDim.x / split) {
        uint64_t n = H.block_counts[b];
        if (n > H.region_cap) { // hits were dropped: the host grows the hit sink and redoes the call
            n = H.region_cap;
            if (GK.abort_flag && threadIdx.x == 0) *GK.abort_flag = 1;
        }
        const uint4 *rec = H.recs + (uint64_t)b * H.region_cap * 2;
        for (uint64_t i = (blockIdx.x % split) * 256 + threadIdx.x; i < n; i += split * 256) {
            const uint4 h = rec[2 * i], w = rec[2 * i + 1];
            if (ablate & 64) { if (h.x == 0x12345678u && w.y == 77) emit_key(K, 1, 1, 1); continue; }
            verify_hit(A, G, K, stream, len, ((uint64_t)h.y << 32) | h.x, h.z,
                       ((uint64_t)w.y << 32) | w.x, ((uint64_t)w.w << 32) | w.z, ablate);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) GK.block_counts[blockIdx.x] = lcount;
}

template <int Q>
__global__ __launch_bounds__(1024) void k1b_prefilter(DevAutomaton A, const DevAutomaton *Ad,
                                                      Segments G, Sink GK,
                                                      const uint8_t *__restrict__ hay,
                                                      uint64_t len, uint64_t lead,
                                                      uint64_t tile_begin, uint64_t tile_end,
                                                      uint32_t ablate) {
    // Scans the 4 KiB tiles [tile_begin, tile_end) of the stream (a chunk of a pipelined
    // call, or everything).  `hay` is 16-byte aligned; the first `lead` bytes (< 16) precede the real
    // stream and are never candidates.  Stream position = index - lead.
    __shared__ __attribute__((aligned(16))) K1bLds L;
    // the wave index is wave-uniform: say so, and the tile index, its byte offset, the
    // interior test and most of the prefetch address arithmetic move from VALU to SALU
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint16_t *q1 = L.q1[wave];
    // sink of prefix hits (two quads per record): every WAVE owns a region and keeps its cursor
    // in an SGPR -- no atomic, no cross-lane traffic on the push path
    const uint32_t region = blockIdx.x * 16 + wave;
    uint4 *const hrec = GK.recs + (uint64_t)region * GK.region_cap * 2;
    const uint32_t hcap = (uint32_t)(GK.region_cap < 0xFFFFFFFFull ? GK.region_cap : 0xFFFFFFFFull);
    uint32_t hcur = 0;
    {
        const uint4 *src = (const uint4 *)A.filterA;
        uint4 *dst = (uint4 *)L.xy;
        for (uint32_t i = threadIdx.x; i < sizeof(L.xy) / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    constexpr int GB = Q - 1; // bytes of the shared gram
    constexpr uint32_t GMASK = GB >= 4 ? 0xFFFFFFFFu : ((1u << (8 * GB)) - 1u);
    const uint8_t *stream = hay + lead;
    const uint64_t total = lead + len;          // bytes addressable from `hay`
    const uint64_t total16 = (total + 15) & ~15ull;
    const uint64_t last_start = total >= A.min_len ? total - A.min_len : 0; // last index a pattern can start at
    const bool any_start = total >= lead + A.min_len;
    const uint64_t tile_bytes = (uint64_t)K1B_ROWS * 1024;
    const uint64_t all_tiles = any_start ? (total + tile_bytes - 1) / tile_bytes : 0;
    const uint64_t ntiles = all_tiles < tile_end ? all_tiles : tile_end;
    const uint64_t gw = tile_begin + (uint64_t)blockIdx.x * 16 + wave;
    const uint64_t nw = (uint64_t)gridDim.x * 16;
    const uint32_t q2len = A.filter_q2;
    const uint64_t q2mask = q2len >= 8 ? ~0ull : ((1ull << (8 * q2len)) - 1);
    const uint32_t ptab_log2 = A.ptab_log2;
    uint32_t q1c = 0; // wave-uniform queue fill

    // ---- level-2 pipeline registers (tile-synchronous, one entry per lane)
    uint32_t nB = 0, nC = 0;            // wave-uniform counts
    uint64_t tbA = 0, tbB = 0, tbC = 0; // tile bases of the entries in Q1 / phase B / phase C
    uint64_t winB = 0, winB1 = 0, winC = 0, winC1 = 0;
    uint32_t offB = 0, offC = 0;
    uint4 entC = make_uint4(0, 0, PREFIX_EMPTY, 0);

    // Prefix hits (p, st) of the lanes with found == true go to the wave's region THROUGH an LDS
    // buffer that is stored in bursts of up to K1B_HB records: on this architecture stores count
    // in vmcnt like loads, so a store issued every iteration makes the `s_waitcnt vmcnt(0)` in
    // front of the next tile wait for HBM write latency every iteration; a burst every ~7
    // iterations does not, and its records are contiguous (measured: 1-2 % of the kernel).
    uint4 (*const hb)[2] = L.hb[wave];
    uint32_t hbn = 0; // wave-uniform fill of the buffer
#define K1B_HIT_FLUSH                                                                            \
    {                                                                                            \
        if (lane < hbn) {                                                                        \
            const uint32_t s_ = hcur + lane;                                                     \
            if (s_ < hcap) { hrec[2 * s_] = hb[lane][0]; hrec[2 * s_ + 1] = hb[lane][1]; }       \
        }                                                                                        \
        hcur += hbn; /* keeps counting past the capacity */                                      \
        hbn = 0;                                                                                 \
    }
#define K1B_HIT_PUSH(FOUND, P, ST, W0, W1)                                                       \
    {                                                                                            \
        unsigned long long fm_ = __ballot(FOUND);                                                \
        if (fm_) {                                                                               \
            const uint32_t np_ = (uint32_t)__popcll(fm_);                                        \
            if (hbn + np_ > K1B_HB) K1B_HIT_FLUSH                                                \
            const uint32_t rk_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm_ >> 32),                \
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)fm_, 0)); \
            const uint64_t p_ = (P), a_ = (W0), b_ = (W1);                                       \
            const uint4 r0_ = make_uint4((uint32_t)p_, (uint32_t)(p_ >> 32), (ST), 0);           \
            const uint4 r1_ = make_uint4((uint32_t)a_, (uint32_t)(a_ >> 32), (uint32_t)b_,       \
                                         (uint32_t)(b_ >> 32));                                  \
            if (np_ > K1B_HB) { /* more hits at once than the buffer holds: straight to HBM */   \
                if ((FOUND) && hcur + rk_ < hcap) { hrec[2 * (hcur + rk_)] = r0_; hrec[2 * (hcur + rk_) + 1] = r1_; } \
                hcur += np_;                                                                     \
            } else {                                                                             \
                if (FOUND) { hb[hbn + rk_][0] = r0_; hb[hbn + rk_][1] = r1_; }                   \
                hbn += np_;                                                                      \
            }                                                                                    \
        }                                                                                        \
    }

    // Tile loads are UNCONDITIONAL (addresses clamped to the last 16-byte block of
    // the stream) so that exactly five loads are in flight per prefetch: garbage
    // read for out-of-range blocks only ever feeds positions that are masked off.
    // nxtL = the 8 bytes that follow the tile (lane 63's look-ahead), read by all
    // lanes from one address.
    const uint64_t last_block = total16 - 16;
    u32x4 nxt0, nxt1, nxt2, nxt3;
    uint2 nxtL;
#define K1B_LOAD16(DST, PTR) DST = (ablate & 16) ? *(const u32x4 *)(PTR) : load16_stream(PTR); /* 16: plain loads */
#define K1B_ISSUE_ROW(DST, TILE, R)                                                              \
    {                                                                                            \
        uint64_t off_ = (TILE) * tile_bytes + (uint64_t)(R) * 1024 + lane * 16;                  \
        const uint8_t *ptr_ = hay + (off_ < last_block ? off_ : last_block);                     \
        K1B_LOAD16(DST, ptr_)                                                                    \
    }
    // The tile index is wave-uniform (SGPRs): a tile that lies wholly inside the stream -- all
    // but the last one -- is addressed as scalar base + lane * 16 + immediate row offset, no
    // VALU address arithmetic; both branches issue the same five loads.
#define K1B_ISSUE_TILE(TILE)                                                                     \
    {                                                                                            \
        const uint64_t tb_ = (TILE) * tile_bytes;                                                \
        if (tb_ + tile_bytes <= last_block) {                                                    \
            const uint8_t *tp_ = hay + tb_ + lane * 16;                                          \
            K1B_LOAD16(nxt0, tp_) K1B_LOAD16(nxt1, tp_ + 1024) K1B_LOAD16(nxt2, tp_ + 2048)      \
            K1B_LOAD16(nxt3, tp_ + 3072)                                                         \
            nxtL = *(const uint2 *)(hay + tb_ + tile_bytes);                                     \
        } else {                                                                                 \
            K1B_ISSUE_ROW(nxt0, TILE, 0) K1B_ISSUE_ROW(nxt1, TILE, 1) K1B_ISSUE_ROW(nxt2, TILE, 2) \
            K1B_ISSUE_ROW(nxt3, TILE, 3)                                                         \
            uint64_t off_ = tb_ + tile_bytes;                                                    \
            nxtL = *(const uint2 *)(hay + (off_ < last_block ? off_ : last_block));              \
        }                                                                                        \
    }
    K1B_ISSUE_TILE(gw)

    // three extra iterations drain the level-2 pipeline
    for (uint64_t tile = gw; tile < ntiles + 3 * nw; tile += nw) {
        // Everything loaded during the previous iteration (the tile prefetch and the
        // level-2 windows/slots) is consumed from here on.  Passing the tile through
        // an empty asm makes the compiler wait for those loads HERE, not with a vmcnt(0)
        // somewhere in the middle of level 1.
        u32x4 v0 = nxt0, v1 = nxt1, v2 = nxt2, v3 = nxt3;
        uint2 vL = nxtL;
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(vL.x), "+v"(vL.y));
        // ---- phase C: compare the slots fetched one tile ago with their grams
        if (nC) {
            bool act = lane < nC;
            bool same = (((uint64_t)entC.y << 32) | entC.x) == (winC & q2mask);
            // a home slot holding another gram proves absence unless PREFIX_MORE is set
            bool found = act && entC.z != PREFIX_EMPTY && (same || (entC.z & PREFIX_MORE));
            uint32_t st = same ? entC.w : HIT_RETRY;
            if (!(ablate & 2)) K1B_HIT_PUSH(found, tbC + offC - lead, st, winC, winC1)
        }
        // ---- phase B: hash the windows fetched one tile ago, fetch their home slots
        if (nB) {
            if (lane < nB) {
                uint32_t idx = prefix_slot(gram_hash2(winB & q2mask), ptab_log2);
                entC = *(const uint4 *)(A.ptab + (size_t)idx * 4);
            }
            offC = offB; winC = winB; winC1 = winB1;
        }
        nC = nB; tbC = tbB;
        // ---- phase A: fetch the 8-byte windows of the previous tile's survivors
        if (ablate & 1) q1c = 0;
        if (q1c) {
            if (lane < q1c) {
                offB = q1[lane];
                load_window16(stream, len, tbA + offB - lead, &winB, &winB1);
            }
        }
        nB = q1c; tbB = tbA; q1c = 0;
        __builtin_amdgcn_wave_barrier();
        if (tile >= ntiles) continue;

        // ---- level 1 on this tile
        const uint64_t tbase = tile * tile_bytes;
        tbA = tbase;
        uint32_t mrow0 = 0, mrow1 = 0, mrow2 = 0, mrow3 = 0;
        // a register whose LOW byte is byte k of the lane's 24-byte view: an odd window,
        // a dword, or a dword shifted by 16 (only bits [4:0] are consumed)
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
            uint32_t m_ = 0;                                                                     \
            if (!(ablate & 4)) {                                                                 \
                _Pragma("unroll") for (int j = 0; j < 16; j += 2) {                              \
                    const uint32_t W_ = w_[j + 1] & GMASK;                                       \
                    const uint32_t H_ = hash_mul24(W_, HASH_K1) + W_;                            \
                    const uint2 e_ = *(const uint2 *)((const uint8_t *)L.xy +                    \
                        ((H_ >> (32 - FILTER_ENTRIES_LOG2 - 3)) & ((FILTER_WORDS * 4 - 1) & ~7u))); \
                    /* byte j: low byte of d_[j / 4] (j % 4 == 0) or of d_ >> 16 (j % 4 == 2) */  \
                    const uint32_t bx_ = K1B_BYTE_REG(j);                                        \
                    const uint32_t by_ = K1B_BYTE_REG(j + Q);                                    \
                    const uint32_t g_ = e_.x >> (W_ & 31); /* the gate both tests share */       \
                    const uint32_t tx_ = (e_.x >> (bx_ & 31)) & g_;                              \
                    const uint32_t ty_ = (e_.y >> (by_ & 31)) & g_;                              \
                    m_ = __builtin_amdgcn_alignbit(tx_, m_, 1); /* position j:   X, byte j   */  \
                    m_ = __builtin_amdgcn_alignbit(ty_, m_, 1); /* position j+1: Y, byte j+Q */  \
                }                                                                                \
                m_ >>= 16;                                                                       \
            } else {                                                                             \
                m_ = (d_[0] ^ d_[1] ^ d_[2] ^ d_[3] ^ d_[4]) == 0x12345678u ? 1u : 0u;           \
            }                                                                                    \
            if (!interior) {                                                                     \
                const uint64_t p0_ = tbase + (uint64_t)(RI) * 1024 + lane * 16;                  \
                uint32_t keep_ = 0;                                                              \
                _Pragma("unroll") for (int j = 0; j < 16; j++)                                   \
                    if (p0_ + j >= lead && p0_ + j <= last_start) keep_ |= 1u << j;              \
                m_ &= keep_;                                                                     \
            }                                                                                    \
            MROW = m_;                                                                           \
        }
        // every position of an interior tile is a legal start: no per-row masking
        const bool interior = tbase >= lead && tbase + tile_bytes <= last_start;
        K1B_ROW(0, v0, __builtin_amdgcn_readfirstlane(v1.x), __builtin_amdgcn_readfirstlane(v1.y), mrow0)
        K1B_ROW(1, v1, __builtin_amdgcn_readfirstlane(v2.x), __builtin_amdgcn_readfirstlane(v2.y), mrow1)
        K1B_ROW(2, v2, __builtin_amdgcn_readfirstlane(v3.x), __builtin_amdgcn_readfirstlane(v3.y), mrow2)
        K1B_ROW(3, v3, vL.x, vL.y, mrow3)
        // Prefetch of the wave's next tile, issued LATE: the compaction below, the level-2 phases
        // at the top of the next iteration and the three other waves of the SIMD cover its
        // latency.  Measured (K1b, T): issued before row 0: 310 us; after row 1: 299; after row 2:
        // 293; here: 291; no prefetch at all (loads at the top of the tile's own iteration): 308.
        // The kernel runs within 5 % of the streaming ceiling of the fabric, and five 16-byte
        // loads per lane that sit in flight for a whole iteration are in the way of everything
        // else in the memory pipeline.
        K1B_ISSUE_TILE(tile + nw)
        // ---- ballot-compact the survivors of the tile into Q1, one per lane per round
        uint32_t mlo = mrow0 | (mrow1 << 16), mhi = mrow2 | (mrow3 << 16);
        while (true) {
            unsigned long long act = __ballot((mlo | mhi) != 0);
            if (!act) break;
            uint32_t np = __popcll(act);
            if (q1c + np > K1B_Q1CAP) {
                // queue pressure (dense survivors): hand everything in Q1 to the walk
                // kernel unprobed (it probes the prefix table itself)
                bool found = lane < q1c;
                uint64_t p = found ? tbase + q1[lane] - lead : 0;
                uint64_t a0 = found ? load_window(stream, len, p) : 0;
                uint64_t a1 = found ? load_window(stream, len, p + 8) : 0;
                K1B_HIT_PUSH(found, p, HIT_RETRY, a0, a1)
                q1c = 0;
            }
            if (mlo | mhi) {
                uint32_t pos;
                if (mlo) { pos = __builtin_ctz(mlo); mlo &= mlo - 1; }
                else { pos = 32 + __builtin_ctz(mhi); mhi &= mhi - 1; }
                uint32_t slot = q1c + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                q1[slot] = (uint16_t)(((pos >> 4) << 10) + lane * 16 + (pos & 15)); // offset in the tile
            }
            q1c += np;
            __builtin_amdgcn_wave_barrier();
        }
    }
    K1B_HIT_FLUSH
    if (lane == 0) GK.block_counts[region] = hcur;
#undef K1B_HIT_FLUSH
#undef K1B_ISSUE_ROW
#undef K1B_LOAD16
#undef K1B_ISSUE_TILE
#undef K1B_HIT_PUSH
#undef K1B_ROW
#undef K1B_BYTE_REG
}

size_t prefilter_lds_bytes() { return sizeof(K1bLds); }

uint32_t prefilter_grid(const uint8_t *d_hay, uint64_t len, int n_cus) {
    uint64_t lead = (uintptr_t)d_hay & 15;
    uint64_t total = lead + len;
    uint64_t ntiles = (total + (uint64_t)K1B_ROWS * 1024 - 1) / ((uint64_t)K1B_ROWS * 1024);
    uint64_t blocks = (ntiles + 15) / 16;
    if (blocks > (uint64_t)n_cus) blocks = n_cus;
    return blocks ? (uint32_t)blocks : 1;
}

static uint32_t ablation_flags() {
    static int v = -1;
    if (v < 0) {
        const char *e = std::getenv("ACX_ABLATE"); // profiling only: 1 = level 1 only, 2 = skip level 3, 4 = loads only
        v = e ? std::atoi(e) : 0;
    }
    return (uint32_t)v;
}

uint32_t walk_hits_grid(uint32_t hit_regions) { return hit_regions * K_WALK_SPLIT; }

hipError_t launch_walk_hits(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G,
                            const Sink &hits, uint32_t hit_grid, uint32_t split, const Sink &occ,
                            const uint8_t *d_hay, uint64_t len, hipStream_t st) {
    if (split == 0 || split > K_WALK_SPLIT) split = K_WALK_SPLIT;
    hipLaunchKernelGGL(k_walk_hits, dim3(hit_grid * split), dim3(256), 0, st, A, Ad, G, hits, hit_grid,
                       split, occ, d_hay, len, ablation_flags());
    return hipGetLastError();
}

uint32_t prefilter_hit_regions(uint32_t grid) { return grid * 16; } // one per wave

uint64_t prefilter_tiles(const uint8_t *d_hay, uint64_t len) {
    const uint64_t total = ((uintptr_t)d_hay & 15) + len;
    return (total + (uint64_t)K1B_ROWS * 1024 - 1) / ((uint64_t)K1B_ROWS * 1024);
}

hipError_t launch_prefilter(const DevAutomaton &A, const DevAutomaton *Ad, const Segments &G,
                            const Sink &K, const uint8_t *d_hay, uint64_t len, uint32_t grid,
                            uint64_t tile_begin, uint64_t tile_end, hipStream_t st, hipEvent_t ev_start,
                            hipEvent_t ev_stop) {
    if (len == 0 || A.filter_q == 0) return hipSuccess;
    uint64_t lead = (uintptr_t)d_hay & 15;
    const uint8_t *base = d_hay - lead;
    dim3 g(grid), b(1024);
    uint32_t ab = ablation_flags();
    // the events (measurement only) ride on the dispatch itself: no barrier packets, no gaps
#define ACX_K1B(Q)                                                                                    \
    hipExtLaunchKernelGGL(k1b_prefilter<Q>, g, b, 0, st, ev_start, ev_stop, 0, A, Ad, G, K, base, len, lead, \
                          tile_begin, tile_end, ab)
    switch (A.filter_q) {
    case 1: ACX_K1B(1); break;
    case 2: ACX_K1B(2); break;
    case 3: ACX_K1B(3); break;
    case 4: ACX_K1B(4); break;
    default: ACX_K1B(5); break;
    }
#undef ACX_K1B
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
    const uint4 *r = recs + (uint64_t)blockIdx.x * region_cap;
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
    size_t m = a > b ? a : b;
    return m > c ? m : c;
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
// K2b: sparse path -- tile kernels over the bucket slots
// ---------------------------------------------------------------------------
// In slot mode the scan's emission has already grouped the occurrences by 4 KiB bucket
// of their key position (Sink::slots).  The launch geometry of everything that follows
// depends only on the number of buckets (known to the host), never on the number of
// occurrences (known only to the device), so the whole post stage is queued behind the
// scan without a host round trip.  A workgroup owns a tile of TILE_BUCKETS consecutive
// buckets (256 KiB of stream position); its occurrences (at most TILE_MAX, else the
// abort flag -> region mode + radix sort) are staged in LDS so that the per-bucket serial
// work runs at LDS latency and every global access is coalesced.  These kernels are
// latency chains (a few dependent HBM round trips per tile), so the workgroups are small
// (two waves, 16 KiB of LDS).
//   k_tile_sort     gather the slots, per-bucket insertion sort, sync-point flag of
//                   every occurrence
//   k_tile_resolve  greedy chains, re-derived per bucket from the nearest sync point;
//                   reported count of the tile
//   k_tile_scan     exclusive scan of the tile counts, totals (one workgroup)
//   k_tile_write    compaction into the final (pattern, start, end) records
constexpr uint32_t DST_NONE = 0xFFFFFFFFu;
constexpr uint32_t TILE_THREADS = 128;
constexpr uint32_t TILE_PER_THREAD = TILE_MAX / TILE_THREADS; // k_tile_write
static_assert(TILE_BUCKETS <= 64, "one wave owns the buckets of a tile");
static_assert(TILE_THREADS % TILE_BUCKETS == 0 && TILE_PER_THREAD % 4 == 0, "tile geometry");

// span of an occurrence record {key lo, key hi, pid, pattern length}
__device__ __forceinline__ void span_of(uint32_t rank_bits, int key_mode, uint4 v, uint64_t *s, uint64_t *e) {
    const uint64_t x = (((uint64_t)v.y << 32) | v.x) >> rank_bits;
    if (key_mode == 0) { *e = x; *s = x - v.w; }
    else { *s = x; *e = x + v.w; }
}

// bo[0 .. TILE_BUCKETS] = offsets of the tile's buckets inside the tile (wave 0; the caller syncs)
__device__ __forceinline__ void tile_offsets(const TileSpace &T, uint32_t B0, uint32_t *bo) {
    const uint32_t t = threadIdx.x;
    if (t < 64) {
        uint32_t c = t < TILE_BUCKETS && B0 + t < T.n_buckets ? T.bcnt[B0 + t] : 0;
        c = c < BUCKET_SLOTS ? c : BUCKET_SLOTS; // overfull: the emitter raised the abort flag
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)t >= o) incl += v;
        }
        if (t < TILE_BUCKETS) bo[t + 1] = incl;
        if (t == 0) bo[0] = 0;
    }
}

// largest end among the (unsorted) occurrences of the buckets [b0, b1)
__device__ __forceinline__ uint64_t raw_max_end(uint32_t rank_bits, int key_mode, const TileSpace &T,
                                                uint32_t b0, uint32_t b1) {
    uint64_t mx = 0;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t c = T.bcnt[b];
        c = c < BUCKET_SLOTS ? c : BUCKET_SLOTS;
        for (uint32_t r = 0; r < c; r++) {
            uint64_t s, e;
            span_of(rank_bits, key_mode, T.slots[(uint64_t)b * BUCKET_SLOTS + r], &s, &e);
            mx = max(mx, e);
        }
    }
    return mx;
}

// Occurrence i is a "sync point" when every earlier occurrence (in sorted order) ends at or
// before its start: whatever the greedy did before, i is reported.
__global__ __launch_bounds__(TILE_THREADS) void k_tile_sort(uint32_t rank_bits, uint32_t max_len,
                                                            int key_mode, int overlapping, TileSpace T,
                                                            uint32_t tile0, uint32_t *abort_flag) {
    __shared__ uint64_t k[TILE_MAX];
    __shared__ uint2 pl[TILE_MAX]; // {pid, pattern length}
    __shared__ uint32_t bo[TILE_BUCKETS + 1];
    __shared__ uint32_t stop;
    const uint32_t t = threadIdx.x, tile = tile0 + blockIdx.x, B0 = tile * TILE_BUCKETS;
    tile_offsets(T, B0, bo);
    if (t == 0) stop = *abort_flag;
    __syncthreads();
    const uint32_t n = bo[TILE_BUCKETS];
    if (stop) return;
    if (n > TILE_MAX) { if (t == 0) *abort_flag = 1; return; }
    if (t == 0) T.tile_n[tile] = n;
    { // gather: TILE_THREADS / 64 threads per bucket
        constexpr uint32_t TPB = TILE_THREADS / TILE_BUCKETS;
        const uint32_t bq = t / TPB, a = bo[bq], c = bo[bq + 1] - a;
        for (uint32_t r = t % TPB; r < c; r += TPB) {
            const uint4 v = T.slots[(uint64_t)(B0 + bq) * BUCKET_SLOTS + r];
            k[a + r] = ((uint64_t)v.y << 32) | v.x;
            pl[a + r] = make_uint2(v.z, v.w);
        }
    }
    __syncthreads();
    if (t < TILE_BUCKETS) {
        const uint32_t a = bo[t], e = bo[t + 1];
        for (uint32_t i = a + 1; i < e; i++) {
            const uint64_t kk = k[i];
            const uint2 pp = pl[i];
            uint32_t j = i;
            while (j > a && k[j - 1] > kk) { k[j] = k[j - 1]; pl[j] = pl[j - 1]; j--; }
            k[j] = kk; pl[j] = pp;
        }
    }
    __syncthreads();
    const uint64_t gi = (uint64_t)tile * TILE_MAX;
    for (uint32_t i = t; i < n; i += TILE_THREADS) {
        const uint64_t key = k[i];
        const uint2 pp = pl[i];
        T.trecs[gi + i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pp.x, pp.y);
        if (overlapping) continue;
        const uint64_t x = key >> rank_bits;
        const uint64_t s = key_mode == 0 ? x - pp.y : x;
        uint64_t mx = 0; // largest end among the occurrences before i
        if (key_mode == 0) { // sorted by end: the previous end is the maximum
            if (i > 0) mx = k[i - 1] >> rank_bits;
            else if ((s >> BUCKET_BITS) < B0) // anything before the tile that ends beyond s ends in these buckets
                mx = raw_max_end(rank_bits, key_mode, T, (uint32_t)(s >> BUCKET_BITS), B0);
        } else { // sorted by start: only occurrences that start within max_len of s can end beyond it
            bool open = true;
            for (uint32_t j = i; j > 0;) {
                j--;
                const uint64_t sj = k[j] >> rank_bits;
                if (sj + max_len <= s) { open = false; break; }
                mx = max(mx, sj + pl[j].y);
            }
            const uint64_t p0 = s > max_len ? s - max_len : 0;
            if (open && (p0 >> BUCKET_BITS) < B0)
                mx = max(mx, raw_max_end(rank_bits, key_mode, T, (uint32_t)(p0 >> BUCKET_BITS), B0));
        }
        T.syncf[gi + i] = mx <= s ? 1 : 0;
    }
}

__global__ __launch_bounds__(TILE_THREADS) void k_tile_resolve(uint32_t rank_bits, int key_mode,
                                                               int overlapping, TileSpace T, uint32_t tile0,
                                                               const uint32_t *abort_flag) {
    __shared__ uint32_t rel[TILE_MAX]; // key position relative to the tile's first byte
    __shared__ uint32_t len[TILE_MAX];
    __shared__ uint8_t sy[TILE_MAX], ac[TILE_MAX];
    __shared__ uint32_t bo[TILE_BUCKETS + 1];
    __shared__ uint32_t stop;
    const uint32_t t = threadIdx.x, tile = tile0 + blockIdx.x, B0 = tile * TILE_BUCKETS;
    tile_offsets(T, B0, bo);
    if (t == 0) stop = *abort_flag;
    __syncthreads();
    if (stop) return;
    const uint32_t n = bo[TILE_BUCKETS];
    const uint64_t base = (uint64_t)B0 << BUCKET_BITS, gi = (uint64_t)tile * TILE_MAX;
    uint32_t cnt = 0; // reported occurrences of bucket t (wave 0 only)
    if (overlapping) {
        if (t < TILE_BUCKETS) cnt = bo[t + 1] - bo[t];
    } else {
        for (uint32_t i = t; i < n; i += TILE_THREADS) {
            const uint4 v = T.trecs[gi + i];
            rel[i] = (uint32_t)(((((uint64_t)v.y << 32) | v.x) >> rank_bits) - base);
            len[i] = v.w;
            sy[i] = T.syncf[gi + i];
        }
        __syncthreads();
        const uint32_t a = t < TILE_BUCKETS ? bo[t] : 0, e = t < TILE_BUCKETS ? bo[t + 1] : 0;
        if (e > a) {
            // re-derive the greedy chain from the nearest sync point at or before a
            uint32_t j = a;
            while (j > 0 && !sy[j]) j--;
            uint64_t pos = 0; // end of the last reported match
            if (!sy[j]) { // the chain enters the tile from before it: follow it in HBM (rare)
                uint32_t tt = tile, q = 0;
                for (;;) { // the first occurrence of the stream is a sync point: this terminates
                    while (q == 0) q = T.tile_n[--tt];
                    q--;
                    if (T.syncf[(uint64_t)tt * TILE_MAX + q]) break;
                }
                for (; tt < tile; tt++, q = 0) {
                    const uint32_t nn = T.tile_n[tt];
                    for (; q < nn; q++) {
                        uint64_t s, en;
                        span_of(rank_bits, key_mode, T.trecs[(uint64_t)tt * TILE_MAX + q], &s, &en);
                        if (s >= pos) pos = en;
                    }
                }
            }
            for (uint32_t q = j; q < e; q++) {
                const uint64_t x = base + rel[q], l = len[q];
                const uint64_t s = key_mode == 0 ? x - l : x, en = key_mode == 0 ? x : x + l;
                const bool take = s >= pos;
                if (take) pos = en;
                if (q >= a) { ac[q] = take; cnt += take; }
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < n; i += TILE_THREADS) T.accf[gi + i] = ac[i];
    }
    if (t < 64) {
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if (t == 0) T.btot[tile] = cnt;
    }
}

// One workgroup, for the tiles [tile0, tile1) of one chunk of the call (chunks run in order):
// bbase = exclusive scan of btot continuing the previous chunk's; summary[0] = occurrences,
// summary[4] = reported matches, summary[2] / [3] = prefix hits kept / largest hit region, all
// accumulated over the chunks.  The last chunk clears the abort flag the NEXT call will use and
// publishes {[0] occurrences, [2] hits, [3] largest hit region, [4] matches, [5] aborted} to
// host_out (pinned host memory).
__global__ __launch_bounds__(1024) void k_tile_scan(TileSpace T, uint32_t tile0, uint32_t tile1, int first,
                                                    const uint64_t *hit_counts, uint32_t hit_grid,
                                                    uint64_t hit_cap, uint64_t *summary,
                                                    const uint32_t *abort_flag, uint32_t *next_flag,
                                                    uint64_t *host_out) {
    using scan_t = rocprim::block_scan<uint32_t, 1024>;
    __shared__ typename scan_t::storage_type scan_tmp;
    __shared__ uint64_t red[3][16];
    const uint32_t t = threadIdx.x;
    const bool stop = *abort_flag != 0; // stable: its writers completed
    const uint64_t n_before = first ? 0 : summary[0], h_before = first ? 0 : summary[2],
                   hmax_before = first ? 0 : summary[3];
    const uint32_t base = first ? 0 : (uint32_t)summary[4];
    uint64_t hsum = 0, hmax = 0, nsum = 0;
    if (hit_counts)
        for (uint32_t g = t; g < hit_grid; g += 1024) {
            const uint64_t c = hit_counts[g];
            hmax = max(hmax, c);
            hsum += c < hit_cap ? c : hit_cap;
        }
    const uint32_t per = (tile1 - tile0 + 1023) / 1024, g0 = tile0 + t * per;
    uint32_t mine = 0, excl = 0;
    // (unrolled so that the loads of several tiles are in flight together)
    if (!stop)
        _Pragma("unroll 8") for (uint32_t g = g0; g < g0 + per && g < tile1; g++) {
            mine += T.btot[g];
            nsum += T.tile_n[g];
        }
    scan_t().exclusive_scan(mine, excl, 0u, scan_tmp);
    excl += base;
    if (!stop)
        _Pragma("unroll 8") for (uint32_t g = g0; g < g0 + per && g < tile1; g++) {
            T.bbase[g] = excl;
            excl += T.btot[g];
        }
    for (int o = 32; o > 0; o >>= 1) {
        hsum += __shfl_down(hsum, o);
        nsum += __shfl_down(nsum, o);
        hmax = max(hmax, __shfl_down(hmax, o));
    }
    if ((t & 63) == 0) { red[0][t >> 6] = hsum; red[1][t >> 6] = hmax; red[2][t >> 6] = nsum; }
    __syncthreads();
    if (t == 1023) {
        summary[4] = excl;
        if (host_out) host_out[4] = excl;
    }
    if (t == 0) {
        uint64_t a = h_before, b = hmax_before, c = n_before;
        for (int i = 0; i < 16; i++) { a += red[0][i]; b = max(b, red[1][i]); c += red[2][i]; }
        summary[2] = a; summary[3] = b; summary[0] = c;
        if (next_flag) *next_flag = 0;
        // last chunk: the totals go straight to pinned host memory (no copy operation)
        if (host_out) { host_out[0] = c; host_out[2] = a; host_out[3] = b; host_out[5] = stop ? 1 : 0; }
    }
}

// zero_lo..zero_hi: the arrival counters this launch leaves clean for the next call
// seg_counts != null (batch of haystacks, byte offsets): the records get offsets local to
// their haystack and the per-haystack counts are taken here -- one atomic per run of matches
// of the same haystack inside the tile instead of a separate pass with one atomic per match.
__global__ __launch_bounds__(TILE_THREADS) void k_tile_write(uint32_t rank_bits, int key_mode, int overlapping,
                                                             TileSpace T, uint32_t tile0, uint32_t zero_lo,
                                                             uint32_t zero_hi, acx_match_t *out,
                                                             const uint32_t *abort_flag, Segments G,
                                                             uint64_t *seg_counts) {
    __shared__ __attribute__((aligned(16))) uint8_t ac[TILE_MAX];
    __shared__ uint32_t dst[TILE_MAX];
    using scan_t = rocprim::block_scan<uint32_t, TILE_THREADS>;
    __shared__ typename scan_t::storage_type scan_tmp;
    const uint32_t t = threadIdx.x, tile = tile0 + blockIdx.x;
    for (uint32_t i = zero_lo + blockIdx.x * TILE_THREADS + t; i < zero_hi; i += gridDim.x * TILE_THREADS)
        T.bcnt[i] = 0;
    if (*abort_flag) return; // stable by now: its writers completed
    const uint32_t n = T.tile_n[tile], base = T.bbase[tile];
    const uint64_t gi = (uint64_t)tile * TILE_MAX;
    if (overlapping) {
        for (uint32_t i = t; i < n; i += TILE_THREADS) dst[i] = base + i;
    } else {
        for (uint32_t i = t; i < TILE_MAX; i += TILE_THREADS) ac[i] = i < n ? T.accf[gi + i] : 0;
        __syncthreads();
        // thread t owns the occurrences [t * TILE_PER_THREAD, (t + 1) * TILE_PER_THREAD)
        uint32_t mine = 0, excl = 0;
        _Pragma("unroll") for (uint32_t w = 0; w < TILE_PER_THREAD / 4; w++)
            mine += __popc(*(const uint32_t *)&ac[t * TILE_PER_THREAD + 4 * w]);
        scan_t().exclusive_scan(mine, excl, 0u, scan_tmp);
        uint32_t d = base + excl;
        for (uint32_t q = t * TILE_PER_THREAD; q < (t + 1) * TILE_PER_THREAD; q++) dst[q] = ac[q] ? d++ : DST_NONE;
    }
    __syncthreads();
    __shared__ uint32_t hs[TILE_MAX]; // haystack index of the tile's reported matches, in output order
    for (uint32_t i = t; i < n; i += TILE_THREADS) {
        const uint32_t d = dst[i];
        if (d == DST_NONE) continue;
        const uint4 v = T.trecs[gi + i];
        uint64_t s, e;
        span_of(rank_bits, key_mode, v, &s, &e);
        if (seg_counts) {
            uint64_t h, hbase;
            if (G.uniform_len) { h = s / G.uniform_len; hbase = h * G.uniform_len; }
            else { h = upper_bound_u64(G.offsets, G.n_hay + 1, s) - 1; hbase = G.offsets[h]; }
            s -= hbase; e -= hbase;
            hs[d - base] = (uint32_t)h;
        }
        out[d].pattern = v.z; out[d].start = s; out[d].end = e;
    }
    if (seg_counts) {
        __syncthreads();
        const uint32_t total = T.btot[tile];
        for (uint32_t c = t; c < total; c += TILE_THREADS) {
            const uint32_t h = hs[c];
            if (c > 0 && hs[c - 1] == h) continue; // not the head of its run
            uint32_t run = 1;
            while (c + run < total && hs[c + run] == h) run++;
            atomicAdd((unsigned long long *)&seg_counts[h], (unsigned long long)run);
        }
    }
}

// Sort (within buckets), resolve and compact the slotted occurrences of the tiles
// [tile0, tile1) into out[] (capacity n_tiles * TILE_MAX suffices for a whole call).  Chunks of
// one call come in tile order; `first` / `last` mark the ends.  summary[0] = occurrences,
// [2] = prefix hits kept, [3] = largest hit region, [4] = matches written (accumulated over the
// chunks); *abort_flag != 0: the output did not fit the sparse path (or hits were dropped) and
// out[] / summary[0], [4] are meaningless.  The last chunk zeroes T.bcnt and *next_flag.
hipError_t tile_post(const DevAutomaton &A, int key_mode, bool overlapping, const TileSpace &T,
                     uint32_t tile0, uint32_t tile1, bool first, bool last, const uint64_t *hit_counts,
                     uint32_t hit_grid, uint64_t hit_cap, acx_match_t *out, uint64_t *summary,
                     uint32_t *abort_flag, uint32_t *next_flag, uint64_t *host_out, const Segments &G,
                     uint64_t *seg_counts, hipStream_t st) {
    const int ov = overlapping ? 1 : 0;
    const uint32_t tiles = tile1 - tile0;
    if (tiles) {
        hipLaunchKernelGGL(k_tile_sort, dim3(tiles), dim3(TILE_THREADS), 0, st, A.rank_bits, A.max_len, key_mode,
                           ov, T, tile0, abort_flag);
        hipLaunchKernelGGL(k_tile_resolve, dim3(tiles), dim3(TILE_THREADS), 0, st, A.rank_bits, key_mode, ov, T,
                           tile0, abort_flag);
    }
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, T, tile0, tile1, first ? 1 : 0, hit_counts,
                       hit_grid, hit_cap, summary, abort_flag, last ? next_flag : nullptr,
                       last ? host_out : nullptr);
    if (tiles)
        hipLaunchKernelGGL(k_tile_write, dim3(tiles), dim3(TILE_THREADS), 0, st, A.rank_bits, key_mode, ov, T,
                           tile0, 0u, last ? T.n_buckets + 1 : 0u, out, abort_flag, G, seg_counts);
    return hipGetLastError();
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
// (zero-copy): the host-memory entry point then costs one launch and one sync.
// res[0] = matches written, res[1] = 0, or 1: too many occurrences, nothing written.
__global__ __launch_bounds__(1024) void k0_small(DevAutomaton A, const uint8_t *__restrict__ hay,
                                                 uint32_t len, int key_mode, int overlapping,
                                                 int codepoints, acx_match_t *out, uint64_t *res) {
    __shared__ __attribute__((aligned(16))) uint8_t sh[SMALL_MAX_LEN + 16];
    __shared__ uint8_t cls[256];
    __shared__ uint4 occ[SMALL_MAX_OCC]; // {key lo, key hi, pid, pattern length}
    __shared__ uint16_t order[SMALL_MAX_OCC];
    __shared__ uint8_t syn[SMALL_MAX_OCC], acc[SMALL_MAX_OCC];
    __shared__ uint32_t cpre[1025]; // code points before every 16-byte slice
    __shared__ uint32_t nocc;
    using scan_t = rocprim::block_scan<uint32_t, 1024>;
    __shared__ typename scan_t::storage_type scan_tmp;
    const uint32_t t = threadIdx.x;
    if (t == 0) nocc = 0;
    if (t < 256) cls[t] = A.classes[t];
    for (uint32_t i = t; i < len; i += 1024) sh[i] = hay[i];
    __syncthreads();
    // ---- all occurrences: anchored walk from every position
    for (uint32_t pos = t; pos < len; pos += 1024) {
        uint32_t s = 0;
        for (uint32_t d = 0; pos + d < len;) {
            const uint32_t e = A.table[((size_t)s << A.stride2) + cls[sh[pos + d]]];
            const uint32_t id = e & ID_MASK;
            d++;
            if (id < A.level_start[d]) break; // shallower than d: a failure transition, not an edge
            s = id;
            if (!(e & FLAG_OWN)) continue;
            const uint32_t one = A.own1[s];
            uint32_t b = 0, en = 1;
            if (one == OWN1_MANY) { b = A.own_off[s]; en = A.own_off[s + 1]; }
            for (uint32_t k = b; k < en; k++) { // the patterns that are exactly hay[pos, pos + d)
                const uint32_t pid = one == OWN1_MANY ? A.own_pid[k] : one;
                const uint64_t key = key_mode == 0   ? ((uint64_t)(pos + d) << A.rank_bits) | A.rank[pid]
                                     : key_mode == 1 ? ((uint64_t)pos << A.rank_bits) | pid
                                                     : ((uint64_t)pos << A.rank_bits) | A.rank[pid];
                const uint32_t slot = atomicAdd(&nocc, 1u);
                if (slot < SMALL_MAX_OCC) occ[slot] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), pid, d);
            }
        }
    }
    __syncthreads();
    const uint32_t n = nocc;
    if (n > SMALL_MAX_OCC) { // dense: the general pipeline takes the call
        if (t == 0) { res[0] = 0; res[1] = 1; }
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
    if (codepoints) {
        uint32_t leads = 0;
        for (uint32_t i = t * 16; i < t * 16 + 16 && i < len; i++) leads += (sh[i] & 0xC0) != 0x80;
        uint32_t before = 0;
        scan_t().exclusive_scan(leads, before, 0u, scan_tmp);
        cpre[t] = before;
        if (t == 1023) cpre[1024] = before + leads;
        __syncthreads();
    }
    uint32_t dst = 0;
    scan_t().exclusive_scan(mine, dst, 0u, scan_tmp);
    if (mine) {
        const uint4 v = occ[order[t]];
        uint64_t s, e;
        K0_SPAN(t, s, e)
        if (codepoints) {
            uint32_t cs = cpre[s >> 4], ce = cpre[e >> 4];
            for (uint32_t i = (uint32_t)s & ~15u; i < s; i++) cs += (sh[i] & 0xC0) != 0x80;
            for (uint32_t i = (uint32_t)e & ~15u; i < e; i++) ce += (sh[i] & 0xC0) != 0x80;
            s = cs; e = ce;
        }
        out[dst].pattern = v.z; out[dst].start = s; out[dst].end = e;
    }
    if (t == 1023) { res[0] = dst + mine; res[1] = 0; }
#undef K0_SPAN
}

hipError_t launch_small(const DevAutomaton &A, const uint8_t *hay, uint32_t len, int key_mode, bool overlapping,
                        bool codepoints, acx_match_t *out, uint64_t *res, hipStream_t st) {
    hipLaunchKernelGGL(k0_small, dim3(1), dim3(1024), 0, st, A, hay, len, key_mode, overlapping ? 1 : 0,
                       codepoints ? 1 : 0, out, res);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K3: UTF-8 code-point indexes
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lead_bytes_in_word(uint32_t w) {
    // continuation byte: bit7 = 1 and bit6 = 0
    uint32_t cont = w & 0x80808080u & ((~w) << 1);
    return 4 - __popc(cont);
}

// one wave per 1 KiB block
__global__ __launch_bounds__(256) void k_count_leads(const uint8_t *__restrict__ hay,
                                                     uint64_t len, uint64_t *cnt,
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
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane == 0) cnt[blk] = c;
}

hipError_t count_lead_bytes(const uint8_t *d_hay, uint64_t len, uint64_t *cnt, hipStream_t st) {
    uint64_t nblocks = (len + 1023) / 1024;
    uint64_t waves = nblocks + 1;
    hipLaunchKernelGGL(k_count_leads, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, d_hay,
                       len, cnt, nblocks);
    return hipGetLastError();
}

hipError_t prefix_sum_u64(void *temp, size_t temp_bytes, const uint64_t *in, uint64_t *out,
                          uint64_t n, hipStream_t st) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, (uint64_t)0, (size_t)n,
                                   rocprim::plus<uint64_t>(), st);
}

// non-continuation (lead) bytes in [p, end): bytes up to an 8-byte boundary, then words
__device__ __forceinline__ uint64_t lead_bytes_between(const uint8_t *p, const uint8_t *end) {
    uint64_t c = 0;
    while (p < end && ((uintptr_t)p & 7)) { c += (*p & 0xC0) != 0x80; p++; }
    for (; p + 8 <= end; p += 8) {
        const uint64_t w = *(const uint64_t *)p;
        c += 8 - __popcll(w & 0x8080808080808080ull & ((~w) << 1)); // continuation: bit7 = 1, bit6 = 0
    }
    for (; p < end; p++) c += (*p & 0xC0) != 0x80;
    return c;
}

// code-point index of byte offset x = number of non-continuation bytes in [0, x)
__device__ __forceinline__ uint64_t code_point_of(const uint8_t *__restrict__ hay,
                                                  const uint64_t *blockpre, uint64_t x) {
    const uint64_t blk = x >> 10;
    return blockpre[blk] + lead_bytes_between(hay + (blk << 10), hay + x);
}

// one thread per match: the start from its 1 KiB block's prefix, the end from the start
__global__ void k_to_code_points(const uint8_t *__restrict__ hay, const uint64_t *blockpre,
                                 acx_match_t *m, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = m[i].start, e = m[i].end;
    const uint64_t cs = code_point_of(hay, blockpre, s);
    m[i].start = cs;
    m[i].end = cs + lead_bytes_between(hay + s, hay + e);
}

hipError_t to_code_points(const uint8_t *d_hay, uint64_t len, const uint64_t *blockpre,
                          acx_match_t *m, uint64_t n, hipStream_t st) {
    (void)len;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_to_code_points, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                       d_hay, blockpre, m, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// batch: local offsets + per-haystack counts
// ---------------------------------------------------------------------------
__global__ void k_localize(Segments G, const uint8_t *__restrict__ hay, uint64_t len,
                           const uint64_t *blockpre, int codepoints, acx_match_t *m,
                           uint64_t n, uint64_t *counts) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = m[i].start, e = m[i].end;
    uint64_t h, base;
    if (G.uniform_len) { h = s / G.uniform_len; base = h * G.uniform_len; }
    else { h = upper_bound_u64(G.offsets, G.n_hay + 1, s) - 1; base = G.offsets[h]; }
    (void)len;
    if (codepoints) {
        const uint64_t cs = code_point_of(hay, blockpre, s) - code_point_of(hay, blockpre, base);
        m[i].start = cs;
        m[i].end = cs + lead_bytes_between(hay + s, hay + e);
    } else {
        m[i].start = s - base;
        m[i].end = e - base;
    }
    atomicAdd((unsigned long long *)&counts[h], 1ull);
}

hipError_t localize(const Segments &G, const uint8_t *d_hay, uint64_t len,
                    const uint64_t *blockpre, int codepoints, acx_match_t *m, uint64_t n,
                    uint64_t *counts, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_localize, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, G, d_hay,
                       len, blockpre, codepoints, m, n, counts);
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
