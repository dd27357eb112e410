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

#include <cstring>

#include <rocprim/rocprim.hpp>

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
    uint64_t *keys;
    uint32_t *pids;
    uint32_t *lcount; // LDS
    uint64_t region_cap;
    int key_mode;
};

__device__ __forceinline__ BlockSink block_sink(const Sink &K, uint32_t *lcount) {
    return BlockSink{K.keys + (uint64_t)blockIdx.x * K.region_cap,
                     K.pids + (uint64_t)blockIdx.x * K.region_cap, lcount, K.region_cap,
                     K.key_mode};
}

__device__ __forceinline__ void emit_one(const DevAutomaton &A, const BlockSink &K, uint32_t pid,
                                         uint64_t end) {
    uint64_t key;
    if (K.key_mode == 0) {
        key = (end << 24) | A.rank[pid];
    } else {
        uint64_t start = end - A.plen[pid];
        key = (start << 24) | (K.key_mode == 1 ? pid : A.rank[pid]);
    }
    uint32_t slot = atomicAdd(K.lcount, 1u); // LDS atomic
    if (slot < K.region_cap) {
        K.keys[slot] = key;
        K.pids[slot] = pid;
    }
}

// every pattern that ends at state s (own patterns, then the dictionary-suffix
// chain: progressively shorter suffixes)
__device__ __noinline__ void emit_state(const DevAutomaton &A, const BlockSink &K, uint32_t s,
                                        uint64_t end) {
    for (uint32_t t = s; t != NONE; t = A.dlink[t]) {
        uint32_t b = A.own_off[t], e = A.own_off[t + 1];
        for (uint32_t k = b; k < e; k++) emit_one(A, K, A.own_pid[k], end);
    }
}

// only the patterns that end exactly at state s with depth(s) == length
__device__ __noinline__ void emit_own(const DevAutomaton &A, const BlockSink &K, uint32_t s,
                                      uint64_t end) {
    uint32_t b = A.own_off[s], e = A.own_off[s + 1];
    for (uint32_t k = b; k < e; k++) emit_one(A, K, A.own_pid[k], end);
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
__device__ __forceinline__ uint32_t walk_span(const DevAutomaton &A, const WalkCtx &W,
                                              const BlockSink &K, const uint8_t *hay, uint64_t pos,
                                              uint64_t lim, uint32_t s) {
#define ACX_STEP(BYTE, POS)                                            \
    {                                                                  \
        uint32_t e_ = dfa_step(A, W, s, (BYTE));                       \
        s = e_ & ID_MASK;                                              \
        if (EMIT && (e_ & FLAG_OUT)) emit_state(A, K, s, (POS) + 1);   \
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

__global__ __launch_bounds__(1024) void k1a_dfa_walk(DevAutomaton A, Segments G, Sink GK,
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
            s = emit ? walk_span<true>(A, W, K, hay, pos, lim, s)
                     : walk_span<false>(A, W, K, hay, pos, lim, s);
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

hipError_t launch_dfa_walk(const DevAutomaton &A, const Segments &G, const Sink &K,
                           const uint8_t *d_hay, uint64_t len, uint32_t grid, size_t max_lds,
                           hipStream_t st) {
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
    hipLaunchKernelGGL(k1a_dfa_walk, dim3((uint32_t)blocks), dim3(1024), lds, st, A, G, K,
                       d_hay, len, chunk, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1b: LDS q-gram prefilter + anchored DFA verification
// ---------------------------------------------------------------------------
// Geometry: 1024-thread workgroups (16 waves), one per CU, persistent.
// A wave owns tiles of K1B_ROWS rows; one row = 64 lanes x 16 B = 1 KiB read by
// ONE coalesced global_load_dwordx4 per lane.  Lane l tests the 16 positions
// that start inside its 16 bytes; the bytes it needs beyond them (up to 5) come
// from lane l+1 by cross-lane shuffle.
//
// LDS: [0, 128 KiB) bitmap (1 bit per hashed q-gram of a pattern prefix),
//      then 256 B class map, then 16 per-wave candidate queues (128 x u64).
constexpr int K1B_ROWS = 4;
constexpr uint32_t K1B_QCAP = 128;
constexpr size_t K1B_LDS_BITMAP = (size_t)1 << (FILTER_BITS_LOG2 - 3);
constexpr size_t K1B_LDS_CLS = K1B_LDS_BITMAP;
constexpr size_t K1B_LDS_COUNT = K1B_LDS_CLS + 256;
constexpr size_t K1B_LDS_QUEUE = K1B_LDS_COUNT + 16;
constexpr size_t K1B_LDS_TOTAL = K1B_LDS_QUEUE + 16 * K1B_QCAP * 8;

__device__ __forceinline__ uint32_t hash_mul24(uint32_t a, uint32_t k) {
    return __umul24(a, k); // v_mul_u32_u24: uses bits [23:0] of each operand
}

// anchored verification of one candidate start position p
__device__ __forceinline__ void verify_candidate(const DevAutomaton &A, const Segments &G,
                                                 const BlockSink &K, const uint8_t *lcls,
                                                 const uint8_t *__restrict__ hay,
                                                 uint64_t len, uint64_t p) {
    uint64_t end = segment_end(G, len, p);
    uint64_t maxd = end - p;
    if (maxd > A.max_len) maxd = A.max_len;
    uint32_t s = 0;
    for (uint32_t d = 0; d < maxd; d++) {
        uint32_t c = lcls[hay[p + d]];
        uint32_t e = A.table[((size_t)s << A.stride2) + c];
        uint32_t t = e & ID_MASK;
        if (t < A.level_start[d + 1]) return; // not a trie edge: no pattern continues
        if (e & FLAG_OWN) emit_own(A, K, t, p + d + 1);
        s = t;
    }
}

template <int Q>
__global__ __launch_bounds__(1024) void k1b_prefilter(DevAutomaton A, Segments G, Sink GK,
                                                      const uint8_t *__restrict__ hay,
                                                      uint64_t len, uint64_t lead) {
    // `hay` is 16-byte aligned; the first `lead` bytes (< 16) precede the real
    // stream and are never candidates.  Stream position = index - lead.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lbits = smem;
    uint8_t *lcls = smem + K1B_LDS_CLS;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t *queue = (uint64_t *)(smem + K1B_LDS_QUEUE) + wave * K1B_QCAP;
    uint32_t *lcount = (uint32_t *)(smem + K1B_LDS_COUNT);
    if (threadIdx.x == 0) *lcount = 0;
    const BlockSink K = block_sink(GK, lcount);
    {
        const uint4 *src = (const uint4 *)A.filterA;
        uint4 *dst = (uint4 *)lbits;
        for (uint32_t i = threadIdx.x; i < K1B_LDS_BITMAP / 16; i += blockDim.x) dst[i] = src[i];
        for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x)
            ((uint32_t *)lcls)[i] = ((const uint32_t *)A.classes)[i];
    }
    __syncthreads();

    const uint64_t total = lead + len;          // bytes addressable from `hay`
    const uint64_t total16 = (total + 15) & ~15ull;
    // last index at which a pattern can still start
    const uint64_t last_start = total >= A.min_len ? total - A.min_len : 0;
    const bool any_start = total >= lead + A.min_len;
    const uint64_t tile_bytes = (uint64_t)K1B_ROWS * 1024;
    const uint64_t ntiles = (total + tile_bytes - 1) / tile_bytes;
    const uint64_t gw = (uint64_t)blockIdx.x * 16 + wave;
    const uint64_t nw = (uint64_t)gridDim.x * 16;
    uint32_t qcount = 0; // wave-uniform

    for (uint64_t tile = gw; tile < ntiles && any_start; tile += nw) {
        const uint64_t tbase = tile * tile_bytes;
        u32x4 v[K1B_ROWS + 1];
#pragma unroll
        for (int r = 0; r <= K1B_ROWS; r++) {
            uint64_t off = tbase + (uint64_t)r * 1024 + lane * 16;
            // row K1B_ROWS is only needed by lane 63 (its look-ahead = lane 0's bytes)
            bool need = (r < K1B_ROWS || lane == 0) && off < total16;
            v[r] = need ? load16_stream(hay + off) : (u32x4)(0u);
        }
#pragma unroll
        for (int r = 0; r < K1B_ROWS; r++) {
            // look-ahead dwords: lane l+1's first two dwords (lane 63: next row's lane 0)
            uint32_t nx = __shfl_down(v[r].x, 1), ny = __shfl_down(v[r].y, 1);
            uint32_t rx = __shfl(v[r + 1].x, 0), ry = __shfl(v[r + 1].y, 0);
            uint32_t d[6] = {v[r].x, v[r].y, v[r].z, v[r].w, lane == 63 ? rx : nx,
                             lane == 63 ? ry : ny};
            // 32-bit little-endian windows starting at byte j
            uint32_t w[16 + (Q > 4 ? (Q == 5 ? 1 : 3) : 0)];
#pragma unroll
            for (int j = 0; j < (int)(sizeof(w) / sizeof(w[0])); j++) {
                w[j] = (j & 3) == 0 ? d[j >> 2]
                                    : __builtin_amdgcn_alignbyte(d[(j >> 2) + 1], d[j >> 2], j & 3);
            }
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                uint32_t h;
                if (Q == 1) h = hash_mul24(w[j] & 0xFFu, HASH_K1);
                else if (Q == 2) h = hash_mul24(w[j] & 0xFFFFu, HASH_K1);
                else if (Q == 3) h = hash_mul24(w[j], HASH_K1);
                else if (Q == 4) h = hash_mul24(w[j], HASH_K1) + hash_mul24(w[j] >> 24, HASH_K2);
                else if (Q == 5) h = hash_mul24(w[j], HASH_K1) + hash_mul24(w[j + 1] >> 16, HASH_K2);
                else h = hash_mul24(w[j], HASH_K1) + hash_mul24(w[j + 3], HASH_K2);
                uint32_t byte = lbits[(h >> 12) & 0x1FFFFu]; // bits 12..28 -> byte of the bitmap
                uint32_t t = byte >> (h >> 29);              // bits 29..31 -> bit in that byte
                m = __builtin_amdgcn_alignbit(t, m, 1);      // shift the verdict in at bit 31
            }
            m >>= 16; // position j -> bit j
            // mask positions outside [lead, last_start]
            const uint64_t p0 = tbase + (uint64_t)r * 1024 + lane * 16;
            if (p0 < lead || p0 + 15 > last_start) {
                uint32_t keep = 0;
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (p0 + j >= lead && p0 + j <= last_start) keep |= 1u << j;
                m &= keep;
            }
            // ballot-compact survivors into the wave's LDS queue, one per lane per round
            while (true) {
                unsigned long long act = __ballot(m != 0);
                if (!act) break;
                if (m) {
                    uint32_t j = __builtin_ctz(m);
                    m &= m - 1;
                    uint32_t slot = qcount + __builtin_amdgcn_mbcnt_hi(
                                                 (uint32_t)(act >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                    queue[slot] = p0 + j - lead; // stream position
                }
                qcount += __popcll(act);
                __builtin_amdgcn_wave_barrier();
                if (qcount >= 64) {
                    qcount -= 64;
                    uint64_t p = queue[qcount + lane];
                    verify_candidate(A, G, K, lcls, hay + lead, len, p);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < qcount) {
        uint64_t p = queue[lane];
        verify_candidate(A, G, K, lcls, hay + lead, len, p);
    }
    __syncthreads();
    if (threadIdx.x == 0) GK.block_counts[blockIdx.x] = *lcount;
}

size_t prefilter_lds_bytes() { return K1B_LDS_TOTAL; }

uint32_t prefilter_grid(const uint8_t *d_hay, uint64_t len, int n_cus) {
    uint64_t lead = (uintptr_t)d_hay & 15;
    uint64_t total = lead + len;
    uint64_t ntiles = (total + (uint64_t)K1B_ROWS * 1024 - 1) / ((uint64_t)K1B_ROWS * 1024);
    uint64_t blocks = (ntiles + 15) / 16;
    if (blocks > (uint64_t)n_cus) blocks = n_cus;
    return blocks ? (uint32_t)blocks : 1;
}

hipError_t launch_prefilter(const DevAutomaton &A, const Segments &G, const Sink &K,
                            const uint8_t *d_hay, uint64_t len, uint32_t grid, hipStream_t st) {
    if (len == 0 || A.filter_q == 0) return hipSuccess;
    uint64_t lead = (uintptr_t)d_hay & 15;
    const uint8_t *base = d_hay - lead;
    uint64_t blocks = grid;
    uint32_t q = A.filter_q;
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[6] = {(const void *)k1b_prefilter<1>, (const void *)k1b_prefilter<2>,
                              (const void *)k1b_prefilter<3>, (const void *)k1b_prefilter<4>,
                              (const void *)k1b_prefilter<5>, (const void *)k1b_prefilter<6>};
        for (auto f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)K1B_LDS_TOTAL);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    dim3 g((uint32_t)blocks), b(1024);
    switch (q) {
    case 1: hipLaunchKernelGGL(k1b_prefilter<1>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    case 2: hipLaunchKernelGGL(k1b_prefilter<2>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    case 3: hipLaunchKernelGGL(k1b_prefilter<3>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    case 4: hipLaunchKernelGGL(k1b_prefilter<4>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    case 5: hipLaunchKernelGGL(k1b_prefilter<5>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    default: hipLaunchKernelGGL(k1b_prefilter<6>, g, b, K1B_LDS_TOTAL, st, A, G, K, base, len, lead); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sink bookkeeping: totals + compaction of the per-workgroup regions
// ---------------------------------------------------------------------------
// summary[0] = total occurrences, summary[1] = max per region, offsets[b] =
// exclusive prefix of min(count, region_cap)
__global__ __launch_bounds__(256) void k_sink_summary(const uint64_t *block_counts, uint32_t grid,
                                                      uint64_t region_cap, uint64_t *summary,
                                                      uint64_t *offsets) {
    __shared__ uint64_t part[256];
    // grid <= a few hundred: one thread per block, serial prefix by thread 0
    uint64_t mx = 0;
    for (uint32_t b = threadIdx.x; b < grid; b += 256) mx = block_counts[b] > mx ? block_counts[b] : mx;
    part[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t m = 0, run = 0;
        for (int i = 0; i < 256; i++) m = part[i] > m ? part[i] : m;
        for (uint32_t b = 0; b < grid; b++) {
            offsets[b] = run;
            uint64_t c = block_counts[b];
            run += c < region_cap ? c : region_cap;
        }
        offsets[grid] = run;
        summary[0] = run;
        summary[1] = m;
    }
}

__global__ __launch_bounds__(256) void k_sink_compact(const uint64_t *keys, const uint32_t *pids,
                                                      const uint64_t *offsets, uint64_t region_cap,
                                                      uint64_t *keys_out, uint32_t *pids_out) {
    uint64_t o0 = offsets[blockIdx.x], n = offsets[blockIdx.x + 1] - o0;
    const uint64_t *k = keys + (uint64_t)blockIdx.x * region_cap;
    const uint32_t *p = pids + (uint64_t)blockIdx.x * region_cap;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        keys_out[o0 + i] = k[i];
        pids_out[o0 + i] = p[i];
    }
}

hipError_t sink_summary(const uint64_t *block_counts, uint32_t grid, uint64_t region_cap,
                        uint64_t *summary, uint64_t *offsets, hipStream_t st) {
    hipLaunchKernelGGL(k_sink_summary, dim3(1), dim3(256), 0, st, block_counts, grid, region_cap,
                       summary, offsets);
    return hipGetLastError();
}

hipError_t sink_compact(const uint64_t *keys, const uint32_t *pids, const uint64_t *offsets,
                        uint32_t grid, uint64_t region_cap, uint64_t *keys_out, uint32_t *pids_out,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_sink_compact, dim3(grid), dim3(256), 0, st, keys, pids, offsets, region_cap,
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
    uint64_t x = keys[i] >> 24;
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

// code-point index of byte offset x = number of non-continuation bytes in [0, x)
__device__ __forceinline__ uint64_t code_point_of(const uint8_t *__restrict__ hay,
                                                  const uint64_t *blockpre, uint64_t x) {
    uint64_t blk = x >> 10;
    uint64_t c = blockpre[blk];
    for (uint64_t k = blk << 10; k < x; k++) c += (hay[k] & 0xC0) != 0x80;
    return c;
}

__global__ void k_to_code_points(const uint8_t *__restrict__ hay, const uint64_t *blockpre,
                                 acx_match_t *m, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n) return;
    uint64_t *field = (i & 1) ? &m[i >> 1].end : &m[i >> 1].start;
    *field = code_point_of(hay, blockpre, *field);
}

hipError_t to_code_points(const uint8_t *d_hay, uint64_t len, const uint64_t *blockpre,
                          acx_match_t *m, uint64_t n, hipStream_t st) {
    (void)len;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_to_code_points, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, st,
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
        uint64_t cb = code_point_of(hay, blockpre, base);
        m[i].start = code_point_of(hay, blockpre, s) - cb;
        m[i].end = code_point_of(hay, blockpre, e) - cb;
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
