// acx_api.cpp -- implementation of the C ABI declared in include/acx.h.
// Host orchestration of the device pipeline (kernels.hip).  There is no CPU
// matching path here: without a HIP device every find call fails (ACX_EDEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/acx.h"
#include "automaton.hpp"
#include "kernels.hpp"

using namespace acx;

namespace {

thread_local std::string g_err;
thread_local int g_device = -1; // -1: use the current HIP device

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hipfail(hipError_t e, const char *what) {
    return fail(ACX_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(expr)                                   \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) return hipfail(e__, #expr); \
    } while (0)

// Small process-wide cache of device buffers for results, so that a find call
// does not pay hipMalloc/hipFree (each tens of microseconds and a device sync).
struct BufCache {
    struct Ent { void *p; size_t bytes; int dev; };
    std::mutex mu;
    std::vector<Ent> free_list;
    size_t cached = 0;
    static constexpr size_t MAX_CACHED = (size_t)2 << 30;
    hipError_t get(void **out, size_t bytes, int dev) {
        bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)free_list.size(); i++)
                if (free_list[i].dev == dev && free_list[i].bytes >= bytes &&
                    free_list[i].bytes <= bytes * 4 + 65536 &&
                    (best < 0 || free_list[i].bytes < free_list[best].bytes))
                    best = i;
            if (best >= 0) {
                *out = free_list[best].p;
                cached -= free_list[best].bytes;
                free_list.erase(free_list.begin() + best);
                return hipSuccess;
            }
        }
        // round up so that slightly larger requests can reuse the buffer later
        size_t alloc = bytes + bytes / 4;
        alloc = (alloc + 4095) / 4096 * 4096;
        hipError_t e = hipMalloc(out, alloc);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            sizes.push_back({*out, alloc, dev});
        }
        return e;
    }
    void put(void *p, int dev) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        size_t bytes = 0;
        for (auto &e : sizes) if (e.p == p) { bytes = e.bytes; break; }
        if (!bytes || cached + bytes > MAX_CACHED || free_list.size() >= 16) {
            for (size_t i = 0; i < sizes.size(); i++) if (sizes[i].p == p) { sizes.erase(sizes.begin() + i); break; }
            (void)hipSetDevice(dev);
            (void)hipFree(p);
            return;
        }
        free_list.push_back({p, bytes, dev});
        cached += bytes;
    }
    std::vector<Ent> sizes; // every live buffer handed out by get()
};
BufCache g_bufs;

struct Workspace {
    uint64_t cap = 0; // occurrence capacity of the dense path (region mode + radix sort)
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *pids[2] = {nullptr, nullptr};
    uint64_t *S = nullptr, *E = nullptr, *M = nullptr;
    uint32_t *flags = nullptr, *idx = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    uint4 *recs = nullptr;            // occurrence sink: cap records of 16 B in per-workgroup regions
    uint4 *hrecs = nullptr;           // K1b prefix-hit sink: hit_total records of 32 B
    uint64_t hit_total = 0;
    uint64_t *hit_counts = nullptr;   // device: one per K1b workgroup
    uint64_t *summary = nullptr;      // device: [0] occurrences kept, [1] max per region, [2..3] same for
                                      // hits, [4] matches written, [5] abort flag of the sparse path
    uint64_t *block_counts = nullptr; // device: one per scan workgroup
    uint64_t *region_off = nullptr;   // device: exclusive prefix of the kept counts
    uint64_t *h_pinned = nullptr;     // pinned host scratch (16 x u64; [8], [9] = result of K0)
    uint8_t *pin_hay = nullptr;       // small calls: pinned copy of a host haystack (read by K0 in place)
    acx_match_t *pin_out = nullptr;   // small calls: pinned output of K0 (host entry point)
    uint64_t *blockcnt = nullptr, *blockpre = nullptr;
    uint64_t block_cap = 0;
    TileSpace T{};                    // sparse path (slot mode + tile kernels)
    uint64_t tile_buckets = 0;        // buckets T is allocated for
    bool sparse_dirty = true;         // T.bcnt / the abort flag are not known to be zero
    acx_match_t *final = nullptr;     // sparse path: output buffer the next call writes into
    uint64_t final_cap = 0;
    uint8_t *hay = nullptr; // staging buffer of the host-memory entry points
    uint64_t hay_cap = 0;
    uint64_t *offsets = nullptr;
    uint64_t offsets_cap = 0;
};

} // namespace

struct acx_automaton {
    Automaton host;
    int device = 0;
    hipStream_t stream = nullptr;
    DevAutomaton dev{};
    const DevAutomaton *d_dev = nullptr; // the same struct, resident in HBM
    std::vector<void *> allocs;
    int kernel = ACX_KERNEL_DFA_WALK;
    int n_cus = 1;
    size_t max_lds = 65536;
    uint64_t table_bytes = 0;
    std::mutex mu;       // guards the workspace + device pipeline
    std::mutex stage_mu; // guards the host staging buffers (taken before mu)
    Workspace ws;
    bool prof = false;
    bool post_pending = false;  // profiling: ev[2] of the last call has not been read yet
    bool kernel_forced = false; // the scan kernel was chosen explicitly: K0 never takes a call
    int dense_hold = 0; // > 0: the output was too dense for the sparse path; calls left in region mode
    acx_profile_t profile{};
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // pipelined sparse path: the scan of chunk c + 1 (stream) overlaps the verification and
    // tile kernels of chunk c (post_stream); chunk_ev[c] = scan of chunk c done
    static constexpr int MAX_CHUNKS = 8;
    hipStream_t post_stream = nullptr;
    hipEvent_t chunk_ev[MAX_CHUNKS] = {};
    hipEvent_t post_ev = nullptr; // the post stream caught up (start of a call)
    int flag_idx = 0;             // which of the two abort flags the next sparse attempt uses
};

struct acx_host_automaton {
    Automaton host;
};

struct acx_result {
    int device = 0;
    acx_match_t *d_matches = nullptr;
    uint64_t n = 0;
    uint64_t *d_counts = nullptr;
    uint64_t n_hay = 0;
};

namespace {

template <typename T>
int upload(acx_automaton *a, const T *src, size_t count, const T **dst) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    bytes = (bytes + 15) / 16 * 16;
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, bytes));
    a->allocs.push_back(d);
    HIPCHK(hipMemsetAsync(d, 0, bytes, a->stream));
    if (count) HIPCHK(hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, a->stream));
    *dst = (const T *)d;
    return ACX_OK;
}

void free_tiles(Workspace &w) {
    TileSpace &T = w.T;
    (void)hipFree(T.slots); (void)hipFree(T.bcnt); (void)hipFree(T.trecs);
    (void)hipFree(T.syncf); (void)hipFree(T.accf); (void)hipFree(T.tile_n); (void)hipFree(T.btot);
    (void)hipFree(T.bbase);
    T = TileSpace{};
    w.tile_buckets = 0;
}

void free_ws(Workspace &w, int device) {
    for (int i = 0; i < 2; i++) { (void)hipFree(w.keys[i]); (void)hipFree(w.pids[i]); }
    (void)hipFree(w.S); (void)hipFree(w.E); (void)hipFree(w.M);
    (void)hipFree(w.flags); (void)hipFree(w.idx); (void)hipFree(w.temp);
    (void)hipFree(w.summary); (void)hipFree(w.block_counts); (void)hipFree(w.region_off);
    (void)hipFree(w.recs); (void)hipFree(w.hrecs); (void)hipFree(w.hit_counts);
    free_tiles(w);
    g_bufs.put(w.final, device);

    (void)hipFree(w.blockcnt); (void)hipFree(w.blockpre);
    (void)hipFree(w.hay); (void)hipFree(w.offsets);
    if (w.h_pinned) (void)hipHostFree(w.h_pinned);
    if (w.pin_hay) (void)hipHostFree(w.pin_hay);
    if (w.pin_out) (void)hipHostFree(w.pin_out);
    w = Workspace();
}

int ensure_common(acx_automaton *a) {
    Workspace &w = a->ws;
    if (!w.summary) {
        HIPCHK(hipMalloc((void **)&w.summary, 64));
        HIPCHK(hipMalloc((void **)&w.block_counts, 8 * 8192));
        HIPCHK(hipMalloc((void **)&w.region_off, 8 * 8193));
        HIPCHK(hipMalloc((void **)&w.hit_counts, 8 * 16 * 1024 * acx_automaton::MAX_CHUNKS));
        HIPCHK(hipHostMalloc((void **)&w.h_pinned, 128, hipHostMallocDefault));
        w.sparse_dirty = true;
    }
    return ACX_OK;
}

// prefix-hit sink of K1b
int ensure_hits(acx_automaton *a, uint64_t want) {
    Workspace &w = a->ws;
    if (want <= w.hit_total) return ACX_OK;
    (void)hipFree(w.hrecs); w.hrecs = nullptr; w.hit_total = 0;
    HIPCHK(hipMalloc((void **)&w.hrecs, want * 32));
    w.hit_total = want;
    return ACX_OK;
}

// dense path: occurrence regions + everything the radix sort / resolve pipeline needs
int ensure_occ_capacity(acx_automaton *a, uint64_t want) {
    Workspace &w = a->ws;
    if (want <= w.cap) return ACX_OK;
    uint64_t cap = std::max<uint64_t>(want, 1u << 16);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(w.keys[i]); (void)hipFree(w.pids[i]);
        w.keys[i] = nullptr; w.pids[i] = nullptr;
    }
    (void)hipFree(w.S); (void)hipFree(w.E); (void)hipFree(w.M);
    (void)hipFree(w.flags); (void)hipFree(w.idx); (void)hipFree(w.temp);
    (void)hipFree(w.recs);
    w.S = w.E = w.M = nullptr; w.flags = w.idx = nullptr; w.temp = nullptr; w.cap = 0;
    w.temp_bytes = 0;
    w.recs = nullptr;
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc((void **)&w.keys[i], cap * 8));
        HIPCHK(hipMalloc((void **)&w.pids[i], cap * 4));
    }
    HIPCHK(hipMalloc((void **)&w.recs, cap * 16));
    HIPCHK(hipMalloc((void **)&w.S, cap * 8));
    HIPCHK(hipMalloc((void **)&w.E, cap * 8));
    HIPCHK(hipMalloc((void **)&w.M, cap * 8));
    HIPCHK(hipMalloc((void **)&w.flags, (cap + 1) * 4));
    HIPCHK(hipMalloc((void **)&w.idx, (cap + 1) * 4));
    w.temp_bytes = std::max(sort_temp_bytes(cap), scan_temp_bytes(cap)) + 256;
    HIPCHK(hipMalloc(&w.temp, w.temp_bytes));
    w.cap = cap;
    return ACX_OK;
}

// sparse path: slots and tile arrays for nb buckets
int ensure_tiles(acx_automaton *a, uint64_t nb) {
    Workspace &w = a->ws;
    TileSpace &T = w.T;
    const uint64_t tiles = (nb + TILE_BUCKETS - 1) / TILE_BUCKETS;
    if (nb > w.tile_buckets) {
        free_tiles(w);
        if (w.final) { g_bufs.put(w.final, a->device); w.final = nullptr; }
        const uint64_t ents = tiles * TILE_MAX;
        HIPCHK(hipMalloc((void **)&T.slots, nb * BUCKET_SLOTS * 16));
        HIPCHK(hipMalloc((void **)&T.bcnt, (nb + 1) * 4));
        HIPCHK(hipMalloc((void **)&T.trecs, ents * 16));
        HIPCHK(hipMalloc((void **)&T.syncf, ents));
        HIPCHK(hipMalloc((void **)&T.accf, ents));
        HIPCHK(hipMalloc((void **)&T.tile_n, tiles * 4));
        HIPCHK(hipMalloc((void **)&T.btot, tiles * 4));
        HIPCHK(hipMalloc((void **)&T.bbase, tiles * 4));
        w.tile_buckets = nb;
        w.sparse_dirty = true;
    }
    T.n_buckets = (uint32_t)nb;
    T.n_tiles = (uint32_t)tiles;
    return ACX_OK;
}

int ensure_blocks(acx_automaton *a, uint64_t nblocks_plus1) {
    Workspace &w = a->ws;
    if (nblocks_plus1 <= w.block_cap) {
        // the scan temp storage may need to cover this size too
    } else {
        (void)hipFree(w.blockcnt); (void)hipFree(w.blockpre);
        w.blockcnt = w.blockpre = nullptr; w.block_cap = 0;
        HIPCHK(hipMalloc((void **)&w.blockcnt, nblocks_plus1 * 8));
        HIPCHK(hipMalloc((void **)&w.blockpre, nblocks_plus1 * 8));
        w.block_cap = nblocks_plus1;
    }
    size_t need = scan_temp_bytes(nblocks_plus1) + 256;
    if (need > w.temp_bytes) {
        (void)hipFree(w.temp); w.temp = nullptr;
        HIPCHK(hipMalloc(&w.temp, need));
        w.temp_bytes = need;
    }
    return ACX_OK;
}

int bits_for(uint64_t x) { // number of bits needed to represent x
    int b = 0;
    while (x) { b++; x >>= 1; }
    return b;
}

// K0 takes the call when the haystack is small and nobody asked for a particular scan kernel
bool small_ok(const acx_automaton *a, uint64_t len) {
    static const bool off = std::getenv("ACX_NO_SMALL") != nullptr;
    return !off && !a->kernel_forced && len > 0 && len <= SMALL_MAX_LEN && a->host.n_patterns > 0;
}

// One K0 launch + one sync.  hay / out: anything the device can address (HBM or pinned host);
// out holds SMALL_MAX_OCC records.  *done = false: too many occurrences, use the general path.
// Caller holds a->mu.
int run_small(acx_automaton *a, const uint8_t *hay, uint64_t len, int overlapping, int codepoints,
              acx_match_t *out, uint64_t *n_out, bool *done) {
    *done = false;
    int rc = ensure_common(a);
    if (rc) return rc;
    Workspace &w = a->ws;
    const int key_mode = overlapping ? 0 : a->host.match_kind;
    HIPCHK(launch_small(a->dev, hay, (uint32_t)len, key_mode, overlapping != 0, codepoints != 0, out,
                        w.h_pinned + 8, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    if (w.h_pinned[9] == 0) { *n_out = w.h_pinned[8]; *done = true; a->profile.small_calls++; }
    return ACX_OK;
}

// post_ms of the previous profiled call: ev[1] (end of the scan) .. ev[2] (end of the call's device work)
void settle_post_profile(acx_automaton *a) {
    if (!a->post_pending) return;
    a->post_pending = false;
    float ms = 0;
    if (hipEventSynchronize(a->ev[2]) == hipSuccess && hipEventElapsedTime(&ms, a->ev[1], a->ev[2]) == hipSuccess)
        a->profile.post_ms += ms;
}

void add_scan_profile(acx_automaton *a, uint64_t len) {
    if (!a->prof) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, a->ev[0], a->ev[1]) != hipSuccess) return;
    a->profile.scan_ms += ms;
    a->profile.scan_launches++;
    a->profile.scan_bytes += len;
}

// ---------------------------------------------------------------------------
// The device pipeline.  d_hay: device pointer, len bytes.
//
//   small haystack:          K0, the whole call in one workgroup                 one launch
//   sparse output (default): scan (K1a, or K1b + walk) emits into bucket slots -> tile kernels
//                            (sort, resolve, scan, write) -> final matches       ONE host round trip
//   dense output:            scan emits into regions -> compact -> radix sort -> spans ->
//                            resolve -> offsets -> write                         (two round trips)
// ---------------------------------------------------------------------------
#define HIPCHK_RC(expr)                                  \
    do {                                                 \
        hipError_t e__ = (expr);                         \
        if (e__ != hipSuccess) return hipfail(e__, #expr); \
    } while (0)

// what one call works on (all attempts of it)
struct FindCall {
    acx_automaton *a;
    const uint8_t *d_hay;
    uint64_t len;
    const Segments &G;
    bool overlapping, codepoints, segmented;
    acx_result *r;
    int key_mode;
    bool pre;           // K1b + walk (else K1a)
    uint32_t scan_grid; // workgroups of the scan kernel
    uint32_t hit_grid;  // hit regions of one K1b launch
    uint32_t grid;      // emitting workgroups (= occurrence regions in region mode)
    uint32_t bshift;    // bucket = key >> bshift
    uint64_t nb;        // buckets of 4 KiB of stream position
    // results
    uint64_t n_raw = 0, n_final = 0;
    bool pending = false;   // work queued on the stream that nobody waited for yet
    bool localized = false; // batch: offsets are already local and the counts taken
};
enum class Attempt { Done, Again, GoDense };

// ---- sparse output: slot mode + tile kernels, ONE host round trip
int attempt_sparse(FindCall &c, Attempt *what) {
    acx_automaton *a = c.a;
    Workspace &w = a->ws;
    hipStream_t st = a->stream;
    int rc = ensure_tiles(a, c.nb);
    if (rc) return rc;
    // Chunked variant (ACX_CHUNKS=n, experiments only): K1b in n launches, each chunk's walk +
    // tile kernels on the post stream underneath the next chunk's scan.  Measured on MI355X: no
    // gain -- K1b's 16 waves x 128 VGPRs per CU fill the register file, so the other kernels only
    // get on the CUs when a K1b workgroup retires.
    static const int chunk_env = std::getenv("ACX_CHUNKS") ? std::atoi(std::getenv("ACX_CHUNKS")) : 1;
    const int chunks = c.pre ? std::max(1, std::min(chunk_env, (int)acx_automaton::MAX_CHUNKS)) : 1;
    const uint64_t hit_regions = (uint64_t)c.hit_grid * chunks; // every chunk has its own hit regions
    const uint64_t hit_cap = c.pre ? w.hit_total / hit_regions : 0;
    const Sink H{w.hrecs, nullptr, w.hit_counts, hit_cap, 0, c.key_mode, nullptr, nullptr};
    const TileSpace &T = w.T;
    const uint64_t out_cap = (uint64_t)T.n_tiles * TILE_MAX;
    if (w.final && w.final_cap < out_cap) { g_bufs.put(w.final, a->device); w.final = nullptr; }
    if (!w.final) {
        HIPCHK_RC(g_bufs.get((void **)&w.final, out_cap * sizeof(acx_match_t), a->device));
        w.final_cap = out_cap;
    }
    if (w.sparse_dirty) {
        HIPCHK_RC(hipMemsetAsync(T.bcnt, 0, (c.nb + 1) * 4, st));
        HIPCHK_RC(hipMemsetAsync(w.summary + 5, 0, 16, st));
    }
    w.sparse_dirty = true;
    // two abort flags used in turn: this attempt's tile kernels clear the other one
    uint32_t *abort_flag = (uint32_t *)(w.summary + 5 + a->flag_idx);
    uint32_t *next_flag = (uint32_t *)(w.summary + 5 + (a->flag_idx ^ 1));
    a->flag_idx ^= 1;
    const Sink K{nullptr, T.bcnt, w.block_counts, 0, c.bshift, c.key_mode, T.slots, abort_flag};
    // batch with byte offsets: the write kernel localises and counts per haystack itself
    uint64_t *seg_counts = c.segmented && !c.codepoints ? c.r->d_counts : nullptr;
    if (!c.pre) {
        if (a->prof) HIPCHK_RC(hipEventRecord(a->ev[0], st));
        HIPCHK_RC(launch_dfa_walk(a->dev, a->d_dev, c.G, K, c.d_hay, c.len, c.scan_grid, a->max_lds, st));
        if (a->prof) HIPCHK_RC(hipEventRecord(a->ev[1], st));
        HIPCHK_RC(tile_post(a->dev, c.key_mode, c.overlapping, T, 0, T.n_tiles, true, true, nullptr, 0, 0,
                            w.final, w.summary, abort_flag, next_flag, w.h_pinned, c.G, seg_counts, st));
        HIPCHK_RC(hipStreamSynchronize(st));
    } else {
        // K1b in chunks on `st`; each chunk's walk + tile kernels on the post stream as soon as its
        // scan is done.  After the scan of K1b tiles [0, t1) every bucket below t1 - 1 is final (a key
        // position never precedes the start of its occurrence), so the post stage trails by a tile.
        const uint64_t k_tiles = prefilter_tiles(c.d_hay, c.len);
        const uint64_t per = ((k_tiles + chunks - 1) / chunks + TILE_BUCKETS - 1) / TILE_BUCKETS * TILE_BUCKETS;
        hipStream_t ps = chunks > 1 ? a->post_stream : st;
        if (ps != st) {
            HIPCHK_RC(hipEventRecord(a->post_ev, st)); // memsets above / earlier work on st
            HIPCHK_RC(hipStreamWaitEvent(ps, a->post_ev, 0));
        }
        uint32_t tile0 = 0;
        bool first = true;
        for (int k = 0; k < chunks; k++) {
            const uint64_t t0 = std::min<uint64_t>((uint64_t)k * per, k_tiles);
            const uint64_t t1 = k == chunks - 1 ? k_tiles : std::min<uint64_t>(t0 + per, k_tiles);
            const bool last = k == chunks - 1;
            if (t1 == t0 && !last) continue;
            Sink Hc = H;
            Hc.recs = w.hrecs + (uint64_t)k * c.hit_grid * hit_cap * 2;
            Hc.block_counts = w.hit_counts + (uint64_t)k * c.hit_grid;
            // measurement: the event pair rides on the dispatch (first chunk's start, last chunk's stop)
            HIPCHK_RC(launch_prefilter(a->dev, a->d_dev, c.G, Hc, c.d_hay, c.len, c.scan_grid, t0, t1, st,
                                       a->prof && k == 0 ? a->ev[0] : nullptr,
                                       a->prof && last ? a->ev[1] : nullptr));
            if (ps != st) {
                HIPCHK_RC(hipEventRecord(a->chunk_ev[k], st));
                HIPCHK_RC(hipStreamWaitEvent(ps, a->chunk_ev[k], 0));
            }
            HIPCHK_RC(launch_walk_hits(a->dev, a->d_dev, c.G, Hc, c.hit_grid, 0, K, c.d_hay, c.len, ps));
            const uint32_t tile1 = last ? T.n_tiles
                                        : (uint32_t)std::min<uint64_t>((t1 - 1) / TILE_BUCKETS, T.n_tiles);
            HIPCHK_RC(tile_post(a->dev, c.key_mode, c.overlapping, T, tile0, std::max(tile0, tile1), first, last,
                                Hc.block_counts, c.hit_grid, hit_cap, w.final, w.summary, abort_flag, next_flag,
                                w.h_pinned, c.G, seg_counts, ps));
            tile0 = std::max(tile0, tile1);
            first = false;
        }
        HIPCHK_RC(hipStreamSynchronize(ps));
        if (ps != st) HIPCHK_RC(hipStreamSynchronize(st));
    }
    w.sparse_dirty = false; // the tile kernels left the counters and the next flag clean
    add_scan_profile(a, c.len);
    const bool aborted = w.h_pinned[5] != 0;
    const uint64_t hit_max = c.pre ? w.h_pinned[3] : 0;
    if (aborted) { // the sparse path gave up
        if (seg_counts) // (a chunked call may already have counted the matches of its first chunks)
            HIPCHK_RC(hipMemsetAsync(c.r->d_counts, 0, std::max<uint64_t>(c.G.n_hay, 1) * 8, st));
        if (hit_max > hit_cap) { // prefix hits were dropped: grow their sink, redo
            if ((rc = ensure_hits(a, hit_regions * (hit_max + hit_max / 8 + 64))) != ACX_OK) return rc;
            *what = Attempt::Again;
        } else { // a bucket or a tile overflowed: dense output, use the region mode
            a->dense_hold = 8;
            *what = Attempt::GoDense;
        }
        return ACX_OK;
    }
    c.n_raw = w.h_pinned[0];
    c.n_final = w.h_pinned[4];
    c.r->d_matches = w.final; // hand the buffer over; the next call takes a fresh one
    w.final = nullptr;
    c.localized = seg_counts != nullptr;
    if (c.localized) c.pending = false; // the stream has drained, counts included
    *what = Attempt::Done;
    return ACX_OK;
}

// ---- dense output: region mode -> compact -> radix sort -> resolve (two round trips)
int attempt_dense(FindCall &c, Attempt *what) {
    acx_automaton *a = c.a;
    Workspace &w = a->ws;
    hipStream_t st = a->stream;
    int rc = ensure_occ_capacity(a, std::max<uint64_t>(1u << 16, c.len / 64));
    if (rc) return rc;
    const uint64_t hit_cap = c.pre ? w.hit_total / c.hit_grid : 0;
    const uint64_t region_cap = w.cap / c.grid;
    const Sink H{w.hrecs, nullptr, w.hit_counts, hit_cap, 0, c.key_mode, nullptr, nullptr};
    const Sink K{w.recs, nullptr, w.block_counts, region_cap, c.bshift, c.key_mode, nullptr, nullptr};
    if (a->prof) HIPCHK_RC(hipEventRecord(a->ev[0], st));
    HIPCHK_RC(c.pre ? launch_prefilter(a->dev, a->d_dev, c.G, H, c.d_hay, c.len, c.scan_grid, 0, ~0ull, st)
                    : launch_dfa_walk(a->dev, a->d_dev, c.G, K, c.d_hay, c.len, c.scan_grid, a->max_lds, st));
    if (a->prof) HIPCHK_RC(hipEventRecord(a->ev[1], st));
    if (c.pre) HIPCHK_RC(launch_walk_hits(a->dev, a->d_dev, c.G, H, c.hit_grid, 0, K, c.d_hay, c.len, st));
    HIPCHK_RC(sink_summary(w.block_counts, c.grid, region_cap, c.pre ? w.hit_counts : nullptr, c.hit_grid, hit_cap,
                           w.summary, w.region_off, st));
    HIPCHK_RC(hipMemcpyAsync(w.h_pinned, w.summary, 32, hipMemcpyDeviceToHost, st));
    HIPCHK_RC(hipStreamSynchronize(st));
    add_scan_profile(a, c.len);
    const uint64_t n_raw = w.h_pinned[0], region_max = w.h_pinned[1], hit_max = c.pre ? w.h_pinned[3] : 0;
    if (region_max > region_cap || hit_max > hit_cap) { // a sink region overflowed: grow, redo
        if (hit_max > hit_cap && (rc = ensure_hits(a, (uint64_t)c.hit_grid * (hit_max + hit_max / 8 + 64))) != ACX_OK)
            return rc;
        // hits that overflowed were dropped, so the occurrence count is a lower bound
        uint64_t want = (uint64_t)c.grid * (region_max + region_max / 8 + 64);
        if (hit_max > hit_cap) want = std::max(want, w.cap * 4);
        if ((rc = ensure_occ_capacity(a, want)) != ACX_OK) return rc;
        *what = Attempt::Again;
        return ACX_OK;
    }
    if (n_raw >= (1ull << 32) - 2) return fail(ACX_ETOOBIG, "more than 2^32 occurrences");
    if (n_raw > 8 * c.nb) a->dense_hold = 8;
    else if (a->dense_hold > 0) a->dense_hold--;
    c.n_raw = n_raw;
    *what = Attempt::Done;
    if (n_raw == 0) return ACX_OK;
    if (a->prof) HIPCHK_RC(hipEventRecord(a->ev[1], st));
    HIPCHK_RC(sink_compact(w.recs, w.region_off, c.grid, region_cap, w.keys[1], w.pids[1], st));
    const int end_bit = std::min(64, (int)a->dev.rank_bits + bits_for(c.len));
    HIPCHK_RC(sort_occurrences(w.temp, w.temp_bytes, w.keys[1], w.keys[0], w.pids[1], w.pids[0], n_raw, end_bit, st));
    HIPCHK_RC(make_spans(a->dev, c.key_mode, w.keys[0], w.pids[0], w.S, w.E, n_raw, st));
    if (c.overlapping) {
        c.n_final = n_raw;
    } else {
        // Standard: sorted by end, so the running max of the ends IS the array of ends
        const uint64_t *M = w.E;
        if (c.key_mode != 0) {
            HIPCHK_RC(prefix_max(w.temp, w.temp_bytes, w.E, w.M, n_raw, st));
            M = w.M;
        }
        HIPCHK_RC(hipMemsetAsync(w.flags + n_raw, 0, 4, st));
        HIPCHK_RC(resolve_greedy(w.S, w.E, M, w.flags, n_raw, st));
        HIPCHK_RC(flag_offsets(w.temp, w.temp_bytes, w.flags, w.idx, n_raw, st));
        HIPCHK_RC(hipMemcpyAsync(w.h_pinned + 6, w.idx + n_raw, 4, hipMemcpyDeviceToHost, st));
        HIPCHK_RC(hipStreamSynchronize(st));
        c.n_final = *(uint32_t *)(w.h_pinned + 6);
    }
    HIPCHK_RC(g_bufs.get((void **)&c.r->d_matches, std::max<uint64_t>(c.n_final, 1) * sizeof(acx_match_t),
                         a->device));
    HIPCHK_RC(write_matches(w.pids[0], w.S, w.E, c.overlapping ? nullptr : w.flags, c.overlapping ? nullptr : w.idx,
                            c.r->d_matches, n_raw, st));
    c.pending = true;
    return ACX_OK;
}

// everything after the matches exist: code points (str API), local offsets + counts (batches)
int finish_matches(FindCall &c) {
    acx_automaton *a = c.a;
    Workspace &w = a->ws;
    hipStream_t st = a->stream;
    if (!c.n_final || !(c.codepoints || (c.segmented && !c.localized))) return ACX_OK;
    if (c.codepoints) {
        const uint64_t nb1 = (c.len + 1023) / 1024 + 1;
        int rc = ensure_blocks(a, nb1);
        if (rc) return rc;
        HIPCHK_RC(count_lead_bytes(c.d_hay, c.len, w.blockcnt, st));
        HIPCHK_RC(prefix_sum_u64(w.temp, w.temp_bytes, w.blockcnt, w.blockpre, nb1, st));
    }
    if (c.segmented)
        HIPCHK_RC(localize(c.G, c.d_hay, c.len, w.blockpre, c.codepoints, c.r->d_matches, c.n_final, c.r->d_counts, st));
    else
        HIPCHK_RC(to_code_points(c.d_hay, c.len, w.blockpre, c.r->d_matches, c.n_final, st));
    c.pending = true;
    return ACX_OK;
}

// the general pipeline on an allocated result; caller holds a->mu
int run_pipeline(FindCall &c) {
    acx_automaton *a = c.a;
    int rc = ensure_common(a);
    if (rc) return rc;
    c.pre = a->kernel == ACX_KERNEL_PREFILTER;
    // K1b emits prefix hits; k_walk_hits turns them into occurrences.  K1a emits occurrences.
    c.scan_grid = c.pre ? prefilter_grid(c.d_hay, c.len, a->n_cus) : dfa_walk_grid(a->dev, c.len, a->n_cus);
    c.hit_grid = c.pre ? prefilter_hit_regions(c.scan_grid) : 0;
    c.grid = c.pre ? walk_hits_grid(c.hit_grid) : c.scan_grid;
    c.bshift = a->dev.rank_bits + BUCKET_BITS;
    c.nb = (c.len >> BUCKET_BITS) + 2;
    static const bool no_bucket_env = std::getenv("ACX_NO_BUCKET") != nullptr; // profiling only
    bool sparse = a->dense_hold == 0 && !no_bucket_env && c.nb < (1ull << 31);
    if (c.pre && (rc = ensure_hits(a, std::max<uint64_t>(1u << 16, c.len / 64))) != ACX_OK) return rc;
    for (int attempt = 0;; attempt++) {
        if (attempt == 5) return fail(ACX_EDEVICE, "occurrence buffer overflow persisted");
        Attempt what = Attempt::Done;
        if ((rc = sparse ? attempt_sparse(c, &what) : attempt_dense(c, &what)) != ACX_OK) return rc;
        if (what == Attempt::GoDense) sparse = false;
        if (what == Attempt::Done) break;
    }
    if (a->prof) {
        a->profile.raw_occurrences += c.n_raw;
        a->profile.prefix_hits += c.pre ? a->ws.h_pinned[2] : 0;
    }
    c.r->n = c.n_final;
    if ((rc = finish_matches(c)) != ACX_OK) return rc;
    if (a->prof) { // end of the post stage: read lazily (next call / acx_profile_read), no extra sync here
        HIPCHK_RC(hipEventRecord(a->ev[2], a->stream));
        a->post_pending = true;
    }
    return ACX_OK;
}

int run_find(acx_automaton *a, const uint8_t *d_hay, uint64_t len, const Segments &G,
             int overlapping, int codepoints, acx_result **out, bool allow_small = true) {
    *out = nullptr;
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD) {
        static const char *names[3] = {"Standard", "LeftmostFirst", "LeftmostLongest"};
        return fail(ACX_EOVERLAP, std::string("match kind ") + names[a->host.match_kind] +
                                      " does not support overlapping searches");
    }
    if (len >= (1ull << 40)) return fail(ACX_ETOOBIG, "haystack stream of 2^40 bytes or more");
    std::lock_guard<std::mutex> lock(a->mu);
    HIPCHK(hipSetDevice(a->device));
    settle_post_profile(a);
    hipStream_t st = a->stream;
    const bool segmented = G.uniform_len != 0 || G.offsets != nullptr;
    acx_result *r = new (std::nothrow) acx_result();
    if (!r) return fail(ACX_ENOMEM, "out of memory");
    r->device = a->device;
    r->n_hay = segmented ? G.n_hay : 0;
    FindCall c{a, d_hay, len, G, overlapping != 0, codepoints != 0, segmented, r,
               overlapping ? 0 : a->host.match_kind};
    auto body = [&]() -> int {
        if (segmented) {
            HIPCHK_RC(g_bufs.get((void **)&r->d_counts, std::max<uint64_t>(G.n_hay, 1) * 8, a->device));
            HIPCHK_RC(hipMemsetAsync(r->d_counts, 0, std::max<uint64_t>(G.n_hay, 1) * 8, st));
            c.pending = true;
        }
        if (allow_small && !segmented && small_ok(a, len)) { // small haystack: the whole call in one workgroup (K0)
            HIPCHK_RC(g_bufs.get((void **)&r->d_matches, SMALL_MAX_OCC * sizeof(acx_match_t), a->device));
            bool done = false;
            int rc = run_small(a, d_hay, len, overlapping, codepoints, r->d_matches, &r->n, &done);
            if (rc || done) return rc;
            g_bufs.put(r->d_matches, a->device); // dense: the general pipeline takes over
            r->d_matches = nullptr;
        }
        if (len > 0 && a->host.n_patterns > 0) {
            int rc = run_pipeline(c);
            if (rc) return rc;
        }
        if (c.pending) HIPCHK_RC(hipStreamSynchronize(st));
        return ACX_OK;
    };
    const int rc = body();
    if (rc != ACX_OK) { acx_free_result(r); return rc; }
    *out = r;
    return ACX_OK;
}
#undef HIPCHK_RC

int stage_host(acx_automaton *a, const uint8_t *hay, uint64_t len, const uint64_t *offsets,
               uint64_t n_off) {
    Workspace &w = a->ws;
    HIPCHK(hipSetDevice(a->device));
    if (len > w.hay_cap) {
        (void)hipFree(w.hay); w.hay = nullptr; w.hay_cap = 0;
        uint64_t cap = std::max<uint64_t>(len + len / 8, 4096);
        HIPCHK(hipMalloc((void **)&w.hay, cap));
        w.hay_cap = cap;
    }
    if (len) HIPCHK(hipMemcpyAsync(w.hay, hay, len, hipMemcpyHostToDevice, a->stream));
    if (n_off) {
        if (n_off > w.offsets_cap) {
            (void)hipFree(w.offsets); w.offsets = nullptr; w.offsets_cap = 0;
            HIPCHK(hipMalloc((void **)&w.offsets, n_off * 8));
            w.offsets_cap = n_off;
        }
        HIPCHK(hipMemcpyAsync(w.offsets, offsets, n_off * 8, hipMemcpyHostToDevice, a->stream));
    }
    return ACX_OK;
}

} // namespace

// ---------------------------------------------------------------------------
extern "C" {

int acx_version(void) { return ACX_VERSION; }
const char *acx_last_error(void) { return g_err.c_str(); }

int acx_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return hipfail(e, "hipGetDeviceCount"); }
    *n = c;
    return ACX_OK;
}

int acx_set_device(int ordinal) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) return hipfail(e, "hipGetDeviceCount");
    if (ordinal < 0 || ordinal >= c) return fail(ACX_EINVAL, "device ordinal out of range");
    g_device = ordinal;
    HIPCHK(hipSetDevice(ordinal));
    return ACX_OK;
}

int acx_build(const uint8_t *blob, const uint64_t *offsets, uint64_t n_patterns, int match_kind,
              int implementation, acx_automaton_t **out) {
    if (!out) return fail(ACX_EINVAL, "null output pointer");
    *out = nullptr;
    if (n_patterns && (!offsets || (!blob && offsets[n_patterns] != offsets[0])))
        return fail(ACX_EINVAL, "null pattern buffer");
    if (implementation < ACX_IMPL_AUTO || implementation > ACX_IMPL_DFA)
        return fail(ACX_EINVAL, "unknown implementation hint");
    acx_automaton *a = new (std::nothrow) acx_automaton();
    if (!a) return fail(ACX_ENOMEM, "out of memory");
    static const uint64_t zero_off[1] = {0};
    int code = ACX_OK;
    std::string err;
    try {
        err = compile(blob, n_patterns ? offsets : zero_off, n_patterns, match_kind, a->host, code);
    } catch (const std::bad_alloc &) {
        delete a;
        return fail(ACX_ENOMEM, "out of host memory while compiling the automaton");
    }
    if (code != ACX_OK) { delete a; return fail(code, err); }

    // ---- device side.  No device => no matcher (there is no CPU fallback).
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        delete a;
        return fail(ACX_EDEVICE, std::string("no HIP device available: ") +
                                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    }
    int dev = g_device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    a->device = dev;
    auto destroy = [&](int rc) { acx_free_automaton(a); return rc; };
#define HIPCHK_A(expr)                                            \
    do {                                                          \
        hipError_t e__ = (expr);                                  \
        if (e__ != hipSuccess) return destroy(hipfail(e__, #expr)); \
    } while (0)
    HIPCHK_A(hipSetDevice(dev));
    {
        int v = 0;
        HIPCHK_A(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        a->n_cus = std::max(v, 1);
        int l1 = 0, l2 = 0;
        (void)hipDeviceGetAttribute(&l1, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        (void)hipDeviceGetAttribute(&l2, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev);
        a->max_lds = (size_t)std::max(std::max(l1, l2), 65536);
        if (a->max_lds > 160 * 1024) a->max_lds = 160 * 1024;
    }
    HIPCHK_A(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    for (auto &ev : a->ev) HIPCHK_A(hipEventCreate(&ev));
    HIPCHK_A(hipStreamCreateWithFlags(&a->post_stream, hipStreamNonBlocking));
    for (auto &ev : a->chunk_ev) HIPCHK_A(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK_A(hipEventCreateWithFlags(&a->post_ev, hipEventDisableTiming));

    Automaton &H = a->host;
    DevAutomaton &D = a->dev;
    D.n_patterns = H.n_patterns; D.n_states = H.n_states; D.stride2 = H.stride2;
    D.min_len = H.min_len; D.max_len = H.max_len; D.filter_q = H.filter_q;
    D.ptab_log2 = H.ptab_log2; D.filter_q2 = H.filter_q2;
    D.rank_bits = (uint32_t)std::max(1, bits_for(H.n_patterns ? H.n_patterns - 1 : 0));
    // compact u16 copy of the hot (lowest-id) rows for K1a's LDS tile
    uint32_t hot_rows = dfa_walk_hot_rows(H.n_states, H.stride2, 160 * 1024);
    std::vector<uint16_t> hot16(((size_t)hot_rows << H.stride2) + 8, 0xFFFF);
    for (size_t i = 0; i < ((size_t)hot_rows << H.stride2); i++) {
        uint32_t en = H.table[i], id = en & ID_MASK;
        hot16[i] = id < 0x3FFFu ? (uint16_t)(id | ((en >> 30) << 14)) : (uint16_t)0xFFFF;
    }
    D.hot_rows = hot_rows;
    int rc;
#define UP(vec, field)                                                                   \
    if ((rc = upload(a, (vec).data(), (vec).size(), &D.field)) != ACX_OK) return destroy(rc);
    UP(H.table, table)
    UP(hot16, hot16)
    UP(H.own_off, own_off)
    UP(H.own_pid, own_pid)
    UP(H.own1, own1)
    UP(H.dlink, dlink)
    UP(H.level_start, level_start)
    UP(H.plen, plen)
    UP(H.rank, rank)
    UP(H.filterA, filterA)
    UP(H.ptab, ptab)
    UP(H.blist, blist)
    {
        const uint32_t *pi = nullptr;
        if ((rc = upload(a, H.pinfo.data(), H.pinfo.size(), &pi)) != ACX_OK) return destroy(rc);
        D.pinfo = reinterpret_cast<const uint4 *>(pi);
    }
    H.blob.resize(H.blob.size() + 16, 0); // the walk kernel compares 8 bytes at a time
    UP(H.blob, pat_blob)
    UP(H.offsets, pat_off)
#undef UP
    if ((rc = upload(a, H.classes, (size_t)256, &D.classes)) != ACX_OK) return destroy(rc);
    if ((rc = upload(a, &a->dev, (size_t)1, &a->d_dev)) != ACX_OK) return destroy(rc);
    HIPCHK_A(hipStreamSynchronize(a->stream));
    a->table_bytes = H.table.size() * 4;
    // the big host copy of the table is no longer needed
    std::vector<uint32_t>().swap(H.table);
    // kernel selection
    bool prefilter_ok = H.filter_q >= 3 && a->max_lds >= prefilter_lds_bytes();
    if (implementation == ACX_IMPL_NONCONTIGUOUS_NFA || implementation == ACX_IMPL_CONTIGUOUS_NFA)
        a->kernel = ACX_KERNEL_DFA_WALK;
    else
        a->kernel = prefilter_ok ? ACX_KERNEL_PREFILTER : ACX_KERNEL_DFA_WALK;
    if (const char *envk = std::getenv("ACX_KERNEL")) {
        if (!std::strcmp(envk, "dfa_walk")) { a->kernel = ACX_KERNEL_DFA_WALK; a->kernel_forced = true; }
        else if (!std::strcmp(envk, "prefilter") && H.filter_q >= 1 &&
                 a->max_lds >= prefilter_lds_bytes()) {
            a->kernel = ACX_KERNEL_PREFILTER;
            a->kernel_forced = true;
        }
    }
#undef HIPCHK_A
    *out = a;
    return ACX_OK;
}

int acx_compile_host(const uint8_t *blob, const uint64_t *offsets, uint64_t n_patterns,
                     int match_kind, acx_host_automaton_t **out) {
    if (!out) return fail(ACX_EINVAL, "null output pointer");
    *out = nullptr;
    if (n_patterns && (!offsets || (!blob && offsets[n_patterns] != offsets[0])))
        return fail(ACX_EINVAL, "null pattern buffer");
    acx_host_automaton *h = new (std::nothrow) acx_host_automaton();
    if (!h) return fail(ACX_ENOMEM, "out of memory");
    static const uint64_t zero_off[1] = {0};
    int code = ACX_OK;
    std::string err;
    try {
        err = compile(blob, n_patterns ? offsets : zero_off, n_patterns, match_kind, h->host, code);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(ACX_ENOMEM, "out of host memory while compiling the automaton");
    }
    if (code != ACX_OK) { delete h; return fail(code, err); }
    *out = h;
    return ACX_OK;
}

int acx_host_tables(const acx_host_automaton_t *h, acx_host_tables_t *out) {
    if (!h || !out) return fail(ACX_EINVAL, "null argument");
    const Automaton &A = h->host;
    out->n_patterns = A.n_patterns; out->n_states = A.n_states;
    out->n_classes = A.n_classes; out->stride = A.stride;
    out->min_pattern_len = A.min_len; out->max_pattern_len = A.max_len;
    out->classes = A.classes; out->table = A.table.data();
    out->own_off = A.own_off.data(); out->own_pid = A.own_pid.data();
    out->dlink = A.dlink.data(); out->level_start = A.level_start.data();
    out->pattern_len = A.plen.data(); out->rank = A.rank.data();
    out->filter_xy = A.filterA.data();
    out->prefix_table = A.ptab.data();
    out->filter_q = A.filter_q; out->filter_q2 = A.filter_q2;
    out->filter_entries_log2 = FILTER_ENTRIES_LOG2; out->prefix_table_log2 = A.ptab_log2;
    out->filter_density = A.filter_density;
    return ACX_OK;
}

uint32_t acx_filter_hash(uint32_t gram) { return filter_hash(gram); }
uint32_t acx_prefix_slot(uint64_t gram, uint32_t log2) { return prefix_slot(gram_hash2(gram), log2); }

void acx_free_host(acx_host_automaton_t *h) { delete h; }

void acx_free_automaton(acx_automaton_t *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    for (void *p : a->allocs) (void)hipFree(p);
    free_ws(a->ws, a->device);
    for (auto &ev : a->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : a->chunk_ev) if (ev) (void)hipEventDestroy(ev);
    if (a->post_ev) (void)hipEventDestroy(a->post_ev);
    if (a->post_stream) { (void)hipStreamSynchronize(a->post_stream); (void)hipStreamDestroy(a->post_stream); }
    if (a->stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

int acx_automaton_info(const acx_automaton_t *a, acx_info_t *out) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->n_patterns = a->host.n_patterns;
    out->n_states = a->host.n_states;
    out->n_classes = a->host.n_classes;
    out->stride = a->host.stride;
    out->min_pattern_len = a->host.min_len;
    out->max_pattern_len = a->host.max_len;
    out->table_bytes = a->table_bytes;
    out->lds_hot_rows = std::min(a->dev.hot_rows,
                                 dfa_walk_hot_rows(a->host.n_states, a->host.stride2, a->max_lds));
    out->kernel = a->kernel;
    out->match_kind = a->host.match_kind;
    out->device = a->device;
    out->filter_q = a->host.filter_q;
    return ACX_OK;
}

int acx_set_kernel(acx_automaton_t *a, int kernel) {
    if (!a) return fail(ACX_EINVAL, "null automaton");
    if (kernel == ACX_KERNEL_DFA_WALK) { a->kernel = kernel; a->kernel_forced = true; return ACX_OK; }
    if (kernel == ACX_KERNEL_PREFILTER) {
        if (a->host.filter_q == 0 || a->max_lds < prefilter_lds_bytes())
            return fail(ACX_EINVAL, "prefilter kernel unavailable for this automaton/device");
        a->kernel = kernel;
        a->kernel_forced = true;
        return ACX_OK;
    }
    if (kernel == ACX_KERNEL_AUTO) {
        a->kernel_forced = false;
        a->kernel = (a->host.filter_q >= 3 && a->max_lds >= prefilter_lds_bytes())
                        ? ACX_KERNEL_PREFILTER : ACX_KERNEL_DFA_WALK;
        return ACX_OK;
    }
    return fail(ACX_EINVAL, "unknown kernel");
}

int acx_find_device(acx_automaton_t *a, const void *d_hay, uint64_t len, const uint64_t *d_offsets,
                    uint64_t n_hay, uint64_t uniform_len, int overlapping, int codepoints,
                    acx_result_t **out) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    if (len && !d_hay) return fail(ACX_EINVAL, "null haystack");
    Segments G{nullptr, 1, 0};
    if (uniform_len) {
        if (n_hay * uniform_len != len) return fail(ACX_EINVAL, "n_hay * uniform_len != len");
        G.uniform_len = uniform_len; G.n_hay = n_hay;
    } else if (d_offsets) {
        G.offsets = d_offsets; G.n_hay = n_hay;
    }
    return run_find(a, (const uint8_t *)d_hay, len, G, overlapping, codepoints, out);
}

uint64_t acx_result_count(const acx_result_t *r) { return r ? r->n : 0; }
const acx_match_t *acx_result_device_matches(const acx_result_t *r) { return r ? r->d_matches : nullptr; }
const uint64_t *acx_result_device_counts(const acx_result_t *r) { return r ? r->d_counts : nullptr; }

int acx_result_copy(const acx_result_t *r, acx_match_t *host_out) {
    if (!r) return fail(ACX_EINVAL, "null result");
    if (!r->n) return ACX_OK;
    HIPCHK(hipSetDevice(r->device));
    HIPCHK(hipMemcpy(host_out, r->d_matches, r->n * sizeof(acx_match_t), hipMemcpyDeviceToHost));
    return ACX_OK;
}

int acx_result_copy_counts(const acx_result_t *r, uint64_t *host_counts) {
    if (!r) return fail(ACX_EINVAL, "null result");
    if (!r->d_counts || !r->n_hay) return ACX_OK;
    HIPCHK(hipSetDevice(r->device));
    HIPCHK(hipMemcpy(host_counts, r->d_counts, r->n_hay * 8, hipMemcpyDeviceToHost));
    return ACX_OK;
}

void acx_free_result(acx_result_t *r) {
    if (!r) return;
    g_bufs.put(r->d_matches, r->device);
    g_bufs.put(r->d_counts, r->device);
    delete r;
}

int acx_find(acx_automaton_t *a, const uint8_t *hay, uint64_t len, int overlapping, int codepoints,
             acx_match_t **out, uint64_t *n_out) {
    if (!a || !out || !n_out) return fail(ACX_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    if (len && !hay) return fail(ACX_EINVAL, "null haystack");
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD) {
        acx_result_t *dummy = nullptr; // produces the error message, touches no device state
        return run_find(a, nullptr, 0, Segments{nullptr, 1, 0}, overlapping, codepoints, &dummy);
    }
    acx_result_t *r = nullptr;
    int rc;
    const bool try_small = small_ok(a, len);
    if (try_small) {
        // small haystack: copy it into pinned memory, ONE launch (K0 reads and writes pinned host
        // memory in place), one sync -- no H2D / D2H copies at all
        std::lock_guard<std::mutex> lk(a->stage_mu);
        std::lock_guard<std::mutex> lock(a->mu);
        HIPCHK(hipSetDevice(a->device));
        Workspace &w = a->ws;
        if (!w.pin_hay) {
            HIPCHK(hipHostMalloc((void **)&w.pin_hay, SMALL_MAX_LEN + 16, hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&w.pin_out, SMALL_MAX_OCC * sizeof(acx_match_t), hipHostMallocDefault));
        }
        std::memcpy(w.pin_hay, hay, len);
        uint64_t n = 0;
        bool done = false;
        rc = run_small(a, w.pin_hay, len, overlapping, codepoints, w.pin_out, &n, &done);
        if (rc != ACX_OK) return rc;
        if (done) {
            if (n) {
                acx_match_t *m = (acx_match_t *)std::malloc(n * sizeof(acx_match_t));
                if (!m) return fail(ACX_ENOMEM, "out of memory");
                std::memcpy(m, w.pin_out, n * sizeof(acx_match_t));
                *out = m;
            }
            *n_out = n;
            return ACX_OK;
        }
    }
    {
        // the staging buffer is shared by the host-memory entry points
        std::lock_guard<std::mutex> lk(a->stage_mu);
        rc = stage_host(a, hay, len, nullptr, 0);
        if (rc == ACX_OK)
            rc = run_find(a, a->ws.hay, len, Segments{nullptr, 1, 0}, overlapping, codepoints, &r, !try_small);
    }
    if (rc != ACX_OK) return rc;
    uint64_t n = acx_result_count(r);
    if (n) {
        acx_match_t *m = (acx_match_t *)std::malloc(n * sizeof(acx_match_t));
        if (!m) { acx_free_result(r); return fail(ACX_ENOMEM, "out of memory"); }
        rc = acx_result_copy(r, m);
        if (rc != ACX_OK) { std::free(m); acx_free_result(r); return rc; }
        *out = m;
    }
    *n_out = n;
    acx_free_result(r);
    return ACX_OK;
}

void acx_free_matches(acx_match_t *m) { std::free(m); }

int acx_find_batch(acx_automaton_t *a, const uint8_t *hay, const uint64_t *offsets, uint64_t n_hay,
                   int overlapping, int codepoints, acx_match_t **out, uint64_t *n_out,
                   uint64_t *counts) {
    if (!a || !out || !n_out || !offsets) return fail(ACX_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    for (uint64_t i = 0; i < n_hay; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ACX_EINVAL, "offsets not monotone");
        if (counts) counts[i] = 0;
    }
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD) {
        acx_result_t *dummy = nullptr;
        return run_find(a, nullptr, 0, Segments{nullptr, 1, 0}, overlapping, codepoints, &dummy);
    }
    if (n_hay == 0) return ACX_OK;
    uint64_t base = offsets[0], len = offsets[n_hay] - base;
    std::vector<uint64_t> rel(n_hay + 1);
    for (uint64_t i = 0; i <= n_hay; i++) rel[i] = offsets[i] - base;
    acx_result_t *r = nullptr;
    int rc;
    {
        std::lock_guard<std::mutex> lk(a->stage_mu);
        rc = stage_host(a, hay ? hay + base : nullptr, len, rel.data(), n_hay + 1);
        if (rc == ACX_OK) {
            Segments G{a->ws.offsets, n_hay, 0};
            rc = run_find(a, a->ws.hay, len, G, overlapping, codepoints, &r);
        }
    }
    if (rc != ACX_OK) return rc;
    uint64_t n = acx_result_count(r);
    if (n) {
        acx_match_t *m = (acx_match_t *)std::malloc(n * sizeof(acx_match_t));
        if (!m) { acx_free_result(r); return fail(ACX_ENOMEM, "out of memory"); }
        rc = acx_result_copy(r, m);
        if (rc != ACX_OK) { std::free(m); acx_free_result(r); return rc; }
        *out = m;
    }
    if (counts && rc == ACX_OK) rc = acx_result_copy_counts(r, counts);
    *n_out = n;
    acx_free_result(r);
    return rc;
}

int acx_profile_enable(acx_automaton_t *a, int on) {
    if (!a) return fail(ACX_EINVAL, "null automaton");
    a->prof = on != 0;
    return ACX_OK;
}

int acx_profile_read(acx_automaton_t *a, acx_profile_t *out, int reset) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(a->mu);
    settle_post_profile(a);
    *out = a->profile;
    if (reset) a->profile = acx_profile_t{};
    return ACX_OK;
}

int acx_device_alloc(void **d_ptr, uint64_t bytes) {
    if (!d_ptr) return fail(ACX_EINVAL, "null argument");
    HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 16));
    return ACX_OK;
}
int acx_device_free(void *d_ptr) { HIPCHK(hipFree(d_ptr)); return ACX_OK; }
int acx_device_upload(void *d_dst, const void *h_src, uint64_t bytes) {
    if (bytes) HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ACX_OK;
}
int acx_device_download(void *h_dst, const void *d_src, uint64_t bytes) {
    if (bytes) HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ACX_OK;
}
int acx_device_synchronize(void) { HIPCHK(hipDeviceSynchronize()); return ACX_OK; }

int acx_generate_haystack(acx_automaton_t *a, void *d_dst, uint64_t len, int kind, uint64_t seed,
                          uint64_t stream_offset) {
    if (!a || (!d_dst && len)) return fail(ACX_EINVAL, "null argument");
    if (kind != 0 && kind != 1) return fail(ACX_EINVAL, "unknown haystack kind");
    if (kind == 1 && (stream_offset % 1024)) return fail(ACX_EINVAL, "stream_offset must be a multiple of 1024");
    std::lock_guard<std::mutex> lock(a->mu);
    HIPCHK(hipSetDevice(a->device));
    HIPCHK(generate(a->dev, (uint8_t *)d_dst, len, kind, seed, stream_offset, a->stream));
    HIPCHK(hipStreamSynchronize(a->stream));
    return ACX_OK;
}

} // extern "C"
