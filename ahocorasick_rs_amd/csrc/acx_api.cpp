// acx_api.cpp -- implementation of the C ABI declared in include/acx.h.
// Host orchestration of the device pipeline (kernels.hip).  There is no CPU
// matching path here: without a HIP device every find call fails (ACX_EDEVICE).
//
// Concurrency (reference: methods take a shared PyRef and release the GIL, the module is
// gil_used = false -- /root/reference/src/lib.rs:238, 261, 433, 438): a handle owns a small pool of
// *contexts* (stream + workspace + pinned scratch); every call leases one, so calls from different
// threads on ONE automaton run side by side on different streams instead of queueing on a lock.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
// The device's occurrence indexes are 32 bits wide: a call that would enumerate more is cut into byte ranges (run_chunked).
// The limit of ONE pass (ACX_MAX_OCC lowers it: tests) and what a pass returns when it hits it -- run_find's business,
// never the caller's.
// ACX_HOST_TRACE=1 (measurements): where the host's microseconds of a device-resident call go -- steady-clock stamps at ten
// points of acx_find_device .. acx_free_result, the mean of every interval printed when the process ends.
struct HostTrace {
    static constexpr int N = 10;
    bool on = std::getenv("ACX_HOST_TRACE") != nullptr;
    int64_t t[N] = {}, sum[N] = {};
    uint64_t rounds = 0;
    static int64_t now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(int i) {
        if (!on) return;
        const int64_t v = now();
        if (i == 0 && t[N - 1]) sum[0] += v - t[N - 1]; // (from the end of the last round: the caller's own time)
        if (i > 0 && t[i - 1]) sum[i] += v - t[i - 1];
        t[i] = v;
        if (i == N - 1) rounds++;
    }
    ~HostTrace() {
        if (!on || !rounds) return;
        static const char *what[N] = {"caller (free .. next call)", "lease", "up to the scan's launch", "the scan's launch", "the post kernels' launches",
                                      "events up to the wait", "wait for the totals' line", "return", "caller (return .. free)", "free"};
        std::fprintf(stderr, "ACX_HOST_TRACE: %llu rounds, mean microseconds per interval\n", (unsigned long long)rounds);
        for (int i = 0; i < N; i++) std::fprintf(stderr, "  %-32s %8.2f\n", what[i], sum[i] / 1e3 / rounds);
    }
};
HostTrace g_trace;
// (the same for the host-memory entry point acx_find beyond K0's sizes: copy in, pipeline, copy out)
struct HostTrace3 {
    bool on = std::getenv("ACX_HOST_TRACE") != nullptr;
    int64_t t0 = 0, sum[4] = {};
    uint64_t rounds = 0;
    void begin() { if (on) t0 = HostTrace::now(); }
    void lap(int i) { if (!on) return; const int64_t v = HostTrace::now(); sum[i] += v - t0; t0 = v; if (i == 3) rounds++; }
    ~HostTrace3() {
        if (!on || !rounds) return;
        static const char *what[4] = {"stage (host -> device copy queued)", "pipeline until the totals are known", "wait + device -> host copy", "free"};
        std::fprintf(stderr, "ACX_HOST_TRACE acx_find: %llu rounds, mean microseconds\n", (unsigned long long)rounds);
        for (int i = 0; i < 4; i++) std::fprintf(stderr, "  %-40s %8.2f\n", what[i], sum[i] / 1e3 / rounds);
    }
};
HostTrace3 g_trace_find;
constexpr int TOO_MANY_OCC = -1006;
uint64_t occ_limit() {
    const char *e = std::getenv("ACX_MAX_OCC");
    const uint64_t hard = (1ull << 32) - 2;
    if (!e) return hard;
    const uint64_t v = std::strtoull(e, nullptr, 10);
    return v && v < hard ? v : hard;
}
int fail_occ() { return fail(TOO_MANY_OCC, "more than 2^32 occurrences in one pass"); }
int hipfail(hipError_t e, const char *what) {
    (void)hipGetLastError(); // the runtime's "last error" is sticky: the next launch check must not see this one
    return fail(e == hipErrorOutOfMemory ? ACX_ENOMEM : ACX_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(expr)                                   \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) return hipfail(e__, #expr); \
    } while (0)

inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

// makes `dev` the calling thread's HIP device for a scope and restores the previous one
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------
// process-wide pools
// ---------------------------------------------------------------------------
// events that mark "this result's device work is done"
struct EventPool {
    std::mutex mu;
    std::vector<std::pair<int, hipEvent_t>> free_list;
    hipEvent_t get(int dev) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].first == dev) {
                    hipEvent_t e = free_list[i].second;
                    free_list.erase(free_list.begin() + i);
                    return e;
                }
        }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        return e;
    }
    void put(int dev, hipEvent_t e) {
        if (!e) return;
        std::lock_guard<std::mutex> lk(mu);
        if (free_list.size() >= 64) { (void)hipEventDestroy(e); return; }
        free_list.push_back({dev, e});
    }
};
EventPool g_events;

// Small cache of device buffers for results, so that a find call does not pay
// hipMalloc/hipFree (each tens of microseconds and a device sync).  A buffer may come back
// while the kernel that fills it is still running (the call returned as soon as the totals
// were known): it waits in `deferred` until its event has fired.
struct BufCache {
    struct Ent { void *p; size_t bytes; int dev; };
    struct Deferred { void *p; void *p2; int dev; hipEvent_t ev; }; // (p2: a second buffer behind the same event, or null)
    std::mutex mu;
    std::vector<Ent> free_list;
    std::vector<Deferred> deferred;
    std::unordered_map<void *, Ent> live; // every buffer handed out by get()
    size_t cached = 0;
    static constexpr size_t MAX_CACHED = (size_t)4 << 30;

    void release_locked(void *p, int dev) {
        auto it = live.find(p);
        const size_t bytes = it == live.end() ? 0 : it->second.bytes;
        if (!bytes || cached + bytes > MAX_CACHED || free_list.size() >= 24) {
            if (it != live.end()) live.erase(it);
            DeviceScope ds(dev);
            (void)hipFree(p);
            return;
        }
        free_list.push_back({p, bytes, dev});
        cached += bytes;
    }
    void sweep_locked() {
        for (size_t i = 0; i < deferred.size();) {
            if (hipEventQuery(deferred[i].ev) == hipErrorNotReady) { i++; continue; }
            g_events.put(deferred[i].dev, deferred[i].ev);
            release_locked(deferred[i].p, deferred[i].dev);
            if (deferred[i].p2) release_locked(deferred[i].p2, deferred[i].dev);
            deferred.erase(deferred.begin() + i);
        }
    }
    hipError_t get(void **out, size_t bytes, int dev) {
        bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!deferred.empty()) sweep_locked();
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
        DeviceScope ds(dev);
        hipError_t e = hipMalloc(out, alloc);
        if (e == hipErrorOutOfMemory) { // give back everything the cache holds idle, then try once more
            (void)hipGetLastError();
            {
                std::lock_guard<std::mutex> lk(mu);
                sweep_locked();
                for (const Ent &f : free_list) {
                    live.erase(f.p);
                    DeviceScope fs(f.dev);
                    (void)hipFree(f.p);
                }
                free_list.clear();
                cached = 0;
            }
            e = hipMalloc(out, alloc);
            if (e != hipSuccess) e = hipMalloc(out, alloc = bytes);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            live[*out] = {*out, alloc, dev};
        }
        return e;
    }
    // ev != null: work that writes the buffer may still be running; ev fires when it is done
    // (ownership of the event passes to the cache)
    void put(void *p, int dev, hipEvent_t ev = nullptr, void *p2 = nullptr) {
        if (!p) { p = p2; p2 = nullptr; }
        if (!p) { g_events.put(dev, ev); return; }
        std::lock_guard<std::mutex> lk(mu);
        if (ev) {
            if (hipEventQuery(ev) == hipErrorNotReady) { deferred.push_back({p, p2, dev, ev}); return; }
            g_events.put(dev, ev);
        }
        release_locked(p, dev);
        if (p2) release_locked(p2, dev);
    }
};
BufCache g_bufs;

// Pinned host buffers handed to callers as acx_find results (the D2H copy lands in them and
// the caller reads them in place: no second copy).  acx_free_matches() gives them back.
struct PinnedResults {
    struct Ent { void *p; size_t bytes; bool used; };
    std::mutex mu;
    std::vector<Ent> all;
    size_t total = 0;
    static constexpr size_t MAX_TOTAL = (size_t)2 << 30;
    void *get(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (int i = 0; i < (int)all.size(); i++)
            if (!all[i].used && all[i].bytes >= bytes && (best < 0 || all[i].bytes < all[best].bytes)) best = i;
        if (best >= 0) { all[best].used = true; return all[best].p; }
        size_t alloc = std::max<size_t>(bytes + bytes / 4, 1 << 20);
        if (total + alloc > MAX_TOTAL) { // drop idle buffers, then give up (the caller falls back to malloc)
            for (size_t i = 0; i < all.size();)
                if (!all[i].used) { (void)hipHostFree(all[i].p); total -= all[i].bytes; all.erase(all.begin() + i); }
                else i++;
            if (total + alloc > MAX_TOTAL) return nullptr;
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, alloc, hipHostMallocDefault) != hipSuccess) return nullptr;
        all.push_back({p, alloc, true});
        total += alloc;
        return p;
    }
    bool put(void *p) { // false: not one of ours
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : all)
            if (e.p == p) { e.used = false; return true; }
        return false;
    }
};
PinnedResults g_pinned_results;

// Host threads that copy a caller's pageable bytes into pinned staging chunks (one thread moves
// ~10 GB/s; the PCIe link wants ~55): created on the first large host-memory call.
class CopyPool {
public:
    static CopyPool &get() {
        static CopyPool p;
        return p;
    }
    int threads() const { return (int)workers_.size() + 1; }
    // memcpy(dst, src, n) by all threads; returns when done.  One job at a time.
    void copy(void *dst, const void *src, size_t n) {
        const int T = threads();
        if (n < (1u << 20) || T == 1) { std::memcpy(dst, src, n); return; }
        std::lock_guard<std::mutex> job(job_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = (uint8_t *)dst; src_ = (const uint8_t *)src; n_ = n;
            pending_ = T - 1;
            gen_++;
        }
        cv_.notify_all();
        slice(0, T);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    CopyPool() {
        int T = (int)std::thread::hardware_concurrency() / 2;
        if (const char *e = std::getenv("ACX_COPY_THREADS")) T = std::atoi(e);
        T = std::max(1, std::min(T, 16));
        for (int i = 1; i < T; i++) workers_.emplace_back([this, i, T] { run(i, T); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void slice(int i, int T) {
        const size_t per = ((n_ + T - 1) / T + 4095) & ~(size_t)4095;
        const size_t lo = std::min(n_, per * i), hi = std::min(n_, per * (i + 1));
        if (hi > lo) std::memcpy(dst_ + lo, src_ + lo, hi - lo);
    }
    void run(int i, int T) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            slice(i, T);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    uint8_t *dst_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t n_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// ---------------------------------------------------------------------------
// pinned host scratch of a context, in 64-bit words (Workspace::h_pinned)
constexpr uint32_t PIN_TOTALS = 24, PIN_HOT_TOTALS = 32, PIN_SPEC_TOTALS = 40, PIN_RESIDENT = 48, PIN_K0 = 64, PINNED_WORDS = 96;
struct Workspace {
    uint64_t cap = 0; // occurrence capacity of the dense path (region mode + radix sort)
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *pids[2] = {nullptr, nullptr};
    uint64_t *S = nullptr, *E = nullptr, *M = nullptr;
    uint32_t *flags = nullptr, *idx = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    uint4 *recs = nullptr;            // occurrence sink: cap records of 16 B in per-workgroup regions
    uint4 *hrecs = nullptr;           // dense path, K1b prefix-hit sink: hit_total records of 32 B
    uint64_t hit_total = 0;
    uint64_t *hit_counts = nullptr;   // device: one per K1b wave
    uint64_t *summary = nullptr;      // device: [0] occurrences kept, [1] max per region, [2..3] same for
                                      // hits, [4] matches written, [5..6] abort flags of the sparse path
    uint64_t *block_counts = nullptr; // device: one per scan workgroup
    uint64_t *region_off = nullptr;   // device: exclusive prefix of the kept counts
    uint64_t *h_pinned = nullptr;     // pinned host scratch (PINNED_WORDS x u64, 64-byte lines): [0 .. 15] the dense paths' totals
                                      // (copied behind a stream synchronisation; [8], [9] = result of an unpolled K0), then the
                                      // POLLED lines, each written by one store and accepted on its check word (kernels.hpp):
                                      // [64 .. 95] K0's result (up to four lines), [24 .. 31] the sparse path's totals, [32 .. 39] the hot pipeline's
                                      // early total (hot_totals), [40 .. 47] the speculative hot pipeline's, [48] the epoch of
                                      // the last resident K0 that has left (Resident)
    uint64_t t_line[8] = {};          // the sparse path's totals: the verified copy of the line (PIN_TOTALS / PIN_HOT_TOTALS)
    uint64_t h_lines[K0_RESULT_LINES][8] = {}; // K0, polled: the verified copies of the call's result lines (words 1 .. 6 of each)
    acx_match_t *pin_final = nullptr; // host entry point, mid-size calls: pinned host memory the write kernel's records go to
    uint64_t pin_final_cap = 0;       // (records)
    uint8_t *pin_mid = nullptr;       // mid-size calls: pinned copy of a host haystack the scan reads in place
    uint64_t pin_mid_cap = 0;
    uint64_t *mailbox = nullptr;      // small calls: coherent pinned memory -- the resident K0's command word (kernels.hpp), then
    uint8_t *pin_hay = nullptr;       //   (K0_MAILBOX_HAY bytes behind it) the copy of a host haystack K0 reads in place
    acx_match_t *pin_out = nullptr;   // small calls: pinned output of K0 (host entry point)
    uint64_t *blockcnt = nullptr, *blockpre = nullptr; // lead bytes per 1 KiB block / their prefix
    uint8_t *blocksub = nullptr;                        // lead bytes per 64 bytes of a block
    uint64_t block_cap = 0;
    DenseTiles dt{};                  // dense path, tile-ordered: occurrence buckets by key tile
    TileSpace TD{};                   //   its groups' reported occurrences (64-bit words), counts, supergroup words
    uint64_t dt_cap = 0;              //   tiles both are allocated for
    TileSpace T{};                    // sparse path (hit slots + tile kernels)
    uint64_t tile_cap = 0;            // tiles T is allocated for
    uint64_t group_cap = 0;           // groups T.gstate is allocated for
    uint32_t trecs_gmax = 0;          // records per group T.trecs is allocated for (GROUP_MAX; GROUP_MAX_WIDE once a call needed it)
    bool flags_dirty = true;          // the control blocks' counters are not known to be zero
    uint32_t *ctl = nullptr;          // device: the sparse path's two control blocks (device_types.hpp), used by the calls in turn
    uint4 *ovf_recs = nullptr;        // K1b's hits beyond a tile's slots: OVF_LISTS lists of ovf_cap records of 32 B
    uint64_t ovf_cap = 0;             //   records per list
    uint32_t *ovf_counts = nullptr;   //   the lists' fill counters, two sets (one per control block), a cache line each
    uint32_t *hot_list = nullptr;     // groups left to the hot pipeline (group_cap ids)
    acx_match_t *final = nullptr;     // sparse path: output buffer the next call writes into
    uint64_t final_cap = 0;
    uint8_t *hay = nullptr;           // device staging buffer of the host-memory entry points
    uint64_t hay_cap = 0;
    uint64_t *offsets = nullptr;
    uint64_t offsets_cap = 0;
    // pipelined host -> device staging: ring of pinned chunks
    static constexpr int RING = 3;
    uint8_t *pin_chunk[RING] = {nullptr, nullptr, nullptr};
    hipEvent_t chunk_ev[RING] = {nullptr, nullptr, nullptr};
    size_t chunk_bytes = 0;
};

// The resident K0 of a context (kernels.hip, k0_resident): one workgroup that stays on the device between the calls of a
// loop over short haystacks and is fed through the workspace's mailbox, so that a call costs a poll on either side instead
// of a launch.  At most one of them per context, and nothing else of the context runs beside it (streams may share a
// hardware queue: whatever else the context launches first tells the kernel to leave -- stop_resident).  It leaves by
// itself after idle_us without a call and life_us after its launch (ACX_RESIDENT_IDLE_US, ACX_RESIDENT_LIFE_US;
// ACX_NO_RESIDENT=1: every small call is a launch, as until round 5).
struct Resident {
    hipStream_t stream = nullptr; // its own: created with the first launch
    uint64_t epoch = 0;           // the number of the last launch; h_pinned[PIN_RESIDENT] == epoch: that kernel has left
    bool live = false;            // a kernel has been launched and has not been seen to have left
    int mode = -1, overlapping = 0; // what it was launched for (small_mode; the tables' view and the key follow from overlapping)
    uint32_t delay = 0;           // ticks the kernel waits behind a result before it polls (k0_resident: what the last kernel
                                  // ended with -- h_pinned[PIN_RESIDENT + 5])
    uint64_t secret = 0;          // keys the check of the haystack bytes that travel with the poll (kernels.hpp, k0_hay_check)
    uint32_t switches = 0, calls = 0; // launches for another mode within the last calls: a loop that alternates between two
    uint32_t off = 0;                 //   kinds of call pays a launch per call either way -- small calls left as plain launches
};

// everything one in-flight call needs
struct Ctx {
    hipStream_t stream = nullptr, copy_stream = nullptr;
    Resident res;
    // profiling: [0], [1] and [3], [4]: scan start / stop, two pairs used by the calls in turn (the time
    // of a call's scan is read while the NEXT call's kernels run, off the path between two calls);
    // [2]: end of the call
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t prof_calls = 0;   // calls of this context while profiling was on (sampling)
    int ev_pair = 0;           // the pair the next scan launch takes
    bool scan_pending = false; // a scan's time has not been read yet
    int pend_pair = 0;
    uint64_t pend_len = 0;
    hipEvent_t copy_done = nullptr;                 // staging: the last chunk has landed
    hipEvent_t fork_ev = nullptr, join_ev = nullptr; // str API: the code-point prefix runs beside k_tile_main
    Workspace ws;
    bool post_pending = false; // profiling: ev[2] of the last call has not been read yet
    int dense_hold = 0;        // > 0: the output was too dense for the sparse path; calls left in region mode
    bool hold_dense_input = false; // why: the INPUT was dense (the hold ends with the first call that is not) -- or the sparse
                                   // kernels gave up on it for another reason (counted down: one failed attempt in nine calls)
    uint32_t spec_hot = 0;     // > 0: the last call had this many hot groups (few): the next one queues the hot pipeline ahead of
                               // knowing that it needs it (kernels.hip: hot_groups_here), spec_ovf: that call's fullest overflow list
    uint32_t spec_ovf = 0;
    uint64_t hot_inline = 16;  // hot groups the output buffer of a sparse attempt has room for (HOT_INLINE .. HOT_INLINE_MAX)
    int dense_full = 0;        // > 0: calls left for which the tile-ordered dense path runs k_dense_main in its full form (a group
                               // did not fit the compact stage)
    bool wide = false;         // the sparse path's post stage runs in its WIDE form (device_types.hpp: GROUP_MAX_WIDE): the last
                               // call's groups mostly gave up on the narrow one (a match every 100 - 500 bytes)
    int flag_idx = 0;          // which of the two abort flags the next sparse attempt uses
    uint64_t seq = 0;          // sequence number the write kernel publishes in the totals' line (h_pinned + PIN_TOTALS)
    uint64_t small_seq = 0;    // K0 (host entry point): the number its result line carries (h_pinned + PIN_K0)
};

} // namespace

struct acx_automaton {
    Automaton host;
    int device = 0;
    DevAutomaton dev{};
    const DevAutomaton *d_dev = nullptr; // the same struct, resident in HBM
    // Copies of a pattern (Standard automata keep them: an overlapping search reports every copy).  A NON-overlapping
    // search can only ever report the lowest id of a string, and the device enumerates every occurrence it is given --
    // hundreds of copies of every pattern on text where every position matches were hundreds of times the work
    // (tools/gpu_fuzz.py, seed 40404).  dev_nov = dev with two tables replaced: own1 holds the lowest id of every
    // state's string (all patterns that end in a trie state ARE one string: never OWN1_MANY), and blist has every
    // candidate list's first-of-their-string ids in front and counts only those (same list indexes: the prefix table
    // and the short patterns' codes are shared).  The DFA walk reports through the own lists (own_off / own_pid: one
    // entry per state in the view) and through its trie records (grec: own1 in their third word).  Taken by the
    // kernels that read those tables (K0, k_tile_main, k_walk_hits, k_dense_verify; k1a_walk, the walks' emit
    // paths through the device-resident copy d_dev_nov) when the call is not overlapping; has_nov = false: no copies.
    // Round 5: an OVERLAPPING search takes the view too -- one occurrence per string under its lowest id -- and the result
    // is expanded where it is complete (expand_copies: every occurrence becomes the run of its string's copies, ids
    // ascending, as the reference reports them): the copies cost their records, not a verification, a sort slot and a
    // 2^32-limited index each.  x_cnt[pid] = the later copies of a lowest id (0 otherwise), x_off[pid] = where their ids
    // begin in x_ids (host: K0's pinned result is expanded on the host; d_x*: the same in HBM).
    DevAutomaton dev_nov{};
    const DevAutomaton *d_dev_nov = nullptr;
    bool has_nov = false;
    // expand_ov: overlapping searches do that -- when at least a quarter of the ids are later copies (ACX_EXPAND_COPIES=1 / 0:
    // whenever there is one / never).  The expansion is a pass over the complete result and a round trip for its size
    // (cfg4's 100 000 random patterns hold half a dozen accidental duplicates: 0.79 -> 0.87 ms when it was taken for them);
    // a set with a few copies enumerates them on the device as before, at the cost of those few.
    bool expand_ov = false;
    std::vector<uint32_t> x_cnt, x_off, x_ids;
    const uint32_t *d_xcnt = nullptr, *d_xoff = nullptr, *d_xids = nullptr;
    std::vector<void *> allocs;
    int kernel = ACX_KERNEL_DFA_WALK;
    int implementation = ACX_IMPL_AUTO; // the caller's hint (replicas are built with the same one)
    int n_cus = 1;
    size_t max_lds = 65536;
    uint64_t table_bytes = 0;
    bool kernel_forced = false; // the scan kernel was chosen explicitly: K0 never takes a call
    bool sparse_ok = true;      // tile_lookback(max_len) <= MAX_LOOKBACK
    // contexts
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    std::vector<Ctx *> ctxs, idle;
    int max_ctx = 4;
    // profiling (accumulated over the contexts)
    std::mutex prof_mu;
    std::atomic<bool> prof{false};
    std::atomic<int> prof_every{1}; // profiling events on every N-th call of a context
    acx_profile_t profile{};
    std::atomic<uint64_t> path[ACX_PATH_STATS] = {}; // acx_path_stats
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
    hipEvent_t done = nullptr; // non-null: device work that fills the buffers may still be running
    bool borrowed = false;     // d_matches is the context's pinned host buffer (acx_find: the write kernel's records land where
                               // the host reads them); never handed to a caller, never given to the buffer cache
};

namespace {

// the result's buffers are complete after this
int result_wait(const acx_result *r) {
    if (r && r->done) {
        DeviceScope ds(r->device);
        HIPCHK(hipEventSynchronize(r->done));
    }
    return ACX_OK;
}

template <typename T>
int upload(acx_automaton *a, hipStream_t st, const T *src, size_t count, const T **dst) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    bytes = (bytes + 15) / 16 * 16;
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, bytes));
    a->allocs.push_back(d);
    HIPCHK(hipMemsetAsync(d, 0, bytes, st));
    if (count) HIPCHK(hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    *dst = (const T *)d;
    return ACX_OK;
}

void free_tiles(Workspace &w) {
    TileSpace &T = w.T;
    (void)hipFree(T.hslots); (void)hipFree(T.hcnt); (void)hipFree(T.trecs); (void)hipFree(T.btot); (void)hipFree(T.sgw);
    (void)hipFree(w.hot_list); (void)hipFree(w.ovf_recs);
    w.hot_list = nullptr; w.ovf_recs = nullptr; w.ovf_cap = 0;
    T = TileSpace{};
    w.tile_cap = 0;
    w.group_cap = 0;
    w.trecs_gmax = 0;
}

void free_dense_tiles(Workspace &w) {
    (void)hipFree(w.dt.words); (void)hipFree(w.dt.counts);
    (void)hipFree(w.TD.trecs); (void)hipFree(w.TD.btot); (void)hipFree(w.TD.sgw);
    w.dt = DenseTiles{};
    w.TD = TileSpace{};
    w.dt_cap = 0;
}

void free_ws(Workspace &w, int device) {
    free_dense_tiles(w);
    for (int i = 0; i < 2; i++) { (void)hipFree(w.keys[i]); (void)hipFree(w.pids[i]); }
    (void)hipFree(w.S); (void)hipFree(w.E); (void)hipFree(w.M);
    (void)hipFree(w.flags); (void)hipFree(w.idx); (void)hipFree(w.temp);
    (void)hipFree(w.summary); (void)hipFree(w.ctl); (void)hipFree(w.ovf_counts); (void)hipFree(w.block_counts); (void)hipFree(w.region_off);
    (void)hipFree(w.recs); (void)hipFree(w.hrecs); (void)hipFree(w.hit_counts);
    free_tiles(w);
    g_bufs.put(w.final, device);
    (void)hipFree(w.blockcnt); (void)hipFree(w.blockpre); (void)hipFree(w.blocksub);
    (void)hipFree(w.hay); (void)hipFree(w.offsets);
    if (w.h_pinned) (void)hipHostFree(w.h_pinned);
    if (w.pin_final) (void)hipHostFree(w.pin_final);
    if (w.mailbox) (void)hipHostFree(w.mailbox);
    if (w.pin_mid) (void)hipHostFree(w.pin_mid);
    if (w.pin_out) (void)hipHostFree(w.pin_out);
    for (int i = 0; i < Workspace::RING; i++) {
        if (w.pin_chunk[i]) (void)hipHostFree(w.pin_chunk[i]);
        if (w.chunk_ev[i]) (void)hipEventDestroy(w.chunk_ev[i]);
    }
    w = Workspace();
}

// the context's resident K0 is told to leave, and has left when this returns
void resident_left(Ctx *c); // (below: trace + the delay the kernel ended with)
void trace_resident(Ctx *c) { // (ACX_RESIDENT_TRACE=1: what the kernel that has just left did -- kernels.hip, k0_resident)
    static const bool on = std::getenv("ACX_RESIDENT_TRACE") != nullptr;
    if (!on) return;
    const uint64_t *s = c->ws.h_pinned + PIN_RESIDENT;
    std::fprintf(stderr, "acx resident K0 epoch %llu: %llu calls, %llu with their bytes in the poll, %.2f us busy per call, %llu polls, delay %llu ticks\n",
                 (unsigned long long)s[0], (unsigned long long)s[1], (unsigned long long)s[2],
                 s[1] ? (double)s[3] / 100.0 / (double)s[1] : 0.0, (unsigned long long)s[4], (unsigned long long)s[5]);
}

void resident_left(Ctx *c) {
    trace_resident(c);
    const uint64_t d = c->ws.h_pinned[PIN_RESIDENT + 5];
    c->res.delay = d < 1000 ? (uint32_t)d : 0;
}

void stop_resident(Ctx *c) {
    Resident &R = c->res;
    if (!R.live) return;
    R.live = false;
    volatile uint64_t *status = c->ws.h_pinned + PIN_RESIDENT;
    struct AtExit { Ctx *c; ~AtExit() { resident_left(c); } } at_exit{c};
    if (*status == R.epoch) return;
    // (the word's call number is one the kernel is not waiting for: the quit flag is all it reads)
    __atomic_store_n(c->ws.mailbox, k0_mailbox_word(0, 0, false, true), __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *status != R.epoch; spins++) {
        cpu_relax();
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) {
            (void)hipStreamSynchronize(R.stream);
            break;
        }
    }
}

void destroy_ctx(Ctx *c, int device) {
    if (!c) return;
    if (c->res.stream) {
        stop_resident(c);
        (void)hipStreamSynchronize(c->res.stream);
        (void)hipStreamDestroy(c->res.stream);
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    free_ws(c->ws, device);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

Ctx *create_ctx() { // the automaton's device is current
    Ctx *c = new (std::nothrow) Ctx();
    if (!c) return nullptr;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) == hipSuccess;
    if (!ok) { destroy_ctx(c, 0); return nullptr; }
    return c;
}

// a context of the automaton for the duration of one call (and the automaton's device as the
// calling thread's current device)
struct Lease {
    acx_automaton *a;
    Ctx *c = nullptr;
    DeviceScope dev;
    // keep_resident: the call may go to the context's resident K0 (acx_find); every other call has the context to itself
    explicit Lease(acx_automaton *a_, bool keep_resident = false) : a(a_), dev(a_->device) {
        take();
        if (c && !keep_resident) stop_resident(c);
    }
    void take() {
        std::unique_lock<std::mutex> lk(a->pool_mu);
        for (;;) {
            if (!a->idle.empty()) { c = a->idle.back(); a->idle.pop_back(); return; }
            if ((int)a->ctxs.size() < a->max_ctx) {
                c = create_ctx();
                if (c) { a->ctxs.push_back(c); return; }
                if (a->ctxs.empty()) return; // nothing to wait for: the caller reports the failure
            }
            a->pool_cv.wait(lk);
        }
    }
    ~Lease() {
        if (!c) return;
        {
            std::lock_guard<std::mutex> lk(a->pool_mu);
            a->idle.push_back(c);
        }
        a->pool_cv.notify_one();
    }
};

int ensure_common(Ctx *c) {
    Workspace &w = c->ws;
    if (!w.summary) {
        HIPCHK(hipMalloc((void **)&w.summary, 128)); // [0..4] totals, [8], [9] scratch, [10], [11] flags of the dense / hot pipeline, [12], [13] the cut of a byte range
        HIPCHK(hipMemsetAsync(w.summary, 0, 128, c->stream)); // (its flags are cleared by the kernels that use them; recycled memory is not zero)
        HIPCHK(hipMalloc((void **)&w.ctl, 2 * CTL_WORDS * 4));
        // (every clearing of the workspace is queued on the CONTEXT'S stream: the stream does not wait for the null stream
        // (hipStreamNonBlocking), and a hipMemset there has been seen to run behind this context's first scan when another
        // thread kept the device busy -- round 6, tools/stress: a fresh handle's first batch lost its overflow hits)
        HIPCHK(hipMemsetAsync(w.ctl, 0, 2 * CTL_WORDS * 4, c->stream));
        HIPCHK(hipMalloc((void **)&w.ovf_counts, 2 * OVF_LISTS * OVF_COUNT_STRIDE * 4));
        HIPCHK(hipMemsetAsync(w.ovf_counts, 0, 2 * OVF_LISTS * OVF_COUNT_STRIDE * 4, c->stream));
        HIPCHK(hipMalloc((void **)&w.block_counts, 8 * 16400)); // counts of <= 8192 regions + their exact bases
        HIPCHK(hipMalloc((void **)&w.region_off, 8 * 8193));
        HIPCHK(hipMalloc((void **)&w.hit_counts, 8 * 16 * 1024));
        // polled by the host while kernels still run: system-coherent
        HIPCHK(hipHostMalloc((void **)&w.h_pinned, PINNED_WORDS * 8, hipHostMallocCoherent));
        std::memset(w.h_pinned, 0, PINNED_WORDS * 8);
        w.flags_dirty = true;
    }
    return ACX_OK;
}

// dense path: prefix-hit sink of K1b
int ensure_hits(Ctx *c, uint64_t want) {
    Workspace &w = c->ws;
    if (want <= w.hit_total) return ACX_OK;
    (void)hipFree(w.hrecs); w.hrecs = nullptr; w.hit_total = 0;
    HIPCHK(hipMalloc((void **)&w.hrecs, want * 32));
    w.hit_total = want;
    return ACX_OK;
}

// dense path: occurrence regions + everything the radix sort / resolve pipeline needs
int ensure_occ_capacity(Ctx *c, uint64_t want) {
    Workspace &w = c->ws;
    if (want <= w.cap) return ACX_OK;
    uint64_t cap = std::max<uint64_t>(want, 1u << 16);
    if (std::getenv("ACX_DEBUG_MEM")) {
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        std::fprintf(stderr, "acx: ensure_occ_capacity want %llu (old cap %llu), device free %.1f of %.1f GiB\n",
                     (unsigned long long)want, (unsigned long long)w.cap, fr / 1073741824.0, tot / 1073741824.0);
    }
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

// sparse path: the list of K1b's hits beyond their tiles' slots (dense stretches of the input; device_types.hpp: control
// block) -- room for `want` records; both control blocks learn where it is (and where the hot list is).  The stream is idle.
constexpr uint64_t OVF_PER_TILE = 16; // records per tile to start with (a quarter of the slots; grown when an input needs more)
int set_overflow_room(Ctx *c, uint64_t want) { // want: records per list
    Workspace &w = c->ws;
    want = std::min<uint64_t>(std::max<uint64_t>(want, 64), 0xFFFFFFF0ull / OVF_LISTS);
    if (want > w.ovf_cap) {
        // (the new lists first: a failed allocation leaves the old ones -- and the control blocks that point at them -- as they are)
        HIPCHK(hipStreamSynchronize(c->stream));
        uint4 *fresh = nullptr;
        HIPCHK(hipMalloc((void **)&fresh, want * OVF_LISTS * 32));
        (void)hipFree(w.ovf_recs);
        w.ovf_recs = fresh;
        w.ovf_cap = want;
    }
    uint32_t h[2 * CTL_WORDS] = {};
    for (int b = 0; b < 2; b++) {
        uint32_t *blk = h + b * CTL_WORDS;
        blk[CTL_OVF_CAP] = (uint32_t)w.ovf_cap;
        const uint64_t recs = (uint64_t)(uintptr_t)w.ovf_recs, list = (uint64_t)(uintptr_t)w.hot_list;
        const uint64_t counts = (uint64_t)(uintptr_t)(w.ovf_counts + (size_t)b * OVF_LISTS * OVF_COUNT_STRIDE);
        std::memcpy(blk + CTL_OVF_RECS, &recs, 8);
        std::memcpy(blk + CTL_HOT_LIST, &list, 8);
        std::memcpy(blk + CTL_OVF_COUNTS, &counts, 8);
    }
    HIPCHK(hipMemcpyAsync(w.ctl, h, sizeof h, hipMemcpyHostToDevice, c->stream)); // (the counters with them: clear)
    HIPCHK(hipMemsetAsync(w.ovf_counts, 0, 2 * OVF_LISTS * OVF_COUNT_STRIDE * 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream)); // (h is this function's; and on the context's stream: see ensure_common)
    w.flags_dirty = false;
    return ACX_OK;
}

// sparse path: hit slots for `tiles` tiles of index space, group arrays.  One bucket beyond the
// last tile exists (an occurrence may END exactly at the end of the last tile).
// Nothing is marked as allocated before every allocation has succeeded: a failure leaves the workspace without its tile
// arrays (free_tiles), never with control blocks that point at freed memory.
int ensure_tiles(acx_automaton *a, Ctx *c, uint64_t tiles, uint32_t gmax) {
    Workspace &w = c->ws;
    TileSpace &T = w.T;
    const uint64_t groups = (tiles + 1 + GROUP_TILES - 1) / GROUP_TILES;
    if (tiles > w.tile_cap || gmax > w.trecs_gmax) {
        HIPCHK(hipStreamSynchronize(c->stream));
        // (grown, never shrunk: what the larger of the two demands -- tiles, records per group -- had is kept)
        const uint64_t cap_tiles = tiles > w.tile_cap ? tiles + tiles / 8 + GROUP_TILES : w.tile_cap;
        const uint32_t cap_gmax = std::max(gmax, w.trecs_gmax);
        free_tiles(w);
        if (w.final) { g_bufs.put(w.final, a->device); w.final = nullptr; }
        const uint64_t cap_groups = (cap_tiles + 1 + GROUP_TILES - 1) / GROUP_TILES;
        const uint64_t cap_super = (cap_groups + 63) / 64;
        int rc = ACX_OK;
        auto grab = [&](void **p, uint64_t bytes) { if (rc == ACX_OK && hipMalloc(p, bytes) != hipSuccess) rc = hipfail(hipGetLastError(), "hipMalloc (tile workspace)"); };
        grab((void **)&T.hslots, cap_tiles * HIT_SLOTS * 32);
        grab((void **)&T.hcnt, (cap_tiles + 16 * 1024 + 16) * 4); // + one slot per K1b wave (layout slack)
        grab((void **)&T.trecs, cap_groups * cap_gmax * 16);
        grab((void **)&T.btot, cap_groups * 4);
        grab((void **)&T.sgw, 4 * cap_super * 8);
        grab((void **)&w.hot_list, cap_groups * 4);
        if (rc == ACX_OK && hipMemsetAsync(T.sgw, 0, 4 * cap_super * 8, c->stream) != hipSuccess) rc = hipfail(hipGetLastError(), "hipMemset"); // both sets start clear
        if (rc == ACX_OK) rc = set_overflow_room(c, (cap_tiles * OVF_PER_TILE + OVF_LISTS - 1) / OVF_LISTS);
        if (rc != ACX_OK) { free_tiles(w); return rc; }
        T.sg_cap = (uint32_t)cap_super;
        w.group_cap = cap_groups;
        w.tile_cap = cap_tiles;
        w.trecs_gmax = cap_gmax;
    }
    T.n_tiles = (uint32_t)tiles;
    T.n_groups = (uint32_t)groups;
    T.gmax = gmax;
    return ACX_OK;
}

// dense path, tile-ordered: buckets of DT_SLOTS words per key tile (tiles + 1 of them), DT_GMAX words per group
int ensure_dense_tiles(Ctx *c, uint64_t tiles) {
    Workspace &w = c->ws;
    const uint64_t key_tiles = tiles + 1;
    if (key_tiles > w.dt_cap) {
        free_dense_tiles(w);
        const uint64_t cap = key_tiles + key_tiles / 8 + DT_GROUP;
        const uint64_t cap_groups = (cap + DT_GROUP - 1) / DT_GROUP, cap_super = (cap_groups + 63) / 64;
        HIPCHK(hipMalloc((void **)&w.dt.words, cap * DT_SLOTS * 8));
        HIPCHK(hipMalloc((void **)&w.dt.counts, (cap + 16) * 4));
        HIPCHK(hipMalloc((void **)&w.TD.trecs, cap_groups * DT_GMAX * 8));
        HIPCHK(hipMalloc((void **)&w.TD.btot, cap_groups * 4));
        HIPCHK(hipMalloc((void **)&w.TD.sgw, 4 * cap_super * 8));
        HIPCHK(hipMemsetAsync(w.TD.sgw, 0, 4 * cap_super * 8, c->stream));
        w.TD.sg_cap = (uint32_t)cap_super;
        w.dt_cap = cap;
    }
    w.dt.n_tiles = (uint32_t)key_tiles;
    w.TD.n_tiles = (uint32_t)key_tiles;
    w.TD.n_groups = (uint32_t)((key_tiles + DT_GROUP - 1) / DT_GROUP);
    return ACX_OK;
}

int ensure_blocks(Ctx *c, uint64_t nblocks_plus1) {
    Workspace &w = c->ws;
    if (nblocks_plus1 > w.block_cap) {
        (void)hipFree(w.blockcnt); (void)hipFree(w.blockpre); (void)hipFree(w.blocksub);
        w.blockcnt = w.blockpre = nullptr; w.blocksub = nullptr; w.block_cap = 0;
        HIPCHK(hipMalloc((void **)&w.blockcnt, nblocks_plus1 * 8));
        HIPCHK(hipMalloc((void **)&w.blockpre, nblocks_plus1 * 8));
        HIPCHK(hipMalloc((void **)&w.blocksub, nblocks_plus1 * 64)); // one lead-byte count per 16 bytes
        w.block_cap = nblocks_plus1;
    }
    size_t need = std::max<size_t>(scan_temp_bytes(nblocks_plus1), 32768) + 256; // the scan temp storage must cover this size too (block_prefix: 32 KiB of partial sums)
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
    return !off && !a->kernel_forced && len > 0 && a->host.n_patterns > 0 &&
           (len <= SMALL_MAX_LEN || (len <= SMALL_PF_MAX_LEN && small_prefilter_ok(a->dev)));
}

// One K0 launch + one sync.  hay / out: anything the device can address (HBM or pinned host);
// out holds SMALL_MAX_OCC records.  *done = false: too many occurrences, use the general path.
// poll: hay and out are host memory the kernel reads / writes in place: wait for the number the kernel publishes
// behind its last store instead of synchronising the stream (tools/ubench_roundtrip.hip: 6 us against 11)
// the device tables a call takes: a non-overlapping search never needs the later copies of a string (acx_automaton::dev_nov)
// (an overlapping search as well since round 5: its result is expanded to the copies afterwards -- expand_copies)
inline const DevAutomaton &view(const acx_automaton *a, bool overlapping) {
    return (overlapping ? a->expand_ov : a->has_nov) ? a->dev_nov : a->dev;
}
inline const DevAutomaton *d_view(const acx_automaton *a, bool overlapping) { // (the same, resident in HBM)
    return (overlapping ? a->expand_ov : a->has_nov) ? a->d_dev_nov : a->d_dev;
}

// (ACX_SMALL_SYNC, measurements: always synchronise the stream -- and then the records are plain acx_match_t)
bool small_polls() {
    static const bool no_poll = std::getenv("ACX_SMALL_SYNC") != nullptr;
    return !no_poll;
}

int wait_line(Ctx *c, uint32_t at, uint64_t seq, uint64_t line[8], const char *what); // (below)
// the result lines behind the first that a polled K0 call with `n` matches wrote (kernels.hpp, K0_RESULT_LINES: the same
// store instruction as the first): verified copies into the workspace
int take_more_lines(Ctx *c, uint64_t seq, uint64_t n) {
    for (uint32_t L = 1; L < k0_result_lines(n); L++) {
        uint64_t line[K0_LINE_WORDS];
        int rc = wait_line(c, PIN_K0 + 8 * L, seq, line, "K0's matches did not arrive");
        if (rc) return rc;
        for (uint32_t i = 1; i < K0_LINE_WORDS - 1; i++) c->ws.h_lines[L][i] = line[i];
    }
    return ACX_OK;
}

int run_small(acx_automaton *a, Ctx *c, const uint8_t *hay, uint64_t len, int overlapping, int codepoints,
              acx_match_t *out, uint64_t *n_out, bool *done, bool poll = false) {
    *done = false;
    int rc = ensure_common(c);
    if (rc) return rc;
    Workspace &w = c->ws;
    const int key_mode = overlapping ? 0 : a->host.match_kind;
    const uint64_t seq = poll && small_polls() ? ++c->small_seq : 0;
    HIPCHK(launch_small(view(a, overlapping != 0), hay, (uint32_t)len, key_mode, overlapping != 0, codepoints != 0, out,
                        seq ? w.h_pinned + PIN_K0 : w.h_pinned + 8, seq, c->stream, !(overlapping && a->expand_ov)));
    if (seq) {
        // the result line (kernels.hpp): complete when its first word carries this call's number and its last word
        // agrees with the six in between as read (wait_line: a copy is checked and used)
        uint64_t line[K0_LINE_WORDS];
        int rc = wait_line(c, PIN_K0, seq, line, "K0 did not publish its result");
        if (rc) return rc;
        for (uint32_t i = 1; i < K0_LINE_WORDS - 1; i++) w.h_lines[0][i] = line[i]; // (what the caller unpacks the matches from)
        const uint64_t w1 = line[1]; // matches | too dense << 32 | hash of pin_out << 33
        if (((w1 >> 32) & 1u) == 0) {
            if ((rc = take_more_lines(c, seq, w1 & 0xFFFFFFFFull)) != ACX_OK) return rc;
            *n_out = w1 & 0xFFFFFFFFull;
            *done = true;
            std::lock_guard<std::mutex> lk(a->prof_mu);
            a->profile.small_calls++;
            a->path[7]++;
        }
        return ACX_OK;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (w.h_pinned[9] == 0) {
        *n_out = w.h_pinned[8];
        *done = true;
        std::lock_guard<std::mutex> lk(a->prof_mu);
        a->profile.small_calls++;
        a->path[7]++;
    }
    return ACX_OK;
}

// post_ms of the previous profiled call: ev[1] (end of the scan) .. ev[2] (end of the call's device work)
void settle_post_profile(acx_automaton *a, Ctx *c) {
    if (!c->post_pending) return;
    c->post_pending = false;
    float ms = 0;
    hipEvent_t scan_end = c->ev[(c->ev_pair ^ 1) ? 4 : 1]; // the pair the last launch took
    if (hipEventSynchronize(c->ev[2]) == hipSuccess && hipEventElapsedTime(&ms, scan_end, c->ev[2]) == hipSuccess) {
        std::lock_guard<std::mutex> lk(a->prof_mu);
        a->profile.post_ms += ms;
    }
}

inline hipEvent_t scan_start_ev(Ctx *c) { return c->ev[c->ev_pair ? 3 : 0]; }
inline hipEvent_t scan_stop_ev(Ctx *c) { return c->ev[c->ev_pair ? 4 : 1]; }

// the scan time of the last profiled launch of this context, if it has not been read yet
void settle_scan_profile(acx_automaton *a, Ctx *c) {
    if (!c->scan_pending) return;
    c->scan_pending = false;
    float ms = 0;
    hipEvent_t e0 = c->ev[c->pend_pair ? 3 : 0], e1 = c->ev[c->pend_pair ? 4 : 1];
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(a->prof_mu);
    a->profile.scan_ms += ms;
    a->profile.scan_launches++;
    a->profile.scan_bytes += c->pend_len;
}

// the scan just launched with the current pair of events: its time is read later (the next launch
// of this context takes the other pair)
void add_scan_profile(acx_automaton *a, Ctx *c, uint64_t len, bool timed) {
    if (!timed) return;
    settle_scan_profile(a, c); // (normally settled already, behind this call's own launches)
    c->scan_pending = true;
    c->pend_pair = c->ev_pair;
    c->pend_len = len;
    c->ev_pair ^= 1;
}

// Wait until a kernel has published the line that carries `seq` at pinned word `at` (kernels.hpp, k0_line_check: one
// 64-byte line, one store instruction, [0] seq, [1 .. 6] payload, [7] seq ^ check(payload)) and take a COPY of it: the line
// is complete when its first word carries the number and its last word agrees with the six in between AS READ HERE --
// nothing is read twice, and nothing beside the line is read at all (separate device writes to host memory arrive in no
// particular order).  The wake-up of a blocking stream synchronisation costs 10-20 us; polling costs one PCIe round trip.
// Falls back to the stream after a few milliseconds.
int wait_line(Ctx *c, uint32_t at, uint64_t seq, uint64_t line[8], const char *what) {
    volatile uint64_t *p = c->ws.h_pinned + at;
    auto complete = [&]() -> bool {
        if (p[0] != seq) return false;
        std::atomic_thread_fence(std::memory_order_acquire);
        for (uint32_t i = 1; i < 8; i++) line[i] = p[i];
        return line[7] == (seq ^ k0_line_check(line + 1));
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; !complete(); spins++) {
        cpu_relax();
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(8)) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!complete()) return fail(ACX_EDEVICE, what);
            break;
        }
    }
    line[0] = seq;
    return ACX_OK;
}

// A small call of the host-memory entry point through the context's RESIDENT K0 (Resident above; kernels.hip k0_resident).
// *taken = false: this call is a plain launch (run_small) -- residency is
// switched off, or the loop alternates between kinds of call.  Otherwise as run_small with poll = true.
bool resident_on() {
    static const bool off = std::getenv("ACX_NO_RESIDENT") != nullptr;
    return !off && small_polls();
}
uint64_t env_ticks(const char *name, uint64_t dflt_us) { // microseconds -> ticks of the device's 100 MHz clock
    const char *e = std::getenv(name);
    const uint64_t us = e && *e ? std::strtoull(e, nullptr, 10) : dflt_us;
    return us * 100;
}

int run_resident(acx_automaton *a, Ctx *c, const uint8_t *hay, uint64_t len, int overlapping, int codepoints, uint64_t *n_out,
                 bool *done, bool *taken) {
    *done = false;
    *taken = false;
    Resident &R = c->res;
    if (!resident_on()) return ACX_OK;
    if (R.off) { R.off--; stop_resident(c); return ACX_OK; }
    int rc = ensure_common(c);
    if (rc) return rc;
    Workspace &w = c->ws;
    const DevAutomaton &A = view(a, overlapping != 0);
    const int mode = small_mode(A, (uint32_t)len, !(overlapping && a->expand_ov));
    // (beyond 16 KiB a launch is as good or better -- 60 000 bytes: 28 us launched, 39 through the mailbox, measured; 16 000: 30 and 17)
    if (mode < 0 || len > SMALL_MAX_LEN) { stop_resident(c); return ACX_OK; }
    static const uint64_t idle_ticks = env_ticks("ACX_RESIDENT_IDLE_US", 200), life_ticks = env_ticks("ACX_RESIDENT_LIFE_US", 1000);
    volatile uint64_t *status = w.h_pinned + PIN_RESIDENT;
    const int ov = overlapping ? 1 : 0;
    if (R.live && (R.mode != mode || R.overlapping != ov)) {
        // another kind of call than the kernel was launched for: that one leaves, the next one is launched below
        stop_resident(c);
        if (++R.switches >= 4) { R.switches = 0; R.calls = 0; R.off = 256; return ACX_OK; }
    }
    if (++R.calls >= 64) { R.calls = 0; R.switches = 0; }
    const uint64_t seq = ++c->small_seq;
    const int key_mode = overlapping ? 0 : a->host.match_kind;
    auto launch = [&]() -> int { // (the mailbox holds the call: the kernel takes it as its first)
        if (!R.stream) {
            HIPCHK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
            std::random_device rd;
            R.secret = ((uint64_t)rd() << 32) ^ rd() ^ (uint64_t)(uintptr_t)c;
        }
        R.epoch++;
        R.mode = mode; R.overlapping = ov;
        HIPCHK(launch_resident(d_view(a, overlapping != 0), mode, w.mailbox, key_mode, ov != 0, w.pin_out, w.h_pinned + PIN_K0,
                               w.h_pinned + PIN_RESIDENT, R.epoch, seq - 1, idle_ticks, life_ticks, R.secret, R.delay, R.stream));
        R.live = true;
        std::lock_guard<std::mutex> lk(a->prof_mu);
        a->path[10]++;
        return ACX_OK;
    };
    // the haystack first (acx_find), its check, the word behind them (one aligned store: the kernel takes the bytes that came
    // with the word when the check agrees, and reads the haystack after it has seen the word otherwise)
    if (!R.live || *status == R.epoch) {
        if (R.live) resident_left(c);
        w.mailbox[0] = 0; // (a word of the past -- a quit -- is not for the kernel launched now)
        rc = launch();
        if (rc) return rc;
    }
    // (nothing between the three writes: a poll that reads the mailbox while they are under way fails its check and reads again)
    const uint64_t check = k0_hay_check(hay, (uint32_t)len, seq, R.secret), word = k0_mailbox_word(seq, (uint32_t)len, codepoints != 0, false);
    std::memcpy(w.pin_hay, hay, len);
    std::memset(w.pin_hay + len, 0, (16 - (len & 15)) & 15); // (the check covers whole 16-byte pieces)
    w.mailbox[1] = check;
    __atomic_store_n(w.mailbox, word, __ATOMIC_RELEASE);
    // the result line, as run_small waits for it -- and the kernel's epoch: a kernel that has left (idle, end of its life)
    // has published everything it took before it said so (one release store behind its last line): the line is read once
    // more, and a call the kernel did not take is the first call of the next launch
    volatile uint64_t *p = w.h_pinned + PIN_K0;
    uint64_t line[K0_LINE_WORDS];
    auto complete = [&]() -> bool {
        if (p[0] != seq) return false;
        std::atomic_thread_fence(std::memory_order_acquire);
        for (uint32_t i = 1; i < 8; i++) line[i] = p[i];
        return line[7] == (seq ^ k0_line_check(line + 1));
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; !complete(); spins++) {
        cpu_relax();
        if ((spins & 15) == 15 && *status == R.epoch) {
            std::atomic_thread_fence(std::memory_order_acquire);
            if (complete()) break;
            resident_left(c);
            rc = launch();
            if (rc) return rc;
        }
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(8)) {
            // no answer for 8 ms (a kernel that has not started yet -- its hardware queue may be another context's for a
            // while --, a thread of the host that lost its core): the kernel is told to leave, and when the call is not
            // among what it did, a plain launch answers it
            const uint64_t st0 = *status, word0 = w.mailbox[0], l0 = p[0];
            stop_resident(c);
            HIPCHK(hipStreamSynchronize(R.stream));
            static const bool trace = std::getenv("ACX_RESIDENT_TRACE") != nullptr;
            if (trace)
                std::fprintf(stderr, "acx resident K0: call %llu unanswered for 8 ms (epoch %llu, status %llu, word %llx, line %llu): %s\n",
                             (unsigned long long)seq, (unsigned long long)R.epoch, (unsigned long long)st0, (unsigned long long)word0,
                             (unsigned long long)l0, complete() ? "answered by now" : "a launch takes it");
            if (!complete()) { R.off = 64; return ACX_OK; }
            break;
        }
    }
    *taken = true;
    for (uint32_t i = 1; i < K0_LINE_WORDS - 1; i++) w.h_lines[0][i] = line[i];
    const uint64_t w1 = line[1]; // matches | too dense << 32 | hash of pin_out << 33
    if (((w1 >> 32) & 1u) == 0) {
        if ((rc = take_more_lines(c, seq, w1 & 0xFFFFFFFFull)) != ACX_OK) return rc;
        *n_out = w1 & 0xFFFFFFFFull;
        *done = true;
        std::lock_guard<std::mutex> lk(a->prof_mu);
        a->profile.small_calls++;
        a->path[7]++;
    }
    return ACX_OK;
}

// ---------------------------------------------------------------------------
// The device pipeline.  d_hay: device pointer, len bytes.
//
//   small haystack:          K0, the whole call in one workgroup                 one launch
//   sparse output (default): scan (K1b: prefix hits / K1a: occurrences) into per-tile hit slots ->
//                            k_tile_main (verify, order, match kind) -> k_tile_write (output offsets, final records);
//                            the host returns as soon as the scan kernel has published the totals
//   dense output:            scan emits into regions -> (walk) -> compact -> radix sort -> spans ->
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
    Ctx *c;
    const uint8_t *d_hay;
    uint64_t len;
    const Segments &G;
    bool overlapping, codepoints, segmented;
    acx_result *r;
    int key_mode;
    bool pre;           // K1b (else K1a)
    uint32_t scan_grid; // workgroups of the scan kernel
    uint32_t lead;      // d_hay & 15: index = stream position + lead
    uint64_t tiles;     // 4 KiB tiles of index space
    // results
    uint64_t n_raw = 0, n_final = 0, n_hits = 0;
    bool exact_regions = false; // dense path, second pass: regions at the exclusive prefix of the first pass's counts
    bool chunked_walk = false;  // dense path, K1a: the failureless walk ran out of item room, walk in chunks
    bool no_dense_tiles = false; // dense path: the tile-ordered form gave up on this call (the radix-sort form takes it)
    bool ovf_grown = false;      // sparse path: the overflow list was grown for this call (one more attempt)
    bool wide_tried = false;     // sparse path: the call was repeated with the wide form of the post stage
    bool host_result = false;    // the caller reads the matches on the host right away (acx_find): the sparse path writes them
                                 // to the context's pinned buffer when its capacity fits (PIN_FINAL_MAX), no copy kernel-side
    acx_match_t *out = nullptr;  // sparse path: where the write kernels put the records (w.final, or w.pin_final)
    uint64_t out_cap = 0;
    bool counts_zeroed = false; // batch: the per-haystack counts are zero or being accumulated into
    uint64_t exact_total = 0;
    bool timed = false;         // this call carries the profiling events (every prof_every-th call of a context)
    bool early_event = false;   // the caller returns before the device work is done: fence it with r->done
    bool event_at_post = false; // r->done was recorded right behind the post kernels
    bool leads_counted = false; // the scan has written the lead-byte counts of every 64 bytes (str API)
    bool cp_done = false;       // the write kernel has already converted the offsets to code points
    bool queued = false;    // work queued on the stream that nobody waited for yet
    bool localized = false; // batch: offsets are already local and the counts taken
};
enum class Attempt { Done, GoDense, Again };

// ---- sparse output, groups the tile kernels could not finish (a dense stretch of the input: more hits than a tile's
// slots, a full bucket, more matches than a group's stretch, an uncertifiable chain): the HOT pipeline -- the hot
// groups' hits (slots + overflow list) through the tile-ordered dense machinery, their counts credited to the sparse
// path's groups, then the write kernel again.  One dense region costs the groups it lies in, not the call (it used to
// send the whole call to the dense path and keep the handle there for eight more calls).  *lost: the hot pipeline gave
// up too (a bucket of more than DT_SLOTS occurrences, a chain longer than the context) -- the radix-sort form takes the call.
constexpr uint64_t PIN_FINAL_MAX = 32ull << 20; // bytes of pinned result buffer a context keeps for host calls (acx_find up to ~8 MiB)
// hot groups whose capacity the output buffer has room for anyway: 16 to start with, up to 128 for a context that has seen more
// (Ctx::hot_inline; round 6 -- until then 128 for every call: ~100 MB per result from 32 MiB haystacks on, whatever the input);
// beyond a context's figure: hot_totals, then a buffer of the exact size
constexpr uint64_t HOT_INLINE = 16, HOT_INLINE_MAX = 128;
int run_hot(FindCall &c, uint32_t *abort_flag, uint64_t seq, uint32_t n_hot, uint32_t ovf_max, uint64_t *seg_counts,
            const uint64_t *cp_pre, bool counts_clear, bool *lost) {
    acx_automaton *a = c.a;
    Ctx *x = c.c;
    Workspace &w = x->ws;
    hipStream_t st = x->stream;
    TileSpace &T = w.T;
    *lost = false;
    if (ensure_dense_tiles(x, c.tiles) != ACX_OK) { // (no room for the buckets: the radix-sort form needs less)
        (void)hipGetLastError();
        free_dense_tiles(w);
        c.no_dense_tiles = true;
        *lost = true;
        return ACX_OK;
    }
    uint32_t *hot_abort = (uint32_t *)(w.summary + 10);
    // (the bucket counters and the pipeline's abort flag, summary[10]: cleared by the write kernel that announced the hot
    // groups -- unless the buckets are allocated by this very call)
    if (!counts_clear) HIPCHK_RC(hipMemsetAsync(w.dt.counts, 0, ((uint64_t)w.dt.n_tiles + 1) * 4, st));
    HIPCHK_RC(hot_verify_main(view(a, c.overlapping), c.key_mode, c.overlapping, c.G, T, w.hot_list, n_hot, abort_flag, ovf_max,
                              w.dt, w.TD, c.lead, c.d_hay, c.len, hot_abort, seq, 0, st));
    // the output's room: the groups' capacities bound the matches; with many hot groups the buffer is sized exactly
    // instead (one more round trip, next to that much hot work)
    const uint64_t bound = ((uint64_t)T.n_groups - n_hot) * T.gmax + (uint64_t)n_hot * HOT_SUB * DT_GMAX;
    if (bound > c.out_cap && c.out != w.final) { // (the pinned buffer is not regrown: the dense path takes this call)
        *lost = true;
        return ACX_OK;
    }
    if (bound > w.final_cap && c.out == w.final) {
        const uint64_t pub_t = seq | (1ull << 62);
        HIPCHK_RC(hot_totals(T, seq, w.h_pinned + PIN_HOT_TOTALS, pub_t, st));
        uint64_t early[8];
        int rc = wait_line(x, PIN_HOT_TOTALS, pub_t, early, "the hot pipeline did not publish its total");
        if (rc) return rc;
        const uint64_t n = std::min<uint64_t>(early[1], bound); // (meaningless when the pipeline gave up: bounded all the same)
        if (n >= occ_limit()) {
            // (the call goes on in byte ranges on this context: the control blocks and both sets of supergroup words clear again)
            w.flags_dirty = true;
            HIPCHK_RC(hipStreamSynchronize(st));
            HIPCHK_RC(hipMemsetAsync(T.sgw, 0, 4 * (uint64_t)T.sg_cap * 8, st));
            return fail_occ();
        }
        if (n > w.final_cap) {
            HIPCHK_RC(hipStreamSynchronize(st));
            g_bufs.put(w.final, a->device);
            w.final = nullptr; w.final_cap = 0;
            HIPCHK_RC(g_bufs.get((void **)&w.final, n * sizeof(acx_match_t), a->device));
            w.final_cap = n;
            c.out = w.final; c.out_cap = n;
        }
    }
    // an input that is dense (nearly) everywhere: the dense path proper takes the handle's next calls -- its scan writes
    // the hits where its verification reads them, no sparse attempt in front
    if ((uint64_t)n_hot * 4 > T.n_groups && T.n_groups >= 8) { x->dense_hold = 8; x->hold_dense_input = true; }
    const uint64_t pub = seq | (1ull << 63);
    HIPCHK_RC(hot_write(view(a, c.overlapping), c.key_mode, T, w.TD, w.hot_list, n_hot, c.lead, c.d_hay, c.out, w.summary,
                        abort_flag, hot_abort, w.h_pinned + PIN_TOTALS, seq, pub, c.G, seg_counts, cp_pre, w.blocksub, 0, st));
    if (c.early_event && c.r->done) HIPCHK_RC(hipEventRecord(c.r->done, st)); // (again: behind the kernels queued since)
    int rc = wait_line(x, PIN_TOTALS, pub, w.t_line, "the write kernel did not publish its totals");
    if (rc) return rc;
    if ((w.t_line[4] & 0xFF) != 0) {
        c.no_dense_tiles = true;
        *lost = true;
    }
    return ACX_OK;
}

// ---- sparse output: hit slots + tile kernels; returns when the totals are known
int attempt_sparse(FindCall &c, Attempt *what) {
    acx_automaton *a = c.a;
    Ctx *x = c.c;
    Workspace &w = x->ws;
    hipStream_t st = x->stream;
    // (the wide form of the post stage: K1b's hits only -- the hot pipeline behind it is theirs)
    static const bool force_wide = std::getenv("ACX_FORCE_WIDE") != nullptr; // tests: every K1b call in the wide form
    const bool wide = (x->wide || force_wide) && c.pre;
    const uint32_t gmax = wide ? GROUP_MAX_WIDE : GROUP_MAX;
    int rc = ensure_tiles(a, x, c.tiles, gmax);
    if (rc) return rc;
    TileSpace &T = w.T;
    // automata of at most 32 byte classes: the failureless walk (k1a_scan + k1a_walk) instead of the
    // chunked one (ACX_NO_PFAC: always the chunked walk -- measurements)
    static const bool no_pfac = std::getenv("ACX_NO_PFAC") != nullptr;
    const bool pfac = !c.pre && pfac_available(a->dev, a->max_lds) && !no_pfac;
    const uint32_t pgrid = pfac ? pfac_scan_grid(c.d_hay, c.len, a->n_cus) : 0;
    // hit counts: contiguous per wave of the scan (K1b, k1a_scan); the chunked walk: plain per-tile
    // arrival counters
    T.cnt_nw = c.pre ? c.scan_grid * 16 : pfac ? pgrid * 16 : 1;
    T.cnt_iters = T.cnt_nw > 1 ? (uint32_t)((c.tiles + T.cnt_nw - 1) / T.cnt_nw) : (uint32_t)c.tiles;
    // (room for every group's capacity + what a few hot groups can report beyond it: run_hot)
    // (a host call's records go to the CONTEXT's pinned buffer, which is not regrown in the middle of a call: there the room
    // for hot groups is the full figure while the whole fits PIN_FINAL_MAX; a device result's buffer is the caller's to hold:
    // the context's own figure)
    auto cap_for = [&](uint64_t hot_room) {
        return (uint64_t)T.n_groups * gmax + (c.pre ? std::min<uint64_t>(T.n_groups, hot_room) * HOT_SUB * DT_GMAX : 0);
    };
    const bool pin = c.host_result && !c.segmented && cap_for(HOT_INLINE_MAX) * sizeof(acx_match_t) <= PIN_FINAL_MAX &&
                     !(c.overlapping && a->expand_ov);
    const uint64_t out_cap = cap_for(pin ? HOT_INLINE_MAX : x->hot_inline);
    if (pin && w.pin_final_cap < out_cap) {
        HIPCHK_RC(hipStreamSynchronize(st));
        if (w.pin_final) (void)hipHostFree(w.pin_final);
        w.pin_final = nullptr; w.pin_final_cap = 0;
        HIPCHK_RC(hipHostMalloc((void **)&w.pin_final, out_cap * sizeof(acx_match_t), hipHostMallocDefault));
        w.pin_final_cap = out_cap;
    }
    if (pin) {
        c.out = w.pin_final; c.out_cap = w.pin_final_cap;
    } else {
        if (w.final && w.final_cap < out_cap) { g_bufs.put(w.final, a->device); w.final = nullptr; }
        if (!w.final) {
            HIPCHK_RC(g_bufs.get((void **)&w.final, out_cap * sizeof(acx_match_t), a->device));
            w.final_cap = out_cap;
        }
        c.out = w.final; c.out_cap = w.final_cap;
    }
    if (w.flags_dirty) {
        HIPCHK_RC(hipMemsetAsync(w.ctl, 0, 12, st));
        HIPCHK_RC(hipMemsetAsync(w.ctl + CTL_WORDS, 0, 12, st));
        HIPCHK_RC(hipMemsetAsync(w.ovf_counts, 0, 2 * OVF_LISTS * OVF_COUNT_STRIDE * 4, st));
    }
    w.flags_dirty = true;
    // two control blocks used in turn: this attempt's write kernel clears the other one
    uint32_t *abort_flag = w.ctl + CTL_WORDS * x->flag_idx;
    uint32_t *next_flag = w.ctl + CTL_WORDS * (x->flag_idx ^ 1);
    x->flag_idx ^= 1;
    const Sink K{nullptr, nullptr, 0, c.key_mode, T.hslots, T.hcnt, abort_flag, c.lead, T.cnt_nw, T.cnt_iters};
    // batch with byte offsets: the write kernel localises and counts per haystack itself
    uint64_t *seg_counts = c.segmented && !c.codepoints ? c.r->d_counts : nullptr;
    const bool prof = c.timed;
    hipEvent_t side_after = nullptr;
    if (c.pre) {
        // str API: the scan counts the UTF-8 lead bytes on its way (aligned haystacks: the blocks of
        // the code-point prefix are then the rows of the scan's tiles)
        uint8_t *cp_sub = nullptr;
        if (c.codepoints && c.lead == 0) {
            if ((rc = ensure_blocks(x, 4 * c.tiles + 1)) != ACX_OK) return rc;
            cp_sub = w.blocksub;
        }
        // measurement: the event pair rides on the dispatch
        // (str API, one haystack: the code-point prefix runs on the second stream as soon as the scan
        // is done -- the event it waits for rides on the scan's own dispatch, no packet in between)
        side_after = cp_sub && !c.segmented ? (prof ? scan_stop_ev(x) : x->fork_ev) : nullptr;
        g_trace.mark(2);
        HIPCHK_RC(launch_prefilter(a->dev, K, c.d_hay, c.len, c.scan_grid, st, prof ? scan_start_ev(x) : nullptr,
                                   prof ? scan_stop_ev(x) : side_after, cp_sub));
        g_trace.mark(3);
        c.leads_counted = cp_sub != nullptr;
    } else {
        // the failureless walk: the scan writes the hits it settles itself and every tile's count, the
        // walk appends to them; its survivor records take the place of K1b's dense-path hit sink.  More
        // survivors than their regions hold (1 per 16 haystack bytes): the walk raises the abort flag, the
        // call is redone on the dense path, which walks in chunks.
        if (!pfac) HIPCHK_RC(hipMemsetAsync(T.hcnt, 0, (c.tiles + 1) * 4, st)); // arrival counters of the walk's emission
        uint64_t surv_total = 0;
        if (pfac) {
            surv_total = pfac_workspace_words(c.len, pgrid, false);
            if ((rc = ensure_hits(x, (surv_total + 3) / 4)) != ACX_OK) return rc; // (records of 32 B there, u64 words here)
        }
        if (prof) HIPCHK_RC(hipEventRecord(scan_start_ev(x), st));
        if (pfac)
            HIPCHK_RC(launch_pfac(view(a, c.overlapping), d_view(a, c.overlapping), c.G, K, c.d_hay, c.len, pgrid, (uint64_t *)w.hrecs, w.hit_counts,
                                  pgrid * 16, false, st));
        else
            HIPCHK_RC(launch_dfa_walk(view(a, c.overlapping), d_view(a, c.overlapping), c.G, K, c.d_hay, c.len, c.scan_grid, a->max_lds, st));
        if (prof) HIPCHK_RC(hipEventRecord(scan_stop_ev(x), st));
    }
    // str API, one haystack: the prefix of the lead-byte counts is ready before the write kernel
    // needs it (it depends on the scan only), so the write kernel converts on the way out
    // (three small latency-bound kernels, ~30 us: on the context's second stream, beside k_tile_main,
    // which does not need them; the write kernel waits for both)
    const uint64_t *cp_pre = nullptr;
    hipEvent_t before_write = nullptr;
    if (c.leads_counted && !c.segmented) {
        const uint64_t nb1 = (c.len + 1023) / 1024 + 1;
        hipStream_t side = x->copy_stream;
        if (!side_after) { side_after = x->fork_ev; HIPCHK_RC(hipEventRecord(x->fork_ev, st)); }
        HIPCHK_RC(hipStreamWaitEvent(side, side_after, 0));
        HIPCHK_RC(block_prefix(w.blocksub, w.blockcnt, w.blockpre, nb1 - 1, w.temp, w.temp_bytes, side));
        HIPCHK_RC(hipEventRecord(x->join_ev, side));
        before_write = x->join_ev;
        cp_pre = w.blockpre;
    }
    const uint64_t seq = ++x->seq;
    // (the hot pipeline's bucket counters, when a call of this context has allocated them: the write kernel clears them
    // when it announces hot groups)
    uint32_t *hot_counts = c.pre && w.dt.counts && c.tiles + 1 <= w.dt_cap ? w.dt.counts : nullptr;
    // The hot pipeline queued ahead of the knowledge that the call needs it (round 6): the context's last call had a few hot
    // groups and the buckets are allocated -- grids for spec_bound hot groups, their number read on the device (kernels.hip:
    // hot_groups_here); more of them, or none: the kernels return at once.  (Not beyond the room the output buffer has for
    // hot groups, nor where a call with that many would rather take the wide form.)
    static const bool no_spec = std::getenv("ACX_NO_SPEC_HOT") != nullptr; // measurements
    uint32_t spec_bound = 0;
    if (hot_counts && x->spec_hot && !no_spec) {
        uint64_t b = std::min<uint64_t>(2ull * x->spec_hot, HOT_INLINE_MAX);
        b = std::min<uint64_t>(b, pin ? HOT_INLINE_MAX : x->hot_inline);
        b = std::min<uint64_t>(b, T.n_groups >= 32 ? T.n_groups / 32 : T.n_groups);
        if (b >= x->spec_hot && ensure_dense_tiles(x, c.tiles) == ACX_OK) spec_bound = (uint32_t)b;
    }
    T.w8 = tile_words_narrow(view(a, c.overlapping), cp_pre != nullptr) ? 1u : 0u; // (what the groups' stretches hold: kernels.hpp)
    HIPCHK_RC(tile_post(view(a, c.overlapping), c.key_mode, c.overlapping, T, c.lead, c.d_hay, c.len, c.out, w.summary, abort_flag,
                        next_flag, w.h_pinned + PIN_TOTALS, seq, c.G, seg_counts, cp_pre, w.blocksub, before_write, c.pre, hot_counts,
                        (uint32_t)(c.tiles + 2), st));
    const uint64_t pub_spec = seq | (1ull << 63);
    if (spec_bound) {
        uint32_t *hot_abort = (uint32_t *)(w.summary + 10);
        HIPCHK_RC(hot_verify_main(view(a, c.overlapping), c.key_mode, c.overlapping, c.G, T, w.hot_list, 0, abort_flag, x->spec_ovf, w.dt,
                                  w.TD, c.lead, c.d_hay, c.len, hot_abort, seq, spec_bound, st));
        HIPCHK_RC(hot_write(view(a, c.overlapping), c.key_mode, T, w.TD, w.hot_list, 0, c.lead, c.d_hay, c.out, w.summary, abort_flag,
                            hot_abort, w.h_pinned + PIN_SPEC_TOTALS, seq, pub_spec, c.G, seg_counts, cp_pre, w.blocksub, spec_bound, st));
    }
    g_trace.mark(4);
    if (seg_counts) c.counts_zeroed = true; // (k_tile_main clears them, k_tile_write adds to them)
    // while the kernels run: the scan time of the previous call, and the event the result's
    // accessors wait for (nothing more is queued behind the write kernel unless a fix-up follows)
    settle_scan_profile(a, x);
    if (c.early_event && !c.r->done) {
        c.r->done = g_events.get(a->device);
        if (c.r->done) HIPCHK_RC(hipEventRecord(c.r->done, st));
        c.event_at_post = c.r->done != nullptr;
    }
    g_trace.mark(5);
    if ((rc = wait_line(x, PIN_TOTALS, seq, w.t_line, "the write kernel did not publish its totals")) != ACX_OK) return rc;
    g_trace.mark(6);
    w.flags_dirty = false; // the write kernel left the next control block clean
    add_scan_profile(a, x, c.len, c.timed);
    // the line (k_tile_write): [1] matches, [2] occurrences, [3] prefix hits, [4] why | hot groups << 8, [5] overflow hits |
    // the fullest overflow list << 32
    uint64_t gave_up = w.t_line[4] & 0xFF;
    const uint64_t n_hot = w.t_line[4] >> 8, n_ovf = w.t_line[5] & 0xFFFFFFFFull, ovf_max = w.t_line[5] >> 32;
    if (gave_up == 2 && !c.ovf_grown && ovf_max * OVF_LISTS <= 3 * c.tiles * HIT_SLOTS + (OVF_LISTS << 12)) {
        // K1b found more hits beyond their tiles' slots than an overflow list holds (nothing else is wrong): with lists
        // of the size this input needs the sparse kernels + the hot pipeline take it -- again, once
        HIPCHK_RC(hipStreamSynchronize(st));
        HIPCHK_RC(hipMemsetAsync(T.sgw, 0, 4 * (uint64_t)T.sg_cap * 8, st));
        if ((rc = set_overflow_room(x, ovf_max + ovf_max / 4 + 64)) == ACX_OK) {
            c.ovf_grown = true;
            c.leads_counted = false;
            c.event_at_post = false;
            *what = Attempt::Again;
            return ACX_OK;
        }
        (void)hipGetLastError(); // (no room for it: the dense path)
    }
    // what the context's next call queues ahead: the hot pipeline, when this one had a few hot groups
    x->spec_hot = (!gave_up && n_hot && n_hot <= HOT_INLINE_MAX) ? (uint32_t)n_hot : 0u;
    x->spec_ovf = (uint32_t)ovf_max;
    bool spec_done = false;
    if (spec_bound && !gave_up && n_hot && n_hot <= spec_bound) {
        // the speculative pipeline is this call's: its pass 1 publishes the totals (a line of its own: pass 0's stays readable)
        if ((rc = wait_line(x, PIN_SPEC_TOTALS, pub_spec, w.t_line, "the speculative hot pipeline did not publish its totals")) != ACX_OK) return rc;
        if ((w.t_line[4] & 0xFF) != 0) { c.no_dense_tiles = true; gave_up = 1; }
        else { a->path[1]++; a->path[2] += n_hot; a->path[3] += n_ovf; }
        spec_done = true;
    }
    if (c.ovf_grown && gave_up != 2) a->path[6]++;
    // many groups gave up on the narrow stage although their tiles' slots held the hits (a match every 100 - 500 bytes: more
    // than 24 occurrences in a 4 KiB bucket, more than GROUP_MAX in a group): the context takes the WIDE form of the post stage
    // -- this call again, its next calls from the start -- instead of handing every group to the hot pipeline and the handle
    // to the dense path (until round 5: 2 437 -> 1 057 GB/s between a match every 512 and every 256 bytes)
    static const bool no_wide = std::getenv("ACX_NO_WIDE") != nullptr; // measurements
    if (!gave_up && c.pre && !wide && !c.wide_tried && !no_wide && T.n_groups >= 32 && n_hot * 32 > T.n_groups &&
        n_ovf * 8 <= w.t_line[3]) {
        HIPCHK_RC(hipStreamSynchronize(st));
        HIPCHK_RC(hipMemsetAsync(T.sgw, 0, 4 * (uint64_t)T.sg_cap * 8, st));
        if (seg_counts) HIPCHK_RC(hipMemsetAsync(c.r->d_counts, 0, std::max<uint64_t>(c.G.n_hay, 1) * 8, st));
        x->wide = true;
        c.wide_tried = true;
        c.leads_counted = false;
        c.event_at_post = false;
        a->path[9]++;
        *what = Attempt::Again;
        return ACX_OK;
    }
    if (!gave_up && n_hot && !spec_done) { // groups the sparse kernels could not finish: the hot pipeline, then the write kernel again
        bool lost = false;
        if ((rc = run_hot(c, abort_flag, seq, (uint32_t)n_hot, (uint32_t)ovf_max, seg_counts, cp_pre, hot_counts != nullptr, &lost)) != ACX_OK) return rc;
        if (lost) gave_up = 1;
        else {
            a->path[1]++; a->path[2] += n_hot; a->path[3] += n_ovf;
            if (n_hot > x->hot_inline) x->hot_inline = std::min<uint64_t>(HOT_INLINE_MAX, 2 * n_hot); // (the context's next calls)
        }
    } else if (!gave_up && !spec_done) {
        a->path[0]++;
    }
    if (gave_up != 0) { // the slots could not hold the output: dense path
        HIPCHK_RC(hipStreamSynchronize(st));
        c.event_at_post = false; // (the dense path queues more: the event is recorded again at the end)
        // (both sets of supergroup words clear again, whatever made the call give up)
        HIPCHK_RC(hipMemsetAsync(T.sgw, 0, 4 * (uint64_t)T.sg_cap * 8, st));
        if (seg_counts) HIPCHK_RC(hipMemsetAsync(c.r->d_counts, 0, std::max<uint64_t>(c.G.n_hay, 1) * 8, st));
        x->dense_hold = 8;
        x->hold_dense_input = false; // (unless the dense path finds the input dense: below)
        c.leads_counted = false;
        c.cp_done = false;
        *what = Attempt::GoDense;
        return ACX_OK;
    }
    c.n_raw = w.t_line[2]; // (the hot pipeline's second publication when it ran)
    c.n_hits = w.t_line[3];
    c.n_final = w.t_line[1];
    // (back to the narrow form -- twice the groups in flight -- when the input no longer needs the wide one)
    if (wide && n_hot == 0 && c.n_raw * 5 < (uint64_t)T.n_groups * GROUP_MAX * 2) x->wide = false;
    if (c.out == w.final) {
        c.r->d_matches = w.final; // hand the buffer over; the next call takes a fresh one
        w.final = nullptr;
    } else {
        c.r->d_matches = c.out;   // the context's pinned buffer: the caller (acx_find) copies out of it under its lease
        c.r->borrowed = true;
    }
    c.localized = seg_counts != nullptr;
    c.cp_done = cp_pre != nullptr;
    c.queued = !c.event_at_post; // k_tile_write is still running (and the event that fences it is in place)
    *what = Attempt::Done;
    return ACX_OK;
}

int zero_counts(FindCall &c);

// ---- dense output, tile-ordered (K1b sets whose patterns fit the context tiles): prefix hits in per-wave regions ->
// occurrence words in the bucket of their key tile -> per group: sort + match kind in LDS -> the sparse path's write
// kernel.  One round trip for the totals (the output buffer is sized exactly), a second pass only when the hit
// regions were too small.  Gives up (Attempt::Again with no_dense_tiles) when a bucket overflows -- more than one
// occurrence per 8 bytes -- or a chain of overlapping occurrences is longer than the context.
int attempt_dense_tiles(FindCall &c, Attempt *what) {
    acx_automaton *a = c.a;
    Ctx *x = c.c;
    Workspace &w = x->ws;
    hipStream_t st = x->stream;
    int rc = ensure_hits(x, std::max<uint64_t>(1u << 16, c.len / 64));
    if (rc) return rc;
    if ((rc = ensure_dense_tiles(x, c.tiles)) != ACX_OK) return rc;
    const uint32_t hit_grid = prefilter_hit_regions(c.scan_grid);
    const uint64_t hit_cap = w.hit_total / hit_grid;
    const Sink H{w.hrecs, w.hit_counts, hit_cap, c.key_mode, nullptr, nullptr, nullptr, c.lead, 1, 0};
    uint32_t *abort_flag = (uint32_t *)(w.summary + 10), *zero_flag = (uint32_t *)(w.summary + 11);
    HIPCHK_RC(hipMemsetAsync(w.dt.counts, 0, ((uint64_t)w.dt.n_tiles + 1) * 4, st));
    HIPCHK_RC(hipMemsetAsync(w.TD.sgw, 0, 2 * (uint64_t)w.TD.sg_cap * 8, st));
    HIPCHK_RC(hipMemsetAsync(w.summary + 10, 0, 16, st));
    const bool prof = c.timed;
    HIPCHK_RC(launch_prefilter(a->dev, H, c.d_hay, c.len, c.scan_grid, st, prof ? scan_start_ev(x) : nullptr,
                               prof ? scan_stop_ev(x) : nullptr));
    HIPCHK_RC(dense_tiles_verify(view(a, c.overlapping), c.G, H, hit_grid, w.dt, c.key_mode, c.lead, c.d_hay, c.len, abort_flag, st));
    // (the hit regions' fill: summary[2] = hits kept, [3] = the fullest region)
    HIPCHK_RC(sink_summary(w.hit_counts, hit_grid, hit_cap, w.hit_counts, hit_grid, hit_cap, w.summary, w.region_off, st));
    // (k_dense_main in its compact form -- sixteen groups per CU -- unless a call of this context did not fit it lately)
    static const bool no_compact = std::getenv("ACX_NO_DENSE_COMPACT") != nullptr; // measurements
    bool compact = x->dense_full == 0 && !no_compact;
    if (x->dense_full > 0) x->dense_full--;
    HIPCHK_RC(dense_tiles_main(a->dev, c.key_mode, c.overlapping, w.dt, w.TD, c.lead, abort_flag, w.summary, compact, st));
    // ([0..3]: the hit regions' fill; [8] matches, [9] occurrences, [10] the abort flag -- not [7], [11]: the words the
    // sparse path and K0 publish their sequence numbers in)
    HIPCHK_RC(hipMemcpyAsync(w.h_pinned, w.summary, 32, hipMemcpyDeviceToHost, st));
    HIPCHK_RC(hipMemcpyAsync(w.h_pinned + 8, w.summary + 8, 24, hipMemcpyDeviceToHost, st));
    HIPCHK_RC(hipStreamSynchronize(st));
    add_scan_profile(a, x, c.len, c.timed);
    const uint64_t hit_max = w.h_pinned[3];
    if (hit_max > hit_cap) { // hits were dropped: more room, again (the regions are balanced: a wave's tiles are spread over the stream)
        if ((rc = ensure_hits(x, (uint64_t)hit_grid * (hit_max + hit_max / 8 + 64))) != ACX_OK) return rc;
        *what = Attempt::Again;
        return ACX_OK;
    }
    if (compact && (uint32_t)w.h_pinned[10] == 2) { // a group's occurrences did not fit the compact stage: the kernel again, full
        x->dense_full = 8;
        HIPCHK_RC(hipMemsetAsync(w.TD.sgw, 0, 2 * (uint64_t)w.TD.sg_cap * 8, st));
        HIPCHK_RC(hipMemsetAsync(w.summary + 10, 0, 16, st));
        HIPCHK_RC(dense_tiles_main(a->dev, c.key_mode, c.overlapping, w.dt, w.TD, c.lead, abort_flag, w.summary, false, st));
        HIPCHK_RC(hipMemcpyAsync(w.h_pinned + 8, w.summary + 8, 24, hipMemcpyDeviceToHost, st));
        HIPCHK_RC(hipStreamSynchronize(st));
    }
    if ((uint32_t)w.h_pinned[10] != 0) { // a bucket overflowed / a chain left its context: the radix-sort form
        c.no_dense_tiles = true;
        *what = Attempt::Again;
        return ACX_OK;
    }
    const uint64_t n_final = w.h_pinned[8], n_raw = w.h_pinned[9];
    if (std::max(n_raw, w.h_pinned[2]) >= occ_limit()) return fail_occ(); // (the tiles' counts and their prefixes are 32 bits wide)
    // (round 5: a hold that a dense INPUT set ends with the first input that is not dense -- the dense path on a sparse input
    // costs 2-3x, eight calls of it were the price of one dense call in front: bench.py's 8 GiB run behind its density sweep)
    if (n_raw > 8 * c.tiles) { x->dense_hold = 8; x->hold_dense_input = true; }
    else if (x->dense_hold > 0) x->dense_hold = x->hold_dense_input ? 0 : x->dense_hold - 1;
    c.n_raw = n_raw;
    c.n_hits = w.h_pinned[2];
    c.n_final = n_final;
    *what = Attempt::Done;
    a->path[4]++;
    if (n_final == 0) return ACX_OK;
    HIPCHK_RC(g_bufs.get((void **)&c.r->d_matches, n_final * sizeof(acx_match_t), a->device));
    // batch with byte offsets: the write kernel localises and counts per haystack itself
    uint64_t *seg_counts = c.segmented && !c.codepoints ? c.r->d_counts : nullptr;
    if (seg_counts && (rc = zero_counts(c)) != ACX_OK) return rc;
    HIPCHK_RC(dense_tiles_write(a->dev, c.key_mode, w.TD, c.d_hay, c.r->d_matches, w.summary, zero_flag, w.h_pinned + PIN_TOTALS, c.lead,
                                c.G, seg_counts, nullptr, nullptr, st));
    c.localized = seg_counts != nullptr;
    c.queued = true;
    return ACX_OK;
}

// ---- dense output: region mode -> compact -> radix sort -> resolve (two round trips)
int attempt_dense(FindCall &c, Attempt *what) {
    acx_automaton *a = c.a;
    Ctx *x = c.c;
    Workspace &w = x->ws;
    hipStream_t st = x->stream;
    {
        const bool no_tiles_env = std::getenv("ACX_NO_DENSE_TILES") != nullptr; // tests / measurements: the radix-sort form (read per call)
        if (c.pre && a->sparse_ok && !c.no_dense_tiles && !no_tiles_env && c.tiles < (1ull << 26))
            return attempt_dense_tiles(c, what);
    }
    int rc = ensure_occ_capacity(x, std::max<uint64_t>(1u << 16, c.len / 64));
    if (rc) return rc;
    if (c.pre && (rc = ensure_hits(x, std::max<uint64_t>(1u << 16, c.len / 64))) != ACX_OK) return rc;
    const uint32_t hit_grid = c.pre ? prefilter_hit_regions(c.scan_grid) : 0;
    // K1a: the failureless walk here too (one occurrence region per block of k1a_walk); the chunked walk
    // when the automaton has none, or when its items did not fit
    static const bool no_pfac = std::getenv("ACX_NO_PFAC") != nullptr;
    const bool pfac = !c.pre && pfac_available(a->dev, a->max_lds) && !no_pfac && !c.chunked_walk;
    const uint32_t pgrid = pfac ? pfac_scan_grid(c.d_hay, c.len, a->n_cus) : 0;
    if (pfac && (rc = ensure_hits(x, (pfac_workspace_words(c.len, pgrid, true) + 3) / 4)) != ACX_OK) return rc;
    const uint32_t grid = c.pre ? walk_hits_grid(hit_grid) : pfac ? pgrid * 16 : c.scan_grid; // occurrence regions
    const uint64_t hit_cap = c.pre ? w.hit_total / hit_grid : 0;
    // exact_regions (second pass after an occurrence region overflowed): every region gets the room
    // it asked for in the first pass, at the exclusive prefix of the counts (stored behind the counts)
    const uint64_t region_cap = c.exact_regions ? 0 : w.cap / grid;
    const Sink H{w.hrecs, w.hit_counts, hit_cap, c.key_mode, nullptr, nullptr, nullptr, c.lead, 1, 0};
    uint32_t *items_overflow = (uint32_t *)(w.summary + 4);
    const Sink K{w.recs, w.block_counts, region_cap, c.key_mode, nullptr, nullptr, pfac ? items_overflow : nullptr, c.lead, 1, 0};
    const bool prof = c.timed;
    if (c.pre) {
        HIPCHK_RC(launch_prefilter(a->dev, H, c.d_hay, c.len, c.scan_grid, st, prof ? scan_start_ev(x) : nullptr,
                                   prof ? scan_stop_ev(x) : nullptr));
        HIPCHK_RC(launch_walk_hits(view(a, c.overlapping), c.G, H, hit_grid, K, grid, c.d_hay, c.len, st));
    } else {
        if (pfac) HIPCHK_RC(hipMemsetAsync(w.summary + 4, 0, 8, st));
        if (prof) HIPCHK_RC(hipEventRecord(scan_start_ev(x), st));
        if (pfac)
            HIPCHK_RC(launch_pfac(view(a, c.overlapping), d_view(a, c.overlapping), c.G, K, c.d_hay, c.len, pgrid, (uint64_t *)w.hrecs, w.hit_counts,
                                  grid, true, st));
        else
            HIPCHK_RC(launch_dfa_walk(view(a, c.overlapping), d_view(a, c.overlapping), c.G, K, c.d_hay, c.len, c.scan_grid, a->max_lds, st));
        if (prof) HIPCHK_RC(hipEventRecord(scan_stop_ev(x), st));
    }
    HIPCHK_RC(sink_summary(w.block_counts, grid, c.exact_regions ? ~0ull : region_cap, c.pre ? w.hit_counts : nullptr,
                           hit_grid, hit_cap, w.summary, w.region_off, st));
    HIPCHK_RC(hipMemcpyAsync(w.h_pinned, w.summary, 40, hipMemcpyDeviceToHost, st));
    HIPCHK_RC(hipStreamSynchronize(st));
    add_scan_profile(a, x, c.len, c.timed);
    if (pfac && (uint32_t)w.h_pinned[4] != 0) { // more items than 1 per 16 bytes: the chunked walk has no such limit
        c.chunked_walk = true;
        c.exact_regions = false;
        *what = Attempt::Again;
        return ACX_OK;
    }
    const uint64_t n_raw = w.h_pinned[0], region_max = w.h_pinned[1], hit_max = c.pre ? w.h_pinned[3] : 0;
    if (c.exact_regions) {
        if (n_raw != c.exact_total || (c.pre && hit_max > hit_cap))
            return fail(ACX_EDEVICE, "the second pass of the dense path counted differently");
    } else if (region_max > region_cap || hit_max > hit_cap) { // a sink region overflowed: grow, redo
        if (hit_max > hit_cap) {
            // hits that overflowed were dropped, so the occurrence counts are lower bounds: more room for
            // both, uniform regions (the hit regions are balanced: a wave's tiles are spread over the stream)
            if ((rc = ensure_hits(x, (uint64_t)hit_grid * (hit_max + hit_max / 8 + 64))) != ACX_OK) return rc;
            uint64_t want = std::max((uint64_t)grid * (region_max + region_max / 8 + 64), w.cap * 4);
            // (bounded growth: ~72 B of workspace per record; once the hit regions hold everything the
            // counts are exact and the second pass sizes the occurrence buffer exactly)
            want = std::min<uint64_t>(want, std::max<uint64_t>(w.cap * 4, 1ull << 28));
            if ((rc = ensure_occ_capacity(x, want)) != ACX_OK) return rc;
        } else {
            // the regions' counts are exact (a full region keeps counting): the second pass puts every
            // region at the exclusive prefix of the counts -- room for exactly the occurrences there are,
            // however unevenly they are spread (grid * the fullest region can be 100x that)
            uint64_t *bases = w.block_counts + grid;
            HIPCHK_RC(sink_summary(w.block_counts, grid, ~0ull, nullptr, 0, 0, w.summary + 8, bases, st));
            HIPCHK_RC(hipMemcpyAsync(w.h_pinned + 10, bases + grid, 8, hipMemcpyDeviceToHost, st));
            HIPCHK_RC(hipStreamSynchronize(st));
            c.exact_total = w.h_pinned[10];
            if (c.exact_total >= occ_limit()) return fail_occ();
            // (an eighth of headroom: the NEXT call's uniform regions -- capacity / grid each -- then hold an
            // output that is spread as evenly as this one, and it needs no second pass)
            if ((rc = ensure_occ_capacity(x, c.exact_total + c.exact_total / 8 + 64 * (uint64_t)grid)) != ACX_OK) return rc;
            c.exact_regions = true;
        }
        *what = Attempt::Again;
        return ACX_OK;
    }
    if (n_raw >= occ_limit()) return fail_occ();
    // (round 5: a hold that a dense INPUT set ends with the first input that is not dense -- the dense path on a sparse input
    // costs 2-3x, eight calls of it were the price of one dense call in front: bench.py's 8 GiB run behind its density sweep)
    if (n_raw > 8 * c.tiles) { x->dense_hold = 8; x->hold_dense_input = true; }
    else if (x->dense_hold > 0) x->dense_hold = x->hold_dense_input ? 0 : x->dense_hold - 1;
    c.n_raw = n_raw;
    c.n_hits = c.pre ? w.h_pinned[2] : 0;
    *what = Attempt::Done;
    a->path[5]++;
    if (n_raw == 0) return ACX_OK;
    HIPCHK_RC(sink_compact(w.recs, w.region_off, grid, region_cap, w.keys[1], w.pids[1], st));
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
    c.queued = true;
    return ACX_OK;
}

// batch: the per-haystack counts start at zero.  No memset in front of the scan (it delayed the scan's
// launch by a dispatch and ~15 us of host time on every batch call): the sparse path has k_tile_main
// clear them on its way, every other path clears them here, when it gets to them.
int zero_counts(FindCall &c) {
    if (c.segmented && !c.counts_zeroed) {
        HIPCHK_RC(hipMemsetAsync(c.r->d_counts, 0, std::max<uint64_t>(c.G.n_hay, 1) * 8, c.c->stream));
        c.queued = true;
    }
    c.counts_zeroed = true;
    return ACX_OK;
}

// everything after the matches exist: code points (str API), local offsets + counts (batches)
int finish_matches(FindCall &c) {
    Ctx *x = c.c;
    Workspace &w = x->ws;
    hipStream_t st = x->stream;
    if (!c.n_final || c.cp_done || !(c.codepoints || (c.segmented && !c.localized))) return ACX_OK;
    if (c.codepoints) {
        const uint64_t nb1 = (c.len + 1023) / 1024 + 1;
        int rc = ensure_blocks(x, c.leads_counted ? std::max<uint64_t>(nb1, 4 * c.tiles + 1) : nb1);
        if (rc) return rc;
        if (!c.leads_counted) HIPCHK_RC(count_lead_bytes(c.d_hay, c.len, w.blockcnt, w.blocksub, st));
        HIPCHK_RC(block_prefix(c.leads_counted ? w.blocksub : nullptr, w.blockcnt, w.blockpre, nb1 - 1, w.temp, w.temp_bytes, st));
    }
    if (c.segmented) {
        int rc = zero_counts(c);
        if (rc) return rc;
        HIPCHK_RC(localize(c.G, c.d_hay, c.len, w.blockpre, w.blocksub, c.codepoints, c.r->d_matches, c.n_final,
                           c.r->d_counts, st));
    }
    else
        HIPCHK_RC(to_code_points(c.d_hay, c.len, w.blockpre, w.blocksub, c.r->d_matches, c.n_final, st));
    c.queued = true;
    return ACX_OK;
}

// the general pipeline on an allocated result
int run_pipeline(FindCall &c) {
    acx_automaton *a = c.a;
    Ctx *x = c.c;
    int rc = ensure_common(x);
    if (rc) return rc;
    c.pre = a->kernel == ACX_KERNEL_PREFILTER;
    c.scan_grid = c.pre ? prefilter_grid(c.d_hay, c.len, a->n_cus) : dfa_walk_grid(a->dev, c.len, a->n_cus);
    c.lead = (uint32_t)((uintptr_t)c.d_hay & 15);
    c.tiles = prefilter_tiles(c.d_hay, c.len);
    const bool no_sparse_env = std::getenv("ACX_NO_BUCKET") != nullptr; // tests / profiling: force the dense path (read per call)
    bool sparse = a->sparse_ok && x->dense_hold == 0 && !no_sparse_env && c.tiles < (1ull << 26);
    for (int attempt = 0;; attempt++) {
        if (attempt == 6) return fail(ACX_EDEVICE, "occurrence buffer overflow persisted");
        Attempt what = Attempt::Done;
        if ((rc = sparse ? attempt_sparse(c, &what) : attempt_dense(c, &what)) != ACX_OK) return rc;
        if (what == Attempt::GoDense) sparse = false;
        if (what == Attempt::Done) break;
    }
    if (c.timed) {
        std::lock_guard<std::mutex> lk(a->prof_mu);
        a->profile.raw_occurrences += c.n_raw;
        a->profile.prefix_hits += c.n_hits;
    }
    c.r->n = c.n_final;
    if ((rc = finish_matches(c)) != ACX_OK) return rc;
    if ((rc = zero_counts(c)) != ACX_OK) return rc; // (a batch without a match never got to them)
    static const bool prof_post = std::getenv("ACX_PROFILE_POST") != nullptr;
    if (a->prof && prof_post) { // end of the post stage (costs the next call a wait for this one's last kernel)
        HIPCHK_RC(hipEventRecord(x->ev[2], x->stream));
        x->post_pending = true;
    }
    return ACX_OK;
}

// Overlapping search over a set with copies of a string: the pipeline ran on the view without the later copies (one
// occurrence per string, lowest id); every occurrence becomes the run of its string's copies, ids ascending -- the order
// the reference reports them in (one state's match list, in the order the patterns were added).  In place of r->d_matches;
// batch: the per-haystack counts follow.  One round trip (the number of records).
int expand_copies(acx_automaton *a, Ctx *x, acx_result *r, bool segmented) {
    const uint64_t n = r->n;
    if (!n) return ACX_OK;
    hipStream_t st = x->stream;
    Workspace &w = x->ws;
    int rc = ensure_common(x);
    if (rc) return rc;
    const uint64_t n_hay = segmented ? r->n_hay : 0;
    const size_t tb = scan_temp_bytes(std::max(n, n_hay) + 1) + 256;
    void *temp = nullptr;
    uint64_t *k = nullptr, *offs = nullptr, *incl = nullptr;
    acx_match_t *out = nullptr;
    auto done = [&](int code) -> int {
        (void)hipStreamSynchronize(st); // (the scratch goes back to the pool: nothing may still use it)
        g_bufs.put(temp, a->device); g_bufs.put(k, a->device); g_bufs.put(offs, a->device); g_bufs.put(incl, a->device);
        g_bufs.put(out, a->device);
        return code;
    };
    HIPCHK_RC(g_bufs.get(&temp, tb, a->device));
    if (hipError_t e = g_bufs.get((void **)&k, (n + 1) * 8, a->device); e != hipSuccess) return done(hipfail(e, "expand_copies"));
    if (hipError_t e = g_bufs.get((void **)&offs, (n + 1) * 8, a->device); e != hipSuccess) return done(hipfail(e, "expand_copies"));
    hipError_t e = copy_runs(r->d_matches, n, a->d_xcnt, temp, tb, k, offs, st);
    if (e == hipSuccess) e = hipMemcpyAsync(w.h_pinned + 8, offs + n, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(hipfail(e, "expand_copies"));
    const uint64_t total = w.h_pinned[8];
    if (total == n) return done(ACX_OK); // (no occurrence of a string with copies)
    if ((e = g_bufs.get((void **)&out, total * sizeof(acx_match_t), a->device)) != hipSuccess) return done(hipfail(e, "expand_copies"));
    e = expand_copies_write(r->d_matches, n, offs, a->d_xoff, a->d_xids, out, total, st);
    if (e == hipSuccess && n_hay) {
        if ((e = g_bufs.get((void **)&incl, n_hay * 8, a->device)) == hipSuccess)
            e = expand_copies_counts(temp, tb, r->d_counts, n_hay, incl, offs, st);
    }
    if (e != hipSuccess) return done(hipfail(e, "expand_copies"));
    std::swap(out, r->d_matches); // (the unexpanded buffer goes back to the pool with the scratch)
    r->n = total;
    return done(ACX_OK);
}

// d_hay must stay valid until the result's device work is done (acx_result accessors wait for it)
int run_chunked(acx_automaton *a, Ctx *x, const uint8_t *d_hay, uint64_t len, int overlapping, int codepoints,
                acx_result **out, bool wait, uint64_t piece, int depth);
int run_batch_split(acx_automaton *a, Ctx *x, const uint8_t *d_hay, uint64_t len, const Segments &G, int overlapping,
                    int codepoints, acx_result **out, int depth);
int run_find(acx_automaton *a, Ctx *x, const uint8_t *d_hay, uint64_t len, const Segments &G,
             int overlapping, int codepoints, acx_result **out, bool allow_small, bool wait, int depth = 0,
             bool host_result = false) {
    *out = nullptr;
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD) {
        static const char *names[3] = {"Standard", "LeftmostFirst", "LeftmostLongest"};
        return fail(ACX_EOVERLAP, std::string("match kind ") + names[a->host.match_kind] +
                                      " does not support overlapping searches");
    }
    if (len >= (1ull << 38)) return fail(ACX_ETOOBIG, "haystack stream of 2^38 bytes or more");
    if (!x) return fail(ACX_EDEVICE, "could not create a stream for the call");
    settle_post_profile(a, x);
    hipStream_t st = x->stream;
    const bool segmented = G.uniform_len != 0 || G.offsets != nullptr;
    acx_result *r = new (std::nothrow) acx_result();
    if (!r) return fail(ACX_ENOMEM, "out of memory");
    r->device = a->device;
    r->n_hay = segmented ? G.n_hay : 0;
    FindCall c{a, x, d_hay, len, G, overlapping != 0, codepoints != 0, segmented, r,
               overlapping ? 0 : a->host.match_kind};
    c.host_result = host_result;
    auto body = [&]() -> int {
        if (segmented) {
            HIPCHK_RC(g_bufs.get((void **)&r->d_counts, std::max<uint64_t>(G.n_hay, 1) * 8, a->device));
        }
        if (allow_small && !segmented && small_ok(a, len)) { // small haystack: the whole call in one workgroup (K0)
            HIPCHK_RC(g_bufs.get((void **)&r->d_matches, SMALL_MAX_OCC * sizeof(acx_match_t), a->device));
            bool done = false;
            int rc = run_small(a, x, d_hay, len, overlapping, codepoints, r->d_matches, &r->n, &done);
            if (rc) return rc;
            if (done) return overlapping && a->expand_ov ? expand_copies(a, x, r, false) : ACX_OK;
            g_bufs.put(r->d_matches, a->device); // dense: the general pipeline takes over
            r->d_matches = nullptr;
        }
        if (len > 0 && a->host.n_patterns > 0) {
            int rc = run_pipeline(c);
            if (rc) return rc;
            if (overlapping && a->expand_ov) { // (copies of a string: the view reported the lowest ids)
                if ((rc = expand_copies(a, x, r, segmented)) != ACX_OK) return rc;
                c.queued = false; // (synchronised)
            }
        } else {
            // nothing to scan (a batch of empty haystacks, or no patterns): the per-haystack counts come
            // out of the buffer cache uninitialised -- they are this call's to clear
            int rc = zero_counts(c);
            if (rc) return rc;
        }
        if (c.queued) {
            // the totals are known; what is still running (the write kernel, the fix-ups) is fenced
            // by an event the result's accessors wait for
            if (wait) {
                HIPCHK_RC(hipStreamSynchronize(st));
            } else {
                if (!r->done) r->done = g_events.get(a->device);
                if (!r->done) HIPCHK_RC(hipStreamSynchronize(st));
                else HIPCHK_RC(hipEventRecord(r->done, st)); // (again, if a fix-up was queued behind an early record)
            }
        }
        return ACX_OK;
    };
    c.early_event = !wait;
    c.timed = a->prof && (a->prof_every <= 1 || (x->prof_calls++ % (uint32_t)a->prof_every) == 0);
    // (tests: ACX_CHUNK_BYTES cuts every one-haystack call longer than that, whatever it holds)
    const char *cb = depth == 0 && !segmented ? std::getenv("ACX_CHUNK_BYTES") : nullptr;
    const uint64_t forced = cb ? std::strtoull(cb, nullptr, 10) : 0;
    int rc = forced && len > forced ? TOO_MANY_OCC : body();
    if (rc != ACX_OK) {
        (void)hipStreamSynchronize(st);
        acx_free_result(r);
        if (rc != TOO_MANY_OCC) return rc;
        // more occurrences than one pass can index: the haystack in byte ranges, one after the other -- a batch in two parts,
        // cut at a haystack boundary, where nothing has to be carried over (round 6; until then the error was the caller's)
        const uint64_t m = a->host.max_len ? a->host.max_len - 1 : 0;
        const uint64_t piece = forced && len > forced ? forced : len / 2;
        if (segmented && depth < 40) return run_batch_split(a, x, d_hay, len, G, overlapping, codepoints, out, depth + 1);
        if (segmented || depth >= 40 || piece <= 2 * m + 16)
            return fail(ACX_ETOOBIG, "more than 2^32 occurrences");
        return run_chunked(a, x, d_hay, len, overlapping, codepoints, out, wait, piece, depth + 1);
    }
    *out = r;
    return ACX_OK;
}

// A batch whose occurrences one pass cannot index (2^32, the width of the device's indexes): its haystacks in two parts,
// searched one after the other on the same context -- each part again a batch (or, a part of ONE haystack, a call of its
// own, which may go on in byte ranges) -- and the parts' matches (offsets are local to their haystack: nothing to shift)
// and per-haystack counts put behind one another.  The reference's loop has no limit (/root/reference/src/lib.rs:53, 59).
int run_batch_split(acx_automaton *a, Ctx *x, const uint8_t *d_hay, uint64_t len, const Segments &G, int overlapping,
                    int codepoints, acx_result **out, int depth) {
    *out = nullptr;
    hipStream_t st = x->stream;
    Workspace &w = x->ws;
    const uint64_t n = G.n_hay;
    a->path[8]++;
    auto one_haystack = [&](const uint8_t *h, uint64_t hl, acx_result **r) -> int { // a part of one haystack: its count is its matches
        int rc = run_find(a, x, h, hl, Segments{nullptr, 1, 0}, overlapping, codepoints, r, true, true, depth);
        if (rc != ACX_OK) return rc;
        hipError_t e = g_bufs.get((void **)&(*r)->d_counts, 8, a->device);
        if (e == hipSuccess) e = hipMemcpyAsync((*r)->d_counts, &(*r)->n, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { acx_free_result(*r); *r = nullptr; return hipfail(e, "count of a one-haystack part"); }
        (*r)->n_hay = 1;
        return ACX_OK;
    };
    if (n <= 1) return one_haystack(d_hay, len, out);
    const uint64_t half = n / 2;
    uint64_t cut = 0; // the first byte of haystack `half`
    uint64_t *reb = nullptr; // the second part's offsets, from its first byte
    if (G.uniform_len) {
        cut = half * G.uniform_len;
    } else {
        HIPCHK(hipMemcpyAsync(w.h_pinned + 8, G.offsets + half, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        cut = w.h_pinned[8];
        HIPCHK(g_bufs.get((void **)&reb, (n - half + 1) * 8, a->device));
        hipError_t e = rebase_offsets(reb, G.offsets + half, n - half + 1, cut, st);
        if (e != hipSuccess) { g_bufs.put(reb, a->device); return hipfail(e, "rebase_offsets"); }
    }
    acx_result *ra = nullptr, *rb = nullptr;
    auto part = [&](const uint8_t *h, uint64_t hl, const uint64_t *offs, uint64_t k, acx_result **r) -> int {
        if (k == 1) return one_haystack(h, hl, r);
        const Segments S{G.uniform_len ? nullptr : offs, k, G.uniform_len};
        return run_find(a, x, h, hl, S, overlapping, codepoints, r, true, true, depth);
    };
    int rc = part(d_hay, cut, G.offsets, half, &ra);
    if (rc == ACX_OK) rc = part(d_hay + cut, len - cut, reb, n - half, &rb);
    acx_result *r = rc == ACX_OK ? new (std::nothrow) acx_result() : nullptr;
    if (rc == ACX_OK && !r) rc = fail(ACX_ENOMEM, "out of memory");
    if (rc == ACX_OK) {
        r->device = a->device;
        r->n_hay = n;
        r->n = ra->n + rb->n;
        hipError_t e = g_bufs.get((void **)&r->d_counts, n * 8, a->device);
        if (e == hipSuccess && r->n) e = g_bufs.get((void **)&r->d_matches, r->n * sizeof(acx_match_t), a->device);
        if (e == hipSuccess) e = hipMemcpyAsync(r->d_counts, ra->d_counts, half * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(r->d_counts + half, rb->d_counts, (n - half) * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && ra->n) e = hipMemcpyAsync(r->d_matches, ra->d_matches, ra->n * sizeof(acx_match_t), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && rb->n) e = hipMemcpyAsync(r->d_matches + ra->n, rb->d_matches, rb->n * sizeof(acx_match_t), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = hipfail(e, "result of a batch in two parts");
    }
    hipError_t e2 = hipStreamSynchronize(st); // (the parts' buffers go back to the pool: nothing may still read them)
    if (rc == ACX_OK && e2 != hipSuccess) rc = hipfail(e2, "a batch in two parts");
    if (ra) acx_free_result(ra);
    if (rb) acx_free_result(rb);
    if (reb) g_bufs.put(reb, a->device);
    if (rc != ACX_OK) { if (r) acx_free_result(r); return rc; }
    *out = r;
    return ACX_OK;
}

// One haystack in byte ranges of `piece` bytes, searched one after the other; the pieces' matches, cut where the ranges
// meet, are copied into one result with global offsets.  The reference's loop has no limit on what it reports
// (/root/reference/src/lib.rs:53, 59: an iterator); one pass here has -- 2^32 occurrences, the width of the device's
// indexes.  What couples the ranges is what couples the ranks of a sharded haystack (distributed.py):
//   overlapping      a range reports the occurrences that END in (lo, hi]; it is scanned from max_len - 1 bytes before lo;
//   non-overlapping  a range reports the matches that START in [carry, hi): the iteration resumes at `carry`, the end of
//                    the last match in front (or lo); a match that starts before hi ends at most max_len - 1 bytes behind
//                    it, so the scan stops there, and nothing the truncated window hides beats what it shows.
// Code points (str API): the pieces run on byte offsets, the conversion runs once over the whole result.
int run_chunked(acx_automaton *a, Ctx *x, const uint8_t *d_hay, uint64_t len, int overlapping, int codepoints,
                acx_result **out, bool wait, uint64_t piece, int depth) {
    *out = nullptr;
    hipStream_t st = x->stream;
    Workspace &w = x->ws;
    const uint64_t m = a->host.max_len ? a->host.max_len - 1 : 0;
    struct Piece { acx_match_t *buf; uint64_t first, n, shift; };
    std::vector<Piece> pieces;
    auto drop = [&]() { for (auto &p : pieces) g_bufs.put(p.buf, a->device); pieces.clear(); };
    uint64_t total = 0, carry = 0;
    if (int rc = ensure_common(x)) return rc; // (the context's scratch: the first call of a context may be this one)
    uint64_t *cut = w.summary + 12; // (two device words of the context's scratch)
    for (uint64_t lo = 0; lo < len;) {
        const uint64_t hi = std::min(len, lo + piece);
        const bool last = hi == len;
        uint64_t a0, a1;
        if (overlapping) { a0 = lo > m ? lo - m : 0; a1 = hi; }
        else { carry = std::max(carry, lo); a0 = carry; a1 = last ? len : std::min(len, hi + m); }
        if (a0 >= hi) { lo = hi; continue; } // (a match from the ranges in front covers this one)
        acx_result *r = nullptr;
        a->path[8]++;
        int rc = run_find(a, x, d_hay + a0, a1 - a0, Segments{nullptr, 1, 0}, overlapping, 0, &r, true, true, depth);
        if (rc != ACX_OK) { drop(); return rc; }
        uint64_t first = 0, n = r->n, last_end = 0;
        if (n && ((overlapping && lo > 0) || (!overlapping && !last))) {
            // overlapping: the occurrences that end at or before lo belong to the range in front (a prefix: ordered by
            // end); non-overlapping: the matches that start at or behind hi to the next one (a suffix: ordered by start)
            hipError_t e = overlapping ? cut_point(r->d_matches, n, true, a0, lo + 1, cut, st)
                                       : cut_point(r->d_matches, n, false, a0, hi, cut, st);
            if (e == hipSuccess) e = hipMemcpyAsync(w.h_pinned + 8, cut, 16, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { acx_free_result(r); drop(); return hipfail(e, "cut of a byte range"); }
            if (overlapping) first = w.h_pinned[8];
            else { n = w.h_pinned[8]; last_end = w.h_pinned[9]; }
        } else if (n && !overlapping) {
            last_end = len; // (the last range: nothing follows)
        }
        if (n > first) {
            pieces.push_back(Piece{r->d_matches, first, n - first, a0});
            r->d_matches = nullptr; // (ours now)
            total += n - first;
        }
        acx_free_result(r);
        if (!overlapping) carry = std::max(std::max(carry, hi), last_end);
        lo = hi;
    }
    acx_result *r = new (std::nothrow) acx_result();
    if (!r) { drop(); return fail(ACX_ENOMEM, "out of memory"); }
    r->device = a->device;
    r->n = total;
    auto body = [&]() -> int {
        if (!total) return ACX_OK;
        hipError_t e = g_bufs.get((void **)&r->d_matches, total * sizeof(acx_match_t), a->device);
        if (e != hipSuccess) return hipfail(e, "result of a call in byte ranges");
        uint64_t at = 0;
        for (auto &p : pieces) {
            if ((e = copy_shifted(r->d_matches + at, p.buf + p.first, p.n, p.shift, st)) != hipSuccess) return hipfail(e, "copy_shifted");
            at += p.n;
        }
        if (codepoints) {
            const Segments one{nullptr, 1, 0};
            FindCall c{a, x, d_hay, len, one, overlapping != 0, true, false, r, 0};
            c.n_final = total;
            int rc = finish_matches(c);
            if (rc) return rc;
        }
        return ACX_OK;
    };
    int rc = body();
    hipError_t e = hipStreamSynchronize(st); // (the pieces' buffers go back to the pool: nothing may still read them)
    drop();
    if (rc == ACX_OK && e != hipSuccess) rc = hipfail(e, "a call in byte ranges");
    if (rc != ACX_OK) { acx_free_result(r); return rc; }
    (void)wait; // (synchronised either way)
    *out = r;
    return ACX_OK;
}
#undef HIPCHK_RC

// ---------------------------------------------------------------------------
// host memory -> device staging buffer
// ---------------------------------------------------------------------------
// Small inputs: one hipMemcpyAsync from the caller's (pageable) memory.  Large inputs: the
// runtime's own pageable path moves 20-40 GB/s (one staging thread), so the bytes are copied by
// several host threads into a ring of pinned chunks, each chunk DMA'd by its own asynchronous copy
// while the threads fill the next one; the scan is queued behind the last chunk.
constexpr uint64_t STAGE_DIRECT_MAX = 8ull << 20;

int stage_host(acx_automaton *a, Ctx *c, const uint8_t *hay, uint64_t len, const uint64_t *offsets,
               uint64_t n_off) {
    Workspace &w = c->ws;
    hipStream_t st = c->stream;
    (void)a;
    if (len > w.hay_cap) {
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(w.hay); w.hay = nullptr; w.hay_cap = 0;
        uint64_t cap = std::max<uint64_t>(len + len / 8, 4096);
        HIPCHK(hipMalloc((void **)&w.hay, cap));
        w.hay_cap = cap;
    }
    if (n_off) {
        if (n_off > w.offsets_cap) {
            HIPCHK(hipStreamSynchronize(st));
            (void)hipFree(w.offsets); w.offsets = nullptr; w.offsets_cap = 0;
            HIPCHK(hipMalloc((void **)&w.offsets, n_off * 8));
            w.offsets_cap = n_off;
        }
        HIPCHK(hipMemcpyAsync(w.offsets, offsets, n_off * 8, hipMemcpyHostToDevice, st));
    }
    // ACX_STAGE: 1 = the runtime's own pageable copy (default: measured 54 GB/s on the MI355X box, the
    // link's rate, once the result no longer lands in freshly faulted pages), 2 = pin the caller's
    // pages for the call (54.8 GB/s), 3 = the ring of pinned chunks below (51 GB/s; independent of the
    // runtime's staging)
    static const int mode = std::getenv("ACX_STAGE") ? std::atoi(std::getenv("ACX_STAGE")) : 1;
    if (len <= STAGE_DIRECT_MAX || mode == 1) {
        if (len) HIPCHK(hipMemcpyAsync(w.hay, hay, len, hipMemcpyHostToDevice, st));
        return ACX_OK;
    }
    if (mode == 2) { // measurements: pin the caller's pages for the call, one DMA
        HIPCHK(hipHostRegister((void *)hay, len, hipHostRegisterDefault));
        hipError_t e = hipMemcpyAsync(w.hay, hay, len, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipHostUnregister((void *)hay);
        if (e != hipSuccess) return hipfail(e, "registered host copy");
        return ACX_OK;
    }
    static const size_t chunk = std::getenv("ACX_STAGE_CHUNK") ? (size_t)std::atoll(std::getenv("ACX_STAGE_CHUNK"))
                                                                : ((size_t)16 << 20);
    if (w.chunk_bytes != chunk) {
        for (int i = 0; i < Workspace::RING; i++) {
            if (w.pin_chunk[i]) { (void)hipHostFree(w.pin_chunk[i]); w.pin_chunk[i] = nullptr; }
            HIPCHK(hipHostMalloc((void **)&w.pin_chunk[i], chunk, hipHostMallocDefault));
            if (!w.chunk_ev[i]) HIPCHK(hipEventCreateWithFlags(&w.chunk_ev[i], hipEventDisableTiming));
        }
        w.chunk_bytes = chunk;
    }
    CopyPool &pool = CopyPool::get();
    uint64_t k = 0;
    for (uint64_t off = 0; off < len; off += chunk, k++) {
        const int slot = (int)(k % Workspace::RING);
        const size_t n = (size_t)std::min<uint64_t>(chunk, len - off);
        if (k >= (uint64_t)Workspace::RING) HIPCHK(hipEventSynchronize(w.chunk_ev[slot])); // its last DMA has read it
        pool.copy(w.pin_chunk[slot], hay + off, n);
        HIPCHK(hipMemcpyAsync(w.hay + off, w.pin_chunk[slot], n, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(hipEventRecord(w.chunk_ev[slot], c->copy_stream));
    }
    HIPCHK(hipEventRecord(c->copy_done, c->copy_stream));
    HIPCHK(hipStreamWaitEvent(st, c->copy_done, 0));
    return ACX_OK;
}

// device result -> host array the caller owns (acx_free_matches).  Large results land in a pinned
// buffer that is handed out as it is.
int download_matches(const acx_result *r, acx_match_t **out, uint64_t *n_out) {
    *out = nullptr;
    const uint64_t n = r->n;
    *n_out = n;
    if (!n) return result_wait(r);
    const size_t bytes = n * sizeof(acx_match_t);
    acx_match_t *m = nullptr;
    if (bytes >= (1u << 20)) m = (acx_match_t *)g_pinned_results.get(bytes);
    const bool pinned = m != nullptr;
    if (!m) m = (acx_match_t *)std::malloc(bytes);
    if (!m) return fail(ACX_ENOMEM, "out of memory");
    int rc = acx_result_copy(r, m);
    if (rc != ACX_OK) {
        if (pinned) g_pinned_results.put(m); else std::free(m);
        return rc;
    }
    *out = m;
    return ACX_OK;
}

} // namespace

// ---------------------------------------------------------------------------
extern "C" {

// (comm.cpp reports through the same thread-local message)
int acx_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }

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
        // implementation=DFA asks for the dense table outright (the reference's DFA, README.md:173-177,
        // has no size limit either): keep it up to 16 GiB of the 288 GB
        err = compile(blob, n_patterns ? offsets : zero_off, n_patterns, match_kind, a->host, code,
                      implementation == ACX_IMPL_DFA ? (16ull << 30) : 0);
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
    if (const char *envc = std::getenv("ACX_MAX_CONCURRENCY")) a->max_ctx = std::max(1, std::min(16, std::atoi(envc)));
    DeviceScope scope(dev);
    auto destroy = [&](int rc) { acx_free_automaton(a); return rc; };
#define HIPCHK_A(expr)                                            \
    do {                                                          \
        hipError_t e__ = (expr);                                  \
        if (e__ != hipSuccess) return destroy(hipfail(e__, #expr)); \
    } while (0)
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
    Ctx *c0 = create_ctx();
    if (!c0) return destroy(fail(ACX_EDEVICE, "could not create a HIP stream"));
    a->ctxs.push_back(c0);
    a->idle.push_back(c0);
    hipStream_t st = c0->stream;

    Automaton &H = a->host;
    DevAutomaton &D = a->dev;
    D.n_patterns = H.n_patterns; D.n_states = H.n_states; D.stride2 = H.stride2;
    D.min_len = H.min_len; D.max_len = H.max_len; D.filter_q = H.filter_q;
    D.ptab_log2 = H.ptab_log2; D.filter_q2 = H.filter_q2;
    D.short_min_len = H.n_short ? H.short_min_len : 0; D.k1b_min_len = H.long_min_len;
    D.max_shift = H.max_shift;
    {
        static const char *big_env = std::getenv("ACX_FILTER_BIG"); // measurements: 0 / 1 force the choice
        // 1: the level-1 table is saturated (10^5 patterns): every position put to both tests; 2: it passes nearly every
        // position (10^6 patterns): ... and the survivors' windows captured from the row staged in LDS (kernels.hip: K1bLds)
        D.filter_big = big_env ? (uint32_t)std::atoi(big_env)
                               : (H.filter_q == 5 && H.filter_density > 0.6 ? 2u : H.filter_q == 5 && H.filter_density > 0.2 ? 1u : 0u);
    }
    D.rank_bits = (uint32_t)std::max(1, bits_for(H.n_patterns ? H.n_patterns - 1 : 0));
    // compact u16 copy of the hot (lowest-id) rows for K1a's LDS tile
    uint32_t hot_rows = H.dense ? dfa_walk_hot_rows(H.n_states, H.stride2, 160 * 1024) : 0;
    std::vector<uint16_t> hot16(((size_t)hot_rows << H.stride2) + 8, 0xFFFF);
    for (size_t i = 0; i < ((size_t)hot_rows << H.stride2); i++) {
        uint32_t en = H.table[i], id = en & ID_MASK;
        hot16[i] = id < 0x3FFFu ? (uint16_t)(id | ((en >> 30) << 14)) : (uint16_t)0xFFFF;
    }
    D.hot_rows = hot_rows;
    // K1a's compact table (automata of at most 65 535 states): see DevAutomaton::table16
    std::vector<uint16_t> table16;
    std::vector<uint32_t> walk_bfs;
    D.n_classes = H.n_classes;
    D.walk_plain = 0;
    if (H.dense && H.n_states <= 0xFFFF && H.n_patterns > 0) {
        const uint32_t NS = H.n_states, NC = H.n_classes, S = H.stride;
        std::vector<uint8_t> reports(NS, 0); // FLAG_OUT is a property of the TARGET state
        for (size_t i = 0; i < (size_t)NS * S; i++)
            if (H.table[i] & FLAG_OUT) reports[H.table[i] & ID_MASK] = 1;
        std::vector<uint32_t> walk_of(NS);
        walk_bfs.resize(NS);
        uint32_t k = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t s = 0; s < NS; s++)
                if (reports[s] == pass) { walk_of[s] = k; walk_bfs[k] = s; k++; }
            if (pass == 0) D.walk_plain = k;
        }
        table16.assign((size_t)NS * NC + 8, 0);
        for (uint32_t w = 0; w < NS; w++)
            for (uint32_t c = 0; c < NC; c++)
                table16[(size_t)w * NC + c] = (uint16_t)walk_of[H.table[(size_t)walk_bfs[w] * S + c] & ID_MASK];
    }
    // (K1a's failureless form -- walk_t3b / walk_t3r / walk_grec -- is part of the host compiler's output)
    int rc;
#define UP(vec, field)                                                                   \
    if ((rc = upload(a, st, (vec).data(), (vec).size(), &D.field)) != ACX_OK) return destroy(rc);
    if (H.dense) { UP(H.table, table) } else { D.table = nullptr; }
    UP(H.first_child, first_child)
    UP(H.in_byte, in_byte)
    UP(H.fail, fail)
    UP(H.sflags, sflags)
    UP(H.root_next, root_next)
    UP(hot16, hot16)
    if (!table16.empty()) {
        UP(table16, table16)
        UP(walk_bfs, walk_bfs)
    } else {
        D.table16 = nullptr; D.walk_bfs = nullptr;
    }
    if (!H.walk_t3b.empty()) {
        UP(H.walk_t3b, t3b)
        const uint32_t *p2 = nullptr;
        if ((rc = upload(a, st, H.walk_t3r.data(), H.walk_t3r.size(), &p2)) != ACX_OK) return destroy(rc);
        D.t3r = reinterpret_cast<const uint2 *>(p2);
        if ((rc = upload(a, st, H.walk_grec.data(), H.walk_grec.size(), &p2)) != ACX_OK) return destroy(rc);
        D.grec = reinterpret_cast<const uint4 *>(p2);
    } else {
        D.t3b = nullptr; D.t3r = nullptr; D.grec = nullptr;
    }
    UP(H.own_off, own_off)
    UP(H.own_pid, own_pid)
    UP(H.own1, own1)
    UP(H.dlink, dlink)
    UP(H.level_start, level_start)
    UP(H.plen, plen)
    std::vector<uint32_t> pchars(H.plen.size() + 1, 0);
    for (size_t i = 0; i < H.plen.size(); i++)
        for (uint64_t k = H.offsets[i]; k < H.offsets[i + 1]; k++) pchars[i] += (H.blob[k] & 0xC0) != 0x80;
    UP(pchars, pchars)
    UP(H.rank, rank)
    std::vector<uint32_t> by_rank(H.rank.size() + 1, 0);
    for (uint32_t i = 0; i < H.rank.size(); i++) by_rank[H.rank[i]] = i;
    UP(by_rank, by_rank)
    UP(H.filterA, filterA)
    UP(H.ptab, ptab)
    UP(H.blist, blist)
    if (H.rbloom.empty()) H.rbloom.assign(REDIRECT_BLOOM_WORDS, 0);
    UP(H.rbloom, rbloom)
    if (H.pbits.empty()) H.pbits.assign(4, 0);
    UP(H.pbits, pbits)
    if (H.short_xy.empty()) { H.short_xy.assign(SHORT_XY_WORDS, 0); H.short_codes.assign(4, SHORT_NONE); }
    UP(H.short_xy, short_xy)
    if (H.max_shift) {
        const uint32_t *ph = nullptr;
        if ((rc = upload(a, st, H.phead.data(), H.phead.size(), &ph)) != ACX_OK) return destroy(rc);
        D.phead = reinterpret_cast<const uint4 *>(ph);
    } else {
        D.phead = nullptr;
    }
    UP(H.short_codes, short_codes)
    {
        const uint32_t *pi = nullptr;
        if ((rc = upload(a, st, H.pinfo.data(), H.pinfo.size(), &pi)) != ACX_OK) return destroy(rc);
        D.pinfo = reinterpret_cast<const uint4 *>(pi);
    }
    H.blob.resize(H.blob.size() + 16, 0); // the verification compares 8 bytes at a time
    UP(H.blob, pat_blob)
    UP(H.offsets, pat_off)
#undef UP
    if ((rc = upload(a, st, H.classes, (size_t)256, &D.classes)) != ACX_OK) return destroy(rc);
    if ((rc = upload(a, st, &a->dev, (size_t)1, &a->d_dev)) != ACX_OK) return destroy(rc);
    {
        // the view of a non-overlapping search (struct acx_automaton): only when some string is there more than once
        static const bool no_nov = std::getenv("ACX_NO_COPY_VIEW") != nullptr; // measurements
        std::vector<uint8_t> later(H.n_patterns, 0); // a copy of a string with a lower id
        uint64_t n_later = 0;
        for (uint32_t s2 = 0; s2 < H.n_states && H.match_kind == ACX_MATCH_STANDARD; s2++)
            for (uint32_t k = H.own_off[s2] + 1; k < H.own_off[s2 + 1]; k++) { later[H.own_pid[k]] = 1; n_later++; }
        if (n_later && !no_nov) {
            std::vector<uint32_t> own1_nov(H.n_states, OWN1_NONE);
            for (uint32_t s2 = 0; s2 < H.n_states; s2++)
                if (H.own_off[s2 + 1] > H.own_off[s2]) own1_nov[s2] = H.own_pid[H.own_off[s2]]; // (lists are in id order)
            std::vector<uint32_t> blist_nov(H.blist);
            for (size_t i = 0; i < blist_nov.size();) { // [count, codes ...] records, back to back
                const uint32_t cnt = H.blist[i];
                uint32_t kept = 0;
                for (uint32_t k = 0; k < cnt; k++)
                    if (!later[H.blist[i + 1 + k] & CODE_PID_MASK]) blist_nov[i + 1 + kept++] = H.blist[i + 1 + k];
                uint32_t rest = kept;
                for (uint32_t k = 0; k < cnt; k++)
                    if (later[H.blist[i + 1 + k] & CODE_PID_MASK]) blist_nov[i + 1 + rest++] = H.blist[i + 1 + k];
                blist_nov[i] = kept;
                i += (size_t)cnt + 1;
            }
            std::vector<uint32_t> own_off_nov((size_t)H.n_states + 1, 0), own_pid_nov(H.own_pid.size(), 0);
            for (uint32_t s2 = 0; s2 < H.n_states; s2++) {
                own_off_nov[s2 + 1] = own_off_nov[s2];
                if (own1_nov[s2] != OWN1_NONE) own_pid_nov[own_off_nov[s2 + 1]++] = own1_nov[s2];
            }
            a->dev_nov = a->dev;
            if ((rc = upload(a, st, own1_nov.data(), own1_nov.size(), &a->dev_nov.own1)) != ACX_OK) return destroy(rc);
            if ((rc = upload(a, st, blist_nov.data(), blist_nov.size(), &a->dev_nov.blist)) != ACX_OK) return destroy(rc);
            if ((rc = upload(a, st, own_off_nov.data(), own_off_nov.size(), &a->dev_nov.own_off)) != ACX_OK) return destroy(rc);
            if ((rc = upload(a, st, own_pid_nov.data(), own_pid_nov.size(), &a->dev_nov.own_pid)) != ACX_OK) return destroy(rc);
            if (!H.walk_grec.empty()) { // the failureless walk's trie records: {children bitmap, first child | OWN, own1, ..}
                std::vector<uint32_t> grec_nov(H.walk_grec);
                // (only the records that say "several": a tail record's third word is its leaf's pattern, not own1)
                for (uint32_t s2 = 0; s2 < H.n_states; s2++)
                    if (grec_nov[4 * (size_t)s2 + 2] == OWN1_MANY) grec_nov[4 * (size_t)s2 + 2] = own1_nov[s2];
                const uint32_t *p2 = nullptr;
                if ((rc = upload(a, st, grec_nov.data(), grec_nov.size(), &p2)) != ACX_OK) return destroy(rc);
                a->dev_nov.grec = reinterpret_cast<const uint4 *>(p2);
            }
            if ((rc = upload(a, st, &a->dev_nov, (size_t)1, &a->d_dev_nov)) != ACX_OK) return destroy(rc);
            // the copies of every lowest id, for the expansion of an overlapping search's result
            a->x_cnt.assign(H.n_patterns, 0);
            a->x_off.assign(H.n_patterns, 0);
            a->x_ids.reserve(n_later);
            for (uint32_t s2 = 0; s2 < H.n_states; s2++) {
                const uint32_t b = H.own_off[s2], e2 = H.own_off[s2 + 1];
                if (e2 - b < 2) continue;
                a->x_off[H.own_pid[b]] = (uint32_t)a->x_ids.size();
                a->x_cnt[H.own_pid[b]] = e2 - b - 1;
                for (uint32_t k = b + 1; k < e2; k++) a->x_ids.push_back(H.own_pid[k]);
            }
            if ((rc = upload(a, st, a->x_cnt.data(), a->x_cnt.size(), &a->d_xcnt)) != ACX_OK) return destroy(rc);
            if ((rc = upload(a, st, a->x_off.data(), a->x_off.size(), &a->d_xoff)) != ACX_OK) return destroy(rc);
            if ((rc = upload(a, st, a->x_ids.data(), a->x_ids.size(), &a->d_xids)) != ACX_OK) return destroy(rc);
            a->has_nov = true;
            const char *xe = std::getenv("ACX_EXPAND_COPIES");
            a->expand_ov = xe ? std::atoi(xe) != 0 : n_later * 4 >= (uint64_t)H.n_patterns;
        }
    }
    HIPCHK_A(hipStreamSynchronize(st));
    a->table_bytes = H.table.size() * 4;
    // the big host copy of the table is no longer needed
    std::vector<uint32_t>().swap(H.table);
    a->sparse_ok = tile_lookback(H.max_len) <= MAX_LOOKBACK;
    // kernel selection
    // The Implementation hint never selects a slower scan (the reference's README recommends the
    // contiguous NFA as the sensible default, README.md:173-177: a caller following that advice must
    // not pay for it): it only decides how large a dense table is kept (acx_build above).  The plain
    // DFA walk stays reachable through acx_set_kernel / ACX_KERNEL=dfa_walk.
    bool prefilter_ok = H.filter_q >= 3 && a->max_lds >= prefilter_lds_bytes();
    a->implementation = implementation;
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
    out->classes = A.classes; out->table = A.dense ? A.table.data() : nullptr;
    out->prefix_bitmap = A.pbits.data();
    out->dense = A.dense ? 1 : 0;
    out->first_child = A.first_child.data(); out->in_byte = A.in_byte.data();
    out->fail = A.fail.data(); out->state_flags = A.sflags.data();
    out->walk_t3b = A.walk_t3b.empty() ? nullptr : A.walk_t3b.data();
    out->walk_t3r = A.walk_t3r.empty() ? nullptr : A.walk_t3r.data();
    out->walk_grec = A.walk_grec.empty() ? nullptr : A.walk_grec.data();
    out->own_off = A.own_off.data(); out->own_pid = A.own_pid.data();
    out->dlink = A.dlink.data(); out->level_start = A.level_start.data();
    out->pattern_len = A.plen.data(); out->rank = A.rank.data();
    out->filter_xy = A.filterA.data();
    out->prefix_table = A.ptab.data();
    out->prefix_lists = A.blist.data();
    out->filter_q = A.filter_q; out->filter_q2 = A.filter_q2;
    out->filter_entries_log2 = FILTER_ENTRIES_LOG2; out->prefix_table_log2 = A.ptab_log2;
    out->filter_density = A.filter_density;
    out->n_prefix_keys = A.n_prefix_keys;
    out->n_prefix_lists = (uint32_t)A.blist.size();
    out->max_shift = A.max_shift; out->pattern_shift = A.shift.data();
    out->pattern_head = A.phead.empty() ? nullptr : A.phead.data();
    out->long_min_len = A.long_min_len; out->n_short = A.n_short; out->short_min_len = A.n_short ? A.short_min_len : 0;
    out->short_xy = A.n_short ? A.short_xy.data() : nullptr;
    out->short_codes = A.n_short ? A.short_codes.data() : nullptr;
    return ACX_OK;
}

uint32_t acx_filter_hash(uint32_t gram) { return filter_hash(gram); }
uint32_t acx_prefix_slot(uint64_t gram, uint32_t q2, uint32_t log2) {
    return prefix_slot(prefix_home_hash(q2 >= 8 ? gram : (gram & ((1ull << (8 * q2)) - 1)), q2), log2);
}

void acx_free_host(acx_host_automaton_t *h) { delete h; }

void acx_free_automaton(acx_automaton_t *a) {
    if (!a) return;
    DeviceScope scope(a->device);
    for (Ctx *c : a->ctxs) destroy_ctx(c, a->device);
    for (void *p : a->allocs) (void)hipFree(p);
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
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD)
        return run_find(a, nullptr, nullptr, 0, G, overlapping, codepoints, out, true, true); // the error, no device state
    g_trace.mark(0);
    int rc;
    {
        Lease lease(a);
        g_trace.mark(1);
        // returns when the totals are known; accessors of the result wait for the rest of its device work
        rc = run_find(a, lease.c, (const uint8_t *)d_hay, len, G, overlapping, codepoints, out, true, false);
    }
    g_trace.mark(7);
    return rc;
}

uint64_t acx_result_count(const acx_result_t *r) { return r ? r->n : 0; }
const acx_match_t *acx_result_device_matches(const acx_result_t *r) {
    if (!r || result_wait(r) != ACX_OK) return nullptr;
    return r->d_matches;
}
const uint64_t *acx_result_device_counts(const acx_result_t *r) {
    if (!r || result_wait(r) != ACX_OK) return nullptr;
    return r->d_counts;
}

int acx_result_copy(const acx_result_t *r, acx_match_t *host_out) {
    if (!r) return fail(ACX_EINVAL, "null result");
    int rc = result_wait(r);
    if (rc != ACX_OK || !r->n) return rc;
    DeviceScope ds(r->device);
    HIPCHK(hipMemcpy(host_out, r->d_matches, r->n * sizeof(acx_match_t), hipMemcpyDeviceToHost));
    return ACX_OK;
}

int acx_result_copy_counts(const acx_result_t *r, uint64_t *host_counts) {
    if (!r) return fail(ACX_EINVAL, "null result");
    int rc = result_wait(r);
    if (rc != ACX_OK || !r->d_counts || !r->n_hay) return rc;
    DeviceScope ds(r->device);
    HIPCHK(hipMemcpy(host_counts, r->d_counts, r->n_hay * 8, hipMemcpyDeviceToHost));
    return ACX_OK;
}

void acx_free_result(acx_result_t *r) {
    if (!r) return;
    g_trace.mark(8);
    struct AtExit { ~AtExit() { g_trace.mark(9); } } at_exit;
    // the buffers may still be written by the call's last kernels: the cache holds them back until
    // the event has fired (one event guards both buffers; nobody waits here -- a batch caller that
    // frees a result and starts the next call used to sit out the write kernel in this function)
    g_bufs.put(r->borrowed ? nullptr : r->d_matches, r->device, r->done, r->d_counts);
    delete r;
}

int acx_find(acx_automaton_t *a, const uint8_t *hay, uint64_t len, int overlapping, int codepoints,
             acx_match_t **out, uint64_t *n_out) {
    if (!a || !out || !n_out) return fail(ACX_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    if (len && !hay) return fail(ACX_EINVAL, "null haystack");
    acx_result_t *r = nullptr;
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD) // produces the error message, touches no device state
        return run_find(a, nullptr, nullptr, 0, Segments{nullptr, 1, 0}, overlapping, codepoints, &r, true, true);
    const bool try_small = small_ok(a, len);
    Lease lease(a, try_small);
    Ctx *c = lease.c;
    if (!c) return fail(ACX_EDEVICE, "could not create a stream for the call");
    int rc;
    if (try_small) {
        // small haystack: copy it into pinned memory; the context's resident K0 takes it from there (a poll on either
        // side), or ONE launch does (K0 reads and writes pinned host memory in place) -- no H2D / D2H copies at all
        Workspace &w = c->ws;
        if (!w.mailbox) {
            // (both read by the other side while a kernel runs: system-coherent)
            HIPCHK(hipHostMalloc((void **)&w.mailbox, K0_MAILBOX_HAY + SMALL_PF_MAX_LEN + 32, hipHostMallocCoherent));
            w.mailbox[0] = 0;
            w.pin_hay = (uint8_t *)w.mailbox + K0_MAILBOX_HAY;
            HIPCHK(hipHostMalloc((void **)&w.pin_out, SMALL_MAX_OCC * sizeof(acx_match_t), hipHostMallocCoherent));
        }
        uint64_t n = 0;
        bool done = false, taken = false;
        rc = run_resident(a, c, hay, len, overlapping, codepoints, &n, &done, &taken);
        if (rc != ACX_OK) return rc;
        if (!taken) {
            std::memcpy(w.pin_hay, hay, len);
            rc = run_small(a, c, w.pin_hay, len, overlapping, codepoints, w.pin_out, &n, &done, true);
        }
        if (rc != ACX_OK) return rc;
        if (done) {
            if (n) {
                acx_match_t *m = (acx_match_t *)std::malloc(n * sizeof(acx_match_t));
                if (!m) return fail(ACX_ENOMEM, "out of memory");
                // (polled K0: the first matches ride in the result line, the others are in pin_out; all packed)
                // (the line: the copy run_small checked, not the pinned words themselves)
                auto carried = [&](uint64_t i) -> uint64_t { // (i < K0_LINES_MATCHES: from the lines' verified copies)
                    return i < ACX_K0_LINE_MATCHES ? w.h_lines[0][2 + i]
                                                   : w.h_lines[1 + (i - ACX_K0_LINE_MATCHES) / K0_MORE_MATCHES][1 + (i - ACX_K0_LINE_MATCHES) % K0_MORE_MATCHES];
                };
                volatile const uint64_t *rest = (volatile const uint64_t *)w.pin_out;
                if (!small_polls()) std::memcpy(m, w.pin_out, n * sizeof(acx_match_t));
                else {
                    // pin_out and the line are separate writes of the device to host memory: the line carries a hash of what
                    // pin_out must hold (k0_rest_mix); what is read here is taken when it agrees, read again when not
                    const uint32_t want = (uint32_t)(w.h_lines[0][1] >> K0_REST_HASH_SHIFT);
                    const uint64_t sq = c->small_seq;
                    const auto t0 = std::chrono::steady_clock::now();
                    bool synced = false; // the stream has been synchronised: what is read now is what the kernel wrote
                    for (;;) {
                        uint32_t hx = 0;
                        for (uint64_t i = 0; i < n; i++) {
                            const uint64_t v = i < K0_LINES_MATCHES ? carried(i) : rest[i - K0_LINES_MATCHES];
                            if (i >= K0_LINES_MATCHES) hx ^= k0_rest_mix(v, (uint32_t)(i - K0_LINES_MATCHES), sq);
                            m[i].pattern = v & 0xFFFFFFFFull; m[i].start = (v >> 32) & 0xFFFF; m[i].end = (v >> 48) + 1;
                        }
                        if (n <= K0_LINES_MATCHES || hx == want) break;
                        cpu_relax();
                        // (once the kernel is known to be over its writes have arrived: ONE more reading decides -- a hash that
                        // still disagrees is an error, not a reason to synchronise the stream a million times)
                        if (synced) { std::free(m); return fail(ACX_EDEVICE, "K0's matches did not arrive"); }
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(8)) {
                            if (hipStreamSynchronize(c->stream) != hipSuccess) { std::free(m); return fail(ACX_EDEVICE, "K0's matches did not arrive"); }
                            synced = true;
                        }
                    }
                }
                if (overlapping && a->expand_ov) { // (copies of a string: K0 reported the lowest ids -- expand_copies, on the host)
                    uint64_t total = 0;
                    for (uint64_t i = 0; i < n; i++) total += 1 + a->x_cnt[m[i].pattern];
                    if (total != n) {
                        acx_match_t *m2 = (acx_match_t *)std::malloc(total * sizeof(acx_match_t));
                        if (!m2) { std::free(m); return fail(ACX_ENOMEM, "out of memory"); }
                        uint64_t at = 0;
                        for (uint64_t i = 0; i < n; i++) {
                            m2[at++] = m[i];
                            const uint32_t *ids = a->x_ids.data() + a->x_off[m[i].pattern];
                            for (uint32_t q = 0; q < a->x_cnt[m[i].pattern]; q++) { m2[at] = m[i]; m2[at++].pattern = ids[q]; }
                        }
                        std::free(m);
                        m = m2;
                        n = total;
                    }
                }
                *out = m;
            }
            *n_out = n;
            return ACX_OK;
        }
    }
    stop_resident(c); // (a small call that turned out dense: the pipeline has the context to itself)
    g_trace_find.begin();
    // Mid-size haystacks IN PLACE (round 6): copied into pinned host memory by this thread and read from there by the scan
    // itself -- the runtime's copy of pageable memory is a staging copy of the same size PLUS a DMA the scan's launch waits
    // for (1 MiB: 26 us in the copy call, 21 us in the launch behind it, the DMA's own time before the scan starts).
    // Up to 1 MiB (same-box pairs, profiles/r06/exp_inplace_midsize_pairs.txt: 70 KB 45.2 -> 36.6 us, 128 KiB 49.9 -> 41.5,
    // 512 KiB 69.8 -> 60.5, 1 MiB 103.1 -> 93.0; 2 MiB 112 -> 134: beyond, this thread's copy is what the call waits for).
    // Only while the context's calls stay on the sparse path (a dense input is read several times: from HBM, then).
    static const uint64_t inplace_max = std::getenv("ACX_INPLACE_MAX") ? std::strtoull(std::getenv("ACX_INPLACE_MAX"), nullptr, 10) : (1ull << 20);
    const uint8_t *d_hay = nullptr;
    if (len <= inplace_max && a->kernel == ACX_KERNEL_PREFILTER && a->sparse_ok && c->dense_hold == 0 && !c->wide && c->spec_hot == 0) {
        Workspace &w = c->ws;
        if (w.pin_mid_cap < len + 4096) {
            if (w.pin_mid) (void)hipHostFree(w.pin_mid);
            w.pin_mid = nullptr; w.pin_mid_cap = 0;
            const uint64_t cap = std::max<uint64_t>(len + len / 4 + 4096, 1ull << 20);
            HIPCHK(hipHostMalloc((void **)&w.pin_mid, cap, hipHostMallocDefault));
            w.pin_mid_cap = cap;
        }
        std::memcpy(w.pin_mid, hay, len);
        std::memset(w.pin_mid + len, 0, 64);
        d_hay = w.pin_mid;
        rc = ACX_OK;
        a->path[11]++;
    } else {
        rc = stage_host(a, c, hay, len, nullptr, 0);
        d_hay = c->ws.hay;
    }
    g_trace_find.lap(0);
    if (rc == ACX_OK)
        rc = run_find(a, c, d_hay, len, Segments{nullptr, 1, 0}, overlapping, codepoints, &r, !try_small, false, 0, true);
    if (rc != ACX_OK) return rc;
    g_trace_find.lap(1);
    if (r->borrowed) {
        // the write kernel's records are in the context's pinned buffer: wait for the kernel (a few microseconds behind the
        // totals: polled, a blocking wait's wake-up costs more than the kernel), copy them out -- no device-to-host copy call
        if (r->done) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e;
            for (uint32_t spins = 0; (e = hipEventQuery(r->done)) == hipErrorNotReady; spins++) {
                cpu_relax();
                if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    e = hipEventSynchronize(r->done);
                    break;
                }
            }
            if (e != hipSuccess) { acx_free_result(r); return hipfail(e, "the write kernel"); }
        } else {
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        *n_out = r->n;
        if (r->n) {
            acx_match_t *m = (acx_match_t *)std::malloc(r->n * sizeof(acx_match_t));
            if (!m) { acx_free_result(r); return fail(ACX_ENOMEM, "out of memory"); }
            std::memcpy(m, r->d_matches, r->n * sizeof(acx_match_t));
            *out = m;
        }
        rc = ACX_OK;
    } else
    rc = download_matches(r, out, n_out); // waits for the call's device work: the staging buffer is free again
    g_trace_find.lap(2);
    acx_free_result(r);
    g_trace_find.lap(3);
    return rc;
}

void acx_free_matches(acx_match_t *m) {
    if (!m) return;
    if (!g_pinned_results.put(m)) std::free(m);
}

int acx_find_batch(acx_automaton_t *a, const uint8_t *hay, const uint64_t *offsets, uint64_t n_hay,
                   int overlapping, int codepoints, acx_match_t **out, uint64_t *n_out,
                   uint64_t *counts) {
    if (!a || !out || !n_out || !offsets) return fail(ACX_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    for (uint64_t i = 0; i < n_hay; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ACX_EINVAL, "offsets not monotone");
        if (counts) counts[i] = 0;
    }
    acx_result_t *r = nullptr;
    if (overlapping && a->host.match_kind != ACX_MATCH_STANDARD)
        return run_find(a, nullptr, nullptr, 0, Segments{nullptr, 1, 0}, overlapping, codepoints, &r, true, true);
    if (n_hay == 0) return ACX_OK;
    uint64_t base = offsets[0], len = offsets[n_hay] - base;
    std::vector<uint64_t> rel(n_hay + 1);
    for (uint64_t i = 0; i <= n_hay; i++) rel[i] = offsets[i] - base;
    Lease lease(a);
    Ctx *c = lease.c;
    if (!c) return fail(ACX_EDEVICE, "could not create a stream for the call");
    int rc = stage_host(a, c, hay ? hay + base : nullptr, len, rel.data(), n_hay + 1);
    if (rc == ACX_OK) {
        Segments G{c->ws.offsets, n_hay, 0};
        rc = run_find(a, c, c->ws.hay, len, G, overlapping, codepoints, &r, true, false);
    }
    if (rc != ACX_OK) return rc;
    rc = download_matches(r, out, n_out);
    if (counts && rc == ACX_OK) rc = acx_result_copy_counts(r, counts);
    if (rc != ACX_OK && *out) { acx_free_matches(*out); *out = nullptr; *n_out = 0; }
    acx_free_result(r);
    return rc;
}

int acx_replicate(const acx_automaton_t *a, int device, acx_automaton_t **out) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(ACX_EINVAL, "device ordinal out of range");
    const int saved = g_device;
    g_device = device;
    // (the host copy keeps the pattern bytes and offsets: the replica is compiled from them)
    const int rc = acx_build(a->host.blob.data(), a->host.offsets.data(), a->host.n_patterns, a->host.match_kind,
                             a->implementation, out);
    g_device = saved;
    if (rc == ACX_OK && a->kernel_forced) (void)acx_set_kernel(*out, a->kernel);
    return rc;
}

int acx_automaton_device(const acx_automaton_t *a) { return a ? a->device : -1; }

void acx_shard_range(uint64_t n_items, int shard, int n_shards, uint64_t *lo, uint64_t *hi) {
    if (n_shards <= 0 || shard < 0 || shard >= n_shards) { // no such shard: the empty range
        if (lo) *lo = 0;
        if (hi) *hi = 0;
        return;
    }
    const uint64_t base = n_items / (uint64_t)n_shards, extra = n_items % (uint64_t)n_shards;
    const uint64_t s = (uint64_t)shard;
    *lo = s * base + std::min<uint64_t>(s, extra);
    *hi = *lo + base + (s < extra ? 1 : 0);
}

int acx_find_batch_multi(acx_automaton_t *const *handles, int n_handles, const uint8_t *hay,
                         const uint64_t *offsets, uint64_t n_hay, int overlapping, int codepoints,
                         acx_match_t **out, uint64_t *n_out, uint64_t *counts) {
    if (!handles || n_handles < 1 || !out || !n_out || !offsets) return fail(ACX_EINVAL, "null argument");
    for (int i = 0; i < n_handles; i++)
        if (!handles[i]) return fail(ACX_EINVAL, "null automaton");
    if (n_handles == 1) return acx_find_batch(handles[0], hay, offsets, n_hay, overlapping, codepoints, out, n_out, counts);
    *out = nullptr; *n_out = 0;
    // one host thread per handle, each on its contiguous range of haystacks (acx_shard_range: the
    // same split as distributed.shard_range, so the concatenation over shards is the batch in
    // order); the only thing combined afterwards are the shards' match counts (their exclusive
    // prefix = where a shard's matches go in the output)
    struct Shard { acx_match_t *m = nullptr; uint64_t n = 0; int rc = ACX_OK; std::string err; uint64_t lo = 0, hi = 0; };
    std::vector<Shard> sh((size_t)n_handles);
    std::vector<std::thread> th;
    for (int i = 0; i < n_handles; i++) {
        acx_shard_range(n_hay, i, n_handles, &sh[i].lo, &sh[i].hi);
        th.emplace_back([&, i] {
            Shard &s = sh[(size_t)i];
            s.rc = acx_find_batch(handles[i], hay, offsets + s.lo, s.hi - s.lo, overlapping, codepoints, &s.m, &s.n,
                                  counts ? counts + s.lo : nullptr);
            if (s.rc != ACX_OK) s.err = acx_last_error(); // (thread-local: carried to the caller's thread)
        });
    }
    for (auto &t : th) t.join();
    int rc = ACX_OK;
    uint64_t total = 0;
    for (auto &s : sh) {
        if (s.rc != ACX_OK && rc == ACX_OK) rc = fail(s.rc, s.err);
        total += s.n;
    }
    acx_match_t *all = nullptr;
    if (rc == ACX_OK && total) {
        all = (acx_match_t *)std::malloc(total * sizeof(acx_match_t));
        if (!all) rc = fail(ACX_ENOMEM, "out of memory");
    }
    uint64_t at = 0;
    for (auto &s : sh) {
        if (rc == ACX_OK && s.n) std::memcpy(all + at, s.m, s.n * sizeof(acx_match_t));
        at += s.n;
        acx_free_matches(s.m);
    }
    if (rc != ACX_OK) return rc;
    *out = all;
    *n_out = total;
    return ACX_OK;
}

int acx_profile_enable(acx_automaton_t *a, int on) {
    if (!a) return fail(ACX_EINVAL, "null automaton");
    a->prof = on != 0;
    a->prof_every = on > 1 ? on : 1;
    return ACX_OK;
}

int acx_profile_read(acx_automaton_t *a, acx_profile_t *out, int reset) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    {
        // settle the post-stage time of every context that is not in use right now
        DeviceScope scope(a->device);
        std::vector<Ctx *> idle;
        {
            std::lock_guard<std::mutex> lk(a->pool_mu);
            idle.swap(a->idle);
        }
        for (Ctx *c : idle) { settle_scan_profile(a, c); settle_post_profile(a, c); }
        {
            std::lock_guard<std::mutex> lk(a->pool_mu);
            a->idle.insert(a->idle.end(), idle.begin(), idle.end());
        }
        a->pool_cv.notify_all();
    }
    std::lock_guard<std::mutex> lk(a->prof_mu);
    *out = a->profile;
    if (reset) a->profile = acx_profile_t{};
    return ACX_OK;
}

int acx_path_stats(acx_automaton_t *a, uint64_t out[ACX_PATH_STATS], int reset) {
    if (!a || !out) return fail(ACX_EINVAL, "null argument");
    for (int i = 0; i < ACX_PATH_STATS; i++) out[i] = reset ? a->path[i].exchange(0) : a->path[i].load();
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
int acx_device_synchronize_on(int device) {
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ACX_EINVAL, "device ordinal out of range");
    DeviceScope ds(device);
    HIPCHK(hipDeviceSynchronize());
    return ACX_OK;
}

int acx_generate_haystack(acx_automaton_t *a, void *d_dst, uint64_t len, int kind, uint64_t seed,
                          uint64_t stream_offset) {
    if (!a || (!d_dst && len)) return fail(ACX_EINVAL, "null argument");
    if (kind != 0 && kind != 1) return fail(ACX_EINVAL, "unknown haystack kind");
    if (kind == 1 && (stream_offset % 1024)) return fail(ACX_EINVAL, "stream_offset must be a multiple of 1024");
    Lease lease(a);
    if (!lease.c) return fail(ACX_EDEVICE, "could not create a stream for the call");
    HIPCHK(generate(a->dev, (uint8_t *)d_dst, len, kind, seed, stream_offset, lease.c->stream));
    HIPCHK(hipStreamSynchronize(lease.c->stream));
    return ACX_OK;
}

} // extern "C"
