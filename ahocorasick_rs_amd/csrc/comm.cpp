// comm.cpp -- RCCL behind the C ABI (include/acx.h, "multi-GPU: the count exchange").
//
// The hot path shards by haystack with NO data-path collective; the one exchange it has is the
// per-rank match counts (8 bytes per rank) from which every rank derives where its matches go in
// the global output (the reference has no multi-device form at all -- its one
// call site is /root/reference/benchmarks/test_comparison.py:113-124).  Until round 4 that exchange
// existed only through torch.distributed (ahocorasick_rs_amd/distributed.py); a PyO3 host -- what
// BASELINE.json's north_star names -- has no torch.  These entry points speak RCCL directly:
//   * one process, n devices:   acx_comm_init_all   (ncclCommInitAll)
//   * one process per device:   acx_comm_unique_id on rank 0, the 128 bytes carried to the other
//                               ranks by whatever the host has (a file, a socket, MPI), then
//                               acx_comm_init_rank everywhere (ncclCommInitRank)
//   * acx_comm_allgather_counts (ncclAllGather of one u64 per rank, over xGMI inside a node)
// librccl is loaded on first use (dlopen): a host that never calls these functions needs no RCCL.  (A process
// that has imported a PyTorch wheel holds the wheel's own HIP / HSA runtime and RCCL under other SONAMEs next
// to /opt/rocm's; this communicator is for the hosts that have no torch -- tests/test_gpu_round4.py runs it in
// a process of its own.)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/acx.h"

extern "C" int acx_internal_fail(int code, const char *msg); // acx_api.cpp: sets acx_last_error of this thread

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.h) break;
        }
        if (!R.h) { R.err = std::string("librccl could not be loaded: ") + dlerror(); return; }
#define ACX_SYM(field, sym)                                                        \
        R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.h, sym));            \
        if (!R.field && R.err.empty()) R.err = std::string("librccl lacks ") + sym;
        ACX_SYM(GetUniqueId, "ncclGetUniqueId")
        ACX_SYM(CommInitRank, "ncclCommInitRank")
        ACX_SYM(CommInitAll, "ncclCommInitAll")
        ACX_SYM(CommDestroy, "ncclCommDestroy")
        ACX_SYM(AllGather, "ncclAllGather")
        ACX_SYM(GroupStart, "ncclGroupStart")
        ACX_SYM(GroupEnd, "ncclGroupEnd")
        ACX_SYM(GetErrorString, "ncclGetErrorString")
#undef ACX_SYM
    });
    return &R;
}

int nfail(Rccl *R, ncclResult_t r, const char *what) {
    return acx_internal_fail(ACX_EDEVICE, (std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(r) : "RCCL error")).c_str());
}
int hfail(hipError_t e, const char *what) {
    (void)hipGetLastError();
    return acx_internal_fail(e == hipErrorOutOfMemory ? ACX_ENOMEM : ACX_EDEVICE, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
}

struct DevScope {
    int prev = -1;
    explicit DevScope(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != d) (void)hipSetDevice(d); }
    ~DevScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

} // namespace

struct acx_comm {
    int world = 0;
    std::vector<int> devices;        // the ranks this process holds (init_all: all of them, init_rank: one)
    std::vector<int> ranks;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t *> d_send, d_recv;
};

namespace {

int alloc_rank_buffers(acx_comm *c) {
    for (size_t i = 0; i < c->devices.size(); i++) {
        DevScope ds(c->devices[i]);
        hipStream_t st = nullptr;
        uint64_t *s = nullptr, *r = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&s, 8);
        if (e == hipSuccess) e = hipMalloc((void **)&r, 8 * (size_t)c->world);
        c->streams.push_back(st); c->d_send.push_back(s); c->d_recv.push_back(r);
        if (e != hipSuccess) return hfail(e, "acx_comm: stream / buffers");
    }
    return ACX_OK;
}

} // namespace

extern "C" {

void acx_comm_free(acx_comm_t *c) {
    if (!c) return;
    Rccl *R = rccl();
    for (size_t i = 0; i < c->devices.size(); i++) {
        DevScope ds(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) { (void)hipStreamSynchronize(c->streams[i]); (void)hipStreamDestroy(c->streams[i]); }
        if (i < c->d_send.size()) (void)hipFree(c->d_send[i]);
        if (i < c->d_recv.size()) (void)hipFree(c->d_recv[i]);
        if (i < c->comms.size() && c->comms[i] && R->CommDestroy) (void)R->CommDestroy(c->comms[i]);
    }
    delete c;
}

int acx_comm_init_all(const int *devices, int n, acx_comm_t **out) {
    if (!out) return acx_internal_fail(ACX_EINVAL, "null output pointer");
    *out = nullptr;
    if (!devices || n < 1) return acx_internal_fail(ACX_EINVAL, "acx_comm_init_all: no devices");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return acx_internal_fail(ACX_EDEVICE, "no HIP device available");
    }
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return acx_internal_fail(ACX_EINVAL, "device ordinal out of range");
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return acx_internal_fail(ACX_EINVAL, "acx_comm_init_all: a device is listed twice");
    }
    Rccl *R = rccl();
    if (!R->err.empty()) return acx_internal_fail(ACX_EDEVICE, R->err.c_str());
    acx_comm *c = new (std::nothrow) acx_comm();
    if (!c) return acx_internal_fail(ACX_ENOMEM, "out of memory");
    c->world = n;
    c->devices.assign(devices, devices + n);
    for (int i = 0; i < n; i++) c->ranks.push_back(i);
    c->comms.assign((size_t)n, nullptr);
    ncclResult_t r = R->CommInitAll(c->comms.data(), n, devices);
    if (r != ncclSuccess) { acx_comm_free(c); return nfail(R, r, "ncclCommInitAll"); }
    const int rc = alloc_rank_buffers(c);
    if (rc != ACX_OK) { acx_comm_free(c); return rc; }
    *out = c;
    return ACX_OK;
}

int acx_comm_unique_id(uint8_t id[ACX_COMM_ID_BYTES]) {
    if (!id) return acx_internal_fail(ACX_EINVAL, "null id");
    static_assert(sizeof(ncclUniqueId) == ACX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return acx_internal_fail(ACX_EDEVICE, "no HIP device available");
    }
    Rccl *R = rccl();
    if (!R->err.empty()) return acx_internal_fail(ACX_EDEVICE, R->err.c_str());
    ncclUniqueId u;
    ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return nfail(R, r, "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof(u));
    return ACX_OK;
}

int acx_comm_init_rank(const uint8_t id[ACX_COMM_ID_BYTES], int world, int rank, int device, acx_comm_t **out) {
    if (!out) return acx_internal_fail(ACX_EINVAL, "null output pointer");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return acx_internal_fail(ACX_EINVAL, "acx_comm_init_rank: bad rank / world");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return acx_internal_fail(ACX_EDEVICE, "no HIP device available");
    }
    if (device < 0 || device >= ndev) return acx_internal_fail(ACX_EINVAL, "device ordinal out of range");
    Rccl *R = rccl();
    if (!R->err.empty()) return acx_internal_fail(ACX_EDEVICE, R->err.c_str());
    acx_comm *c = new (std::nothrow) acx_comm();
    if (!c) return acx_internal_fail(ACX_ENOMEM, "out of memory");
    c->world = world;
    c->devices.push_back(device);
    c->ranks.push_back(rank);
    c->comms.assign(1, nullptr);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r;
    {
        DevScope ds(device);
        r = R->CommInitRank(&c->comms[0], world, u, rank);
    }
    if (r != ncclSuccess) { acx_comm_free(c); return nfail(R, r, "ncclCommInitRank"); }
    const int rc = alloc_rank_buffers(c);
    if (rc != ACX_OK) { acx_comm_free(c); return rc; }
    *out = c;
    return ACX_OK;
}

int acx_comm_world(const acx_comm_t *c) { return c ? c->world : 0; }
int acx_comm_local_ranks(const acx_comm_t *c) { return c ? (int)c->devices.size() : 0; }

int acx_comm_allgather_counts(acx_comm_t *c, const uint64_t *local_counts, uint64_t *all_counts) {
    if (!c || !local_counts || !all_counts) return acx_internal_fail(ACX_EINVAL, "null argument");
    Rccl *R = rccl();
    const size_t nl = c->devices.size();
    for (size_t i = 0; i < nl; i++) {
        DevScope ds(c->devices[i]);
        hipError_t e = hipMemcpyAsync(c->d_send[i], &local_counts[i], 8, hipMemcpyHostToDevice, c->streams[i]);
        if (e != hipSuccess) return hfail(e, "acx_comm_allgather_counts: upload");
    }
    ncclResult_t r = R->GroupStart();
    for (size_t i = 0; i < nl && r == ncclSuccess; i++) {
        DevScope ds(c->devices[i]);
        r = R->AllGather(c->d_send[i], c->d_recv[i], 1, ncclUint64, c->comms[i], c->streams[i]);
    }
    const ncclResult_t r2 = R->GroupEnd();
    if (r != ncclSuccess) return nfail(R, r, "ncclAllGather");
    if (r2 != ncclSuccess) return nfail(R, r2, "ncclGroupEnd");
    for (size_t i = 0; i < nl; i++) { // (every local rank is waited for; the first one's copy is the answer)
        DevScope ds(c->devices[i]);
        hipError_t e = hipSuccess;
        if (i == 0) e = hipMemcpyAsync(all_counts, c->d_recv[0], 8 * (size_t)c->world, hipMemcpyDeviceToHost, c->streams[0]);
        if (e == hipSuccess) e = hipStreamSynchronize(c->streams[i]);
        if (e != hipSuccess) return hfail(e, "acx_comm_allgather_counts: download");
    }
    return ACX_OK;
}

void acx_output_offsets(const uint64_t *counts, int world, uint64_t *offsets) {
    uint64_t run = 0;
    for (int r = 0; r < world; r++) { offsets[r] = run; run += counts[r]; }
    offsets[world > 0 ? world : 0] = run;
}

} // extern "C"
