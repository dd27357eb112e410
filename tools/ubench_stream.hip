// ubench_stream.hip -- streaming-read ceilings on MI355X for the tile shapes K1b uses.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const uint8_t *p) {
    if (NT) return __builtin_nontemporal_load((const u32x4 *)p);
    return *(const u32x4 *)p;
}

// classic grid-stride, 256-thread blocks
template <bool NT> __global__ void k_gridstride(const uint8_t *p, uint64_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16) {
        u32x4 v = ld<NT>(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) out[0] = acc;
}

// persistent, 1024-thread blocks, per-wave tiles of ROWS KiB, DEPTH tiles in flight
template <bool NT, int ROWS, int DEPTH, int LDSK> __global__ __launch_bounds__(1024) void k_tiles(const uint8_t *p, uint64_t n, uint32_t *out) {
    __shared__ uint32_t lds[LDSK * 256 + 4];
    if (threadIdx.x == 0) lds[0] = 0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t tile_bytes = (uint64_t)ROWS * 1024, ntiles = n / tile_bytes;
    const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x / 64) + wave, nw = (uint64_t)gridDim.x * (blockDim.x / 64);
    uint32_t acc = 0;
    u32x4 buf[DEPTH][ROWS];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        uint64_t t = gw + d * nw; if (t >= ntiles) t = ntiles - 1;
#pragma unroll
        for (int r = 0; r < ROWS; r++) buf[d][r] = ld<NT>(p + t * tile_bytes + r * 1024 + lane * 16);
    }
    for (uint64_t tile = gw; tile < ntiles; tile += nw * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) { u32x4 v = buf[d][r]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            uint64_t t = tile + (d + DEPTH) * nw; if (t >= ntiles) t = ntiles - 1;
#pragma unroll
            for (int r = 0; r < ROWS; r++) buf[d][r] = ld<NT>(p + t * tile_bytes + r * 1024 + lane * 16);
        }
    }
    if (acc == 0x12345678) out[0] = acc + lds[threadIdx.x & 3];
}

template <typename F> float timeit(F f, int iters = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < iters; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}

int main() {
    const uint64_t n = 1ull << 30;
    uint8_t *d; uint32_t *o;
    CK(hipMalloc(&d, n)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 1, n));
    auto rep = [&](const char *name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, n / ms / 1e6); };
    rep("gridstride 256x2048 plain", timeit([&] { k_gridstride<false><<<2048, 256>>>(d, n, o); }));
    rep("gridstride 256x2048 nt", timeit([&] { k_gridstride<true><<<2048, 256>>>(d, n, o); }));
    rep("gridstride 256x8192 nt", timeit([&] { k_gridstride<true><<<8192, 256>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth1 nt", timeit([&] { k_tiles<true, 4, 1, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth2 nt", timeit([&] { k_tiles<true, 4, 2, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth3 nt", timeit([&] { k_tiles<true, 4, 3, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows8 depth1 nt", timeit([&] { k_tiles<true, 8, 1, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows8 depth2 nt", timeit([&] { k_tiles<true, 8, 2, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth2 plain", timeit([&] { k_tiles<false, 4, 2, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth2 nt LDS150K", timeit([&] { k_tiles<true, 4, 2, 150><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x512 rows4 depth2 nt (2 blk/CU)", timeit([&] { k_tiles<true, 4, 2, 1><<<512, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth1 plain LDS150K (K1b's shape)", timeit([&] { k_tiles<false, 4, 1, 150><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows4 depth2 plain LDS150K", timeit([&] { k_tiles<false, 4, 2, 150><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows2 depth2 nt", timeit([&] { k_tiles<true, 2, 2, 1><<<256, 1024>>>(d, n, o); }));
    rep("tiles 1024x256 rows2 depth4 nt", timeit([&] { k_tiles<true, 2, 4, 1><<<256, 1024>>>(d, n, o); }));
    return 0;
}
