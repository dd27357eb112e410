#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <cstdint>
__global__ void k_flag(volatile uint64_t *f, uint64_t seq) { if (threadIdx.x == 0) { *f = seq; __threadfence_system(); } }
__global__ void k_work(const uint8_t *in, volatile uint64_t *f, uint64_t seq) {
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    uint32_t a = 0;
    for (int i = threadIdx.x; i < 300; i += blockDim.x) a += in[i];
    atomicAdd(&s, a);
    __syncthreads();
    if (threadIdx.x == 0) { f[1] = s; *f = seq; __threadfence_system(); }
}
int main() {
    uint64_t *f; uint8_t *pin;
    hipHostMalloc((void **)&f, 64, hipHostMallocCoherent);
    hipHostMalloc((void **)&pin, 4096, hipHostMallocDefault);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    f[0] = 0;
    auto run = [&](const char *name, int mode) {
        const int N = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= N; i++) {
            uint64_t seq = (uint64_t)mode * 1000000 + i;
            if (mode == 0) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, f, seq); while (*(volatile uint64_t *)f != seq) {} }
            if (mode == 1) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, f, seq); hipStreamSynchronize(st); }
            if (mode == 2) { pin[i & 255] = (uint8_t)i; hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, pin, f, seq); while (*(volatile uint64_t *)f != seq) {} }
        }
        auto t1 = std::chrono::steady_clock::now();
        printf("%-40s %.2f us per call\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    };
    run("warm", 0);
    run("launch + poll pinned flag", 0);
    run("launch + hipStreamSynchronize", 1);
    run("launch(work on pinned input) + poll", 2);
    return 0;
}
