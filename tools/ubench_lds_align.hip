// ubench_lds_align.hip -- what does ds_read_b64 / ds_read_b32 return at an address that is not naturally aligned (gfx950)?
// K1b's level 1 masks the low three bits of every table address (one v_and per pair of positions): if the LDS ignores
// them the mask is dead code; if it honours them (an unaligned read) an LDS-staged tile can be read at byte offsets.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_align.hip -o tools/ubench_lds_align.bin && tools/ubench_lds_align.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(uint64_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) buf[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)buf; // LDS byte address
    if (threadIdx.x < 16) {
        const uint32_t a = base + 64 + threadIdx.x;
        uint64_t v64; uint32_t v32;
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
        out[2 * threadIdx.x] = v64;
        out[2 * threadIdx.x + 1] = v32;
    }
}
int main() {
    uint64_t *d, h[32];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++)
        printf("offset %2d: ds_read_b64 -> first byte %3u (%s), bytes %016llx   ds_read_b32 -> first byte %3u\n", i,
               (unsigned)(h[2 * i] & 0xFF), (h[2 * i] & 0xFF) == (unsigned)(64 + i) ? "UNALIGNED read honoured" :
               (h[2 * i] & 0xFF) == (unsigned)(64 + (i & ~7)) ? "low bits ignored" : "other",
               (unsigned long long)h[2 * i], (unsigned)(h[2 * i + 1] & 0xFF));
    return 0;
}
