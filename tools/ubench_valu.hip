// ubench_valu.hip -- issue cost of the integer VALU ops K1b level 1 can be built from (MI355X).
// Every op is forced with inline asm; 8 independent chains per lane, 16 waves per CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS(X) \
  X(0, "v_lshrrev_b32",   "v_lshrrev_b32 %0, 5, %1") \
  X(1, "v_and_b32",       "v_and_b32 %0, 0x7fffff7, %1") \
  X(2, "v_xor_b32",       "v_xor_b32 %0, %2, %1") \
  X(3, "v_add_u32",       "v_add_u32 %0, %2, %1") \
  X(4, "v_alignbit_b32",  "v_alignbit_b32 %0, %1, %2, 1") \
  X(5, "v_alignbyte_b32", "v_alignbyte_b32 %0, %1, %2, 1") \
  X(6, "v_mad_u32_u24",   "v_mad_u32_u24 %0, %1, %2, %1") \
  X(7, "v_mul_u32_u24",   "v_mul_u32_u24 %0, %1, %2") \
  X(8, "v_mul_lo_u32",    "v_mul_lo_u32 %0, %1, %2") \
  X(9, "v_perm_b32",      "v_perm_b32 %0, %1, %2, %2") \
  X(10, "v_bfe_u32",      "v_bfe_u32 %0, %1, 3, 17") \
  X(11, "v_lshl_or_b32",  "v_lshl_or_b32 %0, %1, 3, %2") \
  X(12, "v_and_or_b32",   "v_and_or_b32 %0, %1, %2, %2") \
  X(13, "v_lshl_add_u32", "v_lshl_add_u32 %0, %1, 3, %2") \
  X(14, "v_xad_u32",      "v_xad_u32 %0, %1, %2, %2") \
  X(15, "v_cndmask_b32",  "v_cndmask_b32 %0, %1, %2, vcc") \
  X(16, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(17, "v_mov_b32 dpp wave_shl:1", "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf") \
  X(18, "v_lshrrev_b64",  "v_lshrrev_b64 %0, 5, %1") \
  X(19, "v_bfi_b32",      "v_bfi_b32 %0, %2, %1, %1") \
  X(20, "v_lshlrev_b32 (var)", "v_lshlrev_b32 %0, %2, %1") \
  X(21, "v_or3_b32",      "v_or3_b32 %0, %1, %2, %2") \
  X(22, "v_sad_u32",      "v_sad_u32 %0, %1, %2, %2") \
  X(23, "v_mbcnt_lo",     "v_mbcnt_lo_u32_b32 %0, %2, %1")

template <int OP> __global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8]; uint32_t y = threadIdx.x * 0x9E3779u + seed;
    uint64_t w[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * (2 * i + 3) + seed; w[i] = a[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
#define X(ID, NAME, ASM) if (OP == ID) { if (ID == 18) asm volatile(ASM : "=v"(w[i]) : "v"(w[i]), "v"(y)); else asm volatile(ASM : "=v"(a[i]) : "v"(a[i]), "v"(y)); }
            OPS(X)
#undef X
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= a[i] ^ (uint32_t)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> float run(uint32_t *o, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<256, 1024>>>(o, iters, 1); hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<256, 1024>>>(o, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t *o; hipMalloc(&o, 256 * 1024 * 4);
    const int iters = 20000;
#define X(ID, NAME, ASM) { float ms = run<ID>(o, iters); double wi = 16.0 * iters * 16 / 4; /* wave-instr per SIMD */ \
        printf("%-26s %.3f ms   %.2f clk per wave-instr per SIMD (at 2.4 GHz)\n", NAME, ms, ms * 1e-3 * 2.4e9 / wi); }
    OPS(X)
#undef X
    return 0;
}
