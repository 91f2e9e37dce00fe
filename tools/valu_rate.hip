// valu_rate.hip -- micro-benchmark: issue cost of the integer VALU ops the k-mer hash uses on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32; typedef unsigned long long u64;
#define REP 256
#define OPS(NAME, BODY)                                                                     \
    __global__ void NAME(u32 *out, int iters) {                                             \
        u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        u32 b = out[0] | 0x9e3779b9u;                                                       \
        for (int it = 0; it < iters; ++it) {                                                \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { BODY }                    \
        }                                                                                   \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }
#define EIGHT(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(b));
#define ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define CNDM(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
OPS(k_add, EIGHT(ADD))
OPS(k_mullo, EIGHT(MULLO))
OPS(k_mulhi, EIGHT(MULHI))
OPS(k_mul24, EIGHT(MUL24))
OPS(k_xor, EIGHT(XOR))
OPS(k_perm, EIGHT(PERM))
OPS(k_align, EIGHT(ALIGN))
OPS(k_add3, EIGHT(ADD3))
__global__ void k_mad64(u32 *out, int iters) {
    u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = out[0] | 0x9e3779b9u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#define MAD(x) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0" : "+v"(x) : "v"(b) : "s10", "s11");
            EIGHT(MAD)
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_lshladd64(u32 *out, int iters) {
    u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u64 b = out[0] | 0x9e3779b9u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#define LA(x) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(x) : "v"(b));
            EIGHT(LA)
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
template <class K> void run(const char *name, K kern, u32 *d, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * 4, threads = 64 * waves_per_simd;  // 4 blocks per CU -> one per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    // each SIMD runs `waves_per_simd` waves x iters x REP instructions
    double instr_per_simd = (double)waves_per_simd * iters * REP;
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    u32 *d; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24);
    for (int w : {1, 4}) {
        run("v_add_u32", k_add, d, w); run("v_xor_b32", k_xor, d, w); run("v_mul_lo_u32", k_mullo, d, w); run("v_mul_hi_u32", k_mulhi, d, w);
        run("v_mul_u32_u24", k_mul24, d, w); run("v_perm_b32", k_perm, d, w); run("v_alignbit", k_align, d, w); run("v_add3_u32", k_add3, d, w);
        run("v_mad_u64_u32", k_mad64, d, w); run("v_lshl_add_u64", k_lshladd64, d, w);
    }
    return 0;
}
