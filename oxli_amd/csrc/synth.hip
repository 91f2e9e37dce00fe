// synth.hip -- device-side synthetic genome / read generator (see include/kct_synth.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kct_synth.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix64(u64 x) {
    u64 z = x + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__global__ void genome_kernel(unsigned char *g, u64 G, u64 seed_g) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += (u64)gridDim.x * blockDim.x)
        g[j] = (unsigned char)((0x54474341u >> (8 * (mix64(seed_g + j) & 3))) & 0xFF);  // "ACGT"
}

__global__ void reads_kernel(unsigned char *out, const unsigned char *__restrict__ genome, u64 G, u64 first, u64 count, unsigned L,
                             u64 seed_r, unsigned sub_ppm = 0, unsigned n_ppm = 0, u64 sorted_total = 0, u64 seed_e = 0) {
    const u64 stride = (u64)L + 1, total = count * stride;
    for (u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (u64)gridDim.x * blockDim.x) {
        const u64 r = idx / stride, j = idx - r * stride, i = first + r;
        unsigned char c = '\n';
        if (j < L) {
            const u64 start = sorted_total ? (i % sorted_total) * (G - L + 1) / sorted_total : mix64(seed_r + 2 * i) % (G - L + 1);
            if ((mix64(seed_r + 2 * i + 1) & 1) == 0) c = genome[start + j];
            else {
                const unsigned char b = genome[start + L - 1 - j];
                c = b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A';
            }
            if (sub_ppm | n_ppm) {  // the error model of include/kct_synth.h
                const u64 e = mix64(seed_e + i * (u64)L + j), u = e % 1000000ULL;
                if (u < n_ppm) c = 'N';
                else if (u < (u64)n_ppm + sub_ppm) {
                    const unsigned code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
                    c = (unsigned char)((0x54474341u >> (8 * ((code + 1u + (unsigned)((e >> 32) % 3ULL)) & 3u))) & 0xFFu);
                }
            }
        }
        out[idx] = c;
    }
}

}  // namespace

extern "C" int kct_synth_genome_device(void *d_genome, uint64_t G, uint64_t seed_g, void *stream) {
    if (!d_genome || !G) return 7;
    hipLaunchKernelGGL(genome_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (unsigned char *)d_genome, (u64)G, (u64)seed_g);
    return hipGetLastError() == hipSuccess ? 0 : 6;
}

extern "C" int kct_synth_reads_device(void *d_reads, const void *d_genome, uint64_t G, uint64_t first, uint64_t count, uint32_t L,
                                      uint64_t seed_r, void *stream) {
    if (!d_reads || !d_genome || G < L || !L) return 7;
    hipLaunchKernelGGL(reads_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (unsigned char *)d_reads,
                       (const unsigned char *)d_genome, (u64)G, (u64)first, (u64)count, (unsigned)L, (u64)seed_r);
    return hipGetLastError() == hipSuccess ? 0 : 6;
}

extern "C" int kct_synth_reads_device_ex(void *d_reads, const void *d_genome, uint64_t G, uint64_t first, uint64_t count, uint32_t L,
                                         uint64_t seed_r, uint32_t sub_ppm, uint32_t n_ppm, uint64_t sorted_total, uint64_t seed_e, void *stream) {
    if (!d_reads || !d_genome || G < L || !L || (uint64_t)sub_ppm + n_ppm > 1000000u) return 7;
    hipLaunchKernelGGL(reads_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (unsigned char *)d_reads,
                       (const unsigned char *)d_genome, (u64)G, (u64)first, (u64)count, (unsigned)L, (u64)seed_r, (unsigned)sub_ppm,
                       (unsigned)n_ppm, (u64)sorted_total, (u64)seed_e);
    return hipGetLastError() == hipSuccess ? 0 : 6;
}
