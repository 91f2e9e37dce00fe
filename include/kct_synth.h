/*
 * kct_synth.h -- synthetic read generator on the device (measurement infrastructure, exported
 * by the same libkct_hip.so).  Not part of the reference's interface: it exists so that
 * bench.py can put BASELINE.json's workloads in HBM without pushing 150 MB - 15 GB over PCIe,
 * byte-identical to what oracle/kct_oracle.c (orc_synth_genome / orc_synth_reads) makes on the
 * CPU (tests/test_gpu_parity.py checks that).
 *
 *   genome[j] = "ACGT"[ mix64(seed_g + j) & 3 ]
 *   read i    : start  = mix64(seed_r + 2i)     mod (G - L + 1)
 *               strand = mix64(seed_r + 2i + 1) & 1      (1 = reverse complement)
 *               L bases followed by one '\n'  (stride L + 1)
 *   mix64(x)  = splitmix64 output function of (x + 0x9e3779b97f4a7c15)
 */
#ifndef KCT_SYNTH_H
#define KCT_SYNTH_H
#include <stddef.h>
#include <stdint.h>
/* Only the entry points below are exported from libkct_hip.so (it is built with -fvisibility=hidden). */
#if defined(KCT_BUILDING_LIBRARY)
#define KCT_API __attribute__((visibility("default")))
#else
#define KCT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* d_genome: device buffer of G bytes.  stream: hipStream_t as void* (NULL = default stream). Returns 0 on success. */
KCT_API int kct_synth_genome_device(void *d_genome, uint64_t G, uint64_t seed_g, void *stream);

/* d_reads: device buffer of count * (L + 1) bytes, reads [first, first + count) of the stream. */
KCT_API int kct_synth_reads_device(void *d_reads, const void *d_genome, uint64_t G, uint64_t first, uint64_t count, uint32_t L,
                           uint64_t seed_r, void *stream);

/* The same stream with a sequencing-error model (SURVEY.md 8d's secondary inputs), byte-identical to
 * oracle/kct_oracle.c orc_synth_reads_ex:
 *   e = mix64(seed_e + i * L + j) for base j of read i;  u = e mod 10^6
 *   u < n_ppm            -> 'N'
 *   u < n_ppm + sub_ppm  -> "ACGT"[(code + 1 + (e >> 32) mod 3) & 3]      (a substitution: never the true base)
 *   sorted_total > 0     -> start = (i mod sorted_total) * (G - L + 1) / sorted_total   (position-sorted reads) */
KCT_API int kct_synth_reads_device_ex(void *d_reads, const void *d_genome, uint64_t G, uint64_t first, uint64_t count, uint32_t L,
                              uint64_t seed_r, uint32_t sub_ppm, uint32_t n_ppm, uint64_t sorted_total, uint64_t seed_e, void *stream);

#ifdef __cplusplus
}
#endif
#endif
