/* A plain-C caller of the early multi-GPU route with the RCCL exchange (include/kct_rccl.h), world = 1: the communicator, the size
 * all-to-all, the asynchronous payload all-to-all (ncclSend / ncclRecv to itself on the helper's stream) and the owner-side count all
 * run as they do on N GPUs -- only the peers are missing; then the LATE route's kct_rccl_merge_across_ranks on the directly counted
 * table.  Prints the digests of the routed table and of a table that counted the same
 * reads directly; tests/test_gpu_api.py compares them (and both with the CPU oracle's).
 *
 *   c_rccl_example <k> <reads> <read length> <genome> <passes>
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "kct.h"
#include "kct_rccl.h"
#include "kct_synth.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s / %s\n", #x, kct_last_error(), kct_rccl_last_error()); return 1; } } while (0)

static int report(const char *name, kct_table *t, uint64_t n) {
    uint64_t len, sum, shc, xhc, ssq;
    CHECK(kct_len(t, &len));
    CHECK(kct_sum_counts(t, &sum));
    CHECK(kct_digest(t, &shc, &xhc, &ssq));
    printf("%s n %llu len %llu sum %llu digests %llu %llu %llu\n", name, (unsigned long long)n, (unsigned long long)len, (unsigned long long)sum,
           (unsigned long long)shc, (unsigned long long)xhc, (unsigned long long)ssq);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const int k = atoi(argv[1]);
    const uint64_t R = strtoull(argv[2], 0, 10), L = strtoull(argv[3], 0, 10), G = strtoull(argv[4], 0, 10), passes = strtoull(argv[5], 0, 10);
    const uint64_t nbytes = R * (L + 1);
    void *d_genome = 0, *d_reads = 0;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d_genome, G) != hipSuccess || hipMalloc(&d_reads, nbytes + 16) != hipSuccess) return 3;
    CHECK(kct_synth_genome_device(d_genome, G, 42, 0));
    CHECK(kct_synth_reads_device(d_reads, d_genome, G, 0, R, (uint32_t)L, 1337, 0));
    if (hipDeviceSynchronize() != hipSuccess) return 3;

    kct_table *plain = 0, *routed = 0;
    uint64_t n = 0;
    CHECK(kct_create((uint8_t)k, G, 0, &plain));
    CHECK(kct_consume_device(plain, d_reads, nbytes, R * L, &n));
    if (report("plain", plain, n)) return 1;

    unsigned char id[KCT_RCCL_ID_BYTES];
    kct_rccl *x = 0;
    CHECK(kct_rccl_unique_id(id));
    CHECK(kct_rccl_create(id, 1, 0, 0, &x));
    CHECK(kct_create((uint8_t)k, G, 0, &routed));
    uint64_t stats[16];
    const uint64_t windows = nbytes - (uint64_t)k + 1, per_pass = ((windows + passes - 1) / passes + 0xFFFF) & ~0xFFFFULL;
    CHECK(kct_consume_device_routed(routed, d_reads, nbytes, R * L, 1, 0, kct_rccl_ops(x), per_pass, &n, stats));
    if (report("routed", routed, n)) return 1;
    uint64_t sent, received;
    double wait_s;
    kct_rccl_stats(x, &sent, &received, &wait_s);
    printf("passes %llu runs %llu counts %llu sent_to_others %llu\n", (unsigned long long)stats[5], (unsigned long long)stats[4], (unsigned long long)stats[11],
           (unsigned long long)sent);
    /* the LATE route's collective (kct_rccl_merge_across_ranks) on the directly counted table: with one rank every pair is sent to
     * itself through the size rounds and the ncclSend / ncclRecv group, the table is cleared, resized and refilled -- same digests */
    uint64_t got = 0, len_before = 0, consumed_before = 0, consumed_after = 0;
    CHECK(kct_len(plain, &len_before));
    CHECK(kct_count_hash(plain, 0, &got));            /* key 0 lives beside the device table and rides on the size round */
    CHECK(kct_consumed(plain, &consumed_before));
    kct_rccl_merge_when_alone(x, 1);
    CHECK(kct_rccl_merge_across_ranks(x, plain, &got));
    CHECK(kct_consumed(plain, &consumed_after));
    uint64_t zero_count = 0;
    CHECK(kct_get_hash(plain, 0, &zero_count));
    printf("merge pairs_received %llu len_before %llu zero %llu consumed_kept %d\n", (unsigned long long)got, (unsigned long long)len_before,
           (unsigned long long)zero_count, consumed_before == consumed_after);
    CHECK(kct_remove_hash(plain, 0, &got));
    if (report("merged", plain, n)) return 1;
    kct_destroy(routed);
    kct_destroy(plain);
    kct_rccl_destroy(x);
    (void)hipFree(d_reads); (void)hipFree(d_genome);
    return 0;
}
