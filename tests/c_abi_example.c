/* c_abi_example.c -- drives libkct_hip.so from plain C through include/kct.h, exactly as a foreign-language
 * binding (the Rust crate's extern "C" block in INTEGRATION.md) would.  tests/test_gpu_api.py compiles it
 * with gcc, runs it on the GPU box and compares its output with the CPU oracle.
 *
 *   usage: c_abi_example <ksize> <sequence>
 *   prints: n, len, sum_counts, consumed, then "hash count" lines sorted by hash
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kct.h"

#define CHECK(call)                                                               \
    do {                                                                          \
        kct_status st_ = (call);                                                  \
        if (st_ != KCT_OK) {                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, kct_last_error()); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    const uint8_t k = (uint8_t)atoi(argv[1]);
    const char *seq = argv[2];
    kct_table *t = NULL, *u = NULL;
    CHECK(kct_create(k, 0, 0, &t));
    CHECK(kct_create(k, 0, 0, &u));
    uint64_t n = 0, n2 = 0, len = 0, sum = 0, consumed = 0, added = 0, fresh = 0;
    CHECK(kct_consume(t, seq, strlen(seq), 1, &n));
    /* the same record through the batch entry point into a second table, then add() it */
    uint64_t offsets[2] = {0, strlen(seq)};
    CHECK(kct_consume_batch(u, seq, offsets, 1, 1, &n2, NULL, NULL));
    CHECK(kct_add(t, u, &added, &fresh));
    CHECK(kct_len(t, &len));
    CHECK(kct_sum_counts(t, &sum));
    CHECK(kct_consumed(t, &consumed));
    printf("n %llu %llu\nadded %llu new %llu\nlen %llu\nsum %llu\nconsumed %llu\n", (unsigned long long)n, (unsigned long long)n2,
           (unsigned long long)added, (unsigned long long)fresh, (unsigned long long)len, (unsigned long long)sum,
           (unsigned long long)consumed);
    uint64_t *hashes = malloc((len ? len : 1) * sizeof *hashes), *counts = malloc((len ? len : 1) * sizeof *counts), got = 0;
    CHECK(kct_dump(t, hashes, counts, len, 1, &got));
    for (uint64_t i = 0; i < got; ++i) printf("%llu %llu\n", (unsigned long long)hashes[i], (unsigned long long)counts[i]);
    /* error mode: the first bad window raises with the number of k-mers counted before it */
    uint64_t before = 0;
    kct_status st = kct_consume(u, "ACGTNACGT", 9, 0, &before);
    printf("error_mode status %d position %llu\n", (int)st, (unsigned long long)before);
    free(hashes); free(counts);
    kct_destroy(t); kct_destroy(u);
    return 0;
}
