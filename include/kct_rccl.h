/*
 * kct_rccl.h -- kct_exchange_ops over RCCL (libkct_rccl.so): the collective that kct_consume_device_routed (include/kct.h) asks its
 * caller for, implemented with ncclSend / ncclRecv groups on a stream of its own, so that the early multi-GPU route runs behind the
 * C ABI with no Python in the process -- what a Rust KmerCountTable links.  One communicator per process (one
 * process per GPU); the exchange moves over xGMI between the GPUs of a node.
 *
 * A separate small library on purpose: libkct_hip.so does not depend on RCCL, and a process that already carries another copy of
 * RCCL (PyTorch ships its own) uses its own collective instead (oxli_amd/distributed.py does, through torch.distributed).
 *
 *   rank 0:      kct_rccl_unique_id(id)            -> 128 bytes, handed to every rank out of band (file, socket, MPI, env)
 *   every rank:  kct_rccl_create(id, world, rank, device, &x)
 *                kct_consume_device_routed(table, ..., world, rank, kct_rccl_ops(x), 0, &n, stats)        (the EARLY route)
 *          or    kct_consume_device(table, own records ...); kct_rccl_merge_across_ranks(x, table, &got)  (the LATE route)
 *                kct_rccl_destroy(x)
 */
#ifndef KCT_RCCL_H
#define KCT_RCCL_H

#include "kct.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KCT_RCCL_ID_BYTES 128

typedef struct kct_rccl kct_rccl;

/* All return 0 on success; kct_rccl_last_error() holds the text of the last failure on this thread. */
int kct_rccl_unique_id(void *id128);
int kct_rccl_create(const void *id128, int world, int rank, int device, kct_rccl **out);
const kct_exchange_ops *kct_rccl_ops(kct_rccl *x);
void kct_rccl_destroy(kct_rccl *x);
const char *kct_rccl_last_error(void);
/* The LATE route's collective -- BASELINE.json north_star's "final RCCL reduce of per-bucket counts"; the reference's add(),
 * lib.rs:778-837 (per-key sum of counts), applied across ranks.  Every rank has counted its OWN records into its own table; every rank
 * calls this; afterwards rank r's table holds exactly the keys of hash slice r -- owner(hash) = floor(hi32(hash) * world / 2^32) --
 * with their global counts, resized for that slice; the global table is the disjoint union over ranks (its len / sum_counts /
 * consumed are sums over ranks; `consumed` of each table stays that rank's own share).  One size round (status, pair counts, the count
 * of key 0 -- kept beside the device table -- to its owner, rank 0), one status round once every rank has room, one ncclSend /
 * ncclRecv group of 16-byte {hash, count} pairs, one status round after every rank's refill.  A failure on any rank before the payload
 * ends the call on EVERY rank with the tables unchanged; a failure after it (the refill) ends it on every rank with the tables'
 * contents UNDEFINED -- clear them.  (An RCCL call that fails on one rank once its peers may have enqueued their halves aborts that
 * rank's communicator: every later call on it fails at once; the peers are ended by the launcher's hang guard.)  *pairs_received (may be NULL) = pairs this rank received, its own included.  A world of one returns at once
 * unless kct_rccl_merge_when_alone(x, 1) asks for the collectives anyway (tests on a one-GPU box). */
int kct_rccl_merge_across_ranks(kct_rccl *x, kct_table *t, uint64_t *pairs_received);
void kct_rccl_merge_when_alone(kct_rccl *x, int on);
/* The merge keeps its two pair buffers (16 B per pair sent / received, + 1/8) between merges.  kct_rccl_release_buffers frees them now;
 * kct_rccl_release_above(x, bytes) makes every merge free them afterwards when one has grown beyond `bytes` (default 4 GiB, 0 = never). */
void kct_rccl_release_buffers(kct_rccl *x);
void kct_rccl_release_above(kct_rccl *x, uint64_t bytes);
/* bytes this communicator has sent to / received from OTHER ranks, and seconds spent blocked in wait() (statistics) */
void kct_rccl_stats(const kct_rccl *x, uint64_t *bytes_sent, uint64_t *bytes_received, double *wait_seconds);

#ifdef __cplusplus
}
#endif
#endif /* KCT_RCCL_H */
