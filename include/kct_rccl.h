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
 *                kct_consume_device_routed(table, ..., world, rank, kct_rccl_ops(x), 0, &n, stats)
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
/* bytes this communicator has sent to / received from OTHER ranks, and seconds spent blocked in wait() (statistics) */
void kct_rccl_stats(const kct_rccl *x, uint64_t *bytes_sent, uint64_t *bytes_received, double *wait_seconds);

#ifdef __cplusplus
}
#endif
#endif /* KCT_RCCL_H */
