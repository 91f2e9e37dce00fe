// kct_rccl.cpp -- kct_exchange_ops (include/kct.h) over RCCL: libkct_rccl.so (include/kct_rccl.h).
//
// The early multi-GPU route needs three things of its caller: device buffers, a small all-to-all of sizes in host memory, and an
// asynchronous all-to-all of byte ranges.  Here they are hipMalloc, and ncclSend / ncclRecv pairs inside one group on a stream that
// belongs to the communicator -- the payload moves while the library's kernels run on the table's stream; wait() is a
// hipStreamSynchronize of that stream.  xGMI is point to point: a group of world - 1 sends and receives uses every link at once.
//
// kct_rccl_merge_across_ranks is the LATE route's collective in the same library: private tables -> owner-bucketed {hash, count} pairs
// (kct_export_by_owner_device) -> one size round + one payload group over the same communicator -> every owner folds what arrives into
// a table resized for its slice (kct_merge_pairs_device); add()'s semantics, lib.rs:778-837.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/kct_rccl.h"

namespace {

thread_local char g_err[512];
void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err("%s: %s", #expr, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { set_err("%s: %s", #expr, ncclGetErrorString(r_)); return 1; } } while (0)

}  // namespace

struct kct_rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int world = 0, rank = 0, device = 0;
    uint64_t *d_sizes = nullptr;     // 2 x world x nvals uint64: what goes out, what comes in
    size_t sizes_cap = 0;
    uint64_t sent = 0, received = 0;
    double wait_s = 0;
    bool merge_when_alone = false;   // a world of one goes through the merge's collectives too (tests on a one-GPU box)
    void *m_send = nullptr, *m_recv = nullptr;   // the late route's pair buffers, kept between merges (grow-only: no hipMalloc / hipFree per job)
    uint64_t m_send_cap = 0, m_recv_cap = 0;
    uint64_t release_above = 4ULL << 30;         // a merge frees its pair buffers afterwards when one has grown beyond this (0 = never)
    kct_exchange_ops ops;
};

namespace {

void *x_alloc(void *user, uint64_t bytes) {
    kct_rccl *x = (kct_rccl *)user;
    void *p = nullptr;
    if (hipSetDevice(x->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { set_err("hipMalloc of %llu bytes failed", (unsigned long long)bytes); return nullptr; }
    return p;
}

void x_release(void *user, void *p) {
    kct_rccl *x = (kct_rccl *)user;
    (void)hipStreamSynchronize(x->stream);
    (void)hipFree(p);
}

// a grow-only device buffer of the communicator (the merge's pair buffers)
void *x_keep(kct_rccl *x, void **p, uint64_t *cap, uint64_t bytes) {
    if (bytes <= *cap) return *p;
    if (*p) { (void)hipStreamSynchronize(x->stream); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const uint64_t want = bytes + bytes / 8;
    if (hipSetDevice(x->device) != hipSuccess || hipMalloc(p, want) != hipSuccess) { set_err("hipMalloc of %llu bytes failed", (unsigned long long)want); *p = nullptr; return nullptr; }
    *cap = want;
    return *p;
}

// A collective that fails on THIS rank after its peers may have enqueued their halves cannot be agreed away (their receives from this
// rank never complete): the open group is closed and the communicator ABORTED, so that every later call on it fails at once instead of
// enqueueing behind a dead operation.  The peers' pending operations end when their own RCCL notices (or the launcher's hang guard
// does); kct_consume_device_routed's "every rank returns an error" holds for failures the protocol can carry, not for this one.
int x_broken(kct_rccl *x, bool group_open, const char *what, ncclResult_t r) {
    set_err("%s: %s -- the communicator is aborted", what, ncclGetErrorString(r));
    if (group_open) (void)ncclGroupEnd();
    if (x->comm) { (void)ncclCommAbort(x->comm); x->comm = nullptr; }
    return 1;
}
#define NCCL_GROUP(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return x_broken(x, true, #expr, r_); } while (0)

int x_sizes(void *user, const uint64_t *send, uint32_t nvals, uint64_t *recv) {
    kct_rccl *x = (kct_rccl *)user;
    const size_t n = (size_t)x->world * nvals;
    if (!x->comm) { set_err("the communicator was aborted by an earlier failure"); return 1; }
    HIP_OK(hipSetDevice(x->device));
    if (2 * n > x->sizes_cap) {
        if (x->d_sizes) HIP_OK(hipFree(x->d_sizes));
        x->d_sizes = nullptr; x->sizes_cap = 0;
        HIP_OK(hipMalloc((void **)&x->d_sizes, 2 * n * sizeof(uint64_t)));
        x->sizes_cap = 2 * n;
    }
    HIP_OK(hipMemcpyAsync(x->d_sizes, send, n * sizeof(uint64_t), hipMemcpyHostToDevice, x->stream));
    ncclResult_t g = ncclGroupStart();
    if (g != ncclSuccess) return x_broken(x, false, "ncclGroupStart", g);
    for (int r = 0; r < x->world; ++r) {
        NCCL_GROUP(ncclSend(x->d_sizes + (size_t)r * nvals, nvals, ncclUint64, r, x->comm, x->stream));
        NCCL_GROUP(ncclRecv(x->d_sizes + n + (size_t)r * nvals, nvals, ncclUint64, r, x->comm, x->stream));
    }
    g = ncclGroupEnd();
    if (g != ncclSuccess) return x_broken(x, false, "ncclGroupEnd", g);
    HIP_OK(hipMemcpyAsync(recv, x->d_sizes + n, n * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
    HIP_OK(hipStreamSynchronize(x->stream));
    return 0;
}

int x_start(void *user, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes) {
    kct_rccl *x = (kct_rccl *)user;
    if (!x->comm) { set_err("the communicator was aborted by an earlier failure"); return 1; }
    HIP_OK(hipSetDevice(x->device));
    ncclResult_t g = ncclGroupStart();
    if (g != ncclSuccess) return x_broken(x, false, "ncclGroupStart", g);
    for (int r = 0; r < x->world; ++r) {
        if (send_bytes[r]) NCCL_GROUP(ncclSend((const char *)d_send + send_off[r], send_bytes[r], ncclUint8, r, x->comm, x->stream));
        if (recv_bytes[r]) NCCL_GROUP(ncclRecv((char *)d_recv + recv_off[r], recv_bytes[r], ncclUint8, r, x->comm, x->stream));
        if (r != x->rank) { x->sent += send_bytes[r]; x->received += recv_bytes[r]; }
    }
    g = ncclGroupEnd();
    if (g != ncclSuccess) return x_broken(x, false, "ncclGroupEnd", g);
    return 0;
}

int x_wait(void *user) {
    kct_rccl *x = (kct_rccl *)user;
    const double t0 = now_s();
    HIP_OK(hipStreamSynchronize(x->stream));
    x->wait_s += now_s() - t0;
    return 0;
}

}  // namespace

extern "C" {

const char *kct_rccl_last_error(void) { return g_err; }

int kct_rccl_unique_id(void *id128) {
    static_assert(sizeof(ncclUniqueId) <= KCT_RCCL_ID_BYTES, "ncclUniqueId does not fit KCT_RCCL_ID_BYTES");
    if (!id128) { set_err("null argument"); return 1; }
    ncclUniqueId id;
    NCCL_OK(ncclGetUniqueId(&id));
    memset(id128, 0, KCT_RCCL_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return 0;
}

int kct_rccl_create(const void *id128, int world, int rank, int device, kct_rccl **out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { set_err("bad argument"); return 1; }
    *out = nullptr;
    kct_rccl *x = new (std::nothrow) kct_rccl;
    if (!x) { set_err("out of memory"); return 1; }
    x->world = world; x->rank = rank; x->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess) { set_err("no stream on device %d", device); delete x; return 1; }
    const ncclResult_t r = ncclCommInitRank(&x->comm, world, id, rank);
    if (r != ncclSuccess) { set_err("ncclCommInitRank: %s", ncclGetErrorString(r)); (void)hipStreamDestroy(x->stream); delete x; return 1; }
    x->ops.user = x;
    x->ops.alloc = x_alloc; x->ops.release = x_release; x->ops.exchange_sizes = x_sizes; x->ops.start = x_start; x->ops.wait = x_wait;
    *out = x;
    return 0;
}

const kct_exchange_ops *kct_rccl_ops(kct_rccl *x) { return x ? &x->ops : nullptr; }

void kct_rccl_stats(const kct_rccl *x, uint64_t *bytes_sent, uint64_t *bytes_received, double *wait_seconds) {
    if (!x) return;
    if (bytes_sent) *bytes_sent = x->sent;
    if (bytes_received) *bytes_received = x->received;
    if (wait_seconds) *wait_seconds = x->wait_s;
}

// The late route (BASELINE.json north_star: "a final RCCL reduce of per-bucket counts"): every rank has counted its own records into its
// own table; afterwards rank r holds every key of hash slice r -- owner(hash) = floor(hi32(hash) * world / 2^32) -- with its GLOBAL
// count: add() (lib.rs:778-837: per-key sum) applied across ranks.  `consumed` stays this rank's own share; len / sum_counts / consumed
// of the global table are sums over ranks.  Collective: every rank must call it; a failure on any rank is learnt by all in the size
// round (slot 0 = status) or the second one-word round, before any payload moves (tables unchanged), or in the third one-word round
// after the refill (tables undefined: clear them).  *pairs_received (may be NULL) = pairs this rank received, its own included.
int kct_rccl_merge_across_ranks(kct_rccl *x, kct_table *t, uint64_t *pairs_received) {
    if (!x || !t) { set_err("null argument"); return 1; }
    if (pairs_received) *pairs_received = 0;
    const int world = x->world;
    if (world == 1 && !x->merge_when_alone) return 0;
    uint64_t status = 0;    // this rank's own failure so far (it stays in step until every rank has learnt of it)
    auto fail = [&](const char *what) { if (!status) { set_err("%s: %s", what, kct_last_error()); status = 1; } };
    uint64_t n = 0, zero = 0, consumed = 0, got = 0;
    if (kct_len(t, &n) != KCT_OK || kct_get_hash(t, 0, &zero) != KCT_OK || kct_consumed(t, &consumed) != KCT_OK) fail("reading the table");
    std::vector<uint64_t> part(world, 0), send((size_t)world * 3, 0), recv((size_t)world * 3, 0);
    void *d_send = nullptr, *d_recv = nullptr;
    if (!status && n) {
        d_send = x_keep(x, &x->m_send, &x->m_send_cap, n * 16);
        if (!d_send) status = 1;
        else if (kct_export_by_owner_device(t, (uint32_t)world, d_send, n, part.data(), &got) != KCT_OK) fail("kct_export_by_owner_device");
    }
    // round 1: [status, pairs for that owner, this rank's count of key 0 (kept beside the device table: 0 is its EMPTY sentinel) -> owner 0]
    for (int r = 0; r < world; ++r) { send[3 * r] = status; send[3 * r + 1] = status ? 0 : part[r]; send[3 * r + 2] = r == 0 && !status ? zero : 0; }
    if (x_sizes(x, send.data(), 3, recv.data()) != 0) return 1;   // (the collective itself failed: nothing to agree through)
    uint64_t total = 0, zero_total = 0, failed = status;
    for (int r = 0; r < world; ++r) { failed |= recv[3 * r]; total += recv[3 * r + 1]; zero_total += recv[3 * r + 2]; }
    if (!failed && total) { d_recv = x_keep(x, &x->m_recv, &x->m_recv_cap, total * 16); if (!d_recv) status = 1; }
    // round 2: has every rank room for what it is to receive?
    std::vector<uint64_t> s1(world, status | failed), r1(world, 0);
    if (x_sizes(x, s1.data(), 1, r1.data()) != 0) return 1;
    for (int r = 0; r < world; ++r) failed |= r1[r];
    if (failed) {
        if (!status) set_err("the merge failed on another rank before any pair moved: this rank's table is unchanged");
        return 1;
    }
    std::vector<uint64_t> soff(world), sbytes(world), roff(world), rbytes(world);
    uint64_t so = 0, ro = 0;
    for (int r = 0; r < world; ++r) { soff[r] = so; sbytes[r] = part[r] * 16; so += sbytes[r]; roff[r] = ro; rbytes[r] = recv[3 * r + 1] * 16; ro += rbytes[r]; }
    int rc = x_start(x, d_send, soff.data(), sbytes.data(), d_recv, roff.data(), rbytes.data());
    if (rc == 0) rc = x_wait(x);
    // the owner's table: cleared, sized for its slice of the key space (2^27 slots per GPU for C4 instead of 2^30), refilled
    uint64_t a = 0, b = 0;
    const uint64_t zk = 0;
    if (rc == 0 && (kct_clear(t) != KCT_OK || kct_resize(t, total) != KCT_OK || (zero_total && kct_merge_host(t, &zk, &zero_total, 1, &a, &b) != KCT_OK) ||
                    (total && kct_merge_pairs_device(t, d_recv, total, &a, &b) != KCT_OK) || kct_add_consumed(t, consumed) != KCT_OK || kct_sync(t) != KCT_OK)) {
        set_err("folding the received pairs: %s", kct_last_error());
        rc = 1;
    }
    if (pairs_received) *pairs_received = total;
    // round 3: one word -- did every rank's refill succeed?  Every rank returns the same verdict; after a failure HERE (the payload has
    // moved, tables were cleared) the tables' contents are undefined on every rank and must be cleared.  (A rank whose communicator was
    // aborted cannot take part: x_sizes fails at once, it returns 1, and its peers wait for the launcher's hang guard -- see x_broken.)
    std::vector<uint64_t> s2(world, (uint64_t)(rc != 0)), r2(world, 0);
    if (x_sizes(x, s2.data(), 1, r2.data()) != 0) return 1;
    for (int r = 0; r < world; ++r)
        if (r2[r] && rc == 0) { set_err("the merge failed on rank %d after the pairs had moved: every rank's table must be cleared", r); rc = 1; }
    if (x->release_above && (x->m_send_cap > x->release_above || x->m_recv_cap > x->release_above)) kct_rccl_release_buffers(x);
    return rc;
}

// The merge's pair buffers are kept between merges (no hipMalloc / hipFree per job) -- after a big table's merge that is tens of GB of
// HBM held for the life of the communicator.  This gives them back; kct_rccl_release_above(x, bytes) makes every merge do so by itself
// when a buffer has grown beyond `bytes` (default 4 GiB; 0 = keep everything).
void kct_rccl_release_buffers(kct_rccl *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->m_send) { (void)hipFree(x->m_send); x->m_send = nullptr; x->m_send_cap = 0; }
    if (x->m_recv) { (void)hipFree(x->m_recv); x->m_recv = nullptr; x->m_recv_cap = 0; }
}
void kct_rccl_release_above(kct_rccl *x, uint64_t bytes) { if (x) x->release_above = bytes; }

void kct_rccl_merge_when_alone(kct_rccl *x, int on) { if (x) x->merge_when_alone = on != 0; }

void kct_rccl_destroy(kct_rccl *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm) (void)ncclCommDestroy(x->comm);
    if (x->d_sizes) (void)hipFree(x->d_sizes);
    if (x->m_send) (void)hipFree(x->m_send);
    if (x->m_recv) (void)hipFree(x->m_recv);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

}  // extern "C"
