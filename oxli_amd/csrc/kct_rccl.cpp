// kct_rccl.cpp -- kct_exchange_ops (include/kct.h) over RCCL: libkct_rccl.so (include/kct_rccl.h).
//
// The early multi-GPU route needs three things of its caller: device buffers, a small all-to-all of sizes in host memory, and an
// asynchronous all-to-all of byte ranges.  Here they are hipMalloc, and ncclSend / ncclRecv pairs inside one group on a stream that
// belongs to the communicator -- the payload moves while the library's kernels run on the table's stream; wait() is a
// hipStreamSynchronize of that stream.  xGMI is point to point: a group of world - 1 sends and receives uses every link at once.
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

int x_sizes(void *user, const uint64_t *send, uint32_t nvals, uint64_t *recv) {
    kct_rccl *x = (kct_rccl *)user;
    const size_t n = (size_t)x->world * nvals;
    HIP_OK(hipSetDevice(x->device));
    if (2 * n > x->sizes_cap) {
        if (x->d_sizes) HIP_OK(hipFree(x->d_sizes));
        x->d_sizes = nullptr; x->sizes_cap = 0;
        HIP_OK(hipMalloc((void **)&x->d_sizes, 2 * n * sizeof(uint64_t)));
        x->sizes_cap = 2 * n;
    }
    HIP_OK(hipMemcpyAsync(x->d_sizes, send, n * sizeof(uint64_t), hipMemcpyHostToDevice, x->stream));
    NCCL_OK(ncclGroupStart());
    for (int r = 0; r < x->world; ++r) {
        NCCL_OK(ncclSend(x->d_sizes + (size_t)r * nvals, nvals, ncclUint64, r, x->comm, x->stream));
        NCCL_OK(ncclRecv(x->d_sizes + n + (size_t)r * nvals, nvals, ncclUint64, r, x->comm, x->stream));
    }
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipMemcpyAsync(recv, x->d_sizes + n, n * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
    HIP_OK(hipStreamSynchronize(x->stream));
    return 0;
}

int x_start(void *user, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes) {
    kct_rccl *x = (kct_rccl *)user;
    HIP_OK(hipSetDevice(x->device));
    NCCL_OK(ncclGroupStart());
    for (int r = 0; r < x->world; ++r) {
        if (send_bytes[r]) NCCL_OK(ncclSend((const char *)d_send + send_off[r], send_bytes[r], ncclUint8, r, x->comm, x->stream));
        if (recv_bytes[r]) NCCL_OK(ncclRecv((char *)d_recv + recv_off[r], recv_bytes[r], ncclUint8, r, x->comm, x->stream));
        if (r != x->rank) { x->sent += send_bytes[r]; x->received += recv_bytes[r]; }
    }
    NCCL_OK(ncclGroupEnd());
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

void kct_rccl_destroy(kct_rccl *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm) (void)ncclCommDestroy(x->comm);
    if (x->d_sizes) (void)hipFree(x->d_sizes);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

}  // extern "C"
