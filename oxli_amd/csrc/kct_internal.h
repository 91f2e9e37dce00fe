// kct_internal.h -- host-side internals shared by the translation units of libkct_hip.so.
// Nothing here is exported (the library is built with -fvisibility=hidden); the C ABI is include/kct.h.
#pragma once
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kct.h"
#include "partition_args.h"

typedef uint64_t u64;     // host-side 64-bit values (matches the ABI's uint64_t)
typedef kct::u64 du64;    // words that live in device memory (unsigned long long, what HIP atomics take)

namespace kcth {

extern thread_local char g_err[512];
void set_err(const char *fmt, ...);
double now_ms();  // steady clock, for KCT_DEBUG lines

// Host worker threads that outlive a call (starting eight threads costs more than packing a million reads does).
// start(n, fn) runs fn(i), i = 0..n-1, on pool threads and returns at once; wait() blocks until they are done.  One job at
// a time per process (callers serialise on the pool's own lock, released by wait()).
class WorkerPool {
public:
    static WorkerPool &instance();
    void start(size_t n, std::function<void(size_t)> fn);
    bool pinned() const { return pin_; }
    int nodes() const;                  // NUMA nodes the workers are spread over: worker i lives on node i % nodes() (1 = not bound)
    int nodes_hint();                   // ... before the first job: reads the topology (and KCT_PACK_PIN) if that has not happened yet
    void wait();
private:
    void worker(size_t id);
    std::mutex job_lock_;               // held from start() to wait()
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    std::function<void(size_t)> fn_;
    size_t want_ = 0, generation_ = 0, running_ = 0;
    int pid_ = 0;
    bool pin_ = false;                  // KCT_PACK_PIN=1: workers bound to NUMA nodes
    std::vector<cpu_set_t> node_sets_;
    bool topo_read_ = false;
};
#define KCT_DBG(t, ...) do { if ((t)->debug) { fprintf(stderr, "[kct %11.3f ms] ", kcth::now_ms()); fprintf(stderr, __VA_ARGS__); } } while (0)

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? KCT_ERR_NOMEM : KCT_ERR_HIP;                 \
        }                                                                                   \
    } while (0)

#define KCT_TRY(expr)                 \
    do {                              \
        kct_status s_ = (expr);       \
        if (s_ != KCT_OK) return s_;  \
    } while (0)

constexpr u64 kDefaultSlots = 1ULL << 16;
constexpr u64 kMinSlots = 1ULL << 10;
constexpr double kMaxLoad = 0.65;                // grow between launches once load exceeds this
constexpr u64 kChunkPositions = 1ULL << 28;      // stream bytes per launch (bounds the spill list)
constexpr int kNumCounters = kct::kCounterShards * kct::kCounterStride;

inline u64 next_pow2(u64 v) {
    u64 p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct DevBuf {  // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    kct_status reserve(size_t n) {
        if (n <= cap) return KCT_OK;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)4096);
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return KCT_OK;
    }
    // the same, keeping the first `keep` bytes (waits for `stream`)
    kct_status reserve_keep(size_t n, size_t keep, hipStream_t stream) {
        if (n <= cap) return KCT_OK;
        size_t want = std::max(std::max(n, 2 * cap), (size_t)4096);
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, want));
        if (p && keep) HIP_TRY(hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (p) HIP_TRY(hipFree(p));
        p = q; cap = want;
        return KCT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinnedBuf {  // grow-only pinned host buffer
    void *p = nullptr;
    size_t cap = 0;
    kct_status reserve(size_t n) {
        if (n <= cap) return KCT_OK;
        if (p) HIP_TRY(hipHostFree(p));
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)4096);
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return KCT_OK;
    }
    // the same, keeping the first `keep` bytes
    kct_status reserve_keep(size_t n, size_t keep) {
        if (n <= cap) return KCT_OK;
        void *q = nullptr;
        HIP_TRY(hipHostMalloc(&q, n, hipHostMallocDefault));
        if (p && keep) memcpy(q, p, keep);
        if (p) (void)hipHostFree(p);
        p = q; cap = n;
        return KCT_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Measurement-only switches (A/B experiments of tools/*.sh).  Read from the environment ONCE, by kct_create, and only in a
// library built with -DKCT_DEBUG_ENV (`make variant V=dbg EXTRA=-DKCT_DEBUG_ENV`); the shipped build compiles the defaults in.
struct Tuning {
    int pbits = -1;            // KCT_PBITS: first-level bins of a two-level pass
    int k1b_lines = 0;         // KCT_K1B_LINES: 64-byte lines a second-level bin flushes together
    bool pairs_nopersist = false, k2_nopersist = false;  // KCT_PAIRS_NOPERSIST / KCT_K2_NOPERSIST: one workgroup per block
    bool flush_atomic = false; // KCT_FLUSH_ATOMIC: conversions by random table access instead of the partitioned pair route
    bool k1b_half = false;     // KCT_K1B_HALF: two ring flushes per slab in the 64-bit second level
    int sub_chunks = 0;        // KCT_SUB_CHUNKS: sub-chunks of a 64-bit two-level pass that does not fit in one (0 = the default, 4)
    int ablate = 0;            // KCT_ABLATE: skip work (results INVALID)
    int pack_threads = 32;     // KCT_PACK_THREADS (operational, read in every build): host threads packing a batch (16 until round 6: with the encoder's scalar tails gone, 32 threads are as fast at their best and steadier from process to process: tools/e2e_diag.py)
    int k1_flushers = 0;       // KCT_K1_FLUSHERS (operational, read in every build): 4 (or 2) = the wave-specialised K1 (k1ws_kernel.h) with that many flusher waves; 0 = the barrier-synchronised K1 (the default: DESIGN.md section 9, round 6)
};

struct ProfEntry { std::string name; u64 launches = 0; double ms = 0; };
struct ProfPending { int entry; hipEvent_t a, b; };

}  // namespace kcth

struct kct_table {
    int device = 0;
    uint8_t k = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t mem_free_seen = 0;            // hipMemGetInfo's last answer and when it was given (try_stage, kct_entry.hip)
    double mem_free_seen_ms = 0;
    hipStream_t copy_stream = nullptr;   // kct_consume_file's uploads (made at the first file; two staging halves take turns)

    du64 *slots = nullptr;  // 2 * cap words (device)
    u64 slots_alloc = 0;    // slots the allocation has room for (>= cap: kct_resize makes an EMPTY table smaller or larger in place)
    u64 cap = 0;
    int block_bits = 0;     // log2(slots per probing block) = min(13, log2 cap)
    bool lazy_empty = false;  // kct_clear() was called and the memset has not been issued yet
    int num_cus = 256;
    // Deferred mode (on by default; kct_set_deferred): per-record consume() calls only append to a pinned host buffer
    // and return; the buffered records are counted in one device pass when the buffer fills or anything reads
    // the table.  The k-mer count returned to the caller comes from a host-side validity scan.
    bool deferred = true;
    kcth::PinnedBuf h_pending;
    size_t pending_used = 0;
    u64 pending_records = 0;
    // ... and its device-side twin: kct_consume_device calls that are small for the table (fewer than 4 window starts per slot) are
    // COPIED behind one another into d_defer (their good windows counted on the way, for the n the call returns) and counted together
    // when anything else touches the table, when 32 window starts per slot have gathered, or when the buffer is full -- a caller that
    // feeds a large input in pieces gets the passes (and the path choice) of one large call
    bool defer_device = true;
    kcth::DevBuf d_defer;
    size_t defer_used = 0;
    u64 defer_windows = 0;
    // One caller at a time (the reference's `&mut self` under the GIL, lib.rs:546).  Call glue may release the GIL around a device
    // pass (csrc/pyfast.c, ctypes, a pyo3 allow_threads), so a second thread CAN arrive: it is turned away (KCT_ERR_BUSY, pyo3's
    // "Already borrowed") instead of racing the first one.  owner = the thread inside an entry point (0 = none); the same thread may
    // nest (an entry point that calls another).
    std::atomic<unsigned long long> owner{0};
    int owner_depth = 0;
    bool poisoned = false;  // a device pass over buffered records failed half-way: counts already reported are missing for good,
                            // so every later call on this table fails too (kct_clear resets it)
    // packed input (2-bit codes + validity bits, 16 bases per group): set while a packed stream is being counted; the `d_stream`
    // pointers handed around inside kct_consume.hip are then offsets from packed_base (never dereferenced)
    const unsigned int *packed_codes = nullptr;
    const unsigned short *packed_valid = nullptr;
    const unsigned char *packed_base = nullptr;
    // super-k-mer input (the early route's owner side, kct_route.hip): set while received runs are being counted; `d_stream` pointers
    // are then WINDOW offsets from runs_base (never dereferenced) and runs_in.groups describes window 0's group
    kct::RunsInput runs_in;
    const unsigned char *runs_base = nullptr;
    bool packed_upload = true;  // kct_consume_batch packs large skip-bad batches on the host and uploads 0.375 B per base
    bool auto_sized = true; // no capacity hint / reserve yet: bulk ingest ramps its launch size up with the table
    kcth::Tuning tune;      // measurement switches, fixed at create time
    int ablate = 0;         // = tune.ablate
    bool debug = false;     // KCT_DEBUG at create time: one stderr line per partitioned pass
    int force_path = 0;     // 0 = choose per pass, 1 = direct atomic kernel only, 2 = partitioned whenever the geometry allows, 3 = dedupe-first
    bool dedupe_off = false;  // a dedupe-first pass found too many distinct k-mers: this table goes back to hashing every window
    // dedupe-first path (k <= 32): a shadow table keyed by packed k-mers holds counts that are still PENDING for `slots`
    du64 *shadow = nullptr;     // as many slots as `slots`, same block-SoA layout (allocated on first use, re-created when the table grows)
    u64 shadow_cap = 0;
    int shadow_block_bits = 0;
    bool shadow_empty = true;   // no keys yet: K2 starts its blocks from zeros instead of loading them
    bool shadow_dirty = false;  // pending counts exist: anything that reads `slots` flushes first (use())
    u64 shadow_keys = 0;
    // compact variant (k <= 21): 2^s32_sbits blocks x 8192 slots of u32 key + u32 count -- 1024 blocks (64 MiB) beside a table
    // of up to 1024 blocks, otherwise as many blocks as the table has (at least 2^16): two partition levels
    unsigned int *shadow32 = nullptr;
    int s32_sbits = 10;
    // ... of which this table holds the blocks of first-level bins [s32_bin0, s32_bin0 + s32_nbins): all 1024 on one GPU, an
    // owner's range of them in the multi-GPU early route (kct_route.hip); blocks = s32_nbins << (s32_sbits - 10)
    unsigned int s32_bin0 = 0, s32_nbins = 1024;
    bool s32_empty = true, s32_dirty = false, compact_off = false;
    u64 s32_keys = 0, s32_windows = 0;  // keys it holds; window starts counted into it since its last flush (u32 counts!)
    // 128-bit variant (33 <= k <= 64): 1024 blocks x 4096 slots of {x, y, u32 count} (80 MiB), keyed by mix128 pairs; whatever the
    // table's size (its conversion inserts with atomics) -- for inputs of up to ~2.5 M distinct k-mers
    du64 *shadow128 = nullptr;
    bool s128_empty = true, s128_dirty = false, dedupe128_off = false;
    u64 s128_keys = 0, s128_windows = 0;
    // the dedupe probe's own small shadows (kept between calls, swapped in for the probe pass only)
    du64 *probe_shadow = nullptr;
    unsigned int *probe_shadow32 = nullptr;
    // {hash, count} pairs a dedupe-first pass set aside instead of inserting them into a lazily empty table (merge_overflow_kernel's
    // PendingList): merged by the next conversion.  The device cursor is d_counters[kNumCounters + 8] (zero_counters leaves it).
    kcth::DevBuf d_pending;
    u64 pending_pairs = 0;
    u64 windows_since_read = 0; // window starts consumed since anything last read the table (use()): how long the caller's runs are
    u64 call_windows_left = 0;  // window starts the running consume call still has to count (no read can come before them)
    bool wide_bursty = false;      // ... the same for the 8-byte-entry modes (hashing, 64-bit dedupe-first): a flush every 2 windows instead of 4
    bool compact_bursty = false;   // a compact dedupe-first pass sent more than 2 % of its entries over the overflow route (k-mers arriving in bursts:
                                   // position-sorted reads): K1 flushes its ring every 4 windows instead of 8 from then on (kct_consume.hip)
    int fault_point = 0;        // kct_debug_inject_fault: the next routed call fails once at this point (kct_route.hip) ...
    u64 fault_pass = 0;         // ... of this pass
    u64 more_windows = 0;       // ... and those of later passes of the same job, announced by the early route (kct_route.hip)
    bool expect_new_keys = false;  // the dedupe probe found (nearly) every k-mer new: K2's fast path claims slots itself (AggregateArgs::claim)
    bool dedupe_hint = false;   // the last dedupe-first pass paid off: a cleared table starts with that path again
    u64 n_keys = 0;        // distinct non-zero hashes in `slots`
    u64 consumed = 0;      // lib.rs:36
    bool zero_present = false;  // key 0 lives host-side (0 is the EMPTY sentinel on the device)
    u64 zero_count = 0;

    du64 *d_counters = nullptr;  // kNumCounters tallies + 8 scratch words (device)
    u64 *h_counters = nullptr;   // pinned mirror
    kcth::DevBuf d_stream, d_spill, d_aux, d_aux2, d_scratch, d_regions, d_irr, d_sort, d_scratch2, d_regions2, d_irr2, d_pairs_ovf,
        d_unpack,  // the ASCII image of a packed chunk (or of a chunk of super-k-mer windows), for the kernels that read bytes
        d_sk_bases, d_sk_starts, d_sk_meta, d_sk_lists, d_sk_dir, d_sk_send, d_sk_recv, d_sk_inbox,  // early route: (workgroup, owner) regions, counts,
                                                                                 // gather lists, run directory, loop-back slabs
        d_failed,  // K2: the numbers of the blocks it abandoned (partition_kernels.h FailedBlocks)
        d_prefix;  // error mode: the offending record's valid prefix (its own buffer: consume_stream reuses d_aux2 / d_spill)
    kcth::PinnedBuf h_stage;
    double batch_tl[16] = {0};  // kct_batch_timeline: the last packed-upload batch
    std::vector<kcth::PinnedBuf> h_file;  // kct_consume_file's chunk buffers (two per parser thread), kept between calls
    std::vector<std::vector<unsigned char>> file_text;   // ... and its BGZF slot threads' text buffers

    bool prof_on = false;
    std::vector<kcth::ProfEntry> prof;
    std::vector<kcth::ProfPending> prof_pending;
    std::vector<hipEvent_t> event_pool;
};

namespace kcth {

struct Borrow {  // held by every C-ABI entry point for as long as it runs
    kct_table *t;
    bool ok = true;
    explicit Borrow(kct_table *t_) : t(t_) {
        if (!t) return;
        const unsigned long long me = (unsigned long long)pthread_self() | 1ULL;
        unsigned long long none = 0;
        if (t->owner.load(std::memory_order_relaxed) == me) { ++t->owner_depth; return; }
        if (t->owner.compare_exchange_strong(none, me, std::memory_order_acquire)) { t->owner_depth = 1; return; }
        ok = false;
        set_err("Already borrowed: another thread is inside a call on this table (a kct_table takes one caller at a time)");
    }
    ~Borrow() {
        if (t && ok && --t->owner_depth == 0) t->owner.store(0, std::memory_order_release);
    }
    Borrow(const Borrow &) = delete;
};
#define KCT_BORROW(t) kcth::Borrow borrow_##t(t); if (!borrow_##t.ok) return KCT_ERR_BUSY

struct ProfScope {
    kct_table *t;
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(kct_table *t_, const char *name) : t(t_) {
        if (!t->prof_on) return;
        for (size_t i = 0; i < t->prof.size(); ++i)
            if (t->prof[i].name == name) idx = (int)i;
        if (idx < 0) { t->prof.push_back(ProfEntry{name}); idx = (int)t->prof.size() - 1; }
        a = take(); b = take();
        if (a && b) (void)hipEventRecord(a, t->stream);
    }
    ~ProfScope() {
        if (idx < 0 || !a || !b) return;
        (void)hipEventRecord(b, t->stream);
        t->prof[idx].launches++;
        t->prof_pending.push_back(ProfPending{idx, a, b});
    }
    hipEvent_t take() {
        if (!t->event_pool.empty()) { hipEvent_t e = t->event_pool.back(); t->event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

void prof_collect(kct_table *t);
kct_status use(kct_table *t);         // select the device, count whatever deferred mode has buffered, convert the shadow table's pending counts
kct_status use_consume(kct_table *t); // the same without the shadow flush: what consume entry points call
kct_status flush_shadow(kct_table *t);   // both shadows' pending counts and the pending pair list -> the real table
void parallel_memcpy(void *dst, const void *src, size_t nbytes);  // several threads above 8 MiB
kct_status use_device(kct_table *t);  // select the device only
kct_status flush_pending(kct_table *t);
kct_status flush_deferred_device(kct_table *t);  // kct_entry.hip: counts what kct_consume_device has staged
kct::TableGeom geom(const kct_table *t);
kct::TableView view(kct_table *t, u64 spill_cap);
int log2_u64(u64 v);
void set_geometry(kct_table *t);
kct_status materialize(kct_table *t);
kct_status zero_counters(kct_table *t);
kct_status read_counters(kct_table *t, u64 out[4], u64 *spill_n);
int merge_grid(u64 n);
kct_status grow_to(kct_table *t, u64 new_cap);
u64 spill_growth_target(const kct_table *t, u64 spilled);  // the capacity to grow to when `spilled` entries found no room
kct_status maybe_grow(kct_table *t);
kct_status merge_pairs(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]);
kct_status replay_spill(kct_table *t, u64 spilled, u64 *n_out);
// kct_consume.hip: {hash, count} pairs through the LDS-ring partition + per-block LDS merge (tables of up to 1024 blocks)
bool pairs_partition_pays(const kct_table *t, u64 n);
kct_status merge_pairs_partitioned(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]);
kct_status point_add(kct_table *t, u64 h, u64 *count_out);
// kct_consume.hip
kct_status consume_stream(kct_table *t, const unsigned char *d_stream, u64 nbytes, u64 *n_out);
kct_status consume_stream_packed(kct_table *t, const unsigned int *d_codes, const unsigned short *d_valid, u64 nbases, u64 *n_out);
kct_status unpack_stream(kct_table *t, const unsigned int *d_codes, const unsigned short *d_valid, u64 ng);  // kct_entry.hip
kct_status consume_stream_runs(kct_table *t, const kct::RunsInput &in, u64 ngroups, u64 *n_out);  // received super-k-mers (kct_route.hip)
// kct_runs.hip: K1's super-k-mer instantiations and the early route's own kernels
kct_status consume_device_staged(kct_table *t, const unsigned char *d_stream, size_t nbytes, u64 *n_total);   // kct_entry.hip
kct_status stage_piece_async(kct_table *t, const unsigned char *d_stream, size_t nbytes, bool *staged);   // (no waiting; good windows added to ...
du64 *staged_good_word(kct_table *t);                                                                       // ... this device word)
void launch_partition_runs(kct_table *t, int mode, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa);
bool launch_partition_ws(kct_table *t, int mode, const unsigned char *d_stream, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa);   // kct_k1ws.hip
void launch_split(kct_table *t, const unsigned char *d_stream, u64 nbytes, u64 ntiles, const kct::SplitArgs &sa);
inline int split_streams(const kct_table *t) { return t->num_cus; }   // workgroups of the split = streams a rank sends to every owner
void launch_gather_units(kct_table *t, const void *src, const du64 *src_off, const du64 *dst_off, const unsigned int *n, unsigned int count, void *dst);
void launch_run_directory(kct_table *t, const kct::RunStream *streams, unsigned int nstreams, const du64 *starts, kct::RunGroup *groups);
void launch_expand_runs(kct_table *t, const kct::RunsInput &in, u64 ngroups, unsigned char *out);
// ... its kernels' launchers and sizing rules, for kct_route.hip (which does not instantiate the kernels itself)
void launch_partition(kct_table *t, int mode /* 0 hashes, 1 mix64 values, 2 compact */, const unsigned char *d_stream, u64 chunk_bytes, u64 ntiles,
                      const kct::PartitionArgs &pa);
void launch_repartition(kct_table *t, int mode, unsigned grid, const kct::RepartitionArgs &ra, bool whole_slab);
void launch_aggregate32(kct_table *t, unsigned grid, const kct::Aggregate32Args &aa);                 // two-level variant
void launch_aggregate64(kct_table *t, unsigned grid, const kct::AggregateArgs &aa, bool shadow);      // two-level variant
void launch_merge_overflow(kct_table *t, int mode, const du64 *regions, const unsigned int *counts, int nregions, unsigned int region_cap,
                           const du64 *abort, const kct::TableView &tv, const du64 *total);
// (region_capacity / overflow_capacity / repartition_min_lines and the path choices: path_policy.h)
kct_status failed_blocks(kct_table *t, u64 nblocks, kct::FailedBlocks *fb);
kct_status recount_failed(kct_table *t, int mode, const void *scratch, u64 seg_stride, u64 block_stride, const unsigned int *region_count, int nregions,
                          u64 nfailed, u64 entries, int sbits, u64 tallies[4]);
kct_status overflow_total(kct_table *t, const unsigned int *d_counts_a, size_t na, const unsigned int *d_counts_b, size_t nb, u64 *total);
kct_status ensure_shadow(kct_table *t, u64 want_cap, bool *ok);
kct_status ensure_shadow32(kct_table *t, int want_sbits, bool *ok, unsigned int nbins = 1024, unsigned int bin0 = 0);
u64 compact_slots(const kct_table *t);

}  // namespace kcth
