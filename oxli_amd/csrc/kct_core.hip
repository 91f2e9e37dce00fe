// kct_core.hip -- the device-resident table behind the C ABI (include/kct.h): lifetime, growth and re-hash,
// point updates and lookups, dump / export / merge, stream and kernel-timing plumbing.
// Bulk ingest lives in kct_consume.hip, file parsing in kct_ingest.hip.
#include "kct_internal.h"
#include <unistd.h>
#include "table_kernels.h"

extern "C" int kx_sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned long long *vals_in,
                                 unsigned long long *vals_out, size_t n, void *tmp, size_t *tmp_bytes, void *stream);  // sort.hip

namespace kcth {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

double now_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void prof_collect(kct_table *t) {
    if (t->prof_pending.empty()) return;
    (void)hipStreamSynchronize(t->stream);
    for (auto &p : t->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) t->prof[p.entry].ms += ms;
        t->event_pool.push_back(p.a);
        t->event_pool.push_back(p.b);
    }
    t->prof_pending.clear();
}

void parallel_memcpy(void *dst, const void *src, size_t nbytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = nbytes >= ((size_t)8 << 20) ? std::min<size_t>(8, hw ? hw : 1) : 1;
    if (nthreads <= 1) { if (nbytes) memcpy(dst, src, nbytes); return; }
    std::vector<std::thread> pool;
    for (size_t i = 0; i < nthreads; ++i) {
        const size_t lo = (nbytes * i / nthreads) & ~(size_t)4095, hi = i + 1 == nthreads ? nbytes : (nbytes * (i + 1) / nthreads) & ~(size_t)4095;
        if (hi > lo) pool.emplace_back([=]() { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (auto &th : pool) th.join();
}

WorkerPool &WorkerPool::instance() {
    static WorkerPool *pool = new WorkerPool();  // never destroyed: its threads may outlive static destruction
    return *pool;
}

// The CPUs of every NUMA node that this process may run on (/sys/devices/system/node/nodeN/cpulist); empty = one node, or unknown.
static std::vector<cpu_set_t> numa_cpu_sets() {
    std::vector<cpu_set_t> out;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    for (int nd = 0; nd < 64; ++nd) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
        FILE *f = fopen(path, "r");
        if (!f) break;
        cpu_set_t set;
        CPU_ZERO(&set);
        char buf[4096];
        if (fgets(buf, sizeof buf, f)) {
            for (char *q = buf; *q && *q != '\n';) {   // "0-63,128-191"
                char *e;
                const long a_ = strtol(q, &e, 10);
                if (e == q) break;
                long b_ = a_;
                if (*e == '-') { q = e + 1; b_ = strtol(q, &e, 10); }
                for (long x = a_; x <= b_ && x < CPU_SETSIZE; ++x) if (x >= 0 && CPU_ISSET(x, &allowed)) CPU_SET(x, &set);
                if (*e != ',') break;
                q = e + 1;
            }
        }
        fclose(f);
        if (CPU_COUNT(&set) > 0) out.push_back(set);
    }
    if (out.size() < 2) out.clear();
    return out;
}

int WorkerPool::nodes() const { return node_sets_.empty() ? 1 : (int)node_sets_.size(); }

int WorkerPool::nodes_hint() {
    std::unique_lock<std::mutex> lk(m_);
    if (pid_ != (int)getpid() && !topo_read_) {   // (start() reads it again for a forked child)
        const char *pin = getenv("KCT_PACK_PIN");
        if (pin && atoi(pin) != 0) node_sets_ = numa_cpu_sets();
        topo_read_ = true;
    }
    return node_sets_.empty() ? 1 : (int)node_sets_.size();
}

void WorkerPool::worker(size_t id) {
    // KCT_PACK_PIN=1 (a measurement switch): worker i lives on NUMA node i % nodes (any CPU of it), and kct_consume_batch hands each node's
    // share of a batch (move_pages in query mode) to that node's workers.  Round 6 measured it on the two-socket EPYC 9575F test boxes:
    // 3.2-3.8 ms per C2 batch against 2.5-2.8 ms with the scheduler's own placement (tools/e2e_diag.py) -- so it is off by default.
    if (!node_sets_.empty()) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_sets_[id % node_sets_.size()]);
    size_t seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (id >= want_) continue;  // this job uses fewer threads
        auto fn = fn_;
        lk.unlock();
        fn(id);
        lk.lock();
        if (--running_ == 0) done_cv_.notify_all();
    }
}

void WorkerPool::start(size_t n, std::function<void(size_t)> fn) {
    job_lock_.lock();
    std::unique_lock<std::mutex> lk(m_);
    if (pid_ != (int)getpid()) {  // first use, or a forked child (threads do not survive a fork): a fresh set
        auto *gone = new std::vector<std::thread>(std::move(threads_));  // (never joined or destroyed: they are not this process's)
        (void)gone;
        threads_.clear();
        pid_ = (int)getpid();
        const char *pin = getenv("KCT_PACK_PIN");   // 1: bind the workers to NUMA nodes (measured SLOWER than the scheduler's placement on the two-socket test boxes: off by default)
        pin_ = pin && atoi(pin) != 0;
        node_sets_.clear();
        if (pin_) node_sets_ = numa_cpu_sets();
        pin_ = !node_sets_.empty();
    }
    fn_ = std::move(fn);
    want_ = n;
    running_ = n;
    ++generation_;
    // a worker born now starts with seen == 0 != generation_ and takes this job as soon as the lock is released
    while (threads_.size() < n) {
        const size_t id = threads_.size();
        threads_.emplace_back([this, id]() { worker(id); });
    }
    lk.unlock();
    cv_.notify_all();
}

void WorkerPool::wait() {
    {
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return running_ == 0; });
        fn_ = nullptr;
    }
    job_lock_.unlock();
}

kct_status use_device(kct_table *t) {
    if (!t) { set_err("null table handle"); return KCT_ERR_ARG; }
    if (t->poisoned) { set_err("an earlier device pass over buffered records failed: the table is missing counts it has reported (kct_clear resets it)"); return KCT_ERR_HIP; }
    HIP_TRY(hipSetDevice(t->device));
    return KCT_OK;
}

kct_status use_consume(kct_table *t) {
    KCT_TRY(use_device(t));
    if (t->pending_used) KCT_TRY(flush_pending(t));  // (keeps the order of consumed records irrelevant, and `consumed` exact)
    if (t->defer_used) KCT_TRY(flush_deferred_device(t));
    return KCT_OK;
}

kct_status use(kct_table *t) {
    KCT_TRY(use_consume(t));
    if (t->shadow_dirty || t->s32_dirty || t->s128_dirty || t->pending_pairs) {   // reads must observe every earlier consume()
        KCT_DBG(t, "use(): converting pending counts (%llu / %llu shadow keys)\n", (unsigned long long)t->shadow_keys, (unsigned long long)t->s32_keys);
        KCT_TRY(flush_shadow(t));
        KCT_DBG(t, "use(): converted\n");
    }
    t->windows_since_read = 0;
    return KCT_OK;
}

kct::TableGeom geom(const kct_table *t) {
    kct::TableGeom g;
    g.mask = t->cap - 1;
    g.block_bits = t->block_bits;
    return g;
}

kct::TableView view(kct_table *t, u64 spill_cap) {
    kct::TableView v;
    v.words = t->slots;
    v.g = geom(t);
    v.spill = (du64 *)t->d_spill.p;
    v.spill_cap = spill_cap;
    v.spill_n = t->d_counters + kNumCounters;  // scratch word 0
    v.max_groups = kct::kMaxProbeGroups;
    return v;
}

int log2_u64(u64 v) { int b = 0; while ((1ULL << b) < v) ++b; return b; }

void set_geometry(kct_table *t) { t->block_bits = std::min(kct::kBlockBitsMax, log2_u64(t->cap)); }  // cap >= kMinSlots = 1024 > one group

// kct_clear() defers its memset: the partitioned path rewrites every block from zeros anyway.
// Anything else that touches `slots` calls this first.
kct_status materialize(kct_table *t) {
    if (t->lazy_empty) {
        KCT_DBG(t, "materialize: memset of %llu slots\n", (unsigned long long)t->cap);
        HIP_TRY(hipMemsetAsync(t->slots, 0, t->cap * 16, t->stream));
        t->lazy_empty = false;
        if (t->debug) { HIP_TRY(hipStreamSynchronize(t->stream)); KCT_DBG(t, "materialize: done\n"); }
    }
    return KCT_OK;
}

kct_status zero_counters(kct_table *t) {
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 8) * sizeof(u64), t->stream));
    return KCT_OK;
}

// copies the tallies back and folds the shards; waits for the stream
kct_status read_counters(kct_table *t, u64 out[4], u64 *spill_n) {
    HIP_TRY(hipMemcpyAsync(t->h_counters, t->d_counters, (kNumCounters + 9) * sizeof(u64), hipMemcpyDeviceToHost, t->stream));  // (+ the pending-list cursor)
    HIP_TRY(hipStreamSynchronize(t->stream));
    for (int c = 0; c < 4; ++c) out[c] = 0;
    for (int s = 0; s < kct::kCounterShards; ++s)
        for (int c = 0; c < 4; ++c) out[c] += t->h_counters[s * kct::kCounterStride + c];
    *spill_n = t->h_counters[kNumCounters];
    return KCT_OK;
}

kct_status alloc_slots(int, u64 cap, hipStream_t stream, du64 **out) {
    du64 *p = nullptr;
    HIP_TRY(hipMalloc((void **)&p, cap * 16));
    hipError_t e = hipMemsetAsync(p, 0, cap * 16, stream);
    if (e != hipSuccess) { (void)hipFree(p); set_err("hipMemsetAsync: %s", hipGetErrorString(e)); return KCT_ERR_HIP; }
    *out = p;
    return KCT_OK;
}

// A spill list counts OCCURRENCES of the keys that found no room, not distinct keys, so sizing the new table
// for all of them can overshoot by the coverage of the data.  Grow by at most 4x per step; what still does
// not fit spills again and the caller's loop grows once more.
u64 spill_growth_target(const kct_table *t, u64 spilled) {
    const u64 want = next_pow2((u64)((double)(t->n_keys + spilled) / kMaxLoad) + 1);
    return std::max<u64>(2 * t->cap, std::min<u64>(want, 4 * t->cap));
}

int merge_grid(u64 n) { return (int)std::min<u64>((n + kct::kBlock - 1) / kct::kBlock, 256 * 8); }

// Folds n (hash, count) pairs into the table, growing and replaying the spill list until every
// pair is placed.  tallies[] accumulates CTR_* sums.  `stride` 1 = separate arrays, 2 = slot array.
kct_status grow_to(kct_table *t, u64 new_cap);

kct_status merge_pairs(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]) {
    if (pairs_partition_pays(t, n) && (const void *)d_keys != t->d_aux2.p) return merge_pairs_partitioned(t, d_keys, d_counts, n, stride, tallies);
    KCT_TRY(materialize(t));
    // Round 1 probes kMaxProbeGroups groups per key (the hot paths' bound).  What spills is retried with whole-block
    // probing: keys that merely share their low bits (count_hash / __setitem__ / load of arbitrary u64 keys -- the
    // reference's HashMap takes any) then find room without the table growing.  Only what spills from a whole-block
    // round (its block is FULL) or a table past its load limit makes the table grow.
    bool whole_block = false;
    u64 stuck = 0;
    while (n > 0) {
        KCT_TRY(t->d_spill.reserve(n * 16));
        KCT_TRY(zero_counters(t));
        kct::TableView tv = view(t, n);
        if (whole_block) tv.max_groups = 0;
        {
            ProfScope ps(t, "merge_pairs_kernel");
            hipLaunchKernelGGL(kct::merge_pairs_kernel, dim3(merge_grid(n)), dim3(kct::kBlock), 0, t->stream, d_keys, d_counts, n,
                               (const du64 *)nullptr, (const du64 *)nullptr, stride, tv, t->d_counters);
        }
        HIP_TRY(hipGetLastError());
        u64 c[4], spilled;
        KCT_TRY(read_counters(t, c, &spilled));
        for (int i = 0; i < 4; ++i) tallies[i] += c[i];
        t->n_keys += c[kct::CTR_NEWKEYS];
        if (spilled == 0) break;
        // move the spill list aside; then either probe further or grow, and replay it
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        const bool too_full = (double)(t->n_keys + 1) > kMaxLoad * (double)t->cap;
        if (whole_block || too_full) {
            // A block that is full although the table is nearly empty holds keys that agree in every bit a larger table
            // would index with (e.g. i << 40): growing cannot separate them.  Give up cleanly instead of doubling until
            // HBM runs out.
            if (whole_block && !too_full && (double)t->n_keys < 0.05 * (double)t->cap && ++stuck >= 3) {
                set_err("%llu keys do not fit: they collide in one %u-slot table block at every table size tried (their low bits agree)",
                        (unsigned long long)spilled, 1u << t->block_bits);
                return KCT_ERR_NOMEM;
            }
            KCT_TRY(grow_to(t, spill_growth_target(t, spilled)));
        }
        whole_block = true;
        d_keys = (const du64 *)t->d_aux2.p;
        d_counts = d_keys + 1;
        stride = 2;
        n = spilled;
    }
    return KCT_OK;
}

kct_status rehash_into(kct_table *t, u64 new_cap);

kct_status grow_to(kct_table *t, u64 new_cap) {
    new_cap = std::max(next_pow2(new_cap), kMinSlots);
    if (new_cap <= t->cap) new_cap = t->cap * 2;
    return rehash_into(t, new_cap);
}

// every key into a fresh array of new_cap slots (a power of two with room for them)
kct_status rehash_into(kct_table *t, u64 new_cap) {
    const u64 old_keys = t->n_keys;
    du64 *fresh = nullptr;
    KCT_TRY(alloc_slots(t->device, new_cap, t->stream, &fresh));
    u64 placed = 0;
    if (t->slots && old_keys > 0 && !t->lazy_empty) {
        // re-insert every occupied slot, probing whole blocks: a key that fitted the old block fits the new one
        kct::TableView nv;
        nv.words = fresh;
        nv.g.mask = new_cap - 1;
        nv.g.block_bits = std::min(kct::kBlockBitsMax, log2_u64(new_cap));
        nv.spill = nullptr; nv.spill_cap = 0; nv.spill_n = t->d_counters + kNumCounters;
        nv.max_groups = 0;
        kct_status st = zero_counters(t);
        if (st == KCT_OK) {
            {
                ProfScope ps(t, "rehash_kernel");
                hipLaunchKernelGGL(kct::rehash_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t), nv,
                                   t->d_counters);
            }
            if (hipGetLastError() != hipSuccess) { set_err("rehash_kernel launch failed"); st = KCT_ERR_HIP; }
        }
        u64 c[4] = {0, 0, 0, 0}, spilled = 0;
        if (st == KCT_OK) st = read_counters(t, c, &spilled);
        placed = c[kct::CTR_NEWKEYS];
        if (st == KCT_OK && (spilled != 0 || placed != old_keys)) {
            set_err("re-hash lost keys: %llu of %llu placed, %llu spilled", (unsigned long long)placed, (unsigned long long)old_keys,
                    (unsigned long long)spilled);
            st = KCT_ERR_HIP;
        }
        if (st != KCT_OK) { (void)hipStreamSynchronize(t->stream); (void)hipFree(fresh); return st; }
    } else {
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    du64 *old = t->slots;
    t->slots = fresh;
    t->slots_alloc = new_cap;
    t->cap = new_cap;
    set_geometry(t);
    t->n_keys = placed;
    t->lazy_empty = false;  // (a lazily cleared old array held no keys: it is simply dropped)
    if (old) HIP_TRY(hipFree(old));
    return KCT_OK;
}

kct_status maybe_grow(kct_table *t) {
    if ((double)t->n_keys > kMaxLoad * (double)t->cap) {
        u64 target = t->cap;
        while ((double)t->n_keys > 0.25 * (double)target) target <<= 1;
        return grow_to(t, target);
    }
    return KCT_OK;
}

// Replays a spill list (already copied to d_aux2) after growing; adds what it counted to *n_out.
kct_status replay_spill(kct_table *t, u64 spilled, u64 *n_out) {
    KCT_TRY(grow_to(t, spill_growth_target(t, spilled)));
    u64 tl[4] = {0, 0, 0, 0};
    KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux2.p, (const du64 *)t->d_aux2.p + 1, spilled, 2, tl));
    *n_out += tl[kct::CTR_TOTAL_ADDED];
    return KCT_OK;
}

kct_status point_add(kct_table *t, u64 h, u64 *count_out) {
    if (h == 0) {  // 0 is the device EMPTY sentinel: kept host-side (count_hash(0) is legal, lib.rs:100)
        t->zero_present = true;
        *count_out = ++t->zero_count;
        return KCT_OK;
    }
    KCT_TRY(maybe_grow(t));
    KCT_TRY(t->h_stage.reserve(64));
    KCT_TRY(t->d_aux.reserve(64));
    u64 *hp = (u64 *)t->h_stage.p;
    hp[0] = h; hp[1] = 1;
    HIP_TRY(hipMemcpyAsync(t->d_aux.p, hp, 16, hipMemcpyHostToDevice, t->stream));
    u64 tl[4] = {0, 0, 0, 0};
    KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux.p, (const du64 *)t->d_aux.p + 1, 1, 1, tl));
    return kct_get_hash(t, h, count_out);
}

}  // namespace kcth

using namespace kcth;

// ================================ C ABI =====================================================

extern "C" {

const char *kct_last_error(void) { return g_err; }

int kct_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

kct_status kct_create(uint8_t ksize, uint64_t capacity_hint, int device, kct_table **out) {
    if (!out) { set_err("out is null"); return KCT_ERR_ARG; }
    *out = nullptr;
    if (ksize == 0) { set_err("ksize must be >= 1"); return KCT_ERR_ARG; }
    int ndev = kct_device_count();
    if (ndev <= 0) { set_err("no HIP device visible: the k-mer engine has no CPU fallback"); return KCT_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { set_err("device %d out of range (0..%d)", device, ndev - 1); return KCT_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return KCT_ERR_NO_DEVICE;
    }
    kct_table *t = new (std::nothrow) kct_table();
    if (!t) return KCT_ERR_NOMEM;
    t->device = device;
    t->k = ksize;
    kct_status st = KCT_OK;
    auto fail = [&](kct_status s) { kct_destroy(t); return s; };
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); return fail(KCT_ERR_HIP); }
    t->own_stream = true;
    if (hipMalloc((void **)&t->d_counters, (kNumCounters + 16) * sizeof(u64)) != hipSuccess) { set_err("hipMalloc(counters) failed"); return fail(KCT_ERR_NOMEM); }
    if (hipMemset(t->d_counters, 0, (kNumCounters + 16) * sizeof(u64)) != hipSuccess) { set_err("hipMemset(counters) failed"); return fail(KCT_ERR_HIP); }
    if (hipHostMalloc((void **)&t->h_counters, (kNumCounters + 16) * sizeof(u64), hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc failed"); return fail(KCT_ERR_NOMEM); }
    u64 cap = capacity_hint ? next_pow2((u64)((double)capacity_hint / kMaxLoad) + 1) : kDefaultSlots;
    t->auto_sized = capacity_hint == 0;
    cap = std::max(cap, kMinSlots);
    st = alloc_slots(device, cap, t->stream, &t->slots);
    if (st != KCT_OK) return fail(st);
    t->cap = cap;
    t->slots_alloc = cap;
    set_geometry(t);
    t->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#ifdef KCT_DEBUG_ENV
    if (const char *e = getenv("KCT_ABLATE")) t->tune.ablate = atoi(e);
    if (const char *e = getenv("KCT_PBITS")) t->tune.pbits = atoi(e);
    if (const char *e = getenv("KCT_K1B_LINES")) t->tune.k1b_lines = std::max(1, std::min(4, atoi(e)));
    t->tune.pairs_nopersist = getenv("KCT_PAIRS_NOPERSIST") != nullptr;
    t->tune.k2_nopersist = getenv("KCT_K2_NOPERSIST") != nullptr;
    t->tune.flush_atomic = getenv("KCT_FLUSH_ATOMIC") != nullptr;
    t->tune.k1b_half = getenv("KCT_K1B_HALF") != nullptr;
    if (const char *e = getenv("KCT_SUB_CHUNKS")) t->tune.sub_chunks = std::max(1, std::min(15, atoi(e)));
#endif
    if (const char *e = getenv("KCT_PACK_THREADS")) t->tune.pack_threads = std::max(1, atoi(e));
    if (const char *e = getenv("KCT_K1_FLUSHERS")) { const int f = atoi(e); t->tune.k1_flushers = f == 2 || f == 4 ? f : 0; }
    t->ablate = t->tune.ablate;
    t->debug = getenv("KCT_DEBUG") != nullptr;
    if (hipStreamSynchronize(t->stream) != hipSuccess) { set_err("stream sync failed"); return fail(KCT_ERR_HIP); }
    *out = t;
    return KCT_OK;
}

void kct_destroy(kct_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    for (auto &p : t->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : t->event_pool) (void)hipEventDestroy(e);
    if (t->slots) (void)hipFree(t->slots);
    if (t->d_counters) (void)hipFree(t->d_counters);
    if (t->h_counters) (void)hipHostFree(t->h_counters);
    t->d_stream.release(); t->d_spill.release(); t->d_aux.release(); t->d_aux2.release();
    t->d_scratch.release(); t->d_regions.release(); t->d_irr.release(); t->d_sort.release();
    t->d_scratch2.release(); t->d_regions2.release(); t->d_irr2.release(); t->d_pairs_ovf.release(); t->d_prefix.release(); t->d_pending.release(); t->d_failed.release();
    t->h_stage.release(); t->h_pending.release();
    if (t->shadow) (void)hipFree(t->shadow);
    if (t->shadow32) (void)hipFree(t->shadow32);
    if (t->probe_shadow) (void)hipFree(t->probe_shadow);
    if (t->probe_shadow32) (void)hipFree(t->probe_shadow32);
    if (t->shadow128) (void)hipFree(t->shadow128);
    t->d_unpack.release(); t->d_defer.release();
    for (kcth::DevBuf *b : {&t->d_sk_bases, &t->d_sk_starts, &t->d_sk_meta, &t->d_sk_lists, &t->d_sk_dir, &t->d_sk_send, &t->d_sk_recv, &t->d_sk_inbox}) b->release();
    for (auto &b : t->h_file) b.release();
    if (t->own_stream && t->stream) (void)hipStreamDestroy(t->stream);
    if (t->copy_stream) (void)hipStreamDestroy(t->copy_stream);
    delete t;
}

kct_status kct_clear(kct_table *t) {
    KCT_BORROW(t);
    KCT_TRY(use_device(t));
    t->pending_used = 0; t->pending_records = 0;  // buffered records are forgotten with everything else
    t->defer_used = 0; t->defer_windows = 0;
    t->poisoned = false;
    t->lazy_empty = true;  // the memset is issued by materialize() only if something needs it
    t->shadow_empty = true; t->shadow_dirty = false; t->shadow_keys = 0; t->dedupe_off = false;  // pending counts are forgotten too
    t->s32_empty = true; t->s32_dirty = false; t->s32_keys = 0; t->s32_windows = 0; t->compact_off = false;
    t->s128_empty = true; t->s128_dirty = false; t->s128_keys = 0; t->s128_windows = 0; t->dedupe128_off = false;
    t->n_keys = 0; t->consumed = 0; t->zero_present = false; t->zero_count = 0;
    t->expect_new_keys = false;
    if (t->pending_pairs) { t->pending_pairs = 0; HIP_TRY(hipMemsetAsync(t->d_counters + kNumCounters + 8, 0, 8, t->stream)); }
    return KCT_OK;
}

kct_status kct_reserve(kct_table *t, uint64_t distinct) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 want = next_pow2((u64)((double)distinct / kMaxLoad) + 1);
    t->auto_sized = false;
    if (want > t->cap) return grow_to(t, want);
    return KCT_OK;
}

kct_status kct_resize(kct_table *t, uint64_t distinct) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 want = std::max(next_pow2((u64)((double)std::max<u64>(distinct, t->n_keys) / kMaxLoad) + 1), kMinSlots);
    t->auto_sized = false;
    if (want == t->cap) return KCT_OK;
    // (nothing is pending after use(); shadows of another geometry are re-made by the next pass that wants one)
    if (t->n_keys == 0 || t->lazy_empty) {
        // an empty table changes its capacity IN PLACE when the allocation has room (no hipMalloc / hipFree: the multi-GPU
        // merge resizes twice per job); its slots are cleared lazily
        if (want > t->slots_alloc) {
            HIP_TRY(hipStreamSynchronize(t->stream));
            du64 *fresh = nullptr;
            HIP_TRY(hipMalloc((void **)&fresh, want * 16));
            HIP_TRY(hipFree(t->slots));
            t->slots = fresh; t->slots_alloc = want;
        }
        t->cap = want; set_geometry(t);
        t->n_keys = 0; t->lazy_empty = true;
        return KCT_OK;
    }
    if (want > t->cap) return grow_to(t, want);
    return rehash_into(t, want);  // smaller, with keys: a fresh array of the wanted size, the keys re-inserted
}

kct_status kct_count_hash(kct_table *t, uint64_t hash, uint64_t *count_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 c = 0;
    KCT_TRY(point_add(t, hash, &c));
    if (count_out) *count_out = c;
    return KCT_OK;
}

kct_status kct_get_hash_array(kct_table *t, const uint64_t *hashes, size_t n, uint64_t *counts_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (n == 0) return KCT_OK;
    if (!hashes || !counts_out) { set_err("null argument"); return KCT_ERR_ARG; }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_aux.reserve(n * 16));
    du64 *d_in = (du64 *)t->d_aux.p, *d_out = d_in + n;
    HIP_TRY(hipMemcpyAsync(d_in, hashes, n * 8, hipMemcpyHostToDevice, t->stream));
    {
        ProfScope ps(t, "get_hashes_kernel");
        hipLaunchKernelGGL(kct::get_hashes_kernel, dim3((unsigned)((n + kct::kBlock - 1) / kct::kBlock)), dim3(kct::kBlock), 0, t->stream,
                           (const du64 *)t->slots, geom(t), (const du64 *)d_in, (u64)n, d_out);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts_out, d_out, n * 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    for (size_t i = 0; i < n; ++i)
        if (hashes[i] == 0) counts_out[i] = t->zero_present ? t->zero_count : 0;
    return KCT_OK;
}

kct_status kct_get_hash(kct_table *t, uint64_t hash, uint64_t *count_out) {
    KCT_BORROW(t);
    if (!count_out) { set_err("null argument"); return KCT_ERR_ARG; }
    if (hash == 0) { KCT_TRY(use(t)); *count_out = t->zero_present ? t->zero_count : 0; return KCT_OK; }  // host-side key
    return kct_get_hash_array(t, &hash, 1, count_out);
}

kct_status kct_set_hash(kct_table *t, uint64_t hash, uint64_t count) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (hash == 0) { t->zero_present = true; t->zero_count = count; return KCT_OK; }
    // make sure the key exists (adding 0 creates it without changing its count), then overwrite
    KCT_TRY(maybe_grow(t));
    KCT_TRY(t->h_stage.reserve(64));
    KCT_TRY(t->d_aux.reserve(64));
    u64 *hp = (u64 *)t->h_stage.p;
    hp[0] = hash; hp[1] = 0;
    HIP_TRY(hipMemcpyAsync(t->d_aux.p, hp, 16, hipMemcpyHostToDevice, t->stream));
    u64 tl[4] = {0, 0, 0, 0};
    KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux.p, (const du64 *)t->d_aux.p + 1, 1, 1, tl));
    du64 *d_found = t->d_counters + kNumCounters + 2;
    hipLaunchKernelGGL(kct::set_hash_kernel, dim3(1), dim3(1), 0, t->stream, t->slots, geom(t), (u64)hash, (u64)count, d_found);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    return KCT_OK;
}

kct_status kct_len(kct_table *t, uint64_t *out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    *out = t->n_keys + (t->zero_present ? 1 : 0);
    return KCT_OK;
}

kct_status kct_sum_counts(kct_table *t, uint64_t *out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    KCT_TRY(materialize(t));
    du64 *d_sum = t->d_counters + kNumCounters + 3;
    HIP_TRY(hipMemsetAsync(d_sum, 0, 8, t->stream));
    {
        ProfScope ps(t, "sum_counts_kernel");
        hipLaunchKernelGGL(kct::sum_counts_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t), d_sum);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_sum, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *out = t->h_counters[0] + (t->zero_present ? t->zero_count : 0);
    return KCT_OK;
}

kct_status kct_consumed(kct_table *t, uint64_t *out) {
    KCT_BORROW(t); KCT_TRY(use_device(t)); *out = t->consumed; return KCT_OK; }
kct_status kct_add_consumed(kct_table *t, uint64_t delta) {
    KCT_BORROW(t); KCT_TRY(use_device(t)); t->consumed += delta; return KCT_OK; }
uint8_t kct_ksize(const kct_table *t) { return t ? t->k : 0; }
kct_status kct_capacity(kct_table *t, uint64_t *slots_out) {
    KCT_BORROW(t); KCT_TRY(use(t)); *slots_out = t->cap; return KCT_OK; }

kct_status kct_export_device(kct_table *t, void *d_hashes, void *d_counts, size_t cap, uint64_t *n_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (!n_out || (cap && (!d_hashes || !d_counts))) { set_err("null argument"); return KCT_ERR_ARG; }
    KCT_TRY(materialize(t));
    du64 *d_n = t->d_counters + kNumCounters + 4;
    HIP_TRY(hipMemsetAsync(d_n, 0, 8, t->stream));
    {
        ProfScope ps(t, "compact_kernel");
        hipLaunchKernelGGL(kct::compact_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t), (du64 *)d_hashes,
                           (du64 *)d_counts, (u64)cap, d_n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_n, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *n_out = t->h_counters[0];
    return KCT_OK;
}

kct_status kct_export_by_owner_device(kct_table *t, uint32_t nparts, void *d_pairs, size_t cap, uint64_t *part_counts, uint64_t *n_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (!n_out || !part_counts || (cap && !d_pairs)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (nparts == 0 || nparts > (uint32_t)kct::kMaxParts) { set_err("nparts must be 1..%d", kct::kMaxParts); return KCT_ERR_ARG; }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_aux.reserve((size_t)nparts * 16));
    du64 *d_counts = (du64 *)t->d_aux.p, *d_cursor = d_counts + nparts;
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nparts * 8, t->stream));
    {
        ProfScope ps(t, "count_owners_kernel");
        hipLaunchKernelGGL(kct::count_owners_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t),
                           (unsigned int)nparts, d_counts);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_counts, (size_t)nparts * 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    u64 total = 0;
    std::vector<u64> base(nparts);
    for (uint32_t p = 0; p < nparts; ++p) { part_counts[p] = t->h_counters[p]; base[p] = total; total += part_counts[p]; }
    *n_out = total;
    if (total == 0 || cap == 0) return KCT_OK;
    for (uint32_t p = 0; p < nparts; ++p) t->h_counters[p] = base[p];
    HIP_TRY(hipMemcpyAsync(d_cursor, t->h_counters, (size_t)nparts * 8, hipMemcpyHostToDevice, t->stream));
    {
        ProfScope ps(t, "scatter_owners_kernel");
        const unsigned grid = (unsigned)std::min<u64>((t->cap + 16 * kct::kBlock - 1) / (16 * kct::kBlock), 2048);
        hipLaunchKernelGGL(kct::scatter_owners_kernel, dim3(grid), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t),
                           (unsigned int)nparts, d_cursor, (du64 *)d_pairs, (u64)cap);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    return KCT_OK;
}

kct_status kct_merge_pairs_device(kct_table *t, const void *d_pairs, size_t n, uint64_t *total_added, uint64_t *new_keys) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 tl[4] = {0, 0, 0, 0};
    if (n) {
        if (!d_pairs) { set_err("null argument"); return KCT_ERR_ARG; }
        if ((double)(t->n_keys + n) > kMaxLoad * (double)t->cap) KCT_TRY(grow_to(t, next_pow2((u64)((double)(t->n_keys + n) / kMaxLoad) + 1)));
        KCT_TRY(merge_pairs(t, (const du64 *)d_pairs, (const du64 *)d_pairs + 1, n, 2, tl));
    }
    if (total_added) *total_added = tl[kct::CTR_TOTAL_ADDED];
    if (new_keys) *new_keys = tl[kct::CTR_NEW_BY_ZERO];
    return KCT_OK;
}

kct_status kct_dump(kct_table *t, uint64_t *hashes_out, uint64_t *counts_out, size_t cap, int order, uint64_t *n_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (!n_out) { set_err("null argument"); return KCT_ERR_ARG; }
    const u64 n_dev = t->n_keys;
    const u64 n = n_dev + (t->zero_present ? 1 : 0);
    *n_out = n;
    if (cap == 0 || n == 0) return KCT_OK;
    if (!hashes_out || !counts_out) { set_err("null argument"); return KCT_ERR_ARG; }
    // device -> pinned staging (full-rate D2H) -> the caller's arrays (several threads: the destination is
    // usually freshly allocated, so the copy is page-fault bound)
    KCT_TRY(t->h_stage.reserve(n_dev * 16 + 16));
    const u64 *hk = (const u64 *)t->h_stage.p, *hc = hk + n_dev;
    if (n_dev) {
        // compact on the device, sort on the device (rocPRIM radix sort, stable), copy out
        KCT_TRY(t->d_aux.reserve(n_dev * 16));
        du64 *dk = (du64 *)t->d_aux.p, *dc = dk + n_dev;
        u64 got = 0;
        KCT_TRY(kct_export_device(t, dk, dc, n_dev, &got));
        if (got != n_dev) { set_err("table scan found %llu keys, expected %llu", (unsigned long long)got, (unsigned long long)n_dev); return KCT_ERR_HIP; }
        if (order == 1 || order == 2) {
            KCT_TRY(t->d_aux2.reserve(n_dev * 16));
            du64 *sk = (du64 *)t->d_aux2.p, *sc = sk + n_dev;
            size_t tmp_bytes = 0;
            if (kx_sort_pairs_u64(dk, sk, dc, sc, n_dev, nullptr, &tmp_bytes, t->stream) != 0) { set_err("rocprim size query failed"); return KCT_ERR_HIP; }
            KCT_TRY(t->d_sort.reserve(tmp_bytes + 16));
            {
                ProfScope ps(t, "radix_sort_pairs(by hash)");
                if (kx_sort_pairs_u64(dk, sk, dc, sc, n_dev, t->d_sort.p, &tmp_bytes, t->stream) != 0) { set_err("rocprim radix sort failed"); return KCT_ERR_HIP; }
            }
            if (order == 2) {  // (count, hash): stable sort by count of the hash-sorted pairs (lib.rs:353-356)
                ProfScope ps(t, "radix_sort_pairs(by count)");
                if (kx_sort_pairs_u64(sc, dc, sk, dk, n_dev, t->d_sort.p, &tmp_bytes, t->stream) != 0) { set_err("rocprim radix sort failed"); return KCT_ERR_HIP; }
            } else { dk = sk; dc = sc; }
        }
        HIP_TRY(hipMemcpyAsync((void *)hk, dk, n_dev * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync((void *)hc, dc, n_dev * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    // hash 0 lives host-side: `at` is where the order wants it (n_dev = at the end / not present)
    size_t at = n_dev;
    if (t->zero_present && order == 1) at = 0;
    if (t->zero_present && order == 2) at = (size_t)(std::lower_bound(hc, hc + n_dev, t->zero_count) - hc);  // smallest hash among equal counts
    // output element i comes from staging element i (i < at), the zero key (i == at), staging element i - 1 (i > at)
    const size_t ncopy = std::min<size_t>(cap, n);
    const size_t head = std::min(ncopy, at);
    parallel_memcpy(hashes_out, hk, head * 8);
    parallel_memcpy(counts_out, hc, head * 8);
    if (t->zero_present && at < ncopy) { hashes_out[at] = 0; counts_out[at] = t->zero_count; }
    if (ncopy > at + 1) {
        parallel_memcpy(hashes_out + at + 1, hk + at, (ncopy - at - 1) * 8);
        parallel_memcpy(counts_out + at + 1, hc + at, (ncopy - at - 1) * 8);
    }
    return KCT_OK;
}

kct_status kct_merge_device(kct_table *t, const void *d_hashes, const void *d_counts, size_t n, uint64_t *total_added,
                            uint64_t *new_keys) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 tl[4] = {0, 0, 0, 0};
    if (n) {
        if (!d_hashes || !d_counts) { set_err("null argument"); return KCT_ERR_ARG; }
        // make room up front: at most n new keys
        if ((double)(t->n_keys + n) > kMaxLoad * (double)t->cap) KCT_TRY(grow_to(t, next_pow2((u64)((double)(t->n_keys + n) / kMaxLoad) + 1)));
        KCT_TRY(merge_pairs(t, (const du64 *)d_hashes, (const du64 *)d_counts, n, 1, tl));
    }
    if (total_added) *total_added = tl[kct::CTR_TOTAL_ADDED];
    if (new_keys) *new_keys = tl[kct::CTR_NEW_BY_ZERO];
    return KCT_OK;
}

kct_status kct_merge_host(kct_table *t, const uint64_t *hashes, const uint64_t *counts, size_t n, uint64_t *total_added,
                          uint64_t *new_keys) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (total_added) *total_added = 0;
    if (new_keys) *new_keys = 0;
    if (n == 0) return KCT_OK;
    if (!hashes || !counts) { set_err("null argument"); return KCT_ERR_ARG; }
    // key 0 cannot live on the device: fold it host-side
    u64 zero_total = 0, zero_new = 0;
    for (size_t i = 0; i < n; ++i)
        if (hashes[i] == 0) {
            if (!t->zero_present || t->zero_count == 0) zero_new = 1;
            t->zero_present = true;
            t->zero_count += counts[i];
            zero_total += counts[i];
        }
    KCT_TRY(t->d_aux.reserve(n * 16));
    du64 *dk = (du64 *)t->d_aux.p, *dc = dk + n;
    HIP_TRY(hipMemcpyAsync(dk, hashes, n * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(dc, counts, n * 8, hipMemcpyHostToDevice, t->stream));
    u64 ta = 0, nk = 0;
    KCT_TRY(kct_merge_device(t, dk, dc, n, &ta, &nk));
    if (total_added) *total_added = ta + zero_total;
    if (new_keys) *new_keys = nk + zero_new;
    return KCT_OK;
}

kct_status kct_add(kct_table *dst, kct_table *src, uint64_t *total_added, uint64_t *new_keys) {
    KCT_BORROW(dst);
    KCT_BORROW(src);
    if (!dst || !src) { set_err("null table handle"); return KCT_ERR_ARG; }
    if (dst->k != src->k) { set_err("KmerCountTables must have the same ksize"); return KCT_ERR_KSIZE_MISMATCH; }
    // snapshot src (lib.rs:791-795), then fold it into dst (lib.rs:798-806)
    u64 n = 0;
    KCT_TRY(kct_len(src, &n));
    u64 ta = 0, nk = 0;
    if (dst->device == src->device && dst != src) {
        // both tables on one GPU: compact src into a device buffer of dst's and merge from there -- nothing crosses PCIe
        KCT_TRY(use(dst));  // (anything pending in dst is counted before its d_aux is borrowed)
        const u64 n_dev = src->n_keys;
        if (n_dev) {
            KCT_TRY(dst->d_aux.reserve(n_dev * 16));
            du64 *dk = (du64 *)dst->d_aux.p, *dc = dk + n_dev;
            u64 got = 0;
            KCT_TRY(kct_export_device(src, dk, dc, n_dev, &got));
            if (got != n_dev) { set_err("table scan found %llu keys, expected %llu", (unsigned long long)got, (unsigned long long)n_dev); return KCT_ERR_HIP; }
            KCT_TRY(use_device(dst));
            KCT_TRY(kct_merge_device(dst, dk, dc, n_dev, &ta, &nk));
        }
        if (src->zero_present) {  // key 0 lives host-side in both tables (lib.rs:801-803: new when its current count is 0)
            if (!dst->zero_present || dst->zero_count == 0) ++nk;
            dst->zero_present = true;
            dst->zero_count += src->zero_count;
            ta += src->zero_count;
        }
    } else {
        std::vector<u64> hk(n ? n : 1), hc(n ? n : 1);
        u64 got = 0;
        KCT_TRY(kct_dump(src, hk.data(), hc.data(), n, 0, &got));
        KCT_TRY(kct_merge_host(dst, hk.data(), hc.data(), n, &ta, &nk));
    }
    dst->consumed += src->consumed;  // lib.rs:808
    if (total_added) *total_added = ta;
    if (new_keys) *new_keys = nk;
    return KCT_OK;
}

kct_status kct_set_stream(kct_table *t, void *hip_stream) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    HIP_TRY(hipStreamSynchronize(t->stream));
    prof_collect(t);
    if (t->own_stream) { HIP_TRY(hipStreamDestroy(t->stream)); t->own_stream = false; }
    t->stream = (hipStream_t)hip_stream;
    return KCT_OK;
}

void *kct_get_stream(kct_table *t) { return t ? (void *)t->stream : nullptr; }

kct_status kct_release_scratch(kct_table *t) {
    KCT_BORROW(t);
    KCT_TRY(use(t));  // nothing may be pending in a buffer that is about to go
    HIP_TRY(hipStreamSynchronize(t->stream));
    for (DevBuf *b : {&t->d_stream, &t->d_spill, &t->d_aux, &t->d_aux2, &t->d_scratch, &t->d_regions, &t->d_irr, &t->d_sort, &t->d_scratch2,
                      &t->d_regions2, &t->d_irr2, &t->d_pairs_ovf, &t->d_prefix, &t->d_pending, &t->d_failed, &t->d_sk_bases, &t->d_sk_starts, &t->d_sk_meta,
                      &t->d_sk_lists, &t->d_sk_dir, &t->d_sk_send, &t->d_sk_recv, &t->d_sk_inbox, &t->d_defer})
        b->release();
    if (t->shadow) { (void)hipFree(t->shadow); t->shadow = nullptr; t->shadow_cap = 0; t->shadow_empty = true; t->shadow_keys = 0; }
    if (t->shadow32) { (void)hipFree(t->shadow32); t->shadow32 = nullptr; t->s32_empty = true; t->s32_keys = 0; t->s32_windows = 0; }
    if (t->probe_shadow) { (void)hipFree(t->probe_shadow); t->probe_shadow = nullptr; }
    if (t->probe_shadow32) { (void)hipFree(t->probe_shadow32); t->probe_shadow32 = nullptr; }
    if (t->shadow128) { (void)hipFree(t->shadow128); t->shadow128 = nullptr; t->s128_empty = true; t->s128_keys = 0; t->s128_windows = 0; }
    t->d_unpack.release();
    t->h_stage.release(); t->h_pending.release();
    for (auto &b : t->h_file) b.release();
    return KCT_OK;
}

kct_status kct_sync(kct_table *t) {
    KCT_BORROW(t);
    KCT_DBG(t, "sync: call begins\n");
    KCT_TRY(use(t));  // flushes what deferred mode has buffered
    HIP_TRY(hipStreamSynchronize(t->stream));
    return KCT_OK;
}

kct_status kct_set_deferred(kct_table *t, int on) {
    KCT_BORROW(t);
    KCT_TRY(use(t));  // flushes what is buffered
    t->deferred = on != 0;
    t->defer_device = on != 0;
    return KCT_OK;
}

kct_status kct_set_path(kct_table *t, int mode) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (mode < 0 || mode > 3) { set_err("mode must be 0, 1, 2 or 3"); return KCT_ERR_ARG; }
    if (mode == 3 && t->k > 64) { set_err("the dedupe-first paths need k <= 64"); return KCT_ERR_ARG; }
    t->force_path = mode;
    t->dedupe_off = false;
    t->compact_off = false;
    t->dedupe128_off = false;
    t->dedupe_hint = false;  // what earlier passes taught this table about its input is forgotten too
    return KCT_OK;
}

kct_status kct_profile_enable(kct_table *t, int on) {
    KCT_BORROW(t); KCT_TRY(use(t)); t->prof_on = on != 0; return KCT_OK; }

kct_status kct_profile_reset(kct_table *t) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    prof_collect(t);
    t->prof.clear();
    return KCT_OK;
}

kct_status kct_profile_read(kct_table *t, int index, char *name_out, size_t name_cap, uint64_t *launches, double *total_ms) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    prof_collect(t);
    if (index < 0 || (size_t)index >= t->prof.size()) return KCT_ERR_ARG;
    const ProfEntry &e = t->prof[index];
    if (name_out && name_cap) { strncpy(name_out, e.name.c_str(), name_cap - 1); name_out[name_cap - 1] = 0; }
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    return KCT_OK;
}

}  // extern "C"

