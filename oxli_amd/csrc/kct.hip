// kct.hip -- host side of libkct_hip.so: the C ABI declared in include/kct.h.
//
// Owns the device-resident table (HBM), the staging buffers and the launch policy.  All
// hashing and counting happens in the gfx950 kernels of kernels.h; nothing here falls back
// to the CPU -- without a device every entry point fails.
#include <hip/hip_runtime.h>

#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/kct.h"
#include "kernels.h"

typedef uint64_t u64;     // host-side 64-bit values (matches the ABI's uint64_t)
typedef kct::u64 du64;    // words that live in device memory (unsigned long long, what HIP atomics take)

extern "C" int kx_sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned long long *vals_in,
                                 unsigned long long *vals_out, size_t n, void *tmp, size_t *tmp_bytes, void *stream);  // sort.hip

namespace {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

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

u64 next_pow2(u64 v) {
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
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct ProfEntry { std::string name; u64 launches = 0; double ms = 0; };
struct ProfPending { int entry; hipEvent_t a, b; };

}  // namespace

struct kct_table {
    int device = 0;
    uint8_t k = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    du64 *slots = nullptr;  // 2 * cap words (device)
    u64 cap = 0;
    int block_bits = 0;     // log2(slots per probing block) = min(13, log2 cap)
    bool lazy_empty = false;  // kct_clear() was called and the memset has not been issued yet
    int num_cus = 256;
    bool auto_sized = true; // no capacity hint / reserve yet: bulk ingest ramps its launch size up with the table
    int ablate = 0;         // KCT_ABLATE at create time: measurement-only switches that skip work (results invalid)
    bool debug = false;     // KCT_DEBUG at create time: one stderr line per partitioned pass
    int force_path = 0;     // 0 = choose per pass, 1 = direct atomic kernel only, 2 = partitioned whenever the geometry allows
    u64 n_keys = 0;        // distinct non-zero hashes in `slots`
    u64 consumed = 0;      // lib.rs:36
    bool zero_present = false;  // key 0 lives host-side (0 is the EMPTY sentinel on the device)
    u64 zero_count = 0;

    du64 *d_counters = nullptr;  // kNumCounters tallies + 8 scratch words (device)
    u64 *h_counters = nullptr;   // pinned mirror
    DevBuf d_stream, d_spill, d_aux, d_aux2, d_scratch, d_regions, d_irr, d_sort, d_scratch2, d_regions2, d_irr2;
    PinnedBuf h_stage;

    bool prof_on = false;
    std::vector<ProfEntry> prof;
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> event_pool;
};

namespace {

// ---- profiling: HIP events around each launch, on the stream the kernel runs on ---------------------
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

kct_status use(kct_table *t) {
    if (!t) { set_err("null table handle"); return KCT_ERR_ARG; }
    HIP_TRY(hipSetDevice(t->device));
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
    return v;
}

int log2_u64(u64 v) { int b = 0; while ((1ULL << b) < v) ++b; return b; }

void set_geometry(kct_table *t) { t->block_bits = std::min(kct::kBlockBitsMax, log2_u64(t->cap)); }  // cap >= kMinSlots = 1024 > one group

// kct_clear() defers its memset: the partitioned path rewrites every block from zeros anyway.
// Anything else that touches `slots` calls this first.
kct_status materialize(kct_table *t) {
    if (t->lazy_empty) {
        HIP_TRY(hipMemsetAsync(t->slots, 0, t->cap * 16, t->stream));
        t->lazy_empty = false;
    }
    return KCT_OK;
}

kct_status zero_counters(kct_table *t) {
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 8) * sizeof(u64), t->stream));
    return KCT_OK;
}

// copies the tallies back and folds the shards; waits for the stream
kct_status read_counters(kct_table *t, u64 out[4], u64 *spill_n) {
    HIP_TRY(hipMemcpyAsync(t->h_counters, t->d_counters, (kNumCounters + 8) * sizeof(u64), hipMemcpyDeviceToHost, t->stream));
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

int merge_grid(u64 n) { return (int)std::min<u64>((n + kct::kBlock - 1) / kct::kBlock, 256 * 8); }

// Folds n (hash, count) pairs into the table, growing and replaying the spill list until every
// pair is placed.  tallies[] accumulates CTR_* sums.  `stride` 1 = separate arrays, 2 = slot array.
kct_status grow_to(kct_table *t, u64 new_cap);

kct_status merge_pairs(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]) {
    KCT_TRY(materialize(t));
    while (n > 0) {
        KCT_TRY(t->d_spill.reserve(n * 16));
        KCT_TRY(zero_counters(t));
        {
            ProfScope ps(t, "merge_pairs_kernel");
            hipLaunchKernelGGL(kct::merge_pairs_kernel, dim3(merge_grid(n)), dim3(kct::kBlock), 0, t->stream, d_keys, d_counts, n,
                               (const du64 *)nullptr, (const du64 *)nullptr, stride, view(t, n), t->d_counters);
        }
        HIP_TRY(hipGetLastError());
        u64 c[4], spilled;
        KCT_TRY(read_counters(t, c, &spilled));
        for (int i = 0; i < 4; ++i) tallies[i] += c[i];
        t->n_keys += c[kct::CTR_NEWKEYS];
        if (spilled == 0) break;
        // the table is too full for these keys: move the spill list aside, grow, replay it
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(grow_to(t, next_pow2((u64)((double)(t->n_keys + spilled) / kMaxLoad) + 1)));
        d_keys = (const du64 *)t->d_aux2.p;
        d_counts = d_keys + 1;
        stride = 2;
        n = spilled;
    }
    return KCT_OK;
}

// Re-hash into a table of new_cap slots (no-op if not larger).
kct_status grow_to(kct_table *t, u64 new_cap) {
    new_cap = std::max(next_pow2(new_cap), kMinSlots);
    if (new_cap <= t->cap) new_cap = t->cap * 2;
    du64 *old = t->slots;
    const u64 old_cap = t->cap, old_keys = t->n_keys;
    const kct::TableGeom old_g = geom(t);
    du64 *fresh = nullptr;
    KCT_TRY(alloc_slots(t->device, new_cap, t->stream, &fresh));
    t->slots = fresh;
    t->cap = new_cap;
    set_geometry(t);
    t->n_keys = 0;
    if (t->lazy_empty) t->lazy_empty = false;  // the old array was never cleaned, but it holds no keys: drop it
    if (old && old_keys > 0) {
        // re-insert every occupied slot; the new table is at most ~half full so nothing spills
        KCT_TRY(t->d_spill.reserve(16));
        KCT_TRY(zero_counters(t));
        {
            ProfScope ps(t, "rehash_kernel");
            hipLaunchKernelGGL(kct::rehash_kernel, dim3(merge_grid(old_cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)old, old_g,
                               view(t, 0), t->d_counters);
        }
        HIP_TRY(hipGetLastError());
        u64 c[4], spilled;
        KCT_TRY(read_counters(t, c, &spilled));
        t->n_keys = c[kct::CTR_NEWKEYS];
        if (spilled != 0 || t->n_keys != old_keys) {
            set_err("re-hash lost keys: %llu of %llu placed, %llu spilled", (unsigned long long)t->n_keys,
                    (unsigned long long)old_keys, (unsigned long long)spilled);
            return KCT_ERR_HIP;
        }
    } else {
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
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

template <template <int, int> class Launcher, class... Args>
void dispatch_k(int k, Args &&...args) {
    if (k == 21) Launcher<1, 21>::run(args...);
    else if (k == 31) Launcher<1, 31>::run(args...);
    else if (k == 51) Launcher<2, 51>::run(args...);
    else if (k <= 32) Launcher<1, 0>::run(args...);
    else if (k <= 64) Launcher<2, 0>::run(args...);
    else Launcher<0, 0>::run(args...);
}

template <int KW, int KC>
struct CountLauncher {
    static void run(hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, int k, kct::TableView tv, du64 *ctr) {
        hipLaunchKernelGGL((kct::count_windows_kernel<KW, KC>), dim3(grid), dim3(kct::kBlock), 0, s, stream, nbytes, k, tv, ctr);
    }
};

template <int KW, int KC>
struct HashLauncher {
    static void run(hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, int k, u64 nwin, du64 *out, du64 *fb) {
        hipLaunchKernelGGL((kct::hash_windows_kernel<KW, KC>), dim3(grid), dim3(kct::kBlock), 0, s, stream, nbytes, k, nwin, out, fb);
    }
};

template <int KW, int KC>
struct PartitionLauncher {
    static void run(hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, int k, u64 ntiles, kct::PartitionArgs a) {
        hipLaunchKernelGGL((kct::partition_windows_kernel<KW, KC>), dim3(grid), dim3(kct::kPartThreads), 0, s, stream, nbytes, k, ntiles, a);
    }
};

// Replays a spill list (already copied to d_aux2) after growing; adds what it counted to *n_out.
kct_status replay_spill(kct_table *t, u64 spilled, u64 *n_out) {
    KCT_TRY(grow_to(t, next_pow2((u64)((double)(t->n_keys + spilled) / kMaxLoad) + 1)));
    u64 tl[4] = {0, 0, 0, 0};
    KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux2.p, (const du64 *)t->d_aux2.p + 1, spilled, 2, tl));
    *n_out += tl[kct::CTR_TOTAL_ADDED];
    return KCT_OK;
}

// The partitioned path pays 16 B (one level) or 32 B (two levels) of streaming scratch traffic per
// k-mer plus 32 B per table slot per pass; the direct path pays one memory-side atomic per k-mer.
// It wins once a pass brings a fair fraction as many windows as the table has slots.
bool partition_geometry_ok(const kct_table *t) {
    const u64 nblocks = t->cap >> t->block_bits;
    return nblocks >= 16 && t->cap <= (1ULL << 32);
}

bool partition_pays(const kct_table *t, u64 npos) {
    const u64 nblocks = t->cap >> t->block_bits;
    if (!partition_geometry_ok(t) || npos < (1ULL << 20)) return false;
    return nblocks <= 1024 ? npos >= t->cap / 4 : npos >= t->cap / 2;
}

unsigned int region_capacity(double avg) {
    return (unsigned int)((((u64)(avg * 1.15 + 8.0 * __builtin_sqrt(avg) + 64.0)) + 7) & ~7ULL);
}

// One pass of the partitioned path over window starts [0, npos) of d_stream.  *handled = false
// (and nothing counted) if the pass had to be abandoned; the caller then uses the direct path.
kct_status consume_partitioned(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled) {
    *handled = false;
    const int k = t->k;
    const int bbits = log2_u64(t->cap >> t->block_bits);      // log2(table blocks)
    const bool two_level = bbits > 10;
    const int pbits = two_level ? 10 : bbits;                 // K1 fans out to 2^pbits bins ...
    const int sub_bits = bbits - pbits;                       // ... each holding 2^sub_bits table blocks
    const u64 P = 1ULL << pbits, B = 1ULL << bbits;
    const int nwg = t->num_cus;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile;
    const u64 tiles_per_wg = (ntiles + nwg - 1) / nwg;
    const unsigned int region_cap = region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)P);
    const unsigned int ovf_cap = (unsigned int)std::max<u64>(4096, tiles_per_wg * kct::kPartTile / 8);
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 8));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve((u64)nwg * ovf_cap * 8 + (u64)nwg * 4));
    KCT_TRY(t->d_spill.reserve(npos * 16));
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + (u64)nwg * ovf_cap);
    const bool fresh = t->lazy_empty;

    kct::PartitionArgs pa;
    pa.mask = t->cap - 1; pa.block_bits = t->block_bits + sub_bits; pa.pbits = pbits;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = t->ablate;  // measurement only; wrong counts when set
    {
        ProfScope ps(t, "partition_windows_kernel");
        dispatch_k<PartitionLauncher>(k, t->stream, nwg, d_stream, chunk_bytes, k, ntiles, pa);
    }
    HIP_TRY(hipGetLastError());

    kct::AggregateArgs aa;
    aa.words = t->slots; aa.block_bits = t->block_bits; aa.pbits = bbits;
    aa.fresh = fresh ? 1 : 0; aa.overflow = d_overflow; aa.ablate = pa.ablate;
    aa.spill = (du64 *)t->d_spill.p; aa.spill_cap = npos; aa.spill_n = t->d_counters + kNumCounters; aa.counters = t->d_counters;
    unsigned int ovf2_cap = 0, *d_ovf2_count = nullptr;
    if (!two_level) {
        aa.scratch = (const du64 *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
        aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    } else {
        // second level: one workgroup per super-bin spreads its hashes over the super-bin's blocks
        const unsigned int out_cap = region_capacity((double)npos / (double)B);
        ovf2_cap = (unsigned int)std::max<u64>(4096, npos / P / 8);
        KCT_TRY(t->d_scratch2.reserve(B * out_cap * 8));
        KCT_TRY(t->d_regions2.reserve(B * 4));
        KCT_TRY(t->d_irr2.reserve(P * ovf2_cap * 8 + P * 4));
        d_ovf2_count = (unsigned int *)((du64 *)t->d_irr2.p + P * ovf2_cap);
        kct::RepartitionArgs ra;
        ra.mask = t->cap - 1; ra.block_bits = t->block_bits; ra.sub_bits = sub_bits;
        ra.in = (const du64 *)t->d_scratch.p; ra.in_cap = region_cap; ra.in_count = (const unsigned int *)t->d_regions.p;
        ra.nseg = nwg; ra.nbins = (int)P;
        ra.out = (du64 *)t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
        ra.ovf = (du64 *)t->d_irr2.p; ra.ovf_cap = ovf2_cap; ra.ovf_count = d_ovf2_count; ra.overflow = d_overflow;
        {
            ProfScope ps(t, "repartition_kernel");
            hipLaunchKernelGGL(kct::repartition_kernel, dim3((unsigned)P), dim3(kct::kPartThreads), 0, t->stream, ra);
        }
        HIP_TRY(hipGetLastError());
        aa.scratch = (const du64 *)t->d_scratch2.p; aa.seg_stride = 0; aa.block_stride = out_cap;
        aa.region_count = (const unsigned int *)t->d_regions2.p; aa.nregions = 1;
    }
    {
        ProfScope ps(t, "aggregate_blocks_kernel");
        hipLaunchKernelGGL(kct::aggregate_blocks_kernel, dim3((unsigned)B), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    {
        // fold the overflow regions with the direct atomic path; the kernel reads the region lengths
        // and the abandon flag from device memory, so no host round trip sits between the launches
        ProfScope ps(t, "merge_overflow_kernel");
        hipLaunchKernelGGL(kct::merge_overflow_kernel, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                           (const unsigned int *)d_ovf_count, nwg, ovf_cap, (const du64 *)d_overflow, view(t, npos), t->d_counters);
        if (two_level)
            hipLaunchKernelGGL(kct::merge_overflow_kernel, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr2.p,
                               (const unsigned int *)d_ovf2_count, (int)P, ovf2_cap, (const du64 *)d_overflow, view(t, npos), t->d_counters);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    if (t->debug) {
        std::vector<unsigned int> oc(nwg);
        (void)hipMemcpy(oc.data(), d_ovf_count, nwg * 4, hipMemcpyDeviceToHost);
        u64 tot = 0;
        for (auto v : oc) tot += v;
        fprintf(stderr, "[kct] partitioned pass: npos=%llu blocks=%llu levels=%d region_cap=%u overflow(K1)=%llu counted=%llu merged=%llu spilled=%llu abandon=%llu\n",
                (unsigned long long)npos, (unsigned long long)B, two_level ? 2 : 1, region_cap, (unsigned long long)tot,
                (unsigned long long)c[kct::CTR_COUNTED], (unsigned long long)c[kct::CTR_TOTAL_ADDED], (unsigned long long)spilled,
                (unsigned long long)t->h_counters[kNumCounters + 6]);
    }
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // abandoned: K2 and the merges exited early, nothing was touched
    t->lazy_empty = false;
    *handled = true;
    *n_out += c[kct::CTR_COUNTED] + c[kct::CTR_TOTAL_ADDED];
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (spilled) {
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, n_out));
    }
    return KCT_OK;
}

// Counts every good window of a device-resident record stream.  *n_out = k-mers counted.
kct_status consume_stream(kct_table *t, const unsigned char *d_stream, u64 nbytes, u64 *n_out) {
    *n_out = 0;
    const int k = t->k;
    if (nbytes < (u64)k) return KCT_OK;
    u64 done = 0;
    const u64 last_start = nbytes - k;  // last window start position
    const u64 cap_at_entry = t->cap;
    // Launch chunk.  The partitioned path on a large table re-reads and re-writes every table block
    // once per pass, so it wants passes of several windows per slot; its scratch + spill lists cost
    // ~36 B per window start, which bounds the pass by HBM (this is what 288 GB is for).  Decided
    // once per call: buffers this table already holds are reused, so they count as available.
    u64 chunk_limit = kChunkPositions;
    if (t->force_path != 1 && partition_geometry_ok(t) && (t->cap >> t->block_bits) > 1024) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const u64 held = t->d_scratch.cap + t->d_scratch2.cap + t->d_spill.cap + t->d_irr.cap + t->d_irr2.cap;
            const u64 by_mem = ((u64)free_b + held) / 2 / 36;
            chunk_limit = std::max<u64>(kChunkPositions, std::min<u64>(4 * t->cap, by_mem));
            chunk_limit &= ~(u64)0xFFFF;  // keeps `d_stream + done` 16-byte aligned
        }
    }
    while (done <= last_start) {
        KCT_TRY(maybe_grow(t));
        // a chunk owns window starts [done, done + npos); its loads reach k-1 bytes further
        // A table that was never sized by its owner starts tiny: feed it launches of at most a few windows
        // per slot, so that what cannot be placed (and must be replayed after growing) stays small while
        // the table finds its size; launches grow with it.
        const u64 ramp = t->auto_sized ? std::max<u64>(1ULL << 20, 4 * t->cap) : ~0ULL;
        const u64 npos = std::min<u64>({chunk_limit, ramp, last_start + 1 - done});
        const u64 chunk_bytes = std::min<u64>(nbytes - done, npos + k - 1);
        if (partition_geometry_ok(t) && t->force_path != 1 && (t->force_path == 2 || partition_pays(t, npos))) {
            bool handled = false;
            KCT_TRY(consume_partitioned(t, d_stream + done, chunk_bytes, npos, n_out, &handled));
            if (handled) { done += npos; continue; }
        }
        KCT_TRY(materialize(t));
        KCT_TRY(t->d_spill.reserve(npos * 16));
        KCT_TRY(zero_counters(t));
        const int grid = (int)((npos + kct::kTile - 1) / kct::kTile);
        // the kernel derives window ownership from tile positions, so hand it a stream that ends
        // where this chunk's last window ends
        {
            ProfScope ps(t, "count_windows_kernel");
            dispatch_k<CountLauncher>(k, t->stream, grid, d_stream + done, chunk_bytes, k, view(t, npos), t->d_counters);
        }
        HIP_TRY(hipGetLastError());
        u64 c[4], spilled;
        KCT_TRY(read_counters(t, c, &spilled));
        *n_out += c[kct::CTR_COUNTED];
        t->n_keys += c[kct::CTR_NEWKEYS];
        if (spilled) {
            KCT_TRY(t->d_aux2.reserve(spilled * 16));
            HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
            KCT_TRY(replay_spill(t, spilled, n_out));
        }
        done += npos;
    }
    if (t->auto_sized && t->cap == cap_at_entry && nbytes >= (1u << 20)) t->auto_sized = false;  // the table has found its size
    return KCT_OK;
}

// host bytes -> pinned staging -> device stream buffer (padded with '\n' to a multiple of 16)
kct_status upload_stream(kct_table *t, size_t nbytes) {
    const size_t padded = (nbytes + 15) & ~(size_t)15;
    KCT_TRY(t->d_stream.reserve(padded + 16));
    HIP_TRY(hipMemcpyAsync(t->d_stream.p, t->h_stage.p, padded, hipMemcpyHostToDevice, t->stream));
    return KCT_OK;
}

kct_status stage_single(kct_table *t, const char *seq, size_t len) {
    const size_t padded = (len + 15) & ~(size_t)15;
    KCT_TRY(t->h_stage.reserve(padded + 16));
    memcpy(t->h_stage.p, seq, len);
    memset((char *)t->h_stage.p + len, '\n', padded + 16 - len);
    return upload_stream(t, len);
}

// hashes of all windows of the staged stream [0, nbytes) into d_aux; returns first bad window index
kct_status hash_stream(kct_table *t, u64 nbytes, u64 nwin, u64 *first_bad) {
    KCT_TRY(t->d_aux.reserve(nwin * 8));
    du64 *d_fb = t->d_counters + kNumCounters + 1;  // scratch word 1
    HIP_TRY(hipMemsetAsync(d_fb, 0xFF, 8, t->stream));
    const int grid = (int)((nwin + kct::kTile - 1) / kct::kTile);
    {
        ProfScope ps(t, "hash_windows_kernel");
        dispatch_k<HashLauncher>((int)t->k, t->stream, grid, (const unsigned char *)t->d_stream.p, nbytes, (int)t->k, nwin,
                                 (du64 *)t->d_aux.p, d_fb);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_fb, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *first_bad = t->h_counters[0] == ~0ULL ? nwin : t->h_counters[0];
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

}  // namespace

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
    if (hipMalloc((void **)&t->d_counters, (kNumCounters + 8) * sizeof(u64)) != hipSuccess) { set_err("hipMalloc(counters) failed"); return fail(KCT_ERR_NOMEM); }
    if (hipHostMalloc((void **)&t->h_counters, (kNumCounters + 8) * sizeof(u64), hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc failed"); return fail(KCT_ERR_NOMEM); }
    u64 cap = capacity_hint ? next_pow2((u64)((double)capacity_hint / kMaxLoad) + 1) : kDefaultSlots;
    t->auto_sized = capacity_hint == 0;
    cap = std::max(cap, kMinSlots);
    st = alloc_slots(device, cap, t->stream, &t->slots);
    if (st != KCT_OK) return fail(st);
    t->cap = cap;
    set_geometry(t);
    t->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *e = getenv("KCT_ABLATE")) t->ablate = atoi(e);
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
    t->d_scratch2.release(); t->d_regions2.release(); t->d_irr2.release();
    t->h_stage.release();
    if (t->own_stream && t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

kct_status kct_clear(kct_table *t) {
    KCT_TRY(use(t));
    t->lazy_empty = true;  // the memset is issued by materialize() only if something needs it
    t->n_keys = 0; t->consumed = 0; t->zero_present = false; t->zero_count = 0;
    return KCT_OK;
}

kct_status kct_reserve(kct_table *t, uint64_t distinct) {
    KCT_TRY(use(t));
    u64 want = next_pow2((u64)((double)distinct / kMaxLoad) + 1);
    t->auto_sized = false;
    if (want > t->cap) return grow_to(t, want);
    return KCT_OK;
}

kct_status kct_hash_windows(kct_table *t, const char *seq, size_t len, uint64_t *hashes_out, size_t cap, uint64_t *n_windows,
                            uint64_t *first_bad) {
    KCT_TRY(use(t));
    if ((!seq && len) || !n_windows || !first_bad) { set_err("null argument"); return KCT_ERR_ARG; }
    const u64 nwin = len >= t->k ? len - t->k + 1 : 0;
    *n_windows = nwin;
    *first_bad = nwin;
    if (nwin == 0) return KCT_OK;
    KCT_TRY(stage_single(t, seq, len));
    KCT_TRY(hash_stream(t, len, nwin, first_bad));
    const size_t ncopy = std::min<size_t>(cap, nwin);
    if (ncopy && hashes_out) HIP_TRY(hipMemcpy(hashes_out, t->d_aux.p, ncopy * 8, hipMemcpyDeviceToHost));
    return KCT_OK;
}

kct_status kct_hash_kmer(kct_table *t, const char *kmer, size_t len, uint64_t *hash_out) {
    KCT_TRY(use(t));
    if (!kmer || !hash_out) { set_err("null argument"); return KCT_ERR_ARG; }
    if ((uint8_t)len != t->k) { set_err("wrong ksize"); return KCT_ERR_WRONG_KSIZE; }  // lib.rs:66 `len as u8`
    u64 nwin, fb, h = 0;
    KCT_TRY(kct_hash_windows(t, kmer, t->k, &h, 1, &nwin, &fb));  // first window only (lib.rs:78 `.next()`)
    if (fb == 0) { set_err("invalid DNA character in k-mer"); return KCT_ERR_INVALID_DNA; }
    *hash_out = h;
    return KCT_OK;
}

kct_status kct_count_hash(kct_table *t, uint64_t hash, uint64_t *count_out) {
    KCT_TRY(use(t));
    u64 c = 0;
    KCT_TRY(point_add(t, hash, &c));
    if (count_out) *count_out = c;
    return KCT_OK;
}

kct_status kct_count(kct_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    KCT_TRY(use(t));
    if ((uint8_t)len != t->k) { set_err("kmer size does not match count table ksize"); return KCT_ERR_WRONG_KSIZE; }
    u64 h;
    KCT_TRY(kct_hash_kmer(t, kmer, len, &h));
    u64 c = 0;
    KCT_TRY(point_add(t, h, &c));
    t->consumed += len;  // lib.rs:153
    if (count_out) *count_out = c;
    return KCT_OK;
}

kct_status kct_get_hash_array(kct_table *t, const uint64_t *hashes, size_t n, uint64_t *counts_out) {
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
    if (!count_out) { set_err("null argument"); return KCT_ERR_ARG; }
    if (hash == 0) { KCT_TRY(use(t)); *count_out = t->zero_present ? t->zero_count : 0; return KCT_OK; }  // host-side key
    return kct_get_hash_array(t, &hash, 1, count_out);
}

kct_status kct_get(kct_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    KCT_TRY(use(t));
    if ((uint8_t)len != t->k) { set_err("kmer size does not match count table ksize"); return KCT_ERR_WRONG_KSIZE; }
    u64 h;
    KCT_TRY(kct_hash_kmer(t, kmer, len, &h));
    return kct_get_hash(t, h, count_out);
}

kct_status kct_set_hash(kct_table *t, uint64_t hash, uint64_t count) {
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

kct_status kct_consume(kct_table *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out) {
    KCT_TRY(use(t));
    if ((!seq && len) || !n_out) { set_err("null argument"); return KCT_ERR_ARG; }
    *n_out = 0;
    const u64 k = t->k;
    if (len < k) { t->consumed += len; return KCT_OK; }  // zero windows (lib.rs: max_index = 0), consumed still grows
    KCT_TRY(stage_single(t, seq, len));
    u64 use_bytes = len;
    bool bad = false;
    if (!skip_bad) {
        const u64 nwin = len - k + 1;
        u64 fb;
        KCT_TRY(hash_stream(t, len, nwin, &fb));  // validity of every window, on the device
        if (fb < nwin) { bad = true; use_bytes = fb + k - 1; }  // windows 0..fb-1 end before byte fb+k-1
    }
    KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, use_bytes, n_out));
    if (bad) { set_err("bad k-mer encountered at position %llu", (unsigned long long)*n_out); return KCT_ERR_BAD_KMER; }
    t->consumed += len;
    return KCT_OK;
}

kct_status kct_consume_batch(kct_table *t, const char *bytes, const uint64_t *offsets, size_t nrec, int skip_bad,
                             uint64_t *n_total, uint64_t *bad_record, uint64_t *bad_position) {
    KCT_TRY(use(t));
    if (!n_total || (nrec && !offsets)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (nrec && !bytes && offsets[nrec] != offsets[0]) { set_err("null argument"); return KCT_ERR_ARG; }  // all-empty records need no bytes
    *n_total = 0;
    if (bad_record) *bad_record = nrec;
    if (bad_position) *bad_position = 0;
    if (nrec == 0) return KCT_OK;
    const u64 total = offsets[nrec] - offsets[0];
    const u64 stream_len = total + nrec;  // one '\n' after every record
    const size_t padded = (stream_len + 15) & ~(size_t)15;
    const size_t off_bytes = skip_bad ? 0 : (nrec + 1) * 8;
    KCT_TRY(t->h_stage.reserve(padded + 16 + off_bytes));
    char *dst = (char *)t->h_stage.p;
    u64 *rec_off = (u64 *)(dst + padded + 16);  // 16-aligned since padded is
    // Pack the records into the record stream: record r lands at (offsets[r] - offsets[0]) + r, one
    // separator behind it.  Positions are known up front, so large batches are packed by several threads.
    for (size_t r = 0; r < nrec; ++r)
        if (offsets[r + 1] < offsets[r]) { set_err("offsets must be non-decreasing"); return KCT_ERR_ARG; }
    const u64 base0 = offsets[0];
    auto pack_range = [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const u64 n = offsets[r + 1] - offsets[r], w = (offsets[r] - base0) + r;
            if (!skip_bad) rec_off[r] = w;
            memcpy(dst + w, bytes + offsets[r], n);
            dst[w + n] = '\n';
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = stream_len >= (8u << 20) ? std::min<size_t>({(size_t)8, hw ? hw : 1, nrec}) : 1;
    if (nthreads <= 1) pack_range(0, nrec);
    else {
        std::vector<std::thread> pool;
        for (size_t i = 0; i < nthreads; ++i) {
            // split by bytes, not by record count: records may be ragged
            const u64 lo = base0 + total * i / nthreads, hi = base0 + total * (i + 1) / nthreads;
            const size_t r0 = (size_t)(std::lower_bound(offsets, offsets + nrec, lo) - offsets);
            const size_t r1 = i + 1 == nthreads ? nrec : (size_t)(std::lower_bound(offsets, offsets + nrec, hi) - offsets);
            if (r1 > r0) pool.emplace_back(pack_range, r0, r1);
        }
        for (auto &th : pool) th.join();
    }
    const u64 w = stream_len;
    if (!skip_bad) rec_off[nrec] = w;
    memset(dst + w, '\n', padded + 16 - w);
    KCT_TRY(upload_stream(t, stream_len));

    if (!skip_bad) {
        KCT_TRY(t->d_aux.reserve(off_bytes));
        HIP_TRY(hipMemcpyAsync(t->d_aux.p, rec_off, off_bytes, hipMemcpyHostToDevice, t->stream));
        du64 *d_q = t->d_counters + kNumCounters + 1;
        HIP_TRY(hipMemsetAsync(d_q, 0xFF, 8, t->stream));
        const u64 nthreads = (stream_len + 15) / 16;
        {
            ProfScope ps(t, "first_bad_byte_kernel");
            hipLaunchKernelGGL(kct::first_bad_byte_kernel, dim3((unsigned)((nthreads + kct::kBlock - 1) / kct::kBlock)), dim3(kct::kBlock), 0,
                               t->stream, (const unsigned char *)t->d_stream.p, stream_len, (int)t->k, (const du64 *)t->d_aux.p, (u64)nrec, d_q);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_q, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const u64 q = t->h_counters[0];
        if (q != ~0ULL) {
            // Record r holds q.  The per-record loop the reference runs would count records
            // [0, r) whole, then the windows of r before its first bad one, then raise.
            const size_t r = (size_t)(std::upper_bound(rec_off, rec_off + nrec + 1, q) - rec_off) - 1;
            const u64 in_rec = q - rec_off[r];
            const u64 fbw = in_rec + 1 >= t->k ? in_rec + 1 - t->k : 0;  // index of r's first bad window
            u64 n_before = 0, n_prefix = 0;
            KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, rec_off[r], &n_before));
            const u64 prefix = fbw + t->k - 1;  // bytes of r that its windows 0..fbw-1 cover
            if (fbw > 0) {
                KCT_TRY(t->d_aux2.reserve(((prefix + 15) & ~(u64)15) + 16));
                HIP_TRY(hipMemcpyAsync(t->d_aux2.p, (const char *)t->d_stream.p + rec_off[r], prefix, hipMemcpyDeviceToDevice, t->stream));
                KCT_TRY(consume_stream(t, (const unsigned char *)t->d_aux2.p, prefix, &n_prefix));
            }
            t->consumed += offsets[r] - offsets[0];  // r raised before lib.rs:604
            *n_total = n_before + n_prefix;
            if (bad_record) *bad_record = r;
            if (bad_position) *bad_position = n_prefix;
            set_err("bad k-mer encountered at position %llu (record %llu)", (unsigned long long)n_prefix, (unsigned long long)r);
            return KCT_ERR_BAD_KMER;
        }
    }
    KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, stream_len, n_total));
    t->consumed += total;
    return KCT_OK;
}

kct_status kct_consume_device(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes, uint64_t *n_total) {
    KCT_TRY(use(t));
    if (!n_total || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    KCT_TRY(consume_stream(t, (const unsigned char *)d_stream, nbytes, n_total));
    t->consumed += consumed_bytes;
    return KCT_OK;
}

kct_status kct_len(kct_table *t, uint64_t *out) {
    KCT_TRY(use(t));
    *out = t->n_keys + (t->zero_present ? 1 : 0);
    return KCT_OK;
}

kct_status kct_sum_counts(kct_table *t, uint64_t *out) {
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

kct_status kct_consumed(kct_table *t, uint64_t *out) { KCT_TRY(use(t)); *out = t->consumed; return KCT_OK; }
kct_status kct_add_consumed(kct_table *t, uint64_t delta) { KCT_TRY(use(t)); t->consumed += delta; return KCT_OK; }
uint8_t kct_ksize(const kct_table *t) { return t ? t->k : 0; }
kct_status kct_capacity(kct_table *t, uint64_t *slots_out) { KCT_TRY(use(t)); *slots_out = t->cap; return KCT_OK; }

kct_status kct_export_device(kct_table *t, void *d_hashes, void *d_counts, size_t cap, uint64_t *n_out) {
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
    KCT_TRY(use(t));
    if (!n_out) { set_err("null argument"); return KCT_ERR_ARG; }
    const u64 n_dev = t->n_keys;
    const u64 n = n_dev + (t->zero_present ? 1 : 0);
    *n_out = n;
    if (cap == 0 || n == 0) return KCT_OK;
    if (!hashes_out || !counts_out) { set_err("null argument"); return KCT_ERR_ARG; }
    std::vector<u64> hk(n), hc(n);
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
        HIP_TRY(hipMemcpyAsync(hk.data(), dk, n_dev * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(hc.data(), dc, n_dev * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    if (t->zero_present) {  // hash 0 lives host-side: put it where the order wants it
        size_t at = n_dev;
        if (order == 1) at = 0;
        else if (order == 2) {
            at = 0;
            while (at < n_dev && hc[at] < t->zero_count) ++at;  // smallest hash among equal counts
        }
        hk.insert(hk.begin() + at, 0); hk.pop_back();
        hc.insert(hc.begin() + at, t->zero_count); hc.pop_back();
    }
    const size_t ncopy = std::min<size_t>(cap, n);
    memcpy(hashes_out, hk.data(), ncopy * 8);
    memcpy(counts_out, hc.data(), ncopy * 8);
    return KCT_OK;
}

kct_status kct_merge_device(kct_table *t, const void *d_hashes, const void *d_counts, size_t n, uint64_t *total_added,
                            uint64_t *new_keys) {
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
    if (!dst || !src) { set_err("null table handle"); return KCT_ERR_ARG; }
    if (dst->k != src->k) { set_err("KmerCountTables must have the same ksize"); return KCT_ERR_KSIZE_MISMATCH; }
    // snapshot src (lib.rs:791-795), then fold it into dst (lib.rs:798-806)
    u64 n = 0;
    KCT_TRY(kct_len(src, &n));
    std::vector<u64> hk(n ? n : 1), hc(n ? n : 1);
    u64 got = 0;
    KCT_TRY(kct_dump(src, hk.data(), hc.data(), n, 0, &got));
    u64 ta = 0, nk = 0;
    KCT_TRY(kct_merge_host(dst, hk.data(), hc.data(), n, &ta, &nk));
    dst->consumed += src->consumed;  // lib.rs:808
    if (total_added) *total_added = ta;
    if (new_keys) *new_keys = nk;
    return KCT_OK;
}

}  // extern "C" (reopened below)

// ---- FASTA / FASTQ ingestion: the caller side of the path (README.md:89-99) ---------------------------
// The reference delegates parsing to screed and calls consume() once per record.  Here a host
// parser turns the file (plain or gzip) into record-stream chunks in pinned memory while a worker
// thread uploads and counts the previous chunk, so parsing and device work overlap.  A record longer
// than a chunk is cut with a (k-1)-base overlap, which keeps every window counted exactly once.
namespace {

struct FileChunk {
    PinnedBuf host;
    DevBuf dev;
    size_t used = 0;
};

struct RecordParser {
    gzFile f = nullptr;
    std::vector<unsigned char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    int fmt = 0;  // '>' FASTA, '@' FASTQ, 0 unknown yet
    bool fill() {
        if (eof) return false;
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) { eof = true; return false; }
        pos = 0; end = (size_t)n;
        return true;
    }
    int peek() { if (pos >= end && !fill()) return -1; return buf[pos]; }
    int get() { int c = peek(); if (c >= 0) ++pos; return c; }
    void skip_line() { int c; while ((c = get()) >= 0 && c != '\n') {} }
};

}  // namespace

extern "C" kct_status kct_consume_file(kct_table *t, const char *path, int skip_bad, uint64_t *n_total, uint64_t *n_records,
                                       uint64_t *n_bases) {
    KCT_TRY(use(t));
    if (!path || !n_total) { set_err("null argument"); return KCT_ERR_ARG; }
    if (!skip_bad) { set_err("kct_consume_file supports skip_bad_kmers=True only; use kct_consume_batch for error mode"); return KCT_ERR_ARG; }
    *n_total = 0;
    if (n_records) *n_records = 0;
    if (n_bases) *n_bases = 0;
    RecordParser ps;
    ps.f = gzopen(path, "rb");
    if (!ps.f) { set_err("cannot open %s", path); return KCT_ERR_ARG; }
    gzbuffer(ps.f, 1 << 20);
    ps.buf.resize(1 << 22);
    const size_t k = t->k;
    size_t chunk_cap = (size_t)64 << 20;  // stream bytes per chunk
    if (const char *e = getenv("KCT_FILE_CHUNK")) chunk_cap = std::max<size_t>(1024, (size_t)atoll(e));  // tests shrink it to exercise record splitting
    FileChunk chunks[2];
    kct_status st = KCT_OK;
    for (auto &c : chunks) {
        if (st == KCT_OK) st = c.host.reserve(chunk_cap + 64);
        if (st == KCT_OK) st = c.dev.reserve(chunk_cap + 64);
    }
    // worker: uploads and counts chunk `job` while the parser fills the other one
    std::mutex mu;
    std::condition_variable cv;
    int job = -1;             // chunk index handed to the worker, -1 = none
    bool done = false, busy = false;
    kct_status worker_status = KCT_OK;
    std::string worker_msg;
    u64 counted = 0;
    std::thread worker([&] {
        (void)hipSetDevice(t->device);
        for (;;) {
            int j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return job >= 0 || done; });
                if (job < 0 && done) return;
                j = job; job = -1; busy = true;
            }
            FileChunk &c = chunks[j];
            kct_status ws = KCT_OK;
            const size_t padded = (c.used + 15) & ~(size_t)15;
            memset((char *)c.host.p + c.used, '\n', padded + 16 - c.used);
            if (hipMemcpyAsync(c.dev.p, c.host.p, padded + 16, hipMemcpyHostToDevice, t->stream) != hipSuccess) { set_err("H2D copy failed"); ws = KCT_ERR_HIP; }
            u64 n = 0;
            if (ws == KCT_OK) ws = consume_stream(t, (const unsigned char *)c.dev.p, c.used, &n);
            {
                std::lock_guard<std::mutex> lk(mu);
                counted += n;
                if (ws != KCT_OK && worker_status == KCT_OK) { worker_status = ws; worker_msg = g_err; }
                busy = false;
            }
            cv.notify_all();
        }
    });
    auto submit = [&](int j) {  // hand chunk j to the worker once it is idle
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job < 0 && !busy; });
        job = j;
        lk.unlock();
        cv.notify_all();
    };
    int cur = 0;
    u64 records = 0, bases = 0;
    unsigned char *out = (unsigned char *)chunks[cur].host.p;
    size_t used = 0, rec_len = 0;  // rec_len = bases of the current record already emitted into this chunk run
    auto flush = [&](bool mid_record) {
        // keep the last k-1 bases of an unfinished record: they open the next chunk
        unsigned char tail[256];
        size_t ntail = 0;
        if (mid_record) { ntail = std::min(rec_len, k - 1); memcpy(tail, out + used - ntail, ntail); }
        chunks[cur].used = used;
        submit(cur);
        cur ^= 1;
        // submit() returned once the worker was idle, i.e. the other buffer's chunk is finished: it is free
        out = (unsigned char *)chunks[cur].host.p;
        memcpy(out, tail, ntail);
        used = ntail;
        rec_len = ntail;
    };
    auto emit = [&](const unsigned char *p, size_t n) {  // append sequence bytes of the current record
        while (n) {
            if (used + 1 >= chunk_cap) flush(true);
            const size_t take = std::min(n, chunk_cap - 1 - used);
            memcpy(out + used, p, take);
            used += take; rec_len += take; p += take; n -= take; bases += take;
        }
    };
    auto end_record = [&] {
        if (used + 1 >= chunk_cap) flush(true);
        out[used++] = '\n';
        rec_len = 0;
        ++records;
    };
    // copies the rest of the current line (without CR/LF) into the record; returns its length
    auto emit_line = [&]() -> size_t {
        size_t total = 0;
        for (;;) {
            if (ps.pos >= ps.end && !ps.fill()) break;
            const unsigned char *b = ps.buf.data() + ps.pos;
            const size_t avail = ps.end - ps.pos;
            const unsigned char *nl = (const unsigned char *)memchr(b, '\n', avail);
            size_t n = nl ? (size_t)(nl - b) : avail;
            size_t m = n;
            if (m && b[m - 1] == '\r') --m;
            emit(b, m); total += m;
            ps.pos += n + (nl ? 1 : 0);
            if (nl) break;
        }
        return total;
    };
    if (st == KCT_OK) {
        int c;
        while ((c = ps.peek()) >= 0) {
            if (c == '\n' || c == '\r' || c == ' ' || c == '\t') { ps.get(); continue; }
            if (ps.fmt == 0) {
                if (c != '>' && c != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, c); st = KCT_ERR_ARG; break; }
                ps.fmt = c;
            }
            if (c != ps.fmt) { set_err("%s: malformed record header near record %llu", path, (unsigned long long)records); st = KCT_ERR_ARG; break; }
            ps.skip_line();  // header
            size_t seq_len = 0;
            if (ps.fmt == '>') {
                while ((c = ps.peek()) >= 0 && c != '>') seq_len += emit_line();
            } else {
                while ((c = ps.peek()) >= 0 && c != '+') seq_len += emit_line();
                ps.skip_line();  // '+' line
                size_t q = 0;    // quality: as many characters as the sequence had
                while (q < seq_len && ps.peek() >= 0) {
                    int d = ps.get();
                    if (d != '\n' && d != '\r') ++q;
                }
                ps.skip_line();
            }
            end_record();
        }
        if (st == KCT_OK && used) { chunks[cur].used = used; submit(cur); }
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job < 0 && !busy; });
        done = true;
    }
    cv.notify_all();
    worker.join();
    gzclose(ps.f);
    for (auto &c : chunks) { c.host.release(); c.dev.release(); }
    if (st == KCT_OK && worker_status != KCT_OK) { st = worker_status; set_err("%s", worker_msg.c_str()); }
    if (st != KCT_OK) return st;
    t->consumed += bases;
    *n_total = counted;
    if (n_records) *n_records = records;
    if (n_bases) *n_bases = bases;
    return KCT_OK;
}

extern "C" {

kct_status kct_set_stream(kct_table *t, void *hip_stream) {
    KCT_TRY(use(t));
    HIP_TRY(hipStreamSynchronize(t->stream));
    prof_collect(t);
    if (t->own_stream) { HIP_TRY(hipStreamDestroy(t->stream)); t->own_stream = false; }
    t->stream = (hipStream_t)hip_stream;
    return KCT_OK;
}

void *kct_get_stream(kct_table *t) { return t ? (void *)t->stream : nullptr; }

kct_status kct_set_path(kct_table *t, int mode) {
    KCT_TRY(use(t));
    if (mode < 0 || mode > 2) { set_err("mode must be 0, 1 or 2"); return KCT_ERR_ARG; }
    t->force_path = mode;
    return KCT_OK;
}

kct_status kct_profile_enable(kct_table *t, int on) { KCT_TRY(use(t)); t->prof_on = on != 0; return KCT_OK; }

kct_status kct_profile_reset(kct_table *t) {
    KCT_TRY(use(t));
    prof_collect(t);
    t->prof.clear();
    return KCT_OK;
}

kct_status kct_profile_read(kct_table *t, int index, char *name_out, size_t name_cap, uint64_t *launches, double *total_ms) {
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
