// kct_consume.hip -- bulk ingest behind the C ABI: the launch policy of the direct and the partitioned
// (one- and two-level) counting paths, staging of host input, and the entry points that hash windows.
#include "kct_internal.h"

#include <type_traits>
#include "partition_kernels.h"

namespace kcth {

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

// K1 is instruction-issue bound, and a compile-time k is worth ~15 % there (constant shifts and masks, the
// MurmurHash3 block/tail structure unrolled, only the needed table look-ups): every k up to 64 gets its own
// instantiation.  (The direct kernels are bound by the atomic rate, so they keep the three popular k only.)
template <int K>
struct PartitionByK {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        if (k == K) PartitionLauncher<(K <= 32 ? 1 : 2), K>::run(s, grid, stream, nbytes, k, ntiles, a);
        else PartitionByK<K - 1>::run(k, s, grid, stream, nbytes, ntiles, a);
    }
};
// the dedupe-first variant (k <= 32): emits mix64(packed canonical k-mer + 1) instead of hashes
template <int K>
struct PartitionRawByK {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        if (k == K) hipLaunchKernelGGL((kct::partition_windows_kernel<1, K, true>), dim3(grid), dim3(kct::kPartThreads), 0, s, stream, nbytes, k, ntiles, a);
        else PartitionRawByK<K - 1>::run(k, s, grid, stream, nbytes, ntiles, a);
    }
};
template <>
struct PartitionRawByK<0> {
    static void run(int, hipStream_t, int, const unsigned char *, u64, u64, const kct::PartitionArgs &) {}
};
// the compact variant (k <= 21): mix42 values, 32-bit entries
template <int K>
struct PartitionCompactByK {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        if (k == K) hipLaunchKernelGGL((kct::partition_windows_kernel<1, K, 2>), dim3(grid), dim3(kct::kPartThreads), 0, s, stream, nbytes, k, ntiles, a);
        else PartitionCompactByK<K - 1>::run(k, s, grid, stream, nbytes, ntiles, a);
    }
};
template <>
struct PartitionCompactByK<0> {
    static void run(int, hipStream_t, int, const unsigned char *, u64, u64, const kct::PartitionArgs &) {}
};
template <>
struct PartitionByK<0> {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        PartitionLauncher<0, 0>::run(s, grid, stream, nbytes, k, ntiles, a);  // k > 64: the bytewise path
    }
};

// The partitioned path pays 16 B (one level) or 32 B (two levels) of streaming scratch traffic per
// k-mer plus 32 B per table slot per pass; the direct path pays one memory-side atomic per k-mer.
// It wins once a pass brings a fair fraction as many windows as the table has slots.
bool partition_geometry_ok(const kct_table *t) {
    const u64 nblocks = t->cap >> t->block_bits;
    return nblocks >= 16 && t->cap <= (1ULL << 33);  // two levels of 1024 bins x 8192 slots: 128 GiB of table
}

bool partition_pays(const kct_table *t, u64 npos) {
    const u64 nblocks = t->cap >> t->block_bits;
    if (!partition_geometry_ok(t) || npos < (1ULL << 20)) return false;
    return nblocks <= 1024 ? npos >= t->cap / 4 : npos >= t->cap / 2;
}

unsigned int region_capacity(double avg) {
    return (unsigned int)((((u64)(avg * 1.15 + 8.0 * __builtin_sqrt(avg) + 64.0)) + 7) & ~7ULL);
}

// Dedupe-first pass (k <= 32).  Reads that cover a small genome deeply repeat every k-mer tens of times per pass, and
// ~55 % of K1's instructions are MurmurHash3 plus the ASCII re-expansion.  So the pass counts PACKED k-mers: K1 (RAW)
// partitions mix64(packed canonical k-mer + 1) values, and the unchanged K2 counts them into a SHADOW table -- 1024
// blocks x 8192 slots in HBM with the real table's layout, keyed by those values -- holding counts that are PENDING:
// the real (hash-keyed) table only gets them when something needs it (flush_shadow: every k-mer with a pending
// count is hashed once and added with the direct insert).  Many passes, one conversion; MurmurHash3 and the table's
// random accesses are paid per distinct k-mer per flush instead of per occurrence.  Reads of the table flush first
// (use()), so nothing observes the difference.  Chosen by dedupe_pays(); counts are identical either way.
kct::TableGeom shadow_geom(const kct_table *t) {
    kct::TableGeom g;
    g.mask = t->shadow_cap - 1;
    g.block_bits = t->shadow_block_bits;
    return g;
}

bool dedupe_pays(const kct_table *t, u64 npos) {
    if (t->k > 32 || t->dedupe_off || npos < (1ULL << 22) || !partition_geometry_ok(t)) return false;
    if (t->force_path == 3) return true;
    if (t->force_path != 0 || !partition_pays(t, npos)) return false;  // the shadow mirrors the table's geometry
    // few distinct k-mers, each many times?  What the table (or the shadow) holds so far is the best guess.
    const u64 known = std::max({t->n_keys, t->shadow_keys, t->s32_keys});
    if (known == 0) return t->dedupe_hint;  // nothing counted yet (new or cleared table): go by how the last pass went
    // A flush costs ~0.11 ns per pending k-mer (one random table access each), a dedupe-first pass saves ~3-4 ps per
    // window (no MurmurHash3 in K1): converting pays once ~32 windows have been counted per distinct k-mer since the
    // table was last read.  The caller's run so far is the evidence that reads are that rare.
    // (a table of up to 1024 blocks is flushed by partitioning the pairs: half the cost per k-mer)
    const u64 per_key = (t->cap >> t->block_bits) <= 1024 ? 16 : 32;
    return known * per_key <= t->windows_since_read + npos;
}

// The shadow mirrors the real table's capacity (the same k-mers live in both).  (Re)allocated empty when that changes.
kct_status ensure_shadow(kct_table *t, bool *ok) {
    *ok = true;
    if (t->shadow && t->shadow_cap == t->cap) return KCT_OK;
    KCT_TRY(flush_shadow(t));
    if (t->shadow) { (void)hipFree(t->shadow); t->shadow = nullptr; t->shadow_cap = 0; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < 3.0 * (double)t->cap * 16.0) { *ok = false; return KCT_OK; }  // not with HBM this tight
    if (hipMalloc((void **)&t->shadow, t->cap * 16) != hipSuccess) { (void)hipGetLastError(); t->shadow = nullptr; *ok = false; return KCT_OK; }
    t->shadow_cap = t->cap;
    t->shadow_block_bits = t->block_bits;
    t->shadow_empty = true;
    t->shadow_keys = 0;
    return KCT_OK;
}

// Pending counts -> the real table.  The shadow keeps its keys (they will be met again), its counts return to zero.
constexpr int kCompactBlockBits = 10;  // the compact shadow: 1024 blocks x 8192 slots
constexpr u64 kCompactSlots = 1ULL << (kCompactBlockBits + kct::kBlockBitsMax);

// Tables of 16..1024 blocks: hash the pending k-mers, radix-partition the {hash, count} pairs by table block and merge
// each block in LDS -- the table is read and written once, sequentially, instead of once per k-mer at random.
// src 0 / 1: the compact / 64-bit shadow's pending counts; src 2: a flat list of n {hash, count} pairs (keys[i * stride],
// counts[i * stride]) -- add(), load(), the multi-GPU merge.  tallies (may be null) += CTR_* of the pass.
kct_status partitioned_pairs_pass(kct_table *t, int src, const du64 *keys, const du64 *counts, u64 n, int stride, u64 *tallies) {
    const bool compact = src == 0;
    const int pbits = log2_u64(t->cap >> t->block_bits);
    const u64 P = 1ULL << pbits;
    const u64 sslots = src == 2 ? n : compact ? kCompactSlots : t->shadow_cap, skeys = src == 2 ? n : compact ? t->s32_keys : t->shadow_keys;
    const int nwg = src == 2 ? (int)std::min<u64>(t->num_cus, (n + 8 * kct::kPartThreads - 1) / (8 * kct::kPartThreads))
                             : (int)std::min<u64>(t->num_cus, sslots >> kct::kBlockBitsMax);  // a workgroup takes whole shadow blocks
    const double fill = std::min(1.0, (double)skeys / (double)sslots);
    const unsigned int region_cap = region_capacity((double)(sslots / nwg + kct::kPartThreads) * fill / (double)P);  // pairs per (workgroup, block)
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 16));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_pairs_ovf.reserve(sslots * 16));  // pairs that found ring or region full (its own buffer: d_aux may hold the input)
    KCT_TRY(t->d_spill.reserve(sslots * 16));  // pairs that found their table block full
    KCT_TRY(zero_counters(t));
    du64 *d_ovf_n = t->d_counters + kNumCounters + 5;
    const bool fresh = t->lazy_empty;
    kct::FlushPartitionArgs fa;
    fa.shadow = compact ? (void *)t->shadow32 : (void *)t->shadow; fa.shadow_blocks = src == 2 ? 0u : (unsigned int)(sslots >> kct::kBlockBitsMax); fa.k = t->k;
    fa.pair_keys = keys; fa.pair_counts = counts; fa.pair_stride = stride; fa.npairs = n; fa.table_block_bits = t->block_bits; fa.pbits = pbits;
    fa.scratch = (ulonglong2 *)t->d_scratch.p; fa.region_cap = region_cap; fa.region_count = (unsigned int *)t->d_regions.p;
    fa.ovf = (du64 *)t->d_pairs_ovf.p; fa.ovf_cap = sslots; fa.ovf_n = d_ovf_n;
    {
        ProfScope ps(t, "flush_partition_kernel");
        if (src == 0) hipLaunchKernelGGL(kct::flush_partition_kernel<0>, dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else if (src == 1) hipLaunchKernelGGL(kct::flush_partition_kernel<1>, dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else hipLaunchKernelGGL(kct::flush_partition_kernel<2>, dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
    }
    HIP_TRY(hipGetLastError());
    kct::AggregatePairsArgs pa;
    pa.words = t->slots; pa.block_bits = t->block_bits;
    pa.scratch = (const ulonglong2 *)t->d_scratch.p; pa.seg_stride = P * region_cap; pa.block_stride = region_cap;
    pa.region_count = (const unsigned int *)t->d_regions.p; pa.nregions = nwg;
    pa.fresh = fresh ? 1 : 0;
    pa.spill = (du64 *)t->d_spill.p; pa.spill_cap = sslots; pa.spill_n = t->d_counters + kNumCounters; pa.counters = t->d_counters;
    {
        ProfScope ps(t, "aggregate_pairs_kernel");
        hipLaunchKernelGGL(kct::aggregate_pairs_kernel, dim3((unsigned)P), dim3(kct::kPartThreads), 0, t->stream, pa);
    }
    HIP_TRY(hipGetLastError());
    t->lazy_empty = false;
    // the (normally few) pairs that did not fit ring or region: the direct insert; the list length is read on the device
    launch_merge_pairs(t, (const du64 *)t->d_pairs_ovf.p, (const du64 *)t->d_pairs_ovf.p + 1, sslots, (const du64 *)d_ovf_n, 2, view(t, sslots));
    HIP_TRY(hipGetLastError());
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (tallies) for (int i = 0; i < 4; ++i) tallies[i] += c[i];
    if (spilled) {  // pairs that found their table block full: grow, then the direct insert (its tallies join ours)
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(grow_to(t, t->cap * 2));
        u64 tl[4] = {0, 0, 0, 0};
        KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux2.p, (const du64 *)t->d_aux2.p + 1, spilled, 2, tl));
        if (tallies) for (int i = 0; i < 4; ++i) tallies[i] += tl[i];
    }
    return KCT_OK;
}

kct_status flush_partitioned(kct_table *t, bool compact) { return partitioned_pairs_pass(t, compact ? 0 : 1, nullptr, nullptr, 0, 0, nullptr); }

// merge_pairs' fast route: tables of 16..1024 blocks, enough pairs to be worth two kernels
bool pairs_partition_pays(const kct_table *t, u64 n) {
    const u64 blocks = t->cap >> t->block_bits;
    return blocks >= 16 && blocks <= 1024 && t->block_bits == kct::kBlockBitsMax && n >= (1ULL << 18) && !getenv("KCT_FLUSH_ATOMIC");
}

kct_status merge_pairs_partitioned(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]) {
    return partitioned_pairs_pass(t, 2, d_keys, d_counts, n, stride, tallies);
}

kct_status flush_compact(kct_table *t) {
    if (!t->s32_dirty) return KCT_OK;
    t->s32_dirty = false;
    t->s32_windows = 0;
    {
        const u64 blocks = t->cap >> t->block_bits;
        if (blocks >= 16 && blocks <= 1024 && t->block_bits == kct::kBlockBitsMax && !getenv("KCT_FLUSH_ATOMIC")) return flush_partitioned(t, true);
    }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(kCompactSlots * 16));
    KCT_TRY(zero_counters(t));
    {
        ProfScope ps(t, "shadow32_flush_kernel");
        hipLaunchKernelGGL(kct::shadow32_flush_kernel, dim3(merge_grid(kCompactSlots)), dim3(kct::kBlock), 0, t->stream, t->shadow32,
                           (int)kct::kBlockBitsMax, kCompactSlots, view(t, kCompactSlots), (int)t->k, t->d_counters);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (spilled) {
        u64 ignored = 0;
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, &ignored));
    }
    return KCT_OK;
}

kct_status flush_shadow(kct_table *t) {
    KCT_TRY(flush_compact(t));
    if (!t->shadow_dirty) return KCT_OK;
    t->shadow_dirty = false;
    {
        const u64 blocks = t->cap >> t->block_bits, sblocks = t->shadow_cap >> kct::kBlockBitsMax;
        if (blocks >= 16 && blocks <= 1024 && t->block_bits == kct::kBlockBitsMax && t->shadow_block_bits == kct::kBlockBitsMax && sblocks >= 1 &&
            !getenv("KCT_FLUSH_ATOMIC"))
            return flush_partitioned(t, false);
    }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(t->shadow_cap * 16));
    KCT_TRY(zero_counters(t));
    {
        ProfScope ps(t, "shadow_flush_kernel");
        hipLaunchKernelGGL(kct::shadow_flush_kernel, dim3(merge_grid(t->shadow_cap)), dim3(kct::kBlock), 0, t->stream, t->shadow, shadow_geom(t),
                           view(t, t->shadow_cap), (int)t->k, t->d_counters);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (spilled) {
        u64 ignored = 0;
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, &ignored));
    }
    return KCT_OK;
}

// Compact dedupe-first pass (k <= 21, one level): K1 MODE 2 writes 32-bit entries, aggregate_blocks32_kernel counts them
// into the compact shadow (u32 keys, u32 counts).  Half the partition traffic of the 64-bit variant and half as many
// ring flushes.  Same contract as consume_partitioned(raw = true).
bool compact_pays(const kct_table *t, u64 npos) {
    if (t->k > 21 || t->compact_off || !dedupe_pays(t, npos)) return false;
    const u64 known = std::max(t->n_keys, t->s32_keys);
    return known <= (u64)(kCompactSlots * 0.6);
}

kct_status consume_compact(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled) {
    *handled = false;
    const int k = t->k;
    const int pbits = kCompactBlockBits;
    const u64 P = 1ULL << pbits;
    const int nwg = t->num_cus;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile;
    const u64 tiles_per_wg = (ntiles + nwg - 1) / nwg;
    const unsigned int region_cap = (region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)P) + 15u) & ~15u;  // 16-entry lines
    const unsigned int ovf_cap = (unsigned int)std::max<u64>(4096, tiles_per_wg * kct::kPartTile / 8);
    if (t->s32_windows + npos >= (1ULL << 31)) KCT_TRY(flush_compact(t));  // u32 counts: no k-mer can have been seen 2^32 times
    if (!t->shadow32) {
        if (hipMalloc((void **)&t->shadow32, kCompactSlots * 8) != hipSuccess) {  // no room: this table does without
            (void)hipGetLastError();
            t->shadow32 = nullptr;
            t->compact_off = true;
            return KCT_OK;
        }
        t->s32_empty = true;
        t->s32_keys = 0;
    }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 4));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve((u64)nwg * ovf_cap * 8 + (u64)nwg * 4));
    KCT_TRY(t->d_spill.reserve(2 * npos * 16));  // two lists: pairs that found their shadow block full, then the merges' own spills
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + (u64)nwg * ovf_cap);

    kct::PartitionArgs pa;
    pa.mask = kCompactSlots - 1; pa.block_bits = kct::kBlockBitsMax; pa.pbits = pbits;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = t->ablate;
    {
        ProfScope ps(t, "partition_windows_kernel<compact>");
        PartitionCompactByK<21>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa);
    }
    HIP_TRY(hipGetLastError());
    kct::Aggregate32Args aa;
    aa.words = t->shadow32; aa.block_bits = kct::kBlockBitsMax;
    aa.scratch = (const unsigned int *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
    aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    aa.fresh = t->s32_empty ? 1 : 0; aa.overflow = d_overflow; aa.ablate = t->ablate;
    aa.spill = (du64 *)t->d_spill.p; aa.spill_cap = npos; aa.spill_n = t->d_counters + kNumCounters; aa.counters = t->d_counters;
    {
        ProfScope ps(t, "aggregate_blocks32_kernel");
        hipLaunchKernelGGL(kct::aggregate_blocks32_kernel, dim3((unsigned)P), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    // What does not fit the shadow goes to the real table in the same submission (no host round trip in between): K1's
    // overflow regions, and the pairs that found their shadow block full (normally none; their number is read on the
    // device).  These inserts spill -- if the table lacks room -- into the SECOND half of the spill buffer.
    kct::TableView mv = view(t, npos);
    mv.spill = (du64 *)t->d_spill.p + 2 * npos;
    mv.spill_n = t->d_counters + kNumCounters + 5;
    {
        ProfScope ps(t, "merge_overflow_kernel");
        hipLaunchKernelGGL(kct::merge_overflow_kernel<2>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                           (const unsigned int *)d_ovf_count, nwg, ovf_cap, (const du64 *)d_overflow, mv, t->d_counters, k);
        hipLaunchKernelGGL(kct::merge_mixed_pairs_kernel<2>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_spill.p,
                           (const du64 *)(t->d_counters + kNumCounters), (u64)npos, mv, (int)k, t->d_counters);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], blocked;
    KCT_TRY(read_counters(t, c, &blocked));
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // K1 gave up: every kernel after it exited early, nothing was touched
    *handled = true;
    const u64 counted = c[kct::CTR_COUNTED], new_keys = c[kct::CTR_NEW_BY_ZERO], spilled2 = t->h_counters[kNumCounters + 5];
    const u64 *c2 = c;  // the merges' tallies: CTR_TOTAL_ADDED / CTR_NEWKEYS
    t->s32_empty = false;
    t->s32_dirty = true;
    t->s32_keys += new_keys;
    t->s32_windows += npos;
    if (t->debug)
        fprintf(stderr, "[kct] compact dedupe pass: npos=%llu region_cap=%u counted=%llu new keys=%llu (total %llu) blocked=%llu merged=%llu spilled=%llu\n",
                (unsigned long long)npos, region_cap, (unsigned long long)counted, (unsigned long long)new_keys, (unsigned long long)t->s32_keys,
                (unsigned long long)blocked, (unsigned long long)c2[kct::CTR_TOTAL_ADDED], (unsigned long long)spilled2);
    *n_out += counted + c2[kct::CTR_TOTAL_ADDED];  // (see consume_partitioned about n and a MurmurHash3 value of 0)
    t->n_keys += c2[kct::CTR_NEWKEYS];
    if (spilled2) {
        KCT_TRY(t->d_aux2.reserve(spilled2 * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, (du64 *)t->d_spill.p + 2 * npos, spilled2 * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled2, n_out));
    }
    if (new_keys * 3 > npos) {  // too few repeats for any dedupe-first variant
        KCT_TRY(flush_compact(t));
        if (t->force_path != 3) { t->dedupe_off = true; t->dedupe_hint = false; }
    } else {
        t->dedupe_hint = true;
        if (blocked * 50 > npos || t->s32_keys > (u64)(kCompactSlots * 0.65)) {  // outgrown: the table-sized 64-bit shadow takes over
            KCT_TRY(flush_compact(t));
            t->compact_off = true;
        }
    }
    return KCT_OK;
}

// One pass of the partitioned path over window starts [0, npos) of d_stream.  *handled = false
// (and nothing counted) if the pass had to be abandoned; the caller then uses the direct path.
// raw = false: K1 hashes, K2 counts into the real table.
// raw = true (dedupe-first, k <= 32): K1 emits mix64(packed k-mer) values, K2 counts them into the shadow table
//        (same geometry); what does not fit (overflow regions, pairs that found their block full) is hashed and goes
//        to the real table at once.
kct_status consume_partitioned(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled, bool raw) {
    *handled = false;
    const int k = t->k;
    if (raw) {
        bool ok = true;
        KCT_TRY(ensure_shadow(t, &ok));
        if (!ok) { t->dedupe_off = true; return KCT_OK; }
        KCT_TRY(materialize(t));  // what does not fit the shadow goes straight to the real table
    }
    du64 *words = raw ? t->shadow : t->slots;
    const int bbits = log2_u64(t->cap >> t->block_bits);      // log2(table blocks)
    const bool two_level = bbits > 10;
    // K1 fans out to 2^pbits bins, each holding 2^sub_bits table blocks that K1b separates.  K1b wants >= 64 bins per
    // super-bin: with 16 its lanes fight over a handful of LDS cursors (2.5x slower per k-mer), so small tables give K1
    // FEWER bins, and W workgroups share a super-bin so that K1b still fills the chip.
    int pbits = bbits;
    if (two_level) pbits = bbits <= 14 ? bbits - 6 : std::min(10, bbits - 7);
    if (const char *e = getenv("KCT_PBITS")) if (two_level) pbits = std::max(bbits - 10, std::min(10, atoi(e)));  // measurement only
    const int sub_bits = bbits - pbits;                       // ... each holding 2^sub_bits table blocks
    const u64 P = 1ULL << pbits, B = 1ULL << bbits;
    const int nwg = t->num_cus;
    const u64 W = two_level ? std::max<u64>(1, (u64)nwg / P) : 1;  // K1b workgroups per super-bin
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
    const bool fresh = raw ? t->shadow_empty : t->lazy_empty;

    kct::PartitionArgs pa;
    pa.mask = t->cap - 1; pa.block_bits = t->block_bits + sub_bits; pa.pbits = pbits;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = t->ablate;  // measurement only; wrong counts when set
    {
        ProfScope ps(t, raw ? "partition_windows_kernel<raw>" : "partition_windows_kernel");
        if (raw) PartitionRawByK<32>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa);
        else PartitionByK<64>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa);
    }
    HIP_TRY(hipGetLastError());

    kct::AggregateArgs aa;
    aa.words = words; aa.block_bits = t->block_bits; aa.pbits = bbits;
    aa.fresh = fresh ? 1 : 0; aa.overflow = d_overflow; aa.ablate = pa.ablate;
    aa.spill = (du64 *)t->d_spill.p; aa.spill_cap = npos; aa.spill_n = t->d_counters + kNumCounters; aa.counters = t->d_counters;
    unsigned int ovf2_cap = 0, *d_ovf2_count = nullptr;
    if (!two_level) {
        aa.scratch = (const du64 *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
        aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    } else {
        // second level: one workgroup per super-bin spreads its hashes over the super-bin's blocks
        const unsigned int out_cap = region_capacity((double)npos / (double)B / (double)W);
        ovf2_cap = (unsigned int)std::max<u64>(4096, npos / P / W / 8);
        KCT_TRY(t->d_scratch2.reserve(B * W * out_cap * 8));
        KCT_TRY(t->d_regions2.reserve(B * W * 4));
        KCT_TRY(t->d_irr2.reserve(P * W * ovf2_cap * 8 + P * W * 4));
        d_ovf2_count = (unsigned int *)((du64 *)t->d_irr2.p + P * W * ovf2_cap);
        kct::RepartitionArgs ra;
        ra.mask = t->cap - 1; ra.block_bits = t->block_bits; ra.sub_bits = sub_bits;
        ra.in = (const du64 *)t->d_scratch.p; ra.in_cap = region_cap; ra.in_count = (const unsigned int *)t->d_regions.p;
        ra.nseg = nwg; ra.nbins = (int)P; ra.writers = (int)W;
        ra.out = (du64 *)t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
        ra.ovf = (du64 *)t->d_irr2.p; ra.ovf_cap = ovf2_cap; ra.ovf_count = d_ovf2_count; ra.overflow = d_overflow;
        {
            ProfScope ps(t, "repartition_kernel");
            hipLaunchKernelGGL(kct::repartition_kernel, dim3((unsigned)(P * W)), dim3(kct::kPartThreads), 0, t->stream, ra);
        }
        HIP_TRY(hipGetLastError());
        aa.scratch = (const du64 *)t->d_scratch2.p; aa.seg_stride = out_cap; aa.block_stride = W * out_cap;
        aa.region_count = (const unsigned int *)t->d_regions2.p; aa.nregions = (int)W;
    }
    {
        ProfScope ps(t, raw ? "aggregate_blocks_kernel<shadow>" : "aggregate_blocks_kernel");
        hipLaunchKernelGGL(kct::aggregate_blocks_kernel, dim3((unsigned)B), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    auto merge_overflows = [&](auto dedupe_tag, const du64 *abort) {
        // fold the overflow regions with the direct atomic path; the kernel reads the region lengths
        // (and the abandon flag) from device memory, so no host round trip sits between the launches
        constexpr int D = decltype(dedupe_tag)::value ? 1 : 0;
        ProfScope ps(t, "merge_overflow_kernel");
        hipLaunchKernelGGL(kct::merge_overflow_kernel<D>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                           (const unsigned int *)d_ovf_count, nwg, ovf_cap, abort, view(t, npos), t->d_counters, k);
        if (two_level)
            hipLaunchKernelGGL(kct::merge_overflow_kernel<D>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr2.p,
                               (const unsigned int *)d_ovf2_count, (int)(P * W), ovf2_cap, abort, view(t, npos), t->d_counters, k);
    };
    if (!raw) {
        merge_overflows(std::false_type{}, (const du64 *)d_overflow);
        HIP_TRY(hipGetLastError());
    }
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    if (t->debug) {
        std::vector<unsigned int> oc(nwg);
        (void)hipMemcpy(oc.data(), d_ovf_count, nwg * 4, hipMemcpyDeviceToHost);
        u64 tot = 0;
        for (auto v : oc) tot += v;
        fprintf(stderr, "[kct] partitioned pass%s: npos=%llu blocks=%llu levels=%d region_cap=%u overflow(K1)=%llu counted=%llu merged=%llu new=%llu spilled=%llu abandon=%llu\n",
                raw ? " (shadow)" : "", (unsigned long long)npos, (unsigned long long)B, two_level ? 2 : 1, region_cap, (unsigned long long)tot,
                (unsigned long long)c[kct::CTR_COUNTED], (unsigned long long)c[kct::CTR_TOTAL_ADDED], (unsigned long long)c[kct::CTR_NEWKEYS],
                (unsigned long long)spilled, (unsigned long long)t->h_counters[kNumCounters + 6]);
    }
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // abandoned: K2 and the merges exited early, nothing was touched
    *handled = true;
    if (!raw) {
        t->lazy_empty = false;
        *n_out += c[kct::CTR_COUNTED] + c[kct::CTR_TOTAL_ADDED];
        t->n_keys += c[kct::CTR_NEWKEYS];
        if (spilled) {
            KCT_TRY(t->d_aux2.reserve(spilled * 16));
            HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
            KCT_TRY(replay_spill(t, spilled, n_out));
        }
        return KCT_OK;
    }
    // ---- shadow pass: c = entries counted into the shadow / new shadow keys; `spilled` = pairs that found their block full
    const u64 counted = c[kct::CTR_COUNTED], new_shadow = c[kct::CTR_NEWKEYS], blocked = spilled;
    t->shadow_empty = false;
    t->shadow_dirty = true;
    t->shadow_keys += new_shadow;
    if (blocked) {
        KCT_TRY(t->d_aux2.reserve(blocked * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, blocked * 16, hipMemcpyDeviceToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->d_counters + kNumCounters + 5, t->d_counters + kNumCounters, 8, hipMemcpyDeviceToDevice, t->stream));
    }
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 1) * sizeof(u64), t->stream));  // the tallies and the spill cursor, not the copied pair count
    merge_overflows(std::true_type{}, (const du64 *)nullptr);
    if (blocked) {
        ProfScope ps(t, "merge_mixed_pairs_kernel");
        hipLaunchKernelGGL(kct::merge_mixed_pairs_kernel, dim3(merge_grid(blocked)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_aux2.p,
                           (const du64 *)(t->d_counters + kNumCounters + 5), (u64)blocked, view(t, npos), (int)k, t->d_counters);
    }
    HIP_TRY(hipGetLastError());
    u64 c2[4], spilled2;
    KCT_TRY(read_counters(t, c2, &spilled2));
    // n counts every window whose k-mer went into the shadow: the (2^-64 per k-mer) case of a MurmurHash3 value of 0,
    // which the reference leaves out of n, is only seen when the shadow is flushed.
    *n_out += counted + c2[kct::CTR_TOTAL_ADDED];
    t->n_keys += c2[kct::CTR_NEWKEYS];
    if (spilled2) {
        KCT_TRY(t->d_aux2.reserve(spilled2 * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled2 * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled2, n_out));
    }
    // Too few repeats to be worth it, or the shadow is filling up: convert what is pending and go back to hashing every window.
    if (t->force_path != 3 && (new_shadow * 3 > npos || blocked * 50 > npos)) {
        KCT_TRY(flush_shadow(t));
        t->dedupe_off = true;
        t->dedupe_hint = false;
    } else t->dedupe_hint = true;
    // the shadow holds as many keys as the table would: grow both (the table's growth re-creates the shadow, flushed)
    if ((double)t->shadow_keys > kMaxLoad * (double)t->shadow_cap) {
        KCT_TRY(flush_shadow(t));
        KCT_TRY(grow_to(t, t->cap * 2));
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
        if (compact_pays(t, npos)) {
            bool handled = false;
            KCT_TRY(consume_compact(t, d_stream + done, chunk_bytes, npos, n_out, &handled));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (dedupe_pays(t, npos)) {
            bool handled = false;
            KCT_TRY(consume_partitioned(t, d_stream + done, chunk_bytes, npos, n_out, &handled, true));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (partition_geometry_ok(t) && t->force_path != 1 && (t->force_path == 2 || partition_pays(t, npos))) {
            bool handled = false;
            KCT_TRY(consume_partitioned(t, d_stream + done, chunk_bytes, npos, n_out, &handled, false));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
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
        t->windows_since_read += npos;
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

// ---- deferred mode -------------------------------------------------------------------------------------------
constexpr size_t kPendingBytes = (size_t)64 << 20;

// Valid k-windows of one record: the host-side twin of the device's window rule (all k bytes in ACGTacgt).
// Used only for the number deferred consume() returns; the counting itself happens on the device at flush.
u64 host_valid_windows(const unsigned char *s, size_t len, size_t k) {
    static const struct Lut { bool ok[256]; Lut() { for (bool &b : ok) b = false; for (unsigned char c : {'A','C','G','T','a','c','g','t'}) ok[c] = true; } } lut;
    u64 n = 0;
    size_t run = 0;
    for (size_t i = 0; i < len; ++i) {
        run = lut.ok[s[i]] ? run + 1 : 0;
        n += run >= k;
    }
    return n;
}

kct_status flush_pending(kct_table *t) {
    const size_t used = t->pending_used;
    if (!used) return KCT_OK;
    t->pending_used = 0;  // consume_stream -> ... -> use() must not re-enter
    t->pending_records = 0;
    const size_t padded = (used + 15) & ~(size_t)15;
    memset((char *)t->h_pending.p + used, '\n', padded + 16 - used);
    KCT_TRY(t->d_stream.reserve(padded + 16));
    HIP_TRY(hipMemcpyAsync(t->d_stream.p, t->h_pending.p, padded + 16, hipMemcpyHostToDevice, t->stream));
    u64 n = 0;
    return consume_stream(t, (const unsigned char *)t->d_stream.p, used, &n);
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

}  // namespace kcth

using namespace kcth;

extern "C" {

kct_status kct_hash_windows(kct_table *t, const char *seq, size_t len, uint64_t *hashes_out, size_t cap, uint64_t *n_windows,
                            uint64_t *first_bad) {
    KCT_TRY(use_consume(t));
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
    KCT_TRY(use_consume(t));
    if (!kmer || !hash_out) { set_err("null argument"); return KCT_ERR_ARG; }
    if ((uint8_t)len != t->k) { set_err("wrong ksize"); return KCT_ERR_WRONG_KSIZE; }  // lib.rs:66 `len as u8`
    u64 nwin, fb, h = 0;
    KCT_TRY(kct_hash_windows(t, kmer, t->k, &h, 1, &nwin, &fb));  // first window only (lib.rs:78 `.next()`)
    if (fb == 0) { set_err("invalid DNA character in k-mer"); return KCT_ERR_INVALID_DNA; }
    *hash_out = h;
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

kct_status kct_get(kct_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    KCT_TRY(use(t));
    if ((uint8_t)len != t->k) { set_err("kmer size does not match count table ksize"); return KCT_ERR_WRONG_KSIZE; }
    u64 h;
    KCT_TRY(kct_hash_kmer(t, kmer, len, &h));
    return kct_get_hash(t, h, count_out);
}

kct_status kct_consume(kct_table *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out) {
    if (t && t->deferred && skip_bad && len + 64 < kPendingBytes / 2) {
        // deferred mode: buffer the record, answer from the host-side validity scan, count later
        KCT_TRY(use_device(t));
        if ((!seq && len) || !n_out) { set_err("null argument"); return KCT_ERR_ARG; }
        *n_out = 0;
        if (len >= t->k) {
            KCT_TRY(t->h_pending.reserve(kPendingBytes + 64));
            if (t->pending_used + len + 1 > kPendingBytes) KCT_TRY(flush_pending(t));
            char *dst = (char *)t->h_pending.p + t->pending_used;
            memcpy(dst, seq, len);
            dst[len] = '\n';
            t->pending_used += len + 1;
            t->pending_records += 1;
            *n_out = host_valid_windows((const unsigned char *)seq, len, t->k);
        }
        t->consumed += len;
        return KCT_OK;
    }
    KCT_TRY(use_consume(t));
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
    KCT_TRY(use_consume(t));
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
    KCT_TRY(t->d_stream.reserve(padded + 16));
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = stream_len >= (8u << 20) ? std::min<size_t>({(size_t)8, hw ? hw : 1, nrec}) : 1;
    if (nthreads <= 1) {
        pack_range(0, nrec);
        if (!skip_bad) rec_off[nrec] = stream_len;
        memset(dst + stream_len, '\n', padded + 16 - stream_len);
        HIP_TRY(hipMemcpyAsync(t->d_stream.p, dst, padded, hipMemcpyHostToDevice, t->stream));
    } else {
        // The stream is cut into slices of records (by bytes: records may be ragged) that the packers take in
        // order, round-robin; this thread uploads slice s as soon as it is packed, so the H2D copy runs under
        // the packing of the slices behind it instead of after all of it.
        const size_t nslices = std::min<size_t>(nrec, 4 * nthreads);
        std::vector<size_t> cut(nslices + 1);
        for (size_t i = 0; i <= nslices; ++i) {
            const u64 lo = base0 + total * i / nslices;
            cut[i] = i == nslices ? nrec : (size_t)(std::lower_bound(offsets, offsets + nrec, lo) - offsets);
        }
        cut[0] = 0;
        std::vector<std::atomic<int>> ready(nslices);
        for (auto &r : ready) r.store(0, std::memory_order_relaxed);
        std::vector<std::thread> pool;
        for (size_t i = 0; i < nthreads; ++i)
            pool.emplace_back([&, i]() {
                for (size_t sl = i; sl < nslices; sl += nthreads) {
                    if (cut[sl + 1] > cut[sl]) pack_range(cut[sl], cut[sl + 1]);
                    ready[sl].store(1, std::memory_order_release);
                }
            });
        hipError_t copy_err = hipSuccess;
        for (size_t sl = 0; sl < nslices; ++sl) {
            while (!ready[sl].load(std::memory_order_acquire)) std::this_thread::yield();
            const u64 b0 = cut[sl] < nrec ? (offsets[cut[sl]] - base0) + cut[sl] : stream_len;
            u64 b1 = cut[sl + 1] < nrec ? (offsets[cut[sl + 1]] - base0) + cut[sl + 1] : stream_len;
            if (sl + 1 == nslices) {  // the tail slice carries the padding
                if (!skip_bad) rec_off[nrec] = stream_len;
                memset(dst + stream_len, '\n', padded + 16 - stream_len);
                b1 = padded;
            }
            if (b1 > b0 && copy_err == hipSuccess)
                copy_err = hipMemcpyAsync((char *)t->d_stream.p + b0, dst + b0, b1 - b0, hipMemcpyHostToDevice, t->stream);
        }
        for (auto &th : pool) th.join();
        HIP_TRY(copy_err);
    }

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
                // a 16-byte aligned copy of the prefix, in a buffer nothing inside consume_stream touches (its passes
                // reallocate / overwrite d_aux2 and d_spill when they replay spills)
                KCT_TRY(t->d_prefix.reserve(((prefix + 15) & ~(u64)15) + 16));
                HIP_TRY(hipMemcpyAsync(t->d_prefix.p, (const char *)t->d_stream.p + rec_off[r], prefix, hipMemcpyDeviceToDevice, t->stream));
                KCT_TRY(consume_stream(t, (const unsigned char *)t->d_prefix.p, prefix, &n_prefix));
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
    KCT_TRY(use_consume(t));
    if (!n_total || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    KCT_TRY(consume_stream(t, (const unsigned char *)d_stream, nbytes, n_total));
    t->consumed += consumed_bytes;
    return KCT_OK;
}

}  // extern "C"
