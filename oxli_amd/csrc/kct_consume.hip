// kct_consume.hip -- the counting passes over a device-resident record stream: the direct and the partitioned (one- and two-level)
// paths, the dedupe-first variants and their conversions, and the launchers of their kernels (which only this file instantiates).
// Which path counts a pass: path_policy.h.  Host staging, deferred mode and the C-ABI entry points: kct_entry.hip.
#include "kct_internal.h"

#include <type_traits>
#include "partition_kernels.h"
#include "path_policy.h"

namespace kcth {

static_assert(kPolicyBlockBitsMax == kct::kBlockBitsMax && kPolicyPartTile == kct::kPartTile && kPolicyRingEntries == kct::kRingEntries,
              "path_policy.h restates these constants");

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
// the 128-bit dedupe-first variant (33 <= k <= 64): mix128 pairs, 16-byte entries.  Reachable only through kct_set_path(t, 3) since round 4
// (1.0x over hashing on its showcase), so it no longer gets an instantiation per k (32 copies of K1, a quarter of this file's compile time):
// k = 51 at compile time, every other k at run time.
struct PartitionRaw128 {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        if (k == 51) hipLaunchKernelGGL((kct::partition_windows_kernel<2, 51, 3>), dim3(grid), dim3(kct::kPartThreads), 0, s, stream, nbytes, k, ntiles, a);
        else hipLaunchKernelGGL((kct::partition_windows_kernel<2, 0, 3>), dim3(grid), dim3(kct::kPartThreads), 0, s, stream, nbytes, k, ntiles, a);
    }
};
template <>
struct PartitionByK<0> {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::PartitionArgs &a) {
        PartitionLauncher<0, 0>::run(s, grid, stream, nbytes, k, ntiles, a);  // k > 64: the bytewise path
    }
};

// K1 over PACKED base arrays, the popular k (and run-time k for the rest of k <= 32 / <= 64 where the mode allows): the instantiations that
// fetch a tile's words sixteen bytes per lane (k1_kernel.h PACKED).  false: no such instantiation (or misaligned arrays) -- the ordinary
// instantiation reads the arrays with its two narrow loads per lane.
template <int KW, int KC, int MODE>
static void k1_packed(kct_table *t, int nwg, u64 chunk_bytes, int k, u64 ntiles, const kct::PartitionArgs &pa) {
    hipLaunchKernelGGL((kct::partition_windows_kernel<KW, KC, MODE, false, true>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, (const unsigned char *)nullptr, chunk_bytes, k, ntiles, pa);
}
static bool launch_partition_packed(kct_table *t, int mode, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    const int k = t->k, nwg = t->num_cus;
    if ((((uintptr_t)pa.pcodes | (uintptr_t)pa.pvalid) & 15) != 0 || k > 64) return false;
    if (mode == 2 && t->compact_bursty) {   // (flush every 4 windows: k1_kernel.h FE)
        if (k == 21) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 21, 2, false, true, 4>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, (const unsigned char *)nullptr, chunk_bytes, k, ntiles, pa);
        else hipLaunchKernelGGL((kct::partition_windows_kernel<1, 0, 2, false, true, 4>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, (const unsigned char *)nullptr, chunk_bytes, k, ntiles, pa);
    } else if (mode == 2) {
        if (k == 21) k1_packed<1, 21, 2>(t, nwg, chunk_bytes, k, ntiles, pa);
        else k1_packed<1, 0, 2>(t, nwg, chunk_bytes, k, ntiles, pa);
    } else if (mode == 1) {
        if (k == 21) k1_packed<1, 21, 1>(t, nwg, chunk_bytes, k, ntiles, pa);
        else if (k == 31) k1_packed<1, 31, 1>(t, nwg, chunk_bytes, k, ntiles, pa);
        else k1_packed<1, 0, 1>(t, nwg, chunk_bytes, k, ntiles, pa);
    } else if (mode == 0) {
        if (k == 21) k1_packed<1, 21, 0>(t, nwg, chunk_bytes, k, ntiles, pa);
        else if (k == 31) k1_packed<1, 31, 0>(t, nwg, chunk_bytes, k, ntiles, pa);
        else if (k == 51) k1_packed<2, 51, 0>(t, nwg, chunk_bytes, k, ntiles, pa);
        else if (k <= 32) k1_packed<1, 0, 0>(t, nwg, chunk_bytes, k, ntiles, pa);
        else k1_packed<2, 0, 0>(t, nwg, chunk_bytes, k, ntiles, pa);
    } else return false;
    return true;
}

void launch_partition(kct_table *t, int mode, const unsigned char *d_stream, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    const int k = t->k, nwg = t->num_cus;
    if (pa.runs.groups) { launch_partition_runs(t, mode, chunk_bytes, ntiles, pa); return; }  // received super-k-mers (kct_runs.hip)
    ProfScope ps(t, mode == 2 ? "partition_windows_kernel<compact>" : mode == 1 ? "partition_windows_kernel<raw>" : "partition_windows_kernel");
#ifdef KCT_K1_STAMPS
    // measurement build (tools/k1_stamps.sh): every wave's cycles per phase come back after the launch (which is waited for) and one
    // JSON line per launch goes to the file KCT_K1_STAMPS_OUT names
    kct::PartitionArgs pas = pa;
    const size_t nst = (size_t)nwg * (kct::kPartThreads / 64) * kct::kStampSlots;
    du64 *d_st = nullptr;
    if (hipMalloc((void **)&d_st, nst * 8) == hipSuccess) { (void)hipMemsetAsync(d_st, 0, nst * 8, t->stream); pas.stamps = d_st; }
    const kct::PartitionArgs &pa_ = pas;
    struct StampsOut {
        kct_table *t; du64 *d; size_t n; int mode; u64 ntiles;
        ~StampsOut() {
            if (!d) return;
            std::vector<u64> h(n);
            if (hipMemcpyAsync(h.data(), d, n * 8, hipMemcpyDeviceToHost, t->stream) == hipSuccess && hipStreamSynchronize(t->stream) == hipSuccess) {
                static const char *names[kct::kStampSlots] = {"hash", "bar_tile", "stage", "bar1", "list", "bar2", "move", "bar3", "tail"};
                u64 sum[kct::kStampSlots] = {0}, wmin = ~0ULL, wmax = 0, tot = 0;
                for (size_t w = 0; w < n / kct::kStampSlots; ++w) {
                    u64 wt = 0;
                    for (int i = 0; i < kct::kStampSlots; ++i) { sum[i] += h[w * kct::kStampSlots + i]; wt += h[w * kct::kStampSlots + i]; }
                    if (wt) { wmin = std::min(wmin, wt); wmax = std::max(wmax, wt); }
                    tot += wt;
                }
                if (t->tune.k1_flushers > 0 && getenv("KCT_K1_STAMPS_OUT")) {   // the wave-specialised K1 writes raw per-wave figures (k1ws_kernel.h)
                    if (FILE *f = fopen(getenv("KCT_K1_STAMPS_OUT"), "a")) {
                        const int F = t->tune.k1_flushers, W = kct::kPartThreads / 64;
                        double hs[2] = {0, 0}, fs[7] = {0, 0, 0, 0, 0, 0, 0};
                        for (size_t w = 0; w < n / kct::kStampSlots; ++w) {
                            const u64 *o = &h[w * kct::kStampSlots];
                            if ((int)(w % W) < F) for (int i = 0; i < 7; ++i) fs[i] += (double)o[i];
                            else { hs[0] += (double)o[0]; hs[1] += (double)o[1]; }
                        }
                        const double nf = (double)(n / kct::kStampSlots / W * F), nh = (double)(n / kct::kStampSlots / W * (W - F));
                        fprintf(f, "{\"ws\": 1, \"mode\": %d, \"k\": %d, \"tiles\": %llu, \"hash_wave_cycles\": %.0f, \"pieces_per_hash_wave\": %.1f, \"flusher_cycles\": %.0f, \"sweeps\": %.1f, "
                                   "\"lines_per_sweep\": %.2f, \"grace_share\": %.3f, \"list_cycles_per_sweep\": %.0f, \"move_cycles_per_sweep\": %.0f}\n", mode, (int)t->k, (unsigned long long)ntiles,
                                hs[0] / nh, hs[1] / nh, fs[0] / nf, fs[1] / nf, fs[1] ? fs[2] / fs[1] : 0.0, fs[1] ? fs[3] / fs[1] : 0.0, fs[1] ? fs[4] / fs[1] : 0.0, fs[1] ? fs[5] / fs[1] : 0.0);
                        fclose(f);
                    }
                } else if (const char *path = getenv("KCT_K1_STAMPS_OUT"))
                    if (FILE *f = fopen(path, "a")) {
                        fprintf(f, "{\"mode\": %d, \"k\": %d, \"tiles\": %llu, \"waves\": %zu, \"wave_cycles_min\": %llu, \"wave_cycles_max\": %llu, \"share\": {", mode, (int)t->k,
                                (unsigned long long)ntiles, n / kct::kStampSlots, (unsigned long long)wmin, (unsigned long long)wmax);
                        for (int i = 0; i < kct::kStampSlots; ++i) fprintf(f, "%s\"%s\": %.4f", i ? ", " : "", names[i], tot ? (double)sum[i] / (double)tot : 0.0);
                        fprintf(f, "}}\n");
                        fclose(f);
                    }
            }
            (void)hipFree(d);
        }
    } stamps_out{t, d_st, nst, mode, ntiles};
#else
    const kct::PartitionArgs &pa_ = pa;
#endif
    if (mode != 3 && launch_partition_ws(t, mode, d_stream, chunk_bytes, ntiles, pa_)) return;   // the wave-specialised K1 (kct_k1ws.hip)
    if (mode != 2 && t->wide_bursty && k <= 64) {
        // bursty input on the 8-byte-entry modes: a flush every 2 windows instead of 4 (k1_kernel.h FE; the popular k at compile time, the rest at run
        // time; packed arrays through the ordinary instantiation's narrow loads)
        const dim3 g(nwg), b(kct::kPartThreads);
        if (mode == 1) {
            if (k == 31) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 31, 1, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
            else hipLaunchKernelGGL((kct::partition_windows_kernel<1, 0, 1, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        } else if (k == 21) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 21, 0, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        else if (k == 31) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 31, 0, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        else if (k == 51) hipLaunchKernelGGL((kct::partition_windows_kernel<2, 51, 0, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        else if (k <= 32) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 0, 0, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        else hipLaunchKernelGGL((kct::partition_windows_kernel<2, 0, 0, false, false, 2>), g, b, 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        return;
    }
    if (pa_.pcodes && launch_partition_packed(t, mode, chunk_bytes, ntiles, pa_)) return;
    if (mode == 2 && t->compact_bursty) {   // (flush every 4 windows: k1_kernel.h FE; k = 21 at compile time, every other k at run time)
        if (k == 21) hipLaunchKernelGGL((kct::partition_windows_kernel<1, 21, 2, false, false, 4>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
        else hipLaunchKernelGGL((kct::partition_windows_kernel<1, 0, 2, false, false, 4>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, d_stream, chunk_bytes, k, ntiles, pa_);
    } else if (mode == 2) PartitionCompactByK<21>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa_);
    else if (mode == 1) PartitionRawByK<32>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa_);
    else PartitionByK<64>::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa_);
}

void launch_repartition(kct_table *t, int mode, unsigned grid, const kct::RepartitionArgs &ra, bool whole_slab) {
    ProfScope ps(t, mode == 2 ? "repartition_kernel<compact>" : "repartition_kernel");
    if (mode == 2) hipLaunchKernelGGL((kct::repartition_kernel<unsigned int, false>), dim3(grid), dim3(kct::kPartThreads), 0, t->stream, ra);
    else if (whole_slab) hipLaunchKernelGGL((kct::repartition_kernel<du64, true>), dim3(grid), dim3(kct::kPartThreads), 0, t->stream, ra);
    else hipLaunchKernelGGL((kct::repartition_kernel<du64, false>), dim3(grid), dim3(kct::kPartThreads), 0, t->stream, ra);
}

void launch_aggregate32(kct_table *t, unsigned grid, const kct::Aggregate32Args &aa) {
    ProfScope ps(t, "aggregate_blocks32_kernel");
    hipLaunchKernelGGL(kct::aggregate_blocks32_kernel<true>, dim3(grid), dim3(kct::kPartThreads), 0, t->stream, aa);
}

void launch_aggregate64(kct_table *t, unsigned grid, const kct::AggregateArgs &aa, bool shadow) {
    ProfScope ps(t, shadow ? "aggregate_blocks_kernel<shadow>" : "aggregate_blocks_kernel");
    hipLaunchKernelGGL((kct::aggregate_blocks_kernel<false, true>), dim3(grid), dim3(kct::kPartThreads), 0, t->stream, aa);
}

void launch_merge_overflow(kct_table *t, int mode, const du64 *regions, const unsigned int *counts, int nregions, unsigned int region_cap,
                           const du64 *abort, const kct::TableView &tv, const du64 *total) {
    ProfScope ps(t, "merge_overflow_kernel");
    const kct::PendingList none;
    if (mode == 2) hipLaunchKernelGGL(kct::merge_overflow_kernel<2>, dim3(256), dim3(kct::kBlock), 0, t->stream, regions, counts, nregions, region_cap, abort, tv, t->d_counters, (int)t->k, total, none);
    else if (mode == 1) hipLaunchKernelGGL(kct::merge_overflow_kernel<1>, dim3(256), dim3(kct::kBlock), 0, t->stream, regions, counts, nregions, region_cap, abort, tv, t->d_counters, (int)t->k, total, none);
    else hipLaunchKernelGGL(kct::merge_overflow_kernel<0>, dim3(256), dim3(kct::kBlock), 0, t->stream, regions, counts, nregions, region_cap, abort, tv, t->d_counters, (int)t->k, total, none);
}


// packed input: the K1 launch reads groups instead of bytes (PartitionArgs::pcodes); super-k-mer input: it walks windows
// (PartitionArgs::runs; d_stream then counts windows from runs_base, 64 per group)
static void packed_args(const kct_table *t, const unsigned char *d_stream, kct::PartitionArgs *pa) {
    if (t->runs_in.groups) {
        pa->runs = t->runs_in;
        pa->runs.groups += (u64)(d_stream - t->runs_base) >> 6;
        return;
    }
    if (!t->packed_codes) return;
    const u64 g0 = (u64)(d_stream - t->packed_base) >> 4;
    pa->pcodes = t->packed_codes + g0;
    pa->pvalid = t->packed_valid + g0;
}


// Dedupe-first pass (k <= 32).  Reads that cover a small genome deeply repeat every k-mer tens of times per pass, and
// ~55 % of K1's instructions are MurmurHash3 plus the ASCII re-expansion.  So the pass counts PACKED k-mers: K1 (RAW)
// partitions mix64(packed canonical k-mer + 1) values, and the unchanged K2 counts them into a SHADOW table -- in HBM, with
// the real table's layout, keyed by those values -- holding counts that are PENDING:
// the real (hash-keyed) table only gets them when something needs it (flush_shadow: every k-mer with a pending
// count is hashed once and added).  Many passes, one conversion; MurmurHash3 and the table's
// random accesses are paid per distinct k-mer per flush instead of per occurrence.  Reads of the table flush first
// (use()), so nothing observes the difference.  Chosen by dedupe_pays(); counts are identical either way.
kct::TableGeom shadow_geom(const kct_table *t) {
    kct::TableGeom g;
    g.mask = t->shadow_cap - 1;
    g.block_bits = t->shadow_block_bits;
    return g;
}

kct_status flush_compact(kct_table *t);

// The shadow mirrors the real table's capacity (the same k-mers live in both) -- except for the dedupe probe's, which is
// small.  (Re)allocated empty, after converting what is pending, when the wanted capacity changes.
kct_status ensure_shadow(kct_table *t, u64 want_cap, bool *ok) {
    *ok = true;
    if (t->shadow && t->shadow_cap == want_cap) return KCT_OK;
    KCT_TRY(flush_shadow(t));
    if (t->shadow) { (void)hipFree(t->shadow); t->shadow = nullptr; t->shadow_cap = 0; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < 3.0 * (double)want_cap * 16.0) { *ok = false; return KCT_OK; }  // not with HBM this tight
    if (hipMalloc((void **)&t->shadow, want_cap * 16) != hipSuccess) { (void)hipGetLastError(); t->shadow = nullptr; *ok = false; return KCT_OK; }
    t->shadow_cap = want_cap;
    t->shadow_block_bits = std::min(kct::kBlockBitsMax, log2_u64(want_cap));
    t->shadow_empty = true;
    t->shadow_keys = 0;
    return KCT_OK;
}

u64 compact_slots(const kct_table *t) { return ((u64)t->s32_nbins << (t->s32_sbits - kCompactBlockBits)) << kct::kBlockBitsMax; }

kct_status ensure_shadow32(kct_table *t, int want, bool *ok, unsigned int nbins, unsigned int bin0) {
    *ok = true;
    if (t->shadow32 && t->s32_sbits == want && t->s32_nbins == nbins && t->s32_bin0 == bin0) return KCT_OK;
    KCT_TRY(flush_compact(t));
    if (t->shadow32) { (void)hipFree(t->shadow32); t->shadow32 = nullptr; }
    t->s32_nbins = 1024; t->s32_bin0 = 0; t->s32_sbits = kCompactBlockBits;
    const u64 bytes = (((u64)nbins << (want - kCompactBlockBits)) << kct::kBlockBitsMax) * 8;
    size_t free_b = 0, total_b = 0;
    if (want > kCompactBlockBits && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < 3.0 * (double)bytes)) { *ok = false; return KCT_OK; }
    if (hipMalloc((void **)&t->shadow32, bytes) != hipSuccess) { (void)hipGetLastError(); t->shadow32 = nullptr; *ok = false; return KCT_OK; }  // no room: this table does without
    t->s32_sbits = want; t->s32_nbins = nbins; t->s32_bin0 = bin0;
    t->s32_empty = true;
    t->s32_keys = 0;
    return KCT_OK;
}

// {hash, count} pairs into the table by the k-mers' own route: hash the pending k-mers (or take a flat pair list),
// radix-partition the PAIRS by table block through the LDS ring (one level for up to 1024 blocks, two beyond) and merge
// each block in LDS -- the table is read and written once, sequentially, instead of once per k-mer at random.
// src 0 / 1: the compact / 64-bit shadow's pending counts; src 2: a flat list of n {hash, count} pairs (keys[i * stride],
// counts[i * stride]) -- add(), load(), the multi-GPU merge.  tallies (may be null) += CTR_* of the pass.
kct_status partitioned_pairs_pass(kct_table *t, int src, const du64 *keys, const du64 *counts, u64 n, int stride, u64 *tallies) {
    const bool compact = src == 0;
    const Levels L = levels_for(log2_u64(t->cap >> t->block_bits), t->num_cus, t->tune.pbits);
    const u64 P = L.P;
    const u64 sslots = src == 2 ? n : compact ? compact_slots(t) : t->shadow_cap;
    const u64 npairs = std::max<u64>(1, src == 2 ? n : compact ? t->s32_keys : t->shadow_keys);  // at most this many pairs exist
    const int nwg = src == 2 ? (int)std::min<u64>(t->num_cus, (n + 8 * kct::kPartThreads - 1) / (8 * kct::kPartThreads))
                             : (int)std::min<u64>(t->num_cus, sslots >> kct::kBlockBitsMax);  // a workgroup takes whole shadow blocks
    const double fill = std::min(1.0, (double)npairs / (double)sslots);
    const unsigned int region_cap = region_capacity((double)(sslots / nwg + kct::kPartThreads) * fill / (double)P);  // pairs per (workgroup, bin)
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 16));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_pairs_ovf.reserve(npairs * 16 + 64));  // pairs that found ring or region full (its own buffer: d_aux may hold the input)
    KCT_TRY(zero_counters(t));
    du64 *d_ovf_n = t->d_counters + kNumCounters + 5;
    const bool fresh = t->lazy_empty;
    kct::FlushPartitionArgs fa;
    fa.shadow = compact ? (void *)t->shadow32 : (void *)t->shadow; fa.shadow_blocks = src == 2 ? 0u : (unsigned int)(sslots >> kct::kBlockBitsMax); fa.k = t->k;
    fa.shadow_sbits = t->s32_sbits; fa.shadow_bin0 = t->s32_bin0;
    fa.pair_keys = keys; fa.pair_counts = counts; fa.pair_stride = stride; fa.npairs = n;
    fa.table_block_bits = t->block_bits + L.sub_bits; fa.pbits = L.pbits;  // (two levels: the first-level bins are super-bins)
    fa.scratch = (ulonglong2 *)t->d_scratch.p; fa.region_cap = region_cap; fa.region_count = (unsigned int *)t->d_regions.p;
    fa.ovf = (du64 *)t->d_pairs_ovf.p; fa.ovf_cap = npairs; fa.ovf_n = d_ovf_n;
    {
        ProfScope ps(t, "flush_partition_kernel");
        // (the popular k get the MurmurHash3 structure at compile time: the kernel is bound by hashing the pending k-mers)
        if (src == 0 && t->k == 21) hipLaunchKernelGGL((kct::flush_partition_kernel<0, 21>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else if (src == 0) hipLaunchKernelGGL((kct::flush_partition_kernel<0, 0>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else if (src == 1 && t->k == 31) hipLaunchKernelGGL((kct::flush_partition_kernel<1, 31>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else if (src == 1) hipLaunchKernelGGL((kct::flush_partition_kernel<1, 0>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
        else hipLaunchKernelGGL((kct::flush_partition_kernel<2, 0>), dim3(nwg), dim3(kct::kPartThreads), 0, t->stream, fa);
    }
    HIP_TRY(hipGetLastError());
    kct::AggregatePairsArgs pa;
    pa.words = t->slots; pa.block_bits = t->block_bits;
    pa.scratch = (const ulonglong2 *)t->d_scratch.p; pa.seg_stride = P * region_cap; pa.block_stride = region_cap;
    pa.region_count = (const unsigned int *)t->d_regions.p; pa.nregions = nwg;
    if (L.two) {
        const unsigned int out_cap = (region_capacity((double)npairs / (double)L.B / (double)L.W) + 15u) & ~15u;  // pairs per (block, writer): 256-byte multiples
        KCT_TRY(t->d_scratch2.reserve(L.B * L.W * out_cap * 16));
        KCT_TRY(t->d_regions2.reserve(L.B * L.W * 4));
        kct::RepartitionArgs ra;
        ra.mask = t->cap - 1; ra.block_bits = t->block_bits; ra.sub_bits = L.sub_bits;
        ra.in = t->d_scratch.p; ra.in_cap = region_cap; ra.in_count = (const unsigned int *)t->d_regions.p;
        ra.nseg = nwg; ra.nbins = (int)P; ra.writers = (int)L.W;
        ra.out = t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
        ra.ovf = (du64 *)t->d_pairs_ovf.p; ra.ovf_cap = (unsigned int)std::min<u64>(npairs, 0xFFFFFFFFu); ra.ovf_count = nullptr; ra.overflow = nullptr; ra.ovf_n = d_ovf_n;
        ra.min_lines = repartition_min_lines(t, kct::kRingEntries / 2, L.sub_bits, 16);
        {
            ProfScope ps(t, "repartition_kernel<pairs>");
            hipLaunchKernelGGL((kct::repartition_kernel<ulonglong2, false>), dim3((unsigned)(P * L.W)), dim3(kct::kPartThreads), 0, t->stream, ra);
        }
        HIP_TRY(hipGetLastError());
        pa.scratch = (const ulonglong2 *)t->d_scratch2.p; pa.seg_stride = out_cap; pa.block_stride = L.W * out_cap;
        pa.region_count = (const unsigned int *)t->d_regions2.p; pa.nregions = (int)L.W;
    }
    pa.fresh = fresh ? 1 : 0;
    pa.nblocks = (unsigned int)L.B;
    KCT_TRY(failed_blocks(t, L.B, &pa.failed));
    pa.counters = t->d_counters;
    {
        ProfScope ps(t, "aggregate_pairs_kernel");
        // two levels: one workgroup per CU walks the blocks (a block's stores drain under the next block's merge)
        unsigned grid = (unsigned)L.B;
        if (L.two && !t->tune.pairs_nopersist) grid = (unsigned)std::min<u64>(L.B, (u64)t->num_cus);
        hipLaunchKernelGGL(kct::aggregate_pairs_kernel, dim3(grid), dim3(kct::kPartThreads), 0, t->stream, pa);
    }
    HIP_TRY(hipGetLastError());
    t->lazy_empty = false;
    // the (normally few) pairs that did not fit ring or region: the direct insert, into a spill list of their number
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    const u64 nfailed = t->h_counters[kNumCounters + 7], failed_entries = t->h_counters[kNumCounters + 3];
    const u64 n_ovf = std::min<u64>(t->h_counters[kNumCounters + 5], npairs);
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (tallies) for (int i = 0; i < 4; ++i) tallies[i] += c[i];
    u64 tl[4] = {0, 0, 0, 0};
    if (n_ovf) {
        // (a copy: merge_pairs may take the partitioned route again, which writes d_pairs_ovf)
        KCT_TRY(t->d_aux2.reserve(n_ovf * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_pairs_ovf.p, n_ovf * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux2.p, (const du64 *)t->d_aux2.p + 1, n_ovf, 2, tl));
    }
    if (nfailed)  // table blocks that overflowed: the table grows, their pairs are merged with the direct insert
        KCT_TRY(recount_failed(t, 3, pa.scratch, pa.seg_stride, pa.block_stride, pa.region_count, pa.nregions, nfailed, failed_entries, 0, tl));
    if (tallies) for (int i = 0; i < 4; ++i) tallies[i] += tl[i];
    return KCT_OK;
}

kct_status flush_partitioned(kct_table *t, bool compact) { return partitioned_pairs_pass(t, compact ? 0 : 1, nullptr, nullptr, 0, 0, nullptr); }

// The partitioned pair route reads and writes the whole table once: worth it for a table of up to 1024 blocks (128 MiB)
// from 2^18 pairs on, for a larger one once the pairs are a sixteenth of its slots (a random table access costs ~0.1 ns,
// streaming a slot ~7 ps).
bool pairs_partition_pays(const kct_table *t, u64 n) {
    const u64 blocks = t->cap >> t->block_bits;
    if (!partition_geometry_ok(t) || t->block_bits != kct::kBlockBitsMax || t->tune.flush_atomic || t->force_path == 1) return false;
    return blocks <= 1024 ? n >= (1ULL << 18) : n >= t->cap / 16;
}

kct_status merge_pairs_partitioned(kct_table *t, const du64 *d_keys, const du64 *d_counts, u64 n, int stride, u64 tallies[4]) {
    return partitioned_pairs_pass(t, 2, d_keys, d_counts, n, stride, tallies);
}

kct_status flush_compact(kct_table *t) {
    if (!t->s32_dirty) return KCT_OK;
    t->s32_dirty = false;
    t->s32_windows = 0;
    const u64 blocks = t->cap >> t->block_bits;
    if (pairs_partition_pays(t, std::max<u64>(t->s32_keys, blocks <= 1024 ? (1ULL << 18) : 0))) return flush_partitioned(t, true);
    const u64 slots = compact_slots(t);
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(std::max<u64>(t->s32_keys, 1) * 16));
    KCT_TRY(zero_counters(t));
    {
        ProfScope ps(t, "shadow32_flush_kernel");
        hipLaunchKernelGGL(kct::shadow32_flush_kernel, dim3(merge_grid(slots)), dim3(kct::kBlock), 0, t->stream, t->shadow32,
                           (int)kct::kBlockBitsMax, slots, view(t, std::max<u64>(t->s32_keys, 1)), (int)t->k, t->d_counters, t->s32_sbits, t->s32_bin0);
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

kct_status flush_shadow64(kct_table *t);

kct_status flush_shadow128(kct_table *t);

kct_status flush_shadow(kct_table *t) {
    KCT_TRY(flush_compact(t));
    KCT_TRY(flush_shadow64(t));
    KCT_TRY(flush_shadow128(t));
    if (t->pending_pairs) {  // pairs that dedupe-first passes set aside while the table was lazily empty
        const u64 n = t->pending_pairs;
        t->pending_pairs = 0;
        u64 tl[4] = {0, 0, 0, 0};
        KCT_TRY(merge_pairs(t, (const du64 *)t->d_pending.p, (const du64 *)t->d_pending.p + 1, n, 2, tl));
        HIP_TRY(hipMemsetAsync(t->d_counters + kNumCounters + 8, 0, 8, t->stream));
    }
    return KCT_OK;
}

kct_status flush_shadow64(kct_table *t) {
    if (!t->shadow_dirty) return KCT_OK;
    t->shadow_dirty = false;
    {
        const u64 blocks = t->cap >> t->block_bits, sblocks = t->shadow_cap >> kct::kBlockBitsMax;
        if (t->shadow_block_bits == kct::kBlockBitsMax && sblocks >= 1 &&
            pairs_partition_pays(t, std::max<u64>(t->shadow_keys, blocks <= 1024 ? (1ULL << 18) : 0)))
            return flush_partitioned(t, false);
    }
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(std::max<u64>(t->shadow_keys, 1) * 16));
    KCT_TRY(zero_counters(t));
    {
        ProfScope ps(t, "shadow_flush_kernel");
        hipLaunchKernelGGL(kct::shadow_flush_kernel, dim3(merge_grid(t->shadow_cap)), dim3(kct::kBlock), 0, t->stream, t->shadow, shadow_geom(t),
                           view(t, std::max<u64>(t->shadow_keys, 1)), (int)t->k, t->d_counters);
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

// The list K2 writes the numbers of the blocks it abandons to; the counters are d_counters scratch words 7 (blocks) and 3
// (their entries), zeroed with the tallies.
kct_status failed_blocks(kct_table *t, u64 nblocks, kct::FailedBlocks *fb) {
    KCT_TRY(t->d_failed.reserve(nblocks * 4 + 64));
    fb->list = (unsigned int *)t->d_failed.p;
    fb->n = t->d_counters + kNumCounters + 7;
    fb->entries = t->d_counters + kNumCounters + 3;
    return KCT_OK;
}

// Overflow entries of a pass, exactly (the per-workgroup counts are read back; waits for the stream).
kct_status overflow_total(kct_table *t, const unsigned int *d_counts_a, size_t na, const unsigned int *d_counts_b, size_t nb, u64 *total) {
    std::vector<unsigned int> c(na + nb);
    if (na) HIP_TRY(hipMemcpyAsync(c.data(), d_counts_a, na * 4, hipMemcpyDeviceToHost, t->stream));
    if (nb) HIP_TRY(hipMemcpyAsync(c.data() + na, d_counts_b, nb * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *total = 0;
    for (unsigned int v : c) *total += v;
    return KCT_OK;
}

// The abandoned blocks' entries -> the real table with the direct insert (recount_failed_kernel), after making room.
// mode 0: hashes that the TABLE's own blocks could not take (the table grows first); 1 / 2: mix64 / compact entries of
// shadow blocks (hashed on the way); 3: {hash, count} pairs of a pair merge (the table grows first).
// tallies[] += CTR_* of what was placed, replays included.
kct_status recount_failed(kct_table *t, int mode, const void *scratch, u64 seg_stride, u64 block_stride, const unsigned int *region_count, int nregions,
                          u64 nfailed, u64 entries, int sbits, u64 tallies[4]) {
    KCT_DBG(t, "recounting %llu abandoned blocks (%llu entries, mode %d)\n", (unsigned long long)nfailed, (unsigned long long)entries, mode);
    if (mode == 0 || mode == 3) KCT_TRY(grow_to(t, spill_growth_target(t, entries)));
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(std::max<u64>(entries, 1) * 16));
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 1) * sizeof(u64), t->stream));  // tallies and the spill cursor
    const kct::TableView tv = view(t, std::max<u64>(entries, 1));
    const unsigned grid = (unsigned)std::min<u64>(nfailed, 2048);
    const unsigned int *fl = (const unsigned int *)t->d_failed.p;
    {
        ProfScope ps(t, "recount_failed_kernel");
        if (mode == 0) hipLaunchKernelGGL(kct::recount_failed_kernel<0>, dim3(grid), dim3(kct::kBlock), 0, t->stream, scratch, seg_stride, block_stride, region_count, nregions, fl, nfailed, tv, t->d_counters, (int)t->k, sbits);
        else if (mode == 1) hipLaunchKernelGGL(kct::recount_failed_kernel<1>, dim3(grid), dim3(kct::kBlock), 0, t->stream, scratch, seg_stride, block_stride, region_count, nregions, fl, nfailed, tv, t->d_counters, (int)t->k, sbits);
        else if (mode == 2) hipLaunchKernelGGL(kct::recount_failed_kernel<2>, dim3(grid), dim3(kct::kBlock), 0, t->stream, scratch, seg_stride, block_stride, region_count, nregions, fl, nfailed, tv, t->d_counters, (int)t->k, sbits, t->s32_bin0);
        else hipLaunchKernelGGL(kct::recount_failed_kernel<3>, dim3(grid), dim3(kct::kBlock), 0, t->stream, scratch, seg_stride, block_stride, region_count, nregions, fl, nfailed, tv, t->d_counters, (int)t->k, sbits);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], spilled;
    KCT_TRY(read_counters(t, c, &spilled));
    for (int i = 0; i < 4; ++i) tallies[i] += c[i];
    t->n_keys += c[kct::CTR_NEWKEYS];
    if (spilled) {
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(grow_to(t, spill_growth_target(t, spilled)));
        KCT_TRY(merge_pairs(t, (const du64 *)t->d_aux2.p, (const du64 *)t->d_aux2.p + 1, spilled, 2, tallies));
    }
    return KCT_OK;
}

// Room in the pending pair list for every overflow entry of a pass (the per-workgroup counts are read back: exact).
kct_status reserve_pending(kct_table *t, const unsigned int *d_counts_a, size_t na, const unsigned int *d_counts_b, size_t nb, kct::PendingList *pl) {
    std::vector<unsigned int> c(na + nb);
    if (na) HIP_TRY(hipMemcpyAsync(c.data(), d_counts_a, na * 4, hipMemcpyDeviceToHost, t->stream));
    if (nb) HIP_TRY(hipMemcpyAsync(c.data() + na, d_counts_b, nb * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    u64 total = 0;
    for (unsigned int v : c) total += v;
    KCT_TRY(t->d_pending.reserve_keep((t->pending_pairs + total) * 16 + 64, t->pending_pairs * 16, t->stream));
    pl->pairs = (du64 *)t->d_pending.p;
    pl->cap = t->pending_pairs + total;
    pl->n = t->d_counters + kNumCounters + 8;
    return KCT_OK;
}

// What a dedupe-first pass found, for the caller's decisions.
struct DedupeOutcome {
    u64 new_keys = 0;   // k-mers the pass met for the first time (new shadow keys + what went past the shadow)
    u64 blocked = 0;    // entries that found their shadow block full
    u64 counted = 0;    // entries counted into the shadow (dry runs)
};

// After a dedupe-first pass: too few repeats to be worth it, or the shadow is filling up?  `probe`: the pass was the
// bounded sample of a large call (consume_stream decides from its new-key ratio instead).
kct_status after_dedupe_pass(kct_table *t, bool compact, u64 npos, const DedupeOutcome &o, bool probe) {
    if (!probe && t->force_path != 3 && o.new_keys * 3 > npos) {  // too few repeats for any dedupe-first variant
        KCT_TRY(flush_shadow(t));
        t->dedupe_off = true;
        t->dedupe_hint = false;
        return KCT_OK;
    }
    if (!probe) t->dedupe_hint = true;
    if (compact) {
        if (o.blocked * 50 > npos || (double)t->s32_keys > kMaxLoad * (double)compact_slots(t)) {  // outgrown
            KCT_TRY(flush_compact(t));
            // The k-mers live in the table too: normally it is as full as the shadow and grows before the next pass
            // (maybe_grow), and a table of another geometry gets a new shadow (ensure_shadow32).  If it stays as it is,
            // the table-sized 64-bit shadow takes over.
            if (compact_sbits_for(t) == t->s32_sbits && (double)t->n_keys <= kMaxLoad * (double)t->cap) t->compact_off = true;
        }
    } else if (o.blocked * 50 > npos && t->force_path != 3) {
        KCT_TRY(flush_shadow(t));
        t->dedupe_off = true;
        t->dedupe_hint = false;
    }
    return KCT_OK;
}

// (compact_pays: path_policy.h.)  Same contract as consume_partitioned(raw = true).
// dry (the probe of a large table, counting into its own small shadow): K1 and K2 only -- nothing reaches the real table,
// no state changes, *n_out stays; what the pass saw comes back in *dry.
kct_status consume_compact(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled, bool probe,
                           DedupeOutcome *dry = nullptr) {
    *handled = false;
    const int k = t->k;
    {
        bool ok = true;
        KCT_TRY(ensure_shadow32(t, probe ? kCompactBlockBits : compact_sbits_for(t), &ok));  // (the probe's shadow is the small one, swapped in by the caller)
        if (!ok) { t->compact_off = true; return KCT_OK; }
    }
    // u32 counts: a pending count grows by at most 1/256 of the window starts consumed (aggregate_blocks32_kernel) -- 1/128 with the short flush
    // interval of bursty input (a bin's 32-entry stretch per 4096 appends): converted before 2^38 window starts either way
    if (t->s32_dirty && t->s32_windows + npos >= (1ULL << 38)) KCT_TRY(flush_compact(t));
    const int sbits = t->s32_sbits;
    const bool two_level = sbits > kCompactBlockBits;
    const int pbits = kCompactBlockBits;                             // K1's bins are always the value's top 10 bits
    const int gbits = compact_group_bits(sbits), sub_bits = sbits - pbits + gbits;  // second level: 2^gbits bins per super-bin, 2^sub_bits blocks each
    const u64 nsuper = (1ULL << pbits) >> gbits;
    const u64 W2 = two_level_writers(nsuper, t->num_cus);           // workgroups per super-bin, so that the second level fills the chip
    const u64 P = 1ULL << pbits, B = 1ULL << sbits;
    const int nwg = t->num_cus;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile;
    const u64 tiles_per_wg = (ntiles + nwg - 1) / nwg;
    const unsigned int region_cap = (region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)P) + 15u) & ~15u;  // 16-entry lines
    const unsigned int ovf_cap = overflow_capacity(tiles_per_wg * kct::kPartTile);
    if (!dry && !two_level) KCT_TRY(materialize(t));  // (two levels: only if something must go to the table at once, below)
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 4));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve((u64)nwg * ovf_cap * 8 + (u64)nwg * 4));
    // the overflow merges' own spill list (entries the real table has no room for): one level -- everything is one
    // submission -- sized for every overflow region being full; two levels: sized after the fact, below
    if (!two_level) KCT_TRY(t->d_spill.reserve((u64)nwg * ovf_cap * 16));
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + (u64)nwg * ovf_cap);

    kct::PartitionArgs pa;
    pa.mask = compact_slots(t) - 1; pa.block_bits = kct::kBlockBitsMax; pa.pbits = pbits;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = t->ablate;
    packed_args(t, d_stream, &pa);
    launch_partition(t, 2, d_stream, chunk_bytes, ntiles, pa);
    HIP_TRY(hipGetLastError());
    kct::Aggregate32Args aa;
    aa.words = t->shadow32; aa.block_bits = kct::kBlockBitsMax; aa.sbits = sbits;
    aa.scratch = (const unsigned int *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
    aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    unsigned int ovf2_cap = 0, *d_ovf2_count = nullptr;
    u64 n2 = 0;
    if (two_level) {
        // second level: one workgroup per super-bin spreads its entries over the super-bin's 2^sub_bits shadow blocks
        const unsigned int out_cap = (region_capacity((double)npos / (double)B / (double)W2) + 63u) & ~63u;  // 256-byte multiples
        n2 = nsuper * W2;   // second-level workgroups
        ovf2_cap = overflow_capacity(npos / n2);
        KCT_TRY(t->d_scratch2.reserve(B * W2 * out_cap * 4));
        KCT_TRY(t->d_regions2.reserve(B * W2 * 4));
        KCT_TRY(t->d_irr2.reserve(n2 * ovf2_cap * 8 + n2 * 4));
        d_ovf2_count = (unsigned int *)((du64 *)t->d_irr2.p + n2 * ovf2_cap);
        kct::RepartitionArgs ra;
        ra.mask = compact_slots(t) - 1; ra.block_bits = kct::kBlockBitsMax; ra.sub_bits = sub_bits; ra.gbits = gbits;
        ra.in = t->d_scratch.p; ra.in_cap = region_cap; ra.in_count = (const unsigned int *)t->d_regions.p;
        ra.nseg = nwg << gbits; ra.nbins = (int)P; ra.writers = (int)W2;
        ra.out = t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
        ra.ovf = (du64 *)t->d_irr2.p; ra.ovf_cap = ovf2_cap; ra.ovf_count = d_ovf2_count; ra.overflow = d_overflow; ra.ovf_n = nullptr;
        ra.min_lines = repartition_min_lines(t, kct::kRingEntries * 2, sub_bits, 4);
        {
            ProfScope ps(t, "repartition_kernel<compact>");
            // One flush per slab (16 appends per thread) where a sub-bin's stretch of the ring is deep enough for it: up to 128 sub-bins,
            // 256 entries each -- an interval brings 128 on average; K1b -5 %.  With shallower stretches the overflow entries eat the gain.
            if (sub_bits <= 7) hipLaunchKernelGGL((kct::repartition_kernel<unsigned int, true>), dim3((unsigned)n2), dim3(kct::kPartThreads), 0, t->stream, ra);
            else hipLaunchKernelGGL((kct::repartition_kernel<unsigned int, false>), dim3((unsigned)n2), dim3(kct::kPartThreads), 0, t->stream, ra);
        }
        HIP_TRY(hipGetLastError());
        aa.scratch = (const unsigned int *)t->d_scratch2.p; aa.seg_stride = out_cap; aa.block_stride = W2 * out_cap;
        aa.region_count = (const unsigned int *)t->d_regions2.p; aa.nregions = (int)W2;
    }
    aa.fresh = t->s32_empty ? 1 : 0; aa.overflow = d_overflow; aa.ablate = t->ablate;
    KCT_TRY(failed_blocks(t, B, &aa.failed));
    aa.counters = t->d_counters;
    {
        ProfScope ps(t, "aggregate_blocks32_kernel");
        // two workgroups per CU walk the blocks: a block's stores drain under the next block's load instead of in front of
        // the next workgroup's start (K2-32 -1.3 %)
        aa.nblocks = (unsigned int)B;
        const unsigned grid2 = t->tune.k2_nopersist ? (unsigned)B : (unsigned)std::min<u64>(B, 2 * (u64)t->num_cus);
        // (bursty input: folding equal neighbours among the four entries a lane holds before the LDS add was measured -- K2-32 0.581 against
        // 0.489 ms on position-sorted C2: the same k-mer's entries lie a lane apart, not inside one lane's four -- and is not kept)
        if (two_level) hipLaunchKernelGGL(kct::aggregate_blocks32_kernel<true>, dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
        else hipLaunchKernelGGL(kct::aggregate_blocks32_kernel<false>, dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    // What does not fit the shadow goes to the real table: K1's (and the second level's) overflow regions with the direct
    // insert, and -- after the pass -- the entries of shadow blocks that K2 had to abandon (normally none).  One level: the
    // overflow merge rides in the same submission (no host round trip).  Two levels (passes of 10^8+ windows): after the
    // counters have been read; a table that is still lazily empty stays untouched (pending pair list).
    u64 c[4], unused;
    if (dry) {
        KCT_TRY(read_counters(t, c, &unused));
        if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;
        *handled = true;
        dry->counted = c[kct::CTR_COUNTED]; dry->new_keys = c[kct::CTR_NEW_BY_ZERO]; dry->blocked = t->h_counters[kNumCounters + 3];
        KCT_DBG(t, "compact dedupe pass (dry probe): npos=%llu counted=%llu new keys=%llu blocked=%llu\n", (unsigned long long)npos,
                (unsigned long long)dry->counted, (unsigned long long)dry->new_keys, (unsigned long long)dry->blocked);
        return KCT_OK;
    }
    kct::TableView mv = view(t, (u64)nwg * ovf_cap);
    kct::PendingList pend;  // a lazily empty table stays untouched: the overflow entries wait in the pending pair list
    if (two_level) {
        KCT_TRY(read_counters(t, c, &unused));
        if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // K1 / K1b gave up: K2 exited early, nothing was touched
        if (t->lazy_empty && t->h_counters[kNumCounters + 7] == 0) KCT_TRY(reserve_pending(t, d_ovf_count, nwg, d_ovf2_count, n2, &pend));
        else {
            u64 total = 0;
            KCT_TRY(overflow_total(t, d_ovf_count, nwg, d_ovf2_count, n2, &total));
            KCT_TRY(materialize(t));
            KCT_TRY(t->d_spill.reserve(std::max<u64>(total, 1) * 16));
            mv = view(t, std::max<u64>(total, 1));
        }
    }
    mv.spill_n = t->d_counters + kNumCounters + 5;
    {
        ProfScope ps(t, "merge_overflow_kernel");
        hipLaunchKernelGGL(kct::merge_overflow_kernel<2>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                           (const unsigned int *)d_ovf_count, nwg, ovf_cap, (const du64 *)d_overflow, mv, t->d_counters, k, (const du64 *)nullptr, pend);
        if (two_level)
            hipLaunchKernelGGL(kct::merge_overflow_kernel<2>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr2.p,
                               (const unsigned int *)d_ovf2_count, (int)n2, ovf2_cap, (const du64 *)d_overflow, mv, t->d_counters, k, (const du64 *)nullptr, pend);
    }
    HIP_TRY(hipGetLastError());
    u64 c2[4];
    KCT_TRY(read_counters(t, c2, &unused));
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // K1 gave up: every kernel after it exited early, nothing was touched
    *handled = true;
    // (two levels: the second read holds the first submission's tallies too -- the counters were not zeroed in between)
    const u64 counted = c2[kct::CTR_COUNTED], new_keys = c2[kct::CTR_NEW_BY_ZERO], spilled2 = t->h_counters[kNumCounters + 5];
    const u64 nfailed = t->h_counters[kNumCounters + 7], blocked = t->h_counters[kNumCounters + 3];  // abandoned shadow blocks, their entries
    t->s32_empty = false;
    t->s32_dirty = true;
    t->s32_keys += new_keys;
    t->s32_windows += npos;
    if (pend.pairs) t->pending_pairs = t->h_counters[kNumCounters + 8];
    KCT_DBG(t, "compact dedupe pass%s: npos=%llu blocks=%llu region_cap=%u counted=%llu new keys=%llu (total %llu) abandoned=%llu blocks / %llu entries merged=%llu spilled=%llu\n",
            probe ? " (probe)" : "", (unsigned long long)npos, (unsigned long long)B, region_cap, (unsigned long long)counted, (unsigned long long)new_keys,
            (unsigned long long)t->s32_keys, (unsigned long long)nfailed, (unsigned long long)blocked, (unsigned long long)c2[kct::CTR_TOTAL_ADDED],
            (unsigned long long)spilled2);
    *n_out += counted + c2[kct::CTR_TOTAL_ADDED];  // (see consume_partitioned about n and a MurmurHash3 value of 0)
    // K-mers that arrive in bursts (position-sorted reads) overflow a bin's stretch of the ring between two flushes: when more than 2 % of a
    // pass took the overflow route (C2 sorted by position: 4.4 % of the probe, 6.6 % of the pass), K1 flushes twice as often from the next
    // pass on (k1_kernel.h FE = 4: 0.35 % there).  It stays that way for the table's life: what a pass with FE = 4 overflows says nothing about
    // what the same input would do to FE = 8 (an "until a pass stays under 0.25 %" rule flipped back and forth between the probe and the
    // pass of every call), and the price on input that would not have needed it is 3 % of K1.
    {
        const u64 over = c2[kct::CTR_TOTAL_ADDED], all = counted + over;
        if (all >= (1ULL << 20) && over * 50 > all) t->compact_bursty = true;
    }
    t->n_keys += c2[kct::CTR_NEWKEYS];
    u64 new_in_table = c2[kct::CTR_NEWKEYS];
    if (spilled2) {
        KCT_TRY(t->d_aux2.reserve(spilled2 * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, mv.spill, spilled2 * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled2, n_out));
    }
    if (nfailed) {  // shadow blocks that overflowed: their entries are hashed and counted into the real table
        u64 tl[4] = {0, 0, 0, 0};
        KCT_TRY(recount_failed(t, 2, aa.scratch, aa.seg_stride, aa.block_stride, aa.region_count, aa.nregions, nfailed, blocked, sbits, tl));
        *n_out += tl[kct::CTR_TOTAL_ADDED];
        new_in_table += tl[kct::CTR_NEWKEYS];
    }
    DedupeOutcome o;
    o.new_keys = new_keys + new_in_table;
    o.blocked = blocked;
    return after_dedupe_pass(t, true, npos, o, probe);
}


// ---- 128-bit dedupe-first pass (33 <= k <= 64) ------------------------------------------------------------------------------------
// K1 (MODE 3) partitions mix128 pairs of the packed k-mers by x into 1024 bins; aggregate_blocks128_kernel counts them into the
// 128-bit shadow (1024 blocks x 4096 slots, one level whatever the table's size); what overflows a ring (or a shadow block) is
// hashed and goes to the real table with the direct insert; reading the table converts the pending counts (shadow128_flush_kernel).
kct_status flush_shadow128(kct_table *t) {
    if (!t->s128_dirty) return KCT_OK;
    t->s128_dirty = false;
    t->s128_windows = 0;
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_spill.reserve(std::max<u64>(t->s128_keys, 1) * 16));
    KCT_TRY(zero_counters(t));
    {
        ProfScope ps(t, "shadow128_flush_kernel");
        hipLaunchKernelGGL(kct::shadow128_flush_kernel, dim3(merge_grid(1024ULL << kct::kBlockBits128)), dim3(kct::kBlock), 0, t->stream, t->shadow128, 1024u,
                           view(t, std::max<u64>(t->s128_keys, 1)), (int)t->k, t->d_counters);
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

kct_status consume_raw128(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled) {
    *handled = false;
    const int k = t->k, nwg = t->num_cus;
    if (!t->shadow128) {
        if (hipMalloc((void **)&t->shadow128, 1024 * kct::kBlockWords128 * 8) != hipSuccess) { (void)hipGetLastError(); t->dedupe128_off = true; return KCT_OK; }
        t->s128_empty = true; t->s128_keys = 0;
    }
    // u32 counts with bit 31 taken: a pending count grows by at most the window starts consumed
    if (t->s128_dirty && t->s128_windows + npos >= (1ULL << 31)) KCT_TRY(flush_shadow128(t));
    const u64 P = 1024;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile, tiles_per_wg = (ntiles + nwg - 1) / nwg;
    const unsigned int region_cap = (region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)P) + 3u) & ~3u;  // entries of 16 B: four per line
    const unsigned int ovf_cap = overflow_capacity(tiles_per_wg * kct::kPartTile);
    KCT_TRY(materialize(t));
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 16));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve((u64)nwg * ovf_cap * 16 + (u64)nwg * 4));
    KCT_TRY(t->d_spill.reserve((u64)nwg * ovf_cap * 16));
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + 2 * (u64)nwg * ovf_cap);
    kct::PartitionArgs pa;
    pa.mask = (P << kct::kBlockBits128) - 1; pa.block_bits = kct::kBlockBits128; pa.pbits = 10;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = t->ablate;
    packed_args(t, d_stream, &pa);
    {
        ProfScope ps(t, "partition_windows_kernel<raw128>");
        PartitionRaw128::run(k, t->stream, nwg, d_stream, chunk_bytes, ntiles, pa);
    }
    HIP_TRY(hipGetLastError());
    kct::Aggregate128Args aa;
    aa.words = t->shadow128; aa.scratch = (const ulonglong2 *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
    aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    aa.fresh = t->s128_empty ? 1 : 0; aa.overflow = d_overflow; aa.nblocks = (unsigned int)P;
    KCT_TRY(failed_blocks(t, P, &aa.failed));
    aa.counters = t->d_counters;
    {
        ProfScope ps(t, "aggregate_blocks128_kernel");
        hipLaunchKernelGGL(kct::aggregate_blocks128_kernel, dim3((unsigned)std::min<u64>(P, (u64)nwg)), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    // K1's overflow entries: hashed, into the real table with the direct insert (same submission)
    kct::TableView mv = view(t, (u64)nwg * ovf_cap);
    mv.spill_n = t->d_counters + kNumCounters + 5;
    {
        ProfScope ps(t, "merge_entries128_kernel");
        hipLaunchKernelGGL(kct::merge_entries128_kernel, dim3(256), dim3(kct::kBlock), 0, t->stream, (const ulonglong2 *)t->d_irr.p, (const unsigned int *)d_ovf_count,
                           nwg, (u64)ovf_cap, 0ULL, 0ULL, (const unsigned int *)nullptr, 0ULL, (const du64 *)d_overflow, mv, t->d_counters, k);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], unused;
    KCT_TRY(read_counters(t, c, &unused));
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // K1 gave up: every kernel after it exited early, nothing was touched
    *handled = true;
    const u64 counted = c[kct::CTR_COUNTED], new_keys = c[kct::CTR_NEW_BY_ZERO], spilled = t->h_counters[kNumCounters + 5];
    const u64 nfailed = t->h_counters[kNumCounters + 7], blocked = t->h_counters[kNumCounters + 3];
    t->s128_empty = false; t->s128_dirty = true;
    t->s128_keys += new_keys; t->s128_windows += npos;
    KCT_DBG(t, "128-bit dedupe pass: npos=%llu counted=%llu new keys=%llu (total %llu) abandoned=%llu blocks / %llu entries merged=%llu spilled=%llu\n",
            (unsigned long long)npos, (unsigned long long)counted, (unsigned long long)new_keys, (unsigned long long)t->s128_keys, (unsigned long long)nfailed,
            (unsigned long long)blocked, (unsigned long long)c[kct::CTR_TOTAL_ADDED], (unsigned long long)spilled);
    *n_out += counted + c[kct::CTR_TOTAL_ADDED];
    t->n_keys += c[kct::CTR_NEWKEYS];
    u64 new_in_table = c[kct::CTR_NEWKEYS];
    if (spilled) {
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, mv.spill, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, n_out));
    }
    if (nfailed) {  // shadow blocks that overflowed: their entries are hashed and counted into the real table
        KCT_TRY(t->d_spill.reserve(std::max<u64>(blocked, 1) * 16));
        HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 1) * sizeof(u64), t->stream));
        const kct::TableView tv = view(t, std::max<u64>(blocked, 1));
        {
            ProfScope ps(t, "merge_entries128_kernel");
            hipLaunchKernelGGL(kct::merge_entries128_kernel, dim3((unsigned)std::min<u64>(nfailed * nwg, 4096)), dim3(kct::kBlock), 0, t->stream, aa.scratch, aa.region_count,
                               aa.nregions, 0ULL, aa.seg_stride, aa.block_stride, (const unsigned int *)t->d_failed.p, nfailed, (const du64 *)nullptr, tv, t->d_counters, k);
        }
        HIP_TRY(hipGetLastError());
        u64 c3[4], sp3;
        KCT_TRY(read_counters(t, c3, &sp3));
        *n_out += c3[kct::CTR_TOTAL_ADDED];
        t->n_keys += c3[kct::CTR_NEWKEYS];
        new_in_table += c3[kct::CTR_NEWKEYS];
        if (sp3) {
            KCT_TRY(t->d_aux2.reserve(sp3 * 16));
            HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, sp3 * 16, hipMemcpyDeviceToDevice, t->stream));
            KCT_TRY(replay_spill(t, sp3, n_out));
        }
    }
    // too few repeats, or the fixed-size shadow is filling up: convert and go back to hashing every window
    if (t->force_path != 3 && ((new_keys + new_in_table) * 3 > npos || blocked * 50 > npos || t->s128_keys > kShadow128Keys)) {
        KCT_TRY(flush_shadow128(t));
        t->dedupe128_off = true;
        t->dedupe_hint = false;
    } else t->dedupe_hint = true;
    return KCT_OK;
}

// One pass of the partitioned path over window starts [0, npos) of d_stream.  *handled = false
// (and nothing counted) if the pass had to be abandoned; the caller then uses the direct path.
// raw = false: K1 hashes, K2 counts into the real table.
// raw = true (dedupe-first, k <= 32): K1 emits mix64(packed k-mer) values, K2 counts them into the shadow table
//        (same geometry); what does not fit (overflow regions, pairs that found their block full) is hashed and goes
//        to the real table at once.
kct_status consume_partitioned(kct_table *t, const unsigned char *d_stream, u64 chunk_bytes, u64 npos, u64 *n_out, bool *handled, bool raw, bool probe = false,
                               DedupeOutcome *dry = nullptr, int chunks = 1) {
    *handled = false;
    const int k = t->k;
    if (raw) {
        bool ok = true;
        KCT_TRY(ensure_shadow(t, probe ? std::min(t->cap, kProbeShadowSlots) : t->cap, &ok));  // (the probe's shadow is a small one)
        if (!ok) { t->dedupe_off = true; return KCT_OK; }
    }
    du64 *words = raw ? t->shadow : t->slots;
    // the geometry K1 / K1b / K2 work in: the table's, or the shadow's
    const u64 gcap = raw ? t->shadow_cap : t->cap;
    const int gbb = raw ? t->shadow_block_bits : t->block_bits;
    const int nwg = t->num_cus;
    const Levels L = levels_for(log2_u64(gcap >> gbb), nwg, t->tune.pbits);
    const bool two_level = L.two;
    const int bbits = L.bbits, pbits = L.pbits, sub_bits = L.sub_bits;
    const u64 P = L.P, B = L.B, W = L.W;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile;
    // Two levels: the pass may be cut into SUB-CHUNKS -- K1 and K1b run once per sub-chunk, K1's scratch regions are reused, and
    // every sub-chunk's K1b writes its own region slots of the blocks (RepartitionArgs::writer0 / wtot) -- so that ONE K2 pass
    // counts them all: 9.6 / C + 9.6 B of scratch per window start instead of 19.2, and the 100 M-read k = 31 / k = 51
    // configurations load and store their 16 GiB table (or shadow) once instead of twice.
    const u64 C = two_level ? (u64)std::max(1, std::min<int>(chunks, (int)((ntiles + 255) / 256))) : 1;
    const u64 tiles_per_chunk = (ntiles + C - 1) / C;
    const u64 tiles_per_wg = (tiles_per_chunk + nwg - 1) / nwg;
    const unsigned int region_cap = region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)P);
    const unsigned int ovf_cap = overflow_capacity(tiles_per_wg * kct::kPartTile);
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * 8));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve(C * nwg * ovf_cap * 8 + C * nwg * 4));
    // the overflow merges' own spill list (entries the real table has no room for): one level -- everything is one
    // submission -- sized for every overflow region being full; two levels: sized after the fact, below
    if (!two_level) KCT_TRY(t->d_spill.reserve((u64)nwg * ovf_cap * 16));
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + C * nwg * ovf_cap);  // [C][nwg]
    const bool fresh = raw ? t->shadow_empty : t->lazy_empty;

    kct::AggregateArgs aa;
    aa.words = words; aa.block_bits = gbb; aa.pbits = bbits;
    aa.fresh = fresh ? 1 : 0; aa.overflow = d_overflow; aa.ablate = t->ablate;
    KCT_TRY(failed_blocks(t, B, &aa.failed));
    aa.counters = t->d_counters;
    unsigned int ovf2_cap = 0, *d_ovf2_count = nullptr, out_cap = 0;
    if (two_level) {
        // second level: one workgroup per super-bin (and writer) spreads its hashes over the super-bin's blocks
        out_cap = (region_capacity((double)(tiles_per_chunk * kct::kPartTile) / (double)B / (double)W) + 31u) & ~31u;  // 256-byte multiples
        ovf2_cap = overflow_capacity(tiles_per_chunk * kct::kPartTile / P / W);
        KCT_TRY(t->d_scratch2.reserve(B * C * W * out_cap * 8));
        KCT_TRY(t->d_regions2.reserve(B * C * W * 4));
        KCT_TRY(t->d_irr2.reserve(C * P * W * ovf2_cap * 8 + C * P * W * 4));
        d_ovf2_count = (unsigned int *)((du64 *)t->d_irr2.p + C * P * W * ovf2_cap);  // [C][P * W]
    }
    for (u64 c = 0; c < C; ++c) {
        const u64 tile0 = c * tiles_per_chunk, ntiles_c = std::min(tiles_per_chunk, ntiles > tile0 ? ntiles - tile0 : 0);
        const u64 off = tile0 * kct::kPartTile;
        const u64 bytes_c = off < chunk_bytes ? std::min<u64>(chunk_bytes - off, ntiles_c * kct::kPartTile + k - 1) : 0;
        kct::PartitionArgs pa;
        pa.mask = gcap - 1; pa.block_bits = gbb + sub_bits; pa.pbits = pbits;
        pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
        pa.ovf = (du64 *)t->d_irr.p + c * nwg * ovf_cap; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count + c * nwg; pa.overflow = d_overflow;
        pa.ablate = t->ablate;  // measurement only; wrong counts when set
        packed_args(t, d_stream + off, &pa);
        launch_partition(t, raw ? 1 : 0, d_stream + off, bytes_c, ntiles_c, pa);
        HIP_TRY(hipGetLastError());
        if (!two_level) break;
        kct::RepartitionArgs ra;
        ra.mask = gcap - 1; ra.block_bits = gbb; ra.sub_bits = sub_bits;
        ra.in = t->d_scratch.p; ra.in_cap = region_cap; ra.in_count = (const unsigned int *)t->d_regions.p;
        ra.nseg = nwg; ra.nbins = (int)P; ra.writers = (int)W; ra.writer0 = (int)(c * W); ra.wtot = (int)(C * W);
        ra.out = t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
        ra.ovf = (du64 *)t->d_irr2.p + c * P * W * ovf2_cap; ra.ovf_cap = ovf2_cap; ra.ovf_count = d_ovf2_count + c * P * W; ra.overflow = d_overflow; ra.ovf_n = nullptr;
        ra.min_lines = repartition_min_lines(t, kct::kRingEntries, sub_bits, 8);
        {
            ProfScope ps(t, "repartition_kernel");
            // With >= 64 ring entries per bin the ring is flushed once per slab of eight entries per thread (half the ring per
            // interval) instead of twice, a bin's lines leaving two at a time: the flush machinery is ~a third of K1b's
            // instructions (K1b -6 %).
            if ((kct::kRingEntries >> sub_bits) >= 64 && !t->tune.k1b_half) {
                ra.min_lines = std::max(1u, ra.min_lines / 2);
                hipLaunchKernelGGL((kct::repartition_kernel<du64, true>), dim3((unsigned)(P * W)), dim3(kct::kPartThreads), 0, t->stream, ra);
            } else hipLaunchKernelGGL((kct::repartition_kernel<du64, false>), dim3((unsigned)(P * W)), dim3(kct::kPartThreads), 0, t->stream, ra);
        }
        HIP_TRY(hipGetLastError());
    }
    if (!two_level) {
        aa.scratch = (const du64 *)t->d_scratch.p; aa.seg_stride = P * region_cap; aa.block_stride = region_cap;
        aa.region_count = (const unsigned int *)t->d_regions.p; aa.nregions = nwg;
    } else {
        aa.scratch = (const du64 *)t->d_scratch2.p; aa.seg_stride = out_cap; aa.block_stride = C * W * out_cap;
        aa.region_count = (const unsigned int *)t->d_regions2.p; aa.nregions = (int)(C * W);
    }
    const u64 n_ovf1 = C * nwg, n_ovf2 = C * P * W;  // overflow regions of the first / second level (all sub-chunks)
    {
        ProfScope ps(t, raw ? "aggregate_blocks_kernel<shadow>" : "aggregate_blocks_kernel");
        // (a pass expected to bring mostly NEW k-mers into an empty table: the variant whose fast path claims slots itself)
        const bool claim = !raw && fresh && t->expect_new_keys, two = aa.nregions < kct::kPartThreads / 64;
        aa.nblocks = (unsigned int)B;
        // one workgroup per CU walks the blocks: a block's stores drain under the next block's load (K2 -3 % on C3 / C5, -7 % on C4's shard)
        const unsigned grid2 = t->tune.k2_nopersist ? (unsigned)B : (unsigned)std::min<u64>(B, (u64)t->num_cus);
        if (claim && two) hipLaunchKernelGGL((kct::aggregate_blocks_kernel<true, true>), dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
        else if (claim) hipLaunchKernelGGL((kct::aggregate_blocks_kernel<true, false>), dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
        else if (two) hipLaunchKernelGGL((kct::aggregate_blocks_kernel<false, true>), dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
        else hipLaunchKernelGGL((kct::aggregate_blocks_kernel<false, false>), dim3(grid2), dim3(kct::kPartThreads), 0, t->stream, aa);
    }
    HIP_TRY(hipGetLastError());
    // K1's (and K1b's) overflow regions go to the real table with the direct insert; so do -- after the pass -- the entries of
    // blocks that K2 had to abandon (normally none).  A counting pass into a table of one level: the overflow merge rides in
    // the same submission.  Otherwise the counters are read first: a shadow pass keeps K2's tallies (shadow keys) apart from
    // the merges' (table keys) and can leave a lazily empty table untouched (the overflow entries then wait in the pending
    // pair list); two levels size the merges' spill list exactly.
    kct::PendingList pend;
    kct::TableView mv = view(t, (u64)nwg * ovf_cap);
    u64 c[4] = {0, 0, 0, 0}, unused;
    if (raw || two_level) {
        KCT_TRY(read_counters(t, c, &unused));
        if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // abandoned by K1 / K1b: K2 exited early, nothing was touched
        if (dry) {  // (raw only) the probe of a large table: K1 and K2 into its own small shadow, nothing else
            *handled = true;
            dry->counted = c[kct::CTR_COUNTED]; dry->new_keys = c[kct::CTR_NEWKEYS]; dry->blocked = t->h_counters[kNumCounters + 3];
            return KCT_OK;
        }
        if (raw) HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 1) * sizeof(u64), t->stream));  // the tallies (and the unused spill cursor) only
        if (raw && two_level && t->lazy_empty && t->h_counters[kNumCounters + 7] == 0) KCT_TRY(reserve_pending(t, d_ovf_count, n_ovf1, d_ovf2_count, n_ovf2, &pend));
        else {
            if (raw) KCT_TRY(materialize(t));
            if (two_level) {
                u64 total = 0;
                KCT_TRY(overflow_total(t, d_ovf_count, n_ovf1, d_ovf2_count, n_ovf2, &total));
                KCT_TRY(t->d_spill.reserve(std::max<u64>(total, 1) * 16));
                mv = view(t, std::max<u64>(total, 1));
            }
        }
    }
    const u64 nfailed_k2 = t->h_counters[kNumCounters + 7], failed_entries_k2 = t->h_counters[kNumCounters + 3];  // (valid after a round trip)
    {
        // the kernel reads the region lengths (and the abandon flag) from device memory
        ProfScope ps(t, "merge_overflow_kernel");
        if (raw) {
            hipLaunchKernelGGL(kct::merge_overflow_kernel<1>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                               (const unsigned int *)d_ovf_count, (int)n_ovf1, ovf_cap, (const du64 *)d_overflow, mv, t->d_counters, k, (const du64 *)nullptr, pend);
            if (two_level)
                hipLaunchKernelGGL(kct::merge_overflow_kernel<1>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr2.p,
                                   (const unsigned int *)d_ovf2_count, (int)n_ovf2, ovf2_cap, (const du64 *)d_overflow, mv, t->d_counters, k, (const du64 *)nullptr, pend);
        } else {
            hipLaunchKernelGGL(kct::merge_overflow_kernel<0>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p,
                               (const unsigned int *)d_ovf_count, (int)n_ovf1, ovf_cap, (const du64 *)d_overflow, mv, t->d_counters, k);
            if (two_level)
                hipLaunchKernelGGL(kct::merge_overflow_kernel<0>, dim3(256), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr2.p,
                                   (const unsigned int *)d_ovf2_count, (int)n_ovf2, ovf2_cap, (const du64 *)d_overflow, mv, t->d_counters, k);
        }
    }
    HIP_TRY(hipGetLastError());
    u64 c2[4], spilled;
    KCT_TRY(read_counters(t, c2, &spilled));  // counting pass: K2's tallies and the merges' together; shadow pass: the merges' only
    const u64 nfailed = raw ? nfailed_k2 : t->h_counters[kNumCounters + 7], failed_entries = raw ? failed_entries_k2 : t->h_counters[kNumCounters + 3];
    KCT_DBG(t, "partitioned pass%s%s: npos=%llu blocks=%llu levels=%d region_cap=%u counted=%llu merged=%llu new=%llu spilled=%llu abandoned=%llu blocks / %llu entries abandon=%llu\n",
            raw ? " (shadow)" : "", probe ? " (probe)" : "", (unsigned long long)npos, (unsigned long long)B, two_level ? 2 : 1, region_cap,
            (unsigned long long)(raw ? c[kct::CTR_COUNTED] : c2[kct::CTR_COUNTED]), (unsigned long long)c2[kct::CTR_TOTAL_ADDED],
            (unsigned long long)(raw ? c[kct::CTR_NEWKEYS] : c2[kct::CTR_NEWKEYS]), (unsigned long long)spilled, (unsigned long long)nfailed,
            (unsigned long long)failed_entries, (unsigned long long)t->h_counters[kNumCounters + 6]);
    if (t->h_counters[kNumCounters + 6] != 0) return KCT_OK;  // abandoned: K2 and the merges exited early, nothing was touched
    *handled = true;
    {   // bursty input (consume_compact has the story): more than 2 % of the pass over the overflow route -> K1 flushes every 2 windows from now on
        const u64 over = c2[kct::CTR_TOTAL_ADDED], all = (raw ? c[kct::CTR_COUNTED] : c2[kct::CTR_COUNTED]) + over;
        if (all >= (1ULL << 20) && over * 50 > all) t->wide_bursty = true;
    }
    if (pend.pairs) t->pending_pairs = t->h_counters[kNumCounters + 8];
    if (!raw) {
        t->lazy_empty = false;
        *n_out += c2[kct::CTR_COUNTED] + c2[kct::CTR_TOTAL_ADDED];
        t->n_keys += c2[kct::CTR_NEWKEYS];
        if (spilled) {
            KCT_TRY(t->d_aux2.reserve(spilled * 16));
            HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
            KCT_TRY(replay_spill(t, spilled, n_out));
        }
        if (nfailed) {  // table blocks that overflowed: the table grows, their entries are counted with the direct insert
            u64 tl[4] = {0, 0, 0, 0};
            KCT_TRY(recount_failed(t, 0, aa.scratch, aa.seg_stride, aa.block_stride, aa.region_count, aa.nregions, nfailed, failed_entries, 0, tl));
            *n_out += tl[kct::CTR_TOTAL_ADDED];
        }
        return KCT_OK;
    }
    // ---- shadow pass: c = entries counted into the shadow / new shadow keys; c2 = what the merges put into the real table
    const u64 counted = c[kct::CTR_COUNTED], new_shadow = c[kct::CTR_NEWKEYS], blocked = failed_entries;
    t->shadow_empty = false;
    t->shadow_dirty = true;
    t->shadow_keys += new_shadow;
    // n counts every window whose k-mer went into the shadow: the (2^-64 per k-mer) case of a MurmurHash3 value of 0,
    // which the reference leaves out of n, is only seen when the shadow is flushed.
    *n_out += counted + c2[kct::CTR_TOTAL_ADDED];
    t->n_keys += c2[kct::CTR_NEWKEYS];
    u64 new_in_table = c2[kct::CTR_NEWKEYS];
    if (spilled) {
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, n_out));
    }
    if (nfailed) {  // shadow blocks that overflowed: their entries are hashed and counted into the real table
        u64 tl[4] = {0, 0, 0, 0};
        KCT_TRY(recount_failed(t, 1, aa.scratch, aa.seg_stride, aa.block_stride, aa.region_count, aa.nregions, nfailed, failed_entries, 0, tl));
        *n_out += tl[kct::CTR_TOTAL_ADDED];
        new_in_table += tl[kct::CTR_NEWKEYS];
    }
    DedupeOutcome o;
    o.new_keys = new_shadow + new_in_table;
    o.blocked = blocked;
    KCT_TRY(after_dedupe_pass(t, false, npos, o, probe));
    // the shadow holds as many keys as the table would: grow both (the table's growth re-creates the shadow, flushed)
    if (t->shadow && t->shadow_cap == t->cap && (double)t->shadow_keys > kMaxLoad * (double)t->shadow_cap) {
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
    // once per pass, so it wants passes of several windows per slot; its scratch costs
    // ~11-21 B per window start, which bounds the pass by HBM (this is what 288 GB is for).  Decided
    // once per call: buffers this table already holds are reused, so they count as available.
    u64 chunk_limit = kChunkPositions;
    int sub_chunks = 1;
    if (t->force_path != 1 && partition_geometry_ok(t) && (t->cap >> t->block_bits) > 1024) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const u64 held = t->d_scratch.cap + t->d_scratch2.cap + t->d_spill.cap + t->d_irr.cap + t->d_irr2.cap;
            double avail = (double)free_b + (double)held;
            // a shadow table this call may still have to allocate; per window: two levels of 4- or 8-byte entries with
            // ~20 % of slack, plus overflow regions
            const bool may_dedupe = t->k <= 32 && !t->dedupe_off && t->force_path != 2, may_compact = may_dedupe && t->k <= 21 && !t->compact_off;
            if (may_compact && !(t->shadow32 && t->s32_sbits == compact_sbits_for(t))) avail -= (double)(1ULL << (compact_sbits_for(t) + kct::kBlockBitsMax)) * 8.0;
            else if (may_dedupe && !may_compact && !(t->shadow && t->shadow_cap == t->cap)) avail -= (double)t->cap * 16.0;
            const double per_window = may_compact ? 11.0 : 21.0;
            const u64 by_mem = avail > 0 ? (u64)(avail * 0.8 / per_window) : 0;
            chunk_limit = std::max<u64>(kChunkPositions, std::min<u64>(16 * t->cap, by_mem));
            chunk_limit &= ~(u64)0xFFFF;  // keeps `d_stream + done` 16-byte aligned
            // The 64-bit paths can cut a pass into four sub-chunks that share K1's scratch (consume_partitioned): 12.5 B per
            // window start instead of 21 -- taken when that saves whole passes over the table.
            const int want_chunks = t->tune.sub_chunks ? t->tune.sub_chunks : 4;
            if (!may_compact && want_chunks > 1 && nbytes - k + 1 > chunk_limit) {
                const double per4 = 9.6 / want_chunks + 9.6 + 0.5;
                u64 limit4 = std::max<u64>(kChunkPositions, std::min<u64>(16 * t->cap, avail > 0 ? (u64)(avail * 0.8 / per4) : 0)) & ~(u64)0xFFFF;
                const u64 w = nbytes - k + 1;
                if ((w + limit4 - 1) / limit4 < (w + chunk_limit - 1) / chunk_limit) { chunk_limit = limit4; sub_chunks = want_chunks; }
            }
        }
    }
    // A large call into a table that knows nothing about its input (no keys, no hint from earlier passes): is this deep
    // coverage of few k-mers (dedupe-first pays) or mostly distinct ones?  The first 2^22 window starts go through the
    // dedupe-first path as a PROBE; the share of first sightings among them gives the number of distinct k-mers the
    // input draws from (as if uniformly -- position-sorted input shows its repeats even sooner), hence the windows per
    // distinct k-mer of the whole call.  Either way the probe's k-mers are counted; a wrong guess costs speed only.
    const u64 here = last_start + 1;
    const u64 call_windows = here + t->more_windows;  // (the early route feeds one job as several calls: the policy looks at the whole of it)
    KCT_DBG(t, "consume_stream: %llu window starts (+ %llu announced), chunk limit %llu, table %llu slots\n", (unsigned long long)here,
            (unsigned long long)t->more_windows, (unsigned long long)chunk_limit, (unsigned long long)t->cap);
    bool probe = probe_wanted(t, call_windows);
    if (!probe && mostly_new_expected(t, call_windows)) t->expect_new_keys = true;
    if (here > chunk_limit) {  // passes of equal size
        const u64 passes = (here + chunk_limit - 1) / chunk_limit;
        chunk_limit = std::min(chunk_limit, (((here + passes - 1) / passes) + 0xFFFF) & ~(u64)0xFFFF);
    }
    while (done <= last_start) {
        KCT_TRY(maybe_grow(t));
        // a chunk owns window starts [done, done + npos); its loads reach k-1 bytes further
        // A table that was never sized by its owner starts tiny: feed it launches of at most a few windows
        // per slot, so that what cannot be placed (and must be replayed after growing) stays small while
        // the table finds its size; launches grow with it.
        const u64 ramp = t->auto_sized ? std::max<u64>(1ULL << 20, 4 * t->cap) : ~0ULL;
        const u64 npos = std::min<u64>({probe ? kProbeWindows : chunk_limit, ramp, last_start + 1 - done});
        const u64 chunk_bytes = std::min<u64>(nbytes - done, npos + k - 1);
        t->call_windows_left = last_start + 1 - done + t->more_windows;
        if (probe) {
            probe = false;
            // A table of up to 1024 blocks has a small shadow anyway: the probe is an ordinary dedupe-first pass whose counts stay
            // pending.  A larger table's probe is a DRY RUN into a small shadow of its own, swapped in for this pass (nothing is
            // pending and no shadow holds a key -- that is when a probe is made -- so the table-sized shadows are simply set
            // aside): nothing reaches the real table, which may stay lazily empty, and the probe's windows are counted again with
            // the rest of the call.
            const bool use_compact = t->k <= 21 && !t->compact_off;
            const bool swap32 = use_compact && compact_sbits_for(t) != kCompactBlockBits, swap64 = !use_compact && t->cap > kProbeShadowSlots;
            const bool dry_run = swap32 || swap64;
            unsigned int *big32 = t->shadow32; const int big_sbits = t->s32_sbits; const unsigned int big_nbins = t->s32_nbins, big_bin0 = t->s32_bin0;
            du64 *big64 = t->shadow; const u64 big_cap = t->shadow_cap; const int big_bb = t->shadow_block_bits;
            if (swap32) { t->shadow32 = t->probe_shadow32; t->s32_sbits = kCompactBlockBits; t->s32_nbins = 1024; t->s32_bin0 = 0; t->s32_empty = true; }
            if (swap64) { t->shadow = t->probe_shadow; t->shadow_cap = t->shadow ? kProbeShadowSlots : 0; t->shadow_block_bits = kct::kBlockBitsMax; t->shadow_empty = true; }
            const u64 table_before = t->n_keys, n_before = *n_out;
            DedupeOutcome seen, seen1;
            bool handled = false;
            kct_status st = KCT_OK;
            // a dry run is made in two halves: the first sightings at two depths tell k-mers that never repeat (sequencing errors) from
            // deep coverage of few k-mers (path_policy.h per_distinct_two_depths)
            const u64 h1 = dry_run && npos >= (1ULL << 18) ? (npos / 2) & ~(u64)0xFFFF : 0;
            auto probe_piece = [&](u64 off, u64 n, DedupeOutcome *out) -> kct_status {
                const u64 bytes = std::min<u64>(nbytes - done - off, n + k - 1);
                handled = false;
                kct_status s_ = KCT_OK;
                if (use_compact) s_ = consume_compact(t, d_stream + done + off, bytes, n, n_out, &handled, true, dry_run ? out : nullptr);
                if (s_ == KCT_OK && !handled && !t->dedupe_off && (!dry_run || swap64))
                    s_ = consume_partitioned(t, d_stream + done + off, bytes, n, n_out, &handled, true, true, dry_run ? out : nullptr);
                return s_;
            };
            if (h1) {
                st = probe_piece(0, h1, &seen1);
                if (st == KCT_OK && handled) {
                    if (swap32) t->s32_empty = false;   // (the second half meets what the first one left in the probe's shadow)
                    if (swap64) t->shadow_empty = false;
                    st = probe_piece(h1, npos - h1, &seen);
                    if (!handled) seen1 = DedupeOutcome();
                }
            } else st = probe_piece(0, npos, &seen);
            if (swap32) { t->probe_shadow32 = t->shadow32; t->shadow32 = big32; t->s32_sbits = big_sbits; t->s32_nbins = big_nbins; t->s32_bin0 = big_bin0; t->s32_empty = true; }
            if (swap64) { t->probe_shadow = t->shadow; t->shadow = big64; t->shadow_cap = big_cap; t->shadow_block_bits = big_bb; t->shadow_empty = true; }
            KCT_TRY(st);
            if (handled) {
                u64 valid, fresh_keys;
                if (dry_run) { valid = std::max<u64>(1, seen.counted + seen.blocked + seen1.counted + seen1.blocked); fresh_keys = seen.new_keys + seen.blocked + seen1.new_keys + seen1.blocked; }
                else {
                    done += npos; t->windows_since_read += npos;
                    valid = std::max<u64>(1, *n_out - n_before);                                 // window starts that held a k-mer
                    fresh_keys = std::max(t->shadow_keys, t->s32_keys) + (t->n_keys - table_before);
                }
                const double x = draws_per_distinct((double)fresh_keys / (double)valid);         // the probe's k-mers per distinct k-mer
                double per_key = x * (double)call_windows / (double)npos;                        // ... the whole call's
                // Two depths tell never-repeating k-mers from deep coverage -- where the sample is deep enough for its curve to bend
                // (a fifth of the distinct k-mers seen; overlapping reads repeat k-mers in clusters, so a shallow sample's handful of
                // repeats says nothing about curvature: the north-star run's probe sees 0.7 % of its k-mers)
                if (h1 && seen1.counted + seen1.blocked && x >= 0.2) {
                    const double v1 = (double)(seen1.counted + seen1.blocked), f1 = (double)(seen1.new_keys + seen1.blocked);
                    per_key = per_distinct_two_depths(v1, f1, (double)valid, (double)fresh_keys, (double)valid * (double)call_windows / (double)npos);
                }
                const bool pays = probe_verdict(t, per_key, call_windows);
                KCT_DBG(t, "dedupe probe%s: %llu k-mers, %llu first sightings -> ~%.3g k-mers per distinct k-mer over the call: %s\n", dry_run ? " (dry run)" : "",
                        (unsigned long long)valid, (unsigned long long)fresh_keys, per_key, pays ? "dedupe-first" : "hash every window");
                // (a "no" is a verdict on THIS call: later calls are judged by what the table holds and how long the caller's
                // runs between reads are -- dedupe_pays -- so nothing is switched off)
                if (pays) t->dedupe_hint = true;
                else KCT_TRY(flush_shadow(t));
                t->expect_new_keys = per_key < 6.0;  // fewer than six k-mers per distinct one: the first pass is mostly first sightings
                continue;
            }
        }
        if (!t->runs_in.groups && dedupe128_pays(t, npos)) {
            bool handled = false;
            KCT_TRY(consume_raw128(t, d_stream + done, chunk_bytes, npos, n_out, &handled));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (compact_pays(t, npos)) {
            bool handled = false;
            KCT_TRY(consume_compact(t, d_stream + done, chunk_bytes, npos, n_out, &handled, false));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (dedupe_pays(t, npos)) {
            bool handled = false;
            KCT_TRY(consume_partitioned(t, d_stream + done, chunk_bytes, npos, n_out, &handled, true, false, nullptr, sub_chunks));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (partition_geometry_ok(t) && t->force_path != 1 && (t->force_path == 2 || partition_pays(t, npos))) {
            bool handled = false;
            KCT_TRY(consume_partitioned(t, d_stream + done, chunk_bytes, npos, n_out, &handled, false, false, nullptr, sub_chunks));
            if (handled) { done += npos; t->windows_since_read += npos; continue; }
        }
        if (t->runs_in.groups) {  // super-k-mer windows on the direct path (small passes): an ASCII image, k bytes + a separator per window
            const u64 piece = std::min<u64>(npos, 1ULL << 22), g0 = (u64)(d_stream + done - t->runs_base) >> 6;
            kct::RunsInput in = t->runs_in;
            in.groups += g0;
            const u64 ng = (piece + 63) >> 6, img = ng * 64 * (u64)(k + 1);
            KCT_TRY(t->d_unpack.reserve(img + 16));
            launch_expand_runs(t, in, ng, (unsigned char *)t->d_unpack.p);
            HIP_TRY(hipGetLastError());
            const kct::RunsInput keep = t->runs_in;
            const unsigned char *keep_base = t->runs_base;
            t->runs_in = kct::RunsInput(); t->runs_base = nullptr;
            const u64 left = t->call_windows_left, since = t->windows_since_read;
            u64 n_piece = 0;
            const kct_status st = consume_stream(t, (const unsigned char *)t->d_unpack.p, img, &n_piece);   // (the image is the table's own buffer: d_unpack is not used below)
            t->runs_in = keep; t->runs_base = keep_base;
            t->call_windows_left = left; t->windows_since_read = since + piece;
            KCT_TRY(st);
            *n_out += n_piece;
            done += piece;
            continue;
        }
        KCT_TRY(materialize(t));
        KCT_TRY(t->d_spill.reserve(npos * 16));
        KCT_TRY(zero_counters(t));
        const unsigned char *chunk = d_stream + done;
        if (t->packed_codes) {  // the direct kernel reads bytes: the chunk's ASCII image
            const u64 g0 = done >> 4, ng = (chunk_bytes + 15) >> 4;
            KCT_TRY(unpack_stream(t, t->packed_codes + g0, t->packed_valid + g0, ng));
            chunk = (const unsigned char *)t->d_unpack.p;
        }
        const int grid = (int)((npos + kct::kTile - 1) / kct::kTile);
        // the kernel derives window ownership from tile positions, so hand it a stream that ends
        // where this chunk's last window ends
        {
            ProfScope ps(t, "count_windows_kernel");
            dispatch_k<CountLauncher>(k, t->stream, grid, chunk, chunk_bytes, k, view(t, npos), t->d_counters);
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
    t->call_windows_left = 0;
    KCT_DBG(t, "consume_stream: done\n");
    if (t->auto_sized && t->cap == cap_at_entry && nbytes >= (1u << 20)) t->auto_sized = false;  // the table has found its size
    return KCT_OK;
}

// Counts received SUPER-K-MERS (partition_args.h RunsInput; the early route's owner side): 64 * ngroups virtual windows, whatever path
// the table's policy picks -- K1's RUNS instantiations walk the windows directly, the direct path gets an ASCII image.
kct_status consume_stream_runs(kct_table *t, const kct::RunsInput &in, u64 ngroups, u64 *n_out) {
    *n_out = 0;
    if (!ngroups) return KCT_OK;
    if (t->k > 64) { set_err("super-k-mer input needs k <= 64"); return KCT_ERR_ARG; }
    t->runs_in = in;
    t->runs_base = (const unsigned char *)(uintptr_t)0x200000000000ULL;  // an origin for window offsets; never dereferenced
    const kct_status st = consume_stream(t, t->runs_base, ngroups * 64 + t->k - 1, n_out);
    t->runs_in = kct::RunsInput(); t->runs_base = nullptr;
    return st;
}

// Counts a PACKED record stream (window_kernels.h pack_stream_kernel's format): the partition kernels read the groups directly;
// whatever reads bytes (the direct kernel; every kernel at k > 64) gets an unpacked image.
kct_status consume_stream_packed(kct_table *t, const unsigned int *d_codes, const unsigned short *d_valid, u64 nbases, u64 *n_out) {
    *n_out = 0;
    if (nbases < t->k) return KCT_OK;
    const u64 ng = (nbases + 15) >> 4;
    if (t->k > 64) {  // bytewise kernels only
        KCT_TRY(unpack_stream(t, d_codes, d_valid, ng));
        return consume_stream(t, (const unsigned char *)t->d_unpack.p, nbases, n_out);
    }
    t->packed_codes = d_codes; t->packed_valid = d_valid;
    t->packed_base = (const unsigned char *)(uintptr_t)0x100000000000ULL;  // a 16-byte aligned origin for offsets; never dereferenced
    const kct_status st = consume_stream(t, t->packed_base, nbases, n_out);
    t->packed_codes = nullptr; t->packed_valid = nullptr; t->packed_base = nullptr;
    return st;
}

}  // namespace kcth
