// path_policy.h -- which device path counts a pass, and the sizing rules of the partitioned paths: PURE HOST LOGIC.
//
// No HIP, no allocation, no I/O: every function reads a few plain members of its table argument (k, cap, block_bits, the
// path switches and what the table has learnt about its input) and returns a decision.  The functions are templates over the
// table type so that libkct_hip.so instantiates them with kct_table and the CPU test harness (tests/policy_harness.cpp,
// tests/test_policy_cpu.py) with a plain struct of the same member names -- the selection logic is driven with fake
// geometries without a GPU.  What the decisions mean is described where they are used (kct_consume.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace kcth {

typedef uint64_t pu64;

constexpr int kPolicyBlockBitsMax = 13;   // = kct::kBlockBitsMax (static_assert in kct_consume.hip)
constexpr int kPolicyPartTile = 16384;    // = kct::kPartTile
constexpr int kPolicyRingEntries = 16384; // = kct::kRingEntries
constexpr double kPolicyMaxLoad = 0.65;   // = kMaxLoad
constexpr int kCompactBlockBits = 10;     // the compact shadow beside a table of up to 1024 blocks: 1024 blocks
constexpr pu64 kProbeWindows = 1ULL << 22;
constexpr pu64 kProbeShadowSlots = 1ULL << (10 + kPolicyBlockBitsMax);

inline int policy_log2(pu64 v) { int b = 0; while ((2ULL << b) <= v) ++b; return b; }

// The partitioned path pays 16 B (one level) or 32 B (two levels) of streaming scratch traffic per
// k-mer plus 32 B per table slot per pass; the direct path pays one memory-side atomic per k-mer.
// It wins once a pass brings a fair fraction as many windows as the table has slots.
template <class T>
bool partition_geometry_ok(const T *t) {
    const pu64 nblocks = t->cap >> t->block_bits;
    return nblocks >= 16 && t->cap <= (1ULL << 33);  // two levels of 1024 bins x 8192 slots: 128 GiB of table
}

template <class T>
bool partition_pays(const T *t, pu64 npos) {
    const pu64 nblocks = t->cap >> t->block_bits;
    if (!partition_geometry_ok(t) || npos < (1ULL << 20)) return false;
    return nblocks <= 1024 ? npos >= t->cap / 4 : npos >= t->cap / 2;
}

inline unsigned int region_capacity(double avg) {
    return (unsigned int)((((pu64)(avg * 1.15 + 8.0 * std::sqrt(avg) + 64.0)) + 7) & ~7ULL);
}

// How a table of 2^bbits blocks is reached: K1 (or the first level of a pair flush) fans out to 2^pbits bins through its
// LDS ring -- at most 1024 --, each holding 2^sub_bits table blocks that the second level (repartition_kernel) separates.
// The second level wants >= 64 bins per super-bin: with 16 its lanes fight over a handful of LDS cursors (2.5x slower per
// entry), so small two-level tables give the first level FEWER bins, and W workgroups share a super-bin so that the second
// level still fills the chip.
struct Levels {
    int bbits, pbits, sub_bits;
    bool two;
    pu64 P, B, W;
};

inline Levels levels_for(int bbits, int nwg, int forced_pbits = -1) {
    Levels L;
    L.bbits = bbits;
    L.two = bbits > 10;
    L.pbits = bbits;
    if (L.two) L.pbits = bbits <= 14 ? bbits - 6 : std::min(10, bbits - 7);
    if (forced_pbits >= 0 && L.two) L.pbits = std::max(bbits - 10, std::min(10, forced_pbits));  // (Tuning::pbits: measurement only)
    L.sub_bits = bbits - L.pbits;
    L.P = 1ULL << L.pbits;
    L.B = 1ULL << bbits;
    L.W = L.two ? std::max<pu64>(1, (pu64)nwg / L.P) : 1;
    return L;
}

// second partition level: lines of a bin that leave the ring together -- 2 or 4 when the ring is deep enough (its depth in
// lines per bin >= 4x that), so that the scattered 64-byte stores become 128- or 256-byte ones (K1b: 1 -> 2 lines -13 %,
// 2 -> 4 lines another -4 %)
template <class T>
unsigned int repartition_min_lines(const T *t, int ring_entries, int sub_bits, int entry_bytes) {
    if (t->tune.k1b_lines) return (unsigned int)t->tune.k1b_lines;  // (measurement only)
    const int lines_per_bin = (ring_entries >> sub_bits) * entry_bytes / 64;
    return lines_per_bin >= 16 ? 4u : lines_per_bin >= 8 ? 2u : 1u;
}

// overflow regions: an eighth of a workgroup's entries, but few enough that ring positions (21 bits in ring_flush's line
// list) cannot wrap before a hopelessly skewed pass is abandoned
inline unsigned int overflow_capacity(pu64 entries_per_wg) { return (unsigned int)std::min<pu64>(1ULL << 20, std::max<pu64>(4096, entries_per_wg / 8)); }

// windows a dedupe-first pass must amortise per k-mer it leaves pending: a flush costs ~0.04-0.05 ns per pending k-mer when
// the pairs are partitioned (~0.11 ns with one random table access each), a dedupe-first pass saves ~3-5 ps per window
template <class T>
pu64 windows_per_pending_key(const T *t) { return partition_geometry_ok(t) && t->block_bits == kPolicyBlockBitsMax ? 16 : 32; }

// The compact shadow (k <= 21): 2^sbits blocks x 8192 slots of u32 key + u32 count.  A block index is the TOP sbits bits
// of the 42-bit mix42 value, an entry its low 32 bits, so sbits >= 10.  Beside a table of up to 1024 blocks it is the
// fixed 1024-block (64 MiB) one and K1's bins are its blocks; beside a larger table it has as many blocks as the table and K1's
// 1024 bins are super-bins -- or, for shadows of fewer than 2^16 blocks, GROUPS of 2^(16 - sbits) of K1's bins are, so that the
// second partition level still has 64 bins per super-bin (RepartitionArgs::gbits).  (Until round 4 such a shadow was rounded up to
// 2^16 blocks -- 4 GiB beside a 2^24-slot table: K2 and the conversion then visit mostly empty blocks.)
template <class T>
int compact_sbits_for(const T *t) {
    const int bbits = policy_log2(t->cap >> t->block_bits);
    return bbits <= 10 ? kCompactBlockBits : bbits;
}
// workgroups per second-level super-bin: one, unless there are too few super-bins to fill the chip (a writer's sixteen waves take one
// input region each: no more writers than regions / 16)
inline pu64 two_level_writers(pu64 nsuper, int nwg) { return std::max<pu64>(1, std::min<pu64>((pu64)nwg / std::max<pu64>(nsuper, 1), 16)); }
inline int compact_group_bits(int sbits) { return sbits > kCompactBlockBits && sbits < 16 ? 16 - sbits : 0; }

// A dedupe-first run also pays for its SHADOW, whatever the input: the first K2 pass stores every shadow block and the
// conversion reads (and re-zeroes) every slot -- ~2.5 bytes of streaming per shadow byte at ~4.5 TB/s -- while a dedupe-first
// pass saves ~3.75 ps per window over hashing it (K1 0.92 -> 0.35 ms per 1.5x10^8 windows): ~0.15 windows per shadow byte must
// be consumed between two reads of the table.  Negligible for a table of up to 1024 blocks (64 MiB of compact shadow: 10^7
// windows); decisive for a small two-level table, whose compact shadow is 4 GiB at least (C2-sized reads with 1 % substitution
// errors into a 2^26-slot table: 7.2 ms dedupe-first against 2.4 ms hashing every window, tools/err_probe.py).
template <class T>
pu64 shadow_bytes_for(const T *t, bool compact) {  // the shadow a dedupe-first run of this table would use
    return compact ? (1ULL << (compact_sbits_for(t) + kPolicyBlockBitsMax)) * 8 : t->cap * 16;
}
template <class T>
bool shadow_amortises(const T *t, bool compact, pu64 windows) { return (double)windows >= 0.15 * (double)shadow_bytes_for(t, compact); }

// (everything but the shadow's own cost, which differs between the variants)
template <class T>
bool dedupe_pays_but_for_the_shadow(const T *t, pu64 npos) {
    if (t->k > 32 || t->dedupe_off || npos < (1ULL << 22) || !partition_geometry_ok(t)) return false;
    if (t->force_path == 3) return true;
    if (t->force_path != 0 || !partition_pays(t, npos)) return false;  // the shadow mirrors the table's geometry
    // few distinct k-mers, each many times?  What the table (or the shadow) holds so far is the best guess.
    const pu64 known = std::max({(pu64)t->n_keys, (pu64)t->shadow_keys, (pu64)t->s32_keys});
    if (known == 0) return t->dedupe_hint;  // nothing counted yet (new or cleared table): go by how the last pass went
    // Converting pays once ~16 (32) windows have been counted per distinct k-mer between two reads of the table.  The
    // caller's run so far is the evidence that reads are that rare -- and nothing can read before the running call ends.
    return known * windows_per_pending_key(t) <= t->windows_since_read + std::max<pu64>(npos, t->call_windows_left);
}

template <class T>
bool dedupe_pays(const T *t, pu64 npos) {  // the 64-bit variant: a table-sized shadow
    if (!dedupe_pays_but_for_the_shadow(t, npos)) return false;
    return t->force_path == 3 || t->shadow_dirty || shadow_amortises(t, false, t->windows_since_read + std::max<pu64>(npos, t->call_windows_left));
}

// Compact dedupe-first pass (k <= 21): K1 MODE 2 writes 32-bit entries, aggregate_blocks32_kernel counts them
// into the compact shadow (u32 keys, u32 counts).  Half the partition traffic of the 64-bit variant and half as many
// ring flushes.  With a shadow of more than 1024 blocks a second partition level (repartition_kernel<u32>) sits between.
template <class T>
bool compact_pays(const T *t, pu64 npos) {
    if (t->k > 21 || t->compact_off || !dedupe_pays_but_for_the_shadow(t, npos)) return false;
    if (t->force_path != 3 && !t->s32_dirty && !shadow_amortises(t, true, t->windows_since_read + std::max<pu64>(npos, t->call_windows_left))) return false;
    if (compact_sbits_for(t) > kCompactBlockBits) return true;  // a shadow as large as the table
    const pu64 known = std::max<pu64>(t->n_keys, t->s32_keys);
    return known <= (pu64)((double)(1ULL << (kCompactBlockBits + kPolicyBlockBitsMax)) * 0.6);
}

// The 128-bit variant (33 <= k <= 64): a fixed shadow of 1024 x 4096 slots holds ~2.5 M k-mers at most, and its conversion inserts
// with atomics (~0.1 ns per pending k-mer): for deep coverage of SMALL genomes only -- 32 windows per known k-mer between reads.
// Its 16-byte entries flush K1's ring every second window and double K2's stream, so it only beats hashing every window where the
// hash is long: measured at C2 size on a 2 Mbp genome (tools/k51_probe.py) k = 33: 7.9 against 8.6x10^10 k-mers/s hashing, k = 51:
// 7.6 against 6.6, k = 64: 7.3 against 5.5; the hashing K1 has since become 8 % faster (rotl / tail tables: 7.2 at k = 51) -- chosen by
// itself from k = 48 on, forced (set_path 3) at any k in 33..64.
constexpr pu64 kShadow128Keys = (pu64)(0.6 * 1024 * 4096);
// Round 4: OFF in the automatic choice (its gain over the hashing K1, which has since got 8 % faster, is 3 % on its showcase -- bench.py
// k51_deep 1.03x -- for a second shadow layout and a conversion by atomics); kct_set_path(t, 3) still takes it at any k in 33..64.
constexpr int kDedupe128MinK = 65;
template <class T>
bool dedupe128_pays(const T *t, pu64 npos) {
    if (t->k <= 32 || t->k > 64 || t->dedupe128_off || npos < (1ULL << 22)) return false;
    if (t->force_path == 3) return true;
    if (t->force_path != 0 || t->k < kDedupe128MinK) return false;
    const pu64 known = std::max<pu64>(t->n_keys, t->s128_keys);
    if (known == 0) return t->dedupe_hint;
    return known <= kShadow128Keys && known * 32 <= t->windows_since_read + std::max<pu64>(npos, t->call_windows_left);
}

// A large call into a table that knows nothing about its input (no keys, no hint from earlier passes): is this deep
// coverage of few k-mers (dedupe-first pays) or mostly distinct ones?  The first 2^22 window starts go through the
// dedupe-first kernels as a PROBE -- unless no shadow this call could pay for exists anyway.
template <class T>
bool probe_wanted(const T *t, pu64 call_windows) {
    return t->force_path == 0 && t->k <= 32 && !t->dedupe_off && !t->dedupe_hint && !t->auto_sized &&
           std::max({(pu64)t->n_keys, (pu64)t->shadow_keys, (pu64)t->s32_keys}) == 0 && call_windows >= 8 * kProbeWindows &&
           call_windows >= 4 * (pu64)t->cap &&   // (a table its owner sized holds its k-mers at a load of 0.3-0.6: fewer than 4 window starts
                                                  // per slot are fewer than ~10 per distinct k-mer -- hashing, no need to look)
           t->cap >= kProbeShadowSlots && partition_geometry_ok(t) && partition_pays(t, call_windows) &&
           ((t->k <= 21 && !t->compact_off && shadow_amortises(t, true, t->windows_since_read + call_windows)) ||
            shadow_amortises(t, false, t->windows_since_read + call_windows));
}

// No probe because the call is small for the table its owner sized (above): then it is mostly first sightings -- fewer than 2.5
// window starts per slot are fewer than ~6 per distinct k-mer -- and the hashing K2 of an empty table runs the variant whose fast path
// claims slots itself (what a probe's "fewer than six per distinct k-mer" sets too).
template <class T>
bool mostly_new_expected(const T *t, pu64 call_windows) {
    return !t->auto_sized && std::max({(pu64)t->n_keys, (pu64)t->shadow_keys, (pu64)t->s32_keys}) == 0 && 2 * call_windows < 5 * (pu64)t->cap;
}

// From the share r of a uniform sample's n draws that were first sightings: x = n / D solves r = (1 - e^-x) / x.
inline double draws_per_distinct(double r) {
    if (r >= 0.9995) return 0.0;   // (nearly) every draw new: D is beyond what the sample can see
    if (r <= 0.0) return 1e9;
    double lo = 1e-6, hi = 1e6;    // (1 - e^-x) / x falls monotonically from 1 to 0
    for (int i = 0; i < 80; ++i) {
        const double mid = std::sqrt(lo * hi);
        if ((1.0 - std::exp(-mid)) / mid > r) lo = mid; else hi = mid;
    }
    return lo;
}

// The same from TWO depths of one sample -- f1 first sightings among the first n1 draws, f2 among all n2 -- under a model with one
// more unknown: a share e of the draws are k-mers that never repeat (sequencing errors: a substituted base makes up to k new k-mers,
// each seen once), the rest are uniform draws from D:  F(n) = e n + D (1 - exp(-(1 - e) n / D)).  Returns the k-mers per distinct
// k-mer the model predicts for N draws, N / F(N).  With e = 0 this is draws_per_distinct's law; the one-depth estimate reads 1 % of
// substitution errors at 30x coverage as "17 per distinct k-mer" where the truth is 4.5 (round 4: bench.py C2_sub1pct).
inline double per_distinct_two_depths(double n1, double f1, double n2, double f2, double N) {
    if (n2 <= 0 || f2 <= 0 || n1 <= 0 || n1 >= n2) return 1.0;
    f1 = std::min(f1, n1); f2 = std::min(f2, n2);
    // for a given e the second depth gives D (F is increasing in D); the first depth then picks e: F(n1; e, D(e)) falls as e grows
    // (a straighter curve), so bisect on e
    auto model = [](double n, double e, double D) { return e * n + D * (1.0 - std::exp(-(1.0 - e) * n / D)); };
    auto solve_D = [&](double e) {
        const double target = f2 - e * n2;               // D (1 - exp(-(1 - e) n2 / D)) = target, at most (1 - e) n2
        if (target >= (1.0 - e) * n2 * 0.99995) return 1e30;   // every uniform draw was new: D is beyond the sample
        if (target <= 0) return 1e-9;
        double lo = 1.0, hi = 1e18;
        for (int i = 0; i < 100; ++i) {
            const double mid = std::sqrt(lo * hi);
            if (mid * (1.0 - std::exp(-(1.0 - e) * n2 / mid)) < target) lo = mid; else hi = mid;
        }
        return lo;
    };
    double elo = 0.0, ehi = std::min(0.999, f2 / n2);
    for (int i = 0; i < 50; ++i) {
        const double e = 0.5 * (elo + ehi);
        if (model(n1, e, solve_D(e)) > f1) elo = e; else ehi = e;
    }
    const double e = 0.5 * (elo + ehi), D = solve_D(e);
    const double FN = e * N + (D >= 1e29 ? (1.0 - e) * N : D * (1.0 - std::exp(-(1.0 - e) * N / D)));
    return FN > 0 ? N / FN : 1e9;
}

// the probe's verdict: k-mers per distinct k-mer over the whole call (per_key), and a shadow the call can pay for
template <class T>
bool probe_verdict(const T *t, double per_key, pu64 call_windows) {
    const bool use_compact = t->k <= 21 && !t->compact_off;
    return per_key >= (double)windows_per_pending_key(t) &&
           (shadow_amortises(t, use_compact, t->windows_since_read + call_windows) || shadow_amortises(t, false, t->windows_since_read + call_windows));
}

// The path of a pass of npos window starts, in the order consume_stream tries them (a path that abandons a pass hands it to
// the next): 3 = compact dedupe-first, 2 = 64-bit dedupe-first, 1 = partitioned (hash every window), 0 = direct atomic kernel.
enum PassPath { PASS_DIRECT = 0, PASS_PARTITIONED = 1, PASS_DEDUPE64 = 2, PASS_COMPACT = 3, PASS_DEDUPE128 = 4 };
template <class T>
PassPath choose_path(const T *t, pu64 npos) {
    if (dedupe128_pays(t, npos)) return PASS_DEDUPE128;
    if (compact_pays(t, npos)) return PASS_COMPACT;
    if (dedupe_pays(t, npos)) return PASS_DEDUPE64;
    if (partition_geometry_ok(t) && t->force_path != 1 && (t->force_path == 2 || partition_pays(t, npos))) return PASS_PARTITIONED;
    return PASS_DIRECT;
}

}  // namespace kcth
