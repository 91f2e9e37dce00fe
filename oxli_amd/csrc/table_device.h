// table_device.h -- device-resident open-addressed {u64 hash -> u64 count} table in HBM.
//
// Mirrors the reference's `counts: HashMap<u64,u64>` (lib.rs:33) and `count_hash` (lib.rs:100-104).
//
// Layout ("block-SoA, 8-slot groups")
//   The table is `cap` slots (power of two) cut into aligned BLOCKS of S = min(cap, 8192) slots.
//   A block is 2*S u64 words: S keys, then S counts -- 128 KiB at full size, exactly what one
//   workgroup can hold in LDS.  Key 0 = EMPTY (consume never inserts hash 0, lib.rs:589; the
//   host keeps a side counter for count_hash(0)).
//   Inside a block, slots form GROUPS of 8: one group's keys are one aligned 64-byte line.
//   home group of h = (h & (cap-1)) >> 3   (MurmurHash3 output is already well mixed)
//   probe sequence  = home group, then the following groups, wrapping INSIDE the block;
//                     within a group slots are tried in order 0..7.
//   A key lives in the first slot of that sequence that was empty when it arrived, so a lookup
//   may stop at the first empty slot it meets.  One probe round = one 64-byte line = 8
//   candidates, in HBM for the direct path and in LDS for the partitioned path.
//   Because a key never leaves the block its home group names, one workgroup can own a block
//   outright: the partitioned path loads it into LDS, counts there, and stores it back with
//   plain coalesced traffic.
// Direct path: claim = 64-bit CAS on the key word, increment = 64-bit atomic add on the count
//   word; both are agent-scope HBM atomics (per-XCD L2s are not coherent, so they execute
//   memory-side).  Keys never change once written, so the plain loads of the key line in
//   front of the CAS are safe: a stale EMPTY only sends the lane to the CAS, which returns the
//   truth.
// A lane that cannot place its key within kMaxProbeGroups groups appends {hash, count} to the
// spill list; the host replays the list with whole-block probing (TableView::max_groups = 0) and grows
// the table when it is too full, so nothing is ever dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kct {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kMaxProbeGroups = 32;  // 256 slots
constexpr int kBlockBitsMax = 13;    // 8192 slots: 64 KiB keys + 64 KiB counts = one CU's LDS
constexpr int kGroupBits = 3;        // 8 slots = one 64-byte line of keys
constexpr int kGroup = 1 << kGroupBits;

struct TableGeom {
    u64 mask;        // cap - 1
    int block_bits;  // log2(S), >= kGroupBits
};

struct TableView {
    u64 *words;      // 2 * cap u64 words, block-SoA
    TableGeom g;
    u64 *spill;      // 2 * spill_cap words
    u64 spill_cap;
    u64 *spill_n;    // device counter
    int max_groups;  // groups table_add probes before it spills: kMaxProbeGroups on the hot paths, the whole block
                     // (0 = S / 8) for re-hash and spill replay, so that keys sharing their low bits still find room
};

// word index of the KEY of global slot s; its count sits S words further
__device__ __host__ __forceinline__ u64 key_word(const TableGeom &g, u64 s) {
    const u64 bmask = (1ULL << g.block_bits) - 1;
    return ((s >> g.block_bits) << (g.block_bits + 1)) + (s & bmask);
}
__device__ __host__ __forceinline__ u64 block_slots(const TableGeom &g) { return 1ULL << g.block_bits; }

// first slot of the home group / of the next group in the probe sequence
__device__ __forceinline__ u64 home_group_slot(const TableGeom &g, u64 h) { return (h & g.mask) & ~(u64)(kGroup - 1); }
__device__ __forceinline__ u64 next_group_slot(const TableGeom &g, u64 s) {
    const u64 bmask = (1ULL << g.block_bits) - 1;
    return (s & ~bmask) | ((s + kGroup) & bmask);
}

struct AddResult {
    u64 old;       // count before the add (only when WANT_OLD)
    bool claimed;  // this call created the key
    bool spilled;  // not placed: appended to the spill list, the caller must not tally it
    u64 word;      // word index of the key's slot (its count sits block_slots further); valid unless spilled
};

__device__ __forceinline__ void spill_pair(const TableView &t, u64 h, u64 c) {
    u64 i = atomicAdd(t.spill_n, 1ULL);
    if (i < t.spill_cap) { t.spill[2 * i] = h; t.spill[2 * i + 1] = c; }
}

// Adds `c` to the count of `h` (h != 0) with HBM atomics.
template <bool WANT_OLD>
__device__ __forceinline__ AddResult table_add(const TableView &t, u64 h, u64 c) {
    const u64 S = block_slots(t.g);
    u64 s = home_group_slot(t.g, h);
    AddResult r{0, false, false, 0};
    const int max_groups = t.max_groups > 0 ? t.max_groups : (int)(S >> kGroupBits);
    for (int probe = 0; probe < max_groups; ++probe) {
        u64 *kw = t.words + key_word(t.g, s);
        // one 64-byte line: the eight candidate keys of this group
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(kw), b = *reinterpret_cast<const ulonglong2 *>(kw + 2);
        const ulonglong2 d = *reinterpret_cast<const ulonglong2 *>(kw + 4), e = *reinterpret_cast<const ulonglong2 *>(kw + 6);
        const u64 k[kGroup] = {a.x, a.y, b.x, b.y, d.x, d.y, e.x, e.y};
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            u64 key = k[i];
            if (key == 0) {
                key = atomicCAS(kw + i, 0ULL, h);
                if (key == 0) { r.claimed = true; key = h; }
            }
            if (key == h) {
                if (WANT_OLD) r.old = atomicAdd(kw + i + S, c);
                else atomicAdd(kw + i + S, c);
                r.word = key_word(t.g, s) + (u64)i;
                return r;
            }
        }
        s = next_group_slot(t.g, s);
    }
    spill_pair(t, h, c);
    r.spilled = true;
    return r;
}

// Word index of the key of `h`, or ~0 if absent.  Only valid once all writers have completed
// (kernel boundary).
__device__ __forceinline__ u64 table_find(const u64 *words, const TableGeom &g, u64 h) {
    const u64 S = block_slots(g);
    u64 s = home_group_slot(g, h);
    for (u64 probe = 0; probe < (S >> kGroupBits); ++probe) {
        const u64 kw = key_word(g, s);
        for (int i = 0; i < kGroup; ++i) {
            const u64 key = words[kw + i];
            if (key == h) return kw + i;
            if (key == 0) return ~0ULL;
        }
        s = next_group_slot(g, s);
    }
    return ~0ULL;
}

}  // namespace kct
