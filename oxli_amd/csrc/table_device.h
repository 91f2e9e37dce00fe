// table_device.h -- device-resident open-addressed {u64 hash -> u64 count} table in HBM.
//
// Mirrors the reference's `counts: HashMap<u64,u64>` (lib.rs:33) and `count_hash` (lib.rs:100-104):
//   slot s = 16 bytes {key, count} at slots[2s], slots[2s+1]; key 0 = EMPTY (consume never
//   inserts hash 0, lib.rs:589; the host keeps a side counter for count_hash(0)).
//   home slot = hash & mask (MurmurHash3 output is already well mixed).
//   Linear probing that WRAPS INSIDE AN ALIGNED BLOCK of 2^block_bits slots (8192 slots = 128 KiB,
//   or the whole table when it is smaller).  A key therefore always lives in the block its home
//   slot names, which lets one workgroup own a block outright: the partitioned path loads the
//   block into LDS, counts there, and stores it back with plain coalesced traffic.
//   The direct path claims with a 64-bit CAS on the key word and increments with a 64-bit
//   atomic add on the count word; both are agent-scope HBM atomics (per-XCD L2s are not
//   coherent, so they execute memory-side).  Keys never change once written, so the plain key
//   load in front of the CAS is safe: a stale EMPTY only sends the lane to the CAS, which
//   returns the truth.
// A lane that cannot place its key within kMaxProbe slots appends {hash, count} to the spill
// list; the host grows the table and replays the list, so nothing is ever dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kct {

typedef unsigned long long u64;

constexpr int kMaxProbe = 128;
constexpr int kBlockBitsMax = 13;  // 8192 slots x 16 B = 128 KiB: one block fits a CU's LDS

struct TableView {
    u64 *slots;       // 2 * (mask + 1) words
    u64 mask;         // capacity - 1 (capacity is a power of two)
    u64 block_mask;   // slots per probing block - 1
    u64 *spill;       // 2 * spill_cap words
    u64 spill_cap;
    u64 *spill_n;     // device counter
};

__device__ __forceinline__ u64 next_slot(u64 s, u64 block_mask) { return (s & ~block_mask) | ((s + 1) & block_mask); }

struct AddResult {
    u64 old;       // count before the add (only when WANT_OLD)
    bool claimed;  // this call created the key
    bool spilled;  // not placed: appended to the spill list, the caller must not tally it
};

__device__ __forceinline__ void spill_pair(const TableView &t, u64 h, u64 c) {
    u64 i = atomicAdd(t.spill_n, 1ULL);
    if (i < t.spill_cap) { t.spill[2 * i] = h; t.spill[2 * i + 1] = c; }
}

// Adds `c` to the count of `h` (h != 0).
template <bool WANT_OLD>
__device__ __forceinline__ AddResult table_add(const TableView &t, u64 h, u64 c) {
    u64 s = h & t.mask;
    AddResult r{0, false, false};
    for (int probe = 0; probe < kMaxProbe; ++probe) {
        u64 *slot = t.slots + 2 * s;
        u64 key = *slot;
        if (key == 0) {
            key = atomicCAS(slot, 0ULL, h);
            if (key == 0) { r.claimed = true; key = h; }
        }
        if (key == h) {
            if (WANT_OLD) r.old = atomicAdd(slot + 1, c);
            else atomicAdd(slot + 1, c);
            return r;
        }
        s = next_slot(s, t.block_mask);
    }
    spill_pair(t, h, c);
    r.spilled = true;
    return r;
}

// Count of `h`, 0 if absent.  Only valid after all writers have completed (kernel boundary).
__device__ __forceinline__ u64 table_get(const u64 *slots, u64 mask, u64 block_mask, u64 h) {
    u64 s = h & mask;
    for (u64 probe = 0; probe <= block_mask; ++probe) {
        u64 key = slots[2 * s];
        if (key == h) return slots[2 * s + 1];
        if (key == 0) return 0;
        s = next_slot(s, block_mask);
    }
    return 0;
}

}  // namespace kct
