// table_kernels.h -- kernels over (hash, count) pairs and whole tables: pair merge (add(), spill replay),
// re-hash, lookups, compaction, owner-bucketed export, sums.
#pragma once
#include "device_common.h"

namespace kct {

// ---- (hash, count) pairs -> table: add()'s inner loop (lib.rs:798-806), spill replay, re-hash --------
// pairs are read as keys[i * key_stride], counts[i * count_stride] so that the same kernel folds
// separate arrays (stride 1) and an old slot array (stride 2, keys = slots, counts = slots + 1).
// n_dev (if not null) holds the pair count in device memory, clamped to n; a non-zero *abort
// (if not null) turns the launch into a no-op.
__global__ __launch_bounds__(kBlock) void merge_pairs_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ counts,
                                                             u64 n, const u64 *n_dev, const u64 *abort, int stride, TableView table,
                                                             u64 *counters) {
    __shared__ u64 s_tot, s_new, s_zero;
    if (abort && *abort) return;
    if (n_dev) { const u64 nd = *n_dev; n = nd < n ? nd : n; }
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; s_zero = 0; }
    __syncthreads();
    u64 tot = 0, nk = 0, nz = 0;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        const u64 h = keys[i * stride];
        if (h == 0) continue;
        const u64 c = counts[i * stride];
        const AddResult r = table_add<true>(table, h, c);
        if (!r.spilled) {
            tot += c;
            nk += r.claimed;
            nz += (r.old == 0);  // lib.rs:801-803: a key counts as new when its current count is 0
        }
    }
    tot = wave_sum(tot); nk = wave_sum(nk); nz = wave_sum(nz);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); atomicAdd(&s_zero, nz); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
        if (s_zero) atomicAdd(shard + CTR_NEW_BY_ZERO, s_zero);
    }
}

// ---- re-hash: every occupied slot of an old table -> the new table (growth) ------------------------
__global__ __launch_bounds__(kBlock) void rehash_kernel(const u64 *__restrict__ old_words, TableGeom old_g, TableView table,
                                                        u64 *counters) {
    __shared__ u64 s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const u64 cap = old_g.mask + 1, S = block_slots(old_g);
    u64 nk = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(old_g, s);
        const u64 h = old_words[kw];
        if (h == 0) continue;
        const AddResult r = table_add<false>(table, h, old_words[kw + S]);
        nk += (r.claimed && !r.spilled) ? 1 : 0;
    }
    nk = wave_sum(nk);
    if ((threadIdx.x & 63) == 0 && nk) atomicAdd(&s_new, nk);
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(counters + (blockIdx.x % kCounterShards) * kCounterStride + CTR_NEWKEYS, s_new);
}

// ---- lookups / point update ---------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void get_hashes_kernel(const u64 *__restrict__ words, TableGeom g,
                                                            const u64 *__restrict__ hashes, u64 n, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const u64 w = hashes[i] ? table_find(words, g, hashes[i]) : ~0ULL;
    out[i] = w == ~0ULL ? 0 : words[w + block_slots(g)];
}

// set the count of an existing key (returns 1 in *found) -- __setitem__ (lib.rs:675-681)
__global__ void set_hash_kernel(u64 *words, TableGeom g, u64 h, u64 value, u64 *found) {
    const u64 w = table_find(words, g, h);
    *found = w != ~0ULL;
    if (w != ~0ULL) words[w + block_slots(g)] = value;
}

// ---- whole-table scans ---------------------------------------------------------------------------
// compaction for dump / export: out_n must be zero on entry
__global__ __launch_bounds__(kBlock) void compact_kernel(const u64 *__restrict__ words, TableGeom g, u64 *__restrict__ out_keys,
                                                         u64 *__restrict__ out_counts, u64 out_cap, u64 *out_n) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s);
        const u64 key = words[kw];
        if (key != 0) {
            const u64 i = atomicAdd(out_n, 1ULL);  // hipcc folds this into one add per wave
            if (i < out_cap) { out_keys[i] = key; out_counts[i] = words[kw + S]; }
        }
    }
}

// ---- export bucketed by owner rank (multi-GPU merge) -------------------------------------------------
// owner(h) = floor(hi32(h) * nparts / 2^32): a contiguous slice of hash space per rank.
__device__ __forceinline__ u32 owner_of(u64 h, u32 nparts) { return (u32)(((h >> 32) * (u64)nparts) >> 32); }

constexpr int kMaxParts = 256;

// pass 1: how many occupied slots belong to each owner
__global__ __launch_bounds__(kBlock) void count_owners_kernel(const u64 *__restrict__ words, TableGeom g, u32 nparts, u64 *part_counts) {
    __shared__ u32 hist[kMaxParts];
    for (u32 i = threadIdx.x; i < nparts; i += kBlock) hist[i] = 0;
    __syncthreads();
    const u64 cap = g.mask + 1;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 key = words[key_word(g, s)];
        if (key != 0) atomicAdd(&hist[owner_of(key, nparts)], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nparts; i += kBlock) if (hist[i]) atomicAdd(part_counts + i, (u64)hist[i]);
}

// pass 2: write interleaved {hash, count} pairs, owner p's pairs contiguous from part_base[p].
// Each workgroup reserves one range per owner per chunk of slots, so the global cursors see
// (chunks x nparts) atomics instead of one per key.
__global__ __launch_bounds__(kBlock) void scatter_owners_kernel(const u64 *__restrict__ words, TableGeom g, u32 nparts,
                                                                u64 *part_cursor /* starts at part_base */, u64 *__restrict__ out_pairs,
                                                                u64 out_cap) {
    __shared__ u32 hist[kMaxParts];
    __shared__ u64 base[kMaxParts];
    const u64 cap = g.mask + 1, S = block_slots(g);
    constexpr u64 kChunkSlots = 16 * kBlock;
    for (u64 c0 = (u64)blockIdx.x * kChunkSlots; c0 < cap; c0 += (u64)gridDim.x * kChunkSlots) {
        for (u32 i = threadIdx.x; i < nparts; i += kBlock) hist[i] = 0;
        __syncthreads();
        u64 keys[16], cnts[16];
        u32 rank[16], own[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u64 s = c0 + (u64)j * kBlock + threadIdx.x;
            keys[j] = 0;
            if (s < cap) {
                const u64 kw = key_word(g, s);
                keys[j] = words[kw];
                if (keys[j]) { cnts[j] = words[kw + S]; own[j] = owner_of(keys[j], nparts); rank[j] = atomicAdd(&hist[own[j]], 1u); }
            }
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < nparts; i += kBlock) base[i] = hist[i] ? atomicAdd(part_cursor + i, (u64)hist[i]) : 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (keys[j]) {
                const u64 pos = base[own[j]] + rank[j];
                if (pos < out_cap) { out_pairs[2 * pos] = keys[j]; out_pairs[2 * pos + 1] = cnts[j]; }
            }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void sum_counts_kernel(const u64 *__restrict__ words, TableGeom g, u64 *out) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    u64 acc = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s);
        if (words[kw] != 0) acc += words[kw + S];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace kct
