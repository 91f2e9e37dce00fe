// analytics_kernels.h -- whole-table scans behind histo / min / max / mincut / maxcut / drop / set operations /
// jaccard / cosine (lib.rs:197-267, 464-514, 610-655, 708-765): reductions, filtered compaction and
// table-against-table lookups over the resident slots.  Cold paths next to consume, HBM-streaming bound.
#pragma once
#include "device_common.h"

namespace kct {

__device__ __forceinline__ u64 wave_min(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_down(v, off); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ u64 wave_max(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_down(v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// minmax[0] = min count (start at ~0), minmax[1] = max count (start at 0), *sumsq = sum of count^2 as f64
__global__ __launch_bounds__(kBlock) void count_stats_kernel(const u64 *__restrict__ words, TableGeom g, u64 *minmax, double *sumsq) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    u64 lo = ~0ULL, hi = 0;
    double sq = 0.0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s);
        if (words[kw] != 0) {
            const u64 c = words[kw + S];
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
            sq += (double)c * (double)c;
        }
    }
    lo = wave_min(lo); hi = wave_max(hi); sq = wave_sum_f64(sq);
    if ((threadIdx.x & 63) == 0) {
        if (lo != ~0ULL) atomicMin(minmax, lo);
        if (hi != 0) atomicMax(minmax + 1, hi);
        if (sq != 0.0) atomicAdd(sumsq, sq);
    }
}

// out[0] += sum over keys of hash * count, out[1] ^= xor over keys of hash * count, out[2] += sum of count^2 (all wrapping u64):
// an order-free digest of the table's contents (tests compare it with the CPU oracle's at sizes no dump would be compared at)
__device__ __forceinline__ u64 wave_xor(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v ^= __shfl_down(v, off);
    return v;
}
__global__ __launch_bounds__(kBlock) void digest_kernel(const u64 *__restrict__ words, TableGeom g, u64 *out) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    u64 shc = 0, xhc = 0, sq = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s), h = words[kw];
        if (h != 0) {
            const u64 c = words[kw + S];
            shc += h * c; xhc ^= h * c; sq += c * c;
        }
    }
    shc = wave_sum(shc); xhc = wave_xor(xhc); sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) {
        if (shc) atomicAdd(out, shc);
        if (xhc) atomicXor(out + 1, xhc);
        if (sq) atomicAdd(out + 2, sq);
    }
}

// the counts of all occupied slots, in no particular order; *out_n must be zero on entry
__global__ __launch_bounds__(kBlock) void compact_counts_kernel(const u64 *__restrict__ words, TableGeom g, u64 *__restrict__ out,
                                                                u64 out_cap, u64 *out_n) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s);
        if (words[kw] != 0) {
            const u64 i = atomicAdd(out_n, 1ULL);
            if (i < out_cap) out[i] = words[kw + S];
        }
    }
}

// interleaved {hash, count} pairs of the slots that survive a cut: lo <= count <= hi and hash != drop (0 = none)
__global__ __launch_bounds__(kBlock) void compact_filtered_kernel(const u64 *__restrict__ words, TableGeom g, u64 lo, u64 hi, u64 drop,
                                                                  u64 *__restrict__ out_pairs, u64 out_cap, u64 *out_n) {
    const u64 cap = g.mask + 1, S = block_slots(g);
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(g, s);
        const u64 key = words[kw];
        if (key != 0 && key != drop) {
            const u64 c = words[kw + S];
            if (c >= lo && c <= hi) {
                const u64 i = atomicAdd(out_n, 1ULL);
                if (i < out_cap) { out_pairs[2 * i] = key; out_pairs[2 * i + 1] = c; }
            }
        }
    }
}

// out[0] += keys of a that b also holds, out[1] += sum over those keys of count_a * count_b (wrapping u64, lib.rs:736-744)
__global__ __launch_bounds__(kBlock) void compare_tables_kernel(const u64 *__restrict__ a_words, TableGeom ga, const u64 *__restrict__ b_words,
                                                                TableGeom gb, u64 *out) {
    const u64 cap = ga.mask + 1, Sa = block_slots(ga), Sb = block_slots(gb);
    u64 common = 0, dot = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(ga, s);
        const u64 key = a_words[kw];
        if (key != 0) {
            const u64 w = table_find(b_words, gb, key);
            if (w != ~0ULL) { ++common; dot += a_words[kw + Sa] * b_words[w + Sb]; }
        }
    }
    common = wave_sum(common); dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0 && common) { atomicAdd(out, common); atomicAdd(out + 1, dot); }
}

// appends the keys of a that are in b (want = 1), not in b (want = 0) or all of them (want = 2; b unused).
// b_words == nullptr stands for an empty b.
__global__ __launch_bounds__(kBlock) void select_keys_kernel(const u64 *__restrict__ a_words, TableGeom ga, const u64 *__restrict__ b_words,
                                                             TableGeom gb, int want, u64 *__restrict__ out, u64 out_cap, u64 *out_n) {
    const u64 cap = ga.mask + 1;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 key = a_words[key_word(ga, s)];
        if (key == 0) continue;
        bool take = want == 2;
        if (want != 2) {
            const bool found = b_words != nullptr && table_find(b_words, gb, key) != ~0ULL;
            take = found == (want == 1);
        }
        if (take) {
            const u64 i = atomicAdd(out_n, 1ULL);
            if (i < out_cap) out[i] = key;
        }
    }
}

// drop_hash (lib.rs:213-224) in place: one thread finds the key and closes the gap by backward shifting.  Probing is
// linear from the home group's first slot and never leaves the block, so only the run of occupied slots behind the
// key can be affected: a later key moves into the hole iff the hole still lies at or after its home in probe order.
__global__ void remove_hash_kernel(u64 *words, TableGeom g, u64 h, u64 *found) {
    const u64 w = table_find(words, g, h);
    *found = w != ~0ULL;
    if (w == ~0ULL) return;
    const u64 S = block_slots(g), smask = S - 1, base = w & ~(2 * S - 1);  // first word of the key's block
    u64 i = w - base, j = i;
    for (u64 n = 1; n < S; ++n) {
        j = (j + 1) & smask;
        const u64 kj = words[base + j];
        if (kj == 0) break;
        const u64 home = kj & g.mask & smask & ~(u64)(kGroup - 1);
        if (((j - home) & smask) >= ((j - i) & smask)) {
            words[base + i] = kj;
            words[base + S + i] = words[base + S + j];
            i = j;
        }
    }
    words[base + i] = 0;
    words[base + S + i] = 0;
}

}  // namespace kct
