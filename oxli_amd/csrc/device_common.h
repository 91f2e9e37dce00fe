// device_common.h -- constants and wave helpers shared by every kernel header (gfx950, wave64).
#pragma once
#include "kmer_device.h"
#include "table_device.h"

namespace kct {

constexpr int kBlock = 256;
constexpr int kWPT = 32;                  // windows per thread (direct kernels)
constexpr int kTile = kBlock * kWPT;      // window start positions per workgroup (8192)
constexpr int kHaloMax = 256;             // k - 1 <= 254
constexpr int kPartThreads = 1024;        // partitioned path: one 16-wave workgroup per CU
constexpr int kPartWPT = 16;
constexpr int kPartTile = kPartThreads * kPartWPT;  // 16384 window starts per tile
constexpr int kRingEntries = 16384;       // LDS write-combining ring: 128 KiB of u64, split over the bins
constexpr int kChunk = 8;                 // entries per flush = one 64-byte line
constexpr int kWaveQueue = 160;           // K2: deferred entries per wave
constexpr int kCounterShards = 64;        // per-launch tallies are spread over this many 128-B lines
constexpr int kCounterStride = 16;        // u64 words per shard (128 B)
enum { CTR_COUNTED = 0, CTR_NEWKEYS = 1, CTR_TOTAL_ADDED = 2, CTR_NEW_BY_ZERO = 3 };

// 64-bit value of lane `src` (wave-uniform index) broadcast through SGPRs: two v_readlane_b32,
// no LDS round trip (what __shfl would cost).
__device__ __forceinline__ u64 read_lane64(u64 v, int src) {
    const u32 lo = __builtin_amdgcn_readlane((u32)v, src), hi = __builtin_amdgcn_readlane((u32)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid in lane 0
}

}  // namespace kct
