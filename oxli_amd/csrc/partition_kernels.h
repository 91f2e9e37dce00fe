// partition_kernels.h -- the partitioned counting path: K1 partition_windows_kernel, K1b
// repartition_kernel, K2 aggregate_blocks_kernel and merge_overflow_kernel.
#pragma once
#include <type_traits>

#include "window_kernels.h"
#include "partition_args.h"
#include "k1_kernel.h"

namespace kct {

// =================================================================================================
// Partitioned path: the same consume loop, without one HBM atomic per k-mer.
//
// The direct kernel (window_kernels.h) is bound by the memory-side atomic rate (~2.3e10 64-byte atomic
// requests/s measured on MI355X, profiles/r01a_*), not by HBM bandwidth.  Because probing is
// confined to 128-KiB table blocks (table_device.h), a block can be owned by ONE workgroup:
//
//   K1 partition_windows_kernel  hash every window (same arithmetic as above) and radix-partition
//        the 8-byte hashes by table block into HBM scratch.  Each persistent 1024-thread
//        workgroup keeps a software write-combining ring per block in LDS (128 KiB in all) and
//        flushes 64-byte lines to its own private region of every block's scratch, so the
//        scatter costs no global atomics and every store is a whole line.
//   K2 aggregate_blocks_kernel   one workgroup per table block: load the block (128 KiB) into
//        LDS -- or start from zeros if the table is known empty -- stream the block's hashes,
//        count them with LDS atomics (ds_cmpst_rtn_b64 claim + ds_add_u64), store the block back.
//
// HBM traffic per k-mer: 1.15 B bases + 8 B scratch write + 8 B scratch read, plus 32 B per table
// slot per pass -- all of it coalesced streaming.  A hash that finds its ring slot or its region
// full (many lanes hitting one block at once: homopolymers, tandem repeats) goes to a per-workgroup
// overflow region that merge_overflow_kernel folds in afterwards with the direct atomic insert,
// so results are identical.
// =================================================================================================

// ---- second partition level (tables with more than 1024 blocks) ---------------------------------------
// K1 can only fan out to 1024 bins (the LDS ring).  For larger tables its bins are SUPER-BINS of
// 2^sub_bits consecutive table blocks, and this kernel -- one 1024-thread workgroup per super-bin --
// re-partitions a super-bin's hashes by block with the same LDS write-combining ring.  Exactly one
// workgroup writes a given block's region, so K2 then reads a single region per block.

// T = u64 (MurmurHash3 / mix64 values; sub-bin = value bits block_bits...), u32 (compact dedupe-first entries: the low 32
// bits of a mix42 value whose top 10 bits chose the super-bin; sub-bin = the entry's top sub_bits bits) or ulonglong2
// ({hash, count} pairs of a shadow flush / pair merge; sub-bin from the hash).
// The NEXT slab's loads are issued before the current slab is appended (two register buffers): the flush barriers keep
// the sixteen waves in step, so without the prefetch every HBM round trip was fully exposed.
template <class T, bool WHOLE_SLAB = false>  // WHOLE_SLAB: one flush per slab instead of two (half the ring per interval)
__global__ __launch_bounds__(kPartThreads) void repartition_kernel(RepartitionArgs a) {
    constexpr bool kPair = sizeof(T) == 16, kCompact = sizeof(T) == 4;
    constexpr int kEntries = kRingEntries * 8 / sizeof(T);  // 128 KiB of ring
    constexpr int kLoads = 64 / sizeof(T);                   // 64 bytes per lane per slab
    constexpr int kFlushEvery = WHOLE_SLAB ? kLoads : kLoads / 2;  // appends per thread between flushes: a quarter (half) of the ring
    constexpr u32 kSlab = 64 * kLoads;
    __shared__ __attribute__((aligned(16))) T ring[kEntries];
    __shared__ u64 cur[1024];  // per bin: fill (low half) | flushed (high half)
    __shared__ u32 flist[2048];
    __shared__ u32 fcount, ovf_n, rounds;
    const int W = a.writers, s = blockIdx.x / W, w = blockIdx.x % W, P2 = 1 << a.sub_bits;
    const u32 D = (u32)(kEntries >> a.sub_bits), dmask = D - 1;
    const int dshift = __builtin_ctz((unsigned)kEntries) - a.sub_bits;  // log2 D
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T zero;
    memset(&zero, 0, sizeof zero);
    for (int i = threadIdx.x; i < kEntries; i += kPartThreads) ring[i] = zero;
    for (int i = threadIdx.x; i < 1024; i += kPartThreads) cur[i] = 0;
    if (threadIdx.x == 0) { fcount = 0; ovf_n = 0; rounds = 0; }
    if (threadIdx.x == 0 && a.overflow && *a.overflow) rounds = ~0u;  // K1 (or another super-bin) gave up on this pass
    __syncthreads();
    if (rounds == ~0u) return;  // read through LDS so that the whole workgroup takes the same branch
    __syncthreads();
    const int Wt = a.wtot ? a.wtot : W, wslot = a.writer0 + w;  // (a block's regions: wtot of them, this workgroup writes slot wslot)
    T *my_out = reinterpret_cast<T *>(a.out) + (((u64)s << a.sub_bits) * Wt + wslot) * a.out_cap;
    const u64 bin_stride = (u64)Wt * a.out_cap;
    u64 *my_ovf = a.ovf + (u64)blockIdx.x * a.ovf_cap;
    const int gbits = kCompact ? a.gbits : 0, lowbits = a.sub_bits - gbits, per_bin = a.nseg >> gbits;  // (regions per first-level bin)
    const u64 ovf_hi = kCompact ? (((u64)(((u32)s << gbits) + a.bin0) << 32) | (1ULL << 63)) : 0ULL;
    auto overflow_one = [&](u64 h) {
        const u32 i = atomicAdd(&ovf_n, 1u);
        if (i < a.ovf_cap) my_ovf[i] = h;
        else *a.overflow = 1ULL;
    };
    auto overflow_pair = [&](u64 h, u64 c) {
        const u64 i = atomicAdd(a.ovf_n, 1ULL);
        if (i < a.ovf_cap) { a.ovf[2 * i] = h; a.ovf[2 * i + 1] = c; }  // (cap = every pair: cannot be exceeded)
    };
    auto flush_lines = [&](bool drain) {
        if constexpr (kPair) return ring_flush<2048u, T>(ring, cur, flist, &fcount, P2, D, my_out, a.out_cap, drain, overflow_pair, bin_stride, 0ULL, a.min_lines);
        else return ring_flush<2048u, T>(ring, cur, flist, &fcount, P2, D, my_out, a.out_cap, drain, overflow_one, bin_stride, ovf_hi, a.min_lines, (u32)lowbits);
    };
    // A wave owns the input regions seg = 16 w + wave, + 16 W, ...; work unit = a slab of 64 lanes x 64 bytes.  Every wave
    // runs the same number of rounds so that the flush barriers line up.
    // (grouped super-bins: region `seg` is workgroup seg >> gbits of first-level bin (s << gbits) + (seg & (2^gbits - 1)) -- neighbouring
    // regions, which the sixteen waves read at the same time, belong to DIFFERENT bins, so that the appends of a moment spread over all
    // the sub-bins; taken bin by bin they would all land in a quarter (or less) of the ring and overflow it)
    const u32 *counts0 = a.in_count + (u64)s * a.nseg;
    auto count_of = [&](int seg) -> u32 { return gbits ? counts0[(u64)(seg & ((1 << gbits) - 1)) * per_bin + (seg >> gbits)] : counts0[seg]; };
    const int seg0 = w * (kPartThreads / 64) + wave, seg_step = W * (kPartThreads / 64);
    // The sizes of this wave's regions (its r-th region is seg0 + r seg_step) sit in ONE register, lane r holding region r's: a slab's
    // loads then start from a v_readlane instead of behind a global load of the size -- a round trip that every wave of the workgroup
    // paid at the same moment (the flush barriers keep them in step), once per slab.  (More than 64 regions per wave: loaded as before.)
    const int nreg = seg0 < a.nseg ? (a.nseg - seg0 + seg_step - 1) / seg_step : 0;
    const bool in_lanes = nreg <= 64;
    u32 lane_cnt = 0;
    if (in_lanes && lane < nreg) lane_cnt = count_of(seg0 + lane * seg_step);
    auto region_count = [&](int r) -> u32 {   // r: wave-uniform
        return in_lanes ? (u32)__builtin_amdgcn_readlane((int)lane_cnt, __builtin_amdgcn_readfirstlane(r)) : count_of(seg0 + r * seg_step);
    };
    u32 my_slabs = 0;
    if (in_lanes) {
        u32 v = (lane_cnt + kSlab - 1) / kSlab;
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        my_slabs = v;
    } else for (int r = 0; r < nreg; ++r) my_slabs += (region_count(r) + kSlab - 1) / kSlab;
    if (lane == 0) atomicMax(&rounds, my_slabs);
    __syncthreads();
    const u32 nrounds = rounds;
    int reg = 0;
    u32 off = 0;
    auto is_set = [](const T &v) -> bool {
        if constexpr (kPair) return v.x != 0;
        else return v != 0;
    };
    auto load_slab = [&](T (&v)[kLoads], u32 &binlow) {  // binlow: which of the super-bin's first-level bins the slab's region belongs to
#pragma unroll
        for (int j = 0; j < kLoads; ++j) v[j] = zero;
        u32 cnt = reg < nreg ? region_count(reg) : 0u;
        while (reg < nreg && off >= cnt) { ++reg; off = 0; cnt = reg < nreg ? region_count(reg) : 0u; }  // next non-empty region
        if (reg < nreg) {
            const int seg = seg0 + reg * seg_step;
            binlow = gbits ? (u32)seg & ((1u << gbits) - 1u) : 0u;
            const u64 region = gbits ? ((u64)(seg >> gbits) * a.nbins + ((u64)s << gbits) + binlow) : ((u64)seg * a.nbins + s);
            const T *src = reinterpret_cast<const T *>(a.in) + (a.in_off ? a.in_off[(u64)s * a.nseg + seg] : region * a.in_cap) + off;
            const u32 left = cnt - off;
            if constexpr (!kPair) {
                // SIXTEEN bytes per lane and load (kVec entries: regions are whole 64-byte lines, so a piece is never cut): a slab is the
                // same 64 bytes per lane in a quarter (u32) or half (u64) of the load instructions -- K1b is bound by the requests a CU keeps
                // in flight, and a request of 1 KiB per wave carries what four of 256 bytes did.  (Which lane appends which entry is immaterial.)
                constexpr int kVec = 16 / sizeof(T);
                if (a.in_off == nullptr) {
#pragma unroll
                    for (int p = 0; p < kLoads / kVec; ++p) {
                        const u32 i = kVec * (lane + 64 * p);
                        if (i < left) {
                            const uint4 w = *reinterpret_cast<const uint4 *>(src + i);
                            if constexpr (kCompact) { v[kVec * p] = w.x; v[kVec * p + 1] = w.y; v[kVec * p + 2] = w.z; v[kVec * p + 3] = w.w; }
                            else { v[kVec * p] = ((u64)w.y << 32) | w.x; v[kVec * p + 1] = ((u64)w.w << 32) | w.z; }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kLoads; ++j) { const u32 i = lane + 64 * j; if (i < left) v[j] = src[i]; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < kLoads; ++j) { const u32 i = lane + 64 * j; if (i < left) v[j] = src[i]; }
            }
            off += kSlab;
        }
    };
    T va[kLoads], vb[kLoads];
    u32 bla = 0, blb = 0;
    if (nrounds) load_slab(va, bla);
    for (u32 r = 0; r < nrounds; ++r) {
        if (r + 1 < nrounds) load_slab(vb, blb);
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const T e = va[j];
            if (is_set(e)) {
                u32 b;
                if constexpr (kPair) b = (u32)(e.x >> a.block_bits) & (u32)(P2 - 1);
                else if constexpr (kCompact) b = (bla << lowbits) | (lowbits ? e >> (32 - lowbits) : 0u);
                else b = (u32)(e >> a.block_bits) & (u32)(P2 - 1);
                const u64 cw = atomicAdd(&cur[b], 1ULL);
                const u32 pos = (u32)cw;
                if (pos - (u32)(cw >> 32) < D) ring[(b << dshift) + (pos & dmask)] = e;
                else if constexpr (kPair) overflow_pair(e.x, e.y);
                else if constexpr (kCompact) overflow_one((ovf_hi + ((u64)bla << 32)) | e);
                else overflow_one(e);
            }
            if ((j % kFlushEvery) == kFlushEvery - 1) flush_lines(false);
        }
#pragma unroll
        for (int j = 0; j < kLoads; ++j) va[j] = vb[j];
        bla = blb;
    }
    while (flush_lines(true)) {}
    for (int b = threadIdx.x; b < P2; b += kPartThreads) {
        const u32 f = (u32)(cur[b] >> 32);
        a.out_count[(((u64)s << a.sub_bits) + b) * Wt + wslot] = f < a.out_cap ? f : a.out_cap;
    }
    if constexpr (!kPair) {
        __syncthreads();
        if (threadIdx.x == 0) a.ovf_count[blockIdx.x] = ovf_n < a.ovf_cap ? ovf_n : a.ovf_cap;
    }
}

// A block whose keys do not fit (the table, or the shadow, is too small for what a pass brings) is ABANDONED by its K2
// workgroup: nothing of the pass is stored for it -- its HBM image stays as it was (zeros for a lazily empty table) -- and
// its number goes on this list.  Its entries still sit in its scratch regions; after the pass the host makes room and
// recount_failed_kernel counts them with the direct insert.  So K2 needs no per-window spill list: a list that had to
// be sized for the worst case (16 B per window start of the pass) and was practically never used.


template <bool CLAIM, bool TWO>  // CLAIM: the fast path claims free home-group slots itself (passes of mostly new k-mers); TWO: two partition levels
__global__ __launch_bounds__(kPartThreads) void aggregate_blocks_kernel(AggregateArgs a) {
    __shared__ __attribute__((aligned(16))) u64 tab[2 << kBlockBitsMax];  // S keys then S counts = 128 KiB
    __shared__ u64 wq[(kPartThreads / 64) * kWaveQueue];                  // per-wave queues of deferred entries, 20 KiB
    __shared__ __attribute__((aligned(16))) unsigned char tags[1 << kBlockBitsMax];  // one fingerprint byte per slot, 8 KiB
    __shared__ u64 s_counted, s_new;
    __shared__ u32 s_failed;
    if (*a.overflow) return;  // wave-uniform: K1 gave up, the host reruns the batch on the direct path
    const u32 S = 1u << a.block_bits, smask = S - 1;
    u64 *keys = tab, *cnts = tab + S;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kInFlight = 12;
    constexpr u32 kSlab = 64 * kInFlight;
    constexpr int kWaves = kPartThreads / 64;
    u64 sum_counted = 0, sum_new = 0;  // thread 0: the tallies of the blocks this workgroup has stored
    // A workgroup takes blocks b, b + grid, ...: with two partition levels the grid is one workgroup per CU, so that a
    // block's stores drain under the next block's load instead of in front of the next workgroup's start.
    for (u32 b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
    u64 *gblock = a.words + ((u64)b << (a.block_bits + 1));
    if (threadIdx.x == 0) { s_counted = 0; s_new = 0; s_failed = 0; }
    const u32 *my_counts = a.region_count + (u64)b * a.nregions;
    // Two levels: a few long regions per block, their slabs dealt round-robin to the waves.  The wave's NEXT slab is loaded
    // before the current one is counted, and its FIRST before the block itself is loaded, so that no HBM round trip of the
    // entry stream is exposed (one workgroup per CU: nothing else would hide it).
    int seg = 0;
    u32 sbase = 0, g = (u32)wave, cnt = a.nregions > 0 ? my_counts[0] : 0u;  // sbase = global index of seg's first slab
    auto load_slab = [&](u64 (&v)[kInFlight]) -> bool {
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) v[j] = 0ULL;
        while (seg < a.nregions) {
            const u32 nsl = (cnt + kSlab - 1) / kSlab;
            if (g < sbase + nsl) break;
            sbase += nsl; ++seg;
            cnt = seg < a.nregions ? my_counts[seg] : 0u;
        }
        if (seg >= a.nregions) return false;
        const u32 s0 = (g - sbase) * kSlab, left = cnt - s0;
        const u64 *src = a.scratch + (u64)seg * a.seg_stride + (u64)b * a.block_stride + s0;
        // (sixteen bytes -- two entries -- per lane and load: regions are whole 64-byte lines)
#pragma unroll
        for (int p = 0; p < kInFlight / 2; ++p) {
            const u32 i = 2 * (lane + 64 * p);
            if (i < left) { const uint4 w = *reinterpret_cast<const uint4 *>(src + i); v[2 * p] = ((u64)w.y << 32) | w.x; v[2 * p + 1] = ((u64)w.w << 32) | w.z; }
        }
        g += kWaves;
        return true;
    };
    const bool two_level = TWO && !(a.ablate & 64);
    u64 va[kInFlight], vb[kInFlight];
    bool more = two_level ? load_slab(va) : false;
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    // fingerprint of a key: a hash byte that neither the slot index (low bits) nor the multi-GPU
    // owner (top bits) uses; 0 is reserved for "empty slot"
    auto tag_of = [](u64 h) -> u32 { const u32 t = (u32)(h >> 32) & 0xFFu; return t ? t : 1u; };
    if (a.fresh) {
        for (u32 i = threadIdx.x; i < S; i += kPartThreads) t4[i] = make_uint4(0, 0, 0, 0);
        for (u32 i = threadIdx.x; i < S / 16; i += kPartThreads) reinterpret_cast<uint4 *>(tags)[i] = make_uint4(0, 0, 0, 0);
    } else {
        for (u32 i = threadIdx.x; i < S; i += kPartThreads) t4[i] = reinterpret_cast<const uint4 *>(gblock)[i];
        __syncthreads();
        for (u32 i = threadIdx.x; i < S; i += kPartThreads) { const u64 kk = keys[i]; tags[i] = (unsigned char)(kk ? tag_of(kk) : 0u); }
    }
    __syncthreads();
    u32 counted = 0, newkeys = 0;
    // General insert.  One probe round = the 8 keys of a group = one 64-byte LDS line
    // (4 x ds_read_b128), examined in slot order so a new key lands in the first empty slot of
    // its sequence -- the same arrangement the direct path builds.
    auto insert = [&](u64 h) {
        u32 g = (u32)h & smask & ~(u32)(kGroup - 1);
        bool placed = false;
        for (u32 round = 0; round < (S >> kGroupBits) && !placed; ++round) {
            const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(keys + g);
            const ulonglong2 q0 = kp[0], q1 = kp[1], q2 = kp[2], q3 = kp[3];
            const u64 k[kGroup] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
            int sel = kGroup;  // first slot that holds h or is empty
#pragma unroll
            for (int i = kGroup - 1; i >= 0; --i) if (k[i] == h || k[i] == 0) sel = i;
            while (sel < kGroup) {
                u64 ks = keys[g + sel];
                if (ks == 0) {
                    ks = atomicCAS(&keys[g + sel], 0ULL, h);
                    if (ks == 0) { ++newkeys; ks = h; tags[g + sel] = (unsigned char)tag_of(h); }
                }
                if (ks == h) {
                    if (!(a.ablate & 4)) atomicAdd(&cnts[g + sel], 1ULL);
                    placed = true;
                    break;
                }
                ++sel;  // another lane claimed that slot for a different key: try the following slots
                while (sel < kGroup) { const u64 kk = keys[g + sel]; if (kk == h || kk == 0) break; ++sel; }
            }
            g = (g + kGroup) & smask;
        }
        if (placed) ++counted;
        else s_failed = 1u;  // block full: the whole block is abandoned (FailedBlocks)
    };
    // Fast path: ~93 % of a block's entries are repeat sightings of a key that already sits in
    // its home group.  Those cost one 8-byte read of the group's fingerprints, a SWAR byte match,
    // one 8-byte key read to confirm and one ds_add_u64 -- 24 bytes of LDS traffic, no loops.
    // Everything else (first sightings, keys pushed to a later group, fingerprint collisions) is
    // parked in a per-wave queue and run through the general insert 64 at a time, so the branchy
    // code always runs with full lanes.  A fingerprint written late only costs a detour through
    // the queue; the general insert never trusts fingerprints.
    u64 *myq = wq + wave * kWaveQueue;
    u32 qn = 0;  // wave-uniform
    auto drain = [&](u32 keep_below) {
        while (qn > keep_below) {
            const u32 take = qn < 64 ? qn : 64;
            qn -= take;
            if ((u32)lane < take) insert(myq[qn + lane]);
        }
    };
    auto fast = [&](u64 h) {
        bool miss = h != 0;
        if (h != 0) {
            const u32 g = (u32)h & smask & ~(u32)(kGroup - 1);
            const u64 t8 = *reinterpret_cast<const u64 *>(tags + g);
            const u64 x = t8 ^ ((u64)tag_of(h) * 0x0101010101010101ULL);
            const u64 z = (x - 0x0101010101010101ULL) & ~x & 0x8080808080808080ULL;  // lowest set bit marks the first equal byte
            if (z) {
                const u32 idx = (u32)__builtin_ctzll(z) >> 3;
                if (keys[g + idx] == h) {
                    if (!(a.ablate & 4)) atomicAdd(&cnts[g + idx], 1ULL);
                    ++counted;
                    miss = false;
                }
            } else if constexpr (CLAIM) {
                // No fingerprint matches and the home group has a free slot: a first sighting, claimed right here (the
                // group's first free slot is where the general insert would put it) instead of taking the detour through the
                // queue.  Only for passes the host expects to bring mostly NEW k-mers (low coverage into an empty table: K2
                // -17 % on C4's shard): a branch one lane in ten takes is a branch nearly every wave takes, and where
                // first sightings are rare it costs more than the queue (K2 +12 % in the steady state -- even behind a run-time
                // flag that is off, hence the template parameter).
                const u64 e8 = (t8 - 0x0101010101010101ULL) & ~t8 & 0x8080808080808080ULL;  // zero bytes: free slots
                if (e8) {
                    const u32 idx = (u32)__builtin_ctzll(e8) >> 3;
                    const u64 old = atomicCAS(&keys[g + idx], 0ULL, h);
                    if (old == 0 || old == h) {
                        if (old == 0) { ++newkeys; tags[g + idx] = (unsigned char)tag_of(h); }
                        if (!(a.ablate & 4)) atomicAdd(&cnts[g + idx], 1ULL);
                        ++counted;
                        miss = false;
                    }
                }
            }
        }
        const u64 m = __ballot(miss);
        if (m) {
            const u32 pos = qn + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            if (miss) myq[pos] = h;
            qn += (u32)__popcll(m);
            if (qn > kWaveQueue - 64) drain(31);
        }
    };
    // The block's hashes arrive as `nregions` regions; they are cut into slabs of 64 x kInFlight
    // entries that the 16 waves take round-robin.  All of a lane's loads for a slab are issued before
    // the first insert, so ~12 x 512 B are in flight per wave instead of one dependent load.
    auto do_slab = [&](const u64 *src, u32 left) {
        u64 v[kInFlight];
#pragma unroll
        for (int p = 0; p < kInFlight / 2; ++p) {   // (sixteen bytes -- two entries -- per lane and load)
            const u32 i = 2 * (lane + 64 * p);
            uint4 w = make_uint4(0, 0, 0, 0);
            if (i < left) w = *reinterpret_cast<const uint4 *>(src + i);
            v[2 * p] = ((u64)w.y << 32) | w.x; v[2 * p + 1] = ((u64)w.w << 32) | w.z;
        }
        if (a.ablate & 16) {
#pragma unroll
            for (int j = 0; j < kInFlight; ++j) counted += (u32)v[j];
        } else {
#pragma unroll
            for (int j = 0; j < kInFlight; ++j) fast(v[j]);
        }
    };
    if (a.ablate & 64) {
    } else if constexpr (!TWO) {  // one level: many short regions, one wave each
        for (int seg = wave; seg < a.nregions; seg += kWaves) {
            const u32 cnt = my_counts[seg];
            const u64 *region = a.scratch + (u64)seg * a.seg_stride + (u64)b * a.block_stride;
            for (u32 s0 = 0; s0 < cnt; s0 += kSlab) do_slab(region + s0, cnt - s0);
        }
    } else {  // two levels (the first slab is already in registers)
        auto count_slab = [&](const u64 (&v)[kInFlight]) {
            if (a.ablate & 16) {
#pragma unroll
                for (int j = 0; j < kInFlight; ++j) counted += (u32)v[j];
            } else {
#pragma unroll
                for (int j = 0; j < kInFlight; ++j) fast(v[j]);
            }
        };
        while (more) {
            more = load_slab(vb);
            count_slab(va);
            if (!more) break;
            more = load_slab(va);
            count_slab(vb);
        }
    }
    drain(0);
    u64 wc = wave_sum(counted), wn = wave_sum(newkeys);
    if (lane == 0) { atomicAdd(&s_counted, wc); atomicAdd(&s_new, wn); }
    __syncthreads();
    if (s_failed) {  // (read after the barrier: uniform)
        if (a.fresh) for (u32 i = threadIdx.x; i < S; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) {
            u64 entries = 0;
            for (int r = 0; r < a.nregions; ++r) entries += my_counts[r];
            a.failed.list[atomicAdd(a.failed.n, 1ULL)] = b;
            atomicAdd(a.failed.entries, entries);
        }
    } else {
        for (u32 i = threadIdx.x; i < S; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = t4[i];
        if (threadIdx.x == 0) { sum_counted += s_counted; sum_new += s_new; }
    }
    __syncthreads();  // the block has left LDS (its stores may still be in flight) and the tallies are read
    }
    if (threadIdx.x == 0) {
        u64 *shard = a.counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (sum_counted) atomicAdd(shard + CTR_COUNTED, sum_counted);
        if (sum_new) atomicAdd(shard + CTR_NEWKEYS, sum_new);
    }
}

// mix64 value of a packed canonical k-mer (dedupe-first path) -> its MurmurHash3 value
template <int MODE = 1>
__device__ __forceinline__ u64 hash_of_mixed(u64 m, int k, const u32 *lut) {
    Packed<1> p;
    p.w[0] = MODE == 2 ? unmix42(m & kMask42) : unmix64(m) - 1ULL;  // (compact values travel with bit 63 set)
    left_align(p, k);
    return hash_packed<1, true>(p, k, lut);
}

// ---- dedupe-first path: the shadow table (packed k-mers -> pending counts) into the real table --------------------
// The shadow table has the real table's block-SoA layout, keyed by mix64 values.  Every slot with a pending count is
// turned back into its k-mer, hashed once, and its count added to the real table with the direct insert; the count
// is then zeroed, the key stays (it will be met again).  counters: CTR_NEWKEYS / CTR_TOTAL_ADDED as merge_pairs.
__global__ __launch_bounds__(kBlock) void shadow_flush_kernel(u64 *__restrict__ shadow, TableGeom sg, TableView main, int k, u64 *counters) {
    __shared__ u32 ascii4[256];
    __shared__ u64 s_tot, s_new;
    fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; }
    __syncthreads();
    const u64 cap = sg.mask + 1, S = block_slots(sg);
    u64 tot = 0, nk = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (u64)gridDim.x * kBlock) {
        const u64 kw = key_word(sg, s);
        const u64 c = shadow[kw + S];
        if (c == 0) continue;
        shadow[kw + S] = 0;
        const u64 h = hash_of_mixed(shadow[kw], k, ascii4);
        if (h == 0) continue;  // lib.rs:589: hash 0 is skipped
        const AddResult r = table_add<false>(main, h, c);
        if (!r.spilled) { tot += c; nk += r.claimed; }  // a spilled pair is tallied when it is replayed
    }
    tot = wave_sum(tot); nk = wave_sum(nk);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

// ---- compact dedupe-first path (k <= 21): 32-bit entries, a shadow table of u32 keys and u32 counts ----------------
// A mix42 value is 10 bits of bin and 32 bits of entry; within a bin the entry identifies the k-mer.  K1 (MODE 2)
// wrote the entries; this is K2 for them: same block ownership, fingerprint fast path and deferred queue as
// aggregate_blocks_kernel, at half the LDS and HBM bytes.  Slot = entry bits 0-12; no fingerprints (see the fast path).
// Counts are u32.  No entry reaches this kernel without passing an LDS ring, and a ring bin lets at most D of the 8192 (K1)
// or 8192 (K1b) appends of a flush interval through (the rest take the overflow route to the real table's u64 counts): a
// k-mer's pending count grows by at most 1/256 of the window starts consumed.  The host converts before 2^39 of them.

// TWO workgroups share a CU: 74 KiB of LDS each, and __launch_bounds__(1024, 8) holds the kernel to 64 VGPRs (it compiled to
// 87, which silently left every CU with one workgroup; at 64 a few values spill, and K2-32 is still 13 % faster -- the fast
// path waits on LDS round trips, which a second workgroup's waves fill).
constexpr int kWaveQueue32 = kWaveQueue;
template <bool TWO>  // TWO: a few long regions per block (two partition levels); otherwise one short region per K1 workgroup
__global__ __launch_bounds__(kPartThreads, 8) void aggregate_blocks32_kernel(Aggregate32Args a) {
    __shared__ __attribute__((aligned(16))) u32 tab[2 << kBlockBitsMax];  // S keys then S counts = 64 KiB
    __shared__ u32 wq[(kPartThreads / 64) * kWaveQueue32];
    __shared__ u64 s_counted, s_new;
    __shared__ u32 s_failed;
    if (*a.overflow) return;
    const u32 S = 1u << a.block_bits, smask = S - 1;
    u32 *keys = tab, *cnts = tab + S;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        constexpr int kInFlight = 8;  // (12 and 16 spill more at 64 VGPRs: K2-32 +2 % / +6 %; 4: +2 %)
    constexpr u32 kSlab = 64 * kInFlight;
    constexpr int kWaves = kPartThreads / 64;
    u64 sum_counted = 0, sum_new = 0;  // thread 0: the tallies of the blocks this workgroup has stored
    // (a workgroup takes blocks b, b + grid, ...: two levels run two workgroups per CU over all the blocks, as the 64-bit K2 does)
    for (u32 b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
    u32 *gblock = a.words + ((u64)b << (a.block_bits + 1));
    if (threadIdx.x == 0) { s_counted = 0; s_new = 0; s_failed = 0; }
    const u32 *my_counts = a.region_count + (u64)b * a.nregions;
    // Two levels (large shadow): a few long regions per block (one per K1b writer); their slabs of 768 entries are dealt
    // round-robin to the waves, the next slab in flight while the current one is counted -- and the FIRST requested before
    // the block itself is loaded (one workgroup per CU: nothing else would hide that round trip).
    int seg2 = 0;
    u32 sbase = 0, g2 = (u32)wave, cnt2 = a.nregions > 0 ? my_counts[0] : 0u;
    auto load_slab2 = [&](uint4 (&v)[kInFlight / 4]) -> bool {
#pragma unroll
        for (int j = 0; j < kInFlight / 4; ++j) v[j] = make_uint4(0, 0, 0, 0);
        while (seg2 < a.nregions) {
            const u32 nsl = (cnt2 + kSlab - 1) / kSlab;
            if (g2 < sbase + nsl) break;
            sbase += nsl; ++seg2;
            cnt2 = seg2 < a.nregions ? my_counts[seg2] : 0u;
        }
        if (seg2 >= a.nregions) return false;
        const u32 s0 = (g2 - sbase) * kSlab;
        const u32 *region = a.scratch + (u64)seg2 * a.seg_stride + (u64)b * a.block_stride;
#pragma unroll
        for (int j = 0; j < kInFlight / 4; ++j) {  // regions are zero-padded to 16 entries
            const u32 i = s0 + 4 * (lane + 64 * j);
            if (i < cnt2) v[j] = *reinterpret_cast<const uint4 *>(region + i);
        }
        g2 += kWaves;
        return true;
    };
    // One level: nregions short regions per block, wave w takes regions w, w + 16, ...  Their sizes are fetched
    // with ONE load (lane l holds the size of the wave's l-th region), and the slabs are double-buffered across region
    // boundaries: a region is only ~2 KB, so a wave that waited for each one separately would keep too few bytes in flight.
    // Here too the first slab is requested before the block is loaded.
    constexpr bool two_level = TWO;
    const int my_nreg = two_level ? 0 : (a.nregions - wave + kWaves - 1) / kWaves;  // regions of this wave (64 per batch)
    int r0 = 0, nr = my_nreg < 64 ? my_nreg : 64, ri = 0;  // ri: region index within the batch; off: offset inside the region
    u32 off = 0, my_cnt = (!two_level && lane < nr) ? my_counts[wave + kWaves * lane] : 0u;
    auto load_slab1 = [&](uint4 (&v)[kInFlight / 4]) -> bool {
        u32 cnt = 0;
        while (ri < nr) {
            cnt = (u32)__builtin_amdgcn_readlane((int)my_cnt, ri);
            if (off < cnt) break;
            ++ri; off = 0;
        }
#pragma unroll
        for (int j = 0; j < kInFlight / 4; ++j) v[j] = make_uint4(0, 0, 0, 0);
        if (ri >= nr) return false;
        const u32 *region = a.scratch + (u64)(wave + kWaves * (r0 + ri)) * a.seg_stride + (u64)b * a.block_stride;
#pragma unroll
        for (int j = 0; j < kInFlight / 4; ++j) {  // 16-byte loads: a lane takes four consecutive entries (regions are zero-padded to 16)
            const u32 i = off + 4 * (lane + 64 * j);
            if (i < cnt) v[j] = *reinterpret_cast<const uint4 *>(region + i);
        }
        off += kSlab;
        return true;
    };
    uint4 va2[kInFlight / 4], vb2[kInFlight / 4];
    bool more2 = two_level ? load_slab2(va2) : (my_nreg > 0 ? load_slab1(va2) : false);
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    if (a.fresh) for (u32 i = threadIdx.x; i < S / 2; i += kPartThreads) t4[i] = make_uint4(0, 0, 0, 0);
    else for (u32 i = threadIdx.x; i < S / 2; i += kPartThreads) t4[i] = reinterpret_cast<const uint4 *>(gblock)[i];
    __syncthreads();
    u32 counted = 0, newkeys = 0;
    auto insert = [&](u32 e) {
        u32 g = e & smask & ~(u32)(kGroup - 1);
        bool placed = false;
        for (u32 round = 0; round < (S >> kGroupBits) && !placed; ++round) {
            const uint4 q0 = *reinterpret_cast<const uint4 *>(keys + g), q1 = *reinterpret_cast<const uint4 *>(keys + g + 4);
            const u32 k[kGroup] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            int sel = kGroup;
#pragma unroll
            for (int i = kGroup - 1; i >= 0; --i) if (k[i] == e || k[i] == 0) sel = i;
            while (sel < kGroup) {
                u32 ks = keys[g + sel];
                if (ks == 0) {
                    ks = atomicCAS(&keys[g + sel], 0u, e);
                    if (ks == 0) { ++newkeys; ks = e; }
                }
                if (ks == e) { atomicAdd(&cnts[g + sel], 1u); placed = true; break; }
                ++sel;
                while (sel < kGroup) { const u32 kk = keys[g + sel]; if (kk == e || kk == 0) break; ++sel; }
            }
            g = (g + kGroup) & smask;
        }
        if (placed) ++counted;
        else s_failed = 1u;  // block full: the whole block is abandoned (FailedBlocks)
    };
    u32 *myq = wq + wave * kWaveQueue32;
    u32 qn = 0;
    // (qn is the same in every lane: kept on the scalar unit -- v_readfirstlane -- its tests are s_cmp + s_cbranch, no exec-mask regions)
    auto drain = [&](u32 keep_below) {
        while (qn > keep_below) {
            const u32 take = qn < 64 ? qn : 64;
            qn = (u32)__builtin_amdgcn_readfirstlane((int)(qn - take));
            if ((u32)lane < take) insert(myq[qn + lane]);
        }
    };
    auto fast = [&](u32 e) {
        bool miss = e != 0;
        if (e != 0) {
            const u32 g = e & smask & ~(u32)(kGroup - 1);
            // The home group's eight keys are only 32 bytes: they are read outright and compared -- ONE LDS round trip, where the
            // 64-bit K2 reads the group's fingerprint bytes first and its candidate key second (K2-32 -9 %; no fingerprint
            // array to build when the block is loaded, either).
            const uint4 q0 = *reinterpret_cast<const uint4 *>(keys + g), q1 = *reinterpret_cast<const uint4 *>(keys + g + 4);
            const u32 kk[kGroup] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            int idx = -1;
#pragma unroll
            for (int i = kGroup - 1; i >= 0; --i) if (kk[i] == e) idx = i;
            if (idx >= 0) { atomicAdd(&cnts[g + idx], 1u); ++counted; miss = false; }
            // (Claiming a first sighting's slot right here, as the 64-bit K2 does, was measured and dropped: a path that one lane
            // in ten takes is a path nearly every WAVE takes, and finding the group's first free slot among eight 32-bit keys
            // costs every entry of the wave -- K2-32 +22 % in the steady state, +13 % even on a cold pass.)
        }
        const u64 m = __ballot(miss);
        if (m) {
            const u32 pos = qn + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            if (miss) myq[pos] = e;
            qn = (u32)__builtin_amdgcn_readfirstlane((int)(qn + (u32)__popcll(m)));
            if (qn > kWaveQueue32 - 64) drain(31);
        }
    };
    auto count_slab = [&](const uint4 (&v)[kInFlight / 4]) {
        if (a.ablate & 16) {
#pragma unroll
            for (int j = 0; j < kInFlight / 4; ++j) counted += v[j].x + v[j].y + v[j].z + v[j].w;
            return;
        }
#pragma unroll
        for (int j = 0; j < kInFlight / 4; ++j) { fast(v[j].x); fast(v[j].y); fast(v[j].z); fast(v[j].w); }
    };
    if (two_level) {  // (the first slab is already in registers)
        while (more2) {
            more2 = load_slab2(vb2);
            count_slab(va2);
            if (!more2) break;
            more2 = load_slab2(va2);
            count_slab(vb2);
        }
    }
    // One level: nregions short regions per block (loop state and the first slab: see the top of the kernel).
    for (; r0 < my_nreg; r0 += 64) {
        if (r0 > 0) {  // (more than 64 regions per wave: the following batches)
            nr = my_nreg - r0 < 64 ? my_nreg - r0 : 64;
            my_cnt = lane < nr ? my_counts[wave + kWaves * (r0 + lane)] : 0u;
            ri = 0; off = 0;
            more2 = load_slab1(va2);
        }
        while (more2) {
            more2 = load_slab1(vb2);
            count_slab(va2);
            if (!more2) break;
            more2 = load_slab1(va2);
            count_slab(vb2);
        }
    }
    drain(0);
    u64 wc = wave_sum((u64)counted), wn = wave_sum((u64)newkeys);
    if (lane == 0) { atomicAdd(&s_counted, wc); atomicAdd(&s_new, wn); }
    __syncthreads();
    if (s_failed) {
        if (a.fresh) for (u32 i = threadIdx.x; i < S / 2; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) {
            u64 entries = 0;
            for (int r = 0; r < a.nregions; ++r) entries += my_counts[r];
            a.failed.list[atomicAdd(a.failed.n, 1ULL)] = b;
            atomicAdd(a.failed.entries, entries);
        }
    } else {
        for (u32 i = threadIdx.x; i < S / 2; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = t4[i];
        if (threadIdx.x == 0) { sum_counted += s_counted; sum_new += s_new; }
    }
    __syncthreads();  // the block has left LDS (its stores may still be in flight) and the tallies are read
    }
    if (threadIdx.x == 0) {
        u64 *shard = a.counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (sum_counted) atomicAdd(shard + CTR_COUNTED, sum_counted);
        if (sum_new) atomicAdd(shard + CTR_NEW_BY_ZERO, sum_new);  // new SHADOW keys: kept apart from the merges' new table keys (CTR_NEWKEYS)
    }
}

// the compact shadow table's pending counts -> the real table (cf. shadow_flush_kernel)
__global__ __launch_bounds__(kBlock) void shadow32_flush_kernel(u32 *__restrict__ shadow, int block_bits, u64 slots, TableView main, int k,
                                                                u64 *counters, int sbits, u32 bin0) {
    __shared__ u32 ascii4[256];
    __shared__ u64 s_tot, s_new;
    fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; }
    __syncthreads();
    const u64 S = 1ULL << block_bits;
    u64 tot = 0, nk = 0;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < slots; s += (u64)gridDim.x * kBlock) {
        const u64 blk = s >> block_bits, kw = (blk << (block_bits + 1)) + (s & (S - 1));
        const u32 c = shadow[kw + S];
        if (c == 0) continue;
        shadow[kw + S] = 0;
        const u64 h = hash_of_mixed<2>(((bin0 + (blk >> (sbits - 10))) << 32) | shadow[kw], k, ascii4);
        if (h == 0) continue;  // lib.rs:589: hash 0 is skipped
        const AddResult r = table_add<false>(main, h, (u64)c);
        if (!r.spilled) { tot += c; nk += r.claimed; }
    }
    tot = wave_sum(tot); nk = wave_sum(nk);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

// ---- partitioned flush of the compact shadow (tables of up to 1024 blocks) -----------------------------------------
// One random table access per pending k-mer (shadow32_flush_kernel) runs at HBM's random-access rate, ~0.1 ns each.
// Here the {hash, count} pairs take the k-mers' own route instead: this kernel hashes every pending k-mer and
// radix-partitions the PAIRS by table block through the LDS ring (16-byte entries, four per line), and
// aggregate_pairs_kernel merges each block's pairs in LDS.  The table is read and written once, sequentially.

// SRC 0: the compact shadow; 1: the 64-bit shadow; 2: a flat list of {hash, count} pairs (merges: add(), load(), the multi-GPU merge)
template <int SRC, int KC = 0>  // KC > 0: k at compile time
__global__ __launch_bounds__(kPartThreads) void flush_partition_kernel(FlushPartitionArgs a) {
    constexpr bool COMPACT = SRC == 0;
    const int k = KC > 0 ? KC : a.k;
    using W = typename std::conditional<COMPACT, u32, u64>::type;  // shadow word
    constexpr int kPairs = kRingEntries / 2;  // 8192 pairs = 128 KiB
    __shared__ __attribute__((aligned(16))) ulonglong2 ring[kPairs];
    __shared__ u64 cur[1024];  // per bin: fill (low half) | flushed (high half)
    constexpr u32 kListCap = COMPACT ? 1024 : 2048;  // (the compact variant's lane-compaction queues want the room)
    __shared__ u32 flist[kListCap];
    __shared__ u32 fcount;
    __shared__ u32 ascii4[256];
    // compact shadow: the slots with a pending count are gathered per wave (key | count << 32, and the block they came from) and hashed
    // 64 at a time -- a shadow is 10-50 % occupied, and hashing a row in place left that share of the lanes working
    constexpr u32 kQueue = 96;
    __shared__ u64 qv[COMPACT ? (kPartThreads / 64) * kQueue : 1];
    __shared__ unsigned short qb[COMPACT ? (kPartThreads / 64) * kQueue : 1];
    fill_ascii4_lut(ascii4, threadIdx.x, kPartThreads);
    const int P = 1 << a.pbits;
    const u32 D = (u32)(kPairs >> a.pbits), dmask = D - 1;
    const int dshift = __builtin_ctz((unsigned)kPairs) - a.pbits;
    for (int i = threadIdx.x; i < kPairs; i += kPartThreads) ring[i] = make_ulonglong2(0, 0);
    for (int i = threadIdx.x; i < 1024; i += kPartThreads) cur[i] = 0;
    if (threadIdx.x == 0) fcount = 0;
    __syncthreads();
    ulonglong2 *my_scratch = a.scratch + (u64)blockIdx.x * P * a.region_cap;
    auto overflow_pair = [&](u64 h, u64 c) {
        const u64 i = atomicAdd(a.ovf_n, 1ULL);
        if (i < a.ovf_cap) { a.ovf[2 * i] = h; a.ovf[2 * i + 1] = c; }  // (cap = every pending k-mer: cannot be exceeded)
    };
    auto flush_lines = [&](bool drain) {
        return ring_flush<kListCap, ulonglong2>(ring, cur, flist, &fcount, P, D, my_scratch, a.region_cap, drain, overflow_pair);
    };
    // this workgroup's share of the shadow: whole blocks, kBlocksPerWg of them, one row of 1024 slots per step
    constexpr u32 S = 1u << kBlockBitsMax;
    if constexpr (SRC == 2) {
        // Pair lists are often sorted by the table slot they came from (an export of a table of the same geometry: add(),
        // the multi-GPU merge): read in order, a row's 1024 pairs would all want the same few bins and nearly all of them
        // would overflow the ring (5 M pairs: 1.0 ms here plus 0.5 ms of atomic inserts, against 0.1 ms).  The list is
        // therefore read TRANSPOSED, as a matrix of 8192 columns walked down the columns: neighbouring lanes take pairs
        // 8192 apart -- more than a block's worth of keys.
        constexpr u64 kCols = 8192;
        const u64 mrows = (a.npairs + kCols - 1) / kCols, cells = mrows * kCols;
        const u64 rows = (cells + kPartThreads - 1) / kPartThreads, rows_per_wg = (rows + gridDim.x - 1) / gridDim.x;
        for (u64 r = 0; r < rows_per_wg; ++r) {
            const u64 cell = ((u64)blockIdx.x * rows_per_wg + r) * kPartThreads + threadIdx.x;
            const u64 i = cell < cells ? (cell % mrows) * kCols + cell / mrows : a.npairs;
            const u64 h = i < a.npairs ? a.pair_keys[i * a.pair_stride] : 0ULL;
            if (h) {
                const u64 c = a.pair_counts[i * a.pair_stride];
                const u32 b = (u32)(h >> a.table_block_bits) & (u32)(P - 1);
                const u64 cw = atomicAdd(&cur[b], 1ULL);
                const u32 pos = (u32)cw;
                if (pos - (u32)(cw >> 32) < D) ring[(b << dshift) + (pos & dmask)] = make_ulonglong2(h, c);
                else overflow_pair(h, c);
            }
            flush_lines(false);  // a full row of pairs over up to 1024 bins of 8: every row
        }
    }
    // The shadow is walked a row of 1024 slots at a time, FOUR rows' counts and keys in flight ahead of the row being
    // hashed (one workgroup per CU and a dependent load per row made this kernel latency-bound: ~2 us of HBM round trip
    // for ~1 us of work; lane compaction of the ~47 % occupied slots, tried first, bought nothing).
    const u32 nblocks = SRC == 2 ? 0u : a.shadow_blocks, per_wg = (nblocks + gridDim.x - 1) / gridDim.x;
    constexpr u32 kRowsPerBlock = S / kPartThreads, kAhead = 4;  // (8 ahead: no better at k <= 21, worse with 64-bit words)
    const u32 sb0 = blockIdx.x * per_wg, sb1 = sb0 + per_wg < nblocks ? sb0 + per_wg : nblocks;
    const u32 nrows = sb1 > sb0 ? (sb1 - sb0) * kRowsPerBlock : 0u;
    W *base = reinterpret_cast<W *>(a.shadow);
    auto count_ptr = [&](u32 r) -> W * {  // the count of this thread's slot in row r of this workgroup's share
        const u32 sb = sb0 + r / kRowsPerBlock, row = r % kRowsPerBlock;
        return base + ((u64)sb << (kBlockBitsMax + 1)) + S + row * kPartThreads + threadIdx.x;
    };
    W cq[kAhead], kq[kAhead];
#pragma unroll
    for (u32 j = 0; j < kAhead; ++j) {
        cq[j] = 0; kq[j] = 0;
        if (!COMPACT && j < nrows) { const W *cp = count_ptr(j); cq[j] = *cp; kq[j] = *(cp - S); }
    }
    auto emit_pair = [&](u64 h, u64 c) {
        if (h) {  // lib.rs:589: hash 0 is skipped
            const u32 b = (u32)(h >> a.table_block_bits) & (u32)(P - 1);
            const u64 cw = atomicAdd(&cur[b], 1ULL);
            const u32 pos = (u32)cw;
            if (pos - (u32)(cw >> 32) < D) ring[(b << dshift) + (pos & dmask)] = make_ulonglong2(h, c);
            else overflow_pair(h, c);
        }
    };
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    u32 qn = 0;  // entries in this wave's queue (wave-uniform)
    auto hash_queued = [&](u32 take) {  // the queue's last `take` entries, one per lane
        qn -= take;
        if (lane < take) {
            const u64 e = qv[wave * kQueue + qn + lane];
            const u32 sb = sb0 + qb[wave * kQueue + qn + lane];
            emit_pair(hash_of_mixed<2>(((u64)(a.shadow_bin0 + (sb >> (a.shadow_sbits - 10))) << 32) | (e & 0xFFFFFFFFULL), k, ascii4), e >> 32);
        }
    };
    if constexpr (COMPACT) {
        // The compact shadow is walked 4096 slots a step: every lane takes FOUR neighbouring counts and keys with 16-byte loads (a
        // block is two steps), four steps in flight -- 128 KiB per workgroup: the 4-byte-per-lane walk below left the kernel bound by
        // its few bytes in flight, not by hashing.  A lane's four slots are queued one after the other (four ballots per step).
        (void)cq; (void)kq;
        constexpr u32 kStepsPerBlock = S / (4 * kPartThreads);
        const u32 nsteps = sb1 > sb0 ? (sb1 - sb0) * kStepsPerBlock : 0u;
        auto count4_ptr = [&](u32 st) -> uint4 * {
            const u32 sb = sb0 + st / kStepsPerBlock, part = st % kStepsPerBlock;
            return reinterpret_cast<uint4 *>(base + ((u64)sb << (kBlockBitsMax + 1)) + S + part * (4 * kPartThreads)) + threadIdx.x;
        };
        uint4 c4[kAhead], k4[kAhead];
#pragma unroll
        for (u32 j = 0; j < kAhead; ++j) {
            c4[j] = make_uint4(0, 0, 0, 0); k4[j] = make_uint4(0, 0, 0, 0);
            if (j < nsteps) { const uint4 *cp = count4_ptr(j); c4[j] = *cp; k4[j] = *(cp - S / 4); }
        }
        for (u32 s0 = 0; s0 < nsteps; s0 += kAhead) {
#pragma unroll
            for (u32 j = 0; j < kAhead; ++j) {
                const u32 st = s0 + j;
                if (st >= nsteps) break;
                const uint4 cc = c4[j], kk = k4[j];
                if (st + kAhead < nsteps) { const uint4 *cp = count4_ptr(st + kAhead); c4[j] = *cp; k4[j] = *(cp - S / 4); }
                if (cc.x | cc.y | cc.z | cc.w) *count4_ptr(st) = make_uint4(0, 0, 0, 0);
                const u32 cs[4] = {cc.x, cc.y, cc.z, cc.w}, ks[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 has = __ballot(cs[q] != 0);
                    const u32 n = (u32)__popcll(has);
                    if (qn + n > kQueue) hash_queued(qn < 64u ? qn : 64u);
                    if (cs[q]) {
                        const u32 at = wave * kQueue + qn + __builtin_amdgcn_mbcnt_hi((u32)(has >> 32), __builtin_amdgcn_mbcnt_lo((u32)has, 0u));
                        qv[at] = (u64)ks[q] | ((u64)cs[q] << 32);
                        qb[at] = (unsigned short)(st / kStepsPerBlock);
                    }
                    qn += n;
                    if (qn >= 64u) hash_queued(64u);
                    if (q & 1) flush_lines(false);  // ~1000 pairs per half step over up to 1024 bins of 8
                }
            }
        }
        while (qn) hash_queued(qn < 64u ? qn : 64u);
    } else {
    for (u32 r0 = 0; r0 < nrows; r0 += kAhead) {
#pragma unroll
        for (u32 j = 0; j < kAhead; ++j) {
            const u32 r = r0 + j;
            if (r >= nrows) break;
            const W c = cq[j], key = kq[j];
            if (r + kAhead < nrows) { const W *cp = count_ptr(r + kAhead); cq[j] = *cp; kq[j] = *(cp - S); }  // the row four ahead
            if (c) { *count_ptr(r) = 0; emit_pair(hash_of_mixed<1>((u64)key, k, ascii4), (u64)c); }
            if ((r & 1) == 1) flush_lines(false);  // ~600 pairs per row over up to 1024 bins of 8: every second row
        }
    }
    }
    while (flush_lines(true)) {}
    for (int b = threadIdx.x; b < P; b += kPartThreads) {
        const u32 f = (u32)(cur[b] >> 32);
        a.region_count[(u64)b * gridDim.x + blockIdx.x] = f < a.region_cap ? f : a.region_cap;
    }
}

// One workgroup per table block: the block in LDS, its pairs merged with LDS atomics, the block stored back.
// Few pairs per block (thousands, against 10^5 k-mers in a counting pass): the general insert alone will do.

__global__ __launch_bounds__(kPartThreads) void aggregate_pairs_kernel(AggregatePairsArgs a) {
    __shared__ __attribute__((aligned(16))) u64 tab[2 << kBlockBitsMax];
    __shared__ u64 s_tot, s_new, s_nz;
    __shared__ u32 s_failed;
    const u32 S = 1u << a.block_bits, smask = S - 1;
    u64 *keys = tab, *cnts = tab + S;
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    const int lane = threadIdx.x & 63;
    const bool few = a.nregions < 64;
    u64 sum_tot = 0, sum_new = 0, sum_nz = 0;  // thread 0: the tallies of the blocks this workgroup has stored
    // A workgroup takes blocks b, b + grid, ...: with two partition levels the grid is one workgroup per CU, and a block's
    // 128 KiB of stores drain while the next block is zeroed and merged (a workgroup per block left the CU's share of HBM
    // idle for half of every block's time: nothing overlaps the stores of a workgroup that is ending).
    // Two levels: the block's (few, long) regions.  The first rows of a block's pairs are requested BEFORE the block itself
    // is loaded, so that their HBM round trip runs under the block load instead of after it (one workgroup per CU: nothing
    // else would hide it).  (Requesting them before the PREVIOUS block is stored was slower: 6.9 -> 7.2 ms.)
    constexpr int kPre = 4;
    ulonglong2 pre[kPre];
    auto request_rows = [&](u32 b) {
        const u32 cnt0 = few && a.nregions > 0 && b < a.nblocks ? a.region_count[(u64)b * a.nregions] : 0u;
        const ulonglong2 *region0 = a.scratch + (u64)b * a.block_stride;
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const u32 i = threadIdx.x + (u32)j * kPartThreads;
            pre[j] = i < cnt0 ? region0[i] : make_ulonglong2(0, 0);
        }
    };
    for (u32 b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        u64 *gblock = a.words + ((u64)b << (a.block_bits + 1));
        if (threadIdx.x == 0) { s_tot = 0; s_new = 0; s_nz = 0; s_failed = 0; }
        request_rows(b);
        const u32 *my_counts = a.region_count + (u64)b * a.nregions;
        if (a.fresh) for (u32 i = threadIdx.x; i < S; i += kPartThreads) t4[i] = make_uint4(0, 0, 0, 0);
        else for (u32 i = threadIdx.x; i < S; i += kPartThreads) t4[i] = reinterpret_cast<const uint4 *>(gblock)[i];
        __syncthreads();
        u64 tot = 0, nk = 0, nz = 0;
        auto merge_pair = [&](const ulonglong2 pr) {
            const u64 h = pr.x, c = pr.y;
            if (h == 0) return;  // hole / padding
            u32 g = (u32)h & smask & ~(u32)(kGroup - 1);
            bool placed = false;
            for (u32 round = 0; round < (S >> kGroupBits) && !placed; ++round) {
                const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(keys + g);
                const ulonglong2 q0 = kp[0], q1 = kp[1], q2 = kp[2], q3 = kp[3];
                const u64 kk[kGroup] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
                int sel = kGroup;  // first slot that holds h or is empty
#pragma unroll
                for (int sl = kGroup - 1; sl >= 0; --sl) if (kk[sl] == h || kk[sl] == 0) sel = sl;
                while (sel < kGroup) {
                    u64 ks = keys[g + sel];
                    if (ks == 0) {
                        ks = atomicCAS(&keys[g + sel], 0ULL, h);
                        if (ks == 0) { ++nk; ks = h; }
                    }
                    if (ks == h) { nz += atomicAdd(&cnts[g + sel], c) == 0; placed = true; break; }  // (lib.rs:801-803: new = count was 0)
                    ++sel;  // another lane claimed that slot for a different key: try the following slots
                    while (sel < kGroup) { const u64 k2 = keys[g + sel]; if (k2 == h || k2 == 0) break; ++sel; }
                }
                g = (g + kGroup) & smask;
            }
            if (placed) tot += c;
            else s_failed = 1u;  // block full: the whole block is abandoned (FailedBlocks)
        };
        if (!few) {
            // One level: a region holds only a handful of pairs (one workgroup's share of one block): FOUR lanes take a region,
            // so all 256 regions are in flight at once and a group's loads are one 64-byte line.
            const int q = threadIdx.x & 3;
            for (int seg = threadIdx.x >> 2; seg < a.nregions; seg += kPartThreads / 4) {
                const u32 cnt = my_counts[seg];
                const ulonglong2 *region = a.scratch + (u64)seg * a.seg_stride + (u64)b * a.block_stride;
                for (u32 i = q; i < cnt; i += 4) merge_pair(region[i]);
            }
        } else {
            // Two levels: a few long regions per block (one per second-level writer), the whole workgroup strides over each
            // (the first kPre rows of region 0 are already in registers).
#pragma unroll
            for (int j = 0; j < kPre; ++j) merge_pair(pre[j]);
            for (int seg = 0; seg < a.nregions; ++seg) {
                const u32 cnt = my_counts[seg];
                const ulonglong2 *region = a.scratch + (u64)seg * a.seg_stride + (u64)b * a.block_stride;
                for (u32 i = threadIdx.x + (seg == 0 ? kPre * kPartThreads : 0); i < cnt; i += kPartThreads) merge_pair(region[i]);
            }
        }
        tot = wave_sum(tot); nk = wave_sum(nk); nz = wave_sum(nz);
        if (lane == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); atomicAdd(&s_nz, nz); }
        __syncthreads();
        if (s_failed) {
            if (a.fresh) for (u32 i = threadIdx.x; i < S; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = make_uint4(0, 0, 0, 0);
            if (threadIdx.x == 0) {
                u64 entries = 0;
                for (int r = 0; r < a.nregions; ++r) entries += my_counts[r];
                a.failed.list[atomicAdd(a.failed.n, 1ULL)] = b;
                atomicAdd(a.failed.entries, entries);
            }
        } else {
            for (u32 i = threadIdx.x; i < S; i += kPartThreads) reinterpret_cast<uint4 *>(gblock)[i] = t4[i];
            if (threadIdx.x == 0) { sum_tot += s_tot; sum_new += s_new; sum_nz += s_nz; }
        }
        __syncthreads();  // the block has left LDS (its stores may still be in flight) and the tallies are read
    }
    if (threadIdx.x == 0) {
        u64 *shard = a.counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (sum_tot) atomicAdd(shard + CTR_TOTAL_ADDED, sum_tot);
        if (sum_new) atomicAdd(shard + CTR_NEWKEYS, sum_new);
        if (sum_nz) atomicAdd(shard + CTR_NEW_BY_ZERO, sum_nz);
    }
}


// =================================================================================================
// 128-bit dedupe-first path (33 <= k <= 64).  The packed k-mer is two words; K1 (MODE 3) partitions 16-byte {x, y} = mix128 entries
// by x into 1024 bins; this is K2 for them: one workgroup per shadow block of 4096 slots (x, y, u32 count in LDS), general insert
// only.  Claim protocol for a two-word key: CAS on the slot's x word; the winner then writes y and sets bit 31 of the count word;
// whoever finds its own x in a slot waits for that bit before comparing y.  The loops are wave-uniform, so within a wave every
// claim of a step has been completed (program order) before any lane of that wave waits; across waves the wait is a plain spin.
// =================================================================================================
constexpr int kWaveQueue128 = 96;  // deferred entries per wave (16 B each: 24 KiB for the sixteen waves)
__global__ __launch_bounds__(kPartThreads) void aggregate_blocks128_kernel(Aggregate128Args a) {
    __shared__ __attribute__((aligned(16))) u64 kx[kSlots128], ky[kSlots128];
    __shared__ __attribute__((aligned(16))) u32 cn[kSlots128];
    __shared__ __attribute__((aligned(16))) unsigned char tags[kSlots128];  // one fingerprint byte per slot (0 = not (yet) claimed)
    __shared__ __attribute__((aligned(16))) ulonglong2 wq[(kPartThreads / 64) * kWaveQueue128];
    __shared__ u64 s_counted, s_new;
    __shared__ u32 s_failed;
    if (*a.overflow) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kPartThreads / 64, kInFlight = 6;
    constexpr u32 kSlab = 64 * kInFlight;
    auto tag_of = [](u64 x) -> u32 { const u32 tg = (u32)(x >> 32) & 0xFFu; return tg ? tg : 1u; };
    u64 sum_counted = 0, sum_new = 0;
    for (u32 b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        u64 *gb = a.words + (u64)b * kBlockWords128;
        if (threadIdx.x == 0) { s_counted = 0; s_new = 0; s_failed = 0; }
        uint4 *x4 = reinterpret_cast<uint4 *>(kx), *y4 = reinterpret_cast<uint4 *>(ky), *c4 = reinterpret_cast<uint4 *>(cn);
        const uint4 *g4 = reinterpret_cast<const uint4 *>(gb);
        for (u32 i = threadIdx.x; i < kSlots128 / 2; i += kPartThreads) {
            x4[i] = a.fresh ? make_uint4(0, 0, 0, 0) : g4[i];
            y4[i] = a.fresh ? make_uint4(0, 0, 0, 0) : g4[kSlots128 / 2 + i];
        }
        for (u32 i = threadIdx.x; i < kSlots128 / 4; i += kPartThreads) c4[i] = a.fresh ? make_uint4(0, 0, 0, 0) : g4[kSlots128 + i];
        __syncthreads();
        for (u32 i = threadIdx.x; i < kSlots128 / 4; i += kPartThreads) {  // four slots per thread: their tags as one 4-byte word
            u32 w = 0;
            if (!a.fresh) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const u64 xx = kx[4 * i + q]; w |= (xx ? tag_of(xx) : 0u) << (8 * q); }
            }
            reinterpret_cast<u32 *>(tags)[i] = w;
        }
        __syncthreads();
        u32 counted = 0, newkeys = 0;
        // General insert (wave-uniform loops; `active` lanes carry an entry).  A group is four slots.
        auto insert = [&](const ulonglong2 e, bool active) {
            const u64 x = e.x, y = e.y;
            bool pending = active && x != 0;  // (x = 0: hole / padding)
            u32 g = (u32)x & (kSlots128 - 1) & ~3u;
            for (u32 round = 0; round < kSlots128 / 4; ++round) {
                if (!__any(pending)) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bool mine = false;
                    u64 cur = 0;
                    if (pending) {
                        cur = kx[g + i];
                        if (cur == 0) {
                            cur = atomicCAS(&kx[g + i], 0ULL, x);
                            if (cur == 0) {
                                ky[g + i] = y;
                                __threadfence_block();
                                atomicOr(&cn[g + i], kClaimed128);
                                tags[g + i] = (unsigned char)tag_of(x);  // (last: a tag in place means the slot is complete)
                                mine = true; cur = x; ++newkeys;
                            }
                        }
                    }
                    if (pending && cur == x) {
                        if (!mine) while (!(atomicOr(&cn[g + i], 0u) & kClaimed128)) {}
                        if (ky[g + i] == y) { atomicAdd(&cn[g + i], 1u); pending = false; ++counted; }
                    }
                }
                g = (g + 4) & (kSlots128 - 1);
            }
            if (pending) s_failed = 1u;  // block full: the whole block is abandoned (FailedBlocks)
        };
        ulonglong2 *myq = wq + wave * kWaveQueue128;
        u32 qn = 0;  // wave-uniform
        auto drain = [&](u32 keep_below) {
            while (qn > keep_below) {
                const u32 take = qn < 64 ? qn : 64;
                qn -= take;
                const bool act = (u32)lane < take;
                insert(act ? myq[qn + lane] : make_ulonglong2(0, 0), act);
            }
        };
        // Fast path: a repeat sighting of a key that sits in its home group -- four tag bytes (one 4-byte read), a SWAR byte match,
        // the two key words to confirm, one ds_add.  Everything else is parked in the wave's queue for the general insert.
        auto fast = [&](const ulonglong2 e) {
            bool miss = e.x != 0;
            if (e.x != 0) {
                const u32 g = (u32)e.x & (kSlots128 - 1) & ~3u;
                const u32 t4 = *reinterpret_cast<const u32 *>(tags + g);
                const u32 xo = t4 ^ (tag_of(e.x) * 0x01010101u);
                const u32 z = (xo - 0x01010101u) & ~xo & 0x80808080u;  // lowest set bit marks the first equal byte
                if (z) {
                    const u32 idx = (u32)__builtin_ctz(z) >> 3;
                    if (kx[g + idx] == e.x && ky[g + idx] == e.y) { atomicAdd(&cn[g + idx], 1u); ++counted; miss = false; }
                }
            }
            const u64 m = __ballot(miss);
            if (m) {
                const u32 pos = qn + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                if (miss) myq[pos] = e;
                qn += (u32)__popcll(m);
                if (qn > kWaveQueue128 - 64) drain(31);
            }
        };
        // one level: nregions short regions per block, one wave each; all of a lane's loads of a slab are issued before the first insert
        const u32 *my_counts = a.region_count + (u64)b * a.nregions;
        for (int seg = wave; seg < a.nregions; seg += kWaves) {
            const u32 cnt = my_counts[seg];
            const ulonglong2 *region = a.scratch + (u64)seg * a.seg_stride + (u64)b * a.block_stride;
            for (u32 s0 = 0; s0 < cnt; s0 += kSlab) {
                ulonglong2 v[kInFlight];
#pragma unroll
                for (int j = 0; j < kInFlight; ++j) { const u32 i = s0 + lane + 64 * j; v[j] = i < cnt ? region[i] : make_ulonglong2(0, 0); }
#pragma unroll
                for (int j = 0; j < kInFlight; ++j) fast(v[j]);
            }
        }
        drain(0);
        const u64 wc = wave_sum((u64)counted), wn = wave_sum((u64)newkeys);
        if (lane == 0) { atomicAdd(&s_counted, wc); atomicAdd(&s_new, wn); }
        __syncthreads();
        if (s_failed) {
            if (a.fresh) for (u32 i = threadIdx.x; i < kBlockWords128 / 2; i += kPartThreads) reinterpret_cast<uint4 *>(gb)[i] = make_uint4(0, 0, 0, 0);
            if (threadIdx.x == 0) {
                u64 entries = 0;
                for (int r = 0; r < a.nregions; ++r) entries += my_counts[r];
                a.failed.list[atomicAdd(a.failed.n, 1ULL)] = b;
                atomicAdd(a.failed.entries, entries);
            }
        } else {
            uint4 *o4 = reinterpret_cast<uint4 *>(gb);
            for (u32 i = threadIdx.x; i < kSlots128 / 2; i += kPartThreads) { o4[i] = x4[i]; o4[kSlots128 / 2 + i] = y4[i]; }
            for (u32 i = threadIdx.x; i < kSlots128 / 4; i += kPartThreads) o4[kSlots128 + i] = c4[i];
            if (threadIdx.x == 0) { sum_counted += s_counted; sum_new += s_new; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64 *shard = a.counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (sum_counted) atomicAdd(shard + CTR_COUNTED, sum_counted);
        if (sum_new) atomicAdd(shard + CTR_NEW_BY_ZERO, sum_new);
    }
}

// mix128 pair of a packed canonical k-mer -> its MurmurHash3 value (33 <= k <= 64)
__device__ __forceinline__ u64 hash_of_mixed128(u64 x, u64 y, int k, const u32 *lut) {
    Packed<2> p;
    unmix128(x, y, p.w[0], p.w[1]);
    left_align(p, k);
    return hash_packed<2, true>(p, k, lut);
}

// the 128-bit shadow's pending counts -> the real table: every slot with a count is turned back into its k-mer, hashed once and
// added with the direct insert; the count returns to zero (bit 31 stays), the key stays.  At most 2.7 M k-mers: ~0.1 ns each.
__global__ __launch_bounds__(kBlock) void shadow128_flush_kernel(u64 *__restrict__ shadow, u32 nblocks, TableView main, int k, u64 *counters) {
    __shared__ u32 ascii4[256];
    __shared__ u64 s_tot, s_new;
    fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; }
    __syncthreads();
    u64 tot = 0, nk = 0;
    const u64 slots = (u64)nblocks << kBlockBits128;
    for (u64 s = (u64)blockIdx.x * kBlock + threadIdx.x; s < slots; s += (u64)gridDim.x * kBlock) {
        u64 *blk = shadow + (s >> kBlockBits128) * kBlockWords128;
        const u32 i = (u32)s & (kSlots128 - 1);
        u32 *cp = reinterpret_cast<u32 *>(blk + 2 * kSlots128) + i;
        const u32 c = *cp & ~kClaimed128;
        if (c == 0) continue;
        *cp = kClaimed128;
        const u64 h = hash_of_mixed128(blk[i], blk[kSlots128 + i], k, ascii4);
        if (h == 0) continue;  // lib.rs:589: hash 0 is skipped
        const AddResult r = table_add<false>(main, h, (u64)c);
        if (!r.spilled) { tot += c; nk += r.claimed; }
    }
    tot = wave_sum(tot); nk = wave_sum(nk);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

// {x, y} entries that did not reach the 128-bit shadow -> the real table with the direct insert: K1's overflow regions
// (failed_list == nullptr: nregions regions of up to region_cap entries, counts[r] each) or the scratch regions of the shadow
// blocks K2 abandoned (failed_list: for each listed block its nregions regions at scratch + seg * seg_stride + b * block_stride).
// Equal neighbours are folded per wave first.  Every entry below its region's count is live (x may be 0 in an overflow region).
__global__ __launch_bounds__(kBlock) void merge_entries128_kernel(const ulonglong2 *__restrict__ scratch, const u32 *__restrict__ counts, int nregions,
                                                                  u64 region_cap, u64 seg_stride, u64 block_stride, const u32 *failed_list, u64 nfailed,
                                                                  const u64 *abort, TableView table, u64 *counters, int k) {
    __shared__ u64 s_tot, s_new;
    __shared__ u32 ascii4[256];
    if (abort && *abort) return;
    fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 tot = 0, nk = 0;
    const u64 nwork = failed_list ? nfailed * (u64)nregions : (u64)nregions;
    for (u64 w = blockIdx.x; w < nwork; w += gridDim.x) {
        const ulonglong2 *src;
        u32 cnt;
        bool skip_holes;
        if (failed_list) {
            const u32 b = failed_list[w / nregions];
            const u64 seg = w % nregions;
            cnt = counts[(u64)b * nregions + seg];
            src = scratch + seg * seg_stride + (u64)b * block_stride;
            skip_holes = true;   // ring regions: x = 0 is a hole
        } else { cnt = counts[w]; src = scratch + w * region_cap; skip_holes = false; }
        for (u32 base = 64u * wave; base < cnt; base += kBlock) {
            const u32 i = base + lane;
            ulonglong2 e = make_ulonglong2(0, 0);
            bool pending = i < cnt;
            if (pending) e = src[i];
            if (skip_holes && e.x == 0) pending = false;
            bool is_leader = false;
            u64 c = 0, act;
            while ((act = __ballot(pending)) != 0) {
                const int leader = __ffsll((long long)act) - 1;
                const u64 xl = read_lane64(e.x, leader), yl = read_lane64(e.y, leader);
                const u64 same = __ballot(pending && e.x == xl && e.y == yl);
                if (lane == leader) { is_leader = true; c = (u64)__popcll(same); }
                if ((same >> lane) & 1ULL) pending = false;
            }
            if (is_leader) {
                const u64 hh = hash_of_mixed128(e.x, e.y, k, ascii4);
                if (hh != 0) {
                    const AddResult res = table_add<false>(table, hh, c);
                    if (!res.spilled) { tot += c; nk += res.claimed; }
                }
            }
        }
    }
    tot = wave_sum(tot); nk = wave_sum(nk);
    if (lane == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

// ---- overflow regions of the partitioned path -> table ---------------------------------------------
// regions[r] holds counts[r] hashes (each standing for one k-mer).  Neighbouring entries are often
// equal (that is why they overflowed), so every wave first folds equal hashes: the lowest active
// lane is the leader, all lanes holding the leader's hash retire into one add, repeat.
// DEDUPE: the entries are mix64 values of packed k-mers (dedupe-first path); each group leader hashes its k-mer first.
// pend (optional): instead of inserting, APPEND the folded {hash, count} pairs to a list that the next conversion of the
// pending counts merges -- so that a dedupe-first pass need not touch a table that is still lazily empty (no memset, and
// the conversion then starts every table block from zeros).  The entries are tallied as counted either way.

template <int DEDUPE = 0>  // 0: hashes; 1: mix64 values; 2: mix42 values (bit 63 set)
__global__ __launch_bounds__(kBlock) void merge_overflow_kernel(const u64 *__restrict__ regions, const u32 *__restrict__ counts,
                                                                int nregions, u32 region_cap, const u64 *abort, TableView table,
                                                                u64 *counters, int k = 0, const u64 *total = nullptr, PendingList pend = PendingList()) {
    __shared__ u64 s_tot, s_new;
    __shared__ u32 ascii4[DEDUPE ? 256 : 1];
    if (abort && *abort) return;
    if constexpr (DEDUPE) fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 tot = 0, nk = 0;
    for (int r = blockIdx.x; r < nregions; r += gridDim.x) {  // one workgroup per region
        // region sizes: counts[r], or (counts == nullptr) one list of *total entries cut into region_cap pieces
        u32 cnt;
        if (counts) cnt = counts[r];
        else { const u64 tot_n = *total, lo = (u64)r * region_cap; cnt = tot_n > lo ? (u32)(tot_n - lo < region_cap ? tot_n - lo : region_cap) : 0u; }
        const u64 *src = regions + (u64)r * region_cap;
        for (u32 base = 64u * wave; base < cnt; base += kBlock) {
            const u32 i = base + lane;
            const u64 h = i < cnt ? src[i] : 0ULL;
            // group equal hashes (registers and SGPRs only), then let every group leader add at once
            bool pending = h != 0, is_leader = false;
            u64 c = 0, act;
            while ((act = __ballot(pending)) != 0) {
                const int leader = __ffsll((long long)act) - 1;
                const u64 hl = read_lane64(h, leader);
                const u64 same = __ballot(pending && h == hl);
                if (lane == leader) { is_leader = true; c = (u64)__popcll(same); }
                if ((same >> lane) & 1ULL) pending = false;
            }
            if (is_leader) {
                const u64 hh = DEDUPE ? hash_of_mixed<(DEDUPE == 2 ? 2 : 1)>(h, k, ascii4) : h;
                if (hh != 0) {
                    if (pend.pairs) {
                        const u64 pi = atomicAdd(pend.n, 1ULL);
                        if (pi < pend.cap) { pend.pairs[2 * pi] = hh; pend.pairs[2 * pi + 1] = c; }  // (cap = every overflow entry: cannot be exceeded)
                        tot += c;
                    } else {
                        const AddResult res = table_add<false>(table, hh, c);
                        if (!res.spilled) { tot += c; nk += res.claimed; }
                    }
                }
            }
        }
    }
    tot = wave_sum(tot); nk = wave_sum(nk);
    if (lane == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

// ---- the abandoned blocks of a K2 launch -> the real table, with the direct insert ---------------------------------------
// One 256-thread workgroup per abandoned block walks the block's scratch regions (same layout as K2 read them).
// MODE 0: u64 MurmurHash3 values; 1: u64 mix64 values of packed k-mers (hashed here); 2: u32 compact entries (the block
// number gives the 42-bit value's upper bits; hashed here); 3: {hash, count} pairs.  Equal neighbours are folded per wave
// first (a block's entries repeat its k-mers many times over).  Tallies: CTR_TOTAL_ADDED, CTR_NEWKEYS, CTR_NEW_BY_ZERO.
template <int MODE>
__global__ __launch_bounds__(kBlock) void recount_failed_kernel(const void *scratch, u64 seg_stride, u64 block_stride, const u32 *region_count, int nregions,
                                                                const u32 *failed_list, u64 nfailed, TableView table, u64 *counters, int k, int sbits, u32 bin0 = 0) {
    using T = typename std::conditional<MODE == 3, ulonglong2, typename std::conditional<MODE == 2, u32, u64>::type>::type;
    __shared__ u64 s_tot, s_new, s_nz;
    __shared__ u32 ascii4[(MODE == 1 || MODE == 2) ? 256 : 1];
    if constexpr (MODE == 1 || MODE == 2) fill_ascii4_lut(ascii4, threadIdx.x, kBlock);
    if (threadIdx.x == 0) { s_tot = 0; s_new = 0; s_nz = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 tot = 0, nk = 0, nz = 0;
    for (u64 fi = blockIdx.x; fi < nfailed; fi += gridDim.x) {
        const u32 b = failed_list[fi];
        for (int seg = 0; seg < nregions; ++seg) {
            const u32 cnt = region_count[(u64)b * nregions + seg];
            const T *src = reinterpret_cast<const T *>(scratch) + (u64)seg * seg_stride + (u64)b * block_stride;
            for (u32 base = 64u * wave; base < cnt; base += kBlock) {
                const u32 i = base + lane;
                u64 h = 0, c = 1;
                if (i < cnt) {
                    if constexpr (MODE == 3) { const ulonglong2 pr = src[i]; h = pr.x; c = pr.y; }
                    else if constexpr (MODE == 2) { const u32 e = src[i]; h = e ? ((((u64)(bin0 + (b >> (sbits - 10)))) << 32) | e | (1ULL << 63)) : 0ULL; }
                    else h = src[i];
                }
                bool pending = h != 0, is_leader = false;
                u64 csum = 0, act;
                while ((act = __ballot(pending)) != 0) {  // fold equal values: one insert per distinct value of the wave
                    const int leader = __ffsll((long long)act) - 1;
                    const u64 hl = read_lane64(h, leader);
                    const bool mine = pending && h == hl;
                    const u64 same = __ballot(mine);
                    u64 part = mine ? c : 0ULL;  // (pairs carry their own counts)
                    if constexpr (MODE == 3) {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                    } else part = (u64)__popcll(same);
                    if (lane == leader) { is_leader = true; csum = part; }
                    if (mine) pending = false;
                }
                if (is_leader) {
                    u64 hh = h;
                    if constexpr (MODE == 1) hh = hash_of_mixed<1>(h, k, ascii4);
                    if constexpr (MODE == 2) hh = hash_of_mixed<2>(h, k, ascii4);
                    if (hh != 0) {
                        const AddResult res = table_add<true>(table, hh, csum);
                        if (!res.spilled) { tot += csum; nk += res.claimed; nz += res.old == 0; }
                    }
                }
            }
        }
    }
    tot = wave_sum(tot); nk = wave_sum(nk); nz = wave_sum(nz);
    if (lane == 0) { atomicAdd(&s_tot, tot); atomicAdd(&s_new, nk); atomicAdd(&s_nz, nz); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_tot) atomicAdd(shard + CTR_TOTAL_ADDED, s_tot);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
        if (s_nz) atomicAdd(shard + CTR_NEW_BY_ZERO, s_nz);
    }
}

}  // namespace kct
