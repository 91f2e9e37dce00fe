// partition_args.h -- argument blocks of the partitioned path's kernels (partition_kernels.h), kept apart from the kernels so
// that host translation units which only LAUNCH them through kct_consume.hip's launchers (kct_route.hip) need not
// instantiate them.
#pragma once
#include "device_common.h"

namespace kct {

// A stream of super-k-mers: bases[] holds runs back to back, 16 bases per word, first base in bits 31:30 (as PartitionArgs::pcodes);
// a run of n windows is n + k - 1 bases.  Windows are numbered in the order their runs lie in bases[]; one start bit per window says
// that it is the first of its run.  Several such streams (one per sending workgroup and peer) are made ONE virtual window space of
// 64-window groups by a directory: group g's windows are lanes 0 .. nvalid-1, lane l's k-mer begins at bit
//     base + 2 * (l + (k - 1) * popcount(starts & ((2 << l) - 1)))      of bases[]
// (base accounts for the runs in front of the group and stands k - 1 bases further in front, so that the sum needs no "- 1": a stream
// therefore never begins in the first 2 (k - 1) bits of bases[]; bit 0 = the most significant bit of bases[0]).  One 16-byte load
// per group.
struct RunGroup {
    u64 base_nvalid;     // base (56 bits) | nvalid << 56: windows in this group (64 except for a stream's last group)
    u64 starts;          // the group's start bits, lane l in bit l
};
constexpr u64 kRunBaseMask = (1ULL << 56) - 1ULL;
struct RunsInput {
    const u32 *bases = nullptr;
    const RunGroup *groups = nullptr;   // the groups of THIS launch (null = not a runs launch)
};

// ---- the early route's sender (superkmer_kernels.h) ----
constexpr int kSkThreads = 1024;   // threads of a split workgroup, one per CU (two of 512 threads per CU: +3 %, two of 1024 at 64 VGPRs: +5 %, measured twice); its tile = 16 window starts per thread
constexpr int kSkTile = 16 * kSkThreads;
constexpr u32 kSkMaxWorld = 64;   // owners a split can address (one byte per window in LDS, per-owner LDS staging)
struct SplitArgs {
    u32 world;
    uint4 *bases_out;      // [nwg][world][cap_units] 16-byte units of 64 bases
    uint4 *starts_out;     // [nwg][world][cap_sunits] 16-byte units of 128 start bits
    u32 cap_units, cap_sunits;
    u32 *nwin;             // [nwg][world] windows written
    u32 *nunits;           // [nwg][world] base units the stream needs (more than cap_units: *overflow is set, nothing beyond the region was written)
    u32 *nsunits;          // [nwg][world] start units likewise
    u32 *nruns;            // [nwg][world] runs (statistics)
    u64 *overflow;
    const u32 *pcodes; const unsigned short *pvalid;  // packed input (PartitionArgs::pcodes) instead of ASCII
    u32 ablate;            // KCT_ABLATE of a -DKCT_DEBUG_ENV build (tools/pmc_ablate.sh): phases skipped, results INVALID; 0 otherwise
};

// One received stream: its bases begin at bit bit0 of bases[], its start bits at starts[word0], it holds nwin windows, and its groups
// are group0 .. group0 + ceil(nwin / 64) - 1 of the owner's virtual window space.
struct RunStream {
    u64 bit0;
    u64 group0;
    u64 word0;
    u32 nwin;
};

struct PartitionArgs {
    u64 mask;            // table capacity - 1
    int block_bits;      // log2(slots per block)
    int pbits;           // log2(number of blocks P); P * D = kRingEntries, D >= 16
    u64 *scratch;        // [nwg][P][region_cap] hashes
    u32 region_cap;      // entries per (workgroup, block) region, multiple of kChunk
    u32 *region_count;   // [P][nwg] entries written (multiple of kChunk, zero-padded)
    u64 *ovf;            // [nwg][ovf_cap] hashes that found their ring (or region) full
    u32 ovf_cap;
    u32 *ovf_count;      // [nwg]
    u64 *overflow;       // set to 1 if an overflow region itself overflowed: the pass is abandoned
    int ablate;          // measurement only: bit 0 = skip the ring append, bit 1 = skip the flush phases
    // PACKED input ("packed base arrays", BASELINE north star): the record stream as 2-bit codes + validity bits, sixteen bases
    // per group -- codes[g] (first base in bits 31:30) and valid[g] (first base in bit 15), exactly what encode16 makes of the
    // ASCII stream.  When pcodes is set, `stream` is ignored and group g of this launch is pcodes[g] / pvalid[g].  (k <= 64.)
    const u32 *pcodes = nullptr;
    const unsigned short *pvalid = nullptr;
    // SUPER-K-MER input (the multi-GPU early route's wire format, superkmer_kernels.h): runs of consecutive good windows as
    // 2-bit bases with no separators, one start bit per window.  K1's RUNS instantiations walk the WINDOWS (64 per group), not
    // byte positions: `stream` / `nbytes` then only say how many (nbytes - k + 1 = 64 * groups of this launch).
    RunsInput runs;
    // -DKCT_K1_STAMPS builds only (tools/k1_stamps.sh): [nwg][16 waves][kStampSlots] shader-clock cycles per phase of K1, summed per wave
    u64 *stamps = nullptr;
};

struct RepartitionArgs {
    u64 mask;            // table capacity - 1
    int block_bits;      // log2(slots per block) (u64 / pair entries: the sub-bin is hash bits block_bits ...)
    int sub_bits;        // log2(blocks per super-bin); ring depth D = ring entries >> sub_bits >= 16
    const void *in;      // K1's regions: region (seg, s) at in + (seg * nbins + s) * in_cap entries
    u32 in_cap;
    const u32 *in_count; // [nbins][nseg]
    int nseg, nbins;
    int writers;         // workgroups per super-bin (W): each takes every W-th group of 16 input regions, so that
                         // W x nbins workgroups fill the chip even when there are few super-bins
    void *out;           // region of (block b, writer w) at out + (b * W + w) * out_cap entries
    u32 out_cap;         // multiple of a 64-byte line of entries
    u32 *out_count;      // [blocks][W]
    u64 *ovf; u32 ovf_cap; u32 *ovf_count;  // per workgroup overflow regions (u64 values; pairs: two words each)
    u64 *overflow;       // abandon flag (shared with K1)
    u64 *ovf_n;          // pairs only: ONE shared overflow list instead of per-workgroup regions (ovf_cap = its capacity)
    u32 min_lines;       // 64-byte lines of a bin that leave the ring together (1, 2 or 4; needs a ring depth of >= 4x that)
    const u64 *in_off = nullptr;  // optional [nbins][nseg]: region (seg, s) starts at in + in_off[s * nseg + seg] entries instead (packed
                                  // regions received from other GPUs, kct_route.hip)
    u32 bin0 = 0;        // compact entries: the first-level bin of super-bin 0
    // compact entries, shadows of fewer than 2^16 blocks: a super-bin is 2^gbits consecutive first-level bins (their regions: nseg =
    // workgroups << gbits of them), so that a second-level workgroup still spreads over >= 64 blocks -- the bin's low gbits bits are
    // the top bits of the sub-bin, above the entry's top sub_bits - gbits bits
    int gbits = 0;
    // several launches feeding ONE K2 pass (a pass cut into sub-chunks so that K1's scratch is reused, kct_consume.hip): this
    // launch's writers are slots writer0 .. writer0 + writers - 1 of the wtot regions every block has (0 = writers)
    int writer0 = 0, wtot = 0;
};

struct FailedBlocks {
    u32 *list = nullptr;   // block numbers, one per abandoned block
    u64 *n = nullptr;      // how many
    u64 *entries = nullptr;  // sum of their regions' entry counts
};

struct AggregateArgs {
    u64 *words;          // the table (block-SoA)
    int block_bits;
    int pbits;
    const u64 *scratch;  // region (seg, b) starts at scratch + seg * seg_stride + b * block_stride (u64 words)
    u64 seg_stride, block_stride;
    const u32 *region_count;  // [P][nregions] entries in each region
    int nregions;        // source regions per block: K1's workgroups (one level) or 1 (two levels)
    int fresh;           // table known empty: start every block from zeros instead of loading it
    const u64 *overflow; // K1's abandon flag
    int ablate;          // measurement only: bit 2 (4) = no count add, bit 4 (16) = loads only, bit 6 (64) = no streaming at all
    u32 nblocks;         // table blocks (the grid may be smaller: a workgroup then takes every grid-th block)
    FailedBlocks failed;
    u64 *counters;
};

struct Aggregate32Args {
    u32 *words;          // [blocks][S keys][S counts]
    int block_bits;
    const u32 *scratch;  // region (seg, b) at scratch + seg * seg_stride + b * block_stride (entries)
    u64 seg_stride, block_stride;
    const u32 *region_count;  // [blocks][nregions]
    int nregions;
    int fresh;
    const u64 *overflow; // K1's abandon flag
    FailedBlocks failed; // blocks that overflowed (abandoned whole; the host recounts their regions)
    u64 *counters;
    int ablate;          // measurement only: bit 4 (16) = loads only
    int sbits;           // log2(blocks of the shadow): a block index is the TOP sbits bits of the 42-bit value (>= 10)
    u32 nblocks;         // shadow blocks (the grid may be smaller: a workgroup then takes every grid-th block)
};

// A compact shadow block's place in the 42-bit value space: block b holds the values whose top 10 bits are bin0 + (b >> (sbits - 10))
// (bin0 = 0 and 2^sbits blocks on one GPU; an owner GPU of the early route holds the blocks of ITS range of bins).
struct FlushPartitionArgs {
    void *shadow;        // compact shadow: [1024 blocks][8192 u32 keys][8192 u32 counts]; 64-bit shadow: u64 keys and counts
    u32 shadow_blocks;   // 1024 for the one-level compact shadow
    int shadow_sbits;    // compact shadow: log2(shadow_blocks) (a block index is the top sbits bits of the 42-bit value)
    u32 shadow_bin0 = 0; // compact shadow: first-level bin of block 0
    const u64 *pair_keys, *pair_counts; int pair_stride; u64 npairs;  // SRC 2: a flat list of {hash, count} pairs instead of a shadow
    int k;
    int table_block_bits, pbits;  // the REAL table: slots per block, log2(blocks) (<= 10)
    ulonglong2 *scratch; // [nwg][P][region_cap] pairs
    u32 region_cap;      // pairs, multiple of 4
    u32 *region_count;   // [P][nwg]
    u64 *ovf; u64 ovf_cap; u64 *ovf_n;  // one shared list of pairs that found ring or region full (merge_pairs_kernel takes it)
};

struct AggregatePairsArgs {
    u64 *words; int block_bits;
    const ulonglong2 *scratch; u64 seg_stride, block_stride;  // region (seg, b) at scratch + seg * seg_stride + b * block_stride
    const u32 *region_count; int nregions;
    int fresh;
    u32 nblocks;         // table blocks (the grid may be smaller: a workgroup then takes every grid-th block)
    FailedBlocks failed; // blocks that overflowed (abandoned whole; the host grows the table and recounts their regions)
    u64 *counters;       // CTR_TOTAL_ADDED (counts placed), CTR_NEWKEYS
};

// ---- 128-bit dedupe-first path (33 <= k <= 64): a shadow of 1024 blocks x 4096 slots keyed by mix128 pairs {x, y} ----------------
constexpr int kBlockBits128 = 12;                   // 4096 slots: 32 KiB of x + 32 KiB of y + 16 KiB of counts = 80 KiB, one CU's LDS share
constexpr u32 kSlots128 = 1u << kBlockBits128;
constexpr u64 kBlockWords128 = (u64)kSlots128 * 2 + kSlots128 / 2;  // u64 words per block in HBM: x[S], y[S], count[S] (u32)
constexpr u32 kClaimed128 = 0x80000000u;            // count word, bit 31: the slot's y has been written (its x was claimed by CAS)
struct Aggregate128Args {
    u64 *words;          // [1024 blocks][x[S] | y[S] | count[S] u32]
    const ulonglong2 *scratch;  // region (seg, b) at scratch + seg * seg_stride + b * block_stride (entries)
    u64 seg_stride, block_stride;
    const u32 *region_count;    // [blocks][nregions]
    int nregions;
    int fresh;
    const u64 *overflow; // K1's abandon flag
    u32 nblocks;
    FailedBlocks failed;
    u64 *counters;       // CTR_COUNTED, CTR_NEW_BY_ZERO (new shadow keys)
};

struct PendingList {
    u64 *pairs = nullptr;  // 2 * cap words
    u64 cap = 0;
    u64 *n = nullptr;      // cursor (device), never reset between passes
};

}  // namespace kct
