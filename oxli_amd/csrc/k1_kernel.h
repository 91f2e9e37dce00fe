// k1_kernel.h -- the first kernel of the partitioned path: K1 partition_windows_kernel and the LDS write-combining ring it shares
// with the second partition level (partition_kernels.h has the overview).  Its own header so that the super-k-mer (RUNS)
// instantiations can live in a translation unit of their own (kct_runs.hip) -- every other kernel of the path is instantiated by
// kct_consume.hip only.
#pragma once
#include <type_traits>

#include "window_kernels.h"
#include "partition_args.h"

namespace kct {

// ---- where a K1 wave's time goes (measurement builds, -DKCT_K1_STAMPS: tools/k1_stamps.sh) -------------------------------------
// Every wave reads the shader clock (s_memtime) at the phase boundaries below and adds the cycles since its previous reading to the
// phase that just ended; the sums leave through PartitionArgs::stamps.  A "barrier" phase runs from the reading in front of
// __syncthreads() to the one behind it: the wave's own outstanding LDS / memory operations draining plus the wait for the slowest of
// the other fifteen waves.  The shipped build compiles NoStamps: nothing.
enum { ST_HASH = 0, ST_BAR_TILE, ST_STAGE, ST_BAR1, ST_LIST, ST_BAR2, ST_MOVE, ST_BAR3, ST_TAIL, kStampSlots };
struct NoStamps { __device__ __forceinline__ void mark(int) {} };
// -DKCT_ABLATE_FLUSH_BARRIERS (tools/k1_stamps.sh; results INVALID): ring_flush keeps its work and drops its three workgroup barriers --
// what the barriers themselves cost, as opposed to the work between them
#ifdef KCT_ABLATE_FLUSH_BARRIERS
#define KCT_FLUSH_BARRIER() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#else
#define KCT_FLUSH_BARRIER() __syncthreads()
#endif
#ifdef KCT_K1_STAMPS
struct WaveStamps {
    u64 last, acc[kStampSlots];
    __device__ __forceinline__ void start() { for (int i = 0; i < kStampSlots; ++i) acc[i] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i) { const u64 now = __builtin_amdgcn_s_memtime(); acc[i] += now - last; last = now; }
};
#endif

// ---- LDS write-combining ring shared by both partition levels ------------------------------------------
// Bin b owns ring[b*D .. b*D+D).  Its cursor word cur[b] holds `fill` (positions handed out) in the low half and
// `flushed` (positions that have left the ring) in the high half, so ONE 64-bit LDS add returns both.  An append takes
// position pos = fill++ and lands in slot pos % D unless the slot's previous tenant (position pos - D) has not left
// yet, i.e. pos - flushed >= D;
// such an append goes to the caller's overflow region and its position stays a HOLE in the sequence.
// Position p of bin b is stored at offset p of the bin's output region (zeros = padding / holes).
// ring_flush moves 64-byte lines (8 positions) out:
//   step 1  each bin's thread lists its ready lines.  Lines at or beyond flushed + D were never in the
//           ring (all holes): they are listed as ZERO lines, because their slots alias the lines in front;
//   step 2  groups of four adjacent lanes move one listed line, 16 bytes each: one whole-line store per
//           group, and only as many wave instructions as there are lines.
// Returns (workgroup-uniformly) whether the list was too short for everything that was ready.
// T = u64 (hashes / mix64 values, 8 per line) or u32 (the compact dedupe-first path: the bin number is the value's
// upper half, 16 per line); a value handed to overflow_hash is always the full 64-bit one.
// ovf_hi (u32 rings only): the upper half of a value handed to overflow_hash -- 0 = derive it from the bin (K1: the bin
// IS the value's top 10 bits), otherwise a fixed word (K1b: the super-bin, the sub-bin bits are inside the entry).
template <u32 LISTCAP, class T, class Overflow, class St = NoStamps>
__device__ __forceinline__ bool ring_flush(T *ring, u64 *cur, u32 *flist, u32 *fcount, int P, u32 D,
                                           T *out_base, u32 out_cap, bool drain, Overflow &&overflow_hash, u64 bin_stride = 0, u64 ovf_hi = 0,
                                           u32 min_lines = 1, u32 ovf_shift = 31, St *st = nullptr) {
    // ovf_shift (u32 rings, ovf_hi set): the bin's bits from ovf_shift up are added to the first-level bin in ovf_hi (K1b with grouped
    // super-bins: the sub-bin's top bits are the first-level bin's low bits)
    // min_lines: a bin's lines leave the ring only that many at a time (adjacent lane groups then store adjacent lines:
    // 128- or 256-byte writes instead of lone 64-byte ones); the drain takes whatever is left.
    constexpr u32 CH = 64 / sizeof(T);  // positions per 64-byte line
    const u32 dmask = D - 1;
    if (bin_stride == 0) bin_stride = out_cap;  // distance between the regions of consecutive bins
    const int dshift = __builtin_ctz(D);  // D is a power of two: shifts instead of quarter-rate multiplies
    if (st) st->mark(ST_HASH);
    KCT_FLUSH_BARRIER();  // appends of this interval are in the ring; *fcount == 0
    if (st) st->mark(ST_BAR1);
    for (int b = threadIdx.x; b < P; b += kPartThreads) {
        const u64 cw = cur[b];
        const u32 f0 = (u32)(cw >> 32), top = (u32)cw;
        u32 f = f0;
        while (drain ? (int)(top - f) > 0 : top - f >= CH * min_lines) {
            const u32 want = drain ? 1u : min_lines;
            const u32 slot = atomicAdd(fcount, want);
            const u32 fit = slot >= LISTCAP ? 0u : (LISTCAP - slot < want ? LISTCAP - slot : want);  // every slot below LISTCAP gets written
            for (u32 l = 0; l < fit; ++l, f += CH) flist[slot + l] = (u32)b | (f << 10) | (f - f0 >= D ? 0x80000000u : 0u);
            if (fit < want) break;  // list full: the remaining lines wait for the next call
        }
        if (f != f0) atomicAdd(&cur[b], (u64)(f - f0) << 32);  // (appenders bump the low half concurrently)
    }
    if (st) st->mark(ST_LIST);
    KCT_FLUSH_BARRIER();
    if (st) st->mark(ST_BAR2);
    const u32 listed = *fcount, nlist = listed < LISTCAP ? listed : LISTCAP;
    for (u32 item = threadIdx.x; item < 4 * nlist; item += kPartThreads) {
        const u32 e = flist[item >> 2], q = item & 3u, b = e & 1023u, f = (e >> 10) & 0x1FFFFFu;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(e >> 31)) {
            uint4 *src = reinterpret_cast<uint4 *>(&ring[(b << dshift) + (f & dmask)]) + q;
            v = *src;
            *src = make_uint4(0, 0, 0, 0);
        }
        if (f + CH <= out_cap) {
            reinterpret_cast<uint4 *>(out_base + (u64)b * bin_stride + f)[q] = v;
        } else if constexpr (sizeof(T) == 8) {  // region full (badly skewed input): hand the entries to the overflow region
            const u64 e0 = ((u64)v.y << 32) | v.x, e1 = ((u64)v.w << 32) | v.z;
            if (e0) overflow_hash(e0);
            if (e1) overflow_hash(e1);
        } else if constexpr (sizeof(T) == 16) {  // {hash, count} pairs
            const u64 h = ((u64)v.y << 32) | v.x, c = ((u64)v.w << 32) | v.z;
            if (h) overflow_hash(h, c);
        } else {
            const u64 hi = ovf_hi ? ovf_hi + ((u64)(b >> ovf_shift) << 32) : (((u64)b << 32) | (1ULL << 63));  // compact values travel with bit 63 set (0 stays "nothing")
            if (v.x) overflow_hash(hi | v.x);
            if (v.y) overflow_hash(hi | v.y);
            if (v.z) overflow_hash(hi | v.z);
            if (v.w) overflow_hash(hi | v.w);
        }
    }
    if (st) st->mark(ST_MOVE);
    KCT_FLUSH_BARRIER();
    if (st) st->mark(ST_BAR3);
    if (threadIdx.x == 0) *fcount = 0;
    return listed > LISTCAP;
}


// MODE 0: MurmurHash3 values (u64 entries).  MODE 1 (dedupe-first, k <= 32): mix64(packed k-mer + 1) values (u64).
// MODE 2 (compact dedupe-first, k <= 21): mix42(packed k-mer) values, of which the bin is the top 10 bits and the ring /
// the scratch regions carry only the low 32 (u32 entries: half the partition traffic, a ring twice as deep, so a flush
// every eight windows instead of four); a value whose low half is 0 -- the hole marker -- takes the overflow route.
// RUNS: the input is a stream of super-k-mers (PartitionArgs::runs, the multi-GPU early route's wire format) -- the tile walk is
// replaced by walk_windows_runs, everything behind the sink is the same.
// PACKED: the input is packed base arrays (PartitionArgs::pcodes / pvalid, 16-byte aligned) and the tile's words are fetched SIXTEEN BYTES per
// lane -- 256 lanes take four code words each, 128 lanes eight validity halves, six more the halo -- straight into the places of tcodes /
// tvalid they land in anyway: one wide load per lane of six waves instead of a 4-byte and a 2-byte load per lane of all sixteen.  (An
// instantiation without PACKED still reads packed arrays when pcodes is set, with the two narrow loads: k without a PACKED instantiation.)
// FE: windows between two flushes of the ring (0 = the mode's default: a quarter of the ring per interval -- 8 in compact mode, 4 with 8-byte entries,
// 2 with 16-byte ones).  FE = 4 in compact mode (an eighth of the ring per interval) is for inputs whose k-mers arrive in BURSTS -- position-sorted reads:
// every k-mer of a tile ~25 times within a few hundred windows -- which overflow a bin's 32-entry stretch between two flushes (round 6: the overflow
// route was half of such a pass); it costs the flush's fixed work twice as often.
template <int KW, int KC, int MODE = 0, bool RUNS = false, bool PACKED = false, int FE = 0>
__global__ __launch_bounds__(kPartThreads) void partition_windows_kernel(const unsigned char *__restrict__ stream, u64 nbytes, int k,
                                                                         u64 ntiles, PartitionArgs a) {
    static_assert(!RUNS || KW != 0, "super-k-mer input needs k <= 64");
    static_assert(!PACKED || (KW != 0 && !RUNS), "packed base arrays: k <= 64, not super-k-mers");
    // MODE 3 (33 <= k <= 64, dedupe-first): mix128 values of the two packed words as 16-byte {x, y} entries: bins and slots come from x.
    using T = typename std::conditional<MODE == 2, u32, typename std::conditional<MODE == 3, ulonglong2, u64>::type>::type;
    using PH = typename std::conditional<MODE == 3, u64, T>::type;  // the pending append's (first) word
    constexpr int kEntries = kRingEntries * 8 / sizeof(T);  // the ring is 128 KiB either way
    constexpr int kFlushEvery = FE > 0 ? FE : MODE == 2 ? 8 : MODE == 3 ? 2 : 4;  // windows between flushes: a quarter of the ring per interval
    __shared__ __attribute__((aligned(16))) T ring[kEntries];
    __shared__ u64 cur[1024];  // per bin: fill (low half) | flushed (high half)
    constexpr u32 kListCap = KW == 0 ? 1792 : 2048;  // the bytewise path's raw tile leaves a little less LDS
    __shared__ u32 flist[kListCap];  // lines ready to leave the ring: block | position << 10 | hole << 31
    __shared__ u32 fcount;
    __shared__ u32 ovf_n;
    // the staged tile: raw bytes for the bytewise path (k > 64), pre-encoded 2-bit words otherwise
    constexpr int kTileBytes = KW == 0 ? kPartTile + kHaloMax + 16 : 16;
    constexpr int kTileWords = (KW == 0 || RUNS) ? 4 : kPartThreads + 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kTileBytes];
    __shared__ __attribute__((aligned(16))) u32 tcodes[kTileWords];
    __shared__ __attribute__((aligned(16))) unsigned short tvalid[kTileWords];
    __shared__ __attribute__((aligned(16))) RunGroup tdesc[RUNS ? 2 * (kPartTile / 64) : 1];  // RUNS: the group descriptors of this tile and the next (two buffers, taking turns)
    __shared__ u32 ascii4[(KW == 0 || MODE != 0) ? 1 : 256];  // four packed bases -> four ASCII bytes (only the hashing mode needs it)
    if constexpr (KW != 0 && MODE == 0) fill_ascii4_lut(ascii4, threadIdx.x, kPartThreads);
    // MurmurHash3's first multiply of every whole 16-base block comes out of pre-multiplied tables (kmer_device.h):
    // K1 -1 % at k = 21, -2 % at k = 31, -4 % at k = 51
    constexpr bool kPremul = KW != 0 && MODE == 0 && (KC == 0 || KC >= 16);
    __shared__ u64 mul1[kPremul ? 256 : 1], mul2[kPremul ? 256 : 1];
    if constexpr (kPremul) fill_premul_luts(mul1, mul2, threadIdx.x, kPartThreads);
    const u64 *pm1 = kPremul ? mul1 : nullptr, *pm2 = kPremul ? mul2 : nullptr;
    // ... and the tail's first multiply, where k is known at compile time and the tail's last piece is cut short (kmer_device.h): K1 -3 %
    constexpr bool kTailLut = kPremul && KC > 0 && tail_needs_lut(KC);
    constexpr int kPre = (kPremul ? 1 : 0) | ((kPremul && KC > 0 && (KC & 15) != 0) ? 2 : 0);  // (a whole last piece needs no table of its own)
    __shared__ u64 tmul[kTailLut ? 256 : 1];
    if constexpr (kTailLut) fill_tail_lut(tmul, threadIdx.x, kPartThreads, KC);
    const u64 *ptm = kTailLut ? tmul : nullptr;
    const int P = 1 << a.pbits;
    const u32 D = (u32)(kEntries >> a.pbits), dmask = D - 1;
    const int dshift = __builtin_ctz((unsigned)kEntries) - a.pbits;  // log2 D
    static_assert(kRingEntries == 1 << 14, "dshift assumes a 16384-entry (u64) ring");
    {
        T zero;
        memset(&zero, 0, sizeof zero);
        for (int i = threadIdx.x; i < kEntries; i += kPartThreads) ring[i] = zero;
    }
    for (int i = threadIdx.x; i < 1024; i += kPartThreads) cur[i] = 0;
    if (threadIdx.x == 0) { ovf_n = 0; fcount = 0; }
    T *my_scratch = reinterpret_cast<T *>(a.scratch) + (u64)blockIdx.x * P * a.region_cap;  // region_cap counts entries

    // A hash whose ring slot is still occupied (many lanes hitting one block in the same few steps:
    // homopolymers, tandem repeats) or whose region is full goes to this workgroup's overflow
    // region: an LDS cursor and a plain 8-byte store, no global atomic, no cross-lane traffic.
    // The host folds those regions in afterwards with the direct atomic kernel, which combines
    // equal neighbours -- so the hot loop needs no duplicate detection at all.
    u64 *my_ovf = a.ovf + (u64)blockIdx.x * a.ovf_cap * (MODE == 3 ? 2 : 1);  // (MODE 3: two words per overflow entry)
    auto overflow_hash = [&](u64 h, u64 y = 0) {
        const u32 i = atomicAdd(&ovf_n, 1u);
        if (i < a.ovf_cap) {
            if constexpr (MODE == 3) { my_ovf[2 * i] = h; my_ovf[2 * i + 1] = y; }
            else my_ovf[i] = h;
        } else *a.overflow = 1ULL;
    };
#ifdef KCT_K1_STAMPS
    WaveStamps stamps;
    stamps.start();
    WaveStamps *stp = &stamps;
#else
    NoStamps *stp = nullptr;
#endif
    auto flush_lines = [&](bool drain) {
        return ring_flush<kListCap, T>(ring, cur, flist, &fcount, P, D, my_scratch, a.region_cap, drain, overflow_hash, 0, 0, 1, 31, stp);
    };

    // A tile is kPartTile + k - 1 <= 1024 + 16 sixteen-byte chunks: one per thread plus a halo that
    // the first 16 threads carry.  The NEXT tile's chunks are loaded into registers before the
    // current tile is hashed, so the HBM latency hides under ~100 us of hashing.
    auto load_chunk = [&](u64 tile_base, int c) -> uint4 {
        const u64 off = tile_base + 16ULL * (u64)c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(stream + off);
        else if (off < nbytes) {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = (off + i < nbytes) ? stream[off + i] : (unsigned char)0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        return v;
    };
    // packed input: a group's code word and validity bits travel in .x / .y of the same prefetch registers
    const bool packed = KW != 0 && a.pcodes != nullptr;  // (workgroup-uniform)
    auto load_group = [&](u64 tile_base, int c) -> uint4 {
        const u64 g = (tile_base >> 4) + (u64)c, off = g << 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off < nbytes) {
            v.x = a.pcodes[g];
            u32 vb = a.pvalid[g];
            if (off + 16 > nbytes) vb &= ~((1u << (16 - (u32)(nbytes - off))) - 1u);  // bases at or beyond nbytes do not exist
            v.y = vb;
        }
        return v;
    };
    auto load_any = [&](u64 tile_base, int c) -> uint4 { return packed ? load_group(tile_base, c) : load_chunk(tile_base, c); };
    // PACKED: this lane's 16-byte piece of a tile's packed words (zeros where the stream has ended; lanes without a piece: zeros).  The
    // piece's place: code words [4 t, 4 t + 4) for t < 256, validity halves [8 (t - 256), + 8) for t < 384, then the halo's sixteen groups
    // (t = 384..387: code words, 388..389: halves).
    const u64 pk_ng = (nbytes + 15) >> 4;   // groups this launch's arrays hold
    auto load_piece = [&](u64 tile_base) -> uint4 {
        const u32 t = threadIdx.x;
        const u64 g0 = tile_base >> 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < 256u || (t >= 384u && t < 388u)) {
            const u64 g = g0 + (t < 256u ? 4u * t : 1024u + 4u * (t - 384u));
            if (g + 4 <= pk_ng) v = *reinterpret_cast<const uint4 *>(a.pcodes + g);
            else if (g < pk_ng) { v.x = a.pcodes[g]; if (g + 1 < pk_ng) v.y = a.pcodes[g + 1]; if (g + 2 < pk_ng) v.z = a.pcodes[g + 2]; }
        } else if (t < 390u) {
            const u64 g = g0 + (t < 384u ? 8u * (t - 256u) : 1024u + 8u * (t - 388u));
            if (g + 8 <= pk_ng) v = *reinterpret_cast<const uint4 *>(a.pvalid + g);
            else if (g < pk_ng) {
                u32 w[4] = {0, 0, 0, 0};
                for (int j = 0; j < 8; ++j) if (g + j < pk_ng) w[j >> 1] |= (u32)a.pvalid[g + j] << (16 * (j & 1));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // the stream's last group is cut to nbytes (bases at or beyond it do not exist)
            if ((nbytes & 15) && g < pk_ng && g + 8 >= pk_ng) {
                const u32 j = (u32)(pk_ng - 1 - g), m = ~((1u << (16 - (u32)(nbytes & 15))) - 1u) & 0xFFFFu;
                const u32 keep = (j & 1) ? ((m << 16) | 0xFFFFu) : (0xFFFF0000u | m);
                if ((j >> 1) == 0) v.x &= keep; else if ((j >> 1) == 1) v.y &= keep; else if ((j >> 1) == 2) v.z &= keep; else v.w &= keep;
            }
        }
        return v;
    };
    // super-k-mer input: what is staged per tile is its 256 group descriptors (16 bytes each, one per thread of the first four waves)
    const u64 ngroups = RUNS ? (nbytes - (u64)k + 1 + 63) >> 6 : 0;
    auto load_desc = [&](u64 tile) -> uint4 {
        const u64 g = tile * (kPartTile / 64) + threadIdx.x;
        uint4 v = make_uint4(0, 0, 0, 0);   // (no group: no windows)
        if (threadIdx.x < kPartTile / 64 && g < ngroups) v = reinterpret_cast<const uint4 *>(a.runs.groups)[g];
        return v;
    };
    uint4 pre_main = make_uint4(0, 0, 0, 0), pre_halo = make_uint4(0, 0, 0, 0);
    // (pre_main: the descriptors of the workgroup's NEXT tile -- they go into the buffer that the current tile's walk reads its last
    // steps' prefetches from; pre_halo, first iteration only: the first tile's own)
    if (RUNS && blockIdx.x < ntiles) { pre_halo = load_desc(blockIdx.x); pre_main = load_desc((u64)blockIdx.x + gridDim.x); }
    RunsPipe<(KW > 0 ? KW : 1)> pipe;
    u32 iter = 0;
    if (PACKED && blockIdx.x < ntiles) pre_main = load_piece((u64)blockIdx.x * kPartTile);
    if (!RUNS && !PACKED && blockIdx.x < ntiles) {
        pre_main = load_any((u64)blockIdx.x * kPartTile, threadIdx.x);
        if (threadIdx.x < 16) pre_halo = load_any((u64)blockIdx.x * kPartTile, kPartThreads + threadIdx.x);
    }
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (stp) stp->mark(ST_HASH);
        __syncthreads();  // the previous tile's readers are done with `lds`; ring/fill init is visible
        if (stp) stp->mark(ST_BAR_TILE);
        if constexpr (RUNS) {
            if (threadIdx.x < kPartTile / 64) {
                if (iter == 0) reinterpret_cast<uint4 *>(tdesc)[threadIdx.x] = pre_halo;
                reinterpret_cast<uint4 *>(tdesc)[((iter + 1) & 1u) * (kPartTile / 64) + threadIdx.x] = pre_main;   // (that buffer's readers finished a tile ago)
            }
        } else if constexpr (KW == 0) {
            reinterpret_cast<uint4 *>(lds)[threadIdx.x] = pre_main;
            if (threadIdx.x < 16) reinterpret_cast<uint4 *>(lds)[kPartThreads + threadIdx.x] = pre_halo;
        } else if constexpr (PACKED) {
            const u32 t = threadIdx.x;
            if (t < 256u) reinterpret_cast<uint4 *>(tcodes)[t] = pre_main;
            else if (t < 384u) reinterpret_cast<uint4 *>(tvalid)[t - 256u] = pre_main;
            else if (t < 388u) reinterpret_cast<uint4 *>(tcodes + kPartThreads)[t - 384u] = pre_main;
            else if (t < 390u) reinterpret_cast<uint4 *>(tvalid + kPartThreads)[t - 388u] = pre_main;
        } else {
            u32 c, v;
            if (packed) { c = pre_main.x; v = pre_main.y; }
            else encode16(pre_main, c, v);
            tcodes[threadIdx.x] = c; tvalid[threadIdx.x] = (unsigned short)v;
            if (threadIdx.x < 16) {
                if (packed) { c = pre_halo.x; v = pre_halo.y; }
                else encode16(pre_halo, c, v);
                tcodes[kPartThreads + threadIdx.x] = c; tvalid[kPartThreads + threadIdx.x] = (unsigned short)v;
            }
        }
        __syncthreads();
        if (stp) stp->mark(ST_STAGE);
        const u64 next = tile + gridDim.x;
        if (RUNS) pre_main = load_desc(next + gridDim.x);   // (zeros beyond the last tile)
        if (PACKED && next < ntiles) pre_main = load_piece(next * kPartTile);
        if (!RUNS && !PACKED && next < ntiles) {
            pre_main = load_any(next * kPartTile, threadIdx.x);
            if (threadIdx.x < 16) pre_halo = load_any(next * kPartTile, kPartThreads + threadIdx.x);
        }
        // The append is software-pipelined: window j's ring cursor is bumped (ds_add_rtn_u32) and the
        // block's flush mark is read as soon as its hash exists, but the returned position is only
        // consumed -- and the hash written into the ring -- after window j+1 has been hashed, so the
        // LDS round trip hides under ~130 VALU instructions instead of stalling the wave.
        PH pend_h = 0;
        u64 pend_y = 0, aux_y = 0;  // (MODE 3: the companion word of the pending append / of the window just mixed)
        u32 pend_b = 0, pend_pos = 0, pend_mark = 0;
        auto commit = [&]() {
            if (pend_h) {
                if (pend_pos - pend_mark < D) {  // slot's previous tenant is flushed
                    if constexpr (MODE == 3) ring[(pend_b << dshift) + (pend_pos & dmask)] = make_ulonglong2(pend_h, pend_y);
                    else ring[(pend_b << dshift) + (pend_pos & dmask)] = pend_h;
                } else if constexpr (MODE == 3) overflow_hash(pend_h, pend_y);
                else overflow_hash(MODE == 2 ? (((u64)pend_b << 32) | pend_h | (1ULL << 63)) : (u64)pend_h);  // ring full: position stays a 0 hole
                pend_h = 0;
            }
        };
        auto sink = [&](int j, bool good, u64 h) {
            commit();  // the previous window's append
            if (MODE == 3 && good && h == 0 && !(a.ablate & 1)) overflow_hash(0ULL, aux_y);  // (x = 0, the ring's hole marker: one value in 2^64)
            if (good && h != 0 && !(a.ablate & 1)) {
                if (MODE == 2 && (u32)h == 0) overflow_hash(h);  // (one value in 2^32: its low half is the hole marker)
                else {
                    // bin = the pbits hash bits above the block (or super-bin) offset; the top bits in compact mode
if constexpr (MODE == 2) pend_b = (u32)(h >> 32) & 1023u;
                    else pend_b = (u32)(h >> a.block_bits) & (u32)(P - 1);
                    const u64 cw = atomicAdd(&cur[pend_b], 1ULL);
                    pend_pos = (u32)cw;
                    pend_mark = (u32)(cw >> 32);
                    pend_h = (PH)h;
                    if constexpr (MODE == 3) pend_y = aux_y;
                }
            }
            if ((j & (kFlushEvery - 1)) == kFlushEvery - 1 && !(a.ablate & 2)) {  // every fourth (eighth) step: move every full line out
                commit();
                flush_lines(false);
            }
        };
        if constexpr (RUNS) {
            walk_windows_runs<KW, KC, true, MODE, kPre>(a.runs, tdesc + (iter & 1u) * (kPartTile / 64), tdesc + ((iter + 1) & 1u) * (kPartTile / 64), pipe, iter == 0, k, sink,
                                                        ascii4, pm1, pm2, &aux_y, ptm);
            ++iter;
        }
        else if constexpr (KW == 0) walk_windows<0, 0, kPartWPT>(lds, k, sink);
        else walk_windows_encoded<KW, KC, true, MODE, kPre>(tcodes, tvalid, k, sink, ascii4, pm1, pm2, &aux_y, ptm);
        commit();
    }
    while (flush_lines(true)) {}  // drain: partial lines go out zero-padded; repeat while the list was too short
    for (int b = threadIdx.x; b < P; b += kPartThreads) {
        const u32 f = (u32)(cur[b] >> 32);
        a.region_count[(u64)b * gridDim.x + blockIdx.x] = f < a.region_cap ? f : a.region_cap;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.ovf_count[blockIdx.x] = ovf_n < a.ovf_cap ? ovf_n : a.ovf_cap;
#ifdef KCT_K1_STAMPS
    stamps.mark(ST_TAIL);
    if (a.stamps && (threadIdx.x & 63) == 0)
        for (int i = 0; i < kStampSlots; ++i) a.stamps[((u64)blockIdx.x * (kPartThreads / 64) + (threadIdx.x >> 6)) * kStampSlots + i] = stamps.acc[i];
#endif
}

}  // namespace kct
