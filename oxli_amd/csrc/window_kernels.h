// window_kernels.h -- tile staging, window walkers, and the kernels that hash every window of a record
// stream: hash-only, validity-only, and the direct (one HBM atomic per k-mer) counting kernel.
//
// Input layout ("record stream"): the bytes of all records of a batch back to back, each record
// followed by at least one byte that is not A/C/G/T (the host packer writes '\n').  A k-window
// is good iff its k bytes are all ACGT (either case), so windows that would span two records
// are bad by construction and no offsets are needed on the device.
//
// Decomposition: one 256-thread workgroup owns a tile of kTile consecutive window START
// positions.  It stages kTile + k - 1 bytes in LDS with 16-byte coalesced global loads, then
// every thread walks kWPT consecutive windows with a rolling 2-bit forward word and a rolling
// reverse-complement word (k - 1 warm-up steps).  Long reads need nothing special: a 10 kbp or
// 350 kbp record is simply many tiles.
#pragma once
#include "device_common.h"
#include "partition_args.h"

namespace kct {


// Stage stream[tile_base, tile_base + TILE + k - 1) into LDS; bytes past `nbytes` read as 0.
template <int BLOCK, int TILE>
__device__ __forceinline__ void stage_tile(const unsigned char *__restrict__ stream, u64 nbytes, u64 tile_base, int k,
                                           unsigned char *lds) {
    const int nchunks = (TILE + k - 1 + 15) >> 4;
    for (int c = threadIdx.x; c < nchunks; c += BLOCK) {
        const u64 off = tile_base + 16ULL * (u64)c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off + 16 <= nbytes) {
            v = *reinterpret_cast<const uint4 *>(stream + off);
        } else if (off < nbytes) {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = (off + i < nbytes) ? stream[off + i] : (unsigned char)0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        *reinterpret_cast<uint4 *>(lds + 16 * c) = v;
    }
    __syncthreads();
}

// Walks this thread's WPT windows of the staged tile and calls sink(j, good, hash) for each,
// j = 0..WPT-1 (window start = tile_base + threadIdx.x * WPT + j).  KC > 0 fixes k at compile
// time; KW = 64-bit words of the packed k-mer (k <= 32 * KW).  Every thread of the workgroup
// makes the same WPT calls, so a sink may use wave collectives and workgroup barriers.
template <int KW, int KC, int WPT, class Sink>
__device__ __forceinline__ void walk_windows_packed(const unsigned char *lds, int k_rt, Sink &&sink) {
    const int k = KC > 0 ? KC : k_rt;
    const unsigned char *p = lds + threadIdx.x * WPT;
    Packed<KW> fw, rc;
#pragma unroll
    for (int i = 0; i < KW; ++i) { fw.w[i] = 0; rc.w[i] = 0; }
    int run = 0;  // length of the run of valid bases ending at the current byte
    for (int j = 0; j < k - 1; ++j) {
        u32 code = base_code(p[j]);
        bool ok = code < 4;
        push_fw(fw, code & 3u);
        push_rc(rc, 3u - (code & 3u), k);
        run = ok ? run + 1 : 0;
    }
#pragma unroll 4
    for (int j = 0; j < WPT; ++j) {
        u32 code = base_code(p[k - 1 + j]);
        bool ok = code < 4;
        push_fw(fw, code & 3u);
        mask_k(fw, k);
        push_rc(rc, 3u - (code & 3u), k);
        run = ok ? run + 1 : 0;
        const bool good = run >= k;
        u64 h = 0;
        if (good) {
            Packed<KW> c = less_eq(fw, rc) ? fw : rc;
            left_align(c, k);
            h = hash_packed(c, k);
        }
        sink(j, good, h);
    }
}

// Same contract as walk_windows_packed, over a PRE-ENCODED tile (kmer_device.h encode16):
// codes[c] / valid[c] describe bases 16c .. 16c+15 of the tile.  The first window is assembled
// directly from the packed words (no k-1 warm-up steps); the WPT-1 following bases come out of two
// shift registers.  WPT must be 16 (one code word per thread).
// `lut` (LDS, kmer_device.h fill_ascii4_lut) switches the ASCII re-expansion to table look-ups.
// RAW = 1 (k <= 32 only): the sink receives mix64(packed canonical k-mer + 1) instead of the MurmurHash3 value --
// the dedupe-first path counts k-mers first and hashes each distinct one once (partition_kernels.h).
// RAW = 2 (k <= 21): mix42(packed canonical k-mer), a 42-bit value, with bit 63 set.
// RAW = 3 (33 <= k <= 64): mix128 of the two packed words: the sink receives x, the companion word y is left in *aux.
// walk_windows_words: the thread's NW = 2 KW + 1 code words (bases 0 .. 16 NW - 1 of its stretch) and their validity bits arrive in
// registers -- vbits: validity of the first 64 bases, base n in bit 63 - n (KW = 1 needs 47, KW = 2 needs 79); vtail: bases 64..79 in
// bits 15..0 (KW = 2 only).  walk_windows_encoded reads them from a staged tile; the wave-specialised K1 (k1ws_kernel.h) gets them
// from its neighbouring lanes.
template <int KW, int KC, bool LUT = false, int RAW = 0, int PRE = 0, class Sink>
__device__ __forceinline__ void walk_windows_words(const u32 (&w)[2 * KW + 1], u64 vbits, u32 vtail, int k_rt, Sink &&sink,
                                                   const u32 *lut = nullptr, const u64 *mul1 = nullptr, const u64 *mul2 = nullptr, u64 *aux = nullptr,
                                                   const u64 *tmul = nullptr);

template <int KW, int KC, bool LUT = false, int RAW = 0, int PRE = 0, class Sink>
__device__ __forceinline__ void walk_windows_encoded(const u32 *codes, const unsigned short *valid, int k_rt, Sink &&sink,
                                                     const u32 *lut = nullptr, const u64 *mul1 = nullptr, const u64 *mul2 = nullptr, u64 *aux = nullptr,
                                                     const u64 *tmul = nullptr) {
    constexpr int NW = 2 * KW + 1;  // code words covering bases 0 .. 15 + k
    u32 w[NW];
    u64 vbits = 0;
    u32 vtail = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        w[i] = codes[threadIdx.x + i];
        const u64 v = valid[threadIdx.x + i];
        if (i < 4) vbits |= v << (48 - 16 * i);
        else vtail = (u32)v;
    }
    walk_windows_words<KW, KC, LUT, RAW, PRE>(w, vbits, vtail, k_rt, sink, lut, mul1, mul2, aux, tmul);
}

template <int KW, int KC, bool LUT, int RAW, int PRE, class Sink>
__device__ __forceinline__ void walk_windows_words(const u32 (&w)[2 * KW + 1], u64 vbits, u32 vtail, int k_rt, Sink &&sink,
                                                   const u32 *lut, const u64 *mul1, const u64 *mul2, u64 *aux, const u64 *tmul) {
    constexpr int WPT = 16, NW = 2 * KW + 1;
    const int k = KC > 0 ? KC : k_rt;
    // ---- window 0: bases 0 .. k-1
    Packed<KW> fw;
#pragma unroll
    for (int i = 0; i < KW; ++i) fw.w[i] = ((u64)w[2 * i] << 32) | w[2 * i + 1];
    {   // shift right so that base k-1 sits in the low 2 bits
        const int s = 64 * KW - 2 * k, ws = s >> 6, bs = s & 63;
        Packed<KW> o;
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            u64 lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                if (j == i - ws) lo = fw.w[j];
                if (j == i - ws - 1) hi = fw.w[j];
            }
            o.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        }
        fw = o;
    }
    Packed<KW> rc = revcomp_packed(fw, k);
    // run = valid bases in a row ending at base k-1
    int run;
    {
        u64 inv_hi = ~vbits;                 // invalid bases among 0..63
        if (k < 64) inv_hi &= ~0ULL << (64 - k);  // keep bases 0..k-1 only
        // the last invalid base before k: its distance to base k-1
        run = inv_hi ? (int)__builtin_ctzll(inv_hi) - (64 - k) : k;
        if (k > 64) run = k;  // unreachable for KW <= 2 (k <= 64)
    }
    // ---- streams of the WPT-1 bases k .. k+WPT-2 (and their validity), next one in the top bits
    u32 cs, vs;
    {
        const int idx = k >> 4, sh = 2 * (k & 15);
        u32 a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { if (i == idx) a = w[i]; if (i == idx + 1) b = w[i]; }
        cs = sh ? ((a << sh) | (b >> (32 - sh))) : a;
        // validity of base n: n < 64 -> vbits bit 63-n, else vtail bit 79-n
        const u64 v_lo = k < 64 ? (vbits << k) : 0ULL;                       // base k at bit 63
        const u64 v_hi = k < 64 ? ((u64)vtail << 48) >> (64 - k) : (u64)vtail << 48;  // bases 64.. follow
        vs = (u32)((v_lo | (k ? v_hi : 0ULL)) >> 32);
    }
    // fully unrolled where the window's work is long (hashing: K1 -2 %) or the loop's own branches weigh (compact: -3 %; the
    // every-eighth-window flush test becomes static); the 64-bit raw mode is 2 % faster unrolled by four
    constexpr int kUnroll = (RAW == 1 || RAW == 3) ? 4 : WPT;
#pragma unroll kUnroll
    for (int j = 0; j < WPT; ++j) {
        const bool good = run >= k;
        u64 h = 0;
        if (good) {
            Packed<KW> c = less_eq(fw, rc) ? fw : rc;
            if constexpr (RAW == 1) {
                static_assert(KW == 1, "the dedupe-first path carries one 64-bit word per k-mer");
                h = mix64(c.w[0] + 1ULL);
            } else if constexpr (RAW == 2) {
                static_assert(KW == 1 && (KC == 0 || KC <= 21), "the compact dedupe-first path needs 2k <= 42 bits");
                h = mix42(c.w[0]) | (1ULL << 63);  // bit 63: "not the zero hash" for the sink's h != 0 test; the sink drops it
            } else if constexpr (RAW == 3) {
                static_assert(KW == 2, "two packed words per k-mer");
                mix128(c.w[0], c.w[1], h, *aux);
            } else {
                left_align(c, k);
                h = hash_packed<KW, LUT, PRE>(c, k, lut, mul1, mul2, tmul);
            }
        }
        sink(j, good, h);
        if (j + 1 < WPT) {
            const u32 code = cs >> 30;
            cs <<= 2;
            const bool ok = (int)vs < 0;
            vs <<= 1;
            push_fw(fw, code);
            mask_k(fw, k);
            push_rc(rc, 3u - code, k);
            run = ok ? run + 1 : 0;
        }
    }
}

// Same contract for a SUPER-K-MER launch (partition_args.h RunsInput; the multi-GPU early route's wire format): the thread's WPT
// windows are window (threadIdx.x & 63) of groups group0 + wave, + 16, + 32, ... -- a wave takes one 64-window group per step.  Every
// window is assembled from its run's packed bases (no rolling: there is no "previous window" to roll from at a run start, and
// assembling costs what rolling does); the lanes of a wave read overlapping words, which the L1 serves.  Two dependent loads stand
// between a step and its k-mer (the group's descriptor -- wave-uniform -- then the bases), so the loop is software-pipelined: the
// descriptor of step j + 3 and the bases of step j + 2 are requested before window j is hashed -- across tiles too (RunsPipe).
// What the walk keeps in flight from one tile to the next: the descriptors of the next tile's first three steps and the bases of its
// first two are requested during the current tile's last steps, so that a tile does not start with two dependent round trips (LDS, then
// HBM) that all sixteen waves would sit out together behind the tile's barrier.
constexpr int kRunsFetchAhead = 2;   // steps between requesting a window's bases and using them (its descriptor: one more; 3: no faster)
template <int KW>
struct RunsPipe {
    RunGroup d[kRunsFetchAhead + 1];
    u32 x[kRunsFetchAhead][2 * KW + 1];
    u32 sh[kRunsFetchAhead];
    bool good[kRunsFetchAhead];
};

template <int KW, int KC, bool LUT = false, int RAW = 0, int PRE = 0, class Sink>
__device__ __forceinline__ void walk_windows_runs(const RunsInput &in, const RunGroup *tdesc, const RunGroup *tdesc_next, RunsPipe<KW> &pipe, bool first,
                                                  int k_rt, Sink &&sink, const u32 *lut = nullptr,
                                                  const u64 *mul1 = nullptr, const u64 *mul2 = nullptr, u64 *aux = nullptr, const u64 *tmul = nullptr) {
    constexpr int WPT = 16, NX = 2 * KW + 1;  // words that hold 2k bits at any 2-bit offset
    const int k = KC > 0 ? KC : k_rt;
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = threadIdx.x >> 6;
    constexpr u32 kStep = kPartThreads / 64;
    struct Fetch { u32 x[NX]; u32 sh; bool good; };
    // The tile's 256 descriptors sit in LDS (`tdesc`, staged by the kernel: group wave + 16 j of the tile is step j of this wave; groups
    // past the end are zeros = no windows; `tdesc_next`: the following tile's).  One broadcast ds_read_b128 + four v_readfirstlane make
    // a descriptor wave-uniform, so its arithmetic runs on the scalar unit.  (Scalar LOADS of the descriptors -- s_load_dwordx4 through
    // the constant address space, the first version -- share the lgkm counter with the append's LDS atomics and return out of order:
    // every wait for an LDS result became lgkmcnt(0) and exposed a scalar-cache round trip per step.)
    auto describe = [&](int j) -> RunGroup {   // j >= WPT: step j - WPT of the next tile
        const uint4 q = reinterpret_cast<const uint4 *>(j < WPT ? tdesc : tdesc_next)[wave + kStep * (u32)(j < WPT ? j : j - WPT)];
        RunGroup d;
        d.base_nvalid = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)q.y) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)q.x);
        d.starts = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)q.w) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)q.z);
        return d;
    };
    // (no branches: a lane without a window -- the tail of a stream's last group, groups past the end -- reads the group's first word
    // and its "k-mer" is dropped by the sink; nearly every lane has one)
    auto fetch = [&](const RunGroup &d) -> Fetch {
        Fetch f;
        f.good = lane < (u32)(d.base_nvalid >> 56);
        // run starts at or before this window: the lanes below (v_mbcnt: two instructions) + its own bit
        const u32 own = (u32)(d.starts >> lane) & 1u;
        const u32 runs = __builtin_amdgcn_mbcnt_hi((u32)(d.starts >> 32), __builtin_amdgcn_mbcnt_lo((u32)d.starts, own));
        // the group's base word is wave-uniform (scalar arithmetic); the lane's distance from it -- never negative, the base stands
        // k - 1 bases in front of the group's first window's run (partition_args.h) -- fits 32 bits
        const u64 base = d.base_nvalid & kRunBaseMask;
        u32 rel = ((u32)base & 31u) + 2u * (lane + (u32)(k - 1) * runs);
        rel = f.good ? rel : 0u;
        const u32 *p = in.bases + (base >> 5) + (rel >> 5);
#pragma unroll
        for (int i = 0; i < NX; ++i) f.x[i] = p[i];
        f.sh = rel & 31u;
        return f;
    };
    constexpr int FA = kRunsFetchAhead, DA = FA + 1;
    RunGroup D[WPT + DA];
    Fetch F[WPT + FA];
    if (first) {   // (workgroup-uniform: the workgroup's first tile)
#pragma unroll
        for (int i = 0; i < DA; ++i) D[i] = describe(i);
#pragma unroll
        for (int i = 0; i < FA; ++i) F[i] = fetch(D[i]);
    } else {
#pragma unroll
        for (int i = 0; i < DA; ++i) D[i] = pipe.d[i];
#pragma unroll
        for (int i = 0; i < FA; ++i) {
#pragma unroll
            for (int w = 0; w < NX; ++w) F[i].x[w] = pipe.x[i][w];
            F[i].sh = pipe.sh[i]; F[i].good = pipe.good[i];
        }
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        D[j + DA] = describe(j + DA);
        F[j + FA] = fetch(D[j + FA]);
        const Fetch &cur = F[j];
        u64 h;
        {
            // the 2k bits from bit cur.sh of x[], left-aligned: base 0 in bits 63:62 of top.w[0]
            u32 t32[2 * KW];
#pragma unroll
            for (int i = 0; i < 2 * KW; ++i) t32[i] = (u32)(((((u64)cur.x[i] << 32) | cur.x[i + 1]) << cur.sh) >> 32);
            Packed<KW> top, fw, rc;
#pragma unroll
            for (int i = 0; i < KW; ++i) top.w[i] = ((u64)t32[2 * i] << 32) | t32[2 * i + 1];
            {   // forward strand: shift right so that base k-1 sits in the low 2 bits
                const int s = 64 * KW - 2 * k, ws = s >> 6, bs = s & 63;
#pragma unroll
                for (int i = 0; i < KW; ++i) {
                    u64 lo = 0, hi = 0;
#pragma unroll
                    for (int jj = 0; jj < KW; ++jj) {
                        if (jj == i - ws) lo = top.w[jj];
                        if (jj == i - ws - 1) hi = top.w[jj];
                    }
                    fw.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
                }
            }
#pragma unroll
            for (int i = 0; i < KW; ++i) rc.w[i] = ~reverse_pairs64(top.w[KW - 1 - i]);  // (whatever follows base k-1 in `top` lands above bit 2k)
            mask_k(rc, k);
            Packed<KW> c = less_eq(fw, rc) ? fw : rc;
            if constexpr (RAW == 1) h = mix64(c.w[0] + 1ULL);
            else if constexpr (RAW == 2) h = mix42(c.w[0]) | (1ULL << 63);
            else if constexpr (RAW == 3) mix128(c.w[0], c.w[1], h, *aux);
            else {
                left_align(c, k);
                h = hash_packed<KW, LUT, PRE>(c, k, lut, mul1, mul2, tmul);
            }
        }
        sink(j, cur.good, h);
    }
#pragma unroll
    for (int i = 0; i < DA; ++i) pipe.d[i] = D[WPT + i];
#pragma unroll
    for (int i = 0; i < FA; ++i) {
#pragma unroll
        for (int w = 0; w < NX; ++w) pipe.x[i][w] = F[WPT + i].x[w];
        pipe.sh[i] = F[WPT + i].sh; pipe.good[i] = F[WPT + i].good;
    }
}

// Any k (used for k > 64): validity by run length, canonical choice and hashing bytewise.
template <int WPT, class Sink>
__device__ __forceinline__ void walk_windows_bytes(const unsigned char *lds, int k, Sink &&sink) {
    const unsigned char *p = lds + threadIdx.x * WPT;
    int run = 0;
    for (int j = 0; j < k - 1; ++j) run = base_code(p[j]) < 4 ? run + 1 : 0;
    for (int j = 0; j < WPT; ++j) {
        run = base_code(p[k - 1 + j]) < 4 ? run + 1 : 0;
        const bool good = run >= k;
        u64 h = good ? hash_bytes_canonical(p + j, k) : 0;
        sink(j, good, h);
    }
}

template <int KW, int KC, int WPT, class Sink>
__device__ __forceinline__ void walk_windows(const unsigned char *lds, int k, Sink &&sink) {
    if constexpr (KW == 0) walk_windows_bytes<WPT>(lds, k, sink);
    else walk_windows_packed<KW, KC, WPT>(lds, k, sink);
}

// ---- hash-only kernel: SeqToHashes as consume drives it (lib.rs:576-600) ------------------------
// out[p] = hash of the window starting at p (0 if bad), p in [0, nwindows);
// *first_bad = min index of a bad window (left untouched if none).
template <int KW, int KC>
__global__ __launch_bounds__(kBlock) void hash_windows_kernel(const unsigned char *__restrict__ stream, u64 nbytes, int k,
                                                              u64 nwindows, u64 *__restrict__ out, u64 *first_bad) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kTile + kHaloMax + 16];
    const u64 tile_base = (u64)blockIdx.x * kTile;
    stage_tile<kBlock, kTile>(stream, nbytes, tile_base, k, lds);
    const u64 p0 = tile_base + (u64)threadIdx.x * kWPT;
    u64 my_bad = ~0ULL;
    walk_windows<KW, KC, kWPT>(lds, k, [&](int j, bool good, u64 h) {
        const u64 p = p0 + j;
        if (p < nwindows) {
            out[p] = good ? h : 0;
            if (!good && my_bad == ~0ULL) my_bad = p;
        }
    });
    if (my_bad != ~0ULL) atomicMin(first_bad, my_bad);
}

// ---- the hot kernel: windows -> canonical hash -> scatter-increment -------------------------------
// consume's loop body (lib.rs:586-600) for every window of the stream at once.
template <int KW, int KC>
__global__ __launch_bounds__(kBlock) void count_windows_kernel(const unsigned char *__restrict__ stream, u64 nbytes, int k,
                                                               TableView table, u64 *counters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kTile + kHaloMax + 16];
    __shared__ u64 s_counted, s_new;
    if (threadIdx.x == 0) { s_counted = 0; s_new = 0; }
    const u64 tile_base = (u64)blockIdx.x * kTile;
    stage_tile<kBlock, kTile>(stream, nbytes, tile_base, k, lds);
    const int lane = threadIdx.x & 63;
    int counted = 0;  // signed: a leader whose folded add spills takes back the folded lanes' tallies
    int newkeys = 0;
    walk_windows<KW, KC, kWPT>(lds, k, [&](int, bool good, u64 h) {
        bool active = good && h != 0;  // lib.rs:589: a hash of 0 is skipped and not tallied
        u64 c = 1;
        int tally = 0;
        // Wavefront combining: lanes whose hash equals the first active lane's hash fold into
        // one add.  Tandem repeats and homopolymers put the same k-mer in every lane at once
        // (lanes are kWPT windows apart), which would otherwise serialise on one HBM atomic.
        const u64 act = __ballot(active);
        if (act) {
            const int leader = __ffsll((long long)act) - 1;
            const u64 hl = read_lane64(h, leader);
            const u64 same = __ballot(active && h == hl);
            if (same != (1ULL << leader)) {
                if (lane == leader) c = (u64)__popcll(same);
                else if ((same >> lane) & 1ULL) { active = false; tally = 1; }
            }
        }
        if (active) {
            const AddResult r = table_add<false>(table, h, c);
            // spilled entries are tallied when the host replays them: a leader whose folded
            // add spilled also takes back the folded lanes' tallies (the replay adds c)
            tally = r.spilled ? 1 - (int)c : 1;
            newkeys += r.claimed ? 1 : 0;
        }
        counted += tally;
    });
    u64 wc = wave_sum((u64)(long long)counted), wn = wave_sum((u64)newkeys);
    if (lane == 0) { atomicAdd(&s_counted, wc); atomicAdd(&s_new, wn); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *shard = counters + (blockIdx.x % kCounterShards) * kCounterStride;
        if (s_counted) atomicAdd(shard + CTR_COUNTED, s_counted);
        if (s_new) atomicAdd(shard + CTR_NEWKEYS, s_new);
    }
}

}  // namespace kct
