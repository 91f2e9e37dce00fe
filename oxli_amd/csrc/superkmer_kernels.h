// superkmer_kernels.h -- the multi-GPU early route's SENDER side: cut a record stream into super-k-mers grouped by owner GPU.
//
// Why: every window's k-mer must reach the GPU that owns it (add() semantics, lib.rs:778-837: per-key sums, so the key space is
// partitioned).  Sending one 4- or 8-byte entry per window makes the job xGMI-bound.  Consecutive windows of a read
// overlap in k - 1 bases, so the route sends BASES instead: the owner of a k-mer is a function of its MINIMISER -- the smallest (in a
// scrambled order) canonical m-mer inside it, m = 8 -- which consecutive windows mostly share; a maximal run of windows with one owner
// travels as n + k - 1 bases at 2 bits each plus one start bit per window: ~1 byte per window at k = 21 instead of 4, ~0.9 at k = 51
// instead of 8 (a run is longer there, but drags k - 1 bases along: ~3 bases per window at every k).  The minimiser is taken over canonical m-mers, so a k-mer and its reverse complement (the same key, lib.rs:576-584 via
// sourmash's canonical min) have the same owner whatever strand a read shows.
//
//   split_superkmers_kernel<K>   persistent, one 1024-thread workgroup per CU, tiles of 16,384 window starts (as K1):
//        encode the tile (2-bit codes + validity), scramble every canonical m-mer into LDS, sliding minimum over the k - m + 1 m-mers of
//        every window (packed 16-bit minima, log steps), owner = (hash(minimiser) >> 6) * world >> 10 (sk_owner), runs = maximal stretches of good
//        windows with one owner; every run gets its place in its owner's LDS staging by ONE ds_add_rtn_u64 (windows | bases << 32) and
//        is copied there with a few ds_or_b32; whole 16-byte units leave for this workgroup's private region of the owner (no global
//        atomics), the partial unit stays in LDS for the next tile.
//   gather_units_kernel          packs the (workgroup, owner) regions into one contiguous slab per owner (the send buffer)
//   run_directory_kernel         OWNER side: one RunGroup per 64 windows of every received stream (partition_args.h), so that K1's RUNS
//                                instantiations find any window's bases with one popcount
//   expand_runs_kernel           OWNER side fallback: windows -> an ASCII record stream (k bytes + '\n' each) for the kernels that read
//                                bytes (the direct path on small passes)
#pragma once
#include "partition_args.h"

namespace kct {

constexpr int kSkM = 8;                       // minimiser length; a packed m-mer is 2m = 16 bits
constexpr int kSkStageWords = 12 * 1024;      // LDS staging of outgoing bases, all owners together: 48 KiB = 196,608 bases
constexpr u32 kSkMaxRun = 1024;               // windows: a longer run is cut (bounds the copy one thread makes)
constexpr u32 kSkUnitBases = 64, kSkUnitWindows = 128;  // a 16-byte unit of bases / of start bits

// a bijection of [0, 2^16): the order in which m-mers are compared (so that poly-A is not everybody's minimiser)
__device__ __host__ __forceinline__ u32 sk_scramble(u32 x) {
    x = (x * 0x9E3Bu) & 0xFFFFu; x ^= x >> 7;
    x = (x * 0x6A75u) & 0xFFFFu; x ^= x >> 9;
    return x;
}
// minima crowd near 0: spread them over the owners with one more odd multiply
// owner of a scrambled minimiser: the top TEN bits of its 16-bit hash spread over the ranks (world <= 64, so the product fits the 16-bit lanes
// of sk_pk_owner, which is this function on both halves of a word: the kernel's rule and the documented one are the same arithmetic)
__device__ __host__ __forceinline__ u32 sk_owner(u32 minimiser, u32 world) { return ((((minimiser * 0x9E37u) & 0xFFFFu) >> 6) * world) >> 10; }

__device__ __forceinline__ u32 sk_pkmin(u32 a, u32 b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 r = __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(u32, r);
}

// A[] holds NV = 16 + W - 1 values as 16-bit pairs (value 2j in the low half of A[j]); afterwards the low/high halves of A[0..7] are
// min(v[i .. i + W - 1]) for i = 0..15.  Doubling steps on packed pairs: shifting by an odd number of values costs one v_alignbit.
template <int W, int NP>
__device__ __forceinline__ void sk_sliding_min(u32 (&A)[NP]) {
    int s = 1;
    if constexpr (W >= 2) {
#pragma unroll
        for (int j = 0; j < NP; ++j) A[j] = sk_pkmin(A[j], __builtin_amdgcn_alignbit(j + 1 < NP ? A[j + 1] : 0xFFFFFFFFu, A[j], 16));
        s = 2;
    }
#pragma unroll
    for (int lvl = 0; lvl < 5; ++lvl) {
        if (2 * s <= W) {
#pragma unroll
            for (int j = 0; j < NP; ++j) if (j + s / 2 < NP) A[j] = sk_pkmin(A[j], A[j + s / 2]);
            s *= 2;
        }
    }
    const int d = W - s;  // min over W = min(min over s at i, min over s at i + d)
    if (d > 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (d % 2 == 0) { if (j + d / 2 < NP) A[j] = sk_pkmin(A[j], A[j + d / 2]); }
            else if (j + (d + 1) / 2 < NP) A[j] = sk_pkmin(A[j], __builtin_amdgcn_alignbit(A[j + (d + 1) / 2], A[j + (d - 1) / 2], 16));
        }
    }
}

// 16-bit lanes of a 32-bit word (v_pk_* instructions)
typedef unsigned short sk_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 sk_pk_scramble(u32 x) {  // sk_scramble on both halves
    sk_us2 v = __builtin_bit_cast(sk_us2, x);
    v *= (unsigned short)0x9E3Bu; v ^= v >> (unsigned short)7;
    v *= (unsigned short)0x6A75u; v ^= v >> (unsigned short)9;
    return __builtin_bit_cast(u32, v);
}
__device__ __forceinline__ u32 sk_pk_owner(u32 x, u32 world) {  // sk_owner on both halves (world <= 64: ten bits of the product suffice)
    sk_us2 v = __builtin_bit_cast(sk_us2, x);
    v *= (unsigned short)0x9E37u; v >>= (unsigned short)6;
    v *= (unsigned short)world; v >>= (unsigned short)10;
    return __builtin_bit_cast(u32, v);
}
// flags in bit 7 of the four bytes of f -> bits 0..3
__device__ __forceinline__ u32 sk_gather4(u32 f) { return (((f >> 7) * 0x00204081u) >> 21) & 0xFu; }
// 0x80 in every byte of d that is not zero
__device__ __forceinline__ u32 sk_nonzero_bytes(u32 d) { return (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u; }

template <int K>
__global__ __launch_bounds__(kSkThreads) void split_superkmers_kernel(const unsigned char *__restrict__ stream, u64 nbytes, u64 ntiles, SplitArgs a) {
    constexpr int M = K < kSkM ? K : kSkM, W = K - M + 1;
    constexpr u32 MM = (1u << (2 * M)) - 1u;
    constexpr int NV = 16 + W - 1, NP = (NV + 1) / 2;
    constexpr int HT = (W - 1 + 15) / 16;                       // threads that also scramble the halo's m-mers
    constexpr int kMmWords = kSkTile / 2 + 8 * HT + 8;
    __shared__ u32 tc[1 + kSkThreads + 16 + 2];               // tile codes behind one zero word (a copy may look 32 bits to the left)
    __shared__ unsigned short tv[kSkThreads + 16];
    __shared__ __attribute__((aligned(16))) u32 mm[kMmWords];   // scrambled canonical m-mers, 16 bits each
    __shared__ unsigned short edge[kSkThreads];                 // a thread's first | last << 8 owner byte: what its neighbours need of it
    __shared__ unsigned short emask[kSkThreads + 1];
    __shared__ __attribute__((aligned(16))) u32 sb[kSkStageWords];      // outgoing bases, [world][capBw]
    __shared__ __attribute__((aligned(16))) u32 ss[kSkStageWords / 2];  // outgoing start bits, [world][capBw / 2]
    __shared__ u64 cur[kSkMaxWorld];       // per owner: windows (low half) | bases (high half) staged, carry included
    __shared__ u64 cur0[kSkMaxWorld];      // ... as it stood when the tile began (the carry alone)
    __shared__ __attribute__((aligned(16))) uint4 carryB[kSkMaxWorld], carryS[kSkMaxWorld];  // the carried partial units, for a tile that has to be redone
    __shared__ u32 filledB[kSkMaxWorld], filledS[kSkMaxWorld];
    __shared__ u32 s_over, s_runs, nlist;   // nlist: runs of the current tile in the run list (below)
    const u32 world = a.world, t = threadIdx.x;
    const u32 capBw = (u32)(kSkStageWords / world) & ~7u;      // staging words per owner (whole 16-byte units of bases AND of start bits)
    const u32 capB = capBw * 16u;                              // ... in bases, and as many window bits
    const u32 capU = capBw / 4, capSU = capBw / 8;             // ... in units
    const u32 invU = 0xFFFFFFFFu / capU + 1u, invSU = 0xFFFFFFFFu / capSU + 1u;   // unit index -> owner by a multiply (indices < 2^12)
    // a sub-tile of g windows always fits: at most g / 2 runs of one owner (+ the few that kSkMaxRun cuts), K bases each, behind a
    // carry of < 64 bases.  gsafe = the largest power of two g with (g / 2) K + 16 K + 64 <= capB (16 holds for every world <= 64)
    u32 gsafe = 16;
    while (gsafe < (u32)kSkTile && (u64)gsafe * K + 16 * K + 64 <= capB) gsafe <<= 1;
    for (u32 i = t; i < (u32)kSkStageWords; i += kSkThreads) sb[i] = 0;
    for (u32 i = t; i < (u32)kSkStageWords / 2; i += kSkThreads) ss[i] = 0;
    if (t < kSkMaxWorld) { cur[t] = 0; cur0[t] = 0; filledB[t] = 0; filledS[t] = 0; carryB[t] = make_uint4(0, 0, 0, 0); carryS[t] = make_uint4(0, 0, 0, 0); }
    if (t == 0) { tc[0] = 0; emask[kSkThreads] = 0xFFFF; s_over = 0; s_runs = 0; }
    uint4 *my_bases = a.bases_out + (u64)blockIdx.x * world * a.cap_units;
    uint4 *my_starts = a.starts_out + (u64)blockIdx.x * world * a.cap_sunits;
    u32 my_runs = 0;

    auto load_chunk = [&](u64 tile_base, int c) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a.pcodes) {
            const u64 g = (tile_base >> 4) + (u64)c, off = g << 4;
            if (off < nbytes) {
                v.x = a.pcodes[g];
                u32 vb = a.pvalid[g];
                if (off + 16 > nbytes) vb &= ~((1u << (16 - (u32)(nbytes - off))) - 1u);
                v.y = vb;
            }
            return v;
        }
        const u64 off = tile_base + 16ULL * (u64)c;
        if (off + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(stream + off);
        else if (off < nbytes) {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = (off + i < nbytes) ? stream[off + i] : (unsigned char)0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        return v;
    };
    uint4 pre_main = make_uint4(0, 0, 0, 0), pre_halo = make_uint4(0, 0, 0, 0);
    if (blockIdx.x < ntiles) {
        pre_main = load_chunk((u64)blockIdx.x * kSkTile, t);
        if (t < 16) pre_halo = load_chunk((u64)blockIdx.x * kSkTile, kSkThreads + t);
    }
    __syncthreads();
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- the tile as 2-bit codes + validity bits -------------------------------------------------------------------------------
        {
            u32 c, v;
            if (a.pcodes) { c = pre_main.x; v = pre_main.y; } else encode16(pre_main, c, v);
            tc[1 + t] = c; tv[t] = (unsigned short)v;
            if (t < 16) {
                if (a.pcodes) { c = pre_halo.x; v = pre_halo.y; } else encode16(pre_halo, c, v);
                tc[1 + kSkThreads + t] = c; tv[kSkThreads + t] = (unsigned short)v;
            }
            if (t == 0) { tc[1 + kSkThreads + 16] = 0; tc[1 + kSkThreads + 17] = 0; nlist = 0; }
        }
        __syncthreads();
        {
            const u64 next = tile + gridDim.x;
            if (next < ntiles) {
                pre_main = load_chunk(next * kSkTile, t);
                if (t < 16) pre_halo = load_chunk(next * kSkTile, kSkThreads + t);
            }
        }
        // ---- every m-mer of the tile, canonical and scrambled: out8[i] = value 2 i | value 2 i + 1 << 16 -----------------------------
        auto scramble16 = [&](u32 c0, u32 c1, u32 (&out8)[8]) {  // the m-mers starting at the 16 bases of c0 (c1 = the following 16 bases)
            if constexpr (M == 8) {
                // w_j = bases j .. j + 15: m-mer j in its high half, m-mer j + 8 in its low half -- one v_alignbit per PAIR of m-mers; the
                // reverse complements likewise, out of the reverse complement R of all 32 bases: revcomp(w_j) = R >> 2 j, halves swapped
                u64 y = __builtin_bitreverse64(((u64)c0 << 32) | c1);
                y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
                const u64 R = ~y;
                const u32 Rhi = (u32)(R >> 32), Rlo = (u32)R;
                u32 P[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const u32 w = j ? __builtin_amdgcn_alignbit(c0, c1, 32 - 2 * j) : c0;
                    const u32 r = j ? __builtin_amdgcn_alignbit(Rhi, Rlo, 2 * j) : Rlo;
                    P[j] = sk_pk_scramble(sk_pkmin(w, __builtin_amdgcn_alignbit(r, r, 16)));   // high: value j, low: value j + 8
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    out8[i] = __builtin_amdgcn_perm(P[2 * i + 1], P[2 * i], 0x07060302u);       // value 2 i | value 2 i + 1 << 16
                    out8[4 + i] = __builtin_amdgcn_perm(P[2 * i + 1], P[2 * i], 0x05040100u);   // value 8 + 2 i | value 9 + 2 i << 16
                }
            } else {
                const u64 win = ((u64)c0 << 32) | c1;
                u32 fw = (u32)(win >> (64 - 2 * M)) & MM;
                u32 r = __builtin_bitreverse32(fw) >> (32 - 2 * M);
                u32 rc = (((r >> 1) & 0x5555u) | ((r & 0x5555u) << 1)) ^ MM;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j) {
                        fw = (u32)(win >> (64 - 2 * M - 2 * j)) & MM;
                        rc = (rc >> 2) | ((3u - (fw & 3u)) << (2 * M - 2));
                    }
                    const u32 v = sk_scramble(fw < rc ? fw : rc);
                    if (j & 1) out8[j >> 1] |= v << 16; else out8[j >> 1] = v;
                }
            }
        };
        {
            u32 o8[8];
            scramble16(tc[1 + t], tc[2 + t], o8);
            reinterpret_cast<uint4 *>(mm)[2 * t] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
            reinterpret_cast<uint4 *>(mm)[2 * t + 1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
            if ((int)t < HT) {  // the halo: m-mers that start beyond the tile's last window start
                scramble16(tc[1 + kSkThreads + t], tc[2 + kSkThreads + t], o8);
                reinterpret_cast<uint4 *>(mm)[2 * (kSkThreads + t)] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
                reinterpret_cast<uint4 *>(mm)[2 * (kSkThreads + t) + 1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
            }
        }
        __syncthreads();
        // ---- minimiser and owner of each of this thread's 16 windows, one byte each; 0xFF = not a good window ----------------------------
        u32 ow[4];
        {
            u32 A[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) A[j] = mm[8 * t + j];
            sk_sliding_min<W, NP>(A);
            // good windows: K valid bases in a row -- an erosion of the validity bits (base b in bit 63 - b) by doubling shifts
            constexpr int NWV = (15 + K + 15) / 16;
            u32 good;  // window j in bit j
            if constexpr (NWV <= 4) {
                u64 e = 0;
#pragma unroll
                for (int i = 0; i < NWV; ++i) e |= (u64)tv[t + i] << (48 - 16 * i);
                int sft = 1;
#pragma unroll
                for (int lvl = 0; lvl < 6; ++lvl) if (2 * sft <= K) { e &= e << sft; sft *= 2; }
                if (K - sft > 0) e &= e << (K - sft);
                good = __builtin_bitreverse32((u32)(e >> 32)) & 0xFFFFu;
            } else {
                unsigned __int128 e = 0;
#pragma unroll
                for (int i = 0; i < NWV; ++i) e |= (unsigned __int128)tv[t + i] << (112 - 16 * i);
                int sft = 1;
#pragma unroll
                for (int lvl = 0; lvl < 6; ++lvl) if (2 * sft <= K) { e &= e << sft; sft *= 2; }
                if (K - sft > 0) e &= e << (K - sft);
                good = __builtin_bitreverse32((u32)(e >> 96)) & 0xFFFFu;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 o = __builtin_amdgcn_perm(sk_pk_owner(A[2 * w + 1], world), sk_pk_owner(A[2 * w], world), 0x06040200u);  // four owner bytes
                const u32 ok = ((((good >> (4 * w)) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;                                    // 0xFF per good window
                ow[w] = o | ~ok;
            }
            edge[t] = (unsigned short)((ow[0] & 0xFFu) | ((ow[3] >> 24) << 8));
        }
        __syncthreads();
        // ---- runs: maximal stretches of good windows with one owner, cut every `cut` windows (cut >= 16, a power of two) ---------------
        const u32 prev_o = t ? (u32)edge[t - 1] >> 8 : 0xFFu, next_o = t + 1 < kSkThreads ? (u32)edge[t + 1] & 0xFFu : 0xFFu;
        u32 smask = 0;
        auto build_masks = [&](u32 cut) {
            u32 em = 0;
            smask = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 o = ow[w];
                const u32 before = __builtin_amdgcn_alignbit(o, w ? ow[w - 1] : prev_o << 24, 24);   // every byte's left neighbour
                const u32 after = __builtin_amdgcn_alignbit(w < 3 ? ow[w + 1] : next_o, o, 8);        // ... right neighbour
                const u32 valid = ~o & 0x80808080u;                                                   // (owners are < 64; 0xFF = no window)
                smask |= sk_gather4(sk_nonzero_bytes(o ^ before) & valid) << (4 * w);
                em |= sk_gather4(sk_nonzero_bytes(o ^ after) & valid) << (4 * w);
            }
            if (((16 * t) & (cut - 1)) == 0 && !(ow[0] & 0x80u)) smask |= 1u;
            if (((16 * t + 16) & (cut - 1)) == 0 && !(ow[3] >> 31)) em |= 0x8000u;
            emask[t] = (unsigned short)em;
        };
        // visit(j, n, owner) for every run that starts at this thread's window j, first <= 16 t + j < last
        auto for_runs = [&](u32 first, u32 last, auto &&visit) {
            if (16 * t + 16 <= first || 16 * t >= last) return;
            u32 sm = smask;
            const u32 mye = emask[t];
            while (sm) {
                const u32 j = (u32)__builtin_ctz(sm);
                sm &= sm - 1;
                u32 n;
                const u32 e = mye >> j;
                if (e) n = (u32)__builtin_ctz(e) + 1;
                else {
                    n = 16 - j;
                    u32 tt = t + 1;
                    for (;;) {
                        const u32 e2 = emask[tt];   // (emask[1024] = all ones: the tile's end)
                        if (e2) { n += (u32)__builtin_ctz(e2) + 1; break; }
                        n += 16; ++tt;
                    }
                }
                visit(j, n, (ow[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            }
        };
        // a run takes its place in its owner's staging with ONE ds_add_rtn_u64 and is copied there; a run that would not fit raises s_over
        auto emit = [&](u32 pos, u32 n, u32 o) {   // pos: the run's first window in the tile
            const u32 L = n + K - 1;
            const u64 cw = atomicAdd(&cur[o], (u64)n | ((u64)L << 32));
            const u32 wpos = (u32)cw, bpos = (u32)(cw >> 32);
            if (bpos + L > capB) { s_over = 1u; return; }
            atomicOr(&ss[o * (capBw / 2) + (wpos >> 5)], 1u << (wpos & 31u));
            // L bases from tile bit 2 pos to staging bit 2 bpos (both MSB-first), one destination word at a time.  Every step moves
            // 32 bits on in BOTH streams, so the source's bit offset inside its words is the same for every word of the run: one LDS
            // read and one v_alignbit per word (the previous read is the next word's high half); only the first and the last word
            // are masked.  (The first version recomputed the offset and read two words per step: the copy was 36 % of the kernel.)
            u32 *dst = sb + o * capBw;
            const u32 dbit = 2 * bpos, nbits = 2 * L;
            const u32 dw0 = dbit >> 5, dw1 = (dbit + nbits - 1) >> 5;
            const u32 sp = 32u + 2u * pos - (dbit & 31u);   // source bit of the first destination word's bit 0 (>= 1: tc[0] is a zero word)
            const u32 sh = sp & 31u;
            u32 wi = (sp >> 5) - (sh == 0u ? 1u : 0u);                // v_alignbit(hi, lo, 0) = lo: an aligned source is read one word late
            const u32 shift = (32u - sh) & 31u;
            const u32 m0 = 0xFFFFFFFFu >> (dbit & 31u);               // the run's bits of its first word ...
            const u32 m1 = 0xFFFFFFFFu << (31u - ((dbit + nbits - 1u) & 31u));   // ... and of its last
            u32 hi = tc[wi];
            for (u32 dw = dw0; dw <= dw1; ++dw) {
                const u32 lo = tc[++wi];
                u32 val = __builtin_amdgcn_alignbit(hi, lo, shift);
                hi = lo;
                if (dw == dw0) val &= m0;
                if (dw == dw1) val &= m1;
                atomicOr(&dst[dw], val);
            }
        };
        // whole 16-byte units leave for this workgroup's regions; the partial ones move to the front for the next tile
        auto flush = [&]() {
            for (u32 u = t; u < world * capU; u += kSkThreads) {
                const u32 o = __umulhi(u, invU), ul = u - o * capU;
                if (ul < (u32)(cur[o] >> 32) / kSkUnitBases) {
                    uint4 *src = reinterpret_cast<uint4 *>(sb) + u;
                    const uint4 v = *src;
                    *src = make_uint4(0, 0, 0, 0);
                    const u32 at = filledB[o] + ul;
                    if (at < a.cap_units) my_bases[(u64)o * a.cap_units + at] = v;
                }
            }
            for (u32 u = t; u < world * capSU; u += kSkThreads) {
                const u32 o = __umulhi(u, invSU), ul = u - o * capSU;
                if (ul < (u32)cur[o] / kSkUnitWindows) {
                    uint4 *src = reinterpret_cast<uint4 *>(ss) + u;
                    const uint4 v = *src;
                    *src = make_uint4(0, 0, 0, 0);
                    const u32 at = filledS[o] + ul;
                    if (at < a.cap_sunits) my_starts[(u64)o * a.cap_sunits + at] = v;
                }
            }
            __syncthreads();
            if (t < world) {
                const u64 cw = cur[t];
                const u32 tb = (u32)(cw >> 32), tw = (u32)cw, ub = tb / kSkUnitBases, us = tw / kSkUnitWindows;
                uint4 *bb = reinterpret_cast<uint4 *>(sb + t * capBw), *sbits = reinterpret_cast<uint4 *>(ss + t * (capBw / 2));
                if (ub) { const uint4 v = bb[ub]; bb[ub] = make_uint4(0, 0, 0, 0); bb[0] = v; }
                if (us) { const uint4 v = sbits[us]; sbits[us] = make_uint4(0, 0, 0, 0); sbits[0] = v; }
                carryB[t] = bb[0]; carryS[t] = sbits[0];
                filledB[t] += ub; filledS[t] += us;
                cur[t] = cur0[t] = (u64)(tw % kSkUnitWindows) | ((u64)(tb % kSkUnitBases) << 32);
            }
            __syncthreads();
        };
        build_masks(kSkMaxRun);
        __syncthreads();
        my_runs += (u32)__builtin_popcount(smask);
#ifdef KCT_DEBUG_ENV   // (timing experiments, tools/pmc_ablate.sh: 0x100 no emit, 0x200 cursors only, 0x400 no flush)
        if (a.ablate & 0x200u) for_runs(0, kSkTile, [&](u32, u32 n, u32 o) { atomicAdd(&cur[o], (u64)n | ((u64)(n + K - 1) << 32)); });
        else if (!(a.ablate & 0x100u))
#endif
        {
            // A thread finds 1.9 run starts among its 16 windows on average and 5-6 at worst in a wave: emitted where they are found, the
            // wave's lanes idle two thirds of the copy.  So the runs first go on a LIST (in `mm`, which is dead by now) -- a wave reserves
            // its stretch with ONE atomic: the lanes' counts are summed bit by bit with ballots, no cross-lane traffic -- and then every
            // thread emits list entries t, t + 1024, ...: two rounds with all lanes busy.
            const u32 cnt = (u32)__builtin_popcount(smask), lane = t & 63u;
            u32 before = 0, total = 0;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const u64 m = __ballot((cnt >> b) & 1u);
                before += __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)) << b;
                total += (u32)__popcll(m) << b;
            }
            u32 base = 0;
            if (lane == 0 && total) base = atomicAdd(&nlist, total);
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (base + total <= (u32)kMmWords) {
                u32 slot = base + before;
                for_runs(0, kSkTile, [&](u32 j, u32 n, u32 o) { mm[slot++] = (16u * t + j) | ((n - 1u) << 14) | (o << 24); });
            } else {   // (more runs than the list holds: this wave's go out directly, what it reserved of the list says "no run")
                for (u32 x = base + lane; x < (u32)kMmWords; x += 64u) mm[x] = 0xFFFFFFFFu;
                for_runs(0, kSkTile, [&](u32 j, u32 n, u32 o) { emit(16u * t + j, n, o); });
            }
            __syncthreads();
            const u32 nl = nlist < (u32)kMmWords ? nlist : (u32)kMmWords;
            for (u32 i = t; i < nl; i += kSkThreads) { const u32 e = mm[i]; if (e != 0xFFFFFFFFu) emit(e & 0x3FFFu, ((e >> 14) & 0x3FFu) + 1u, e >> 24); }
        }
        __syncthreads();
#ifdef KCT_DEBUG_ENV
        if (a.ablate & 0x600u) { if (t < world) cur[t] = cur0[t] = 0; __syncthreads(); continue; }
#endif
        if (s_over == 0) flush();
        else {
            // (rare: far more runs than random sequence gives, or nearly all for one owner) -- the staging goes back to what the tile
            // found, and the tile goes out in pieces that cannot overflow it
            __syncthreads();
            for (u32 i = t; i < (u32)kSkStageWords; i += kSkThreads) sb[i] = 0;
            for (u32 i = t; i < (u32)kSkStageWords / 2; i += kSkThreads) ss[i] = 0;
            if (t == 0) s_over = 0;
            __syncthreads();
            if (t < world) {
                *reinterpret_cast<uint4 *>(sb + t * capBw) = carryB[t];
                *reinterpret_cast<uint4 *>(ss + t * (capBw / 2)) = carryS[t];
                cur[t] = cur0[t];
            }
            my_runs -= (u32)__builtin_popcount(smask);
            build_masks(gsafe < kSkMaxRun ? gsafe : kSkMaxRun);
            __syncthreads();
            my_runs += (u32)__builtin_popcount(smask);
            for (u32 first = 0; first < (u32)kSkTile; first += gsafe) {
                for_runs(first, first + gsafe, [&](u32 j, u32 n, u32 o) { emit(16u * t + j, n, o); });
                __syncthreads();
                flush();
            }
        }
    }
    // ---- the partial units, zero-padded; what every stream holds -----------------------------------------------------------------------
    atomicAdd(&s_runs, my_runs);
    __syncthreads();
    if (t < world) {
        const u64 cw = cur[t];
        const u32 rb = (u32)(cw >> 32), rw = (u32)cw;
        u32 fb = filledB[t], fs = filledS[t];
        const u32 nwin = fs * kSkUnitWindows + rw;
        if (rb) {
            if (fb < a.cap_units) my_bases[(u64)t * a.cap_units + fb] = *reinterpret_cast<uint4 *>(sb + t * capBw);
            ++fb;
        }
        if (rw) {
            if (fs < a.cap_sunits) my_starts[(u64)t * a.cap_sunits + fs] = *reinterpret_cast<uint4 *>(ss + t * (capBw / 2));
            ++fs;
        }
        const u64 idx = (u64)blockIdx.x * world + t;
        a.nwin[idx] = nwin; a.nunits[idx] = fb; a.nsunits[idx] = fs; a.nruns[idx] = t == 0 ? s_runs : 0u;
        if (fb > a.cap_units || fs > a.cap_sunits) *a.overflow = 1ULL;
    }
}

// n[i] 16-byte units from src + src_off[i] to dst + dst_off[i] (offsets in units): region packing / slab assembly
__global__ __launch_bounds__(kBlock) void gather_units_kernel(const uint4 *__restrict__ src, const u64 *__restrict__ src_off, const u64 *__restrict__ dst_off,
                                                              const u32 *__restrict__ n, u32 count, uint4 *__restrict__ dst) {
    for (u32 i = blockIdx.x; i < count; i += gridDim.x) {
        const uint4 *s = src + src_off[i];
        uint4 *d = dst + dst_off[i];
        const u32 m = n[i];
        for (u32 u = threadIdx.x; u < m; u += kBlock) d[u] = s[u];
    }
}

// One 256-thread workgroup per stream (strided): running popcount of the start bits -> one RunGroup per 64 windows.
__global__ __launch_bounds__(kBlock) void run_directory_kernel(const RunStream *__restrict__ streams, u32 nstreams, const u64 *__restrict__ starts, int k,
                                                               RunGroup *__restrict__ groups) {
    __shared__ u32 wsum[kBlock / 64];
    __shared__ u64 carry;
    for (u32 s = blockIdx.x; s < nstreams; s += gridDim.x) {
        const RunStream st = streams[s];
        const u32 ng = (st.nwin + 63) >> 6;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (u32 g0 = 0; g0 < ng; g0 += kBlock) {
            const u32 g = g0 + threadIdx.x;
            const u64 word = g < ng ? starts[st.word0 + g] : 0ULL;
            const u32 pc = (u32)__popcll(word);
            // exclusive prefix inside the workgroup: wave scan, then the waves' totals
            u32 incl = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = __shfl_up(incl, off);
                if ((int)(threadIdx.x & 63) >= off) incl += v;
            }
            if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
            __syncthreads();
            u32 before = 0;
            for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
            const u64 runs_before = carry + before + incl - pc;   // run starts in front of this group
            if (g < ng) {
                RunGroup rg;
                const u32 left = st.nwin - 64u * g;
                rg.base_nvalid = (st.bit0 + 2ULL * (64ULL * g + (u64)(k - 1) * runs_before) - 2ULL * (u64)(k - 1)) | ((u64)(left < 64u ? left : 64u) << 56);
                rg.starts = word;
                groups[st.group0 + g] = rg;
            }
            __syncthreads();
            if (threadIdx.x == kBlock - 1) carry += before + incl;
            __syncthreads();
        }
    }
}

// Windows [64 * group0, 64 * (group0 + ngroups)) as an ASCII record stream: window v = k bytes + '\n' at out + (v - 64 * group0) * (k + 1)
// (an absent window: k + 1 separators).  For the kernels that read bytes.
__global__ __launch_bounds__(kBlock) void expand_runs_kernel(RunsInput in, u64 ngroups, int k, unsigned char *__restrict__ out) {
    const u32 lane = threadIdx.x & 63u;
    for (u64 g = (u64)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); g < ngroups; g += (u64)gridDim.x * (kBlock / 64)) {
        const RunGroup d = in.groups[g];
        unsigned char *o = out + (g * 64 + lane) * (u64)(k + 1);
        if (lane < (u32)(d.base_nvalid >> 56)) {
            const u64 below = d.starts & ((2ULL << lane) - 1ULL);
            const u64 bit = (d.base_nvalid & kRunBaseMask) + 2ULL * ((u64)lane + (u64)(k - 1) * (u64)__popcll(below));
            for (int i = 0; i < k; ++i) {
                const u64 b = bit + 2ULL * i;
                o[i] = (unsigned char)((0x54474341u >> (8 * ((in.bases[b >> 5] >> (30 - (b & 31))) & 3u))) & 0xFFu);
            }
            o[k] = '\n';
        } else {
            for (int i = 0; i <= k; ++i) o[i] = '\n';
        }
    }
}

}  // namespace kct
