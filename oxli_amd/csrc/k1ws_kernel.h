// k1ws_kernel.h -- K1 with WAVE SPECIALISATION (round 6): the first kernel of the partitioned path with its hashing waves never at a
// workgroup barrier.
//
// Why.  partition_windows_kernel (k1_kernel.h) keeps its sixteen waves in step: two barriers to stage a tile, and every 4 / 8 windows a
// ring_flush of three barriers around two short, latency-bound LDS phases, during which no wave hashes.  Per-wave shader-clock stamps
// (profiles/r06_k1_wait_split.json) put ~half of a wave's life into those barriers and phases; the VALU pipes are ~55 % busy.  Here
//   * waves F .. 15 (H = 16 - F of them) only HASH: a wave fetches 1 KiB of the stream itself (16 bytes per lane, the next piece
//     prefetched in registers), gets its neighbours' code words across lanes (ds_bpermute: no LDS tile, no barrier), walks its 16
//     windows per lane and appends to the same LDS write-combining ring with the same ONE ds_add_rtn_u64 per window;
//   * waves 0 .. F-1 only FLUSH: they sweep the bins' cursors and move complete 64-byte lines to this workgroup's regions in HBM.
// What replaces the barrier ("every append of the interval has landed") is a grace period.  A hashing wave commits window j's append
// (the ring write) at the start of window j + 1, then publishes its window count (an LDS word per wave: "epoch"), then takes window
// j + 1's position.  A flusher SNAPSHOTS a bin's fill count, afterwards reads every wave's epoch E, and may move the lines below the
// snapshot once every wave's epoch exceeds E (or the wave was between pieces -- nothing pending -- when E was read, or is done): every
// position handed out before the snapshot has then been written, or was refused (ring full: overflow route, the position stays a zero
// hole exactly as in k1_kernel.h).  LDS operations of one wave are executed in program order, and the LDS serves one instruction at a
// time, so "ring write, then epoch write" on one side and "epoch read, then ring read" on the other need no fence -- only the
// compiler must not reorder them (volatile accesses + asm memory clobbers).
// The output (regions, region counts, overflow regions) has k1_kernel.h's format, so K1b / K2 and the host code are unchanged; which
// workgroup region an entry lands in, and the order inside a region, differ -- neither matters to anything downstream.
#pragma once
#include <type_traits>

#include "k1_kernel.h"

namespace kct {

constexpr u32 kEpochQuiet = 1u << 30;   // the wave is between pieces: nothing handed out and not yet written
constexpr u32 kEpochDone = 0xFFFFFFFFu;

// MODE as partition_windows_kernel (0 MurmurHash3 values, 1 mix64, 2 mix42 / u32 entries, 3 mix128 / 16-byte entries); KW != 0 (k <= 64);
// F = flusher waves.  Input: the ASCII record stream, or packed base arrays when a.pcodes is set (a group per lane: 4 + 2 bytes).
template <int KW, int KC, int MODE, int F>
__global__ __launch_bounds__(kPartThreads) void partition_windows_ws_kernel(const unsigned char *__restrict__ stream, u64 nbytes, int k, u64 ntiles, PartitionArgs a) {
    static_assert(KW == 1 || KW == 2, "k <= 64");
    static_assert(F >= 1 && F <= 8, "flusher waves");
    using T = typename std::conditional<MODE == 2, u32, typename std::conditional<MODE == 3, ulonglong2, u64>::type>::type;
    using PH = typename std::conditional<MODE == 3, u64, T>::type;
    constexpr int kEntries = kRingEntries * 8 / sizeof(T);  // 128 KiB of ring
    constexpr u32 CH = 64 / sizeof(T);                       // positions per 64-byte line
    constexpr int kWaves = kPartThreads / 64, H = kWaves - F;
    constexpr u32 kListCap = 2048 / F;                       // lines a flusher wave lists per sweep
    constexpr int NW = 2 * KW + 1;
    __shared__ __attribute__((aligned(16))) T ring[kEntries];
    __shared__ u64 cur[1024];        // per bin: fill (low half) | flushed (high half)
    __shared__ u32 snap[1024];       // per bin: the fill count its flusher saw last
    __shared__ u32 flist[F][kListCap];
    __shared__ u32 epoch[kWaves];
    __shared__ u32 next_piece, ovf_n;
    __shared__ u32 ascii4[MODE != 0 ? 1 : 256];
    if constexpr (MODE == 0) fill_ascii4_lut(ascii4, threadIdx.x, kPartThreads);
    constexpr bool kPremul = MODE == 0 && (KC == 0 || KC >= 16);
    __shared__ u64 mul1[kPremul ? 256 : 1], mul2[kPremul ? 256 : 1];
    if constexpr (kPremul) fill_premul_luts(mul1, mul2, threadIdx.x, kPartThreads);
    const u64 *pm1 = kPremul ? mul1 : nullptr, *pm2 = kPremul ? mul2 : nullptr;
    constexpr bool kTailLut = kPremul && KC > 0 && tail_needs_lut(KC);
    constexpr int kPre = (kPremul ? 1 : 0) | ((kPremul && KC > 0 && (KC & 15) != 0) ? 2 : 0);
    __shared__ u64 tmul[kTailLut ? 256 : 1];
    if constexpr (kTailLut) fill_tail_lut(tmul, threadIdx.x, kPartThreads, KC);
    const u64 *ptm = kTailLut ? tmul : nullptr;

    const int P = 1 << a.pbits;
    const u32 D = (u32)(kEntries >> a.pbits), dmask = D - 1;
    const int dshift = __builtin_ctz((unsigned)kEntries) - a.pbits;  // log2 D
    {
        T zero;
        memset(&zero, 0, sizeof zero);
        for (int i = threadIdx.x; i < kEntries; i += kPartThreads) ring[i] = zero;
    }
    for (int i = threadIdx.x; i < 1024; i += kPartThreads) { cur[i] = 0; snap[i] = 0; }
    if (threadIdx.x < kWaves) epoch[threadIdx.x] = threadIdx.x < F ? kEpochDone : kEpochQuiet;
    if (threadIdx.x == 0) { ovf_n = 0; next_piece = 0; }
    T *my_scratch = reinterpret_cast<T *>(a.scratch) + (u64)blockIdx.x * P * a.region_cap;
    u64 *my_ovf = a.ovf + (u64)blockIdx.x * a.ovf_cap * (MODE == 3 ? 2 : 1);
    auto overflow_hash = [&](u64 h, u64 y = 0) {
        const u32 i = atomicAdd(&ovf_n, 1u);
        if (i < a.ovf_cap) {
            if constexpr (MODE == 3) { my_ovf[2 * i] = h; my_ovf[2 * i + 1] = y; }
            else my_ovf[i] = h;
        } else *a.overflow = 1ULL;
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();   // ring, cursors, epochs and look-up tables are in place: the only barrier before the kernel's end

    if (wave >= F) {
        // ================================================ hashing waves ================================================================
        // This workgroup counts a CONTIGUOUS stretch of the launch's window starts (the same number of tiles as a workgroup of
        // partition_windows_kernel takes, so the host's region sizes hold), cut into PIECES of PC = 64 - (NW - 1) sixteen-base chunks: lane l
        // of a wave fetches and encodes chunk first + l, lanes 0 .. PC - 1 walk the sixteen windows that start in their chunk, and the
        // last NW - 1 lanes only lend their code words to the lanes in front (one encode16 per lane and piece; a piece of 64 chunks with a
        // separately fetched halo cost every lane a second encode16: + 8 VALU per window).  Pieces are handed out first come, first
        // served (an LDS counter): a wave that shares its SIMD with a flusher, or falls behind for any reason, simply takes fewer.
        constexpr u32 PC = 64 - (NW - 1);
        const u64 tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
        const u64 chunk0 = (u64)blockIdx.x * tiles_per_wg * (kPartTile / 16);
        const u64 chunk1 = chunk0 + tiles_per_wg * (kPartTile / 16) < ntiles * (kPartTile / 16) ? chunk0 + tiles_per_wg * (kPartTile / 16) : ntiles * (kPartTile / 16);
        // (relaxed workgroup-scope atomics, not `volatile`: a volatile access to LDS compiles to a FLAT instruction with system-scope
        // cache bits -- 2.4 M of them per launch made the first version of this kernel 11 % slower than the barrier-synchronised K1)
        auto publish = [&](u32 v) { if (lane == 0) __hip_atomic_store(&epoch[wave], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        u32 ep = 0;
#ifdef KCT_K1_STAMPS
        const u64 st_t0 = __builtin_amdgcn_s_memtime();
        u64 st_pieces = 0;
#endif
        const bool packed = a.pcodes != nullptr;   // (workgroup-uniform)
        auto grab = [&]() -> u64 {   // the next piece's first chunk, ~0 when this workgroup's stretch is used up
            u32 c = 0;
            if (lane == 0) c = atomicAdd(&next_piece, 1u);
            c = (u32)__builtin_amdgcn_readfirstlane((int)c);
            const u64 first = chunk0 + (u64)c * PC;
            return first < chunk1 ? first : ~0ULL;
        };
        auto load_one = [&](u64 off) -> uint4 {   // ASCII: the 16 bytes at `off`; packed: the group's code word in .x, validity in .y
            uint4 v = make_uint4(0, 0, 0, 0);
            if (packed) {
                if (off < nbytes) {
                    const u64 g = off >> 4;
                    v.x = a.pcodes[g];
                    u32 vb = a.pvalid[g];
                    if (off + 16 > nbytes) vb &= ~((1u << (16 - (u32)(nbytes - off))) - 1u);  // bases at or beyond nbytes do not exist
                    v.y = vb;
                }
                return v;
            }
            if (off + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(stream + off);
            else if (off < nbytes) {
                unsigned char tmp[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) tmp[i] = (off + i < nbytes) ? stream[off + i] : (unsigned char)0;
                v = *reinterpret_cast<uint4 *>(tmp);
            }
            return v;
        };
        u64 first = grab();
        uint4 pre_main = make_uint4(0, 0, 0, 0);
        if (first != ~0ULL) pre_main = load_one((first + (u64)lane) << 4);
        const u32 Pm1 = (u32)(P - 1);
        while (first != ~0ULL) {
            u32 c0, v0;
            if (packed) { c0 = pre_main.x; v0 = pre_main.y; }
            else encode16(pre_main, c0, v0);
            const bool walker = (u32)lane < PC && first + (u64)lane < chunk1;   // (the other lanes' windows belong to the next piece, or to the next workgroup)
            // the next piece's bytes are requested before this one is hashed
            const u64 next = grab();
            if (next != ~0ULL) pre_main = load_one((next + (u64)lane) << 4);
            // code words / validity of lanes l + 1 .. l + NW - 1
            u32 w[NW];
            u64 vbits = (u64)v0 << 48;
            u32 vtail = 0;
            w[0] = c0;
#pragma unroll
            for (int i = 1; i < NW; ++i) {
                const int src = ((lane + i) & 63) << 2;
                w[i] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)c0);
                const u32 vv = (u32)__builtin_amdgcn_ds_bpermute(src, (int)v0);
                if (i < 4) vbits |= (u64)vv << (48 - 16 * i);
                else vtail = vv;
            }
            if (!walker) { vbits = 0; vtail = 0; }   // no window of this lane is good
            PH pend_h = 0;
            u64 pend_y = 0, aux_y = 0;
            u32 pend_b = 0, pend_pos = 0, pend_mark = 0;
            auto commit = [&]() {
                if (pend_h) {
                    if (pend_pos - pend_mark < D) {
                        if constexpr (MODE == 3) ring[(pend_b << dshift) + (pend_pos & dmask)] = make_ulonglong2(pend_h, pend_y);
                        else ring[(pend_b << dshift) + (pend_pos & dmask)] = pend_h;
                    } else if constexpr (MODE == 3) overflow_hash(pend_h, pend_y);
                    else overflow_hash(MODE == 2 ? (((u64)pend_b << 32) | pend_h | (1ULL << 63)) : (u64)pend_h);
                    pend_h = 0;
                }
            };
            auto sink = [&](int j, bool good, u64 h) {
                commit();                                  // window j - 1's append is in the ring (or on the overflow route) ...
                asm volatile("" ::: "memory");
                ++ep;
                publish(ep);                               // ... before the count that says so
                asm volatile("" ::: "memory");
                if (MODE == 3 && good && h == 0) overflow_hash(0ULL, aux_y);
                if (good && h != 0) {
                    if (MODE == 2 && (u32)h == 0) overflow_hash(h);
                    else {
                        if constexpr (MODE == 2) pend_b = (u32)(h >> 32) & 1023u;
                        else pend_b = (u32)(h >> a.block_bits) & Pm1;
                        const u64 cw = atomicAdd(&cur[pend_b], 1ULL);
                        pend_pos = (u32)cw;
                        pend_mark = (u32)(cw >> 32);
                        pend_h = (PH)h;
                        if constexpr (MODE == 3) pend_y = aux_y;
                    }
                }
            };
            walk_windows_words<KW, KC, true, MODE, kPre>(w, vbits, vtail, k, sink, ascii4, pm1, pm2, &aux_y, ptm);
            commit();
            asm volatile("" ::: "memory");
            ++ep;
            publish(ep | kEpochQuiet);                     // between pieces: nothing of this wave is pending
            asm volatile("" ::: "memory");
            first = next;
#ifdef KCT_K1_STAMPS
            ++st_pieces;
#endif
        }
        asm volatile("" ::: "memory");
        publish(kEpochDone);
#ifdef KCT_K1_STAMPS
        if (a.stamps && lane == 0) {
            u64 *o = a.stamps + ((u64)blockIdx.x * kWaves + wave) * kStampSlots;
            o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = st_pieces;
        }
#endif
    } else {
        // ================================================ flusher waves ================================================================
        // A sweep visits every bin of the wave (lane l of flusher f has bins f + F (l + 64 i): few bins are spread over all flushers) and lists lines for the wave's lanes to
        // move, four lanes a line:
        //   TENTATIVE  the line at `flushed` when `fill` says it has been handed out completely: moved if every entry of it is non-zero
        //              (an entry is never zero: such values take the overflow route) -- the common case, no waiting;
        //   FORCED     every line below the bin's last snapshot once the grace period for that snapshot is over (header comment): what is
        //              still zero in them is a hole (a refused append), and lines at or beyond flushed + D were never in the ring at all.
        // measurement switches (PartitionArgs::ablate, a -DKCT_DEBUG_ENV build's KCT_ABLATE; 0 in the shipped build = the defaults):
        // bits 8-15: units of 64 cycles a flusher sleeps between sweeps (0 = kSweepPause); bit 16: flushers at the hashing waves' priority
        constexpr u32 kSweepPause = 0;
        const u32 pause = ((u32)a.ablate >> 8) & 0xFFu ? (((u32)a.ablate >> 8) & 0xFFu) - 1u : kSweepPause;
        if (!((u32)a.ablate & (1u << 16))) __builtin_amdgcn_s_setprio(2);
        constexpr int NB = 1024 / (64 * F);
        constexpr u32 kTentative = 1u << 30, kHole = 1u << 31, kPosMask = (1u << 20) - 1u;   // (the launcher refuses regions of 2^20 entries or more)
        auto epoch_of = [&](int w) -> u32 { return __hip_atomic_load(&epoch[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        u32 *mylist = flist[wave];
        u32 e_post = kEpochQuiet;             // this lane's hashing wave (F + lane, lane < H) as read after the last snapshots
#ifdef KCT_K1_STAMPS
        const u64 st_t0 = __builtin_amdgcn_s_memtime();
        u64 st_sweeps = 0, st_listed = 0, st_graces = 0, st_list_cyc = 0, st_move_cyc = 0, st_tent = 0;
#endif
        for (;;) {
#ifdef KCT_K1_STAMPS
            const u64 st_a = __builtin_amdgcn_s_memtime();
#endif
            const u32 e_now = lane < H ? epoch_of(F + lane) : kEpochDone;
            const bool ok = e_now == kEpochDone || (e_post & kEpochQuiet) != 0 || (e_now & ~kEpochQuiet) > (e_post & ~kEpochQuiet);
            const bool grace = __ballot(ok) == ~0ULL;                     // the snapshots may be used (and are then taken anew)
            const bool drain = __ballot(e_now == kEpochDone) == ~0ULL;    // every append there will ever be has landed
            bool more = false;                                            // (drain) something of this lane is still in the ring
            asm volatile("" ::: "memory");
            u64 cw[NB];
            u32 sn[NB], delta[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int b = wave + F * (lane + 64 * i);
                cw[i] = b < P ? cur[b] : 0ULL;
                sn[i] = b < P ? snap[b] : 0u;
            }
            // what this lane will list: nl[i] forced lines of bin i (or one tentative line: tent bit i); the wave's lanes take their
            // stretches of the list by a prefix sum over ballots of the counts' bits (no LDS atomic: the compiler expands one on a
            // wave-uniform address into a forty-instruction scan, and there was one per bin)
            u32 nl[NB], want = 0, tent = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                delta[i] = 0;
                const u32 fill = (u32)cw[i], f0 = (u32)(cw[i] >> 32);
                const u32 upto = drain ? fill : sn[i];
                u32 n = (grace || drain) && (int)(upto - f0) > 0 ? (upto - f0) / CH : 0u;
                if (drain && (int)(upto - f0) > 0 && ((upto - f0) & (CH - 1))) ++n;   // the partial line, zero-padded
                if (n > 7u) n = 7u;                                                     // (the rest: next sweep)
                if (!n && (int)(fill - f0) >= (int)CH) { tent |= 1u << i; want += 1; }
                if (drain && (int)(fill - (f0 + n * CH)) > 0) more = true;
                nl[i] = n; want += n;
            }
            u32 before = 0, total = 0;
#pragma unroll
            for (int bit = 0; bit < 6; ++bit) {   // want <= 7 NB <= 56
                const u64 m = __ballot((want >> bit) & 1u);
                before += __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)) << bit;
                total += (u32)__popcll(m) << bit;
            }
            const bool fits = before + want <= kListCap;   // a lane whose stretch does not fit lists nothing this sweep
            if (!fits && want) more = true;
            u32 slot = before;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int b = wave + F * (lane + 64 * i);
                const u32 fill = (u32)cw[i], f0 = (u32)(cw[i] >> 32);
                if (fits) {
                    for (u32 l = 0; l < nl[i]; ++l) {
                        const u32 f = f0 + l * CH;
                        mylist[slot++] = (u32)b | (f << 10) | (f - f0 >= D ? kHole : 0u);   // at or beyond flushed + D: never in the ring (all holes)
                    }
                    delta[i] = nl[i] * CH;
                    if ((tent >> i) & 1u) mylist[slot++] = (u32)b | (f0 << 10) | kTentative;
                }
                if (grace && b < P) snap[b] = fill;
            }
            // (lanes that fit are a prefix of the wave: the list ends where the first lane that does not fit would have begun)
            const u64 nofit = __ballot(!fits && want != 0);
            const u32 nlist = nofit ? (u32)__builtin_amdgcn_readlane((int)before, (int)__builtin_ctzll(nofit)) : total;
            asm volatile("" ::: "memory");
#ifdef KCT_K1_STAMPS
            const u64 st_b = __builtin_amdgcn_s_memtime();
            ++st_sweeps; st_listed += nlist; st_graces += grace ? 1 : 0; st_list_cyc += st_b - st_a;
#endif
            // the moves, kDeep lines per lane group in flight: list entries, then ring lines, are fetched for kDeep items before the first
            // is examined (one item at a time left the wave waiting out two LDS round trips per sixteen lines)
            constexpr int kDeep = 4;
            for (u32 base = 0; base < 4 * nlist; base += 64 * kDeep) {
                u32 e[kDeep];
                uint4 v[kDeep];
#pragma unroll
                for (int d = 0; d < kDeep; ++d) { const u32 item = base + 64 * d + lane; e[d] = item < 4 * nlist ? mylist[item >> 2] : kHole; }
#pragma unroll
                for (int d = 0; d < kDeep; ++d) {
                    const u32 q = (u32)lane & 3u, b = e[d] & 1023u, f = (e[d] >> 10) & kPosMask;
                    v[d] = make_uint4(0, 0, 0, 0);
                    if (!(e[d] & kHole)) v[d] = reinterpret_cast<const uint4 *>(&ring[(b << dshift) + (f & dmask)])[q];
                }
#pragma unroll
                for (int d = 0; d < kDeep; ++d) {
                    const u32 item = base + 64 * d + lane, q = (u32)lane & 3u, b = e[d] & 1023u, f = (e[d] >> 10) & kPosMask;
                    const uint4 vv = v[d];
                    bool full;
                    if constexpr (sizeof(T) == 4) full = vv.x && vv.y && vv.z && vv.w;
                    else if constexpr (sizeof(T) == 8) full = (vv.x | vv.y) && (vv.z | vv.w);
                    else full = (vv.x | vv.y) != 0;
                    const u64 fm = __ballot(full);
                    const bool ready = item < 4 * nlist && (!(e[d] & kTentative) || ((u32)(fm >> (lane & 60)) & 0xFu) == 0xFu);   // else: an append of this line is still on its way
                    if (ready) {
                        if (!(e[d] & kHole)) reinterpret_cast<uint4 *>(&ring[(b << dshift) + (f & dmask)])[q] = make_uint4(0, 0, 0, 0);
                        if (f + CH <= a.region_cap) {
                            reinterpret_cast<uint4 *>(my_scratch + (u64)b * a.region_cap + f)[q] = vv;
                        } else if constexpr (sizeof(T) == 8) {   // region full (badly skewed input): the entries take the overflow route
                            const u64 e0 = ((u64)vv.y << 32) | vv.x, e1 = ((u64)vv.w << 32) | vv.z;
                            if (e0) overflow_hash(e0);
                            if (e1) overflow_hash(e1);
                        } else if constexpr (sizeof(T) == 16) {
                            const u64 hh = ((u64)vv.y << 32) | vv.x, yy = ((u64)vv.w << 32) | vv.z;
                            if (hh) overflow_hash(hh, yy);
                        } else {
                            const u64 hi = ((u64)b << 32) | (1ULL << 63);
                            if (vv.x) overflow_hash(hi | vv.x);
                            if (vv.y) overflow_hash(hi | vv.y);
                            if (vv.z) overflow_hash(hi | vv.z);
                            if (vv.w) overflow_hash(hi | vv.w);
                        }
                        // (the line's slots are zero: only now may appends take them)
                        if ((e[d] & kTentative) && q == 0) atomicAdd(&cur[b], (u64)CH << 32);
                    }
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int b = wave + F * (lane + 64 * i);
                if (delta[i]) atomicAdd(&cur[b], (u64)delta[i] << 32);
            }
            asm volatile("" ::: "memory");
            if (grace) e_post = lane < H ? epoch_of(F + lane) : kEpochDone;   // read AFTER the snapshots: what their grace is measured against
#ifdef KCT_K1_STAMPS
            st_move_cyc += __builtin_amdgcn_s_memtime() - st_b;
#endif
            if (drain && __ballot(more) == 0ULL) break;
            // A bin completes a line every ~10^4 cycles and its ring holds a second line's worth of appends meanwhile: sweeping more often
            // than every ~2000 cycles buys nothing and takes issue slots from the three hashing waves of this wave's SIMD.
            if (!drain) for (u32 z = 0; z < pause; ++z) __builtin_amdgcn_s_sleep(1);
        }
#ifdef KCT_K1_STAMPS
        if (a.stamps && lane == 0) {
            u64 *o = a.stamps + ((u64)blockIdx.x * kWaves + wave) * kStampSlots;
            o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = st_sweeps; o[2] = st_listed; o[3] = st_graces; o[4] = st_list_cyc; o[5] = st_move_cyc; o[6] = st_tent;
        }
#endif
        for (int b = wave + F * lane; b < P; b += 64 * F) {
            const u32 f = (u32)(cur[b] >> 32);
            a.region_count[(u64)b * gridDim.x + blockIdx.x] = f < a.region_cap ? f : a.region_cap;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.ovf_count[blockIdx.x] = ovf_n < a.ovf_cap ? ovf_n : a.ovf_cap;
}

}  // namespace kct
