// stream_kernels.h -- kernels over the record stream itself (no table): packed base arrays <-> ASCII, and the first invalid byte of a
// multi-record stream (error mode).  Included by kct_entry.hip only (plain __global__ functions: one definition per library).
#pragma once
#include "device_common.h"

namespace kct {

// ---- packed base arrays <-> ASCII record stream -----------------------------------------------------------------
// pack: group g = bytes [16g, 16g + 16) of the stream through encode16 (bytes past nbytes are invalid).
__global__ __launch_bounds__(kBlock) void pack_stream_kernel(const unsigned char *__restrict__ stream, u64 nbytes, u32 *__restrict__ codes,
                                                             unsigned short *__restrict__ valid, u64 ngroups) {
    for (u64 g = (u64)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (u64)gridDim.x * kBlock) {
        const u64 off = g << 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(stream + off);
        else {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = (off + i < nbytes) ? stream[off + i] : (unsigned char)0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        u32 c, vb;
        encode16(v, c, vb);
        codes[g] = c; valid[g] = (unsigned short)vb;
    }
}
// unpack: 'A' 'C' 'G' 'T' for valid bases, 'N' for the others (case and the identity of an invalid byte are not kept -- neither
// matters to any count: a window is good iff its k bytes are all ACGT after upper-casing).  For the kernels that read bytes.
__global__ __launch_bounds__(kBlock) void unpack_stream_kernel(const u32 *__restrict__ codes, const unsigned short *__restrict__ valid, u64 ngroups,
                                                               unsigned char *__restrict__ out) {
    for (u64 g = (u64)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (u64)gridDim.x * kBlock) {
        const u32 c = codes[g], vb = valid[g];
        unsigned char b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (vb >> (15 - i)) & 1u ? (unsigned char)((0x54474341u >> (8 * ((c >> (30 - 2 * i)) & 3u))) & 0xFFu) : (unsigned char)'N';
        *reinterpret_cast<uint4 *>(out + (g << 4)) = *reinterpret_cast<uint4 *>(b);
    }
}

// ---- staging a device-resident stream (deferred kct_consume_device calls, kct_entry.hip) -----------------------------------------
// Copies src[0, nbytes) to dst, pads with separators up to the next 16-byte boundary plus one whole unit (so that the next stream
// staged behind it starts in a record of its own), and counts the stream's good windows: the n its consume call returns
// (lib.rs:586-600: a window is good iff its k bytes are all ACGT after upper-casing; windows never reach outside the stream).
// One 1024-thread workgroup per tile of 16 KiB; a window ending at byte p is good iff the last invalid byte at or before p lies
// more than k - 1 bytes back -- a running maximum over the tile (wave scan + one LDS step), seeded from the <= 256 bytes in front.
__global__ __launch_bounds__(kPartThreads) void stage_stream_kernel(const unsigned char *__restrict__ src, u64 nbytes, int k, unsigned char *__restrict__ dst,
                                                                    u64 padded, u64 *good) {
    __shared__ int wave_last[kPartThreads / 64];
    __shared__ int halo_last;
    __shared__ u64 s_good;
    if (threadIdx.x == 0) s_good = 0;
    const u64 ntiles = (padded + kPartTile - 1) / kPartTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 mine = 0;
    auto load16 = [&](u64 off) -> uint4 {
        uint4 v = make_uint4(0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au);
        if (off + 16 <= nbytes) v = *reinterpret_cast<const uint4 *>(src + off);
        else if (off < nbytes) {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = off + i < nbytes ? src[off + i] : (unsigned char)'\n';
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        return v;
    };
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64 base = tile * kPartTile, off = base + 16ULL * threadIdx.x;
        __syncthreads();
        if (threadIdx.x == 0) halo_last = base ? -300 : -1;   // (tile-relative; nothing in front of the stream: position -1 is "invalid")
        __syncthreads();
        if (base && threadIdx.x < 16) {  // the last invalid byte among the 256 in front of the tile
            u32 c, v;
            encode16(load16(base - 256 + 16ULL * threadIdx.x), c, v);
            const u32 inv = ~v & 0xFFFFu;   // byte j of the chunk in bit 15 - j
            if (inv) atomicMax(&halo_last, -256 + 16 * (int)threadIdx.x + 15 - (int)__builtin_ctz(inv));
        }
        const uint4 data = load16(off);
        if (off < padded) *reinterpret_cast<uint4 *>(dst + off) = data;
        u32 c, v;
        encode16(data, c, v);
        const u32 inv = ~v & 0xFFFFu;
        int last = inv ? 16 * (int)threadIdx.x + 15 - (int)__builtin_ctz(inv) : -100000;   // this chunk's last invalid byte (tile-relative)
        int run = last;   // inclusive maximum over the lanes below
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(run, d);
            if (lane >= d) run = run > o ? run : o;
        }
        if (lane == 63) wave_last[wave] = run;
        __syncthreads();
        int before = halo_last;   // the last invalid byte in front of this thread's chunk
        for (int w = 0; w < wave; ++w) before = before > wave_last[w] ? before : wave_last[w];
        const int prev = __shfl_up(run, 1);
        if (lane) before = before > prev ? before : prev;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 16 * (int)threadIdx.x + j;
            if ((inv >> (15 - j)) & 1u) before = p;
            else if (p - before >= k && base + (u64)p < nbytes) ++cnt;
        }
        mine += (u64)cnt;
    }
    mine = wave_sum(mine);
    if (lane == 0 && mine) atomicAdd(&s_good, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_good) atomicAdd(good, s_good);
}

// ---- validity-only kernel for skip_bad_kmers == False over a multi-record stream -------------------
// Finds the smallest stream position q of an invalid byte that lies INSIDE a record of length
// >= k (separators and records too short to have a window do not raise, lib.rs:593-596 only
// fires for a window that exists).  rec_off[r] = stream offset of record r, rec_off[nrec] = end;
// record r spans [rec_off[r], rec_off[r+1] - 1) and is followed by its separator byte.
__global__ __launch_bounds__(kBlock) void first_bad_byte_kernel(const unsigned char *__restrict__ stream, u64 nbytes, int k,
                                                                const u64 *__restrict__ rec_off, u64 nrec, u64 *first_bad_q) {
    const u64 base = ((u64)blockIdx.x * kBlock + threadIdx.x) * 16ULL;
    if (base >= nbytes) return;
    unsigned char b[16];
    if (base + 16 <= nbytes) *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(stream + base);
    else
        for (int i = 0; i < 16; ++i) b[i] = base + i < nbytes ? stream[base + i] : (unsigned char)'A';
    u64 best = ~0ULL;
    for (int i = 0; i < 16; ++i) {
        if (base_code(b[i]) < 4) continue;
        const u64 q = base + i;
        // record holding q: largest r with rec_off[r] <= q
        u64 lo = 0, hi = nrec;  // invariant rec_off[lo] <= q < rec_off[hi]
        while (hi - lo > 1) {
            u64 mid = (lo + hi) >> 1;
            if (rec_off[mid] <= q) lo = mid; else hi = mid;
        }
        const u64 start = rec_off[lo], end = rec_off[lo + 1] - 1;  // end = separator position
        if (q < end && end - start >= (u64)k) { best = q; break; }
    }
    if (best != ~0ULL) atomicMin(first_bad_q, best);
}

}  // namespace kct
