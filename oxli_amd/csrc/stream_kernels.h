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
