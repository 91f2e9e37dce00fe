// parallel_inflate.h -- ONE gzip member inflated by several threads (host code; kct_ingest.hip's single-member .fastq.gz path).
//
// Why: reads arrive as single-member .fastq.gz (README.md:89-99's loop over screed.open(path)), and one deflate stream is one thread's
// work for zlib / libdeflate: kct_consume_file spent 99 % of such a call inflating (round 5: 5.2x10^8 k-mers/s, the 64-thread CPU
// baseline's rate) with the GPU idle.  A deflate stream can be entered at any block boundary if the 32 KiB of text in front of it are
// treated as UNKNOWN (the idea of pugz / rapidgzip, restated here from the format, RFC 1951 / 1952):
//   1  the compressed bytes are cut into chunks; every chunk's thread searches its chunk for the first position where a non-final
//      dynamic-Huffman block header parses cleanly (complete pre-code, complete literal / length code with an end-of-block symbol, a
//      sane distance code): a block start, or -- rarely -- a look-alike;
//   2  every thread inflates from its start until it arrives EXACTLY at the start the next chunk found (a look-alike is simply passed:
//      its chunk is dropped and the thread runs on to the following start).  Output is 16-bit: a literal is itself, a copy that reaches
//      back before the chunk's first byte becomes a MARKER 0x8000 + position in the unknown 32 KiB window;
//   3  the windows are made known front to back (chunk 0 began at the stream's start: no unknowns; each later chunk's last 32 KiB are
//      resolved with its predecessor's) -- sequential, 32 KiB per chunk;
//   4  every thread turns its 16-bit symbols into bytes at their place in the output, markers looked up in the predecessor's window;
//   5  the CRC-32 and the length of the whole text are checked against the member's trailer.  ANY irregularity -- no start found, a
//      chain that does not close, a length or CRC that differs -- makes the call return false and the caller inflates the member the
//      ordinary way: nothing this file produces is used unverified.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   // mremap
#endif
#include <sys/mman.h>
#include <zlib.h>   // crc32, crc32_combine
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <deque>
#include <vector>

namespace pgz {

// The CRC-32 of RFC 1952 over a piece of text (<= 1 GiB a call): zlib's unless the host has set a faster one of the same meaning -- kct_ingest.hip
// hands over libdeflate's (carry-less multiplication: several GB/s where zlib's table walk does ~1) when libdeflate.so.0 is there.
using crc_fn_t = unsigned (*)(unsigned, const void *, size_t);
inline unsigned zlib_crc(unsigned c, const void *p, size_t n) { return (unsigned)crc32(c, (const Bytef *)p, (uInt)n); }
inline crc_fn_t &crc_impl() { static crc_fn_t f = zlib_crc; return f; }

struct Bits {   // LSB-first bit reader (RFC 1951 3.1.1); bytes beyond the end read as zero and set `over`
    const uint8_t *p;
    size_t n;
    size_t byte = 0;
    uint64_t buf = 0;
    unsigned cnt = 0;
    Bits(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    inline void refill() {
        if (byte + 8 <= n) {
            uint64_t w;
            memcpy(&w, p + byte, 8);
            buf |= w << cnt;
            byte += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) { buf |= (uint64_t)(byte < n ? p[byte] : 0) << cnt; ++byte; cnt += 8; }
        }
    }
    inline uint32_t peek(unsigned k) const { return (uint32_t)(buf & ((1ULL << k) - 1)); }
    inline void drop(unsigned k) { buf >>= k; cnt -= k; }
    inline uint32_t get(unsigned k) { const uint32_t v = peek(k); drop(k); return v; }
    inline uint64_t pos() const { return (uint64_t)byte * 8 - cnt; }
    inline bool over() const { return pos() > (uint64_t)n * 8; }
    void seek(uint64_t bitpos) { byte = (size_t)(bitpos >> 3); buf = 0; cnt = 0; refill(); drop((unsigned)(bitpos & 7)); }
};

// canonical Huffman decoder: a direct table for codes of up to FB bits, the canonical walk for longer ones
template <int FB>
struct Huff {
    uint16_t fast[1 << FB];   // symbol << 4 | length; 0 = longer than FB bits (or unused)
    uint16_t count[16], first[16], offs[16], sorted[320];
    int maxlen = 0;
    // 0 = over-subscribed / empty, 1 = incomplete, 2 = complete
    int build(const uint8_t *len, int nsym) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < nsym; ++i) ++count[len[i]];
        count[0] = 0;
        int left = 1, used = 0;
        maxlen = 0;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return 0;
            if (count[l]) { maxlen = l; used += count[l]; }
        }
        if (!used) return 0;
        unsigned code = 0;
        uint16_t o = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; first[l] = (uint16_t)code; offs[l] = o; o += count[l]; }
        uint16_t next[16];
        memcpy(next, offs, sizeof next);
        for (int i = 0; i < nsym; ++i) if (len[i]) sorted[next[len[i]]++] = (uint16_t)i;
        memset(fast, 0, sizeof fast);
        for (int l = 1; l <= FB && l <= 15; ++l)
            for (unsigned j = 0; j < count[l]; ++j) {
                const unsigned c = first[l] + j;   // MSB-first code of length l
                unsigned r = 0;
                for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)((sorted[offs[l] + j] << 4) | l);
                for (unsigned x = r; x < (1u << FB); x += 1u << l) fast[x] = e;
            }
        return left == 0 ? 2 : 1;
    }
    inline int decode(Bits &b) const {   // (the caller has refilled: >= 56 bits are there)
        const uint16_t e = fast[b.peek(FB)];
        if (e) { b.drop(e & 15); return e >> 4; }
        unsigned code = 0;
        for (int l = 1; l <= maxlen; ++l) {
            code = (code << 1) | ((uint32_t)(b.buf >> (l - 1)) & 1u);
            if (l > FB && code - first[l] < count[l] && code >= first[l]) { b.drop(l); return sorted[offs[l] + (code - first[l])]; }
        }
        return -1;
    }
};

using LitLen = Huff<11>;
using Dist = Huff<8>;

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the header of a dynamic block behind its three type bits (RFC 1951 3.2.7).  strict: what the block SEARCH demands of a candidate
// (a complete literal / length code, a distance code) -- zlib's encoder always writes such headers; the decoder proper accepts what
// zlib's inflate accepts.
inline bool dynamic_header(Bits &b, LitLen &ll, Dist &dd, bool strict) {
    b.refill();
    const unsigned hlit = b.get(5) + 257, hdist = b.get(5) + 1, hclen = b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t pre[19] = {0};
    b.refill();
    for (unsigned i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); pre[kPreOrder[i]] = (uint8_t)b.get(3); }
    Huff<7> pc;
    if (pc.build(pre, 19) != 2) return false;   // (zlib's inflate takes no incomplete code-length code either)
    uint8_t len[320];
    unsigned i = 0;
    while (i < hlit + hdist) {
        b.refill();
        const int s = pc.decode(b);
        if (s < 0) return false;
        if (s < 16) { len[i++] = (uint8_t)s; continue; }
        unsigned rep;
        uint8_t v = 0;
        if (s == 16) { if (i == 0) return false; v = len[i - 1]; rep = 3 + b.get(2); }
        else if (s == 17) rep = 3 + b.get(3);
        else rep = 11 + b.get(7);
        if (i + rep > hlit + hdist) return false;
        while (rep--) len[i++] = v;
    }
    if (b.over() || len[256] == 0) return false;
    const int lst = ll.build(len, (int)hlit);
    if (lst == 0 || (lst == 1 && (strict || ll.maxlen > 1))) return false;
    const int dst = dd.build(len + hlit, (int)hdist);
    if (dst == 0) {   // no distance code at all is legal when the block has no matches (all lengths zero): zlib writes one code anyway
        bool any = false;
        for (unsigned j = 0; j < hdist; ++j) any |= len[hlit + j] != 0;
        if (any || strict) return false;
        memset(dd.fast, 0, sizeof dd.fast); dd.maxlen = 0;
    } else if (dst == 1 && dd.maxlen > 1) return false;   // an incomplete distance code: only a single one-bit code is legal (zlib's rule)
    return true;
}

struct FixedTables {
    LitLen ll;
    Dist dd;
    FixedTables() {
        uint8_t l[288], d[30];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 30; ++i) d[i] = 5;
        ll.build(l, 288);
        dd.build(d, 30);
    }
};
inline const FixedTables &fixed_tables() { static FixedTables t; return t; }

// output with a KNOWN history (the stream's first chunk): bytes straight to their place
struct ByteOut {
    uint8_t *dst;
    size_t cap, pos = 0;
    inline bool lit(unsigned c) { if (pos >= cap) return false; dst[pos++] = (uint8_t)c; return true; }
    inline bool copy(unsigned len, unsigned dist) {
        if (dist > pos || pos + len > cap) return false;
        const uint8_t *s = dst + pos - dist;
        uint8_t *d = dst + pos;
        if (dist >= len) memcpy(d, s, len);
        else for (unsigned i = 0; i < len; ++i) d[i] = s[i];
        pos += len;
        return true;
    }
    inline bool raw(const uint8_t *s, size_t n) { if (pos + n > cap) return false; memcpy(dst + pos, s, n); pos += n; return true; }
    size_t size() const { return pos; }
};

// output with a KNOWN 32 KiB in front and no known size (the streaming form's first piece, and its serial fall-back): v = the window, then the text
struct ByteOutW {
    std::vector<uint8_t> v;      // [0, 32768) = the window in front (zeros where the stream is younger than that), text behind it
    size_t limit = ~(size_t)0;
    explicit ByteOutW(const uint8_t *window) : v(window, window + 32768) {}
    inline bool lit(unsigned c) { if (v.size() - 32768 >= limit) return false; v.push_back((uint8_t)c); return true; }
    inline bool copy(unsigned len, unsigned dist) {
        const size_t pos = v.size();
        if (dist > pos || pos - 32768 + len > limit) return false;
        v.resize(pos + len);
        uint8_t *d = v.data() + pos;
        const uint8_t *s = d - dist;
        if (dist >= len) memcpy(d, s, len);
        else for (unsigned i = 0; i < len; ++i) d[i] = s[i];
        return true;
    }
    inline bool raw(const uint8_t *s, size_t n) { if (v.size() - 32768 + n > limit) return false; v.insert(v.end(), s, s + n); return true; }
    size_t size() const { return v.size() - 32768; }
    const uint8_t *text() const { return v.data() + 32768; }
};

// output with an UNKNOWN 32 KiB in front: 16-bit symbols, 0x8000 + j = byte j of that window (j = 32768 - distance before the chunk)
struct MarkOut {
    uint16_t *v = nullptr;
    size_t n = 0, cap = 0, limit = 0, mapped = 0;
    MarkOut() = default;
    MarkOut(const MarkOut &) = delete;
    MarkOut &operator=(const MarkOut &) = delete;
    ~MarkOut() { release(); }
    // (pages first, under the address-space lock held for reading; the munmap behind it -- the write lock, which stops every other thread's
    // page faults and mmaps -- then has nothing to free)
    void release() { if (v) { (void)madvise(v, mapped, MADV_DONTNEED); munmap(v, mapped); } v = nullptr; n = cap = mapped = 0; }
    // The buffers of all pieces together are twice the text: anonymous mappings that ask for huge pages and grow by mremap (with 4 KiB
    // pages, sixty-four threads faulting in 1.2 GB at once spent more time in the kernel's address-space lock than inflating).
    bool room(size_t more) {   // (258 symbols of slack are kept beyond n: lit() and copy() of one code never check again)
        if (n + more + 258 <= cap) return true;
        if (n + more > limit) return false;
        size_t want = std::max<size_t>(cap * 2, n + more + 258 + 65536);
        if (want > limit + 258 + 65536) want = limit + 258 + 65536;
        const size_t bytes = (want * sizeof(uint16_t) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *q = v ? mremap(v, mapped, bytes, MREMAP_MAYMOVE) : mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        v = (uint16_t *)q; mapped = bytes; cap = bytes / sizeof(uint16_t);
        return true;
    }
    inline bool lit(unsigned c) { if (n + 259 > cap && !room(1)) return false; v[n++] = (uint16_t)c; return true; }
    inline bool copy(unsigned len, unsigned dist) {
        if (n + len + 258 > cap && !room(len)) return false;
        if (n + len > limit) return false;
        uint16_t *d = v + n;
        if (dist <= n) {
            const uint16_t *s = d - dist;
            if (dist >= len) memcpy(d, s, len * sizeof(uint16_t));
            else for (unsigned i = 0; i < len; ++i) d[i] = s[i];
        } else {
            for (unsigned i = 0; i < len; ++i) {
                const long long src = (long long)n + i - dist;
                d[i] = src >= 0 ? v[(size_t)src] : (uint16_t)(0x8000 + (32768 + src));
            }
        }
        n += len;
        return true;
    }
    inline bool raw(const uint8_t *s, size_t m) {
        if (!room(m) || n + m > limit) return false;
        for (size_t i = 0; i < m; ++i) v[n + i] = s[i];
        n += m;
        return true;
    }
    size_t size() const { return n; }
};

// 16-bit symbols -> bytes, markers looked up in the 32 KiB window `w` in front of the piece
inline void resolve(const uint16_t *m, size_t n, const uint8_t *w, uint8_t *d) {
    size_t x = 0;
#if defined(__SSE2__)
    for (; x + 16 <= n; x += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(m + x)), b = _mm_loadu_si128((const __m128i *)(m + x + 8));
        if (_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) {   // a marker among the sixteen (bit 15 of a symbol)
            for (size_t y = x; y < x + 16; ++y) { const uint16_t s = m[y]; d[y] = s < 0x8000 ? (uint8_t)s : w[s - 0x8000]; }
        } else _mm_storeu_si128((__m128i *)(d + x), _mm_packus_epi16(a, b));
    }
#endif
    for (; x < n; ++x) { const uint16_t s = m[x]; d[x] = s < 0x8000 ? (uint8_t)s : w[s - 0x8000]; }
}

template <class Out>
inline bool block_codes(Bits &b, const LitLen &ll, const Dist &dd, Out &out) {
    for (;;) {
        b.refill();   // >= 56 bits: two table-decoded literals (<= 22 bits) and one full length / distance pair (<= 20 + 28) fit
        uint16_t e = ll.fast[b.peek(11)];
        if (e && e < (256u << 4)) {
            b.drop(e & 15);
            if (!out.lit(e >> 4)) return false;
            e = ll.fast[b.peek(11)];
            if (e && e < (256u << 4)) {
                b.drop(e & 15);
                if (!out.lit(e >> 4)) return false;
                continue;
            }
            if (b.cnt < 48) b.refill();
        }
        int s = ll.decode(b);
        if (s < 0) return false;
        if (s < 256) { if (!out.lit((unsigned)s)) return false; continue; }
        if (s == 256) return !b.over();
        s -= 257;
        if (s >= 29) return false;
        const unsigned len = kLenBase[s] + b.get(kLenExtra[s]);
        if (b.cnt < 28) b.refill();
        const int d = dd.decode(b);
        if (d < 0 || d >= 30) return false;
        const unsigned dist = kDistBase[d] + b.get(kDistExtra[d]);
        if (!out.copy(len, dist)) return false;
        if (b.over()) return false;
    }
}

enum { RUN_ERROR = 0, RUN_REACHED = 1, RUN_FINAL = 2, RUN_STOPPED = 3 };

// Inflates whole blocks from the reader's position.  targets[ti...] are bit positions (ascending) of later block-start candidates: the
// run ends when a block boundary falls EXACTLY on one (*reached = its index); candidates that are passed are skipped.  Also ends behind
// the final block (*end_bit = the position behind it).
// stop_at (the streaming form): the run also ends at the first block boundary at or behind this bit position (*end_bit = the boundary).
template <class Out>
inline int run_blocks(Bits &b, Out &out, const uint64_t *targets, size_t ntargets, size_t *reached, uint64_t *end_bit, uint64_t stop_at = ~0ULL) {
    size_t ti = 0;
    LitLen ll;
    Dist dd;
    for (;;) {
        const uint64_t at = b.pos();
        while (ti < ntargets && targets[ti] < at) ++ti;
        if (ti < ntargets && targets[ti] == at) { *reached = ti; return RUN_REACHED; }
        if (at >= stop_at) { *end_bit = at; return RUN_STOPPED; }
        b.refill();
        const unsigned final = b.get(1), type = b.get(2);
        if (type == 0) {
            b.drop(b.cnt & 7);   // to the byte boundary
            b.refill();
            const unsigned len = b.get(16), nlen = b.get(16);
            if ((len ^ nlen) != 0xFFFFu) return RUN_ERROR;
            const uint64_t p0 = b.pos();
            if ((p0 & 7) || (p0 >> 3) + len > b.n) return RUN_ERROR;
            if (!out.raw(b.p + (p0 >> 3), len)) return RUN_ERROR;
            b.seek(p0 + 8ULL * len);
        } else if (type == 1) {
            if (!block_codes(b, fixed_tables().ll, fixed_tables().dd, out)) return RUN_ERROR;
        } else if (type == 2) {
            if (!dynamic_header(b, ll, dd, false)) return RUN_ERROR;
            if (!block_codes(b, ll, dd, out)) return RUN_ERROR;
        } else return RUN_ERROR;
        if (final) { *end_bit = b.pos(); return b.over() ? RUN_ERROR : RUN_FINAL; }
    }
}

// the first bit position in [from, to) at which a non-final dynamic block's header parses under the strict rules; ~0 if none
inline uint64_t find_block(const uint8_t *p, size_t n, uint64_t from, uint64_t to) {
    Bits b(p, n);
    LitLen ll;
    Dist dd;
    for (uint64_t q = from; q < to; ++q) {
        // cheap rejection on the first 17 bits: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29
        const size_t by = (size_t)(q >> 3);
        if (by + 4 > n) break;
        uint32_t w;
        memcpy(&w, p + by, 4);
        w >>= (q & 7);
        if ((w & 7u) != 4u) continue;                       // bits: 0 (not final), then 0 1 = type 2 LSB-first
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        // the code-length code must be complete (RFC 1951 3.2.7: (HCLEN + 4) x 3 bits behind the 17): its Kraft sum straight from the bits
        // turns away all but a few per cent of what is left, before any table is built
        {
            const unsigned hclen = ((w >> 13) & 15u) + 4;
            const uint64_t q2 = q + 17;
            const size_t b2 = (size_t)(q2 >> 3);
            if (b2 + 9 > n) break;
            uint64_t lo;
            memcpy(&lo, p + b2, 8);
            const unsigned sh = (unsigned)(q2 & 7);
            if (sh) lo = (lo >> sh) | ((uint64_t)p[b2 + 8] << (64 - sh));
            unsigned kraft = 0;
            for (unsigned i = 0; i < hclen; ++i) {
                const unsigned l = (unsigned)(lo >> (3 * i)) & 7u;   // (57 bits at most)
                kraft += l ? 128u >> l : 0u;
            }
            if (kraft != 128u) continue;
        }
        b.seek(q + 3);
        if (dynamic_header(b, ll, dd, true)) return q;
    }
    return ~0ULL;
}

// gz: one gzip member (RFC 1952); out: room for `out_size` bytes = the member's ISIZE.  true = out holds the member's text, verified by
// length and CRC-32; false = not done (not a single member, too small to be worth it, or anything irregular): inflate it the ordinary way.
inline bool gunzip_parallel(const uint8_t *gz, size_t gz_size, uint8_t *out, size_t out_size, unsigned nthreads, size_t chunk_bytes = 0) {
    if (gz_size < 18 + 8 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8 || (gz[3] & 0xE0)) return false;
    size_t h = 10;
    const unsigned flg = gz[3];
    if (flg & 4) { if (h + 2 > gz_size) return false; h += 2 + (gz[h] | (gz[h + 1] << 8)); }
    if (flg & 8) { while (h < gz_size && gz[h]) ++h; ++h; }
    if (flg & 16) { while (h < gz_size && gz[h]) ++h; ++h; }
    if (flg & 2) h += 2;
    if (h + 8 >= gz_size) return false;
    const uint8_t *def = gz + h;
    const size_t def_size = gz_size - h - 8;
    uint32_t want_crc, want_size;
    memcpy(&want_crc, gz + gz_size - 8, 4);
    memcpy(&want_size, gz + gz_size - 4, 4);
    if ((uint32_t)out_size != want_size) return false;
    if (!chunk_bytes) chunk_bytes = std::max<size_t>(256 << 10, def_size / (2 * (size_t)std::max(1u, nthreads)));   // two even rounds of pieces
    const size_t nchunks = (def_size + chunk_bytes - 1) / chunk_bytes;
    if (nchunks < 2 || nthreads < 2) return false;
    nthreads = (unsigned)std::min<size_t>(nthreads, nchunks);

    const bool timing = getenv("PGZ_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // ---- 1: every chunk's first block start --------------------------------------------------------------------------------------------
    std::vector<uint64_t> start(nchunks, ~0ULL);
    start[0] = 0;
    {
        std::atomic<size_t> next{1};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&] {
                for (size_t c; (c = next.fetch_add(1)) < nchunks;) start[c] = find_block(def, def_size, 8ULL * c * chunk_bytes, 8ULL * std::min(def_size, (c + 1) * chunk_bytes));
            });
        for (auto &x : th) x.join();
    }
    const double t_found = now();
    std::vector<size_t> cand;        // chunks that have a start, ascending
    for (size_t c = 0; c < nchunks; ++c) if (start[c] != ~0ULL) cand.push_back(c);
    std::vector<uint64_t> cstart(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) cstart[i] = start[cand[i]];

    // ---- 2: inflate from every start to the next start that is really a block boundary --------------------------------------------------
    struct Piece { MarkOut marks; int status = RUN_ERROR; size_t next = 0; uint64_t end_bit = 0; size_t bytes0 = 0; };
    std::vector<Piece> piece(cand.size());
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&] {
                for (size_t i; (i = next.fetch_add(1)) < cand.size();) {
                    Piece &pc = piece[i];
                    Bits b(def, def_size);
                    b.seek(cstart[i]);
                    size_t reached = 0;
                    const uint64_t *tg = cstart.data() + i + 1;
                    const size_t ntg = cand.size() - i - 1;
                    if (i == 0) {
                        ByteOut bo{out, out_size};
                        pc.status = run_blocks(b, bo, tg, ntg, &reached, &pc.end_bit);
                        pc.bytes0 = bo.pos;
                    } else {
                        pc.marks.limit = out_size;
                        // (room for this piece's text at the member's overall ratio + a quarter: one mapping, rarely a second)
                        (void)pc.marks.room(std::min<size_t>(out_size, (size_t)((double)out_size / (double)def_size * 1.25 * (double)chunk_bytes) + 65536));
                        pc.status = run_blocks(b, pc.marks, tg, ntg, &reached, &pc.end_bit);
                    }
                    pc.next = i + 1 + reached;
                }
            });
        for (auto &x : th) x.join();
    }
    const double t_inflated = now();
    // the chain of pieces that are real: piece 0, the piece it arrived at, ...; it must end behind the final block, at the trailer
    std::vector<size_t> chain;
    for (size_t i = 0;;) {
        chain.push_back(i);
        if (piece[i].status == RUN_FINAL) { if ((piece[i].end_bit + 7) / 8 != def_size) return false; break; }
        if (piece[i].status != RUN_REACHED || piece[i].next >= cand.size()) return false;
        i = piece[i].next;
    }
    std::vector<size_t> off(chain.size() + 1, 0);
    for (size_t j = 0; j < chain.size(); ++j) off[j + 1] = off[j] + (j == 0 ? piece[chain[0]].bytes0 : piece[chain[j]].marks.size());
    if (off[chain.size()] != out_size) return false;

    // ---- 3: the 32 KiB in front of every piece, front to back ---------------------------------------------------------------------------
    constexpr size_t W = 32768;
    std::vector<std::vector<uint8_t>> win(chain.size());   // win[j]: the window in front of piece j (index 0 = oldest), W bytes (zeros where the text is shorter)
    for (size_t j = 1; j < chain.size(); ++j) {
        win[j].assign(W, 0);
        const size_t have = off[j];                          // bytes of text in front of piece j
        if (j == 1) {
            const size_t n = std::min(W, have);
            memcpy(win[j].data() + W - n, out + have - n, n);
        } else {
            const MarkOut &m = piece[chain[j - 1]].marks;
            const size_t n = std::min(W, m.n);
            resolve(m.v + (m.n - n), n, win[j - 1].data(), win[j].data() + (W - n));
            if (n < W) memcpy(win[j].data(), win[j - 1].data() + n, W - n);   // a short piece: the older part slides down
        }
    }
    const double t_windows = now();
    // ---- 4: symbols -> bytes at their place; 5: CRC ----------------------------------------------------------------------------------------
    std::vector<uint32_t> crcs(chain.size(), 0);
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&] {
                for (size_t j; (j = next.fetch_add(1)) < chain.size();) {
                    if (j) resolve(piece[chain[j]].marks.v, piece[chain[j]].marks.n, win[j].data(), out + off[j]);
                    uint32_t c = 0;
                    for (size_t a = off[j]; a < off[j + 1];) {   // (zlib's crc32 takes a uInt length)
                        const size_t n = std::min<size_t>(off[j + 1] - a, 1u << 30);
                        c = crc_impl()(c, out + a, n);
                        a += n;
                    }
                    crcs[j] = c;
                    // The piece's 16-bit buffer (twice its text) goes back to the kernel here, by the thread that is done with it:
                    // MADV_DONTNEED takes the address-space lock for READING, so the threads do it side by side, and the munmap that
                    // follows finds nothing left to do.  (munmap itself -- 1.2 GB, ~45 ms under the write lock whether one thread did
                    // it or thirty-two -- was first given to a thread nobody waited for, which held that lock against the page faults
                    // of the parsers behind the inflater.)
                    MarkOut &mk = piece[chain[j]].marks;
                    if (mk.v) (void)madvise(mk.v, mk.mapped, MADV_DONTNEED);
                }
            });
        for (auto &x : th) x.join();
    }
    if (timing) fprintf(stderr, "pgz: %zu chunks, %zu starts, %zu pieces in the chain; search %.1f ms, inflate %.1f ms, windows %.1f ms, bytes + crc %.1f ms\n", nchunks, cand.size(),
                        chain.size(), t_found - t_begin, t_inflated - t_found, t_windows - t_inflated, now() - t_windows);
    uint32_t crc = crcs[0];
    for (size_t j = 1; j < chain.size(); ++j) crc = (uint32_t)crc32_combine(crc, crcs[j], (z_off_t)(off[j + 1] - off[j]));
    return crc == want_crc;
}


// ---- the streaming form: a member of any size, a WINDOW of its compressed bytes at a time (kct_ingest.hip's producer for texts beyond what the
// one-piece form above may hold in memory) --------------------------------------------------------------------------------------------------
// The state between windows is a block boundary and the 32 KiB of text in front of it; a window is the blocks that start in the next `span` compressed
// bytes, inflated like a whole member above -- piece 0 begins at the known boundary with the known window and is decoded as bytes, the other
// pieces begin at searched starts with unknown windows -- except that nothing can be verified until the member's trailer: a window whose pieces do
// not chain up is inflated again by ONE thread from its known boundary (always possible), and the CRC-32 of everything is checked at the end of the
// member, where a mismatch is an error (as it is for zlib): text has been handed on by then.
struct MemberStream {
    const uint8_t *def = nullptr;     // the member's deflate data
    size_t def_size = 0;              // ... up to the end of the FILE (the member's own end is found by inflating)
    uint64_t bit = 0;                 // the next block boundary
    uint8_t window[32768];            // the text in front of it
    bool done = false;                // the final block has been inflated: `bit` is the position behind it
    uint32_t crc = 0;
    uint64_t total = 0;
    unsigned windows = 0, serial_windows = 0;   // statistics: windows inflated, windows that fell back to one thread
    MemberStream() { memset(window, 0, sizeof window); }
};

// A window's text: storage that is not initialised and is kept, with its pages, from one window to the next (a std::vector's resize
// wrote 146 MB of zeros per window on one thread -- more time than the inflating took); `lead` bytes stay free in front of the text
// for what the caller carries over from the window before.
struct TextBuf {
    uint8_t *p = nullptr;
    size_t cap = 0, lead = 0, n = 0;
    TextBuf() = default;
    TextBuf(const TextBuf &) = delete;
    TextBuf &operator=(const TextBuf &) = delete;
    ~TextBuf() { if (p) { (void)madvise(p, cap, MADV_DONTNEED); munmap(p, cap); } }
    uint8_t *data() const { return p + lead; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    bool resize(size_t m) {   // (what was there is NOT kept)
        if (!p || lead + m > cap) {
            if (p) munmap(p, cap);
            p = nullptr;
            const size_t bytes = (lead + m + m / 8 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) { cap = n = 0; return false; }
            (void)madvise(q, bytes, MADV_HUGEPAGE);
            p = (uint8_t *)q; cap = bytes;
        }
        n = m;
        return true;
    }
};

// the 16-bit piece buffers of a window, kept for the next one (mapping, faulting in and unmapping ~300 MB per window otherwise)
struct WindowScratch { std::deque<MarkOut> marks; };

// inflates the next window of `st` into `text` (replaced); false = the stream is corrupt (or out of memory)
inline bool inflate_window(MemberStream &st, size_t span, unsigned nthreads, TextBuf &text, WindowScratch &scratch) {
    text.clear();
    if (st.done) return true;
    constexpr size_t W = 32768;
    const uint64_t stop_at = st.bit + 8ULL * span;
    const size_t byte0 = (size_t)(st.bit >> 3);
    const size_t chunk_bytes = std::max<size_t>(256 << 10, span / (2 * (size_t)std::max(1u, nthreads)));
    const size_t nchunks = std::max<size_t>(1, (std::min(span, st.def_size - std::min(st.def_size, byte0)) + chunk_bytes - 1) / chunk_bytes);
    ++st.windows;
    const bool timing = getenv("PGZ_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t_find = t0, t_dec = t0, t_res = t0;
    bool chained = false;
    std::vector<uint64_t> cstart{st.bit};
    if (nthreads >= 2 && nchunks >= 2) {
        std::vector<uint64_t> start(nchunks, ~0ULL);
        {
            std::atomic<size_t> next{1};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < std::min<size_t>(nthreads, nchunks); ++t)
                th.emplace_back([&] {
                    for (size_t c; (c = next.fetch_add(1)) < nchunks;) {
                        const uint64_t from = 8ULL * (byte0 + c * chunk_bytes), to = std::min<uint64_t>(8ULL * st.def_size, std::min<uint64_t>(stop_at, from + 8ULL * chunk_bytes));
                        if (from < to) start[c] = find_block(st.def, st.def_size, from, to);
                    }
                });
            for (auto &x : th) x.join();
        }
        for (size_t c = 1; c < nchunks; ++c) if (start[c] != ~0ULL && start[c] > st.bit) cstart.push_back(start[c]);
    }
    t_find = now();
    struct Piece { MarkOut *marks = nullptr; int status = RUN_ERROR; size_t next = 0; uint64_t end_bit = 0; };
    std::vector<Piece> piece(cstart.size());
    while (scratch.marks.size() < cstart.size()) scratch.marks.emplace_back();
    for (size_t i = 0; i < cstart.size(); ++i) { piece[i].marks = &scratch.marks[i]; piece[i].marks->n = 0; }
    ByteOutW first(st.window);
    if (cstart.size() > 1) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < std::min<size_t>(nthreads, cstart.size()); ++t)
            th.emplace_back([&] {
                for (size_t i; (i = next.fetch_add(1)) < cstart.size();) {
                    Piece &pc = piece[i];
                    Bits b(st.def, st.def_size);
                    b.seek(cstart[i]);
                    size_t reached = 0;
                    const uint64_t *tg = cstart.data() + i + 1;
                    const size_t ntg = cstart.size() - i - 1;
                    if (i == 0) pc.status = run_blocks(b, first, tg, ntg, &reached, &pc.end_bit, stop_at);
                    else {
                        pc.marks->limit = 1032 * (span + chunk_bytes) + 65536;   // (deflate cannot expand further)
                        (void)pc.marks->room(6 * chunk_bytes);
                        pc.status = run_blocks(b, *pc.marks, tg, ntg, &reached, &pc.end_bit, stop_at);
                    }
                    pc.next = i + 1 + reached;
                }
            });
        for (auto &x : th) x.join();
        t_dec = now();
        // the chain: piece 0, the piece it arrived at, ...; it ends where a piece stopped at the window's end or behind the final block
        std::vector<size_t> chain;
        bool ok = true, fin = false;
        uint64_t end_bit = 0;
        for (size_t i = 0;;) {
            chain.push_back(i);
            if (piece[i].status == RUN_FINAL || piece[i].status == RUN_STOPPED) { fin = piece[i].status == RUN_FINAL; end_bit = piece[i].end_bit; break; }
            if (piece[i].status != RUN_REACHED || piece[i].next >= cstart.size()) { ok = false; break; }
            i = piece[i].next;
        }
        if (ok) {
            std::vector<size_t> off(chain.size() + 1, 0);
            for (size_t j = 0; j < chain.size(); ++j) off[j + 1] = off[j] + (j == 0 ? first.size() : piece[chain[j]].marks->size());
            if (!text.resize(off[chain.size()])) return false;
            memcpy(text.data(), first.text(), first.size());
            // the window in front of every piece, front to back: the last 32 KiB of (what lay in front of the previous piece ++ that piece's text)
            std::vector<std::vector<uint8_t>> win(chain.size());
            for (size_t j = 1; j < chain.size(); ++j) {
                win[j].assign(W, 0);
                if (j == 1) memcpy(win[j].data(), first.v.data() + first.v.size() - W, W);   // (first.v begins with the stream's own window: always >= W bytes)
                else {
                    const MarkOut &m = *piece[chain[j - 1]].marks;
                    const size_t n = std::min(W, m.n);
                    resolve(m.v + (m.n - n), n, win[j - 1].data(), win[j].data() + (W - n));
                    if (n < W) memcpy(win[j].data(), win[j - 1].data() + n, W - n);
                }
            }
            std::atomic<size_t> nx{1};
            std::vector<std::thread> th2;
            for (unsigned t = 0; t < std::min<size_t>(nthreads, chain.size()); ++t)
                th2.emplace_back([&] { for (size_t j; (j = nx.fetch_add(1)) < chain.size();) resolve(piece[chain[j]].marks->v, piece[chain[j]].marks->n, win[j].data(), text.data() + off[j]); });
            for (auto &x : th2) x.join();
            st.bit = end_bit; st.done = fin;
            chained = true;
        }
    }
    if (!chained) {   // one piece, or pieces that did not chain up: one thread from the known boundary
        ++st.serial_windows;
        ByteOutW bo(st.window);
        Bits b(st.def, st.def_size);
        b.seek(st.bit);
        size_t reached = 0;
        uint64_t end_bit = 0;
        const int rc = run_blocks(b, bo, nullptr, 0, &reached, &end_bit, stop_at);
        if (rc != RUN_FINAL && rc != RUN_STOPPED) return false;
        if (!text.resize(bo.size())) return false;
        memcpy(text.data(), bo.text(), bo.size());
        st.bit = end_bit; st.done = rc == RUN_FINAL;
    }
    t_res = now();
    // the state for the next window: the last 32 KiB of (window ++ text), the running CRC and length
    if (text.size() >= W) memcpy(st.window, text.data() + text.size() - W, W);
    else if (!text.empty()) { memmove(st.window, st.window + text.size(), W - text.size()); memcpy(st.window + W - text.size(), text.data(), text.size()); }
    {
        const size_t n = text.size(), parts = std::max<size_t>(1, std::min<size_t>(nthreads, n >> 20));
        std::vector<uint32_t> cr(parts, 0);
        std::vector<std::thread> th;
        for (size_t q = 0; q < parts; ++q) th.emplace_back([&, q] { const size_t a = n * q / parts, e = n * (q + 1) / parts; uint32_t c = 0; for (size_t x = a; x < e;) { const size_t m = std::min<size_t>(e - x, 1u << 30); c = crc_impl()(c, text.data() + x, m); x += m; } cr[q] = c; });
        for (auto &x : th) x.join();
        for (size_t q = 0; q < parts; ++q) st.crc = (uint32_t)crc32_combine(st.crc, cr[q], (z_off_t)(n * (q + 1) / parts - n * q / parts));
        st.total += n;
    }
    (void)t_find; (void)t_dec;
    if (timing) fprintf(stderr, "pgz window %llu: %zu pieces, text %zu: find %.1f decode %.1f resolve %.1f crc %.1f ms%s\n", (unsigned long long)st.windows, cstart.size(),
                        text.size(), t_find - t0, t_dec - t_find, t_res - t_dec, now() - t_res, chained ? "" : " (one thread)");
    return true;
}

// (the same into a std::vector, with scratch of its own: tests)
inline bool inflate_window(MemberStream &st, size_t span, unsigned nthreads, std::vector<uint8_t> &text) {
    TextBuf tb;
    WindowScratch sc;
    if (!inflate_window(st, span, nthreads, tb, sc)) return false;
    text.assign(tb.data(), tb.data() + tb.size());
    return true;
}

// the length of a gzip member's header at p (RFC 1952), 0 if it is none
inline size_t gzip_header(const uint8_t *p, size_t n) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    size_t h = 10;
    const unsigned flg = p[3];
    if (flg & 4) { if (h + 2 > n) return 0; h += 2 + (p[h] | (p[h + 1] << 8)); }
    if (flg & 8) { while (h < n && p[h]) ++h; ++h; }
    if (flg & 16) { while (h < n && p[h]) ++h; ++h; }
    if (flg & 2) h += 2;
    return h + 8 <= n ? h : 0;
}

}  // namespace pgz
