// kct_entry.hip -- the C-ABI entry points of bulk ingest and window hashing, and what sits between the caller's host buffers and the
// device passes of kct_consume.hip: pinned staging, the host packer (ASCII -> packed base arrays), deferred mode.
#include "kct_internal.h"

#include <emmintrin.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include "window_kernels.h"
#include "stream_kernels.h"

namespace kcth {

template <template <int, int> class Launcher, class... Args>
void dispatch_k(int k, Args &&...args) {
    if (k == 21) Launcher<1, 21>::run(args...);
    else if (k == 31) Launcher<1, 31>::run(args...);
    else if (k == 51) Launcher<2, 51>::run(args...);
    else if (k <= 32) Launcher<1, 0>::run(args...);
    else if (k <= 64) Launcher<2, 0>::run(args...);
    else Launcher<0, 0>::run(args...);
}

template <int KW, int KC>
struct HashLauncher {
    static void run(hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, int k, u64 nwin, du64 *out, du64 *fb) {
        hipLaunchKernelGGL((kct::hash_windows_kernel<KW, KC>), dim3(grid), dim3(kct::kBlock), 0, s, stream, nbytes, k, nwin, out, fb);
    }
};

// the ASCII image of ng groups of a packed stream, in t->d_unpack (for the kernels that read bytes)
kct_status unpack_stream(kct_table *t, const unsigned int *d_codes, const unsigned short *d_valid, u64 ng) {
    KCT_TRY(t->d_unpack.reserve(ng * 16 + 16));
    ProfScope ps(t, "unpack_stream_kernel");
    hipLaunchKernelGGL(kct::unpack_stream_kernel, dim3((unsigned)std::min<u64>((ng + kct::kBlock - 1) / kct::kBlock, 1u << 16)), dim3(kct::kBlock), 0, t->stream,
                       d_codes, d_valid, ng, (unsigned char *)t->d_unpack.p);
    HIP_TRY(hipGetLastError());
    return KCT_OK;
}

// ---- host packer: ASCII -> 2-bit codes + validity bits (the host twin of kmer_device.h encode16) -----------------------
#include <tmmintrin.h>
#include <immintrin.h>
// [host-packer-begin]  (tests/test_host_packer.py compiles this block alone and checks the SIMD encoders against the scalar one)
static inline void encode16_scalar(const unsigned char *p, unsigned int *codes, unsigned short *valid) {
    unsigned int c = 0, v = 0;
    for (int i = 0; i < 16; ++i) {
        const unsigned b = p[i] | 0x20u;
        const bool ok = b == 'a' || b == 'c' || b == 'g' || b == 't';
        unsigned int x = (p[i] >> 1) & 3u;
        x ^= x >> 1;  // A0 C1 G2 T3
        c = (c << 2) | (ok ? x : 0u);
        v = (v << 1) | (ok ? 1u : 0u);
    }
    *codes = c; *valid = (unsigned short)v;
}
__attribute__((target("ssse3"))) static inline void encode16_ssse3(const unsigned char *p, unsigned int *codes, unsigned short *valid) {
    const __m128i v = _mm_loadu_si128((const __m128i *)p), up = _mm_or_si128(v, _mm_set1_epi8(0x20));
    const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(up, _mm_set1_epi8('a')), _mm_cmpeq_epi8(up, _mm_set1_epi8('c'))),
                                    _mm_or_si128(_mm_cmpeq_epi8(up, _mm_set1_epi8('g')), _mm_cmpeq_epi8(up, _mm_set1_epi8('t'))));
    __m128i x = _mm_and_si128(_mm_srli_epi16(v, 1), _mm_set1_epi8(3));
    x = _mm_xor_si128(x, _mm_and_si128(_mm_srli_epi16(x, 1), _mm_set1_epi8(1)));
    x = _mm_and_si128(x, ok);
    const __m128i p2 = _mm_maddubs_epi16(x, _mm_set1_epi16(0x0104));      // base 2i * 4 + base 2i+1
    const __m128i p4 = _mm_madd_epi16(p2, _mm_set1_epi32(0x00010010));    // pair 2j * 16 + pair 2j+1: four bases per 32-bit lane
    const __m128i sh = _mm_shuffle_epi8(p4, _mm_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12));
    *codes = (unsigned int)_mm_cvtsi128_si32(sh);                         // bases 0-3 in the top byte
    *valid = (unsigned short)(__builtin_bitreverse16((unsigned short)_mm_movemask_epi8(ok)));
}
// whole runs of groups, the loop INSIDE the function that carries the target attribute (a target("...") function is not inlined into a
// caller without it: one call per 16 bases otherwise)
__attribute__((target("ssse3"))) static void encode_run_ssse3(const unsigned char *p, size_t ngroups, unsigned int *codes, unsigned short *valid) {
    for (size_t g = 0; g < ngroups; ++g) encode16_ssse3(p + 16 * g, codes + g, valid + g);
}
// 32 bases per step: the same arithmetic on two 128-bit lanes
__attribute__((target("avx2"))) static void encode_run_avx2(const unsigned char *p, size_t ngroups, unsigned int *codes, unsigned short *valid) {
    const __m256i lower = _mm256_set1_epi8(0x20), ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1), m2 = _mm256_set1_epi16(0x0104), m4 = _mm256_set1_epi32(0x00010010);
    const __m256i pick = _mm256_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12);
    size_t g = 0;
    for (; g + 2 <= ngroups; g += 2) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(p + 16 * g)), up = _mm256_or_si256(v, lower);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(up, ca), _mm256_cmpeq_epi8(up, cc)), _mm256_or_si256(_mm256_cmpeq_epi8(up, cg), _mm256_cmpeq_epi8(up, ct)));
        __m256i x = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
        x = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), one));
        x = _mm256_and_si256(x, ok);
        const __m256i sh = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(x, m2), m4), pick);
        codes[g] = (unsigned int)_mm256_extract_epi32(sh, 0);
        codes[g + 1] = (unsigned int)_mm256_extract_epi32(sh, 4);
        // (one 32-bit reversal: the first group's sixteen bits come out in the upper half.  Two __builtin_bitreverse16 of the halves
        // were miscompiled by this clang at -O2 -- the halves came out swapped; enc unit test in tests/test_host_packer.py)
        const unsigned int rv = __builtin_bitreverse32((unsigned int)_mm256_movemask_epi8(ok));
        valid[g] = (unsigned short)(rv >> 16);
        valid[g + 1] = (unsigned short)rv;
    }
    // the odd group at the end: the same arithmetic on one 128-bit lane.  (Until round 6 it went through encode16_scalar -- and so did every
    // one-group call for the group that straddles a record boundary: two scalar groups per 150 bp read, more time than the read's
    // other eight groups together.  kct_batch_timeline put the packers at 2.9-3.4 GB/s per thread.)
    for (; g < ngroups; ++g) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(p + 16 * g)), up = _mm_or_si128(v, _mm256_castsi256_si128(lower));
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(up, _mm256_castsi256_si128(ca)), _mm_cmpeq_epi8(up, _mm256_castsi256_si128(cc))),
                                        _mm_or_si128(_mm_cmpeq_epi8(up, _mm256_castsi256_si128(cg)), _mm_cmpeq_epi8(up, _mm256_castsi256_si128(ct))));
        __m128i x = _mm_and_si128(_mm_srli_epi16(v, 1), _mm256_castsi256_si128(three));
        x = _mm_xor_si128(x, _mm_and_si128(_mm_srli_epi16(x, 1), _mm256_castsi256_si128(one)));
        x = _mm_and_si128(x, ok);
        const __m128i sh = _mm_shuffle_epi8(_mm_madd_epi16(_mm_maddubs_epi16(x, _mm256_castsi256_si128(m2)), _mm256_castsi256_si128(m4)), _mm256_castsi256_si128(pick));
        codes[g] = (unsigned int)_mm_cvtsi128_si32(sh);
        valid[g] = (unsigned short)(__builtin_bitreverse32((unsigned int)_mm_movemask_epi8(ok)) >> 16);
    }
}
static void encode_groups(const unsigned char *p, size_t ngroups, unsigned int *codes, unsigned short *valid) {
    static const int level = __builtin_cpu_supports("avx2") ? 2 : __builtin_cpu_supports("ssse3") ? 1 : 0;
    if (level == 2) encode_run_avx2(p, ngroups, codes, valid);
    else if (level == 1) encode_run_ssse3(p, ngroups, codes, valid);
    else for (size_t g = 0; g < ngroups; ++g) encode16_scalar(p + 16 * g, codes + g, valid + g);
}

// Records r0 .. r1 of a CSR batch (each followed by one separator, the part padded with separators to whole groups) -> groups from *codes /
// *valid on; returns the number of groups written.  A record's groups are encoded STRAIGHT from the caller's memory wherever sixteen stream
// bytes lie inside one record (nine of ten groups of a 150 bp read); only the group across a record boundary -- the record's tail, its
// separator, the next record's head -- is assembled in a 16-byte carry.  (Until round 5 every byte went through a 4 KiB line buffer first.)
static size_t pack_records_generic(const unsigned char *bytes, const unsigned long long *offsets, size_t r0, size_t r1, unsigned int *codes, unsigned short *valid) {
    unsigned char carry[16];
    size_t g = 0, fill = 0;
    for (size_t r = r0; r < r1; ++r) {
        const unsigned char *src = bytes + offsets[r];
        size_t n = (size_t)(offsets[r + 1] - offsets[r]);
        if (fill) {   // finish the group the previous record (and its separator) began
            const size_t take = n < 16 - fill ? n : 16 - fill;
            memcpy(carry + fill, src, take);
            fill += take; src += take; n -= take;
            if (fill == 16) { encode_groups(carry, 1, codes + g, valid + g); ++g; fill = 0; }
        }
        const size_t whole = n >> 4;
        if (whole) { encode_groups(src, whole, codes + g, valid + g); g += whole; src += 16 * whole; n -= 16 * whole; }
        if (n) { memcpy(carry + fill, src, n); fill += n; }   // (fill is 0 here unless the record ended inside the carried group)
        carry[fill++] = '\n';
        if (fill == 16) { encode_groups(carry, 1, codes + g, valid + g); ++g; fill = 0; }
    }
    if (fill) { while (fill < 16) carry[fill++] = '\n'; encode_groups(carry, 1, codes + g, valid + g); ++g; }
    return g;
}
// The same with the record loop INSIDE the function that carries the target attribute: the encoders' constants stay in registers from record
// to record and nothing is called (three calls per 150 bp read above, each reloading ten constants); the validity bits come out of the
// movemask in stream order because the bytes are reversed within their 16-byte lane first (one shuffle where a 32-bit bit reversal took a
// dozen scalar operations per 32 bases).
__attribute__((target("avx2"))) static size_t pack_records_avx2(const unsigned char *bytes, const unsigned long long *offsets, size_t r0, size_t r1, unsigned int *codes,
                                                                unsigned short *valid) {
    const __m256i lower = _mm256_set1_epi8(0x20), ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1), m2 = _mm256_set1_epi16(0x0104), m4 = _mm256_set1_epi32(0x00010010);
    const __m256i pick = _mm256_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12);
    const __m256i rev = _mm256_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m128i lower1 = _mm256_castsi256_si128(lower), ca1 = _mm256_castsi256_si128(ca), cc1 = _mm256_castsi256_si128(cc), cg1 = _mm256_castsi256_si128(cg),
                  ct1 = _mm256_castsi256_si128(ct), three1 = _mm256_castsi256_si128(three), one1 = _mm256_castsi256_si128(one), m21 = _mm256_castsi256_si128(m2),
                  m41 = _mm256_castsi256_si128(m4), pick1 = _mm256_castsi256_si128(pick), rev1 = _mm256_castsi256_si128(rev);
    // (macros, not lambdas: a lambda's operator() does not inherit the target attribute)
#define KCT_ENC32(q, c, v)                                                                                                                            \
    do {                                                                                                                                              \
        const __m256i x0_ = _mm256_loadu_si256((const __m256i *)(q)), up_ = _mm256_or_si256(x0_, lower);                                               \
        const __m256i ok_ = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(up_, ca), _mm256_cmpeq_epi8(up_, cc)),                                   \
                                            _mm256_or_si256(_mm256_cmpeq_epi8(up_, cg), _mm256_cmpeq_epi8(up_, ct)));                                  \
        __m256i x_ = _mm256_and_si256(_mm256_srli_epi16(x0_, 1), three);                                                                               \
        x_ = _mm256_xor_si256(x_, _mm256_and_si256(_mm256_srli_epi16(x_, 1), one));                                                                    \
        x_ = _mm256_and_si256(x_, ok_);                                                                                                                \
        const __m256i sh_ = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(x_, m2), m4), pick);                                            \
        (c)[0] = (unsigned int)_mm256_extract_epi32(sh_, 0);                                                                                           \
        (c)[1] = (unsigned int)_mm256_extract_epi32(sh_, 4);                                                                                           \
        const unsigned int mk_ = (unsigned int)_mm256_movemask_epi8(_mm256_shuffle_epi8(ok_, rev)); /* byte j of a lane -> bit 15 - j of its half */   \
        (v)[0] = (unsigned short)mk_;                                                                                                                  \
        (v)[1] = (unsigned short)(mk_ >> 16);                                                                                                          \
    } while (0)
#define KCT_ENC16(q, c, v)                                                                                                                            \
    do {                                                                                                                                              \
        const __m128i x0_ = _mm_loadu_si128((const __m128i *)(q)), up_ = _mm_or_si128(x0_, lower1);                                                    \
        const __m128i ok_ = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(up_, ca1), _mm_cmpeq_epi8(up_, cc1)),                                             \
                                         _mm_or_si128(_mm_cmpeq_epi8(up_, cg1), _mm_cmpeq_epi8(up_, ct1)));                                            \
        __m128i x_ = _mm_and_si128(_mm_srli_epi16(x0_, 1), three1);                                                                                    \
        x_ = _mm_xor_si128(x_, _mm_and_si128(_mm_srli_epi16(x_, 1), one1));                                                                            \
        x_ = _mm_and_si128(x_, ok_);                                                                                                                   \
        const __m128i sh_ = _mm_shuffle_epi8(_mm_madd_epi16(_mm_maddubs_epi16(x_, m21), m41), pick1);                                                  \
        *(c) = (unsigned int)_mm_cvtsi128_si32(sh_);                                                                                                   \
        *(v) = (unsigned short)_mm_movemask_epi8(_mm_shuffle_epi8(ok_, rev1));                                                                         \
    } while (0)
    alignas(16) unsigned char carry[16];
    size_t g = 0, fill = 0;
    for (size_t r = r0; r < r1; ++r) {
        const unsigned char *src = bytes + offsets[r];
        size_t n = (size_t)(offsets[r + 1] - offsets[r]);
        if (fill) {
            const size_t take = n < 16 - fill ? n : 16 - fill;
            memcpy(carry + fill, src, take);
            fill += take; src += take; n -= take;
            if (fill == 16) { KCT_ENC16(carry, codes + g, valid + g); ++g; fill = 0; }
        }
        size_t whole = n >> 4;
        n -= 16 * whole;
        for (; whole >= 2; whole -= 2, src += 32, g += 2) KCT_ENC32(src, codes + g, valid + g);
        if (whole) { KCT_ENC16(src, codes + g, valid + g); ++g; src += 16; }
        if (n) { memcpy(carry + fill, src, n); fill += n; }
        carry[fill++] = '\n';
        if (fill == 16) { KCT_ENC16(carry, codes + g, valid + g); ++g; fill = 0; }
    }
    if (fill) { while (fill < 16) carry[fill++] = '\n'; KCT_ENC16(carry, codes + g, valid + g); ++g; }
    return g;
#undef KCT_ENC32
#undef KCT_ENC16
}
static size_t pack_records(const unsigned char *bytes, const unsigned long long *offsets, size_t r0, size_t r1, unsigned int *codes, unsigned short *valid) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? pack_records_avx2(bytes, offsets, r0, r1, codes, valid) : pack_records_generic(bytes, offsets, r0, r1, codes, valid);
}

// [host-packer-end]

// host bytes -> pinned staging -> device stream buffer (padded with '\n' to a multiple of 16)
kct_status upload_stream(kct_table *t, size_t nbytes) {
    const size_t padded = (nbytes + 15) & ~(size_t)15;
    KCT_TRY(t->d_stream.reserve(padded + 16));
    HIP_TRY(hipMemcpyAsync(t->d_stream.p, t->h_stage.p, padded, hipMemcpyHostToDevice, t->stream));
    return KCT_OK;
}

kct_status stage_single(kct_table *t, const char *seq, size_t len) {
    const size_t padded = (len + 15) & ~(size_t)15;
    KCT_TRY(t->h_stage.reserve(padded + 16));
    memcpy(t->h_stage.p, seq, len);
    memset((char *)t->h_stage.p + len, '\n', padded + 16 - len);
    return upload_stream(t, len);
}

// ---- deferred mode -------------------------------------------------------------------------------------------
constexpr size_t kPendingBytes = (size_t)64 << 20;

// Valid k-windows of one record: the host-side twin of the device's window rule (all k bytes in ACGTacgt).
// Used only for the number deferred consume() returns; the counting itself happens on the device at flush.
u64 host_valid_windows(const unsigned char *s, size_t len, size_t k) {
    // sixteen bytes at a time (SSE2, the x86-64 baseline): (c | 0x20) is one of a, c, g, t exactly for the eight valid bytes
    const __m128i lower = _mm_set1_epi8(0x20), ca = _mm_set1_epi8('a'), cc = _mm_set1_epi8('c'), cg = _mm_set1_epi8('g'), ct = _mm_set1_epi8('t');
    u64 n = 0;
    size_t run = 0, i = 0;  // run: valid bytes ending at the current position
    for (; i + 16 <= len; i += 16) {
        const __m128i v = _mm_or_si128(_mm_loadu_si128((const __m128i *)(s + i)), lower);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, ca), _mm_cmpeq_epi8(v, cc)), _mm_or_si128(_mm_cmpeq_epi8(v, cg), _mm_cmpeq_epi8(v, ct)));
        const unsigned m = (unsigned)_mm_movemask_epi8(ok);
        if (m == 0xFFFFu) {  // windows END at each of the 16 positions whose run has reached k
            run += 16;
            if (run >= k) n += std::min<size_t>(16, run - k + 1);
        } else {
            for (int j = 0; j < 16; ++j) {
                run = (m >> j) & 1u ? run + 1 : 0;
                n += run >= k;
            }
        }
    }
    for (; i < len; ++i) {
        const unsigned char c = s[i] | 0x20;
        run = (c == 'a' || c == 'c' || c == 'g' || c == 't') ? run + 1 : 0;
        n += run >= k;
    }
    return n;
}

kct_status flush_pending(kct_table *t) {
    const size_t used = t->pending_used;
    if (!used) return KCT_OK;
    const u64 records = t->pending_records;
    t->pending_used = 0;  // consume_stream -> ... -> use() must not re-enter
    t->pending_records = 0;
    const size_t padded = (used + 15) & ~(size_t)15;
    memset((char *)t->h_pending.p + used, '\n', padded + 16 - used);
    // Nothing has been counted yet if the upload cannot be made: the records stay buffered and the call can be retried
    // (after kct_release_scratch, say).  Once the device pass has started, a failure leaves the table short of counts that
    // earlier consume() calls have already reported: it is poisoned, and every later call fails until kct_clear.
    kct_status st = t->d_stream.reserve(padded + 16);
    if (st == KCT_OK && hipMemcpyAsync(t->d_stream.p, t->h_pending.p, padded + 16, hipMemcpyHostToDevice, t->stream) != hipSuccess) {
        set_err("hipMemcpyAsync of the buffered records failed");
        st = KCT_ERR_HIP;
    }
    if (st != KCT_OK) { t->pending_used = used; t->pending_records = records; return st; }
    u64 n = 0;
    st = consume_stream(t, (const unsigned char *)t->d_stream.p, used, &n);
    if (st != KCT_OK) t->poisoned = true;
    return st;
}

// what kct_consume_device has staged (its calls' streams behind one another, each in records of its own) -> counted as one stream
kct_status flush_deferred_device(kct_table *t) {
    const size_t used = t->defer_used;
    if (!used) return KCT_OK;
    t->defer_used = 0; t->defer_windows = 0;   // consume_stream -> ... -> use() must not re-enter
    u64 n = 0;
    const kct_status st = consume_stream(t, (const unsigned char *)t->d_defer.p, used, &n);
    if (st != KCT_OK) t->poisoned = true;   // (counts already reported to the caller are missing for good)
    return st;
}

// hashes of all windows of the staged stream [0, nbytes) into d_aux; returns first bad window index
kct_status hash_stream(kct_table *t, u64 nbytes, u64 nwin, u64 *first_bad) {
    KCT_TRY(t->d_aux.reserve(nwin * 8));
    du64 *d_fb = t->d_counters + kNumCounters + 1;  // scratch word 1
    HIP_TRY(hipMemsetAsync(d_fb, 0xFF, 8, t->stream));
    const int grid = (int)((nwin + kct::kTile - 1) / kct::kTile);
    {
        ProfScope ps(t, "hash_windows_kernel");
        dispatch_k<HashLauncher>((int)t->k, t->stream, grid, (const unsigned char *)t->d_stream.p, nbytes, (int)t->k, nwin,
                                 (du64 *)t->d_aux.p, d_fb);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_fb, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *first_bad = t->h_counters[0] == ~0ULL ? nwin : t->h_counters[0];
    return KCT_OK;
}

}  // namespace kcth

namespace kcth {

// A device-resident piece of input that is small for the table is STAGED behind the earlier ones (kct_internal.h defer_device) and counted
// with them: the table then sees ONE large pass -- its two-level paths, its dedupe probe -- where the input arrives in pieces
// (kct_consume_device calls; the 8 / 16 MiB chunks of kct_consume_file, which until round 6 were counted one by one and flipped between
// the policy's paths from chunk to chunk).  *n_total = the piece's good windows (counted on the way by the staging kernel).
// (Not the FIRST piece into an empty table if it brings a window start per slot or more: whoever makes one large call and then reads
// the table should not pay a copy for the calls that might have followed -- 4 % of such a call; a stream of calls of that size has
// its first one counted by itself and the rest together.)
du64 *staged_good_word(kct_table *t);
// The staging itself.  `wait`: the caller wants the piece's good windows now (*n_total) and its buffer back -- the stream is waited for; otherwise
// they are ADDED to the device word staged_good_word(t) and nothing is waited for (kct_consume_file's worker: it zeroes the word before its first
// piece and reads it after its last).  *staged = false: the table does not take this piece that way now (see above; or, without `wait`, staging it
// would first need the earlier pieces counted) -- nothing was done.
static kct_status try_stage(kct_table *t, const unsigned char *d_stream, size_t nbytes, bool wait, u64 *n_total, bool *staged) {
    *staged = false;
    const u64 npos = nbytes >= (u64)t->k ? (u64)nbytes - t->k + 1 : 0;
    const bool untouched = t->n_keys == 0 && !t->defer_used && !t->shadow_dirty && !t->s32_dirty && !t->s128_dirty;
    // Nor a call that by itself fills the passes HBM has room for: a pass is bounded by its scratch (~12.5 B per window start on the 64-bit
    // two-level path), so beside a table that takes most of the GPU -- whole C5's 128 GiB one -- gathering calls cannot make passes larger,
    // and the copy's buffer would only take room from the scratch (every pass re-reads and re-writes the whole table: fewer passes matter).
    bool fills_a_pass = false;
    // (hipMemGetInfo asks the driver -- tens of microseconds, as long as a 4 MiB chunk's copy: the file reader's worker, which stages a chunk
    // every ~0.1 ms and whose pieces alone change the figure, goes by an answer up to 20 ms old)
    size_t free_b = t->mem_free_seen, total_b = 0;
    bool have_mem = !wait && t->mem_free_seen && now_ms() - t->mem_free_seen_ms < 20.0;
    if (!have_mem) {
        have_mem = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        t->mem_free_seen = have_mem ? free_b : 0;
        t->mem_free_seen_ms = now_ms();
    }
    if (have_mem && (t->cap >> t->block_bits) > 1024) {
        const double avail = (double)free_b + (double)(t->d_scratch.cap + t->d_scratch2.cap + t->d_irr.cap + t->d_irr2.cap + t->d_defer.cap);
        fills_a_pass = (double)npos * 12.5 >= 0.5 * 0.8 * avail;
    }
    if (fills_a_pass && t->defer_used) { if (!wait) return KCT_OK; KCT_TRY(flush_deferred_device(t)); }
    if (t->defer_device && t->k <= 255 && npos && npos < 4 * t->cap && !(untouched && npos >= t->cap) && !fills_a_pass) {
        const u64 padded = (((u64)nbytes + 15) & ~15ULL) + 16;   // (the stream, separators up to a 16-byte boundary, one unit of separators)
        u64 limit = 32ULL << 30;
        if (have_mem) limit = std::min<u64>(limit, ((u64)free_b + t->d_defer.cap) / 4);
        if (t->defer_used + padded > limit && t->defer_used) { if (!wait) return KCT_OK; KCT_TRY(flush_deferred_device(t)); }
        if (!wait && t->defer_windows + npos >= 32 * t->cap) return KCT_OK;   // (the pieces are due to be counted after this one: the waiting form does that)
        if (padded <= limit) {
            KCT_TRY(t->d_defer.reserve_keep(t->defer_used + padded, t->defer_used, t->stream));
            du64 *d_good = staged_good_word(t);
            if (wait) HIP_TRY(hipMemsetAsync(d_good, 0, 8, t->stream));
            {
                ProfScope ps(t, "stage_stream_kernel");
                const u64 tiles = (padded + kct::kPartTile - 1) / kct::kPartTile;
                hipLaunchKernelGGL(kct::stage_stream_kernel, dim3((unsigned)std::min<u64>(tiles, 4 * (u64)t->num_cus)), dim3(kct::kPartThreads), 0, t->stream,
                                   d_stream, (u64)nbytes, (int)t->k, (unsigned char *)t->d_defer.p + t->defer_used, padded, d_good);
            }
            HIP_TRY(hipGetLastError());
            if (wait) {
                HIP_TRY(hipMemcpyAsync(t->h_counters, d_good, 8, hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));   // (the caller may reuse its buffer when this returns; n is wanted now)
                *n_total = t->h_counters[0];
            }
            t->defer_used += padded;
            t->defer_windows += npos;
            *staged = true;
            if (wait && t->defer_windows >= 32 * t->cap) KCT_TRY(flush_deferred_device(t));
            return KCT_OK;
        }
    }
    return KCT_OK;
}

du64 *staged_good_word(kct_table *t) { return t->d_counters + kNumCounters + 2; }   // scratch word 2

kct_status consume_device_staged(kct_table *t, const unsigned char *d_stream, size_t nbytes, u64 *n_total) {
    bool staged = false;
    KCT_TRY(try_stage(t, d_stream, nbytes, true, n_total, &staged));
    if (staged) return KCT_OK;
    if (t->defer_used) KCT_TRY(flush_deferred_device(t));
    return consume_stream(t, d_stream, nbytes, n_total);
}

kct_status stage_piece_async(kct_table *t, const unsigned char *d_stream, size_t nbytes, bool *staged) {
    u64 unused = 0;
    return try_stage(t, d_stream, nbytes, false, &unused, staged);
}

}  // namespace kcth

using namespace kcth;

extern "C" {

kct_status kct_hash_windows(kct_table *t, const char *seq, size_t len, uint64_t *hashes_out, size_t cap, uint64_t *n_windows,
                            uint64_t *first_bad) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if ((!seq && len) || !n_windows || !first_bad) { set_err("null argument"); return KCT_ERR_ARG; }
    const u64 nwin = len >= t->k ? len - t->k + 1 : 0;
    *n_windows = nwin;
    *first_bad = nwin;
    if (nwin == 0) return KCT_OK;
    KCT_TRY(stage_single(t, seq, len));
    KCT_TRY(hash_stream(t, len, nwin, first_bad));
    const size_t ncopy = std::min<size_t>(cap, nwin);
    if (ncopy && hashes_out) HIP_TRY(hipMemcpy(hashes_out, t->d_aux.p, ncopy * 8, hipMemcpyDeviceToHost));
    return KCT_OK;
}

kct_status kct_hash_kmer(kct_table *t, const char *kmer, size_t len, uint64_t *hash_out) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if (!kmer || !hash_out) { set_err("null argument"); return KCT_ERR_ARG; }
    if ((uint8_t)len != t->k) { set_err("wrong ksize"); return KCT_ERR_WRONG_KSIZE; }  // lib.rs:66 `len as u8`
    u64 nwin, fb, h = 0;
    KCT_TRY(kct_hash_windows(t, kmer, t->k, &h, 1, &nwin, &fb));  // first window only (lib.rs:78 `.next()`)
    if (fb == 0) { set_err("invalid DNA character in k-mer"); return KCT_ERR_INVALID_DNA; }
    *hash_out = h;
    return KCT_OK;
}

kct_status kct_count(kct_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if ((uint8_t)len != t->k) { set_err("kmer size does not match count table ksize"); return KCT_ERR_WRONG_KSIZE; }
    u64 h;
    KCT_TRY(kct_hash_kmer(t, kmer, len, &h));
    u64 c = 0;
    KCT_TRY(point_add(t, h, &c));
    t->consumed += len;  // lib.rs:153
    if (count_out) *count_out = c;
    return KCT_OK;
}

kct_status kct_get(kct_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if ((uint8_t)len != t->k) { set_err("kmer size does not match count table ksize"); return KCT_ERR_WRONG_KSIZE; }
    u64 h;
    KCT_TRY(kct_hash_kmer(t, kmer, len, &h));
    return kct_get_hash(t, h, count_out);
}

int kct_consume_will_defer(const kct_table *t, size_t len, int skip_bad) {
    return t && !t->poisoned && t->deferred && skip_bad && len + 64 < kPendingBytes / 2 && (len < t->k || t->pending_used + len + 1 + 64 <= t->h_pending.cap) ? 1 : 0;
}

kct_status kct_consume(kct_table *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out) {
    KCT_BORROW(t);
    if (t && t->deferred && !t->poisoned && skip_bad && len + 64 < kPendingBytes / 2) {   // (a poisoned table: use_consume below reports it)
        // deferred mode: buffer the record, answer from the host-side validity scan, count later
        if ((!seq && len) || !n_out) { set_err("null argument"); return KCT_ERR_ARG; }
        *n_out = 0;
        if (len >= t->k) {
            if (t->pending_used + len + 1 + 64 > t->h_pending.cap) {  // (the common call touches no HIP API at all)
                KCT_TRY(use_device(t));
                // the pinned buffer grows geometrically from 1 MiB to its full 64 MiB: a table that sees a few records pins little
                if (t->pending_used + len + 1 > kPendingBytes) KCT_TRY(flush_pending(t));
                size_t want = std::max<size_t>(t->h_pending.cap, (size_t)1 << 20);
                while (want < kPendingBytes + 64 && t->pending_used + len + 1 + 64 > want) want *= 2;
                KCT_TRY(t->h_pending.reserve_keep(std::min(want, kPendingBytes + 64), t->pending_used));
            }
            char *dst = (char *)t->h_pending.p + t->pending_used;
            memcpy(dst, seq, len);
            dst[len] = '\n';
            t->pending_used += len + 1;
            t->pending_records += 1;
            *n_out = host_valid_windows((const unsigned char *)seq, len, t->k);
        }
        t->consumed += len;
        return KCT_OK;
    }
    KCT_TRY(use_consume(t));
    if ((!seq && len) || !n_out) { set_err("null argument"); return KCT_ERR_ARG; }
    *n_out = 0;
    const u64 k = t->k;
    if (len < k) { t->consumed += len; return KCT_OK; }  // zero windows (lib.rs: max_index = 0), consumed still grows
    KCT_TRY(stage_single(t, seq, len));
    u64 use_bytes = len;
    bool bad = false;
    if (!skip_bad) {
        const u64 nwin = len - k + 1;
        u64 fb;
        KCT_TRY(hash_stream(t, len, nwin, &fb));  // validity of every window, on the device
        if (fb < nwin) { bad = true; use_bytes = fb + k - 1; }  // windows 0..fb-1 end before byte fb+k-1
    }
    KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, use_bytes, n_out));
    if (bad) { set_err("bad k-mer encountered at position %llu", (unsigned long long)*n_out); return KCT_ERR_BAD_KMER; }
    t->consumed += len;
    return KCT_OK;
}

kct_status kct_consume_batch(kct_table *t, const char *bytes, const uint64_t *offsets, size_t nrec, int skip_bad,
                             uint64_t *n_total, uint64_t *bad_record, uint64_t *bad_position) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if (!n_total || (nrec && !offsets)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (nrec && !bytes && offsets[nrec] != offsets[0]) { set_err("null argument"); return KCT_ERR_ARG; }  // all-empty records need no bytes
    *n_total = 0;
    if (bad_record) *bad_record = nrec;
    if (bad_position) *bad_position = 0;
    for (double &v : t->batch_tl) v = 0;
    if (nrec == 0) return KCT_OK;
    const double tl0 = now_ms();
    struct rusage ru0;
    getrusage(RUSAGE_SELF, &ru0);
    const u64 total = offsets[nrec] - offsets[0];
    const u64 stream_len = total + nrec;  // one '\n' after every record
    const size_t padded = (stream_len + 15) & ~(size_t)15;
    const size_t off_bytes = skip_bad ? 0 : (nrec + 1) * 8;
    KCT_DBG(t, "batch: %zu records, %llu bytes\n", nrec, (unsigned long long)stream_len);
    KCT_TRY(t->h_stage.reserve(padded + 16 + off_bytes));
    char *dst = (char *)t->h_stage.p;
    u64 *rec_off = (u64 *)(dst + padded + 16);  // 16-aligned since padded is
    // Pack the records into the record stream: record r lands at (offsets[r] - offsets[0]) + r, one
    // separator behind it.  Positions are known up front, so large batches are packed by several threads.
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t max_threads = (size_t)t->tune.pack_threads;
    // (pack_threads PER NUMA NODE the pool's workers are bound to: a batch that lies on one node is packed by that node's workers)
    const size_t pool_nodes = (size_t)WorkerPool::instance().nodes_hint();
    const size_t nthreads = stream_len >= (8u << 20) ? std::min<size_t>({max_threads * pool_nodes, hw ? hw : 1, nrec}) : 1;
    {
        // (8 MB of offsets for a million records: 0.2 ms on one thread, cold -- a twentieth of the call; the pool does it in ~20 us)
        std::atomic<int> bad{0};
        auto check = [&](size_t r0, size_t r1) { for (size_t r = r0; r < r1; ++r) if (offsets[r + 1] < offsets[r]) { bad.store(1, std::memory_order_relaxed); break; } };
        if (nthreads > 1) {
            WorkerPool &pool = WorkerPool::instance();
            pool.start(nthreads, [&](size_t tid) { check(nrec * tid / nthreads, nrec * (tid + 1) / nthreads); });
            pool.wait();
        } else check(0, nrec);
        if (bad.load()) { set_err("offsets must be non-decreasing"); return KCT_ERR_ARG; }
    }
    const u64 base0 = offsets[0];
    const double tl_checked = now_ms() - tl0;
    auto pack_range = [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const u64 n = offsets[r + 1] - offsets[r], w = (offsets[r] - base0) + r;
            if (!skip_bad) rec_off[r] = w;
            memcpy(dst + w, bytes + offsets[r], n);
            dst[w + n] = '\n';
        }
    };
    KCT_TRY(t->d_stream.reserve(padded + 16));
    if (skip_bad && t->packed_upload && t->k <= 64 && nthreads > 1) {
        // PACKED upload: every part of the batch starts on a 16-base boundary of the stream (extra separator bytes in front of
        // it -- any number of invalid bases may sit between two records), so the packers can encode their parts independently:
        // each streams its records (+ one separator each) through a small buffer and writes one code word + one validity word
        // per 16 bases -- 0.375 B per base cross the PCIe link instead of 1.
        const size_t nslices = std::min<size_t>(nrec, std::max<size_t>(4, std::min<size_t>(16, stream_len >> 22)));
        const size_t parts = std::min<size_t>(nthreads, std::max<size_t>(1, nrec / nslices));
        const size_t nitems = nslices * parts;
        std::vector<size_t> cut(nitems + 1);
        for (size_t i = 0; i <= nitems; ++i) {
            const u64 lo = base0 + total * i / nitems;
            cut[i] = i == nitems ? nrec : (size_t)(std::lower_bound(offsets, offsets + nrec, lo) - offsets);
        }
        cut[0] = 0;
        std::vector<u64> pos(nitems + 1);
        pos[0] = 0;
        for (size_t i = 0; i < nitems; ++i) pos[i + 1] = (pos[i] + (offsets[cut[i + 1]] - offsets[cut[i]]) + (cut[i + 1] - cut[i]) + 15) & ~(u64)15;
        const u64 nbases = pos[nitems], ng = nbases >> 4;
        const u64 valid_off = (ng * 4 + 255) & ~(u64)255;
        KCT_TRY(t->h_stage.reserve(valid_off + ng * 2 + 64));
        KCT_TRY(t->d_stream.reserve(valid_off + ng * 2 + 64));
        unsigned int *h_codes = (unsigned int *)t->h_stage.p;
        unsigned short *h_valid = (unsigned short *)((char *)t->h_stage.p + valid_off);
        std::vector<std::atomic<int>> packed(nslices);
        for (auto &r : packed) r.store(0, std::memory_order_relaxed);
        // KCT_PACK_PIN=1 (measurement switch, kct_core.hip): the parts are listed per NUMA node of their source (move_pages in query
        // mode, one page per part) and a worker takes from its own node's list first.  Off (one node, one list) by default.
        const size_t nn = std::max<size_t>(1, pool_nodes);
        std::vector<std::vector<unsigned int>> lists(nn);
        {
            std::vector<int> status(nitems, -1);
            if (nn > 1) {
                std::vector<void *> pages(nitems);
                const long psz = sysconf(_SC_PAGESIZE);
                for (size_t i = 0; i < nitems; ++i) pages[i] = (void *)((uintptr_t)(bytes + offsets[std::min(cut[i], nrec - 1)]) & ~(uintptr_t)(psz - 1));
                if (syscall(SYS_move_pages, 0, (unsigned long)nitems, pages.data(), nullptr, status.data(), 0) != 0) std::fill(status.begin(), status.end(), -1);
            }
            for (size_t i = 0; i < nitems; ++i) lists[status[i] >= 0 && (size_t)status[i] < nn ? (size_t)status[i] : i % nn].push_back((unsigned int)i);
        }
        std::vector<std::atomic<size_t>> next_of(nn);
        for (auto &a_ : next_of) a_.store(0, std::memory_order_relaxed);
        // kct_batch_timeline: per packer thread its first start, last end, busy time and the CPU it ran on
        struct ThreadLine { double first = 1e300, last = 0, busy = 0; int cpu = -1; char pad[36]; };
        std::vector<ThreadLine> lines(nthreads);
        const double tl_cut = now_ms() - tl0;
        WorkerPool &pool = WorkerPool::instance();
        pool.start(nthreads, [&](size_t tid) {
            ThreadLine &ln = lines[tid];
            ln.cpu = sched_getcpu();
            // (pack_records: a record's groups are encoded straight from the caller's memory; see the host packer above)
            for (;;) {
                size_t it = nitems;
                for (size_t d = 0; d < nn && it == nitems; ++d) {   // this worker's node first (worker i lives on node i % nn)
                    const size_t nd = (tid + d) % nn;
                    if (next_of[nd].load(std::memory_order_relaxed) >= lists[nd].size()) continue;
                    const size_t ix = next_of[nd].fetch_add(1, std::memory_order_relaxed);
                    if (ix < lists[nd].size()) it = lists[nd][ix];
                }
                if (it >= nitems) break;
                const double it0 = now_ms() - tl0;
                ln.first = std::min(ln.first, it0);
                size_t g = pos[it] >> 4;
                g += pack_records((const unsigned char *)bytes, (const unsigned long long *)offsets, cut[it], cut[it + 1], h_codes + g, h_valid + g);
                for (; g < (pos[it + 1] >> 4); ++g) { h_codes[g] = 0; h_valid[g] = 0; }  // (never: a part's groups are exactly its bytes, padded)
                ln.last = now_ms() - tl0;
                ln.busy += ln.last - it0;
                packed[it / parts].fetch_add(1, std::memory_order_release);
            }
        });
        hipError_t copy_err = hipSuccess;
        char *d_base = (char *)t->d_stream.p;
        for (size_t sl = 0; sl < nslices; ++sl) {
            while (packed[sl].load(std::memory_order_acquire) < (int)parts) std::this_thread::yield();
            const u64 g0 = pos[sl * parts] >> 4, g1 = pos[(sl + 1) * parts] >> 4;
            if (g1 > g0 && copy_err == hipSuccess) copy_err = hipMemcpyAsync(d_base + g0 * 4, h_codes + g0, (g1 - g0) * 4, hipMemcpyHostToDevice, t->stream);
            if (g1 > g0 && copy_err == hipSuccess) copy_err = hipMemcpyAsync(d_base + valid_off + g0 * 2, h_valid + g0, (g1 - g0) * 2, hipMemcpyHostToDevice, t->stream);
        }
        const double tl_h2d = now_ms() - tl0;
        pool.wait();
        HIP_TRY(copy_err);
        KCT_DBG(t, "batch: packed upload of %llu bases enqueued\n", (unsigned long long)nbases);
        KCT_TRY(consume_stream_packed(t, (const unsigned int *)d_base, (const unsigned short *)(d_base + valid_off), nbases, n_total));
        t->consumed += total;
        {
            double *o = t->batch_tl;
            struct rusage ru1;
            getrusage(RUSAGE_SELF, &ru1);
            double first = 1e300, last = 0, busy = 0, busy_max = 0;
            std::vector<int> cpus, nodes;
            for (const ThreadLine &ln : lines) {
                if (ln.last == 0) continue;
                first = std::min(first, ln.first); last = std::max(last, ln.last); busy += ln.busy; busy_max = std::max(busy_max, ln.busy);
                if (std::find(cpus.begin(), cpus.end(), ln.cpu) == cpus.end()) cpus.push_back(ln.cpu);
            }
            for (int c : cpus) {   // the NUMA node of a CPU: the nodeN entry of its sysfs directory
                int node = -1;
                for (int nd = 0; nd < 16 && node < 0; ++nd) {
                    char path[128];
                    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node%d", c, nd);
                    if (access(path, F_OK) == 0) node = nd;
                }
                if (std::find(nodes.begin(), nodes.end(), node) == nodes.end()) nodes.push_back(node);
            }
            o[0] = tl_checked; o[1] = tl_cut; o[2] = first < 1e299 ? first : 0; o[3] = last; o[4] = tl_h2d; o[5] = now_ms() - tl0;
            o[6] = (double)cpus.size() ? (double)std::count_if(lines.begin(), lines.end(), [](const ThreadLine &l) { return l.last != 0; }) : 0; o[7] = busy; o[8] = busy_max;
            o[9] = (double)total; o[10] = (double)(ng * 6); o[11] = (double)(ru1.ru_minflt - ru0.ru_minflt);
            o[12] = (double)cpus.size(); o[13] = (double)nodes.size(); o[14] = (double)sched_getcpu(); o[15] = pool.pinned() ? 1 : 0;
            if (t->debug) {   // where the batch, the staging and the packers were (NUMA nodes): one line
                auto node_of_cpu = [](int c) { for (int nd = 0; nd < 16; ++nd) { char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node%d", c, nd); if (access(path, F_OK) == 0) return nd; } return -1; };
                auto pages_on = [&](const void *base, size_t bytes, int out_[4]) {
                    const long psz = sysconf(_SC_PAGESIZE);
                    void *pg[64]; int stt[64];
                    for (int i = 0; i < 64; ++i) { pg[i] = (void *)(((uintptr_t)base + (bytes / 64) * i) & ~(uintptr_t)(psz - 1)); stt[i] = -1; }
                    for (int i = 0; i < 4; ++i) out_[i] = 0;
                    if (syscall(SYS_move_pages, 0, 64UL, pg, nullptr, stt, 0) == 0) for (int i = 0; i < 64; ++i) ++out_[stt[i] >= 0 && stt[i] < 3 ? stt[i] : 3];
                };
                int src_n[4], stg_n[4];
                pages_on(bytes + offsets[0], (size_t)total, src_n);
                pages_on(h_codes, (size_t)ng * 4, stg_n);
                double busy_n[4] = {0, 0, 0, 0}; int thr_n[4] = {0, 0, 0, 0};
                for (const ThreadLine &ln : lines) if (ln.last != 0) { const int nd = node_of_cpu(ln.cpu); busy_n[nd >= 0 && nd < 3 ? nd : 3] += ln.busy; ++thr_n[nd >= 0 && nd < 3 ? nd : 3]; }
                KCT_DBG(t, "batch: source pages on nodes [%d %d %d ?%d] of 64 samples, staging [%d %d %d ?%d]; packers node0: %d threads busy %.2f ms, node1: %d threads busy %.2f ms; caller cpu %d (node %d)\n",
                        src_n[0], src_n[1], src_n[2], src_n[3], stg_n[0], stg_n[1], stg_n[2], stg_n[3], thr_n[0], busy_n[0], thr_n[1], busy_n[1], sched_getcpu(), node_of_cpu(sched_getcpu()));
            }
        }
        return KCT_OK;
    }
    if (nthreads <= 1) {
        pack_range(0, nrec);
        if (!skip_bad) rec_off[nrec] = stream_len;
        memset(dst + stream_len, '\n', padded + 16 - stream_len);
        HIP_TRY(hipMemcpyAsync(t->d_stream.p, dst, padded, hipMemcpyHostToDevice, t->stream));
    } else {
        // The stream is cut into slices (by bytes: records may be ragged) and every slice into parts; the packers take the
        // parts in order, so ALL of them work on the slice that is uploaded next, and this thread starts a slice's H2D copy
        // as soon as its last part is packed: the copy runs under the packing of the slices behind it and only one slice's
        // copy is left when the packing ends.  (Slices of ~10 MB: every copy costs ~15 us on top of its transfer.)
        const size_t nslices = std::min<size_t>(nrec, std::max<size_t>(4, std::min<size_t>(16, stream_len >> 22)));
        const size_t parts = std::min<size_t>(nthreads, std::max<size_t>(1, nrec / nslices));
        const size_t nitems = nslices * parts;
        std::vector<size_t> cut(nitems + 1);
        for (size_t i = 0; i <= nitems; ++i) {
            const u64 lo = base0 + total * i / nitems;
            cut[i] = i == nitems ? nrec : (size_t)(std::lower_bound(offsets, offsets + nrec, lo) - offsets);
        }
        cut[0] = 0;
        std::vector<std::atomic<int>> packed(nslices);
        for (auto &r : packed) r.store(0, std::memory_order_relaxed);
        std::atomic<size_t> next_item{0};
        WorkerPool &pool = WorkerPool::instance();  // (threads that outlive the call: starting them cost more than the packing)
        pool.start(nthreads, [&](size_t) {
            for (;;) {
                const size_t it = next_item.fetch_add(1, std::memory_order_relaxed);
                if (it >= nitems) break;
                if (cut[it + 1] > cut[it]) pack_range(cut[it], cut[it + 1]);
                packed[it / parts].fetch_add(1, std::memory_order_release);
            }
        });
        hipError_t copy_err = hipSuccess;
        for (size_t sl = 0; sl < nslices; ++sl) {
            while (packed[sl].load(std::memory_order_acquire) < (int)parts) std::this_thread::yield();
            const size_t r0 = cut[sl * parts], r1 = cut[(sl + 1) * parts];
            const u64 b0 = r0 < nrec ? (offsets[r0] - base0) + r0 : stream_len;
            u64 b1 = r1 < nrec ? (offsets[r1] - base0) + r1 : stream_len;
            if (sl + 1 == nslices) {  // the tail slice carries the padding
                if (!skip_bad) rec_off[nrec] = stream_len;
                memset(dst + stream_len, '\n', padded + 16 - stream_len);
                b1 = padded;
            }
            if (b1 > b0 && copy_err == hipSuccess)
                copy_err = hipMemcpyAsync((char *)t->d_stream.p + b0, dst + b0, b1 - b0, hipMemcpyHostToDevice, t->stream);
        }
        KCT_DBG(t, "batch: last slice enqueued\n");
        pool.wait();
        HIP_TRY(copy_err);
        if (t->debug) { HIP_TRY(hipStreamSynchronize(t->stream)); KCT_DBG(t, "batch: upload done\n"); }
    }

    if (!skip_bad) {
        KCT_TRY(t->d_aux.reserve(off_bytes));
        HIP_TRY(hipMemcpyAsync(t->d_aux.p, rec_off, off_bytes, hipMemcpyHostToDevice, t->stream));
        du64 *d_q = t->d_counters + kNumCounters + 1;
        HIP_TRY(hipMemsetAsync(d_q, 0xFF, 8, t->stream));
        const u64 nthreads = (stream_len + 15) / 16;
        {
            ProfScope ps(t, "first_bad_byte_kernel");
            hipLaunchKernelGGL(kct::first_bad_byte_kernel, dim3((unsigned)((nthreads + kct::kBlock - 1) / kct::kBlock)), dim3(kct::kBlock), 0,
                               t->stream, (const unsigned char *)t->d_stream.p, stream_len, (int)t->k, (const du64 *)t->d_aux.p, (u64)nrec, d_q);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_q, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const u64 q = t->h_counters[0];
        if (q != ~0ULL) {
            // Record r holds q.  The per-record loop the reference runs would count records
            // [0, r) whole, then the windows of r before its first bad one, then raise.
            const size_t r = (size_t)(std::upper_bound(rec_off, rec_off + nrec + 1, q) - rec_off) - 1;
            const u64 in_rec = q - rec_off[r];
            const u64 fbw = in_rec + 1 >= t->k ? in_rec + 1 - t->k : 0;  // index of r's first bad window
            u64 n_before = 0, n_prefix = 0;
            KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, rec_off[r], &n_before));
            const u64 prefix = fbw + t->k - 1;  // bytes of r that its windows 0..fbw-1 cover
            if (fbw > 0) {
                // a 16-byte aligned copy of the prefix, in a buffer nothing inside consume_stream touches (its passes
                // reallocate / overwrite d_aux2 and d_spill when they replay spills)
                KCT_TRY(t->d_prefix.reserve(((prefix + 15) & ~(u64)15) + 16));
                HIP_TRY(hipMemcpyAsync(t->d_prefix.p, (const char *)t->d_stream.p + rec_off[r], prefix, hipMemcpyDeviceToDevice, t->stream));
                KCT_TRY(consume_stream(t, (const unsigned char *)t->d_prefix.p, prefix, &n_prefix));
            }
            t->consumed += offsets[r] - offsets[0];  // r raised before lib.rs:604
            *n_total = n_before + n_prefix;
            if (bad_record) *bad_record = r;
            if (bad_position) *bad_position = n_prefix;
            set_err("bad k-mer encountered at position %llu (record %llu)", (unsigned long long)n_prefix, (unsigned long long)r);
            return KCT_ERR_BAD_KMER;
        }
    }
    KCT_TRY(consume_stream(t, (const unsigned char *)t->d_stream.p, stream_len, n_total));
    KCT_DBG(t, "batch: counted\n");
    t->consumed += total;
    return KCT_OK;
}

kct_status kct_consume_device_packed(kct_table *t, const void *d_codes, const void *d_valid, size_t nbases, uint64_t consumed_bytes, uint64_t *n_total) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if (!n_total || ((!d_codes || !d_valid) && nbases)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_codes & 3) != 0 || ((uintptr_t)d_valid & 1) != 0) { set_err("d_codes / d_valid must be 4- / 2-byte aligned"); return KCT_ERR_ARG; }
    KCT_TRY(consume_stream_packed(t, (const unsigned int *)d_codes, (const unsigned short *)d_valid, nbases, n_total));
    t->consumed += consumed_bytes;
    return KCT_OK;
}

kct_status kct_pack_stream_device(const void *d_stream, size_t nbytes, void *d_codes, void *d_valid, void *stream) {
    const u64 ng = ((u64)nbytes + 15) >> 4;
    if (!ng) return KCT_OK;
    if (!d_stream || !d_codes || !d_valid || ((uintptr_t)d_stream & 15)) { set_err("null or misaligned argument"); return KCT_ERR_ARG; }
    hipLaunchKernelGGL(kct::pack_stream_kernel, dim3((unsigned)std::min<u64>((ng + kct::kBlock - 1) / kct::kBlock, 1u << 16)), dim3(kct::kBlock), 0, (hipStream_t)stream,
                       (const unsigned char *)d_stream, (u64)nbytes, (unsigned int *)d_codes, (unsigned short *)d_valid, ng);
    HIP_TRY(hipGetLastError());
    return KCT_OK;
}

kct_status kct_set_packed_upload(kct_table *t, int on) {
    KCT_BORROW(t);
    if (!t) { set_err("null table handle"); return KCT_ERR_ARG; }
    t->packed_upload = on != 0;
    return KCT_OK;
}

kct_status kct_consume_device(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes, uint64_t *n_total) {
    KCT_BORROW(t);
    KCT_TRY(use_device(t));
    if (t->pending_used) KCT_TRY(flush_pending(t));
    if (!n_total || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    KCT_TRY(consume_device_staged(t, (const unsigned char *)d_stream, nbytes, n_total));
    t->consumed += consumed_bytes;
    return KCT_OK;
}

}  // extern "C"

extern "C" kct_status kct_batch_timeline(kct_table *t, double *out16) {
    KCT_BORROW(t);
    if (!out16) { set_err("null argument"); return KCT_ERR_ARG; }
    for (int i = 0; i < 16; ++i) out16[i] = t->batch_tl[i];
    return KCT_OK;
}
