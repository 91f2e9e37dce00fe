// kmer_device.h -- device-side k-mer arithmetic for gfx950 (wave64).
//
// What is computed (bit-exact with the reference path):
//   window good  <=>  all k bytes are A/C/G/T after ASCII upper-casing   (sourmash VALID)
//   canonical    =   bytewise min(forward, reverse complement)           (sourmash SeqToHashes)
//   hash         =   MurmurHash3_x64_128(canonical ASCII bytes, seed 42).h1
//                                                                        (lib.rs:69-76, 576-584)
// How: bases are held 2 bits each, A<C<G<T = 0<1<2<3, first base in the most significant
// position, so lexicographic min of the byte strings == integer min of the packed words.  The
// chosen strand is re-expanded to upper-case ASCII in registers (v_bfrev + two shift/mask
// steps + one v_perm_b32 byte-LUT per four bases) because the hash input is the ASCII text.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kct {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 kSeed = 42;  // lib.rs:75, 582
constexpr u64 kC1 = 0x87c37b91114253d5ULL;
constexpr u64 kC2 = 0x4cf5ad432745937fULL;

// rotl of a 64-bit value as two v_alignbit_b32 (r is a compile-time constant at every call site: MurmurHash3's 27, 31, 33); the
// compiler's own expansion is two 64-bit shifts and two ors (K1 hashing -3.4 %)
__device__ __forceinline__ u64 rotl64(u64 x, int r) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 nlo, nhi;
    if (r < 32) { nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - r); nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - r); }
    else if (r == 32) { nhi = lo; nlo = hi; }
    else { nhi = __builtin_amdgcn_alignbit(lo, hi, 64 - r); nlo = __builtin_amdgcn_alignbit(hi, lo, 64 - r); }
    return ((u64)nhi << 32) | nlo;
}

__device__ __forceinline__ u64 fmix64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

struct Murmur {
    u64 h1, h2;
    __device__ __forceinline__ Murmur() : h1(kSeed), h2(kSeed) {}
    __device__ __forceinline__ void block(u64 k1, u64 k2) {
        k1 *= kC1; k1 = rotl64(k1, 31); k1 *= kC2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= kC2; k2 = rotl64(k2, 33); k2 *= kC1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    // the same with k1 * c1 and k2 * c2 already formed (pre-multiplied tables)
    __device__ __forceinline__ void block_premul(u64 k1c1, u64 k2c2) {
        k1c1 = rotl64(k1c1, 31); k1c1 *= kC2; h1 ^= k1c1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2c2 = rotl64(k2c2, 33); k2c2 *= kC1; h2 ^= k2c2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    // rem = number of tail bytes (1..15); k1/k2 already hold only those bytes (rest zero)
    __device__ __forceinline__ void tail(u64 k1, u64 k2, int rem) {
        if (rem > 8) { k2 *= kC2; k2 = rotl64(k2, 33); k2 *= kC1; h2 ^= k2; }
        k1 *= kC1; k1 = rotl64(k1, 31); k1 *= kC2; h1 ^= k1;
    }
    // the tail with k1 * c1 (and, for rem > 8, k2 * c2) already formed (pre-multiplied tables)
    __device__ __forceinline__ void tail_premul(u64 k1c1, u64 k2c2, int rem) {
        if (rem > 8) { k2c2 = rotl64(k2c2, 33); k2c2 *= kC1; h2 ^= k2c2; }
        k1c1 = rotl64(k1c1, 31); k1c1 *= kC2; h1 ^= k1c1;
    }
    __device__ __forceinline__ u64 finish(u64 len) {
        h1 ^= len; h2 ^= len;
        h1 += h2; h2 += h1;
        h1 = fmix64(h1); h2 = fmix64(h2);
        return h1 + h2;
    }
};

// low `n` bytes kept (n in 0..8), n wave-uniform
__device__ __forceinline__ u64 keep_bytes(u64 v, int n) { return n >= 8 ? v : (v & ((1ULL << (8 * n)) - 1ULL)); }

// ---- base classification --------------------------------------------------------------------
// returns 0..3 for A/C/G/T (either case), 4 otherwise
__device__ __forceinline__ u32 base_code(u32 c) {
    const u32 letters = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20);  // a c g t, as (c & 31)
    bool ok = ((c & 0xC0u) == 0x40u) && ((letters >> (c & 31u)) & 1u);
    u32 x = (c >> 1) & 3u;  // A0 C1 G3 T2
    x ^= x >> 1;            // A0 C1 G2 T3
    return ok ? x : 4u;
}

// four 2-bit codes (bit-swapped within each pair by v_bfrev) -> four ASCII bytes
__device__ __forceinline__ u32 expand4(u32 v8) {
    u32 s = (v8 | (v8 << 12)) & 0x000F000Fu;
    s = (s | (s << 6)) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, 0x54434741u /* 'A','G','C','T' for swapped codes 0,1,2,3 */, s);
}

// 16 bases held in one dword, first base in bits 31:30 -> 16 ASCII bytes as two LE u64
__device__ __forceinline__ void expand16(u32 chunk, u64 &lo, u64 &hi) {
    u32 r = __builtin_bitreverse32(chunk);  // base j now in bits 2j+1:2j, its two bits swapped
    u32 d0 = expand4(r & 0xFFu), d1 = expand4((r >> 8) & 0xFFu);
    u32 d2 = expand4((r >> 16) & 0xFFu), d3 = expand4(r >> 24);
    lo = (u64)d0 | ((u64)d1 << 32);
    hi = (u64)d2 | ((u64)d3 << 32);
}

// The same through a 256-entry table in LDS (one entry per byte of four packed bases, first base in bits 7:6):
// four ds_read_b32 instead of ~17 VALU instructions.  For kernels that are VALU-issue bound and have LDS to spare.
__device__ __forceinline__ void fill_ascii4_lut(u32 *lut, int tid, int nthreads) {
    for (int b = tid; b < 256; b += nthreads) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v |= ((0x54474341u >> (8 * ((b >> (6 - 2 * j)) & 3))) & 0xFFu) << (8 * j);  // "ACGT"[code]
        lut[b] = v;
    }
}

__device__ __forceinline__ void expand16_lut(const u32 *lut, u32 chunk, u64 &lo, u64 &hi) {
    const u32 d0 = lut[chunk >> 24], d1 = lut[(chunk >> 16) & 0xFFu], d2 = lut[(chunk >> 8) & 0xFFu], d3 = lut[chunk & 0xFFu];
    lo = (u64)d0 | ((u64)d1 << 32);
    hi = (u64)d2 | ((u64)d3 << 32);
}

// ---- packed k-mers of KW 64-bit words (k <= 32*KW); w[0] is most significant ------------------
template <int KW>
struct Packed {
    u64 w[KW];
};

template <int KW>
__device__ __forceinline__ void push_fw(Packed<KW> &a, u32 code) {  // a = (a << 2) | code
#pragma unroll
    for (int i = 0; i < KW - 1; ++i) a.w[i] = (a.w[i] << 2) | (a.w[i + 1] >> 62);
    a.w[KW - 1] = (a.w[KW - 1] << 2) | code;
}

template <int KW>
__device__ __forceinline__ void push_rc(Packed<KW> &a, u32 ccode, int k) {  // a = (a >> 2) | ccode << (2k-2)
#pragma unroll
    for (int i = KW - 1; i > 0; --i) a.w[i] = (a.w[i] >> 2) | (a.w[i - 1] << 62);
    a.w[0] >>= 2;
    const int pos = 2 * k - 2;  // bit position counted from the LSB of w[KW-1]
#pragma unroll
    for (int i = 0; i < KW; ++i) {
        const int base = 64 * (KW - 1 - i);
        if (pos >= base && pos < base + 64) a.w[i] |= (u64)ccode << (pos - base);
    }
}

template <int KW>
__device__ __forceinline__ void mask_k(Packed<KW> &a, int k) {  // keep the low 2k bits
#pragma unroll
    for (int i = 0; i < KW; ++i) {
        const int base = 64 * (KW - 1 - i);
        const int bits = 2 * k - base;
        if (bits <= 0) a.w[i] = 0;
        else if (bits < 64) a.w[i] &= (1ULL << bits) - 1ULL;
    }
}

template <int KW>
__device__ __forceinline__ bool less_eq(const Packed<KW> &a, const Packed<KW> &b) {
    if (KW == 1) return a.w[0] <= b.w[0];
    bool le = a.w[KW - 1] <= b.w[KW - 1];
#pragma unroll
    for (int i = KW - 2; i >= 0; --i) le = a.w[i] < b.w[i] || (a.w[i] == b.w[i] && le);
    return le;
}

// shift left so that base 0 sits in bits 63:62 of w[0]
template <int KW>
__device__ __forceinline__ void left_align(Packed<KW> &a, int k) {
    const int s = 64 * KW - 2 * k;  // wave-uniform
    const int ws = s >> 6, bs = s & 63;
    Packed<KW> o;
#pragma unroll
    for (int i = 0; i < KW; ++i) {
        u64 hi = 0, lo = 0;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            if (j == i + ws) hi = a.w[j];
            if (j == i + ws + 1) lo = a.w[j];
        }
        o.w[i] = bs ? ((hi << bs) | (lo >> (64 - bs))) : hi;
    }
    a = o;
}

// Pre-multiplied tables (K1's hashing mode): MurmurHash3 starts every 8-byte word with k *= c1 (first word of a block,
// and of the tail) or k *= c2 (second word).  A word is two table look-ups d0 | d1 << 32, so
//     k * c  =  d0 * c  +  (d1 * c) << 32          (mod 2^64)
// and with mul[b] = ascii4(b) * c held as u64 the product is one 8-byte read, one 4-byte read (only the low half of the
// second product survives the shift) and ONE 32-bit add -- instead of a 64 x 64 multiply (v_mad_u64_u32 + 2 v_mul_lo_u32 +
// adds).  2 x 256 x 8 B of LDS.
__device__ __forceinline__ void fill_premul_luts(u64 *mul1, u64 *mul2, int tid, int nthreads) {
    for (int b = tid; b < 256; b += nthreads) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v |= ((0x54474341u >> (8 * ((b >> (6 - 2 * j)) & 3))) & 0xFFu) << (8 * j);  // "ACGT"[code]
        mul1[b] = (u64)v * kC1;
        mul2[b] = (u64)v * kC2;
    }
}
// The TAIL's first multiply the same way (compile-time k only).  Of the tail's rem = k mod 16 bytes, one 4-byte piece is cut short:
// m = ((rem - 1) & 3) + 1 of its bytes count (the rest are zero bytes, not 'A').  tmul[b] = (ascii4(b) masked to m bytes) * c, with
// c = c1 if that piece belongs to k1 (rem <= 8), c2 if to k2.  m == 4 needs no table (the piece is whole: mul1 / mul2 serve).
__device__ __forceinline__ constexpr int tail_piece_bytes(int k) { return (((k & 15) - 1) & 3) + 1; }
__device__ __forceinline__ constexpr bool tail_needs_lut(int k) { return (k & 15) != 0 && tail_piece_bytes(k) != 4; }
__device__ __forceinline__ void fill_tail_lut(u64 *tmul, int tid, int nthreads, int k) {
    const int m = tail_piece_bytes(k);
    const u64 c = (k & 15) > 8 ? kC2 : kC1;
    const u32 mask = m >= 4 ? 0xFFFFFFFFu : (1u << (8 * m)) - 1u;
    for (int b = tid; b < 256; b += nthreads) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v |= ((0x54474341u >> (8 * ((b >> (6 - 2 * j)) & 3))) & 0xFFu) << (8 * j);  // "ACGT"[code]
        tmul[b] = (u64)(v & mask) * c;
    }
}
__device__ __forceinline__ u64 premul_word(const u64 *mul, u32 b_lo, u32 b_hi) {  // (ascii4(b_lo) | ascii4(b_hi) << 32) * c
    const u64 lo = mul[b_lo];
    const u32 hi = reinterpret_cast<const u32 *>(mul)[2 * b_hi];  // low half of the second product
    return lo + ((u64)hi << 32);
}

// MurmurHash3_x64_128(seed 42).h1 of the ASCII text of a left-aligned packed k-mer.
// One 32-bit chunk of the packed form = 16 bases = exactly one 16-byte murmur block.
// PRE (compile time, so that no dead path stays in the kernel): bit 0 = whole blocks through the pre-multiplied tables mul1 / mul2,
// bit 1 = the tail too (tmul where the tail's last piece is cut short).
template <int KW, bool LUT = false, int PRE = 0>
__device__ __forceinline__ u64 hash_packed(const Packed<KW> &a, int k, const u32 *lut = nullptr, const u64 *mul1 = nullptr, const u64 *mul2 = nullptr,
                                           const u64 *tmul = nullptr) {
    Murmur m;
    const int nblocks = k >> 4, rem = k & 15;
#ifdef KCT_DEBUG_ZERO_KMER  // `make zero` (tests only): one chosen k-mer hashes to 0, the value consume skips (lib.rs:589)
    if (k == KCT_DEBUG_ZERO_K && a.w[0] == KCT_DEBUG_ZERO_KMER) return 0;
#endif
#pragma unroll
    for (int b = 0; b < 2 * KW; ++b) {
        if (b * 16 < k) {
            u32 chunk = (b & 1) ? (u32)a.w[b >> 1] : (u32)(a.w[b >> 1] >> 32);
            if constexpr (LUT && (PRE & 1) != 0) {
                if (b < nblocks) {  // a whole block: both words through the pre-multiplied tables
                    m.block_premul(premul_word(mul1, chunk >> 24, (chunk >> 16) & 0xFFu), premul_word(mul2, (chunk >> 8) & 0xFFu, chunk & 0xFFu));
                    continue;
                }
                if ((PRE & 2) != 0 && b == nblocks && rem != 0) {  // the tail's first multiplies likewise (k at compile time)
                    const u32 B0 = chunk >> 24, B1 = (chunk >> 16) & 0xFFu, B2 = (chunk >> 8) & 0xFFu, B3 = chunk & 0xFFu;
                    const int mp = tail_piece_bytes(k);
                    u64 k1c1, k2c2 = 0;
                    if (rem >= 8) k1c1 = premul_word(mul1, B0, B1);
                    else if (rem > 4) k1c1 = mul1[B0] + ((u64)(u32)(mp == 4 ? mul1[B1] : tmul[B1]) << 32);
                    else k1c1 = mp == 4 ? mul1[B0] : tmul[B0];
                    if (rem > 12) k2c2 = mul2[B2] + ((u64)(u32)(mp == 4 ? mul2[B3] : tmul[B3]) << 32);
                    else if (rem > 8) k2c2 = mp == 4 ? mul2[B2] : tmul[B2];
                    m.tail_premul(k1c1, k2c2, rem);
                    continue;
                }
            }
            u64 k1, k2;
            if constexpr (LUT) expand16_lut(lut, chunk, k1, k2);
            else expand16(chunk, k1, k2);
            if (b < nblocks) m.block(k1, k2);
            else m.tail(keep_bytes(k1, rem), keep_bytes(k2, rem - 8), rem);
        }
    }
    return m.finish((u64)k);
}

// ---- a cheap bijection of u64 for the dedupe-first path ------------------------------------------------
// mix64 scrambles a packed canonical k-mer (+1, so never 0) well enough to serve as a partition / slot / fingerprint
// key in place of the MurmurHash3 value; unmix64 recovers the k-mer once its occurrences have been counted.
// Every step is invertible (xor-shift by 32 is its own inverse; the constant is odd); 0 maps to 0.  One multiply
// between two xor-shifts spreads bins, home groups and fingerprints as evenly as two rounds did (5 Mbp of random sequence).
constexpr u64 kMixA = 0xbf58476d1ce4e5b9ULL;     // splitmix64's first multiplier
constexpr u64 kMixAInv = 0x96de1b173f119089ULL;  // its inverse mod 2^64
__device__ __host__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 32; x *= kMixA;
    x ^= x >> 32;
    return x;
}
__device__ __host__ __forceinline__ u64 unmix64(u64 x) {
    x ^= x >> 32; x *= kMixAInv;
    x ^= x >> 32;
    return x;
}

// Two words (33 <= k <= 64: the packed k-mer is w0 = high word, w1 = low word): two Feistel rounds with mix64 as the round function --
// a bijection of 128 bits whatever mix64 is.  x (the word every partition / slot / bin decision looks at) depends non-linearly on
// every input bit; y is the companion word that makes the pair identify the k-mer.
__device__ __host__ __forceinline__ void mix128(u64 w0, u64 w1, u64 &x, u64 &y) {
    y = w0 ^ mix64(w1 + 0x9e3779b97f4a7c15ULL);
    x = w1 ^ mix64(y + 0x3c6ef372fe94f82aULL);
}
__device__ __host__ __forceinline__ void unmix128(u64 x, u64 y, u64 &w0, u64 &w1) {
    w1 = x ^ mix64(y + 0x3c6ef372fe94f82aULL);
    w0 = y ^ mix64(w1 + 0x9e3779b97f4a7c15ULL);
}

// The same on 42 bits (k <= 21: a packed k-mer is at most 42 bits): the compact dedupe-first path splits the result
// into a 10-bit bin and a 32-bit entry.  A multiplication mod 2^42 by an odd constant and the xor-shift by 21 (its own
// inverse on 42 bits) are bijections of [0, 2^42).
constexpr u64 kMask42 = (1ULL << 42) - 1;
// One multiply and one xor-shift are enough here (bin = the product's top bits, which depend on every input bit; slot
// and fingerprint = its low bits xor its bits 21..41): the spread over bins and home groups measured on 5 Mbp of random
// sequence is the same as with two rounds, and K1 is issue-bound.
__device__ __host__ __forceinline__ u64 mix42(u64 x) {
    x = (x * kMixA) & kMask42; x ^= x >> 21;
    return x;
}
__device__ __host__ __forceinline__ u64 unmix42(u64 x) {
    x ^= x >> 21; x = (x * kMixAInv) & kMask42;
    return x;
}

// ---- tile pre-encoding (partitioned path, k <= 64) -------------------------------------------------
// Sixteen ASCII bases held in a uint4 (byte 0 = first base) -> one 32-bit word of 2-bit codes with
// the first base in bits 31:30, plus a 16-bit validity word with the first base in bit 15.
// SWAR on whole dwords: every base is classified exactly once per tile instead of once per
// window that touches it.
__device__ __forceinline__ void encode4(u32 w, u32 &codes8, u32 &valid4) {
    u32 x = (w >> 1) & 0x03030303u;          // A0 C1 G3 T2 per byte
    x ^= (x >> 1) & 0x01010101u;             // A0 C1 G2 T3
    // the byte must equal the letter its code stands for (case folded); anything else is invalid
    const u32 d = (w | 0x20202020u) ^ __builtin_amdgcn_perm(0u, 0x74676361u /* 'a','c','g','t' */, x);
    u32 t = (d & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    t = ~(t | d | 0x7f7f7f7fu);              // 0x80 in every byte of d that is zero
    codes8 = (x * ((1u << 30) | (1u << 20) | (1u << 10) | 1u)) >> 24;             // byte0 -> bits 7:6 ... byte3 -> bits 1:0
    valid4 = ((t >> 7) * ((1u << 27) | (1u << 18) | (1u << 9) | 1u)) >> 24 & 0xFu; // byte0 -> bit 3 ... byte3 -> bit 0
}

__device__ __forceinline__ void encode16(uint4 v, u32 &codes, u32 &valid) {
    u32 c0, c1, c2, c3, v0, v1, v2, v3;
    encode4(v.x, c0, v0); encode4(v.y, c1, v1); encode4(v.z, c2, v2); encode4(v.w, c3, v3);
    codes = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    valid = (v0 << 12) | (v1 << 8) | (v2 << 4) | v3;
}

// reverse the order of the 2-bit groups of a 64-bit word
__device__ __forceinline__ u64 reverse_pairs64(u64 x) {
    const u64 y = __builtin_bitreverse64(x);
    return ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
}

// reverse complement of a packed k-mer: complement every base, reverse their order
template <int KW>
__device__ __forceinline__ Packed<KW> revcomp_packed(Packed<KW> a, int k) {
    left_align(a, k);
    Packed<KW> r;
#pragma unroll
    for (int i = 0; i < KW; ++i) r.w[i] = ~reverse_pairs64(a.w[KW - 1 - i]);
    mask_k(r, k);
    return r;
}

// ---- generic path for any k (1..255): works on upper-casing bytes in LDS -----------------------
__device__ __forceinline__ u32 upper(u32 c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }
__device__ __forceinline__ u32 comp_ascii(u32 c) {  // c is one of ACGT (upper case)
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
}

// s points at the first byte of a window known to be all-ACGT (any case)
__device__ __forceinline__ u64 hash_bytes_canonical(const unsigned char *s, int k) {
    bool use_rc = false;
    for (int i = 0; i < k; ++i) {
        u32 f = upper(s[i]), r = comp_ascii(upper(s[k - 1 - i]));
        if (f != r) { use_rc = r < f; break; }
    }
    Murmur m;
    const int nblocks = k >> 4, rem = k & 15;
    auto byte_at = [&](int i) -> u64 { return use_rc ? comp_ascii(upper(s[k - 1 - i])) : upper(s[i]); };
    for (int b = 0; b < nblocks; ++b) {
        u64 k1 = 0, k2 = 0;
        for (int i = 7; i >= 0; --i) { k1 = (k1 << 8) | byte_at(16 * b + i); k2 = (k2 << 8) | byte_at(16 * b + 8 + i); }
        m.block(k1, k2);
    }
    if (rem) {
        u64 k1 = 0, k2 = 0;
        for (int i = rem - 1; i >= 8; --i) k2 = (k2 << 8) | byte_at(16 * nblocks + i);
        for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 = (k1 << 8) | byte_at(16 * nblocks + i);
        m.tail(k1, k2, rem);
    }
    return m.finish((u64)k);
}

}  // namespace kct
