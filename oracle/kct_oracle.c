/*
 * kct_oracle.c -- CPU restatement of oxli's KmerCountTable hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The product path
 * (oxli_amd/, include/kct.h) never links, imports or calls anything in this directory.
 *
 * What is restated (all citations relative to /root/reference):
 *   - KmerCountTable state, consume/count/count_hash/get/add:  src/lib.rs:32-39, 65-81,
 *     100-104, 145-194, 545-607, 778-837.
 *   - The arithmetic underneath lives in un-vendored third-party crates and is restated from
 *     their published algorithms: sourmash 0.23.0 (Cargo.toml:20, Cargo.lock:1200-1203):
 *     signature::SeqToHashes, encodings::{revcomp, VALID, COMPLEMENT}, _hash_murmur; and
 *     murmurhash3 0.0.5 (Cargo.lock:628-631): murmurhash3_x64_128 (Appleby's MurmurHash3).
 *     Call sites in the reference: src/lib.rs:69-76, 576-584.
 *
 * Parity pin: the 18 known-answer hashes and the n / len / consumed facts held by the
 * reference's own tests and docs (tests/golden/reference_kats.json lists each with its
 * file:line), checked by tests/test_oracle.py.  The Rust reference itself cannot be built
 * here (no cargo/rustc), so values for k >= 16 (MurmurHash3's 16-byte block loop) are pinned
 * against an independent canonical MurmurHash3_x64_128 (tests/golden/make_golden.py).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_OK 0
#define ORC_ERR_WRONG_KSIZE 1
#define ORC_ERR_INVALID_DNA 2
#define ORC_ERR_BAD_KMER 3
#define ORC_ERR_KSIZE_MISMATCH 4
#define ORC_ERR_NOMEM 5

/* ------------------------------------------------------------------------------------------
 * MurmurHash3_x64_128, low 64 bits.  murmurhash3 0.0.5 `murmurhash3_x64_128(bytes, seed)`;
 * sourmash `_hash_murmur(kmer, seed)` keeps `.0` (h1).  Used by lib.rs:69-76 / 576-584 with
 * seed 42.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

uint64_t orc_murmur64(const uint8_t *data, size_t len, uint64_t seed) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    size_t nblocks = len / 16;
    for (size_t i = 0; i < nblocks; ++i) {
        uint64_t k1 = load_le64(data + 16 * i), k2 = load_le64(data + 16 * i + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    size_t rem = len & 15;
    if (rem > 8) {
        for (size_t i = rem; i-- > 8;) k2 = (k2 << 8) | tail[i];
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (rem > 0) {
        size_t top = rem > 8 ? 8 : rem;
        for (size_t i = top; i-- > 0;) k1 = (k1 << 8) | tail[i];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

/* ------------------------------------------------------------------------------------------
 * sourmash encodings: VALID (only A C G T after upper-casing), COMPLEMENT / revcomp
 * (A<->T, C<->G, N->N, everything else -> 0).
 * ---------------------------------------------------------------------------------------- */
static inline int valid_base(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

static inline uint8_t complement(uint8_t c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'N': return 'N';
        default: return 0;
    }
}

static inline uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

/* ------------------------------------------------------------------------------------------
 * sourmash SeqToHashes (DNA branch) as driven by lib.rs:576-600.
 * State mirrors the crate's iterator: upper-cased copy made at construction, full-sequence
 * reverse complement made on the first step, validity checked incrementally from
 * max(kmer_index, last_position_check).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *seq;   /* upper-cased copy */
    uint8_t *rc;    /* reverse complement of the whole sequence (lazy) */
    size_t len, k, kmer_index, max_index, last_check;
    int force;
    uint64_t seed;
} seq_iter;

static int iter_init(seq_iter *it, const uint8_t *s, size_t len, size_t k, int force, uint64_t seed) {
    it->seq = (uint8_t *)malloc(len ? len : 1);
    if (!it->seq) return ORC_ERR_NOMEM;
    for (size_t i = 0; i < len; ++i) it->seq[i] = ascii_upper(s[i]);
    it->rc = NULL;
    it->len = len; it->k = k; it->kmer_index = 0; it->last_check = 0;
    it->max_index = len >= k ? len - k + 1 : 0;
    it->force = force; it->seed = seed;
    return ORC_OK;
}

static void iter_free(seq_iter *it) { free(it->seq); free(it->rc); }

/* the same iterator over caller-owned buffers of >= len bytes each (the full-size checker walks 10^8 records per thread:
 * no malloc per record); nothing to free */
static void iter_attach(seq_iter *it, const uint8_t *s, size_t len, size_t k, int force, uint64_t seed, uint8_t *seq_buf, uint8_t *rc_buf) {
    it->seq = seq_buf;
    for (size_t i = 0; i < len; ++i) it->seq[i] = ascii_upper(s[i]);
    it->rc = rc_buf;
    for (size_t i = 0; i < len; ++i) it->rc[i] = complement(it->seq[len - 1 - i]);
    it->len = len; it->k = k; it->kmer_index = 0; it->last_check = 0;
    it->max_index = len >= k ? len - k + 1 : 0;
    it->force = force; it->seed = seed;
}

/* returns 0 = end, 1 = Ok(*out), 2 = Err(InvalidDNA) */
static int iter_next(seq_iter *it, uint64_t *out) {
    if (it->kmer_index >= it->max_index) return 0;
    if (!it->rc) {
        it->rc = (uint8_t *)malloc(it->len ? it->len : 1);
        for (size_t i = 0; i < it->len; ++i) it->rc[i] = complement(it->seq[it->len - 1 - i]);
    }
    size_t i = it->kmer_index, k = it->k;
    size_t j0 = i > it->last_check ? i : it->last_check;
    for (size_t j = j0; j < i + k; ++j) {
        if (!valid_base(it->seq[j])) {
            if (!it->force) return 2;
            it->kmer_index += 1;
            *out = 0;
            return 1;
        }
        it->last_check += 1;
    }
    const uint8_t *fw = it->seq + i;
    const uint8_t *rv = it->rc + (it->len - k - i);
    const uint8_t *canon = memcmp(fw, rv, k) <= 0 ? fw : rv;
    *out = orc_murmur64(canon, k, it->seed);
    it->kmer_index += 1;
    return 1;
}

/* Per-window hashes of `seq` (0 for a skipped window when force).  Returns the number of
 * values written (<= cap); *err = 1 if iteration stopped on an invalid window (force == 0). */
size_t orc_seq_to_hashes(const char *seq, size_t len, size_t k, int force, uint64_t *out, size_t cap, int *err) {
    seq_iter it;
    *err = 0;
    if (iter_init(&it, (const uint8_t *)seq, len, k, force, 42) != ORC_OK) { *err = 2; return 0; }
    size_t n = 0;
    uint64_t h;
    int r;
    while ((r = iter_next(&it, &h)) != 0) {
        if (r == 2) { *err = 1; break; }
        if (n < cap) out[n] = h;
        ++n;
    }
    iter_free(&it);
    return n;
}

/* lib.rs:65-81 hash_kmer: `(kmer.len() as u8) != ksize` -> "wrong ksize"; else first item of
 * SeqToHashes(force = false). */
int orc_hash_kmer(const char *kmer, size_t len, uint8_t ksize, uint64_t *out) {
    if ((uint8_t)len != ksize) return ORC_ERR_WRONG_KSIZE;
    seq_iter it;
    if (iter_init(&it, (const uint8_t *)kmer, len, ksize, 0, 42) != ORC_OK) return ORC_ERR_NOMEM;
    int r = iter_next(&it, out);
    iter_free(&it);
    return r == 1 ? ORC_OK : ORC_ERR_INVALID_DNA;
}

/* ------------------------------------------------------------------------------------------
 * u64 -> u64 map standing in for Rust's std HashMap (hashbrown + SipHash-1-3, lib.rs:3,32-33):
 * SipHash-1-3 of the key with per-table random keys, open addressing, growth by doubling at
 * 7/8 load.  An occupancy byte array lets key 0 be stored (count_hash(0) is legal, lib.rs:100).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t *keys, *vals;
    uint8_t *used;
    size_t cap, len; /* cap is a power of two or 0 */
    uint64_t k0, k1;
} u64map;

#define SIPROUND(v0, v1, v2, v3)                                                   \
    do {                                                                           \
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);              \
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;                                   \
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;                                   \
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);              \
    } while (0)

static inline uint64_t siphash13_u64(uint64_t k0, uint64_t k1, uint64_t m) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
    v3 ^= m; SIPROUND(v0, v1, v2, v3); v0 ^= m;
    uint64_t b = (uint64_t)8 << 56;
    v3 ^= b; SIPROUND(v0, v1, v2, v3); v0 ^= b;
    v2 ^= 0xff;
    SIPROUND(v0, v1, v2, v3); SIPROUND(v0, v1, v2, v3); SIPROUND(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}

static void map_init(u64map *m, uint64_t seed) {
    memset(m, 0, sizeof *m);
    m->k0 = 0x0706050403020100ULL ^ seed;
    m->k1 = 0x0f0e0d0c0b0a0908ULL ^ (seed * 0x9e3779b97f4a7c15ULL);
}

static void map_free(u64map *m) { free(m->keys); free(m->vals); free(m->used); memset(m, 0, sizeof *m); }

static uint64_t *map_slot(u64map *m, uint64_t key, int *fresh);

static int map_grow(u64map *m) {
    u64map n = *m;
    n.cap = m->cap ? m->cap * 2 : 4;
    n.len = 0;
    n.keys = (uint64_t *)malloc(n.cap * sizeof(uint64_t));
    n.vals = (uint64_t *)malloc(n.cap * sizeof(uint64_t));
    n.used = (uint8_t *)calloc(n.cap, 1);
    if (!n.keys || !n.vals || !n.used) return ORC_ERR_NOMEM;
    for (size_t i = 0; i < m->cap; ++i)
        if (m->used[i]) {
            int fresh;
            *map_slot(&n, m->keys[i], &fresh) = m->vals[i];
        }
    free(m->keys); free(m->vals); free(m->used);
    *m = n;
    return ORC_OK;
}

/* room for n keys without further growth (an EMPTY map only; the checker's shard sets know how many keys to expect) */
static void map_reserve(u64map *m, size_t n) {
    if (m->len || !n) return;
    size_t cap = 4;
    while (cap * 7 < (n + 1) * 8) cap *= 2;
    if (cap <= m->cap) return;
    free(m->keys); free(m->vals); free(m->used);
    m->cap = cap;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->used = (uint8_t *)calloc(cap, 1);
}

/* find-or-insert (value initialised to 0 when fresh) */
static uint64_t *map_slot(u64map *m, uint64_t key, int *fresh) {
    if ((m->len + 1) * 8 > m->cap * 7) map_grow(m);
    size_t mask = m->cap - 1, i = (size_t)siphash13_u64(m->k0, m->k1, key) & mask;
    while (m->used[i]) {
        if (m->keys[i] == key) { *fresh = 0; return &m->vals[i]; }
        i = (i + 1) & mask;
    }
    m->used[i] = 1; m->keys[i] = key; m->vals[i] = 0; m->len++;
    *fresh = 1;
    return &m->vals[i];
}

static uint64_t map_get(const u64map *m, uint64_t key) {
    if (!m->cap) return 0;
    size_t mask = m->cap - 1, i = (size_t)siphash13_u64(m->k0, m->k1, key) & mask;
    while (m->used[i]) {
        if (m->keys[i] == key) return m->vals[i];
        i = (i + 1) & mask;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * KmerCountTable (lib.rs:32-39) -- counts, ksize, consumed.
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_table {
    u64map counts;
    uint8_t ksize;
    uint64_t consumed;
} orc_table;

orc_table *orc_new(uint8_t ksize) {
    static uint64_t ctr = 0;
    orc_table *t = (orc_table *)calloc(1, sizeof *t);
    if (!t) return NULL;
    map_init(&t->counts, ++ctr);
    t->ksize = ksize;
    return t;
}

void orc_free(orc_table *t) { if (t) { map_free(&t->counts); free(t); } }

/* lib.rs:100-104 */
uint64_t orc_count_hash(orc_table *t, uint64_t h) {
    int fresh;
    uint64_t *v = map_slot(&t->counts, h, &fresh);
    return ++*v;
}

/* lib.rs:185-188 */
uint64_t orc_get_hash(const orc_table *t, uint64_t h) { return map_get(&t->counts, h); }

/* lib.rs:145-167 (store_kmers == false).  *count_out = new count. */
int orc_count(orc_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    if ((uint8_t)len != t->ksize) return ORC_ERR_WRONG_KSIZE;
    uint64_t h;
    int st = orc_hash_kmer(kmer, len, t->ksize, &h);
    if (st != ORC_OK) return st;
    *count_out = orc_count_hash(t, h);
    t->consumed += len;
    return ORC_OK;
}

/* lib.rs:170-182 */
int orc_get(const orc_table *t, const char *kmer, size_t len, uint64_t *count_out) {
    if ((uint8_t)len != t->ksize) return ORC_ERR_WRONG_KSIZE;
    uint64_t h;
    int st = orc_hash_kmer(kmer, len, t->ksize, &h);
    if (st != ORC_OK) return st; /* the reference panics here (lib.rs:176) */
    *count_out = map_get(&t->counts, h);
    return ORC_OK;
}

/* lib.rs:545-607, plain branch (574-601).  On ORC_ERR_BAD_KMER *n_out holds the number of
 * k-mers counted before the bad window, they stay counted and `consumed` is unchanged. */
int orc_consume(orc_table *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out) {
    seq_iter it;
    uint64_t n = 0, h;
    int r, st = ORC_OK;
    if (iter_init(&it, (const uint8_t *)seq, len, t->ksize, skip_bad, 42) != ORC_OK) return ORC_ERR_NOMEM;
    while ((r = iter_next(&it, &h)) != 0) {
        if (r == 2) { st = ORC_ERR_BAD_KMER; break; }
        if (h == 0) continue;
        orc_count_hash(t, h);
        ++n;
    }
    iter_free(&it);
    *n_out = n;
    if (st == ORC_OK) t->consumed += len;
    return st;
}

uint64_t orc_len(const orc_table *t) { return t->counts.len; }
uint64_t orc_consumed(const orc_table *t) { return t->consumed; }
uint8_t orc_ksize(const orc_table *t) { return t->ksize; }

/* lib.rs:536-539 */
uint64_t orc_sum_counts(const orc_table *t) {
    uint64_t s = 0;
    for (size_t i = 0; i < t->counts.cap; ++i) if (t->counts.used[i]) s += t->counts.vals[i];
    return s;
}

typedef struct { uint64_t k, v; } kv;
static int kv_cmp(const void *a, const void *b) {
    uint64_t x = ((const kv *)a)->k, y = ((const kv *)b)->k;
    return x < y ? -1 : x > y;
}

/* dump(sortkeys=True) (lib.rs:330-381): (hash, count) ascending by hash.  Returns len. */
uint64_t orc_dump_sorted(const orc_table *t, uint64_t *keys, uint64_t *counts, uint64_t cap) {
    size_t n = t->counts.len;
    kv *tmp = (kv *)malloc((n ? n : 1) * sizeof(kv));
    size_t j = 0;
    for (size_t i = 0; i < t->counts.cap; ++i)
        if (t->counts.used[i]) { tmp[j].k = t->counts.keys[i]; tmp[j].v = t->counts.vals[i]; ++j; }
    qsort(tmp, n, sizeof(kv), kv_cmp);
    for (size_t i = 0; i < n && i < cap; ++i) { keys[i] = tmp[i].k; counts[i] = tmp[i].v; }
    free(tmp);
    return n;
}

/* lib.rs:778-837 add: per-key sum; (sum of other's counts, keys new to self); consumed sum. */
int orc_add(orc_table *dst, const orc_table *src, uint64_t *total_added, uint64_t *new_keys) {
    if (dst->ksize != src->ksize) return ORC_ERR_KSIZE_MISMATCH;
    uint64_t tot = 0, nk = 0;
    for (size_t i = 0; i < src->counts.cap; ++i)
        if (src->counts.used[i]) {
            int fresh;
            uint64_t *v = map_slot(&dst->counts, src->counts.keys[i], &fresh);
            if (*v == 0) ++nk; /* lib.rs:801-803: "new" means current count == 0 */
            *v += src->counts.vals[i];
            tot += src->counts.vals[i];
        }
    dst->consumed += src->consumed;
    *total_added = tot; *new_keys = nk;
    return ORC_OK;
}

/* add raw (hash, count) pairs -- used by tests to model an owner-side shard merge */
void orc_add_pairs(orc_table *t, const uint64_t *keys, const uint64_t *counts, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        int fresh;
        *map_slot(&t->counts, keys[i], &fresh) += counts[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * Synthetic workload (SURVEY.md 8d / DESIGN.md): counter-based so CPU and GPU make the same
 * bytes.  mix64 = splitmix64's output function applied to (x + golden gamma).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

uint64_t orc_mix64(uint64_t x) { return mix64(x); }

void orc_synth_genome(uint8_t *out, uint64_t G, uint64_t seed_g) {
    for (uint64_t j = 0; j < G; ++j) out[j] = (uint8_t)"ACGT"[mix64(seed_g + j) & 3];
}

/* reads [first, first+count) of the stream, each L bases followed by one '\n' separator
 * (stride L+1) -- the record-separated layout the device consumes. */
void orc_synth_reads(uint8_t *out, const uint8_t *genome, uint64_t G, uint64_t first, uint64_t count,
                     uint32_t L, uint64_t seed_r) {
    for (uint64_t r = 0; r < count; ++r) {
        uint64_t i = first + r;
        uint64_t start = mix64(seed_r + 2 * i) % (G - L + 1);
        int strand = (int)(mix64(seed_r + 2 * i + 1) & 1);
        uint8_t *dst = out + r * (uint64_t)(L + 1);
        if (!strand) memcpy(dst, genome + start, L);
        else for (uint32_t j = 0; j < L; ++j) dst[j] = complement(genome[start + L - 1 - j]);
        dst[L] = '\n';
    }
}

/* The same stream with a sequencing-error model (SURVEY.md 8d's secondary inputs; include/kct_synth.h):
 *   e = mix64(seed_e + i * L + j) for base j of read i;  u = e mod 10^6;
 *   u < n_ppm            -> 'N'
 *   u < n_ppm + sub_ppm  -> "ACGT"[(code + 1 + (e >> 32) mod 3) & 3]   (a base that differs from the true one)
 *   sorted_total > 0     -> start = i * (G - L + 1) / sorted_total     (position-sorted reads; the strand stays random) */
static inline uint8_t synth_base(uint8_t c, uint64_t i, uint32_t L, uint32_t j, uint32_t sub_ppm, uint32_t n_ppm, uint64_t seed_e) {
    if (!(sub_ppm | n_ppm)) return c;
    const uint64_t e = mix64(seed_e + i * (uint64_t)L + j), u = e % 1000000u;
    if (u < n_ppm) return (uint8_t)'N';
    if (u < (uint64_t)n_ppm + sub_ppm) {
        const unsigned code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
        return (uint8_t)"ACGT"[(code + 1u + (unsigned)((e >> 32) % 3u)) & 3u];
    }
    return c;
}

void orc_synth_reads_ex(uint8_t *out, const uint8_t *genome, uint64_t G, uint64_t first, uint64_t count, uint32_t L, uint64_t seed_r,
                        uint32_t sub_ppm, uint32_t n_ppm, uint64_t sorted_total, uint64_t seed_e) {
    for (uint64_t r = 0; r < count; ++r) {
        const uint64_t i = first + r;
        const uint64_t start = sorted_total ? (i % sorted_total) * (G - L + 1) / sorted_total : mix64(seed_r + 2 * i) % (G - L + 1);
        const int strand = (int)(mix64(seed_r + 2 * i + 1) & 1);
        uint8_t *dst = out + r * (uint64_t)(L + 1);
        for (uint32_t j = 0; j < L; ++j) {
            const uint8_t c = strand ? complement(genome[start + L - 1 - j]) : genome[start + j];
            dst[j] = synth_base(c, i, L, j, sub_ppm, n_ppm, seed_e);
        }
        dst[L] = '\n';
    }
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline ("port" of the reference CPU path): one orc_consume per record, exactly as the
 * README loop does (README.md:96-98); with T threads each thread owns a private table over a
 * contiguous shard of records and shards are merged with `add` semantics, merge time included.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *reads; uint64_t first, count; uint32_t L; uint8_t k;
    orc_table *t; uint64_t n;
} shard_job;

static void *shard_run(void *p) {
    shard_job *j = (shard_job *)p;
    j->t = orc_new(j->k);
    uint64_t n = 0, m;
    for (uint64_t r = 0; r < j->count; ++r) {
        orc_consume(j->t, (const char *)(j->reads + (j->first + r) * (uint64_t)(j->L + 1)), j->L, 1, &m);
        n += m;
    }
    j->n = n;
    return NULL;
}

typedef struct { orc_table *dst, *src; } merge_job;

static void *merge_run(void *p) {
    merge_job *m = (merge_job *)p;
    uint64_t a, b;
    orc_add(m->dst, m->src, &a, &b);
    orc_free(m->src);
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Returns the merged table (caller frees) so the caller can check it; *seconds covers consume
 * + merge; *kmers = sum of n. */
orc_table *orc_baseline_consume(const uint8_t *reads, uint64_t nreads, uint32_t L, uint8_t k, int threads,
                                uint64_t *kmers, double *seconds) {
    if (threads < 1) threads = 1;
    shard_job *jobs = (shard_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
    double t0 = now_s();
    for (int i = 0; i < threads; ++i) {
        jobs[i].reads = reads; jobs[i].L = L; jobs[i].k = k;
        jobs[i].first = nreads * (uint64_t)i / (uint64_t)threads;
        jobs[i].count = nreads * (uint64_t)(i + 1) / (uint64_t)threads - jobs[i].first;
        if (threads == 1) shard_run(&jobs[i]);
        else pthread_create(&tid[i], NULL, shard_run, &jobs[i]);
    }
    uint64_t n = 0;
    for (int i = 0; i < threads; ++i) {
        if (threads > 1) pthread_join(tid[i], NULL);
        n += jobs[i].n;
    }
    /* merge the shards with add() semantics as a binary tree, each level's pair-merges in parallel
     * (the reference's add() is serial under a mutex, lib.rs:798-806; this is the kindest reading
     * of "rayon-style") */
    merge_job *mj = (merge_job *)calloc((size_t)threads, sizeof *mj);
    for (int stride = 1; stride < threads; stride *= 2) {
        int nm = 0;
        for (int i = 0; i + stride < threads; i += 2 * stride) {
            mj[nm].dst = jobs[i].t; mj[nm].src = jobs[i + stride].t;
            pthread_create(&tid[nm], NULL, merge_run, &mj[nm]);
            ++nm;
        }
        for (int i = 0; i < nm; ++i) pthread_join(tid[i], NULL);
    }
    free(mj);
    orc_table *dst = jobs[0].t;
    *seconds = now_s() - t0;
    *kmers = n;
    free(jobs); free(tid);
    return dst;
}

/* ------------------------------------------------------------------------------------------
 * The same per-record work with the KEY SPACE sharded instead of the reads ("best CPU" row of the
 * bench): thread i hashes its slice of a batch of reads exactly as orc_consume does (upper-cased copy,
 * reverse complement, bytewise min, MurmurHash3 -- lib.rs:576-600) but hands every hash to the thread
 * that owns its slice of hash space; after a barrier each owner counts what it was handed into its
 * own table (count_hash, lib.rs:100-104).  No two threads ever hold the same key, so there is nothing
 * to merge (the reference's add(), lib.rs:778-837, never runs).  *seconds covers both phases of every
 * batch.  The per-owner tables are folded into one AFTER the clock stops, for the caller's check.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t *v; size_t n, cap; } hbuf;

typedef struct {
    const uint8_t *reads; uint64_t nreads, batch; uint32_t L; uint8_t k;
    int id, threads;
    hbuf *out;              /* [threads][threads]: out[id * threads + owner] */
    orc_table **tables;     /* [threads] */
    pthread_barrier_t *bar;
    uint64_t n;
} sharded_job;

static void hbuf_push(hbuf *b, uint64_t h) {
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 4096;
        b->v = (uint64_t *)realloc(b->v, b->cap * sizeof(uint64_t));
    }
    b->v[b->n++] = h;
}

static void *sharded_run(void *p) {
    sharded_job *j = (sharded_job *)p;
    const int T = j->threads;
    orc_table *mine = j->tables[j->id];
    uint64_t n = 0;
    for (uint64_t b0 = 0; b0 < j->nreads; b0 += j->batch) {
        const uint64_t bn = j->nreads - b0 < j->batch ? j->nreads - b0 : j->batch;
        const uint64_t lo = b0 + bn * (uint64_t)j->id / (uint64_t)T, hi = b0 + bn * (uint64_t)(j->id + 1) / (uint64_t)T;
        for (int o = 0; o < T; ++o) j->out[(size_t)j->id * T + o].n = 0;
        for (uint64_t r = lo; r < hi; ++r) {   /* phase 1: hash my reads, route by owner */
            seq_iter it;
            uint64_t h;
            int rc;
            if (iter_init(&it, j->reads + r * (uint64_t)(j->L + 1), j->L, j->k, 1, 42) != ORC_OK) continue;
            while ((rc = iter_next(&it, &h)) != 0) {
                if (rc == 2 || h == 0) continue;
                const int owner = (int)(((h >> 32) * (uint64_t)T) >> 32);
                hbuf_push(&j->out[(size_t)j->id * T + owner], h);
                ++n;
            }
            iter_free(&it);
            mine->consumed += j->L;
        }
        pthread_barrier_wait(j->bar);
        for (int src = 0; src < T; ++src) {     /* phase 2: count what I own */
            const hbuf *b = &j->out[(size_t)src * T + j->id];
            for (size_t i = 0; i < b->n; ++i) orc_count_hash(mine, b->v[i]);
        }
        pthread_barrier_wait(j->bar);
    }
    j->n = n;
    return NULL;
}

orc_table *orc_sharded_consume(const uint8_t *reads, uint64_t nreads, uint32_t L, uint8_t k, int threads, uint64_t batch,
                               uint64_t *kmers, double *seconds) {
    if (threads < 1) threads = 1;
    if (batch < (uint64_t)threads) batch = (uint64_t)threads;
    const int T = threads;
    sharded_job *jobs = (sharded_job *)calloc((size_t)T, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof *tid);
    hbuf *out = (hbuf *)calloc((size_t)T * T, sizeof *out);
    orc_table **tables = (orc_table **)calloc((size_t)T, sizeof *tables);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)T);
    for (int i = 0; i < T; ++i) tables[i] = orc_new(k);
    double t0 = now_s();
    for (int i = 0; i < T; ++i) {
        jobs[i].reads = reads; jobs[i].nreads = nreads; jobs[i].batch = batch; jobs[i].L = L; jobs[i].k = k;
        jobs[i].id = i; jobs[i].threads = T; jobs[i].out = out; jobs[i].tables = tables; jobs[i].bar = &bar;
        pthread_create(&tid[i], NULL, sharded_run, &jobs[i]);
    }
    uint64_t n = 0;
    for (int i = 0; i < T; ++i) { pthread_join(tid[i], NULL); n += jobs[i].n; }
    *seconds = now_s() - t0;
    *kmers = n;
    for (int i = 1; i < T; ++i) {   /* untimed: one table for the caller's check (the owners' key sets are disjoint) */
        uint64_t a, b;
        orc_add(tables[0], tables[i], &a, &b);
        orc_free(tables[i]);
    }
    orc_table *dst = tables[0];
    for (size_t i = 0; i < (size_t)T * T; ++i) free(out[i].v);
    pthread_barrier_destroy(&bar);
    free(out); free(tables); free(jobs); free(tid);
    return dst;
}


/* ------------------------------------------------------------------------------------------
 * Full-size parity: a SHARD SET is orc_sharded_consume's T owner tables kept apart (owner = top bits of the hash), built
 * from reads in memory or -- reads == NULL -- from the synthetic stream generated on the fly (100 M reads need no 15 GB
 * buffer), so that the tables of the 10^8 .. 10^10-k-mer configurations can be checked EXACTLY:
 *   orc_shardset_digest      len, sum_counts, sum(hash * count) mod 2^64, xor(hash * count), min / max count,
 *                            sum(count^2) mod 2^64, n (k-mers counted), consumed
 *   orc_shardset_mismatches  how many of the caller's (key, count) pairs differ from the set's (with len equal: 0 <=> equal maps)
 * Per record this is orc_consume's work (iter_init / iter_next: lib.rs:576-600, count_hash: lib.rs:100-104).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int threads;
    orc_table **tables;
    uint64_t n;
} orc_shardset;

typedef struct {
    const uint8_t *reads; const uint8_t *genome; uint64_t G, first, nreads, batch; uint32_t L; uint8_t k;
    uint64_t seed_r, sorted_total, seed_e; uint32_t sub_ppm, n_ppm;
    int id, threads;
    hbuf *out; orc_table **tables; pthread_barrier_t *bar;
    uint64_t n;
} shardset_job;

static void *shardset_run(void *p) {
    shardset_job *j = (shardset_job *)p;
    const int T = j->threads;
    orc_table *mine = j->tables[j->id];
    uint8_t *buf = (uint8_t *)malloc((size_t)j->L + 1), *seq_buf = (uint8_t *)malloc((size_t)j->L + 1), *rc_buf = (uint8_t *)malloc((size_t)j->L + 1);
    uint64_t n = 0;
    for (uint64_t b0 = 0; b0 < j->nreads; b0 += j->batch) {
        const uint64_t bn = j->nreads - b0 < j->batch ? j->nreads - b0 : j->batch;
        const uint64_t lo = b0 + bn * (uint64_t)j->id / (uint64_t)T, hi = b0 + bn * (uint64_t)(j->id + 1) / (uint64_t)T;
        for (int o = 0; o < T; ++o) j->out[(size_t)j->id * T + o].n = 0;
        for (uint64_t r = lo; r < hi; ++r) {
            const uint8_t *rec;
            if (j->reads) rec = j->reads + r * (uint64_t)(j->L + 1);
            else { orc_synth_reads_ex(buf, j->genome, j->G, j->first + r, 1, j->L, j->seed_r, j->sub_ppm, j->n_ppm, j->sorted_total, j->seed_e); rec = buf; }
            seq_iter it;
            uint64_t h;
            int rc;
            mine->consumed += j->L;
            iter_attach(&it, rec, j->L, j->k, 1, 42, seq_buf, rc_buf);
            while ((rc = iter_next(&it, &h)) != 0) {
                if (rc == 2 || h == 0) continue;
                hbuf_push(&j->out[(size_t)j->id * T + (int)(((h >> 32) * (uint64_t)T) >> 32)], h);
                ++n;
            }
        }
        pthread_barrier_wait(j->bar);
        for (int src = 0; src < T; ++src) {
            const hbuf *b = &j->out[(size_t)src * T + j->id];
            /* (tables of 10^7 keys per owner do not fit any cache: the slot of the hash sixteen ahead is prefetched -- the
             * count_hash calls themselves are unchanged) */
            for (size_t i = 0; i < b->n; ++i) {
                if (i + 16 < b->n && mine->counts.cap) {
                    const size_t at = (size_t)siphash13_u64(mine->counts.k0, mine->counts.k1, b->v[i + 16]) & (mine->counts.cap - 1);
                    __builtin_prefetch(&mine->counts.used[at]);
                    __builtin_prefetch(&mine->counts.keys[at]);
                    __builtin_prefetch(&mine->counts.vals[at], 1);
                }
                orc_count_hash(mine, b->v[i]);
            }
        }
        pthread_barrier_wait(j->bar);
    }
    free(buf); free(seq_buf); free(rc_buf);
    j->n = n;
    return NULL;
}

/* reads != NULL: nreads records of stride L + 1 in memory; reads == NULL: records [first, first + nreads) of the synthetic
 * stream over `genome` (G bytes) with the given error model */
orc_shardset *orc_shardset_build(const uint8_t *reads, const uint8_t *genome, uint64_t G, uint64_t first, uint64_t nreads, uint32_t L,
                                 uint8_t k, uint64_t seed_r, uint32_t sub_ppm, uint32_t n_ppm, uint64_t sorted_total, uint64_t seed_e,
                                 int threads, uint64_t batch, uint64_t expect_keys) {
    if (threads < 1) threads = 1;
    if (batch < (uint64_t)threads) batch = (uint64_t)threads;
    const int T = threads;
    orc_shardset *s = (orc_shardset *)calloc(1, sizeof *s);
    shardset_job *jobs = (shardset_job *)calloc((size_t)T, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof *tid);
    hbuf *out = (hbuf *)calloc((size_t)T * T, sizeof *out);
    s->threads = T;
    s->tables = (orc_table **)calloc((size_t)T, sizeof *s->tables);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)T);
    for (int i = 0; i < T; ++i) { s->tables[i] = orc_new(k); map_reserve(&s->tables[i]->counts, (size_t)(expect_keys / (uint64_t)T + expect_keys / (uint64_t)T / 16)); }
    for (int i = 0; i < T; ++i) {
        shardset_job *j = &jobs[i];
        j->reads = reads; j->genome = genome; j->G = G; j->first = first; j->nreads = nreads; j->batch = batch; j->L = L; j->k = k;
        j->seed_r = seed_r; j->sub_ppm = sub_ppm; j->n_ppm = n_ppm; j->sorted_total = sorted_total; j->seed_e = seed_e;
        j->id = i; j->threads = T; j->out = out; j->tables = s->tables; j->bar = &bar;
        pthread_create(&tid[i], NULL, shardset_run, j);
    }
    for (int i = 0; i < T; ++i) { pthread_join(tid[i], NULL); s->n += jobs[i].n; }
    for (size_t i = 0; i < (size_t)T * T; ++i) free(out[i].v);
    pthread_barrier_destroy(&bar);
    free(out); free(jobs); free(tid);
    return s;
}

void orc_shardset_free(orc_shardset *s) {
    if (!s) return;
    for (int i = 0; i < s->threads; ++i) orc_free(s->tables[i]);
    free(s->tables); free(s);
}

/* out[9] = len, sum_counts, sum(hash * count), xor(hash * count), min count, max count, sum(count^2), n, consumed (all mod 2^64) */
void orc_shardset_digest(const orc_shardset *s, uint64_t out[9]) {
    uint64_t len = 0, sum = 0, shc = 0, xhc = 0, lo = ~0ULL, hi = 0, sq = 0, consumed = 0;
    for (int t = 0; t < s->threads; ++t) {
        const u64map *m = &s->tables[t]->counts;
        consumed += s->tables[t]->consumed;
        for (size_t i = 0; i < m->cap; ++i)
            if (m->used[i]) {
                const uint64_t h = m->keys[i], c = m->vals[i];
                ++len; sum += c; shc += h * c; xhc ^= h * c; sq += c * c;
                if (c < lo) lo = c;
                if (c > hi) hi = c;
            }
    }
    out[0] = len; out[1] = sum; out[2] = shc; out[3] = xhc; out[4] = len ? lo : 0; out[5] = hi; out[6] = sq; out[7] = s->n; out[8] = consumed;
}

uint64_t orc_shardset_get(const orc_shardset *s, uint64_t h) {
    return map_get(&s->tables[(int)(((h >> 32) * (uint64_t)s->threads) >> 32)]->counts, h);
}

typedef struct { const orc_shardset *s; const uint64_t *keys, *counts; uint64_t lo, hi, bad; } mismatch_job;

static void *mismatch_run(void *p) {
    mismatch_job *j = (mismatch_job *)p;
    uint64_t bad = 0;
    for (uint64_t i = j->lo; i < j->hi; ++i) bad += orc_shardset_get(j->s, j->keys[i]) != j->counts[i];
    j->bad = bad;
    return NULL;
}

/* pairs (keys[i], counts[i]) whose count differs from the set's (an absent key reads as 0) */
uint64_t orc_shardset_mismatches(const orc_shardset *s, const uint64_t *keys, const uint64_t *counts, uint64_t n, int threads) {
    if (threads < 1) threads = 1;
    mismatch_job *jobs = (mismatch_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
    uint64_t bad = 0;
    for (int i = 0; i < threads; ++i) {
        jobs[i].s = s; jobs[i].keys = keys; jobs[i].counts = counts;
        jobs[i].lo = n * (uint64_t)i / (uint64_t)threads; jobs[i].hi = n * (uint64_t)(i + 1) / (uint64_t)threads;
        pthread_create(&tid[i], NULL, mismatch_run, &jobs[i]);
    }
    for (int i = 0; i < threads; ++i) { pthread_join(tid[i], NULL); bad += jobs[i].bad; }
    free(jobs); free(tid);
    return bad;
}
