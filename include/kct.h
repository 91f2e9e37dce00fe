/*
 * kct.h -- C ABI of the MI355X k-mer counting engine (libkct_hip.so).
 *
 * This is the drop-in boundary for ONE path of oxli-bio/oxli: KmerCountTable's
 * count / consume / get family (reference src/lib.rs:41-194, 545-607, 778-837).  Each entry
 * point below names the reference interface it replaces.  A Rust `KmerCountTable` would bind
 * these with `extern "C"` (INTEGRATION.md shows the stub); in this repo the binding is the
 * ctypes class oxli_amd.KmerCountTable.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Host pointers unless the name says `_device`.
 *   - Every function returns a kct_status; results come back through out-parameters.
 *     Nothing throws or aborts across the boundary.
 *   - A kct_table is NOT thread-safe: one caller at a time, like the reference's `&mut self`
 *     under the GIL (lib.rs:546).  A second thread that enters while a call is running gets
 *     KCT_ERR_BUSY (pyo3's "Already borrowed") -- it is never let in.  All reads observe all
 *     earlier writes on the same handle.
 *   - Input buffers are borrowed for the duration of the call only.
 *   - There is no CPU fallback: without a usable gfx950 device every call that needs one
 *     fails with KCT_ERR_NO_DEVICE / KCT_ERR_HIP.
 */
#ifndef KCT_H
#define KCT_H

#include <stddef.h>
#include <stdint.h>

/* Only the entry points below are exported from libkct_hip.so (it is built with -fvisibility=hidden). */
#if defined(KCT_BUILDING_LIBRARY)
#define KCT_API __attribute__((visibility("default")))
#else
#define KCT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kct_table kct_table;

typedef enum kct_status {
    KCT_OK = 0,
    KCT_ERR_WRONG_KSIZE = 1,    /* lib.rs:67 "wrong ksize"; lib.rs:147,172 ValueError             */
    KCT_ERR_INVALID_DNA = 2,    /* lib.rs:79 hash_kmer on a non-ACGT k-mer (RuntimeError)          */
    KCT_ERR_BAD_KMER = 3,       /* lib.rs:593-596 consume(skip_bad_kmers=False) hit a bad window   */
    KCT_ERR_KSIZE_MISMATCH = 4, /* lib.rs:780-784 add() of tables with different ksize             */
    KCT_ERR_NOMEM = 5,          /* host or device allocation failed                                */
    KCT_ERR_HIP = 6,            /* a HIP runtime call failed; see kct_last_error()                 */
    KCT_ERR_ARG = 7,            /* null / misaligned / out-of-range argument                       */
    KCT_ERR_NO_DEVICE = 8,      /* no gfx950 device visible                                        */
    KCT_ERR_BUSY = 9            /* another thread is inside a call on this table (pyo3: "Already borrowed", lib.rs:546 &mut self) */
} kct_status;

/* Text of the most recent failure on this thread ("" if none). */
KCT_API const char *kct_last_error(void);

/* Number of visible HIP devices (0 on a CPU-only machine; never an error). */
KCT_API int kct_device_count(void);

/* ---- lifetime ---------------------------------------------------------------------------
 * KmerCountTable::new(ksize, store_kmers=false)                              lib.rs:44-62
 * `capacity_hint` = expected number of distinct k-mers (0 = default); the table grows by
 * itself, the hint only avoids early re-hashes.  `device` = HIP device ordinal. */
KCT_API kct_status kct_create(uint8_t ksize, uint64_t capacity_hint, int device, kct_table **out);
KCT_API void kct_destroy(kct_table *t);

/* Forget all counts and `consumed`; keeps the allocation. (No reference counterpart: a fresh
 * KmerCountTable.) */
KCT_API kct_status kct_clear(kct_table *t);

/* Ensure room for `distinct` keys without further growth. */
KCT_API kct_status kct_reserve(kct_table *t, uint64_t distinct);

/* Give the table the capacity that suits `distinct` keys (at least what it holds) -- smaller than it is, if that is
 * enough: the keys are re-inserted into a fresh array.  What the multi-GPU merge calls so that an owner's table is
 * sized for its slice of the key space, not for the whole genome. */
KCT_API kct_status kct_resize(kct_table *t, uint64_t distinct);

/* ---- hashing ----------------------------------------------------------------------------
 * KmerCountTable::hash_kmer(kmer)                                            lib.rs:65-81
 * WRONG_KSIZE if (uint8_t)len != ksize (the reference compares `len as u8`), INVALID_DNA if
 * the first ksize bytes are not all ACGT/acgt. */
KCT_API kct_status kct_hash_kmer(kct_table *t, const char *kmer, size_t len, uint64_t *hash_out);

/* sourmash SeqToHashes as consume drives it (lib.rs:576-600): one value per k-window of
 * `seq`, 0 for a window holding a non-ACGT byte.  Writes min(windows, cap) values, returns
 * the window count in *n_windows and the index of the first bad window in *first_bad
 * (== *n_windows if none). */
KCT_API kct_status kct_hash_windows(kct_table *t, const char *seq, size_t len, uint64_t *hashes_out, size_t cap,
                            uint64_t *n_windows, uint64_t *first_bad);

/* ---- point updates and lookups ------------------------------------------------------------
 * count_hash(hashval) -> new count                                           lib.rs:100-104
 * count(kmer) -> new count; consumed += len                                  lib.rs:145-167
 * get(kmer) / get_hash(hashval) / get_hash_array(hash_keys)                  lib.rs:170-194  */
KCT_API kct_status kct_count_hash(kct_table *t, uint64_t hash, uint64_t *count_out);
KCT_API kct_status kct_count(kct_table *t, const char *kmer, size_t len, uint64_t *count_out);
KCT_API kct_status kct_get(kct_table *t, const char *kmer, size_t len, uint64_t *count_out);
KCT_API kct_status kct_get_hash(kct_table *t, uint64_t hash, uint64_t *count_out);
KCT_API kct_status kct_get_hash_array(kct_table *t, const uint64_t *hashes, size_t n, uint64_t *counts_out);

/* __setitem__(kmer, count) is built from this                                lib.rs:675-681 */
KCT_API kct_status kct_set_hash(kct_table *t, uint64_t hash, uint64_t count);

/* ---- bulk ingest: the hot path -------------------------------------------------------------
 * consume(seq, skip_bad_kmers=true) -> n                                     lib.rs:545-607
 * *n_out = k-mers counted (valid windows whose hash is not 0).  With skip_bad == 0 and a bad
 * window present: returns KCT_ERR_BAD_KMER, *n_out = k-mers counted before it (they stay
 * counted), `consumed` unchanged -- the message is "bad k-mer encountered at position {n}".
 * DEVIATION in *n_out (every bulk entry point below shares it): the reference leaves a window whose MurmurHash3 value is exactly 0
 * out of n (lib.rs:589).  Passes that hash as they read do the same; DEFERRED calls (the default for skip_bad != 0: n comes from a
 * host-side validity scan), device-side STAGED calls and DEDUPE-FIRST passes (the hash is only computed at conversion time) count
 * such a window in n -- the table itself never receives key 0 on any path, so its contents are the reference's.  Probability
 * 2^-64 per distinct k-mer; demonstrated on every path by the `make zero` build (tests/test_gpu_zero_hash.py). */
KCT_API kct_status kct_consume(kct_table *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out);

/* 1 if kct_consume(t, seq of `len` bytes, skip_bad) would only APPEND to deferred mode's buffer (no device work, ~60 ns), 0 if it
 * would run a device pass (buffer full, error mode, deferred mode off, long record).  For call glue that holds a lock it would
 * rather not hold across a device pass (csrc/pyfast.c and the GIL; a pyo3 shim's `allow_threads`). */
KCT_API int kct_consume_will_defer(const kct_table *t, size_t len, int skip_bad);

/* The reference is called once per FASTA/FASTQ record (README.md:96-98).  This is the same
 * loop in one call: record r is bytes[offsets[r] .. offsets[r+1]); k-mers never span records.
 * *n_total = sum of the per-record n.  With skip_bad == 0 the call stops at the first record
 * holding a bad window exactly as the per-record loop would: KCT_ERR_BAD_KMER, *bad_record =
 * its index, *bad_position = the n of that record's own consume ("bad k-mer encountered at
 * position {n}"), records before it are fully counted, its windows before the bad one are
 * counted, later records are untouched, and `consumed` covers only the records before it.
 * Without an error *bad_record = nrec.  bad_record / bad_position may be NULL. */
KCT_API kct_status kct_consume_batch(kct_table *t, const char *bytes, const uint64_t *offsets, size_t nrec, int skip_bad,
                             uint64_t *n_total, uint64_t *bad_record, uint64_t *bad_position);

/* Where the last kct_consume_batch call of this table spent its time (a large skip_bad batch through the host packer; not in the
 * reference): out[0..15], milliseconds since the call began unless said otherwise --
 *   0 argument checks done             1 parts cut, staging reserved        2 first packer thread started its first part
 *   3 last packer thread finished      4 last slice's H2D copy enqueued     5 the device pass(es) submitted: the call returns
 *   6 packer threads used              7 sum of the threads' busy times     8 longest single thread's busy time
 *   9 source bytes read                10 packed bytes uploaded             11 minor page faults during the call (getrusage)
 *   12 distinct CPUs the packers ran on  13 distinct NUMA nodes of those CPUs  14 the calling thread's CPU  15 1 = packers pinned
 * All zero when the last batch did not take the packed-upload route. */
KCT_API kct_status kct_batch_timeline(kct_table *t, double *out16);

/* Same, for input already resident in HBM: `d_stream` is a 16-byte-aligned device pointer to
 * `nbytes` bytes in which records are separated by at least one non-ACGT byte (e.g. '\n');
 * skip_bad semantics.  `consumed` grows by `consumed_bytes` (the caller knows the record
 * lengths).  Runs on the table's stream; the caller's buffer is free again when the call returns.
 * A call that is SMALL for the table (fewer than 4 window starts per slot; fewer than 1 if it is the first call into an empty
 * table) is, in deferred mode (the default, kct_set_deferred), copied behind the earlier ones in HBM and counted with them -- when anything else touches the table, when 32 window starts per
 * slot have gathered or the staging buffer (<= 32 GiB, a quarter of the free HBM) is full -- so that an input fed in pieces is
 * counted in the passes, and on the path, of ONE large call; *n_total then comes from the copy kernel's own validity scan (the
 * all-ACGT rule of lib.rs:586-600; it differs from the reference's n only if a window's true hash is 0, probability 2^-64). */
KCT_API kct_status kct_consume_device(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes,
                              uint64_t *n_total);

/* The README loop itself (README.md:89-99: `for record in screed.open(file): kct.consume(record.sequence)`)
 * for a FASTA or FASTQ file, plain or gzip: a host parser builds record-stream chunks in pinned
 * memory while the previous chunk is uploaded and counted.  skip_bad must be non-zero (the
 * reference's default); *n_total = sum of the per-record n, *n_records / *n_bases = records and
 * sequence bytes read (`consumed` grows by *n_bases).  Out-parameters other than n_total may be NULL.
 * (The uploads and the device calls behind them are made by ONE helper thread that the library keeps for the life of the process --
 * a thread that has made HIP calls takes milliseconds to exit; a second kct_consume_file call that arrives while it is taken, on
 * another table from another thread, gets a thread of its own for that call.) */
KCT_API kct_status kct_consume_file(kct_table *t, const char *path, int skip_bad, uint64_t *n_total, uint64_t *n_records,
                            uint64_t *n_bases);
/* Which inflater kct_consume_file uses for gzip / BGZF input in this process: "libdeflate" (the system's libdeflate.so.0, found at
 * run time: whole-buffer inflate at 2-3x zlib's rate, several threads for BGZF blocks) or "zlib" (the fallback linked into the
 * library).  A plain gzip file of 4 MiB or more is inflated by several threads of the library's own either way (round 6: its members'
 * deflate streams are entered at block boundaries found by search, verified by length and CRC-32; KCT_NO_PARALLEL_GZIP=1 leaves them to
 * one thread of the inflater named here).  What bench.py records beside its file-input entries. */
KCT_API const char *kct_inflater_name(void);

/* ---- table attributes ---------------------------------------------------------------------
 * __len__ lib.rs:665-667; sum_counts lib.rs:536-539; consumed lib.rs:530-533; ksize lib.rs:34 */
KCT_API kct_status kct_len(kct_table *t, uint64_t *out);
KCT_API kct_status kct_sum_counts(kct_table *t, uint64_t *out);
KCT_API kct_status kct_consumed(kct_table *t, uint64_t *out);
KCT_API kct_status kct_add_consumed(kct_table *t, uint64_t delta);
KCT_API uint8_t kct_ksize(const kct_table *t);
KCT_API kct_status kct_capacity(kct_table *t, uint64_t *slots_out);

/* ---- dump / merge ---------------------------------------------------------------------------
 * dump(file=None, sortcounts, sortkeys) -> [(hash, count)]                   lib.rs:330-381
 * order: 0 = unspecified (the reference's HashMap order is unspecified too), 1 = by hash,
 * 2 = by (count, hash).  Writes min(len, cap) pairs; *n_out = len. */
KCT_API kct_status kct_dump(kct_table *t, uint64_t *hashes_out, uint64_t *counts_out, size_t cap, int order, uint64_t *n_out);

/* add(other) -> (total_counts_added, new_keys_added); consumed += other.consumed
 *                                                                            lib.rs:778-837 */
KCT_API kct_status kct_add(kct_table *dst, kct_table *src, uint64_t *total_added, uint64_t *new_keys);

/* The two halves of add() for shards that live on different GPUs / ranks: compact the table
 * into caller-owned device arrays, and fold (hash, count) pairs into a table with add()'s
 * tallies.  `d_*` are device pointers on the table's device. */
KCT_API kct_status kct_export_device(kct_table *t, void *d_hashes, void *d_counts, size_t cap, uint64_t *n_out);
KCT_API kct_status kct_merge_device(kct_table *t, const void *d_hashes, const void *d_counts, size_t n,
                            uint64_t *total_added, uint64_t *new_keys);
KCT_API kct_status kct_merge_host(kct_table *t, const uint64_t *hashes, const uint64_t *counts, size_t n,
                          uint64_t *total_added, uint64_t *new_keys);

/* The same two halves shaped for an all-to-all between ranks: the export writes interleaved
 * {hash, count} pairs (2 * cap u64 words at d_pairs) bucketed by owner rank,
 * owner(hash) = floor(hi32(hash) * nparts / 2^32), owner p's pairs contiguous and in rank order;
 * part_counts[p] (host, nparts entries) = pairs of owner p; *n_out = their sum.  nparts <= 256.
 * kct_merge_pairs_device folds n interleaved pairs with add()'s tallies. */
KCT_API kct_status kct_export_by_owner_device(kct_table *t, uint32_t nparts, void *d_pairs, size_t cap, uint64_t *part_counts,
                                      uint64_t *n_out);
KCT_API kct_status kct_merge_pairs_device(kct_table *t, const void *d_pairs, size_t n, uint64_t *total_added, uint64_t *new_keys);

/* ---- table analytics on the resident table (reference: lib.rs:197-267, 464-514, 610-655, 708-765) ------------
 * Scans, reductions and table-against-table lookups run on the device; nothing is dumped to the host first. */

/* min() / max() (lib.rs:492-514; both 0 for an empty table) and the sum of squared counts as f64, the
 * magnitude term of cosine() (lib.rs:747-760).  Any output pointer may be NULL. */
KCT_API kct_status kct_count_stats(kct_table *t, uint64_t *min_out, uint64_t *max_out, double *sum_squares_out);

/* ---- PACKED base arrays (BASELINE north star: "over packed base arrays") ------------------------------------------------
 * A record stream as 2 bits per base + 1 validity bit per base, sixteen bases per group: codes[g] (uint32, first base in bits
 * 31:30, A C G T = 0 1 2 3) and valid[g] (uint16, first base in bit 15; 0 = not ACGT / separator / padding).  Records are
 * separated by at least one invalid base, exactly as the ASCII stream separates them by a non-ACGT byte -- so the validity
 * rule of lib.rs:586-600 ("a window is good iff its k bytes are all ACGT after upper-casing") is the AND of k bits.  0.375 B
 * per base instead of 1: what kct_consume_batch uploads for large skip-bad batches (kct_set_packed_upload, on by default),
 * and what the partition kernels read directly.
 *   kct_consume_device_packed  counts nbases bases of a device-resident packed stream (kct_consume_device's twin)
 *   kct_pack_stream_device     ASCII record stream -> packed arrays, on the device (ceil(nbytes / 16) groups)              */
KCT_API kct_status kct_consume_device_packed(kct_table *t, const void *d_codes, const void *d_valid, size_t nbases, uint64_t consumed_bytes,
                                             uint64_t *n_total);
KCT_API kct_status kct_pack_stream_device(const void *d_stream, size_t nbytes, void *d_codes, void *d_valid, void *stream);
KCT_API kct_status kct_set_packed_upload(kct_table *t, int on);

/* ---- multi-GPU "early" route: every k-mer is counted by the GPU that owns it; SUPER-K-MERS cross the wire ---------
 * Independent records (README.md:96-98) and per-key sums (add(), lib.rs:778-837) let the key space be partitioned: owner(k-mer) =
 * (hash16(minimiser) >> 6) * world >> 10 (the top ten bits of a 16-bit hash spread over world <= 64 ranks), the minimiser being the smallest (in a scrambled order) canonical 8-mer inside the k-mer -- the same
 * for a k-mer and its reverse complement, and mostly the same for consecutive windows of a read.  Each rank cuts ITS records into
 * maximal runs of good windows with one owner and sends every run as 2-bit bases plus one start bit per window (~1 byte per window at
 * k = 21, ~0.9 at k = 51); every owner counts what it receives with the table's ordinary bulk path.  The ranks' tables end up a
 * disjoint partition of the key space: len / sum_counts of the global table are sums over ranks.  k <= 64.
 *
 * The library runs the whole call -- passes, pipelining (pass p + 1 is cut while pass p is on the wire, and is on the wire while pass
 * p is counted), error agreement -- and asks the caller only for the collective, through kct_exchange_ops (every rank of the job makes
 * the same call with the same world; all callbacks return 0 on success):
 *   alloc(user, bytes)          a device buffer the collective can send from / receive into; release(user, p) gives it back (may be NULL)
 *   exchange_sizes(user, send, nvals, recv)
 *                               blocking all-to-all in HOST memory: send[r * nvals ..] goes to rank r, recv[r * nvals ..] came from
 *                               rank r (nvals uint64 each)
 *   start(user, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes)
 *                               begins an all-to-all of byte ranges: rank r is sent d_send[send_off[r] .. + send_bytes[r]) and what
 *                               rank r sends lands at d_recv[recv_off[r] .. + recv_bytes[r]) (the sizes were agreed by
 *                               exchange_sizes; the send buffer is complete when start is called); may return at once
 *   wait(user)                  returns when the exchange started last has delivered every byte of d_recv
 * At most one exchange is in flight.  csrc/kct_rccl.cpp is an implementation over RCCL (ncclSend / ncclRecv on its own stream),
 * oxli_amd/distributed.py one over torch.distributed.  world == 1 with ops == NULL is a loop-back; world > 1 with ops == NULL counts,
 * of the records given, only the k-mers that `rank` owns and drops the rest -- for jobs in which every GPU reads ALL the input (no
 * exchange at all: the union over the ranks is again the whole table), and what tools/route_profile.py times one owner's share with.
 *   max_windows   window starts per pass (0 = what HBM allows, at least four passes for a long stream)
 *   *n_owned      k-mers this rank counted as an owner (summed over ranks: the job's n)
 *   stats16       optional: windows sent to / received from other ranks, bytes sent / received, runs cut, passes, microseconds in
 *                 the split / blocked in the exchange / in the owner's counting, region retries, window starts of this rank
 * A failure on any rank (a full HBM, a failed collective) ends the call on EVERY rank with an error: the tables then hold a partial
 * count and must be cleared. */
typedef struct kct_exchange_ops {
    void *user;
    void *(*alloc)(void *user, uint64_t bytes);
    void (*release)(void *user, void *p);
    int (*exchange_sizes)(void *user, const uint64_t *send, uint32_t nvals, uint64_t *recv);
    int (*start)(void *user, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_off,
                 const uint64_t *recv_bytes);
    int (*wait)(void *user);
} kct_exchange_ops;
KCT_API kct_status kct_consume_device_routed(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes, uint32_t world, uint32_t rank,
                                             const kct_exchange_ops *ops, uint64_t max_windows, uint64_t *n_owned, uint64_t *stats16);

/* Fault injection for the early route's error agreement (tests/test_gpu_distributed.py::test_a_failing_rank_ends_the_early_route_on_every_rank;
 * not in the reference): the NEXT kct_consume_device_routed call on this table fails once, on this rank only, at the named point --
 *   1 the HBM query before the first collective     2 the split (cut) of pass `pass`       3 the next slab allocation
 *   4 the start of pass `pass`'s payload             5 the wait for pass `pass`'s payload   6 the owner-side count
 * -- as if the HIP call (or the caller's callback) there had failed; 0 disarms.  The contract under test is the one stated above:
 * every rank of the job then returns an error from the same call, none is left inside a collective. */
#define KCT_FAULT_NONE 0
#define KCT_FAULT_MEMINFO 1
#define KCT_FAULT_SPLIT 2
#define KCT_FAULT_ALLOC 3
#define KCT_FAULT_START 4
#define KCT_FAULT_WAIT 5
#define KCT_FAULT_COUNT 6
KCT_API kct_status kct_debug_inject_fault(kct_table *t, int point, uint64_t pass);

/* The sender half alone (tests, tools, a caller with its own transport): this rank's records cut into super-k-mers for `world` owners.
 * *d_parts (device memory owned by the table, valid until its next bulk call) holds owner o's part at part_off[o] .. + part_bytes[o]:
 * the 16-byte base units (64 bases: 2 bits each, first base in the most significant bits of the first 32-bit word) of its
 * kct_superkmer_streams(t) streams, then their start units (128 windows: bit w % 64 of 64-bit word w / 64 = window w begins a run);
 * dir[o * streams + s] = windows | base units << 32 of stream s.  A run of n windows is n + k - 1 bases; windows are numbered in the
 * order their runs lie in the stream. */
KCT_API kct_status kct_superkmer_split_device(kct_table *t, const void *d_stream, size_t nbytes, uint32_t world, const void **d_parts,
                                              uint64_t *part_off, uint64_t *part_bytes, uint64_t *dir);
KCT_API uint32_t kct_superkmer_streams(const kct_table *t);

/* An order-free digest of the table's contents, computed by one scan on the device: sum and xor over all keys of
 * hash * count, and the sum of count^2 (all wrapping u64).  Not in the reference; it lets a table of 10^8 .. 10^9 keys be
 * compared EXACTLY with the CPU oracle's (tests/test_gpu_scale.py) where comparing dumps would not be practical
 * (cf. the digests SURVEY.md 8c records for doc/example.fa). */
KCT_API kct_status kct_digest(kct_table *t, uint64_t *sum_hc_out, uint64_t *xor_hc_out, uint64_t *sum_sq_out);

/* histo(zero=False) (lib.rs:464-488): the distinct count values in ascending order and how many keys have
 * each.  *n_out = number of distinct values; at most `cap` are written. */
KCT_API kct_status kct_histogram(kct_table *t, uint64_t *values_out, uint64_t *freq_out, size_t cap, uint64_t *n_out);

/* mincut / maxcut (lib.rs:227-267): keep only keys with min_count <= count <= max_count; *removed_out = keys
 * removed.  The survivors are re-inserted on the device (the probe layout has no tombstones). */
KCT_API kct_status kct_retain_counts(kct_table *t, uint64_t min_count, uint64_t max_count, uint64_t *removed_out);

/* drop_hash (lib.rs:213-224): remove one key if present; *removed_out = 0 or 1. */
KCT_API kct_status kct_remove_hash(kct_table *t, uint64_t hash, uint64_t *removed_out);

/* |keys(a) & keys(b)| and the dot product sum(count_a * count_b) over the common keys (u64, wrapping) --
 * jaccard() and cosine() (lib.rs:708-765) are arithmetic on these and the tables' sizes.  Both tables must
 * live on the same device. */
KCT_API kct_status kct_compare(kct_table *a, kct_table *b, uint64_t *common_out, uint64_t *dot_out);

/* union (op 0), intersection (1), difference a - b (2), symmetric difference (3) of the two key sets
 * (lib.rs:610-655), in no particular order.  *n_out = size of the result; at most `cap` hashes are written. */
KCT_API kct_status kct_set_op(kct_table *a, kct_table *b, int op, uint64_t *hashes_out, size_t cap, uint64_t *n_out);

/* Deferred mode (ON by default; `on` = 0 switches it off).  The reference is called once per record
 * (README.md:96-98); one device pass per 150 bp read is launch-bound (~80 us, slower than the CPU loop it
 * replaces).  In deferred mode kct_consume(skip_bad != 0) only appends the record to a pinned host buffer and
 * returns; buffered records are counted in ONE device pass when the buffer (64 MiB) fills or when any other
 * call needs the table (get, len, dump, add, ...), so every read still observes every earlier write.  *n_out
 * then comes from a host-side scan for valid windows (the same all-ACGT rule the device applies); it differs
 * from the reference's n only if a window's true hash is 0 (probability 2^-64 per window).  Hashing and
 * counting still happen on the device only.  The same switch governs the device-side twin (kct_consume_device above).  Error mode
 * (skip_bad == 0) is never deferred: it counts what is buffered first, then runs synchronously.  A failure of the deferred pass itself (out of memory ...) is
 * reported by the call that triggers it. */
KCT_API kct_status kct_set_deferred(kct_table *t, int on);

/* save / load in the reference's wire format (lib.rs:269-322): serde_json of the struct, gzip level 1.
 * kct_save writes {"counts":{"<hash>":count,...}} followed by `tail_json`, the remaining members as text
 * starting with a comma -- e.g. ,"ksize":21,"version":"0.3.0","consumed":0,"store_kmers":false,"hash_to_kmer":null}
 * (they belong to the caller's struct).  The pairs are written in hash order, as one gzip member.
 * kct_load reads gzip or plain JSON, creates a table of the file's ksize on `device` holding its counts, and
 * keeps every member other than counts as one JSON object that kct_load_rest_json() returns (valid until the
 * calling thread loads again).  Malformed input: KCT_ERR_ARG with "Deserialization error: ..." . */
KCT_API kct_status kct_save(kct_table *t, const char *path, const char *tail_json);
KCT_API kct_status kct_load(const char *path, int device, kct_table **out);
KCT_API const char *kct_load_rest_json(void);

/* Bulk ingest keeps its working buffers with the table between calls (partition scratch ~ 8-16 B per window of the
 * largest pass, spill lists, shadow tables, pinned staging): several GB after a 10^8-window pass.  This gives them
 * back (after converting anything pending); the table itself is untouched and the next ingest allocates again. */
KCT_API kct_status kct_release_scratch(kct_table *t);

/* Flush point (SURVEY.md 8b "kct_sync"): counts whatever deferred mode has buffered and waits for the table's
 * stream.  Every other call already returns with its device work finished, so this matters only in deferred
 * mode or after work the caller queued on the table's stream itself. */
KCT_API kct_status kct_sync(kct_table *t);

/* Which device path bulk ingest uses: 0 = chosen per pass (default), 1 = direct path only (one
 * HBM atomic per k-mer), 2 = partitioned path whenever the table geometry allows (radix-partition
 * the hashes by 128-KiB table block, count each block in LDS), 3 = dedupe-first whenever k <= 64 and the
 * pass is large enough (k <= 21: compact 32-bit entries, k <= 32: 64-bit, 33..64: 128-bit mix128 pairs; count PACKED k-mers into a shadow table; their counts stay pending until something
 * reads the table, when every k-mer with a pending count is hashed once and added -- every reading call
 * converts first, so no call can observe the difference).  Mode 0 picks dedupe-first when what the table
 * already holds says that a pass repeats few k-mers many times.  The tables' contents are identical; in
 * dedupe-first passes *n_out counts a k-mer whose MurmurHash3 value is exactly 0 (which the reference
 * skips, lib.rs:589; probability 2^-64 per distinct k-mer) because that is only seen at conversion time. */
KCT_API kct_status kct_set_path(kct_table *t, int mode);

/* ---- streams and in-library kernel timing ----------------------------------------------------
 * The table owns a HIP stream; a caller that has its own (e.g. torch's current stream) can
 * hand it over as a `hipStream_t` cast to void*. */
KCT_API kct_status kct_set_stream(kct_table *t, void *hip_stream);
KCT_API void *kct_get_stream(kct_table *t);

/* When enabled, every kernel launch on this table is bracketed by HIP events on the table's
 * stream.  kct_profile_read(i) returns the i-th kernel name seen since the last reset with
 * its launch count and summed device time; KCT_ERR_ARG past the end. */
KCT_API kct_status kct_profile_enable(kct_table *t, int on);
KCT_API kct_status kct_profile_reset(kct_table *t);
KCT_API kct_status kct_profile_read(kct_table *t, int index, char *name_out, size_t name_cap, uint64_t *launches,
                            double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* KCT_H */
