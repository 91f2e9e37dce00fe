"""``KmerCountTable`` -- the reference's Python class (oxli ``src/lib.rs:29-42``) for the
count / consume / get path, bound to the MI355X engine through the C ABI in ``include/kct.h``.

Same method names, argument meaning, defaults, return values and exception types as the pyo3
class.  Every hash and every count is produced on the GPU; nothing in this module computes
them on the host.
"""
import ctypes as C

import math

import numpy as np

from . import _lib as L

__all__ = ["KmerCountTable", "VERSION", "PanicException"]


class PanicException(BaseException):
    """What a Rust panic surfaces as under pyo3 (``pyo3_runtime.PanicException``, a ``BaseException``): the reference's
    ``get`` panics on invalid DNA of the right length (lib.rs:176 ``.expect("error hashing this k-mer")``)."""

_COMP = str.maketrans("ACGT", "TGCA")

# The reference reports its crate version (lib.rs:27, Cargo.toml:3); the wire-compatible value
# for tables written by this engine.
VERSION = "0.3.0"


try:  # CPython call glue for per-record loops (csrc/pyfast.c); without it the same call goes through ctypes
    from . import _kctfast as _fast
    _fast.bind(C.cast(L.load().kct_consume, C.c_void_p).value, C.cast(L.load().kct_consume_will_defer, C.c_void_p).value)
    _fast_consume = _fast.consume
except ImportError:
    _fast = _fast_consume = None


def _bytes(s):
    if isinstance(s, str):
        return s.encode("utf-8")  # the reference sees the UTF-8 bytes of the str (lib.rs:548, 577)
    if isinstance(s, (bytes, bytearray, memoryview)):
        return bytes(s)
    raise TypeError(f"expected str or bytes, got {type(s).__name__}")


class KmerCountTable:
    """Counts canonical k-mers by their sourmash-compatible 64-bit hash, on one MI355X.

    ``KmerCountTable(ksize, store_kmers=False)`` -- reference ``lib.rs:44-62``.
    Extra keyword arguments (``capacity``, ``device``) size and place the device table; ``deferred`` (default: the
    library's, which is on) batches per-record ``consume`` calls into one device pass.  They do not change any result.
    """

    def __init__(self, ksize, store_kmers=False, *, capacity=0, device=0, deferred=None):
        if not 0 <= int(ksize) <= 255:
            raise OverflowError("out of range integral type conversion attempted")  # pyo3's u8 extraction
        self._lib = L.load()
        self._h = C.c_void_p()
        st = self._lib.kct_create(int(ksize), int(capacity), int(device), C.byref(self._h))
        if st != L.KCT_OK:
            self._h = None
            raise RuntimeError(f"kct_create failed ({st}): {L.last_error()}")
        self.ksize = int(ksize)
        self.version = VERSION
        if deferred is not None:
            self.set_deferred(bool(deferred))
        # lib.rs:37-38: optional hash -> canonical k-mer string map.  It is bookkeeping beside the counted
        # path and stays on the host, exactly like the reference's HashMap<u64, String>; the hashes that
        # key it still come from the device.
        self.store_kmers = bool(store_kmers)
        self._hash_to_kmer = {} if store_kmers else None

    @property
    def _hv(self):
        """The handle as a plain int, for _kctfast (derived from ``_h`` so that every construction path -- ``__init__``,
        ``load`` -- has it)."""
        return self._h.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.kct_destroy(h)
            self._h = None

    # ---- error mapping (reference exception types) ----------------------------------------------
    def _check(self, st):
        if st == L.KCT_OK:
            return
        msg = L.last_error()
        if st in (L.KCT_ERR_NOMEM,):
            raise MemoryError(msg)
        if st == L.KCT_ERR_BUSY:
            raise RuntimeError("Already borrowed")   # pyo3's message when a second thread enters a `&mut self` method (lib.rs:546)
        raise RuntimeError(f"libkct_hip error {st}: {msg}")

    # ---- hashing ----------------------------------------------------------------------------------
    def hash_kmer(self, kmer):
        """lib.rs:65-81.  RuntimeError("wrong ksize") / RuntimeError on invalid DNA (anyhow -> PyErr)."""
        b = _bytes(kmer)
        out = C.c_uint64()
        st = self._lib.kct_hash_kmer(self._h, b, len(b), C.byref(out))
        if st == L.KCT_ERR_WRONG_KSIZE:
            raise RuntimeError("wrong ksize")
        if st == L.KCT_ERR_INVALID_DNA:
            raise RuntimeError(f"invalid DNA character in input k-mer: {kmer}")
        self._check(st)
        return out.value

    def hash_windows(self, seq):
        """Hash of every k-window of ``seq`` as the consume loop sees them (lib.rs:576-600): a
        ``numpy.uint64`` array with 0 for windows holding a non-ACGT byte."""
        b = _bytes(seq)
        n = max(len(b) - self.ksize + 1, 0)
        out = np.zeros(n, dtype=np.uint64)
        nwin, fb = C.c_uint64(), C.c_uint64()
        self._check(self._lib.kct_hash_windows(self._h, b, len(b), out.ctypes.data, n, C.byref(nwin), C.byref(fb)))
        return out

    # ---- point updates / lookups ------------------------------------------------------------------
    def count_hash(self, hashval):
        """lib.rs:100-104: increment, return the new count."""
        out = C.c_uint64()
        self._check(self._lib.kct_count_hash(self._h, int(hashval), C.byref(out)))
        return out.value

    def count(self, kmer):
        """lib.rs:145-167: ValueError on wrong length; returns the new count; consumed += len."""
        b = _bytes(kmer)
        out = C.c_uint64()
        st = self._lib.kct_count(self._h, b, len(b), C.byref(out))
        if st == L.KCT_ERR_WRONG_KSIZE:
            raise ValueError("kmer size does not match count table ksize")
        if st == L.KCT_ERR_INVALID_DNA:
            raise RuntimeError(f"invalid DNA character in input k-mer: {kmer}")
        self._check(st)
        if self.store_kmers:  # lib.rs:155-163
            self._hash_to_kmer[self.hash_kmer(kmer)] = self.canon(kmer)
        return out.value

    def canon(self, kmer):
        """lib.rs:107-142: the lexicographically smaller of the upper-cased k-mer and its reverse complement."""
        if len(_bytes(kmer)) != self.ksize:
            raise ValueError("kmer size does not match count table ksize")
        up = kmer.upper() if isinstance(kmer, str) else _bytes(kmer).decode("utf-8", "replace").upper()
        if not all(c in "ATCG" for c in up):
            raise ValueError("kmer contains invalid characters")
        rc = up[::-1].translate(_COMP)
        return up if up <= rc else rc

    def unhash(self, hashval):
        """lib.rs:84-97."""
        if not self.store_kmers:
            raise ValueError("K-mer storage is not enabled.")
        try:
            return self._hash_to_kmer[int(hashval)]
        except KeyError:
            raise KeyError(f"Warning: Hash {hashval} not found in table.") from None

    def kmers_and_hashes(self, seq, skip_bad_kmers=True):
        """lib.rs:683-703 / 853-950: [(canonical k-mer, hash)] for every window of ``seq``.  Hashes come from
        the device; bad windows are reported on stderr and either skipped or returned as ("", 0)."""
        import sys
        b = _bytes(seq)
        up = b.decode("utf-8", "replace").upper() if not isinstance(seq, str) else seq.upper()
        if len(b) < self.ksize:
            return []  # the reference underflows `seq.len() - ksize + 1` here (lib.rs:872) and panics; an empty list is kinder
        hashes = self.hash_windows(b)
        out = []
        k = self.ksize
        for i, h in enumerate(hashes.tolist()):
            sub = up[i:i + k]
            if h:
                rc = sub[::-1].translate(_COMP)
                out.append((sub if sub < rc else rc, h))
            else:
                print(f"bad k-mer at position {i + 1}: {sub}", file=sys.stderr)
                if not skip_bad_kmers:
                    out.append(("", 0))
        return out

    def get(self, kmer):
        """lib.rs:170-182: ValueError on wrong length; 0 when absent."""
        b = _bytes(kmer)
        out = C.c_uint64()
        st = self._lib.kct_get(self._h, b, len(b), C.byref(out))
        if st == L.KCT_ERR_WRONG_KSIZE:
            raise ValueError("kmer size does not match count table ksize")
        if st == L.KCT_ERR_INVALID_DNA:
            # the reference panics here (lib.rs:176 `.expect`), surfacing pyo3's PanicException
            raise PanicException(f"error hashing this k-mer: invalid DNA character in input k-mer: {kmer}")
        self._check(st)
        return out.value

    def get_hash(self, hashval):
        """lib.rs:185-188."""
        out = C.c_uint64()
        self._check(self._lib.kct_get_hash(self._h, int(hashval), C.byref(out)))
        return out.value

    def get_hash_array(self, hash_keys):
        """lib.rs:191-194: counts in the order of ``hash_keys``."""
        keys = np.ascontiguousarray(np.asarray(list(hash_keys) if not isinstance(hash_keys, np.ndarray) else hash_keys,
                                               dtype=np.uint64))
        out = np.zeros(keys.size, dtype=np.uint64)
        self._check(self._lib.kct_get_hash_array(self._h, keys.ctypes.data, keys.size, out.ctypes.data))
        return out.tolist()

    def __getitem__(self, kmer):
        return self.get(kmer)  # lib.rs:670-672

    def __setitem__(self, kmer, count):
        h = self.hash_kmer(kmer)  # lib.rs:675-681
        self._check(self._lib.kct_set_hash(self._h, h, int(count)))

    # ---- bulk ingest ------------------------------------------------------------------------------
    def consume(self, seq, skip_bad_kmers=True):
        """lib.rs:545-607.  Returns the number of k-mers counted.  With ``skip_bad_kmers=False`` a
        window holding a non-ACGT byte raises ``ValueError("bad k-mer encountered at position n")``
        after the k-mers before it were counted, leaving ``consumed`` unchanged."""
        if _fast_consume is not None and not self.store_kmers:
            r = _fast_consume(self._hv, seq, skip_bad_kmers)  # n, or (status, n), or None for a seq that is neither str nor bytes
            if r.__class__ is int:
                return r
            if r is not None:
                if r[0] == L.KCT_ERR_BAD_KMER:
                    raise ValueError(f"bad k-mer encountered at position {r[1]}")
                self._check(r[0])
        b = _bytes(seq)
        if self.store_kmers:
            # lib.rs:552-573: this branch walks KmersAndHashesIter, which always skips bad windows
            # (force=true) -- so it never raises, whatever skip_bad_kmers says -- and records
            # hash -> canonical k-mer for every counted window.
            for kmer, h in self.kmers_and_hashes(b, skip_bad_kmers):
                if h:
                    self._hash_to_kmer[h] = kmer
            skip_bad_kmers = True
        out = C.c_uint64()
        st = self._lib.kct_consume(self._h, b, len(b), 1 if skip_bad_kmers else 0, C.byref(out))
        if st == L.KCT_ERR_BAD_KMER:
            raise ValueError(f"bad k-mer encountered at position {out.value}")
        self._check(st)
        return out.value

    def consume_batch(self, seqs, skip_bad_kmers=True):
        """The README loop ``for record in ...: kct.consume(record.sequence)`` (README.md:96-98) as
        one device pass.  ``seqs`` is an iterable of str/bytes, or a pair ``(bytes, offsets)`` in CSR
        form.  Returns the total number of k-mers counted; raises like ``consume`` would on the
        first offending record, with everything before it counted."""
        if isinstance(seqs, tuple) and len(seqs) == 2 and not isinstance(seqs[0], str):
            data = np.frombuffer(_bytes(seqs[0]), dtype=np.uint8) if not isinstance(seqs[0], np.ndarray) else seqs[0]
            offsets = np.ascontiguousarray(seqs[1], dtype=np.uint64)
        else:
            if not isinstance(seqs, (list, tuple)):
                seqs = list(seqs)
            packed = _fast.csr(seqs) if _fast is not None else None  # (None: an item that is neither str nor bytes)
            if packed is not None:
                data, offsets = np.frombuffer(packed[0], dtype=np.uint8), np.frombuffer(packed[1], dtype=np.uint64)
            else:
                parts = [_bytes(s) for s in seqs]
                offsets = np.zeros(len(parts) + 1, dtype=np.uint64)
                if parts:
                    offsets[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
                data = np.frombuffer(b"".join(parts), dtype=np.uint8)
        data = np.ascontiguousarray(data)
        nrec = offsets.size - 1 if offsets.size else 0
        n, bad_rec, bad_pos = C.c_uint64(), C.c_uint64(), C.c_uint64()
        st = self._lib.kct_consume_batch(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data, nrec,
                                         1 if skip_bad_kmers else 0, C.byref(n), C.byref(bad_rec), C.byref(bad_pos))
        if st == L.KCT_ERR_BAD_KMER:
            err = ValueError(f"bad k-mer encountered at position {bad_pos.value}")
            err.record = bad_rec.value
            err.counted = n.value
            raise err
        self._check(st)
        return n.value

    def batch_timeline(self):
        """Where the last large ``consume_batch`` call spent its time (``kct_batch_timeline``): a dict of milliseconds and counts."""
        out = (C.c_double * 16)()
        self._check(self._lib.kct_batch_timeline(self._h, out))
        names = ("checked_ms", "cut_ms", "first_packer_start_ms", "last_packer_end_ms", "last_h2d_enqueued_ms", "submitted_ms", "threads", "threads_busy_ms_sum",
                 "thread_busy_ms_max", "source_bytes", "packed_bytes", "minor_faults", "cpus", "numa_nodes", "caller_cpu", "pinned")
        return dict(zip(names, (float(v) for v in out)))

    def consume_file(self, path, skip_bad_kmers=True):
        """``for record in screed.open(path): kct.consume(record.sequence)`` (README.md:89-99) for a FASTA
        or FASTQ file, plain or gzip.  Returns the total number of k-mers counted.  With
        ``skip_bad_kmers=False`` the records are fed through ``consume_batch`` so the first bad k-mer
        raises exactly as the per-record loop would."""
        if not skip_bad_kmers:
            from .io import read_records
            total, batch, size = 0, [], 0
            for _, seq in read_records(path):
                batch.append(seq)
                size += len(seq)
                if size >= (64 << 20):
                    total += self.consume_batch(batch, skip_bad_kmers=False)
                    batch, size = [], 0
            return total + (self.consume_batch(batch, skip_bad_kmers=False) if batch else 0)
        n, nrec, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
        st = self._lib.kct_consume_file(self._h, str(path).encode(), 1, C.byref(n), C.byref(nrec), C.byref(nb))
        if st == L.KCT_ERR_ARG and "cannot open" in L.last_error():
            raise OSError(L.last_error())
        self._check(st)
        self.last_file_records = nrec.value
        return n.value

    def consume_device(self, data_ptr, nbytes, consumed_bytes):
        """Counts a record stream that already sits in HBM (records separated by a non-ACGT byte).
        ``data_ptr`` is a raw device address, e.g. ``tensor.data_ptr()``."""
        n = C.c_uint64()
        self._check(self._lib.kct_consume_device(self._h, C.c_void_p(int(data_ptr)), int(nbytes), int(consumed_bytes), C.byref(n)))
        return n.value

    def consume_device_packed(self, codes_ptr, valid_ptr, nbases, consumed_bytes):
        """Counts a PACKED record stream in HBM: ``codes`` (uint32 per 16 bases) and ``valid`` (uint16 per 16 bases), see
        include/kct.h.  Raw device addresses."""
        n = C.c_uint64()
        self._check(self._lib.kct_consume_device_packed(self._h, C.c_void_p(int(codes_ptr)), C.c_void_p(int(valid_ptr)), int(nbases),
                                                        int(consumed_bytes), C.byref(n)))
        return n.value

    def set_packed_upload(self, on=True):
        """Large skip-bad batches of ``consume_batch`` are packed to 2 bits + 1 validity bit per base on the host before the
        upload (``kct_set_packed_upload``; on by default).  Results do not depend on it."""
        self._check(self._lib.kct_set_packed_upload(self._h, 1 if on else 0))

    # ---- attributes ---------------------------------------------------------------------------------
    def __len__(self):
        out = C.c_uint64()
        self._check(self._lib.kct_len(self._h, C.byref(out)))
        return out.value

    @property
    def consumed(self):
        out = C.c_uint64()
        self._check(self._lib.kct_consumed(self._h, C.byref(out)))
        return out.value

    @property
    def sum_counts(self):
        out = C.c_uint64()
        self._check(self._lib.kct_sum_counts(self._h, C.byref(out)))
        return out.value

    @property
    def capacity(self):
        out = C.c_uint64()
        self._check(self._lib.kct_capacity(self._h, C.byref(out)))
        return out.value

    @property
    def hashes(self):
        """lib.rs:516-519: the keys (order unspecified, as in the reference)."""
        return self.dump_arrays(0)[0].tolist()

    # ---- dump / merge -------------------------------------------------------------------------------
    def dump_arrays(self, order=1):
        """(hashes, counts) as numpy arrays; order 0 = unspecified, 1 = by hash, 2 = by (count, hash)."""
        n = len(self)
        keys = np.zeros(n, dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint64)
        got = C.c_uint64()
        self._check(self._lib.kct_dump(self._h, keys.ctypes.data, counts.ctypes.data, n, order, C.byref(got)))
        return keys, counts

    def dump(self, file=None, sortcounts=False, sortkeys=False):
        """lib.rs:330-381."""
        if sortcounts and sortkeys:
            raise ValueError("Cannot sort by both counts and keys at the same time.")
        # (no sort option: the reference returns its map's iteration order, the same order `list(table)` has --
        # test_dump.py:38-49; here that order is by hash)
        keys, counts = self.dump_arrays(2 if sortcounts else 1)
        if file is not None:
            with open(file, "w") as f:  # OSError on a bad path, as File::create (lib.rs:362)
                f.write("".join(f"{h}\t{c}\n" for h, c in zip(keys.tolist(), counts.tolist())))
            return []
        return list(zip(keys.tolist(), counts.tolist()))

    def add(self, other):
        """lib.rs:778-837: per-key sum; returns (total_counts_added, new_keys_added)."""
        if not isinstance(other, KmerCountTable):
            raise TypeError("argument 'other': expected KmerCountTable")
        a, b = C.c_uint64(), C.c_uint64()
        st = self._lib.kct_add(self._h, other._h, C.byref(a), C.byref(b))
        if st == L.KCT_ERR_KSIZE_MISMATCH:
            raise ValueError("KmerCountTables must have the same ksize")
        self._check(st)
        if self.store_kmers:  # lib.rs:810-828
            if other.store_kmers:
                for h, kmer in other._hash_to_kmer.items():
                    self._hash_to_kmer.setdefault(h, kmer)
            else:
                import sys
                print("Warning: Incoming table does not store k-mers, but target table does. "
                      "K-mer information for new hashes will be missing.", file=sys.stderr)
        print(f"Added {a.value} k-mer counts to the table")  # lib.rs:833-834
        print(f"Added {b.value} new keys to the table")
        return a.value, b.value

    def clear(self):
        self._check(self._lib.kct_clear(self._h))
        if self._hash_to_kmer is not None:
            self._hash_to_kmer = {}

    # ---- table analytics: scans, reductions and table-against-table lookups on the resident table ---------
    # (kct_analytics.hip).  No hash or count is ever produced on the host.
    def __iter__(self):
        """lib.rs:658-662: (hash, count) pairs.  The reference's order is HashMap order; here by hash."""
        keys, counts = self.dump_arrays(1)
        return iter(zip(keys.tolist(), counts.tolist()))

    def _count_stats(self):
        lo, hi, sq = C.c_uint64(), C.c_uint64(), C.c_double()
        self._check(self._lib.kct_count_stats(self._h, C.byref(lo), C.byref(hi), C.byref(sq)))
        return lo.value, hi.value, sq.value

    def digest(self):
        """(sum of hash * count, xor of hash * count, sum of count^2), all mod 2^64, from one device scan (``kct_digest``)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.kct_digest(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    @property
    def min(self):
        """lib.rs:492-501 (0 for an empty table)."""
        return self._count_stats()[0]

    @property
    def max(self):
        """lib.rs:505-514."""
        return self._count_stats()[1]

    def histo(self, zero=True):
        """lib.rs:464-488: [(frequency, number of k-mers with that count)]."""
        cap = 1 << 16
        while True:
            vals, freq, n = np.empty(cap, dtype=np.uint64), np.empty(cap, dtype=np.uint64), C.c_uint64()
            self._check(self._lib.kct_histogram(self._h, vals.ctypes.data, freq.ctypes.data, cap, C.byref(n)))
            if n.value <= cap:
                break
            cap = n.value
        observed = dict(zip(vals[: n.value].tolist(), freq[: n.value].tolist()))
        if zero:
            return [(f, observed.get(f, 0)) for f in range(0, (max(observed) if observed else 0) + 1)]
        return sorted(observed.items())

    def dump_kmers(self, file=None, sortcounts=False, sortkeys=False):
        """lib.rs:389-456."""
        if not self.store_kmers:
            raise ValueError("K-mer storage is disabled. No hash:kmer map is available.")
        if sortcounts and sortkeys:
            raise ValueError("Cannot sort by both counts and kmers at the same time.")
        keys, counts = self.dump_arrays(0)
        have = dict(zip(keys.tolist(), counts.tolist()))
        pairs = [(kmer, have[h]) for h, kmer in self._hash_to_kmer.items() if h in have]
        if sortkeys:
            pairs.sort(key=lambda p: p[0])
        elif sortcounts:
            pairs.sort(key=lambda p: (p[1], p[0]))
        if file is not None:
            with open(file, "w") as f:
                f.write("".join(f"{k}\t{c}\n" for k, c in pairs))
            return []
        return pairs

    # removal happens on the device: the survivors are compacted and re-inserted (kct_retain_counts / kct_remove_hash)
    def drop_hash(self, hashval):
        """lib.rs:213-224."""
        self._check(self._lib.kct_remove_hash(self._h, int(hashval), None))

    def drop(self, kmer):
        """lib.rs:197-210."""
        self.drop_hash(self.hash_kmer(kmer))

    def mincut(self, min_count):
        """lib.rs:227-246: remove k-mers with count < min_count; returns how many were removed."""
        removed = C.c_uint64()
        self._check(self._lib.kct_retain_counts(self._h, int(min_count), 2 ** 64 - 1, C.byref(removed)))
        return removed.value

    def maxcut(self, max_count):
        """lib.rs:249-267: remove k-mers with count > max_count."""
        removed = C.c_uint64()
        self._check(self._lib.kct_retain_counts(self._h, 0, int(max_count), C.byref(removed)))
        return removed.value

    def _set_op(self, other, op):
        if not isinstance(other, KmerCountTable):
            raise TypeError("argument 'other': expected KmerCountTable")
        cap = len(self) if op in (1, 2) else len(self) + len(other)
        out, n = np.empty(max(cap, 1), dtype=np.uint64), C.c_uint64()
        self._check(self._lib.kct_set_op(self._h, other._h, op, out.ctypes.data, cap, C.byref(n)))
        return set(out[: n.value].tolist())

    def union(self, other):
        return self._set_op(other, 0)  # lib.rs:615-617

    def intersection(self, other):
        return self._set_op(other, 1)  # lib.rs:619-624

    def difference(self, other):
        return self._set_op(other, 2)  # lib.rs:626-631

    def symmetric_difference(self, other):
        return self._set_op(other, 3)  # lib.rs:633-638

    __or__, __and__, __sub__, __xor__ = union, intersection, difference, symmetric_difference

    def _compare(self, other):
        common, dot = C.c_uint64(), C.c_uint64()
        self._check(self._lib.kct_compare(self._h, other._h, C.byref(common), C.byref(dot)))
        return common.value, dot.value

    def jaccard(self, other):
        """lib.rs:708-722 (two empty tables are identical: 1.0)."""
        inter = self._compare(other)[0]
        uni = len(self) + len(other) - inter
        return 1.0 if uni == 0 else inter / uni

    def cosine(self, other):
        """lib.rs:727-765: u64 dot product over the common keys, f64 magnitudes."""
        if len(self) == 0 or len(other) == 0:
            return 0.0
        dot = self._compare(other)[1]
        ma, mb = math.sqrt(self._count_stats()[2]), math.sqrt(other._count_stats()[2])
        if ma == 0.0 or mb == 0.0:
            return 0.0
        return float(dot) / (ma * mb)

    def _tail_json(self):
        """The members of the reference's struct other than counts, as serde_json writes them (lib.rs:32-39)."""
        import json
        h2k = "null" if self._hash_to_kmer is None else \
            "{" + ",".join(f'"{h}":{json.dumps(kmer)}' for h, kmer in self._hash_to_kmer.items()) + "}"
        return (',"ksize":' + str(self.ksize) + ',"version":' + json.dumps(self.version) +
                ',"consumed":' + str(self.consumed) + ',"store_kmers":' + ("true" if self.store_kmers else "false") +
                ',"hash_to_kmer":' + h2k + "}")

    def serialize_json(self):
        """lib.rs:269-272: the serde_json image of the struct (counts in hash order)."""
        keys, counts = self.dump_arrays(1)
        body = ",".join(f'"{h}":{c}' for h, c in zip(keys.tolist(), counts.tolist()))
        return '{"counts":{' + body + "}" + self._tail_json()

    def save(self, filepath):
        """lib.rs:274-292: gzip level 1 of ``serialize_json()``; text and deflate are produced natively by
        several threads (``kct_save``).  OSError on a bad path, like ``File::create``."""
        st = self._lib.kct_save(self._h, str(filepath).encode(), self._tail_json().encode())
        if st == L.KCT_ERR_ARG and ("cannot create" in L.last_error() or "writing" in L.last_error()):
            raise OSError(L.last_error())
        self._check(st)

    @staticmethod
    def load(filepath, *, device=0):
        """lib.rs:295-322: accepts gzip or plain JSON (niffler sniffs the format); warns on a version mismatch.
        The counts are parsed and merged natively (``kct_load``); the scalar members come back as JSON text."""
        import json
        import os
        import sys
        lib = L.load()
        if not os.path.exists(filepath):
            # File::open fails inside a function returning anyhow::Result (lib.rs:296-299): pyo3 turns that into RuntimeError
            raise RuntimeError("No such file or directory (os error 2)")
        h = C.c_void_p()
        st = lib.kct_load(str(filepath).encode(), int(device), C.byref(h))
        if st != L.KCT_OK:
            msg = L.last_error()
            raise RuntimeError(msg if msg.startswith("Deserialization error") else f"Deserialization error: {msg}")
        rest = lib.kct_load_rest_json()
        t = KmerCountTable.__new__(KmerCountTable)
        t._lib, t._h = lib, h
        try:
            d = json.loads(rest.decode("utf-8"))
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f"Deserialization error: {e}") from None
        t.ksize = int(d["ksize"])
        t.store_kmers = bool(d.get("store_kmers", False))
        t._hash_to_kmer = {} if t.store_kmers else None
        t._check(lib.kct_add_consumed(h, int(d.get("consumed", 0))))
        if t.store_kmers and d.get("hash_to_kmer"):
            t._hash_to_kmer = {int(k): kmer for k, kmer in d["hash_to_kmer"].items()}
        t.version = d.get("version", VERSION)
        if t.version != VERSION:
            print(f"Version mismatch: loaded version is {t.version}, but current version is {VERSION}", file=sys.stderr)
        return t

    def resize(self, distinct):
        """Capacity for ``distinct`` keys (never less than the table holds), smaller than now if that is enough
        (``kct_resize``)."""
        self._check(self._lib.kct_resize(self._h, int(distinct)))

    def release_scratch(self):
        """Gives back the working buffers bulk ingest keeps between calls (``kct_release_scratch``); the table stays."""
        self._check(self._lib.kct_release_scratch(self._h))

    def sync(self):
        """Counts whatever deferred mode has buffered and waits for the table's stream (``kct_sync``)."""
        self._check(self._lib.kct_sync(self._h))

    def set_deferred(self, on=True):
        """Per-record ``consume()`` calls are buffered on the host and counted in one device pass when the buffer
        fills or the table is read (see ``kct_set_deferred`` in include/kct.h).  On by default."""
        self._check(self._lib.kct_set_deferred(self._h, 1 if on else 0))

    def set_path(self, mode):
        """0 = choose per pass, 1 = direct atomic path only, 2 = partitioned path whenever possible,
        3 = dedupe-first path (k <= 64) whenever the pass is large enough."""
        self._check(self._lib.kct_set_path(self._h, {"auto": 0, "direct": 1, "partitioned": 2, "dedupe": 3}.get(mode, mode)))

    # ---- in-library kernel timing (bench.py) ----------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._check(self._lib.kct_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def profile(self, on=True):
        self._check(self._lib.kct_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.kct_profile_reset(self._h))

    def profile_read(self):
        """{kernel name: (launches, total_ms)} since the last reset."""
        out = {}
        i = 0
        name = C.create_string_buffer(128)
        launches, ms = C.c_uint64(), C.c_double()
        while self._lib.kct_profile_read(self._h, i, name, 128, C.byref(launches), C.byref(ms)) == L.KCT_OK:
            out[name.value.decode()] = (launches.value, ms.value)
            i += 1
        return out
