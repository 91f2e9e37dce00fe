"""``KmerCountTable`` -- the reference's Python class (oxli ``src/lib.rs:29-42``) for the
count / consume / get path, bound to the MI355X engine through the C ABI in ``include/kct.h``.

Same method names, argument meaning, defaults, return values and exception types as the pyo3
class.  Every hash and every count is produced on the GPU; nothing in this module computes
them on the host.
"""
import ctypes as C

import numpy as np

from . import _lib as L

__all__ = ["KmerCountTable", "VERSION"]

# The reference reports its crate version (lib.rs:27, Cargo.toml:3); the wire-compatible value
# for tables written by this engine.
VERSION = "0.3.0"


def _bytes(s):
    if isinstance(s, str):
        return s.encode("utf-8")  # the reference sees the UTF-8 bytes of the str (lib.rs:548, 577)
    if isinstance(s, (bytes, bytearray, memoryview)):
        return bytes(s)
    raise TypeError(f"expected str or bytes, got {type(s).__name__}")


class KmerCountTable:
    """Counts canonical k-mers by their sourmash-compatible 64-bit hash, on one MI355X.

    ``KmerCountTable(ksize, store_kmers=False)`` -- reference ``lib.rs:44-62``.
    Extra keyword arguments (``capacity``, ``device``) size and place the device table; they do
    not change any result.
    """

    def __init__(self, ksize, store_kmers=False, *, capacity=0, device=0):
        if not 0 <= int(ksize) <= 255:
            raise OverflowError("out of range integral type conversion attempted")  # pyo3's u8 extraction
        if store_kmers:
            raise NotImplementedError(
                "store_kmers=True (hash -> k-mer string map, lib.rs:552-573) is outside the accelerated "
                "consume path of this build")
        self._lib = L.load()
        self._h = C.c_void_p()
        st = self._lib.kct_create(int(ksize), int(capacity), int(device), C.byref(self._h))
        if st != L.KCT_OK:
            self._h = None
            raise RuntimeError(f"kct_create failed ({st}): {L.last_error()}")
        self.ksize = int(ksize)
        self.version = VERSION
        self.store_kmers = False

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
        return out.value

    def get(self, kmer):
        """lib.rs:170-182: ValueError on wrong length; 0 when absent."""
        b = _bytes(kmer)
        out = C.c_uint64()
        st = self._lib.kct_get(self._h, b, len(b), C.byref(out))
        if st == L.KCT_ERR_WRONG_KSIZE:
            raise ValueError("kmer size does not match count table ksize")
        if st == L.KCT_ERR_INVALID_DNA:
            # the reference panics here (lib.rs:176 `.expect`), surfacing pyo3's PanicException
            raise RuntimeError("error hashing this k-mer")
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
        b = _bytes(seq)
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

    def consume_device(self, data_ptr, nbytes, consumed_bytes):
        """Counts a record stream that already sits in HBM (records separated by a non-ACGT byte).
        ``data_ptr`` is a raw device address, e.g. ``tensor.data_ptr()``."""
        n = C.c_uint64()
        self._check(self._lib.kct_consume_device(self._h, C.c_void_p(int(data_ptr)), int(nbytes), int(consumed_bytes), C.byref(n)))
        return n.value

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
        keys, counts = self.dump_arrays(1 if sortkeys else 2 if sortcounts else 0)
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
        print(f"Added {a.value} k-mer counts to the table")  # lib.rs:833-834
        print(f"Added {b.value} new keys to the table")
        return a.value, b.value

    def clear(self):
        self._check(self._lib.kct_clear(self._h))

    def set_path(self, mode):
        """0 = choose per pass, 1 = direct atomic path only, 2 = partitioned path whenever possible."""
        self._check(self._lib.kct_set_path(self._h, {"auto": 0, "direct": 1, "partitioned": 2}.get(mode, mode)))

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
