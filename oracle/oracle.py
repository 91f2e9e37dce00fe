"""ctypes binding of oracle/kct_oracle.c (test infrastructure; see that file's header).

``OracleTable`` mirrors the slice of oxli's ``KmerCountTable`` that sits on the hot path
(reference src/lib.rs:41-194, 545-607, 778-837) so parity tests can drive the oracle and the
HIP engine with the same calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

ORC_OK, ORC_ERR_WRONG_KSIZE, ORC_ERR_INVALID_DNA, ORC_ERR_BAD_KMER, ORC_ERR_KSIZE_MISMATCH = 0, 1, 2, 3, 4

SEED_E = 7331  # default seed of the synthetic error model

__all__ = ["ShardSet", "SEED_E", "synth_reads_ex", "OracleTable", "murmur64", "hash_kmer", "seq_to_hashes", "synth_genome", "synth_reads",
           "baseline_consume", "sharded_consume", "build", "lib", "mix64"]


def build(native=False):
    """Compile the oracle with gcc (portable build by default, -march=native on request)."""
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, "libkct_oracle_native.so" if native else "libkct_oracle.so")


_libs = {}


def lib(native=False):
    if native not in _libs:
        path = os.path.join(_HERE, "libkct_oracle_native.so" if native else "libkct_oracle.so")
        src = os.path.join(_HERE, "kct_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build(native)
        L = C.CDLL(path)
        u64, u8p, u64p, sz = C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t
        vp = C.c_void_p
        L.orc_murmur64.restype = u64
        L.orc_murmur64.argtypes = [u8p, sz, u64]
        L.orc_mix64.restype = u64
        L.orc_mix64.argtypes = [u64]
        L.orc_seq_to_hashes.restype = sz
        L.orc_seq_to_hashes.argtypes = [u8p, sz, sz, C.c_int, vp, sz, C.POINTER(C.c_int)]
        L.orc_hash_kmer.restype = C.c_int
        L.orc_hash_kmer.argtypes = [u8p, sz, C.c_uint8, u64p]
        L.orc_new.restype = vp
        L.orc_new.argtypes = [C.c_uint8]
        L.orc_free.argtypes = [vp]
        L.orc_count_hash.restype = u64
        L.orc_count_hash.argtypes = [vp, u64]
        L.orc_get_hash.restype = u64
        L.orc_get_hash.argtypes = [vp, u64]
        L.orc_count.restype = C.c_int
        L.orc_count.argtypes = [vp, u8p, sz, u64p]
        L.orc_get.restype = C.c_int
        L.orc_get.argtypes = [vp, u8p, sz, u64p]
        L.orc_consume.restype = C.c_int
        L.orc_consume.argtypes = [vp, vp, sz, C.c_int, u64p]
        for name in ("orc_len", "orc_consumed", "orc_sum_counts"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [vp]
        L.orc_dump_sorted.restype = u64
        L.orc_dump_sorted.argtypes = [vp, vp, vp, u64]
        L.orc_add.restype = C.c_int
        L.orc_add.argtypes = [vp, vp, u64p, u64p]
        L.orc_add_pairs.argtypes = [vp, vp, vp, u64]
        L.orc_synth_genome.argtypes = [vp, u64, u64]
        L.orc_synth_reads.argtypes = [vp, vp, u64, u64, u64, C.c_uint32, u64]
        L.orc_baseline_consume.restype = vp
        L.orc_baseline_consume.argtypes = [vp, u64, C.c_uint32, C.c_uint8, C.c_int, u64p, C.POINTER(C.c_double)]
        L.orc_sharded_consume.restype = vp
        L.orc_sharded_consume.argtypes = [vp, u64, C.c_uint32, C.c_uint8, C.c_int, u64, u64p, C.POINTER(C.c_double)]
        L.orc_synth_reads_ex.argtypes = [vp, vp, u64, u64, u64, C.c_uint32, u64, C.c_uint32, C.c_uint32, u64, u64]
        L.orc_shardset_build.restype = vp
        L.orc_shardset_build.argtypes = [vp, vp, u64, u64, u64, C.c_uint32, C.c_uint8, u64, C.c_uint32, C.c_uint32, u64, u64, C.c_int, u64, u64]
        L.orc_shardset_free.argtypes = [vp]
        L.orc_shardset_digest.argtypes = [vp, vp]
        L.orc_shardset_get.restype = u64
        L.orc_shardset_get.argtypes = [vp, u64]
        L.orc_shardset_mismatches.restype = u64
        L.orc_shardset_mismatches.argtypes = [vp, vp, vp, u64, C.c_int]
        _libs[native] = L
    return _libs[native]


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def murmur64(data, seed=42):
    data = _b(data)
    return lib().orc_murmur64(data, len(data), seed)


def mix64(x):
    return lib().orc_mix64(x & 0xFFFFFFFFFFFFFFFF)


def hash_kmer(kmer, ksize=None):
    kmer = _b(kmer)
    out = C.c_uint64()
    st = lib().orc_hash_kmer(kmer, len(kmer), len(kmer) & 0xFF if ksize is None else ksize, C.byref(out))
    if st == ORC_ERR_WRONG_KSIZE:
        raise RuntimeError("wrong ksize")
    if st != ORC_OK:
        raise RuntimeError("invalid DNA")
    return out.value


def seq_to_hashes(seq, k, force=True):
    """Per-window hashes (0 for skipped windows).  Returns (np.uint64 array, stopped_on_bad)."""
    seq = _b(seq)
    cap = max(len(seq) - k + 1, 0)
    out = np.zeros(cap, dtype=np.uint64)
    err = C.c_int()
    n = lib().orc_seq_to_hashes(seq, len(seq), k, int(force), out.ctypes.data, cap, C.byref(err))
    return out[:n], bool(err.value)


def synth_genome(G, seed_g=42):
    g = np.empty(G, dtype=np.uint8)
    lib().orc_synth_genome(g.ctypes.data, G, seed_g)
    return g


def synth_reads(genome, first, count, L, seed_r=1337):
    """``count`` reads starting at stream index ``first``; returns uint8 [count, L+1] ('\\n' last)."""
    out = np.empty((count, L + 1), dtype=np.uint8)
    lib().orc_synth_reads(out.ctypes.data, genome.ctypes.data, len(genome), first, count, L, seed_r)
    return out


def synth_reads_ex(genome, first, count, L, seed_r=1337, sub_ppm=0, n_ppm=0, sorted_total=0, seed_e=SEED_E):
    """synth_reads with the error model of include/kct_synth.h (substitutions / N per million bases, position-sorted starts)."""
    out = np.empty((count, L + 1), dtype=np.uint8)
    lib().orc_synth_reads_ex(out.ctypes.data, genome.ctypes.data, len(genome), first, count, L, seed_r, sub_ppm, n_ppm, sorted_total, seed_e)
    return out


class ShardSet:
    """The CPU table of a whole configuration, key space sharded over ``threads`` owner tables (orc_shardset_*): exact
    digests and exact pair comparison for inputs of 10^8 .. 10^10 k-mers.  ``reads`` = uint8 [n, L+1] in memory, or None
    to generate reads [first, first + nreads) of the synthetic stream over ``genome`` on the fly.  ``expect_keys`` (optional)
    pre-sizes the owner tables: no re-hash while they grow."""

    FIELDS = ("len", "sum_counts", "sum_hc", "xor_hc", "min", "max", "sum_sq", "n", "consumed")

    def __init__(self, k, L, reads=None, genome=None, first=0, nreads=0, seed_r=1337, sub_ppm=0, n_ppm=0, sorted_total=0, seed_e=SEED_E,
                 threads=None, batch=262144, native=False, expect_keys=0):
        self._L = lib(native)
        threads = threads or max(1, min(64, len(os.sched_getaffinity(0))))  # (64 of 256 threads is the fastest on the GPU box's host)
        self.threads = threads
        if reads is not None:
            reads = np.ascontiguousarray(reads)
            assert reads.ndim == 2 and reads.shape[1] == L + 1
            nreads, rp = reads.shape[0], reads.ctypes.data
        else:
            rp = None
        gp = genome.ctypes.data if genome is not None else None
        self._keep = (reads, genome)
        self._h = self._L.orc_shardset_build(rp, gp, 0 if genome is None else len(genome), first, nreads, L, k, seed_r, sub_ppm, n_ppm,
                                             sorted_total, seed_e, threads, batch, int(expect_keys))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_shardset_free(self._h)
            self._h = None

    def digest(self):
        out = np.zeros(9, dtype=np.uint64)
        self._L.orc_shardset_digest(self._h, out.ctypes.data)
        return dict(zip(self.FIELDS, (int(v) for v in out)))

    def get_hash(self, h):
        return self._L.orc_shardset_get(self._h, h)

    def mismatches(self, keys, counts):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        assert keys.size == counts.size
        return self._L.orc_shardset_mismatches(self._h, keys.ctypes.data, counts.ctypes.data, keys.size, self.threads)


class OracleTable:
    """Hot-path slice of oxli.KmerCountTable over the C oracle."""

    def __init__(self, ksize, _handle=None, _lib=None):
        self._L = _lib or lib()
        self.ksize = ksize
        self._h = _handle if _handle is not None else self._L.orc_new(ksize)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_free(self._h)
            self._h = None

    def hash_kmer(self, kmer):
        return hash_kmer(kmer, self.ksize)

    def count_hash(self, h):
        return self._L.orc_count_hash(self._h, h)

    def get_hash(self, h):
        return self._L.orc_get_hash(self._h, h)

    def get_hash_array(self, hs):
        return [self.get_hash(h) for h in hs]

    def count(self, kmer):
        kmer = _b(kmer)
        out = C.c_uint64()
        st = self._L.orc_count(self._h, kmer, len(kmer), C.byref(out))
        if st == ORC_ERR_WRONG_KSIZE:
            raise ValueError("kmer size does not match count table ksize")
        if st != ORC_OK:
            raise RuntimeError("invalid DNA")
        return out.value

    def get(self, kmer):
        kmer = _b(kmer)
        out = C.c_uint64()
        st = self._L.orc_get(self._h, kmer, len(kmer), C.byref(out))
        if st == ORC_ERR_WRONG_KSIZE:
            raise ValueError("kmer size does not match count table ksize")
        if st != ORC_OK:
            raise RuntimeError("error hashing this k-mer")
        return out.value

    __getitem__ = get

    def consume(self, seq, skip_bad_kmers=True):
        if isinstance(seq, np.ndarray):
            ptr, n = seq.ctypes.data, seq.size
        else:
            seq = _b(seq)
            ptr, n = C.cast(C.c_char_p(seq), C.c_void_p), len(seq)
        out = C.c_uint64()
        st = self._L.orc_consume(self._h, ptr, n, int(skip_bad_kmers), C.byref(out))
        if st == ORC_ERR_BAD_KMER:
            raise ValueError(f"bad k-mer encountered at position {out.value}")
        return out.value

    def __len__(self):
        return self._L.orc_len(self._h)

    @property
    def consumed(self):
        return self._L.orc_consumed(self._h)

    @property
    def sum_counts(self):
        return self._L.orc_sum_counts(self._h)

    def dump_arrays(self):
        n = len(self)
        keys = np.empty(n, dtype=np.uint64)
        counts = np.empty(n, dtype=np.uint64)
        self._L.orc_dump_sorted(self._h, keys.ctypes.data, counts.ctypes.data, n)
        return keys, counts

    def dump(self, file=None, sortcounts=False, sortkeys=False):
        keys, counts = self.dump_arrays()
        pairs = list(zip(keys.tolist(), counts.tolist()))
        if sortcounts:
            pairs.sort(key=lambda p: (p[1], p[0]))
        return pairs

    def add(self, other):
        a, b = C.c_uint64(), C.c_uint64()
        st = self._L.orc_add(self._h, other._h, C.byref(a), C.byref(b))
        if st != ORC_OK:
            raise ValueError("KmerCountTables must have the same ksize")
        return a.value, b.value

    def add_pairs(self, keys, counts):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        self._L.orc_add_pairs(self._h, keys.ctypes.data, counts.ctypes.data, keys.size)


def baseline_consume(reads, L, k, threads=1, native=True):
    """Time the reference-shaped CPU path over ``reads`` (uint8 [n, L+1]).

    Returns (OracleTable merged, kmers, seconds)."""
    Lb = lib(native)
    reads = np.ascontiguousarray(reads)
    kmers, secs = C.c_uint64(), C.c_double()
    h = Lb.orc_baseline_consume(reads.ctypes.data, reads.shape[0], L, k, threads, C.byref(kmers), C.byref(secs))
    return OracleTable(k, _handle=h, _lib=Lb), kmers.value, secs.value


def sharded_consume(reads, L, k, threads, batch=262144, native=True):
    """The same per-record CPU work with the KEY SPACE sharded over ``threads`` (each thread owns a slice of hash space
    and a private table; nothing to merge).  Returns (OracleTable folded together after the clock stopped, kmers, seconds)."""
    Lb = lib(native)
    reads = np.ascontiguousarray(reads)
    kmers, secs = C.c_uint64(), C.c_double()
    h = Lb.orc_sharded_consume(reads.ctypes.data, reads.shape[0], L, k, threads, batch, C.byref(kmers), C.byref(secs))
    return OracleTable(k, _handle=h, _lib=Lb), kmers.value, secs.value
