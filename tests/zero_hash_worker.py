"""Worker of tests/test_gpu_zero_hash.py, run with KCT_LIB_PATH = oxli_amd/csrc/libkct_zero.so (`make zero`): a build in which the
21-mer ZERO_KMER hashes to 0 wherever the DEVICE hashes a k-mer.  The reference's consume skips a window whose hash is 0 -- the
window is neither counted nor tallied in n (lib.rs:589 `Ok(0) => continue`, before `n += 1`).  Real MurmurHash3 values reach that
branch with probability 2^-64, so this is the only way to run it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ZERO_KMER = "ACCGTTAGGCATTCGATCGGA"          # (the Makefile's `zero` target holds its packed canonical form)
K, L = 21, 150


def packed_canonical(kmer):
    rc = kmer[::-1].translate(str.maketrans("ACGT", "TGCA"))
    v = 0
    for c in min(kmer, rc):
        v = (v << 2) | "ACGT".index(c)
    return v << (64 - 2 * len(kmer))


def main():
    import re

    import torch

    import oracle
    from oxli_amd import KmerCountTable, _lib
    assert _lib.LIB_PATH.endswith("libkct_zero.so"), _lib.LIB_PATH
    mk = open(os.path.join(ROOT, "oxli_amd", "csrc", "Makefile")).read()
    assert int(re.search(r"-DKCT_DEBUG_ZERO_KMER=(0x[0-9a-f]+)ULL", mk).group(1), 16) == packed_canonical(ZERO_KMER)

    genome = oracle.synth_genome(200_000, 42)
    reads = oracle.synth_reads(genome, 0, 40_000, L, 1337)      # 5.2 M window starts: enough for a dedupe-first pass
    rc = ZERO_KMER[::-1].translate(str.maketrans("ACGT", "TGCA"))
    plant = np.frombuffer(ZERO_KMER.encode(), dtype=np.uint8)
    plant_rc = np.frombuffer(rc.encode(), dtype=np.uint8)
    for i in range(0, 40_000, 97):                                # both strands, at varying offsets, some twice in a read
        off = (i * 7) % (L - K + 1)
        reads[i, off:off + K] = plant if (i // 97) % 2 == 0 else plant_rc
        if i % 5 == 0 and off + 2 * K < L:
            reads[i, off + K:off + 2 * K] = plant
    ref = oracle.OracleTable(K)
    tab, n_real, _ = oracle.baseline_consume(reads, L, K, min(8, len(os.sched_getaffinity(0))), native=False)
    rk, rcnt = tab.dump_arrays()
    h0 = ref.hash_kmer(ZERO_KMER)
    occ = int(rcnt[np.searchsorted(rk, h0)])
    assert rk[np.searchsorted(rk, h0)] == h0 and occ >= 400, occ
    keep = rk != h0
    want_k, want_c = rk[keep], rcnt[keep]                          # the reference's table if that k-mer hashed to 0: it is not in it
    n_want = n_real - occ                                          # ... and its windows are not in n
    dev_reads = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda()
    recs = [bytes(r[:L]) for r in reads]

    def check(t, n, n_expected, what):
        dk, dc = t.dump_arrays(1)
        assert np.array_equal(dk, want_k) and np.array_equal(dc, want_c), f"{what}: table differs from the reference's"
        assert t.get_hash(0) == 0 and t.get(ZERO_KMER) == 0 and len(t) == want_k.size and t.sum_counts == n_want, what
        assert n == n_expected, (what, n, n_expected, n_real, occ)

    # paths that hash every window as it is read: n is the reference's n
    for path in ("direct", "partitioned"):
        t = KmerCountTable(K, capacity=400_000)
        t.set_path(path)
        t.set_deferred(False)
        check(t, t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), len(recs) * L), n_want, path)
    # one record at a time, error mode and not deferred: the reference's own loop
    t = KmerCountTable(K, capacity=400_000)
    t.set_deferred(False)
    n = sum(t.consume(r.decode(), skip_bad_kmers=False) for r in recs[:2000])
    occ2k = sum(r.count(ZERO_KMER.encode()) + r.count(rc.encode()) for r in recs[:2000])
    assert occ2k > 0 and n == 2000 * (L - K + 1) - occ2k and t.get(ZERO_KMER) == 0 and t.sum_counts == n
    # dedupe-first passes and deferred calls learn the hash only at conversion: the TABLE is the reference's, n includes the
    # windows of that k-mer (include/kct.h: kct_set_path, kct_set_deferred, kct_consume_device -- the documented difference)
    t = KmerCountTable(K, capacity=400_000)
    t.set_path("dedupe")
    t.set_deferred(False)
    check(t, t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), len(recs) * L), n_real, "dedupe-first (compact)")
    t = KmerCountTable(K, capacity=400_000)      # deferred per-record calls (the default mode)
    n = sum(t.consume(r.decode()) for r in recs)
    check(t, n, n_real, "deferred per-record consume")
    t = KmerCountTable(K, capacity=1 << 26)      # a call that is small for its table is staged on the device and counted later
    n = t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), len(recs) * L)
    check(t, n, n_real, "staged device call")
    print(f"ZERO_HASH_OK occurrences={occ} n_reference={n_want} n_dedupe={n_real}")


if __name__ == "__main__":
    main()
