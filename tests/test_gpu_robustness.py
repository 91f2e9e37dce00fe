"""Robustness of the device table around the hot path (``-m gpu``): cases a reviewer found that the happy-path
parity tests did not reach.  Every expected value comes from the oracle or from plain dict arithmetic.

* error mode (``skip_bad_kmers=False``) on a record whose valid prefix spans several launch chunks and makes the
  auto-sized table grow while the prefix is being counted (reference behaviour: ``lib.rs:593-596``);
* keys that are not MurmurHash3 output and share their low bits (``count_hash`` / ``__setitem__`` / ``add`` accept
  any u64, like the reference's ``HashMap<u64, u64>``, ``lib.rs:100-104, 675-681``);
* ``drop_hash`` in place (``lib.rs:213-224``), including keys displaced from their home group;
* ``add`` between two tables on one GPU without a host round trip (``lib.rs:778-837``).
"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
from oracle import OracleTable  # noqa: E402


@pytest.fixture(scope="module")
def KCT():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from oxli_amd import KmerCountTable
    return KmerCountTable


def assert_same_table(dev, ref):
    dk, dc = dev.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk), "hash sets differ"
    assert np.array_equal(dc, rc), "counts differ"
    assert len(dev) == len(ref) and dev.sum_counts == ref.sum_counts and dev.consumed == ref.consumed


@pytest.mark.parametrize("k", [21, 31])
def test_error_mode_long_prefix_in_an_auto_sized_table(KCT, k):
    """One 3 Mbp record with an 'N' near its end, default (65,536-slot) table, skip_bad_kmers=False: the ~3 M k-mers in
    front of the N must be counted exactly (the table grows several times and replays spill lists meanwhile), the
    call raises with the reference's message, and ``consumed`` covers only the records before the bad one."""
    rng = np.random.default_rng(77 + k)
    genome = rng.integers(0, 4, 3_000_000)
    rec = "".join("ACGT"[i] for i in genome)
    bad_at = len(rec) - 1000
    rec_bad = rec[:bad_at] + "N" + rec[bad_at + 1:]
    first = rec[:5000]
    ref = OracleTable(k)
    ref.consume(first)
    with pytest.raises(ValueError) as e_ref:
        ref.consume(rec_bad, skip_bad_kmers=False)
    dev = KCT(k)  # no capacity hint
    with pytest.raises(ValueError) as e_dev:
        dev.consume_batch([first, rec_bad, "ACGT" * 100], skip_bad_kmers=False)
    assert str(e_dev.value) == str(e_ref.value) == f"bad k-mer encountered at position {bad_at - k + 1}"
    assert e_dev.value.record == 1
    assert_same_table(dev, ref)
    # the single-record entry point takes the same route
    dev1, ref1 = KCT(k), OracleTable(k)
    with pytest.raises(ValueError):
        dev1.consume(rec_bad, skip_bad_kmers=False)
    with pytest.raises(ValueError):
        ref1.consume(rec_bad, skip_bad_kmers=False)
    assert_same_table(dev1, ref1)


def test_keys_that_share_their_low_bits(KCT):
    """300 keys i << 40 have the same home group in every table of up to 2^40 slots.  The reference's HashMap takes
    them; here they must fill their block by whole-block probing, without the table growing for each of them."""
    t = KCT(21)
    want = {}
    for i in range(1, 301):
        h = i << 40
        assert t.count_hash(h) == 1
        want[h] = 1
    for i in range(1, 301, 7):
        h = i << 40
        assert t.count_hash(h) == 2
        want[h] = 2
    assert t.capacity <= 1 << 17, t.capacity          # did not double per colliding key
    assert len(t) == 300 and t.sum_counts == sum(want.values())
    keys = sorted(want)
    assert t.get_hash_array(keys) == [want[h] for h in keys]
    assert t.get_hash(301 << 40) == 0
    assert dict(t.dump()) == want
    # the same keys arriving as one merge (add of another table) and surviving a re-hash
    other = KCT(21)
    assert other.add(t) == (sum(want.values()), 300)
    other._check(other._lib.kct_reserve(other._h, 1_000_000))  # re-hash into a larger table
    assert dict(other.dump()) == want
    # in-place removal of displaced keys keeps every other key reachable
    rng = random.Random(3)
    gone = set(rng.sample(keys, 120))
    for h in gone:
        other.drop_hash(h)
    other.drop_hash(12345)  # absent: no effect
    left = {h: c for h, c in want.items() if h not in gone}
    assert len(other) == len(left) and dict(other.dump()) == left
    assert other.get_hash_array(keys) == [left.get(h, 0) for h in keys]


def test_more_colliding_keys_than_a_block_holds_fail_cleanly(KCT):
    """9,000 keys that agree in their low 40 bits cannot share one 8,192-slot block at any table size: the merge must
    report that (MemoryError from KCT_ERR_NOMEM) instead of doubling the table until HBM runs out, and the table must
    still be usable afterwards."""
    t = KCT(21, capacity=100_000)
    keys = np.arange(1, 9001, dtype=np.uint64) << np.uint64(40)
    counts = np.ones(keys.size, dtype=np.uint64)
    st = t._lib.kct_merge_host(t._h, keys.ctypes.data, counts.ctypes.data, keys.size, None, None)
    from oxli_amd import _lib as L
    assert st == L.KCT_ERR_NOMEM, (st, L.last_error())
    assert "collide" in L.last_error()
    assert t.capacity <= 1 << 24                        # gave up early
    n = len(t)
    assert 8000 <= n <= 8192                            # a full block's worth was placed, the rest reported
    assert t.sum_counts == n
    t.consume("ACGT" * 50)                              # still works
    assert t.sum_counts == n + 200 - 21 + 1


def test_drop_hash_in_place_matches_dict(KCT):
    rng = random.Random(9)
    seq = "".join(rng.choice("ACGT") for _ in range(60000))
    k = 12
    t, ref = KCT(k, capacity=20000), OracleTable(k)   # a loaded table: long probe runs
    t.consume(seq); ref.consume(seq)
    rk, rc = ref.dump_arrays()
    have = dict(zip(rk.tolist(), rc.tolist()))
    victims = rng.sample(sorted(have), 3000)
    for h in victims:
        t.drop_hash(h)
        del have[h]
    assert len(t) == len(have)
    dk, dc = t.dump_arrays(1)
    assert dict(zip(dk.tolist(), dc.tolist())) == have
    probe = sorted(have)[:2000] + victims[:2000]
    assert t.get_hash_array(probe) == [have.get(h, 0) for h in probe]
    # dropped keys can come back
    t.consume(seq[:5000]); ref2 = OracleTable(k); ref2.consume(seq[:5000])
    k2, c2 = ref2.dump_arrays()
    for h, c in zip(k2.tolist(), c2.tolist()):
        have[h] = have.get(h, 0) + c
    dk, dc = t.dump_arrays(1)
    assert dict(zip(dk.tolist(), dc.tolist())) == have


def test_add_on_one_device_matches_oracle(KCT):
    rng = random.Random(21)
    k = 17
    a_seq = "".join(rng.choice("ACGT") for _ in range(300000))
    b_seq = a_seq[100000:250000] + "".join(rng.choice("ACGT") for _ in range(200000))
    a, b, ra, rb = KCT(k), KCT(k), OracleTable(k), OracleTable(k)
    a.consume(a_seq); ra.consume(a_seq)
    b.consume(b_seq); rb.consume(b_seq)
    b.count_hash(0); rb.count_hash(0)               # key 0 lives beside the device table
    a["A" * k] = 0; ra.add_pairs([ra.hash_kmer("A" * k)], [0])   # a zero-valued key counts as new when added to (lib.rs:801-803)
    b.consume("A" * (k + 4)); rb.consume("A" * (k + 4))
    assert a.add(b) == ra.add(rb)
    assert_same_table(a, ra)
    assert_same_table(b, rb)                          # the source is unchanged


@pytest.mark.parametrize("k,path,cap_hint,G,R", [
    (21, "partitioned", 300_000, 3_000_000, 200_000),     # 64 table blocks for 3 M k-mers: most blocks overflow, one level
    (21, "dedupe", 300_000, 3_000_000, 200_000),          # ... the small compact shadow holds them; its conversion overflows the table
    (31, "dedupe", 300_000, 3_000_000, 200_000),          # ... the table-sized 64-bit shadow overflows too
    (25, "partitioned", 6_000_000, 20_000_000, 600_000),  # 2^24 slots for ~19 M k-mers: two levels, blocks overflow
    (21, "dedupe", 6_000_000, 20_000_000, 600_000),       # ... the compact two-level shadow (2^29 slots) holds them; the table does not
    (31, "dedupe", 6_000_000, 20_000_000, 600_000),       # ... the 2^24-slot 64-bit shadow does not
])
def test_undersized_tables_abandon_blocks_and_recount(KCT, k, path, cap_hint, G, R):
    """A capacity hint far below what the input brings: K2 workgroups find their block full and abandon it, the host makes
    room and recounts those blocks' entries with the direct insert (no per-window spill list exists any more).  Exactness must
    not depend on the hint."""
    import os

    import torch
    L = 150
    genome = oracle.synth_genome(G, 77)
    reads = oracle.synth_reads(genome, 0, R, L, 78)
    ref, n_ref, _ = oracle.baseline_consume(reads, L, k, max(1, min(16, len(os.sched_getaffinity(0)))), native=False)
    dev_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=cap_hint)
    t.set_path(path)
    t.profile(True)
    assert t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), R * L) == n_ref
    dk, dc = t.dump_arrays(1)
    prof = t.profile_read()
    assert "recount_failed_kernel" in prof, prof   # blocks really were abandoned
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert len(t) == len(ref) and t.sum_counts == n_ref
    # once more into the (now large enough) table
    assert t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), R * L) == n_ref
    dk, dc = t.dump_arrays(1)
    assert np.array_equal(dk, rk) and np.array_equal(dc, 2 * rc)


def test_batches_from_two_host_threads_share_the_packer_pool(KCT):
    """kct_consume_batch packs with a process-wide pool of worker threads (one job at a time); ctypes releases the GIL, so
    two Python threads ingesting into two tables do meet there.  Large batches (the pool is used from 8 MiB on), several
    rounds, list and CSR inputs; both tables against the oracle."""
    import threading
    k, G, R, L = 21, 200_000, 70_000, 150
    genome = oracle.synth_genome(G)
    results, errors = {}, []

    def work(tag, first):
        try:
            reads = oracle.synth_reads(genome, first, R, L)
            recs = [bytes(r[:L]) for r in reads]
            flat = np.ascontiguousarray(reads[:, :L]).reshape(-1)
            offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
            t = KCT(k, capacity=G)
            n = 0
            for rep in range(3):
                n += t.consume_batch(recs) if rep % 2 == 0 else t.consume_batch((flat, offs))
            results[tag] = (t, n, recs)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i, i * R)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for tag in (0, 1):
        t, n, recs = results[tag]
        ref = OracleTable(k)
        n_ref = sum(ref.consume(r) for r in recs)
        assert n == 3 * n_ref
        dk, dc = t.dump_arrays(1)
        rk, rc = ref.dump_arrays()
        assert np.array_equal(dk, rk) and np.array_equal(dc, 3 * rc)


def test_a_second_thread_is_turned_away_not_raced(KCT):
    """One caller at a time (lib.rs:546: `&mut self` under the GIL; pyo3 raises "Already borrowed" for a second one).  ctypes and the
    call glue release the GIL around device passes, so a second thread can arrive while a pass runs: it must get RuntimeError("Already
    borrowed") -- and the table must come out exactly as the first thread's calls alone leave it."""
    import threading

    import torch
    G, R, L, k = 2_000_000, 400_000, 150, 21
    reads = oracle.synth_reads(oracle.synth_genome(G, 3), 0, R, L, 5)
    dev = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=G)
    t.set_deferred(False)   # every consume_device call is a device pass of ~1 ms
    stop, seen, errors = threading.Event(), [], []

    def intruder():
        while not stop.is_set():
            try:
                t.get_hash(12345)
            except RuntimeError as e:   # noqa: PERF203
                (seen if "Already borrowed" in str(e) else errors).append(str(e))
    th = threading.Thread(target=intruder)
    th.start()
    total = 0
    for _ in range(30):
        while True:   # (whoever comes second is turned away: that can be this thread too)
            try:
                total += t.consume_device(dev.data_ptr(), dev.numel(), R * L)
                break
            except RuntimeError as e:
                if "Already borrowed" not in str(e):
                    raise
    stop.set()
    th.join()
    assert not errors, errors[:3]
    assert seen, "the second thread never met a running call (or was let in)"
    assert total == 30 * R * (L - k + 1) and t.sum_counts == total
    ref, _, _ = oracle.baseline_consume(reads, L, k, 8, native=False)
    dk, dc = t.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, 30 * rc)
