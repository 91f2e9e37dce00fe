"""Parity of the HIP path against the CPU oracle and the reference's known answers.  All tests
here need a real MI355X (``-m gpu``) and go through the C ABI (ctypes -> libkct_hip.so).

Bar: bit-exact -- every hash, every count, n, len, consumed, sum_counts.
The tests re-express the reference's own tests for this path (src/python/tests/test_basic.py,
test_kmers_and_hashes.py, test_attr.py, test_add.py, test_dunders.py, test_dump.py) plus the
edge cases the domain has: empty / short / ragged records, bad bases, lower case, non-ASCII
bytes, heavy hitters, table growth from a tiny capacity, every k from 1 to 255.
"""
import hashlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
from oracle import OracleTable  # noqa: E402


@pytest.fixture(scope="module", params=["default", "sync"])
def KCT(request):
    """The table class as users construct it ("default": per-record consume() calls are deferred and counted in one
    device pass when the table is next read) and with every consume() a device pass of its own ("sync"), so that
    each device path is also driven record by record."""
    import functools

    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from oxli_amd import KmerCountTable
    if request.param == "default":
        return KmerCountTable

    @functools.wraps(KmerCountTable, updated=())
    def make(*args, **kw):
        kw.setdefault("deferred", False)
        return KmerCountTable(*args, **kw)
    return make


def assert_same_table(dev, ref):
    dk, dc = dev.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert dk.size == rk.size, (dk.size, rk.size)
    assert np.array_equal(dk, rk), "hash sets differ"
    assert np.array_equal(dc, rc), "counts differ"
    assert len(dev) == len(ref)
    assert dev.sum_counts == ref.sum_counts
    assert dev.consumed == ref.consumed


def rand_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


# ---- hashes ---------------------------------------------------------------------------------------
def test_reference_hash_kats(KCT, kats):
    tables = {}
    for e in kats["hashes"]:
        k = len(e["kmer"])
        t = tables.setdefault(k, KCT(k))
        assert t.hash_kmer(e["kmer"]) == e["hash"], e
        assert t.hash_kmer(e["kmer"].lower()) == e["hash"], e


def test_reference_window_hash_lists(KCT, kats):
    for e in kats["window_hashes"]:
        assert KCT(e["k"]).hash_windows(e["seq"]).tolist() == e["hashes"], e


def test_hash_windows_equal_oracle_every_k(KCT):
    rng = random.Random(11)
    seq = rand_dna(rng, 3000)
    noisy = list(seq)
    for pos in rng.sample(range(len(noisy)), 25):
        noisy[pos] = rng.choice("NnXx-*RYacgt")
    noisy = "".join(noisy)
    for k in list(range(1, 70)) + [95, 96, 97, 127, 128, 129, 200, 254, 255]:
        t = KCT(k)
        for s in (seq, noisy):
            want, _ = oracle.seq_to_hashes(s, k, force=True)
            got = t.hash_windows(s)
            assert np.array_equal(got, want), f"k={k}"


def test_exception_types_at_the_two_panic_and_anyhow_edges(KCT, tmp_path):
    """get() on invalid DNA of the right length: the reference PANICS (lib.rs:176 `.expect`), which pyo3 surfaces as
    PanicException -- a BaseException, not caught by `except Exception`.  load() of a missing path: File::open fails inside an
    anyhow::Result function (lib.rs:296-299), which pyo3 turns into RuntimeError (save() to a bad path is a PyIOError)."""
    from oxli_amd import KmerCountTable
    from oxli_amd.table import PanicException
    t = KCT(4)
    with pytest.raises(PanicException, match="error hashing this k-mer"):
        t.get("ACGN")
    assert not issubclass(PanicException, Exception)
    with pytest.raises(ValueError):
        t.get("ACG")
    with pytest.raises(RuntimeError, match="No such file or directory"):
        KmerCountTable.load(str(tmp_path / "missing.json.gz"))
    with pytest.raises(OSError, match="No such file or directory"):
        t.save(str(tmp_path / "noexist" / "t.json.gz"))


def test_hash_kmer_errors(KCT):
    t = KCT(4)
    with pytest.raises(RuntimeError, match="wrong ksize"):
        t.hash_kmer("ACG")
    with pytest.raises(RuntimeError):
        t.hash_kmer("ACGN")
    # the reference compares `len as u8` (lib.rs:66): a 260-byte string passes a ksize-4 check
    long = "ACGT" + "T" * 256
    assert t.hash_kmer(long) == oracle.hash_kmer(long, 4) == oracle.hash_kmer("ACGT")


# ---- reference test_basic.py ------------------------------------------------------------------------
def test_count_get(KCT):
    cg = KCT(4)
    assert cg.get("ATCG") == 0
    assert cg.count("ATCG") == 1
    assert cg.get("ATCG") == 1
    assert cg["ATCG"] == 1


def test_count_hash_and_count_agree(KCT):
    kmer = "TAAACCCTAACCCTAACCCTAACCCTAACCC"
    cg = KCT(ksize=31)
    h = cg.hash_kmer(kmer)
    assert h == oracle.hash_kmer(kmer)
    assert cg.get_hash(h) == 0 and cg.get(kmer) == 0
    assert cg.count(kmer) == 1 and cg.count(kmer) == 2 and cg.get(kmer) == 2
    assert cg.count_hash(h) == 3 and cg.get(kmer) == 3 and cg.get_hash(h) == 3
    assert cg.consumed == 62 and cg.sum_counts == 3 and len(cg) == 1


def test_wrong_ksize(KCT):
    cg = KCT(3)
    with pytest.raises(ValueError):
        cg.count("ATCG")
    with pytest.raises(ValueError):
        cg.get("ATCG")


def test_reference_consume_facts(KCT, kats):
    for e in kats["consume"]:
        t = KCT(e["k"])
        assert t.consume(e["seq"]) == e["n"], e
        for kmer, c in e.get("get", {}).items():
            assert t.get(kmer) == c, (e, kmer)
        if "len" in e:
            assert len(t) == e["len"]
        if "sum_counts" in e:
            assert t.sum_counts == e["sum_counts"]
        assert t.consumed == len(e["seq"])


def test_reference_consume_errors(KCT, kats):
    for e in kats["consume_errors"]:
        t = KCT(e["k"])
        with pytest.raises(ValueError, match=f"bad k-mer encountered at position {e['position']}$"):
            t.consume(e["seq"], skip_bad_kmers=False)
        assert t.consumed == 0
        assert t.sum_counts == e["position"]


def test_get_hash_array_order(KCT):
    t = KCT(3)
    for k in ["AAA", "TTT", "AAC"]:
        t.count(k)
    hs = [t.hash_kmer("AAA"), t.hash_kmer("AAC"), t.hash_kmer("GGG")]
    assert t.get_hash_array(hs) == [2, 1, 0]
    assert t.get_hash_array(hs[::-1]) == [0, 1, 2]


# ---- reference doc/api.md + README on doc/example.fa ------------------------------------------------
def test_example_fa(KCT, kats, example_seq, example_digests):
    for e in kats["example_fa"]:
        t = KCT(e["k"])
        assert t.consume(example_seq) == e["n"]
        for kmer, c in e.get("get", {}).items():
            assert t.get(kmer) == c
        if "then_consume_skip" in e:
            x = e["then_consume_skip"]
            with pytest.raises(ValueError, match="bad k-mer encountered at position 0"):
                t.consume(x["seq"], skip_bad_kmers=False)
            assert t.consume(x["seq"]) == x["n"]
            for kmer, c in x["get"].items():
                assert t.get(kmer) == c
    for k, d in example_digests["k"].items():
        t = KCT(int(k))
        assert t.consume(example_seq) == d["n"]
        keys, counts = t.dump_arrays(1)
        assert len(t) == d["distinct"] and int(counts.max()) == d["max"] and t.consumed == d["consumed"]
        tsv = "".join(f"{h}\t{c}\n" for h, c in zip(keys.tolist(), counts.tolist()))
        assert hashlib.sha256(tsv.encode()).hexdigest() == d["sha256_dump_sortkeys_tsv"]
        assert t.hash_windows(example_seq[:100])[:3].tolist() == d["first3"]


# ---- consume == oracle on seeded inputs --------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 4, 15, 16, 17, 21, 31, 32, 33, 51, 64, 65, 100, 255])
def test_consume_matches_oracle(KCT, k):
    rng = random.Random(1000 + k)
    dev, ref = KCT(k), OracleTable(k)
    seqs = [rand_dna(rng, rng.choice([0, 1, k - 1, k, k + 1, 2 * k, 150, 1000, 9000])) for _ in range(12)]
    seqs.append(rand_dna(rng, 4000, "ACGTN"))
    seqs.append(rand_dna(rng, 2000, "acgtACGT"))
    seqs.append("A" * 3000)
    seqs.append("AC" * 1500)
    for s in seqs:
        assert dev.consume(s) == ref.consume(s), (k, len(s))
    assert_same_table(dev, ref)


def test_consume_error_mode_matches_oracle(KCT):
    rng = random.Random(5)
    for k in (4, 21, 33):
        for trial in range(8):
            s = list(rand_dna(rng, rng.randint(k, 400)))
            if trial % 4 != 3:
                s[rng.randrange(len(s))] = "N"
            s = "".join(s)
            dev, ref = KCT(k), OracleTable(k)
            outcomes = []
            for t in (dev, ref):
                try:
                    outcomes.append(("ok", t.consume(s, skip_bad_kmers=False)))
                except ValueError as e:
                    outcomes.append(("err", str(e)))
            assert outcomes[0] == outcomes[1], (k, s)
            assert_same_table(dev, ref)


def test_non_ascii_and_byte_semantics(KCT):
    dev, ref = KCT(4), OracleTable(4)
    for s in ["ACGTéACGT", "ACGT\x00ACGT", "\nACGTAC\r\nGGTTAA\n", "ACGT" * 3 + "中" + "TTGCA"]:
        assert dev.consume(s) == ref.consume(s)
    assert_same_table(dev, ref)  # consumed counts UTF-8 bytes (lib.rs:548)


def test_heavy_hitters(KCT, kats):
    s = kats["stress"]
    seq = s["unit"] * s["repeat"]
    a, b = KCT(s["k"]), KCT(s["k"])
    assert a.consume(seq) == s["n"]
    assert b.consume(seq) == s["n"]
    assert a.add(b) == (s["add_counts_added"], s["add_new_keys"])
    assert a.sum_counts == s["sum_counts_after_add"]
    ref = OracleTable(s["k"])
    ref.consume(seq); ref.consume(seq)
    dk, dc = a.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    # homopolymer: one key, every lane of every wave folds into it
    t = KCT(21)
    assert t.consume("A" * 100000) == 100000 - 20
    assert len(t) == 1 and t.get("T" * 21) == 100000 - 20


def test_growth_from_tiny_capacity(KCT):
    rng = random.Random(3)
    seq = rand_dna(rng, 300000)
    dev, ref = KCT(25, capacity=16), OracleTable(25)
    cap0 = dev.capacity
    assert dev.consume(seq) == ref.consume(seq)
    assert dev.capacity > cap0
    for _ in range(3):
        s = rand_dna(rng, 50000)
        assert dev.consume(s) == ref.consume(s)
    assert_same_table(dev, ref)


# ---- batch API == the per-record loop -----------------------------------------------------------------
def test_consume_batch_equals_loop(KCT):
    rng = random.Random(21)
    recs = [rand_dna(rng, rng.choice([0, 5, 20, 21, 22, 150, 151, 3000]), "ACGTACGTACGTN") for _ in range(300)]
    for k in (21, 40):
        dev, ref = KCT(k), OracleTable(k)
        n = dev.consume_batch(recs)
        assert n == sum(ref.consume(r) for r in recs)
        assert_same_table(dev, ref)
        # CSR form
        dev2 = KCT(k)
        data = "".join(recs).encode()
        offs = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
        assert dev2.consume_batch((data, offs)) == n
        assert_same_table(dev2, ref)
    assert KCT(21).consume_batch([]) == 0
    t = KCT(21)
    assert t.consume_batch(["", b""]) == 0 and t.consumed == 0 and len(t) == 0   # records with no bytes at all


def test_consume_batch_error_mode(KCT):
    rng = random.Random(22)
    k = 21
    recs = [rand_dna(rng, 150) for _ in range(50)]
    recs[10] = "ACGTN"                      # too short to have a window: must NOT raise
    recs[30] = recs[30][:70] + "N" + recs[30][71:]
    dev, ref = KCT(k), OracleTable(k)
    n_ref, err_ref = 0, None
    for r in recs:
        try:
            n_ref += ref.consume(r, skip_bad_kmers=False)
        except ValueError as e:
            err_ref = str(e)
            break
    with pytest.raises(ValueError) as ei:
        dev.consume_batch(recs, skip_bad_kmers=False)
    assert str(ei.value) == err_ref == "bad k-mer encountered at position 50"
    assert ei.value.record == 30
    assert_same_table(dev, ref)
    # no bad window anywhere: plain success
    clean = [rand_dna(rng, 100) for _ in range(20)]
    dev, ref = KCT(k), OracleTable(k)
    assert dev.consume_batch(clean, skip_bad_kmers=False) == sum(ref.consume(r, False) for r in clean)
    assert_same_table(dev, ref)


# ---- merge (reference test_add.py) -----------------------------------------------------------------------
def test_reference_add_facts(KCT, kats):
    for e in kats["add"]:
        a, b = KCT(e["k"]), KCT(e["k"])
        if e["a"]:
            a.consume(e["a"])
        b.consume(e["b"])
        assert a.add(b) == (e["counts_added"], e["new_keys"]), e
        assert a.sum_counts == e["sum_counts"]
        if "len" in e:
            assert len(a) == e["len"]
        assert a.consumed == len(e["a"]) + len(e["b"])
    with pytest.raises(ValueError):
        KCT(5).add(KCT(6))


def test_add_matches_oracle_and_counts_zero_valued_keys_as_new(KCT):
    rng = random.Random(8)
    g = rand_dna(rng, 5000)
    a, b, ra, rb = KCT(17), KCT(17), OracleTable(17), OracleTable(17)
    for t in (a, ra):
        t.consume(g[:3000])
    for t in (b, rb):
        t.consume(g[2000:])
    assert a.add(b) == ra.add(rb)
    assert_same_table(a, ra)
    # __setitem__(kmer, 0) leaves a key whose count is 0; add() reports it as new (lib.rs:801-803)
    x, y = KCT(4), KCT(4)
    x["ACGT"] = 0
    assert len(x) == 1 and x.get("ACGT") == 0
    y.count("ACGT")
    assert x.add(y) == (1, 1)
    x["AAAA"] = 7
    assert x.get("TTTT") == 7 and x.sum_counts == 8


def test_dump_orders(KCT):
    t = KCT(4)
    t.count("AAAA"); t.count("TTTT"); t.count("AATT"); t.count("GGGG"); t.count("GGGG")
    assert t.dump(sortkeys=True) == [(73459868045630124, 2), (382727017318141683, 1), (17832910516274425539, 2)]
    assert t.dump(sortcounts=True) == [(382727017318141683, 1), (73459868045630124, 2), (17832910516274425539, 2)]
    assert sorted(t.dump()) == t.dump(sortkeys=True)
    with pytest.raises(ValueError):
        t.dump(sortcounts=True, sortkeys=True)
    assert KCT(4).dump() == []


def test_hash_zero_is_a_legal_key_for_count_hash(KCT):
    t = KCT(4)
    assert t.get_hash(0) == 0 and len(t) == 0
    assert t.count_hash(0) == 1 and t.count_hash(0) == 2
    assert t.get_hash(0) == 2 and len(t) == 1 and t.sum_counts == 2
    assert t.get_hash_array([0, 5]) == [2, 0]
    assert t.dump(sortkeys=True) == [(0, 2)]


# ---- device-resident input + synthetic workload ------------------------------------------------------------
def test_device_generator_matches_oracle_and_consume_device(KCT):
    import ctypes as C

    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N, k = 50000, 150, 4096, 21
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 42, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 100, N, L, 1337, stream) == 0
    torch.cuda.synchronize()
    genome = oracle.synth_genome(G)
    reads = oracle.synth_reads(genome, 100, N, L)
    assert np.array_equal(g.cpu().numpy(), genome)
    assert np.array_equal(r.cpu().numpy().reshape(N, L + 1), reads)
    dev, ref = KCT(k, capacity=G), OracleTable(k)
    n = dev.consume_device(r.data_ptr(), r.numel(), N * L)
    assert n == sum(ref.consume(reads[i, :L]) for i in range(N)) == N * (L - k + 1)
    assert_same_table(dev, ref)


def test_full_size_properties_1M_reads(KCT):
    """BASELINE config C2 at full size (1 M x 150 bp, k=21): properties that need no oracle run.
    sum_counts == n == reads * 130; consumed == reads * 150; consuming the same reads again
    doubles every count and adds no key; counts of a sampled read's k-mers are >= 1."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N, k = 5_000_000, 150, 1_000_000, 21
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 42, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 1337, stream) == 0
    torch.cuda.synchronize()
    t = KCT(k, capacity=G)
    n = t.consume_device(r.data_ptr(), r.numel(), N * L)
    assert n == N * (L - k + 1) == t.sum_counts
    assert t.consumed == N * L
    distinct = len(t)
    assert 0.9 * G < distinct <= G - k + 1
    keys, counts = t.dump_arrays(1)
    assert np.all(keys[:-1] < keys[1:]) and int(counts.sum()) == n
    assert t.consume_device(r.data_ptr(), r.numel(), N * L) == n
    assert len(t) == distinct
    keys2, counts2 = t.dump_arrays(1)
    assert np.array_equal(keys, keys2) and np.array_equal(counts2, 2 * counts)
    # a slice of the same reads through the oracle: every one of its keys is present with count >= oracle's
    sub = r[: 2000 * (L + 1)].cpu().numpy().reshape(2000, L + 1)
    ref = OracleTable(k)
    for i in range(2000):
        ref.consume(sub[i, :L])
    rk, rc = ref.dump_arrays()
    got = np.array(t.get_hash_array(rk), dtype=np.uint64)
    assert np.all(got >= 2 * rc)


# ---- the partitioned path (radix-partition by table block + LDS counting) == the direct path == oracle ----
def _mixed_inputs(rng, k):
    seqs = [rand_dna(rng, n) for n in (k, 2 * k, 150, 5000, 40000)]
    seqs.append(rand_dna(rng, 300000))
    seqs.append(rand_dna(rng, 20000, "ACGTN"))
    seqs.append(rand_dna(rng, 20000, "acgtACGT"))
    seqs.append("A" * 50000)                       # every lane of every wave holds the same k-mer
    seqs.append("AC" * 30000 + "ACG" * 30000)      # tandem repeats: a handful of k-mers, ring overruns
    seqs.append(rand_dna(rng, 1000) * 60)          # a 1 kbp unit repeated: every k-mer 60 times
    return seqs


@pytest.mark.parametrize("k", [21, 31, 32, 33, 51, 100])
def test_partitioned_path_matches_oracle(KCT, k):
    rng = random.Random(4000 + k)
    seqs = _mixed_inputs(rng, k)
    ref = OracleTable(k)
    n_ref = [ref.consume(s) for s in seqs]
    for path in ("partitioned", "direct", "auto"):
        dev = KCT(k, capacity=400000)
        dev.set_path(path)
        assert [dev.consume(s) for s in seqs] == n_ref, path
        assert_same_table(dev, ref)
    # one batch call (all records in one pass of the pipeline)
    dev = KCT(k, capacity=400000)
    dev.set_path("partitioned")
    assert dev.consume_batch(seqs) == sum(n_ref)
    assert_same_table(dev, ref)


def test_partitioned_path_every_k_up_to_64(KCT):
    """The partition kernel is instantiated once per k <= 64 (compile-time shifts, masks and MurmurHash3 block / tail
    structure): every instantiation against the oracle, on input with bad bytes, lower case and short records."""
    rng = random.Random(64)
    recs = [rand_dna(rng, n, "ACGTACGTACGTACGTNacgt") for n in (0, 1, 31, 32, 33, 63, 64, 65, 100, 150, 151, 1000)] * 6
    recs += [rand_dna(rng, 20000), "AC" * 5000]
    for k in range(1, 65):
        ref = OracleTable(k)
        n_ref = sum(ref.consume(r) for r in recs)
        dev = KCT(k, capacity=200000)
        dev.set_path("partitioned")
        assert dev.consume_batch(recs) == n_ref, k
        assert_same_table(dev, ref)


@pytest.mark.parametrize("k", [1, 5, 16, 21, 31, 32, 33, 41, 51, 64])
def test_dedupe_first_path_matches_oracle(KCT, k):
    """Dedupe-first paths (compact k <= 21, 64-bit k <= 32, 128-bit 33 <= k <= 64): packed k-mers are counted in LDS shadow blocks, each
    distinct one is hashed once.
    Deep coverage of a small genome (its home ground), bad bytes, lower case, homopolymers, short records."""
    rng = random.Random(7000 + k)
    genome = rand_dna(rng, 40000)
    recs = []
    for i in range(32000):
        a = rng.randrange(0, len(genome) - 150)
        r = genome[a:a + 150]
        if i % 2:
            r = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if i % 97 == 0:
            r = r[:70] + "N" + r[71:]
        if i % 89 == 0:
            r = r.lower()
        recs.append(r)
    recs += ["A" * 3000, "T" * 3000, "ACGT" * 700, "", "ACG", rand_dna(rng, 5000, "ACGTN")]
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    dev = KCT(k, capacity=200000)
    dev.set_path("dedupe")
    dev.profile(True)
    assert dev.consume_batch(recs) == n_ref
    want = "aggregate_blocks32_kernel" if k <= 21 else "aggregate_blocks_kernel<shadow>" if k <= 32 else "aggregate_blocks128_kernel"
    assert want in dev.profile_read()                                     # really that path
    assert_same_table(dev, ref)
    # a second pass into the live table (every key exists already), through the automatic choice this time
    auto = KCT(k, capacity=200000)
    assert auto.consume_batch(recs) == n_ref                             # first pass: nothing known yet, standard paths
    auto.profile(True)
    assert auto.consume_batch(recs) == n_ref
    # few keys, many k-mers: dedupe-first was chosen -- the 64-bit variant at every k here: 4.8x10^6 windows do not pay for the compact
    # variant's fixed 64 MiB shadow (0.15 windows per shadow byte, path_policy.h), but do for the table-sized 8 MiB one
    # (k > 32: the 128-bit variant buys 1.0x over hashing every window on its showcase and is never chosen by itself since round 4 --
    # set_path("dedupe") above still runs it)
    assert ("aggregate_blocks_kernel<shadow>" if k <= 32 else "aggregate_blocks_kernel") in auto.profile_read()
    assert "aggregate_blocks128_kernel" not in auto.profile_read()
    for r in recs:
        ref.consume(r)
    assert_same_table(auto, ref)


def test_dedupe_first_pending_counts_are_seen_by_every_read(KCT):
    """The dedupe-first path leaves its counts pending in a shadow table; anything that reads the table converts them
    first.  Interleave passes with point reads, point writes, add(), clear() and the other ingest paths."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N, k = 300_000, 150, 40_000, 25            # 40k reads = 6e6 window starts per pass
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 3, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 4, stream) == 0
    torch.cuda.synchronize()
    reads = r.cpu().numpy().tobytes().decode().split("\n")[:N]
    ref = OracleTable(k)
    n1 = sum(ref.consume(x) for x in reads)
    dev = KCT(k, capacity=G)
    dev.set_path("dedupe")
    assert dev.consume_device(r.data_ptr(), r.numel(), N * L) == n1
    probe = reads[7][10:10 + k]
    assert dev.get(probe) == ref.get(probe)                       # point read after a pass
    assert dev.consume_device(r.data_ptr(), r.numel(), N * L) == n1
    for x in reads:
        ref.consume(x)
    assert dev.count(probe) == ref.count(probe)                   # point write sees the pending counts
    assert dev.consume_device(r.data_ptr(), r.numel(), N * L) == n1
    for x in reads:
        ref.consume(x)
    dev.set_path("direct")                                        # another path on top of pending counts
    assert dev.consume(reads[3]) == ref.consume(reads[3])
    dev.set_path("dedupe")
    assert len(dev) == len(ref) and dev.sum_counts == ref.sum_counts
    assert_same_table(dev, ref)
    other, oref = KCT(k, capacity=G), OracleTable(k)
    other.set_path("dedupe")
    assert other.consume_device(r.data_ptr(), r.numel(), N * L) == n1
    for x in reads:
        oref.consume(x)
    acc, aref = KCT(k), OracleTable(k)
    acc.consume(reads[0]); aref.consume(reads[0])
    assert acc.add(other) == aref.add(oref)                       # src's pending counts are part of src
    assert_same_table(acc, aref)
    other.consume_device(r.data_ptr(), r.numel(), N * L)
    other.clear()                                                 # pending counts are forgotten with everything else
    assert len(other) == 0 and other.sum_counts == 0 and other.get(probe) == 0
    assert other.consume_device(r.data_ptr(), r.numel(), N * L) == n1
    assert other.max == int(oref.dump_arrays()[1].max()) and other.histo(zero=False) == oref_histo(oref)
    import torch as _t
    free0 = _t.cuda.mem_get_info()[0]
    other.release_scratch()                                       # working buffers go, the table and its answers stay
    assert _t.cuda.mem_get_info()[0] > free0 + (1 << 26)
    assert other.max == int(oref.dump_arrays()[1].max()) and len(other) == len(oref)
    assert other.consume_device(r.data_ptr(), r.numel(), N * L) == n1 and other.sum_counts == 2 * oref.sum_counts


def oref_histo(ref):
    _, counts = ref.dump_arrays()
    vals, freq = np.unique(counts, return_counts=True)
    return list(zip(vals.tolist(), freq.tolist()))


def test_compact_dedupe_path_over_many_passes_and_when_outgrown(KCT):
    """k <= 21 takes the compact variant (u32 key + u32 count shadow).  Pending counts pile up over many passes without a
    conversion in between (more than 2^31 window starts in all); and when the input brings more k-mers than the 8.4 M-slot
    shadow of a small table holds, the table grows and a shadow of the new geometry (two partition levels) takes over."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    L, N, k = 150, 1_000_000, 21
    stream = torch.cuda.current_stream().cuda_stream

    def make(G, seed):
        g = torch.empty(G, dtype=torch.uint8, device="cuda")
        r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
        assert lib.kct_synth_genome_device(g.data_ptr(), G, seed, stream) == 0
        assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, seed + 1, stream) == 0
        torch.cuda.synchronize()
        return r

    ra = make(3_000_000, 11)
    std = KCT(k, capacity=5_000_000)
    std.set_path("partitioned")
    n = std.consume_device(ra.data_ptr(), ra.numel(), N * L)
    ka, ca = std.dump_arrays(1)
    dev = KCT(k, capacity=5_000_000)
    assert dev.capacity == 1 << 23                                 # 1024 blocks: the small compact shadow, one level
    dev.set_path("dedupe")
    dev.profile(True)
    passes = 16                                                    # 16 x 1.5e8 window starts > 2^31
    for _ in range(passes):
        assert dev.consume_device(ra.data_ptr(), ra.numel(), N * L) == n
    prof = dev.profile_read()                                      # (reading the profile converts what is pending)
    assert prof["aggregate_blocks32_kernel"][0] == passes and prof["flush_partition_kernel"][0] == 1   # ... once, for all 16 passes
    kd, cd = dev.dump_arrays(1)
    assert np.array_equal(kd, ka) and np.array_equal(cd, passes * ca)
    # a second genome: 3 M + 4 M distinct k-mers do not fit the small shadow (nor 65 % of the table): both grow
    rb = make(4_000_000, 1_000_000_021)        # (the generator indexes one stream by seed + position: seeds far apart)
    nb = std.consume_device(rb.data_ptr(), rb.numel(), N * L)
    dev.profile_reset()
    for _ in range(3):
        assert dev.consume_device(rb.data_ptr(), rb.numel(), N * L) == nb
    prof = dev.profile_read()
    assert dev.capacity > 1 << 23 and "repartition_kernel<compact>" in prof, (dev.capacity, prof)
    std.consume_device(rb.data_ptr(), rb.numel(), N * L); std.consume_device(rb.data_ptr(), rb.numel(), N * L)
    for _ in range(passes - 1):
        std.consume_device(ra.data_ptr(), ra.numel(), N * L)
    ks, cs = std.dump_arrays(1)
    kd, cd = dev.dump_arrays(1)
    assert np.array_equal(kd, ks) and np.array_equal(cd, cs)


def test_consume_file_through_the_dedupe_first_paths(KCT, tmp_path):
    """File ingestion feeds chunks (16 MiB here: the table is too small to stage them) to the same pass machinery: with the dedupe-first paths forced, a FASTA of deep
    coverage must give the table the standard path gives (several parser threads, chunks counted in any order)."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N = 200_000, 150, 400_000
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 77, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 78, stream) == 0
    torch.cuda.synchronize()
    reads = r.cpu().numpy().reshape(N, L + 1)[:, :L]
    fa = tmp_path / "deep.fa"
    with open(fa, "wb") as f:
        for i in range(N):
            f.write(b">r%d\n" % i); f.write(reads[i].tobytes()); f.write(b"\n")
    for k in (21, 29):                                   # compact (k <= 21) and 64-bit (k <= 32) variants
        std = KCT(k, capacity=G)
        std.set_path("partitioned")
        n = std.consume_file(str(fa))
        assert n == N * (L - k + 1)
        sk, sc = std.dump_arrays(1)
        dev = KCT(k, capacity=G)
        dev.set_path("dedupe")
        dev.profile(True)
        assert dev.consume_file(str(fa)) == n and dev.consume_file(str(fa)) == n
        prof = dev.profile_read()
        assert ("aggregate_blocks32_kernel" if k <= 21 else "aggregate_blocks_kernel<shadow>") in prof, prof
        dk, dc = dev.dump_arrays(1)
        assert np.array_equal(dk, sk) and np.array_equal(dc, 2 * sc)
        assert dev.consumed == 2 * std.consumed


def test_dedupe_first_path_with_too_many_distinct_kmers(KCT):
    """All-distinct input overflows the scratch blocks: the overflow goes straight to the table (still exact), and the
    table stops choosing the dedupe-first path."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N, k = 60_000_000, 150, 300_000, 27          # 3.7e7 k-mers, nearly all distinct: > 8.4e6 scratch slots
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 5, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 6, stream) == 0
    torch.cuda.synchronize()
    std = KCT(k, capacity=40_000_000)
    std.set_path("partitioned")
    n = std.consume_device(r.data_ptr(), r.numel(), N * L)
    sk, sc = std.dump_arrays(1)
    ded = KCT(k, capacity=40_000_000)
    ded.set_path("dedupe")
    assert ded.consume_device(r.data_ptr(), r.numel(), N * L) == n
    dk, dc = ded.dump_arrays(1)
    assert np.array_equal(dk, sk) and np.array_equal(dc, sc)
    # automatic choice: input like this never looks worth it (too few windows per k-mer the table already holds)
    auto = KCT(k, capacity=40_000_000)
    n0 = (N // 8) // 16 * 16                                               # (a 16-byte aligned cut of the stream)
    lo = n0 * (L + 1)
    auto.consume_device(r.data_ptr(), lo, n0 * L)
    auto.profile(True)
    auto.consume_device(r.data_ptr() + lo, r.numel() - lo, (N - n0) * L)
    assert "aggregate_blocks_kernel<shadow>" not in auto.profile_read()
    auto.consume_device(r.data_ptr(), r.numel(), N * L)
    ak, ac = auto.dump_arrays(1)
    assert np.array_equal(ak, sk) and np.array_equal(ac, 2 * sc)


@pytest.mark.parametrize("k", [21, 51])
def test_two_level_partitioned_path_matches_oracle(KCT, k):
    """Tables with more than 1024 blocks (> 128 MiB) take two partition levels: K1 into 1024 super-bins,
    repartition_kernel into the table blocks, then the per-block LDS count."""
    rng = random.Random(5000 + k)
    seqs = _mixed_inputs(rng, k)
    ref = OracleTable(k)
    n_ref = [ref.consume(s) for s in seqs]
    dev = KCT(k, capacity=12_000_000)      # 2^25 slots = 4096 blocks: 4 blocks per super-bin
    assert dev.capacity == 1 << 25
    dev.set_path("partitioned")
    assert [dev.consume(s) for s in seqs] == n_ref
    assert_same_table(dev, ref)
    dev.clear()
    assert dev.consume_batch(seqs) == sum(n_ref)   # one pass over everything, starting from a lazily cleared table
    assert_same_table(dev, ref)


def test_partitioned_path_shallow_ring_bursts(KCT):
    """1024 table blocks leave each block a 16-entry ring.  Repeats send far more than 16 hashes to one
    block between two flushes: positions beyond the ring must come out as holes, never as a second
    copy of the lines in front (regression test for an aliasing bug in the compacted flush)."""
    rng = random.Random(4242)
    k = 21
    seqs = [rand_dna(rng, 500) * 400, "ACGT" * 50000, rand_dna(rng, 200000), ("AC" * 40 + rand_dna(rng, 37)) * 2000]
    ref = OracleTable(k)
    n_ref = [ref.consume(s) for s in seqs]
    dev = KCT(k, capacity=5_000_000)       # 2^23 slots = 1024 blocks
    assert dev.capacity == 1 << 23
    dev.set_path("partitioned")
    assert [dev.consume(s) for s in seqs] == n_ref
    assert_same_table(dev, ref)


def test_partitioned_path_updates_a_live_table(KCT):
    """Second and later passes load each table block into LDS instead of starting from zeros; point
    updates made through the atomic path in between must be seen and kept."""
    rng = random.Random(77)
    k = 25
    a, b, c = rand_dna(rng, 200000), rand_dna(rng, 200000), rand_dna(rng, 1000)
    dev, ref = KCT(k, capacity=600000), OracleTable(k)
    dev.set_path("partitioned")
    for t in (dev, ref):
        t.consume(a)
        t.count(c[:k]); t.count(c[:k]); t.count_hash(12345)
        t.consume(b)
        t.consume(a[:100000])
    assert_same_table(dev, ref)
    dev.clear()
    ref = OracleTable(k)
    assert dev.consume(b) == ref.consume(b)   # lazy clear -> fresh pass
    assert_same_table(dev, ref)
    dev.clear()
    assert len(dev) == 0 and dev.sum_counts == 0 and dev.get(c[:k]) == 0 and dev.dump() == []


def test_partitioned_path_grows_when_blocks_fill(KCT):
    rng = random.Random(78)
    k = 27
    s = rand_dna(rng, 1500000)       # ~1.5 M distinct k-mers into a table sized for 100 k
    dev, ref = KCT(k, capacity=100000), OracleTable(k)
    dev.set_path("partitioned")
    cap0 = dev.capacity
    assert dev.consume(s) == ref.consume(s)
    assert dev.capacity > cap0
    assert_same_table(dev, ref)


def test_partitioned_path_abandons_on_pathological_input(KCT):
    """Period-3 repeats keep overrunning one ring; the side list overflows, the pass is abandoned
    and the batch is recounted on the direct path -- same answer."""
    k = 21
    s = "ACG" * 1000000
    dev, ref = KCT(k, capacity=200000), OracleTable(k)
    dev.set_path("partitioned")
    assert dev.consume(s) == ref.consume(s)
    assert_same_table(dev, ref)


# ---- multi-GPU merge building blocks, exercised on one GPU ---------------------------------------------------
def test_owner_bucketed_export_and_pair_merge_simulated_ranks(KCT):
    """Four 'ranks' on one device: each counts its shard, exports pairs bucketed by owner
    (kct_export_by_owner_device), owners fold what they are sent (kct_merge_pairs_device).  The union
    of the owner tables must equal the single-table oracle, and the native bucketing must agree with
    the torch implementation the gloo test covers (oxli_amd.distributed.partition_by_owner)."""
    import ctypes as C

    import torch

    from oxli_amd.distributed import owner_of, partition_by_owner
    world, k, L, per_rank = 4, 21, 150, 5000
    genome = oracle.synth_genome(60000)
    reads = oracle.synth_reads(genome, 0, world * per_rank, L)
    ref = OracleTable(k)
    for i in range(reads.shape[0]):
        ref.consume(reads[i, :L])
    shards, sent = [], []
    for r in range(world):
        t = KCT(k, capacity=100000)
        t.consume_batch([bytes(x[:L]) for x in reads[r * per_rank:(r + 1) * per_rank]])
        n = len(t)
        pairs = torch.empty((n, 2), dtype=torch.int64, device="cuda")
        counts = np.zeros(world, dtype=np.uint64)
        got = C.c_uint64()
        t._check(t._lib.kct_export_by_owner_device(t._h, world, C.c_void_p(pairs.data_ptr()), n, counts.ctypes.data, C.byref(got)))
        assert got.value == n == int(counts.sum())
        own = owner_of(pairs[:, 0], world).cpu().numpy()
        assert np.all(own[1:] >= own[:-1])
        assert np.array_equal(np.bincount(own, minlength=world).astype(np.uint64), counts)
        # same multiset per owner as the torch bucketing of a plain dump
        hk, hc = t.dump_arrays(0)
        tp, tc = partition_by_owner(torch.from_numpy(hk.view(np.int64).copy()), torch.from_numpy(hc.view(np.int64).copy()), world)
        assert np.array_equal(tc.numpy().astype(np.uint64), counts)
        a = pairs.cpu().numpy(); b = tp.numpy()
        off = 0
        for p in range(world):
            c = int(counts[p])
            assert np.array_equal(a[off:off + c][np.argsort(a[off:off + c, 0].view(np.uint64))], b[off:off + c][np.argsort(b[off:off + c, 0].view(np.uint64))])
            off += c
        shards.append(t)
        sent.append((pairs, counts))
    keys_all, counts_all = [], []
    for p in range(world):
        owner = KCT(k, capacity=100000)
        for pairs, counts in sent:
            off = int(counts[:p].sum())
            part = pairs[off:off + int(counts[p])].contiguous()
            a_, b_ = C.c_uint64(), C.c_uint64()
            owner._check(owner._lib.kct_merge_pairs_device(owner._h, C.c_void_p(part.data_ptr()), part.shape[0], C.byref(a_), C.byref(b_)))
            assert a_.value == int(part[:, 1].sum())
        dk, dc = owner.dump_arrays(1)
        keys_all.append(dk); counts_all.append(dc)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(np.concatenate(keys_all), rk)   # owner slices are contiguous in hash space
    assert np.array_equal(np.concatenate(counts_all), rc)


# ---- file ingestion (the README loop) ----------------------------------------------------------------------
def _write_fasta(path, records, width=70, gz=False):
    import gzip
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for i, s in enumerate(records):
            f.write(f">rec{i} some description\n")
            for j in range(0, len(s), width):
                f.write(s[j:j + width] + "\n")


def test_consume_file_example_fa(KCT, kats):
    import os
    fa = os.path.join(os.path.dirname(__file__), "golden", "example.fa")
    for e in kats["example_fa"]:
        t = KCT(e["k"])
        assert t.consume_file(fa) == e["n"]          # README.md:94-99 / doc/api.md:20-25
        assert t.last_file_records == 1 and t.consumed == 349930


def _bgzf(data, rng):
    """`data` as BGZF (SAM specification 4.1): gzip members of random sizes up to 64 KiB of text, each with the 'BC' extra subfield
    holding its compressed size - 1, and the 28-byte end-of-file block."""
    import struct
    import zlib
    out, pos = bytearray(), 0
    while pos <= len(data):
        n = min(len(data) - pos, rng.choice([1, 100, 5000, 30000, 65280]))
        piece = data[pos: pos + n]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(piece) + comp.flush()
        bsize = 18 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, n)
        if n == 0:
            break        # (the empty block marks the end)
        pos += n
    return bytes(out)


def test_large_single_member_gzip_goes_through_the_parallel_inflater(KCT, tmp_path, monkeypatch, capfd):
    """A single-member FASTQ .gz of more than 4 MiB is inflated by several threads (csrc/parallel_inflate.h: entered at block boundaries
    found by search, the unknown 32 KiB in front of every piece resolved afterwards, length and CRC-32 verified); the table must be the
    oracle's -- and the same with the parallel inflater switched off, and for a file whose second half is CORRUPT (the CRC check sends
    it to the ordinary inflater, which reports the error)."""
    import gzip
    import os
    rng = random.Random(77)
    L, k = 150, 21
    genome = rand_dna(rng, 400_000, "ACGT")
    recs = [genome[i:i + L] for i in (rng.randrange(0, len(genome) - L) for _ in range(70_000))]
    recs += [rand_dna(rng, rng.choice([0, 20, 300]), "ACGTNacgt") for _ in range(300)]
    fq = "".join(f"@read{i} lane:1\n{s}\n+\n{''.join(rng.choice('FFFFFF:,#') for _ in s)}\n" for i, s in enumerate(recs)).encode()
    blob = gzip.compress(fq, 6)
    assert len(blob) > (4 << 20), len(blob)
    path = tmp_path / "big.fastq.gz"
    path.write_bytes(blob)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    monkeypatch.setenv("KCT_DEBUG", "1")
    dev = KCT(k, capacity=1_000_000)
    assert dev.consume_file(str(path)) == n_ref
    assert dev.last_file_records == len(recs)
    assert_same_table(dev, ref)
    err = capfd.readouterr().err
    assert "parallel inflate" in err and ": ok" in err.split("parallel inflate", 1)[1].split("\n", 1)[0], err[-600:]   # (it really took that route)
    monkeypatch.delenv("KCT_DEBUG")
    monkeypatch.setenv("KCT_NO_PARALLEL_GZIP", "1")
    dev = KCT(k, capacity=1_000_000)
    assert dev.consume_file(str(path)) == n_ref
    assert_same_table(dev, ref)
    monkeypatch.delenv("KCT_NO_PARALLEL_GZIP")
    # the STREAMING form (texts too large to inflate in one piece -- forced here by KCT_GZIP_WHOLE_MAX=0): windows of 1 MiB of compressed bytes, the
    # text parsed by several threads, the unfinished record carried into the next window; and the same reads as TWO members in one file
    monkeypatch.setenv("KCT_GZIP_WHOLE_MAX", "0")
    monkeypatch.setenv("KCT_GZIP_WINDOW", str(1 << 20))
    monkeypatch.setenv("KCT_DEBUG", "1")
    two = tmp_path / "two.fastq.gz"
    cut = fq.index(b"\n@read35000 ") + 1
    two.write_bytes(gzip.compress(fq[:cut], 6) + gzip.compress(fq[cut:], 1))
    # (KCT_GZIP_LEAD=0: no room in front of a window's text for the unfinished record of the window before -- the two are joined aside)
    for pth, lead in ((path, None), (two, None), (path, "0")):
        if lead is not None:
            monkeypatch.setenv("KCT_GZIP_LEAD", lead)
        dev = KCT(k, capacity=1_000_000)
        assert dev.consume_file(str(pth)) == n_ref
        assert dev.last_file_records == len(recs)
        assert_same_table(dev, ref)
        assert "through the parallel inflater" in capfd.readouterr().err
    monkeypatch.delenv("KCT_GZIP_LEAD")
    # FASTA records of 2.5 Mbases in lines of 70: longer than a window's text, so whole windows are carried forward until a record ends
    nrng = np.random.default_rng(78)
    long_recs = [np.frombuffer(b"ACGT", dtype=np.uint8)[nrng.integers(0, 4, size=2_500_000)].tobytes().decode() for _ in range(7)]
    fa = "".join(f">chr{i}\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n" for i, s in enumerate(long_recs)).encode()
    fa_blob = gzip.compress(fa, 6)
    assert len(fa_blob) > (4 << 20), len(fa_blob)
    (tmp_path / "long.fa.gz").write_bytes(fa_blob)
    ref_fa = OracleTable(k)
    n_fa = sum(ref_fa.consume(r) for r in long_recs)
    for lead in ("1048576", "0"):
        monkeypatch.setenv("KCT_GZIP_LEAD", lead)
        dev = KCT(k, capacity=40_000_000)
        assert dev.consume_file(str(tmp_path / "long.fa.gz")) == n_fa
        assert dev.last_file_records == len(long_recs)
        assert_same_table(dev, ref_fa)
    monkeypatch.delenv("KCT_GZIP_LEAD")
    monkeypatch.delenv("KCT_DEBUG"); monkeypatch.delenv("KCT_GZIP_WHOLE_MAX"); monkeypatch.delenv("KCT_GZIP_WINDOW")
    bad = bytearray(blob)
    bad[len(bad) * 3 // 4] ^= 0x21
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises((RuntimeError, ValueError, OSError)):
        KCT(k).consume_file(str(tmp_path / "bad.fastq.gz"))
    monkeypatch.setenv("KCT_GZIP_WHOLE_MAX", "0")     # ... and the streaming form meets the corruption in a window, or at the member's CRC
    with pytest.raises((RuntimeError, ValueError, OSError)):
        KCT(k).consume_file(str(tmp_path / "bad.fastq.gz"))
    monkeypatch.delenv("KCT_GZIP_WHOLE_MAX")


def test_two_threads_read_files_into_their_tables_at_the_same_time(KCT, tmp_path):
    """kct_consume_file's device calls are made by one long-lived thread per process; a second call that arrives while it is taken (another
    table, another caller thread -- ctypes releases the GIL) gets a thread of its own.  Both tables must be the oracle's, call after call."""
    import threading
    rng = random.Random(91)
    files, refs = [], []
    for j, k in enumerate((21, 31, 25)):
        recs = [rand_dna(rng, rng.choice([30, 150, 151, 4000]), "ACGTACGTN") for _ in range(3000)]
        path = tmp_path / f"t{j}.fa"
        _write_fasta(path, recs, width=80)
        ref = OracleTable(k)
        n_ref = sum(ref.consume(r) for r in recs)
        files.append((str(path), k, n_ref)); refs.append(ref)
    tables = [KCT(k, capacity=2_000_000) for _p, k, _n in files]
    errors = []

    def job(i):
        try:
            for _ in range(4):
                tables[i].clear()
                assert tables[i].consume_file(files[i][0]) == files[i][2]
        except Exception as e:   # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=job, args=(i,)) for i in range(len(files))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors and not any(th.is_alive() for th in threads), errors
    for tbl, ref in zip(tables, refs):
        assert_same_table(tbl, ref)


def test_consume_file_formats_match_oracle(KCT, tmp_path, monkeypatch):
    import gzip
    rng = random.Random(31)
    recs = [rand_dna(rng, rng.choice([0, 10, 21, 150, 151, 1000, 25000]), "ACGTACGTACGTNacgt") for _ in range(200)]
    k = 21
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    fa, fagz, fq = tmp_path / "a.fa", tmp_path / "a.fa.gz", tmp_path / "a.fq"
    _write_fasta(fa, recs)
    _write_fasta(fagz, recs, width=61, gz=True)
    with open(fq, "w") as f:
        for i, s in enumerate(recs):
            f.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
    # BGZF (bgzip's blocked gzip: members of <= 64 KiB announcing their size -- inflated by several threads) and a plain multi-member
    # gzip file (members found only by inflating: one inflater thread)
    fabgz, fqmm = tmp_path / "a.bgz.fa.gz", tmp_path / "a.multi.fq.gz"
    text = open(fa, "rb").read()
    with open(fabgz, "wb") as f:
        f.write(_bgzf(text, rng))
    with open(fqmm, "wb") as f:
        fqtext = open(fq, "rb").read()
        cut = len(fqtext) // 3
        for part in (fqtext[:cut], fqtext[cut: 2 * cut], fqtext[2 * cut:]):
            f.write(gzip.compress(part, 1))
    for path in (fa, fagz, fq, fabgz, fqmm):
        dev = KCT(k)
        assert dev.consume_file(str(path)) == n_ref, path
        assert dev.last_file_records == len(recs)
        assert_same_table(dev, ref)
    monkeypatch.setenv("KCT_FILE_SLOT", "65536")   # text slots of 64 KiB: records (up to 25 kbp) and lines across slot boundaries
    monkeypatch.setenv("KCT_GZIP_WHOLE_MAX", "0")  # (a small single-member file is otherwise inflated in one piece: here the streaming reader)
    for path, threads in ((fagz, "1"), (fabgz, "1"), (fabgz, "5"), (fqmm, "1")):
        monkeypatch.setenv("KCT_FILE_THREADS", threads)
        dev = KCT(k)
        assert dev.consume_file(str(path)) == n_ref, (path, threads)
        assert dev.last_file_records == len(recs)
        assert_same_table(dev, ref)
    monkeypatch.delenv("KCT_FILE_SLOT"); monkeypatch.delenv("KCT_FILE_THREADS"); monkeypatch.delenv("KCT_GZIP_WHOLE_MAX")
    monkeypatch.setenv("KCT_FILE_SEGMENT", "4096")   # the one-piece inflate's text, parsed by several threads in tiny segments
    monkeypatch.setenv("KCT_FILE_THREADS", "4")
    dev = KCT(k)
    assert dev.consume_file(str(fagz)) == n_ref and dev.last_file_records == len(recs)
    assert_same_table(dev, ref)
    monkeypatch.delenv("KCT_FILE_SEGMENT"); monkeypatch.delenv("KCT_FILE_THREADS")
    badgz = bytearray(open(fagz, "rb").read())
    badgz[len(badgz) // 2] ^= 0x55                  # a corrupt single-member file: an error from whichever reader meets it
    (tmp_path / "bad1.fa.gz").write_bytes(bytes(badgz))
    with pytest.raises((RuntimeError, ValueError, OSError)):
        KCT(k).consume_file(str(tmp_path / "bad1.fa.gz"))
    # BGZF slots are inflated AND parsed by several threads, the records across slot boundaries by the caller's thread: FASTQ whose
    # quality lines begin with '@' or '+' (a header is only a header when a '+' line follows its sequence line), empty records, CRLF,
    # and records far longer than a slot (no record start in most slots: everything goes through the stitching parser)
    recs2 = [rand_dna(rng, rng.choice([0, 1, 21, 60, 150, 151, 3000]), "ACGTACGTACGTNacgt") for _ in range(3000)] + [rand_dna(rng, 300_000, "ACGT")]
    rng.shuffle(recs2)
    ref2 = OracleTable(k)
    n_ref2 = sum(ref2.consume(r) for r in recs2)
    fq2, fa2 = tmp_path / "b.fq.gz", tmp_path / "b.fa.gz"
    qual = lambda i, n: (("@" if i % 3 == 0 else "+" if i % 3 == 1 else "I") + "@+I" * n)[:n]   # noqa: E731
    eol = lambda i: "\r\n" if i % 7 == 0 else "\n"                                              # noqa: E731
    fq2.write_bytes(_bgzf("".join(f"@r{i} x{eol(i)}{s}{eol(i)}+{'r%d' % i if i % 2 else ''}{eol(i)}{qual(i, len(s))}{eol(i)}" for i, s in enumerate(recs2)).encode(), rng))
    fa2.write_bytes(_bgzf("".join(f">r{i}\n" + "".join(s[j: j + 70] + "\n" for j in range(0, len(s), 70)) for i, s in enumerate(recs2)).encode(), rng))
    for slot, threads, nolib in (("65536", "1", False), ("65536", "7", False), ("131072", "3", True), (None, None, False)):
        if slot:
            monkeypatch.setenv("KCT_FILE_SLOT", slot); monkeypatch.setenv("KCT_FILE_THREADS", threads)
        if nolib:
            monkeypatch.setenv("KCT_NO_LIBDEFLATE", "1")      # (zlib's inflate: what a system without libdeflate.so.0 runs)
        for path in (fq2, fa2):
            dev = KCT(k)
            assert dev.consume_file(str(path)) == n_ref2, (path, slot, threads)
            assert dev.last_file_records == len(recs2)
            assert_same_table(dev, ref2)
        for v in ("KCT_FILE_SLOT", "KCT_FILE_THREADS", "KCT_NO_LIBDEFLATE"):
            monkeypatch.delenv(v, raising=False)
    bad = bytearray(open(fabgz, "rb").read())
    bad[len(bad) // 2] ^= 0x55                      # a corrupt block: an error, not a short count
    (tmp_path / "bad.fa.gz").write_bytes(bytes(bad))
    with pytest.raises((RuntimeError, ValueError, OSError)):
        KCT(k).consume_file(str(tmp_path / "bad.fa.gz"))
    # records longer than a staging chunk are cut with a (k-1)-base overlap: force tiny chunks
    monkeypatch.setenv("KCT_FILE_CHUNK", "4096")
    dev = KCT(k)
    assert dev.consume_file(str(fa)) == n_ref
    assert_same_table(dev, ref)
    monkeypatch.delenv("KCT_FILE_CHUNK")
    # error mode goes through the per-record semantics
    dev2, ref2 = KCT(k), OracleTable(k)
    err_ref = None
    for r in recs:
        try:
            ref2.consume(r, skip_bad_kmers=False)
        except ValueError as e:
            err_ref = str(e)
            break
    with pytest.raises(ValueError) as ei:
        dev2.consume_file(str(fa), skip_bad_kmers=False)
    assert str(ei.value) == err_ref
    assert_same_table(dev2, ref2)
    with pytest.raises(OSError):
        KCT(k).consume_file(str(tmp_path / "missing.fa"))


def test_consume_file_parallel_parsers_match_oracle(KCT, tmp_path, monkeypatch):
    """Plain files are mapped and cut into segments that several parser threads take in turn: every record must be
    parsed exactly once wherever the cuts fall (tiny segments put cuts inside headers, sequences, quality lines)."""
    rng = random.Random(77)
    recs = [rand_dna(rng, rng.choice([0, 1, 20, 21, 22, 150, 700, 5000]), "ACGTACGTACGTNacgt") for _ in range(400)]
    k = 21
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    fa, facr, fq = tmp_path / "p.fa", tmp_path / "p_crlf.fa", tmp_path / "p.fq"
    _write_fasta(fa, recs, width=60)
    with open(facr, "wb") as f:                                    # CRLF line ends, blank lines, '>' inside headers
        f.write(b"\r\n\r\n")
        for i, s in enumerate(recs):
            f.write(b">r%d >not a record > really\r\n" % i)
            for j in range(0, len(s), 33):
                f.write(s[j:j + 33].encode() + b"\r\n")
            if i % 7 == 0:
                f.write(b"\r\n")
    with open(fq, "w") as f:                                       # quality lines that start with '@' and '+'
        for i, s in enumerate(recs):
            q = "".join(rng.choice("@+I#>") for _ in s)
            f.write(f"@r{i} @x\n{s}\n+r{i}\n{q}\n")
    for seg, threads in (("64", "5"), ("1000", "3"), ("40000", "8")):
        monkeypatch.setenv("KCT_FILE_SEGMENT", seg)
        monkeypatch.setenv("KCT_FILE_THREADS", threads)
        monkeypatch.setenv("KCT_FILE_CHUNK", "30000")
        for path in (fa, facr, fq):
            dev = KCT(k)
            assert dev.consume_file(str(path)) == n_ref, (path, seg)
            assert dev.last_file_records == len(recs), (path, seg)
            assert_same_table(dev, ref)
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>r\nACGT\n")
    with pytest.raises(RuntimeError, match="neither FASTA nor FASTQ"):
        KCT(k).consume_file(str(bad))


def test_multi_chunk_stream_and_large_table_paths(KCT):
    """Two scale checks that need no oracle run:
    (1) a 302 MB record stream crosses the 2^28-position launch chunk inside one consume_device call;
        consuming it whole must equal consuming its two halves (same table, bit for bit);
    (2) BASELINE config C3's shape (k=31, table too large for the partitioned path) on the direct
        atomic path: n, sum_counts, len agree with the partitioned path's result on a table that fits."""
    import torch

    from oxli_amd import _lib
    lib = _lib.load()
    G, L, N = 20_000_000, 150, 2_000_000
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 7, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 99, stream) == 0
    torch.cuda.synchronize()
    assert r.numel() > (1 << 28)
    k = 31
    whole = KCT(k, capacity=5_000_000)            # 2^23 slots: partitioned path, grows while counting
    n = whole.consume_device(r.data_ptr(), r.numel(), N * L)
    assert n == N * (L - k + 1) == whole.sum_counts
    halves = KCT(k, capacity=5_000_000)
    half = (N // 2) * (L + 1)
    n2 = halves.consume_device(r.data_ptr(), half, (N // 2) * L) + halves.consume_device(r.data_ptr() + half, r.numel() - half, (N // 2) * L)
    assert n2 == n
    wk, wc = whole.dump_arrays(1)
    hk, hc = halves.dump_arrays(1)
    assert np.array_equal(wk, hk) and np.array_equal(wc, hc)
    big = KCT(k, capacity=40_000_000)             # 2^26 slots = 1 GiB: 8192 blocks -> direct atomic path
    big.set_path("auto")
    assert big.capacity == 1 << 26
    assert big.consume_device(r.data_ptr(), r.numel(), N * L) == n
    assert len(big) == len(whole) and big.sum_counts == n
    bk, bc = big.dump_arrays(1)
    assert np.array_equal(bk, wk) and np.array_equal(bc, wc)


def test_partitioned_path_on_a_128_GiB_table(KCT):
    """2^33 slots = 128 GiB of table on one GPU (288 GB of HBM): 2^20 blocks, both partition levels at their
    full fan-out of 1024.  Same reads into a small table must give the same (hash, count) set."""
    import torch

    from oxli_amd import _lib
    free, _total = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs 200 GiB of free HBM")
    lib = _lib.load()
    G, L, N, k = 3_000_000, 150, 2_000_000, 31
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 17, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, 23, stream) == 0
    torch.cuda.synchronize()
    small = KCT(k, capacity=G)
    n = small.consume_device(r.data_ptr(), r.numel(), N * L)
    assert n == N * (L - k + 1)
    sk, sc = small.dump_arrays(1)
    del small
    big = KCT(k, capacity=5_000_000_000)          # 5e9 / 0.65 -> 2^33 slots
    assert big.capacity == 1 << 33
    big.set_path("partitioned")
    big.profile(True)
    assert big.consume_device(r.data_ptr(), r.numel(), N * L) == n
    prof = big.profile_read()
    assert "repartition_kernel" in prof and "count_windows_kernel" not in prof, prof   # really the two-level path
    assert len(big) == sk.size and big.sum_counts == n
    bk, bc = big.dump_arrays(1)
    assert np.array_equal(bk, sk) and np.array_equal(bc, sc)
    assert big.get_hash(int(sk[12345])) == int(sc[12345])


# ---- deferred mode (the default): per-record consume() buffered on the host, counted in one device pass ------------
@pytest.mark.parametrize("k", [4, 21, 33, 70])
def test_deferred_consume_matches_oracle(KCT, k):
    rng = random.Random(900 + k)
    dev, ref = KCT(k), OracleTable(k)      # default-constructed: deferred mode is the library's default
    plain = KCT(k, deferred=False)         # every call a device pass of its own
    for i in range(400):
        kind = rng.random()
        if kind < 0.1:
            s = rand_dna(rng, rng.randrange(0, k))                      # shorter than k
        elif kind < 0.4:
            s = rand_dna(rng, rng.randrange(k, 4 * k + 40), "ACGTNacgtn-")  # bad bytes, lower case
        else:
            s = rand_dna(rng, rng.randrange(k, 300))
        n = dev.consume(s, skip_bad_kmers=True)
        assert n == ref.consume(s, skip_bad_kmers=True) == plain.consume(s, skip_bad_kmers=True), (i, s)
        if i % 97 == 0:  # a read in the middle must see everything consumed so far
            probe = rand_dna(rng, k)
            assert dev.get(probe) == ref.get(probe)
            assert len(dev) == len(ref)
    assert_same_table(dev, ref)


def test_deferred_error_mode_and_clear(KCT):
    k = 9
    dev, ref = KCT(k), OracleTable(k)
    dev.consume("ACGTACGTACGTTTGA"); ref.consume("ACGTACGTACGTTTGA")
    with pytest.raises(ValueError) as e_dev:
        dev.consume("ACGTACGTACNTACGTACGT", skip_bad_kmers=False)
    with pytest.raises(ValueError) as e_ref:
        ref.consume("ACGTACGTACNTACGTACGT", skip_bad_kmers=False)
    assert str(e_dev.value) == str(e_ref.value)
    assert_same_table(dev, ref)      # the windows before the bad one were counted, like the reference
    dev.consume("GGGGGGGGGGGGG")     # buffered ...
    dev.clear()                      # ... and forgotten
    assert len(dev) == 0 and dev.sum_counts == 0
    dev.consume("GGGGGGGGGGGGG")
    dev.sync()                       # kct_sync: an explicit flush point
    dev.set_deferred(False)          # switching off also counts what is buffered
    assert dev.get("GGGGGGGGG") == 5
    other = KCT(k)
    other.consume("GGGGGGGGGGGGG")
    dev.add(other)                   # src's buffered records are part of src
    assert dev.get("GGGGGGGGG") == 10


def test_deferred_buffer_rollover(KCT):
    # more than the 64 MiB pending buffer: flushes happen mid-stream, totals stay exact
    k = 31
    rng = np.random.default_rng(5)
    read = "".join("ACGT"[i] for i in rng.integers(0, 4, 1 << 20))
    dev = KCT(k)
    total = 0
    for i in range(70):
        total += dev.consume(read[i:] if i else read)
    per = [len(read) - i - k + 1 for i in range(70)]
    assert total == sum(per) == dev.sum_counts
    assert dev.get(read[100:100 + k]) >= 70
