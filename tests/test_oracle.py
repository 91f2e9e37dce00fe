"""Pins the CPU oracle (oracle/kct_oracle.c) to the reference's own known answers.

Everything here runs without a GPU.  The oracle is only trusted by the -m gpu parity tests
because these pass.
"""
import hashlib

import numpy as np
import pytest

import oracle
from oracle import OracleTable


def py_murmur64(data: bytes, seed: int = 42) -> int:
    """Independent pure-Python MurmurHash3_x64_128 (h1), for a three-way check."""
    M = (1 << 64) - 1
    c1, c2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
    rotl = lambda x, r: ((x << r) | (x >> (64 - r))) & M
    h1 = h2 = seed
    nb = len(data) // 16
    for i in range(nb):
        k1 = int.from_bytes(data[16 * i:16 * i + 8], "little")
        k2 = int.from_bytes(data[16 * i + 8:16 * i + 16], "little")
        k1 = (k1 * c1) & M; k1 = rotl(k1, 31); k1 = (k1 * c2) & M; h1 ^= k1
        h1 = rotl(h1, 27); h1 = (h1 + h2) & M; h1 = (h1 * 5 + 0x52DCE729) & M
        k2 = (k2 * c2) & M; k2 = rotl(k2, 33); k2 = (k2 * c1) & M; h2 ^= k2
        h2 = rotl(h2, 31); h2 = (h2 + h1) & M; h2 = (h2 * 5 + 0x38495AB5) & M
    tail = data[16 * nb:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & M; k2 = rotl(k2, 33); k2 = (k2 * c1) & M; h2 ^= k2
    if len(tail) > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & M; k1 = rotl(k1, 31); k1 = (k1 * c2) & M; h1 ^= k1
    h1 ^= len(data); h2 ^= len(data)
    h1 = (h1 + h2) & M; h2 = (h2 + h1) & M

    def fmix(k):
        k ^= k >> 33; k = (k * 0xFF51AFD7ED558CCD) & M
        k ^= k >> 33; k = (k * 0xC4CEB9FE1A85EC53) & M
        k ^= k >> 33
        return k

    h1, h2 = fmix(h1), fmix(h2)
    return (h1 + h2) & M


def test_reference_hash_kats(kats):
    for e in kats["hashes"]:
        assert oracle.hash_kmer(e["kmer"]) == e["hash"], e


def test_reference_window_hash_lists(kats):
    for e in kats["window_hashes"]:
        hs, stopped = oracle.seq_to_hashes(e["seq"], e["k"], force=True)
        assert hs.tolist() == e["hashes"], e
        assert not stopped


def test_canonical_murmur_vectors(murmur_vectors):
    for v in murmur_vectors:
        b = v["bytes"].encode()
        assert oracle.murmur64(b) == v["h1"]
        assert py_murmur64(b) == v["h1"]


def test_reference_consume_facts(kats):
    for e in kats["consume"]:
        t = OracleTable(e["k"])
        assert t.consume(e["seq"]) == e["n"], e
        for kmer, c in e.get("get", {}).items():
            assert t.get(kmer) == c, (e, kmer)
        if "len" in e:
            assert len(t) == e["len"]
        if "consumed" in e:
            assert t.consumed == e["consumed"]
        if "sum_counts" in e:
            assert t.sum_counts == e["sum_counts"]
        assert t.consumed == len(e["seq"])


def test_reference_consume_errors(kats):
    for e in kats["consume_errors"]:
        t = OracleTable(e["k"])
        with pytest.raises(ValueError, match=f"bad k-mer encountered at position {e['position']}$"):
            t.consume(e["seq"], skip_bad_kmers=False)
        assert t.consumed == 0  # lib.rs:593-596 returns before `consumed` is bumped (lib.rs:604)
        assert t.sum_counts == e["position"]  # k-mers before the bad one stay counted


def test_reference_add_facts(kats):
    for e in kats["add"]:
        a, b = OracleTable(e["k"]), OracleTable(e["k"])
        if e["a"]:
            a.consume(e["a"])
        b.consume(e["b"])
        assert a.add(b) == (e["counts_added"], e["new_keys"]), e
        assert a.sum_counts == e["sum_counts"]
        if "len" in e:
            assert len(a) == e["len"]
        assert a.consumed == len(e["a"]) + len(e["b"])
    with pytest.raises(ValueError):
        OracleTable(5).add(OracleTable(6))


def test_reference_stress_atgc(kats):
    s = kats["stress"]
    seq = s["unit"] * s["repeat"]
    a, b = OracleTable(s["k"]), OracleTable(s["k"])
    assert a.consume(seq) == s["n"]
    assert b.consume(seq) == s["n"]
    assert a.add(b) == (s["add_counts_added"], s["add_new_keys"])
    assert a.sum_counts == s["sum_counts_after_add"]


def test_point_api_matches_reference_tests():
    # test_basic.py:14-32, 112-125, 128-166 (reference)
    t = OracleTable(4)
    assert t.get("ATCG") == 0 and t.count("ATCG") == 1 and t.get("ATCG") == 1
    with pytest.raises(ValueError):
        OracleTable(3).count("ATCG")
    with pytest.raises(ValueError):
        OracleTable(3).get("ATCG")
    kmer = "TAAACCCTAACCCTAACCCTAACCCTAACCC"
    t = OracleTable(31)
    h = t.hash_kmer(kmer)
    assert t.get(kmer) == 0 and t.count(kmer) == 1 and t.count(kmer) == 2 and t.get(kmer) == 2
    assert t.count_hash(h) == 3 and t.get(kmer) == 3
    assert t.consumed == 62  # count adds k each time (lib.rs:153); count_hash adds nothing
    t = OracleTable(3)
    for k in ["AAA", "TTT", "AAC"]:
        t.count(k)
    hs = [t.hash_kmer("AAA"), t.hash_kmer("AAC"), t.hash_kmer("GGG")]
    assert t.get_hash_array(hs) == [2, 1, 0] and t.get_hash_array(hs[::-1]) == [0, 1, 2]
    with pytest.raises(RuntimeError):
        t.hash_kmer("AANA"[:3].replace("A", "N", 1))  # invalid DNA of the right length


def test_example_fa(kats, example_seq, example_digests):
    for e in kats["example_fa"]:
        t = OracleTable(e["k"])
        assert t.consume(example_seq) == e["n"]
        for kmer, c in e.get("get", {}).items():
            assert t.get(kmer) == c
        if "then_consume_skip" in e:
            x = e["then_consume_skip"]
            assert t.consume(x["seq"]) == x["n"]
            for kmer, c in x["get"].items():
                assert t.get(kmer) == c
    for k, d in example_digests["k"].items():
        t = OracleTable(int(k))
        assert t.consume(example_seq) == d["n"]
        keys, counts = t.dump_arrays()
        assert len(t) == d["distinct"] and int(counts.max()) == d["max"] and t.consumed == d["consumed"]
        assert np.all(keys[:-1] < keys[1:])
        tsv = "".join(f"{h}\t{c}\n" for h, c in zip(keys.tolist(), counts.tolist()))
        assert hashlib.sha256(tsv.encode()).hexdigest() == d["sha256_dump_sortkeys_tsv"]
        hs, _ = oracle.seq_to_hashes(example_seq, int(k))
        assert [int(v) for v in hs[:3]] == d["first3"]


def test_window_semantics_edge_cases():
    # short / empty input: zero windows, consumed still grows (test_attr.py:76-82 in the reference)
    t = OracleTable(21)
    assert t.consume("") == 0 and t.consume("ACGT") == 0 and t.consumed == 4 and len(t) == 0
    # every window overlapping a bad byte is skipped; bytes >= 0x80 are just bad bases
    t = OracleTable(4)
    assert t.consume("ACGTNACGT") == 2
    assert t.consume("ACGT\u00e9ACGT") == 2 and t.consumed == 9 + 10  # len is a BYTE count (lib.rs:548)
    # the validity rule is per byte after ASCII upper-casing: IUPAC codes are invalid
    for bad in "NRYKMSWBDHVnx-*":
        hs, _ = oracle.seq_to_hashes("ACG" + bad + "ACGT", 4)
        assert hs.tolist()[:4] == [0, 0, 0, 0] and hs[4] != 0
    # palindrome: forward == reverse complement
    assert oracle.hash_kmer("ACGT") == oracle.murmur64(b"ACGT")
    # canonical = bytewise minimum of forward and reverse complement
    import random
    rng = random.Random(7)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in (1, 2, 15, 16, 17, 21, 31, 32, 33, 51, 63, 64, 65, 127, 200, 255):
        s = "".join(rng.choice("ACGT") for _ in range(k))
        rc = "".join(comp[c] for c in reversed(s))
        assert oracle.hash_kmer(s) == oracle.hash_kmer(rc) == py_murmur64(min(s, rc).encode())
        assert oracle.hash_kmer(s.lower()) == oracle.hash_kmer(s)


def test_synthetic_generator_is_counter_based():
    g = oracle.synth_genome(5000)
    assert set(np.unique(g).tolist()) <= set(b"ACGT")
    assert bytes(g[:8]) == bytes(b"ACGT"[oracle.mix64(42 + j) & 3] for j in range(8))
    a = oracle.synth_reads(g, 0, 64, 150)
    b = oracle.synth_reads(g, 32, 32, 150)
    assert np.array_equal(a[32:], b)  # read i depends only on i
    assert np.all(a[:, 150] == ord("\n"))
    i = 5
    start = oracle.mix64(1337 + 2 * i) % (5000 - 150 + 1)
    strand = oracle.mix64(1337 + 2 * i + 1) & 1
    fw = bytes(g[start:start + 150])
    if strand:
        fw = fw[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    assert bytes(a[i, :150]) == fw


def test_baseline_port_equals_single_table():
    g = oracle.synth_genome(20000)
    reads = oracle.synth_reads(g, 0, 2000, 150)
    ref = OracleTable(21)
    n = sum(ref.consume(reads[i, :150]) for i in range(reads.shape[0]))
    for threads in (1, 3):
        t, kmers, secs = oracle.baseline_consume(reads, 150, 21, threads, native=False)
        assert kmers == n == 2000 * 130 and secs > 0
        k1, c1 = t.dump_arrays()
        k0, c0 = ref.dump_arrays()
        assert np.array_equal(k0, k1) and np.array_equal(c0, c1)
        assert t.consumed == ref.consumed


def test_shardset_equals_the_single_table_oracle():
    """oracle.ShardSet (the full-size checker of tests/test_gpu_scale.py): its digest and its pair comparison against
    OracleTable over the same reads -- in memory and generated on the fly, with and without the error model."""
    import oracle
    G, R, L = 20_000, 3_000, 150
    genome = oracle.synth_genome(G, 42)
    for k, model in ((21, {}), (31, dict(sub_ppm=10_000, n_ppm=5_000)), (51, dict(sorted_total=R)), (5, dict(n_ppm=100_000))):
        reads = oracle.synth_reads_ex(genome, 7, R, L, 1337, **model)
        ref = oracle.OracleTable(k)
        n = sum(ref.consume(reads[i, :L]) for i in range(R))
        rk, rc = ref.dump_arrays()
        want = {"len": len(ref), "sum_counts": ref.sum_counts, "n": n, "consumed": R * L, "min": int(rc.min()), "max": int(rc.max()),
                "sum_hc": int(np.sum(rk * rc, dtype=np.uint64)), "xor_hc": int(np.bitwise_xor.reduce(rk * rc)),
                "sum_sq": int(np.sum(rc * rc, dtype=np.uint64))}
        for ss in (oracle.ShardSet(k, L, reads=reads, threads=3, batch=500),
                   oracle.ShardSet(k, L, genome=genome, first=7, nreads=R, threads=4, batch=1000, **model)):
            assert ss.digest() == want
            assert ss.mismatches(rk, rc) == 0
            bad = rc.copy()
            bad[5] += 1
            assert ss.mismatches(rk, bad) == 1 and ss.mismatches(rk ^ np.uint64(1), rc) == rk.size
            assert ss.get_hash(int(rk[0])) == int(rc[0]) and ss.get_hash(12345) == 0


def test_error_model_of_the_synthetic_stream():
    """include/kct_synth.h: substitutions never restore the true base, N and substitution rates, position-sorted starts; all
    rates zero = the plain stream."""
    import oracle
    G, R, L = 50_000, 4_000, 150
    genome = oracle.synth_genome(G, 42)
    plain = oracle.synth_reads(genome, 3, R, L)
    assert np.array_equal(oracle.synth_reads_ex(genome, 3, R, L), plain)
    sub = oracle.synth_reads_ex(genome, 3, R, L, sub_ppm=10_000)
    diff = sub != plain
    assert 0.008 < diff[:, :L].mean() < 0.012 and not diff[:, L].any()
    assert set(np.unique(sub[:, :L]).tolist()) <= set(b"ACGT")
    both = oracle.synth_reads_ex(genome, 3, R, L, sub_ppm=5_000, n_ppm=20_000)
    isn = both[:, :L] == ord("N")
    assert 0.018 < isn.mean() < 0.022
    changed = (both[:, :L] != plain[:, :L]) & ~isn
    assert 0.004 < changed.mean() < 0.006
    srt = oracle.synth_reads_ex(genome, 0, R, L, sorted_total=R)
    text = genome.tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    starts = []
    for i in range(0, R, 97):
        rec = srt[i, :L].tobytes()
        pos = text.find(rec)
        if pos < 0:
            pos = text.find(rec.translate(comp)[::-1])
        starts.append(pos)
        assert pos == i * (G - L + 1) // R
    assert starts == sorted(starts)
