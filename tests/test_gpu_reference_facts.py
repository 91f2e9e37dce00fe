"""Every fact the reference's own Python tests assert (`src/python/tests/test_*.py`, 114 tests in 12 files), re-expressed
against this engine: one test per reference file, each fact with the line it comes from.  Inputs and expected values are
the reference's; the code is not (tables are built by the helpers below, facts are checked in bulk).  Hashes and counts
always come from the device table."""
import gzip
import json
import math

import pytest

pytestmark = pytest.mark.gpu

AAAA, AATT, GGGG = 17832910516274425539, 382727017318141683, 73459868045630124  # test_dump.py:13-17
K31 = "TAAACCCTAACCCTAACCCTAACCCTAACCC"                                          # test_basic.py:26
K16 = "ACGTACGTACGTACGT"                                                          # test_attr.py:63


@pytest.fixture(scope="module")
def KCT():
    from oxli_amd import KmerCountTable
    return KmerCountTable


def counted(KCT, k, kmers, **kw):
    t = KCT(k, **kw)
    for kmer in kmers:
        t.count(kmer)
    return t


def assigned(KCT, k, pairs):
    t = KCT(ksize=k)
    for kmer, c in pairs:
        t[kmer] = c
    return t


def test_basic_py(KCT):
    t = KCT(4)
    assert (t.get("ATCG"), t.count("ATCG"), t.get("ATCG")) == (0, 1, 1)                       # :14-22
    t = KCT(ksize=31)
    h = t.hash_kmer(K31)
    assert (t.get_hash(h), t.count_hash(h), t.get_hash(h)) == (0, 1, 1)                       # :25-32
    t = counted(KCT, 3, ["AAA", "TTT", "AAC"])
    assert t.hash_kmer("AAA") == t.hash_kmer("TTT")                                           # :35-40
    t3 = KCT(3)
    for call in (t3.count, t3.get):                                                           # :43-52
        with pytest.raises(ValueError):
            call("ATCG")
    t = KCT(4)
    assert t.consume("ATCG") == 1 and t.get("ATCG") == 1                                      # :55-60
    t = KCT(4)
    assert t.consume("ATCGG") == 2 and [t.get(x) for x in ("ATCG", "TCGG", "CCGA")] == [1, 1, 1]   # :63-70
    with pytest.raises(ValueError, match="bad k-mer encountered at position 2"):              # :73-79
        KCT(4).consume("ATCGGX", skip_bad_kmers=False)
    with pytest.raises(ValueError, match="bad k-mer encountered at position 0"):              # :82-88
        KCT(4).consume("XATCGG", skip_bad_kmers=False)
    for kwargs in ({"skip_bad_kmers": True}, {}):                                             # :91-108 (skipping is the default)
        t = KCT(4)
        t.consume("XATCGG", **kwargs)
        assert [t.get(x) for x in ("ATCG", "TCGG", "CCGA")] == [1, 1, 1]
    t = KCT(ksize=31)                                                                         # :111-125
    h = t.hash_kmer(K31)
    assert (t.get(K31), t.count(K31), t.count(K31), t.get(K31), t.count_hash(h), t.get(K31)) == (0, 1, 2, 2, 3, 3)
    t = counted(KCT, 3, ["AAA", "TTT", "AAC"])                                                # :128-146
    assert [t.get_hash(t.hash_kmer(x)) for x in ("AAA", "AAC", "AAG")] == [2, 1, 0]
    keys = [t.hash_kmer(x) for x in ("AAA", "AAC", "GGG")]                                    # :149-165
    assert t.get_hash_array(keys) == [2, 1, 0] and t.get_hash_array(keys[::-1]) == [0, 1, 2]


def test_attr_py(KCT):
    from oxli_amd import VERSION
    t = counted(KCT, 3, ["AAA", "TTT", "AAC"])
    assert set(t.hashes) == {t.hash_kmer(x) for x in ("AAA", "TTT", "AAC")}                   # :13-26
    assert KCT(ksize=31).version == VERSION == "0.3.0"                                        # :41-52 (Cargo.toml:3)
    assert KCT(ksize=31).consumed == 0                                                        # :55-57
    t = KCT(ksize=16)
    t.count(K16)
    assert t.consumed == 16                                                                   # :60-65
    t = KCT(ksize=16)
    t.consume("ACGTACGXACGTACGT", skip_bad_kmers=True)
    assert t.consumed == 16                                                                   # :68-73
    t = KCT(ksize=16)
    t.count(K16); t.consume("GCTAGCTAGCTA")
    assert t.consumed == 28                                                                   # :76-83
    t = KCT(ksize=16)
    assert t.sum_counts == 0                                                                  # :86-88
    t.count(K16)
    assert t.sum_counts == 1                                                                  # :91-96
    t = KCT(ksize=16)
    t.consume(K16 + "A")
    assert t.sum_counts == 2                                                                  # :99-104
    t = KCT(ksize=16)
    t.count(K16); t.consume(K16 + "A")
    assert t.sum_counts == 3                                                                  # :107-113


def test_dunders_py(KCT):
    t = KCT(ksize=16)
    assert len(t) == 0 and list(t) == []                                                      # :8-10, :50-60
    t.count(K16)
    assert len(t) == 1                                                                        # :13-16
    t.count(K16); t.count("C" * 16); t.consume("GCTAGCTAGCTA")
    assert len(t) == 2                                                                        # :19-28
    items = list(counted(KCT, 3, ["AAA", "TTT", "AAC"]))
    assert len(items) == 2 and 2 in [c for _, c in items] and 6579496673972597301 in [h for h, _ in items]   # :31-47
    t = KCT(ksize=16)
    t[K16] = 5
    assert t[K16] == 5 == t.get(K16) and t["C" * 16] == 0                                     # :63-86
    t = KCT(ksize=16)
    t.count(K16)
    t[K16] = 5
    assert t.get(K16) == 5
    t[K16] = 10
    assert t[K16] == 10                                                                       # :89-97


def test_add_py(KCT, capfd):
    def pair(a_seq, b_seq, k=5, **kw):
        a, b = KCT(k, **kw), KCT(k, **kw)
        if a_seq:
            a.consume(a_seq)
        if b_seq:
            b.consume(b_seq)
        return a, b
    a, b = pair("ATGCATGCA", "ATGCATGCA")
    assert a.add(b) == (5, 0) and a.sum_counts == 10                                          # :6-16
    a, b = pair("ATGCATGCA", "TGCATGCATGG")
    assert a.add(b) == (7, 1) and len(a) == 3 and a.sum_counts == 12                          # :19-34
    with pytest.raises(ValueError):
        KCT(5).add(KCT(6))                                                                    # :37-43
    a, b = pair("", "")
    assert a.add(b) == (0, 0) and a.sum_counts == 0                                           # :46-55
    a, b = pair("", "ATGCATGCA")
    assert a.add(b) == (5, 2) and len(a) == 2 and a.sum_counts == 5                           # :58-70
    a, b = pair("ATGCA", "TGCAT")
    before = a.consumed
    a.add(b)
    assert a.consumed == before + b.consumed                                                  # :73-84
    for s1, s2 in ((True, True), (True, False), (False, True), (False, False)):               # :87-109
        a, b = KCT(5, store_kmers=s1), KCT(5, store_kmers=s2)
        a.consume("ATGCA"); b.consume("GGCAT")
        capfd.readouterr()
        assert a.add(b) == (1, 1)
        err = capfd.readouterr().err
        if s1 and not s2:
            assert "Warning: Incoming table does not store k-mers" in err
        if s1 and s2:
            assert a.dump_kmers(sortkeys=True) == [("ATGCA", 1), ("ATGCC", 1)]
    a, b = pair("ATGC" * 100000, "ATGC" * 100000)
    assert a.add(b) == (399996, 0) and a.sum_counts == 799992                                 # :112-125
    a, b = pair("ATGCA", "TGCAT")
    c = KCT(5)
    c.consume("GCATG")
    a.add(b)
    assert a.add(c) == (1, 1) and a.sum_counts == 3                                           # :128-141


def test_canonicalization_py(KCT):
    t = KCT(ksize=4, store_kmers=True)
    assert [t.canon(x) for x in ("AAAA", "TTTT", "ATCG", "CGAT")] == ["AAAA", "AAAA", "ATCG", "ATCG"]   # :6-14
    t.count("TTTT"); t.count("TTTT")
    h = t.hash_kmer("TTTT")
    assert t.unhash(h) == "AAAA" and t.get_hash(h) == 2                                       # :17-30
    for bad in ("AAA", "AAAAA"):                                                              # :33-47
        with pytest.raises(ValueError, match="kmer size does not match count table ksize"):
            t.canon(bad)
    assert t.canon("gggg") == "CCCC"                                                          # :57-58
    for bad in ("ATXG", "aTbG"):                                                              # :61-69
        with pytest.raises(ValueError, match="kmer contains invalid characters"):
            t.canon(bad)


def test_remove_py(KCT):
    def fixture():  # :7-19: AAAA/TTTT = 2, ATAT = 1, CCCC/GGGG = 3
        return counted(KCT, 4, ["AAAA", "CCCC", "ATAT", "GGGG", "TTTT", "CCCC"])
    t = fixture()
    t.drop("GGGG")
    assert t.get("GGGG") == 0
    t.drop("AAAA")
    assert t.get("AAAA") == 0 and t.get("TTTT") == 0
    t.drop("GGGA")
    assert t.get("GGGA") == 0                                                                 # :22-39
    t = fixture()
    h = t.hash_kmer("CCCC")
    t.drop_hash(h)
    assert (t.get_hash(h), t.get("CCCC"), t.get("GGGG")) == (0, 0, 0)
    t.drop_hash(999999999)
    assert t.get_hash(999999999) == 0                                                         # :42-61
    t = fixture()
    assert t.mincut(3) == 2 and t.get("GGGG") == 3
    assert t.mincut(10) == 1 and len(t.hashes) == 0                                           # :64-79
    t = fixture()
    assert t.maxcut(2) == 1 and t.get("GGGG") == 0 and t.get("AAAA") == 2
    assert t.maxcut(10) == 0 and len(t.hashes) == 2
    assert t.maxcut(0) == 2 and len(t.hashes) == 0                                            # :82-108


def test_histo_py(KCT):
    t = KCT(ksize=4)
    assert t.min == 0 and t.max == 0                                                          # :13-26
    assert t.histo(zero=False) == [] and t.histo(zero=True) == [(0, 0)]                       # :51-68
    t.count("AAAA"); t.count("TTTT"); t.consume("CCCCCC")
    assert t.min == 2                                                                         # :29-37
    t = counted(KCT, 4, ["AAAA", "TTTT", "CCCC"])
    assert t.max == 2                                                                         # :40-48
    t = counted(KCT, 4, ["AAAA", "AAAA", "TTTT", "CCCC"])
    assert t.histo(zero=False) == [(1, 1), (3, 1)]                                            # :71-84
    assert t.histo(zero=True) == [(0, 0), (1, 1), (2, 0), (3, 1)]                             # :87-105
    t = counted(KCT, 4, ["AAAA"] * 5)
    assert t.histo(zero=True) == [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 1)]             # :108-125


def test_setops_py(KCT):
    a, b = counted(KCT, 3, ["AAA", "AAC"]), counted(KCT, 3, ["AAC", "AAG"])
    sa, sb = set(a.hashes), set(b.hashes)
    assert a.union(b) == sa | sb == a.__or__(b)                                               # :8-15, :54-70
    assert a.intersection(b) == sa & sb == a.__and__(b) and len(sa & sb) == 1                 # :18-27
    assert a.difference(b) == sa - sb == a.__sub__(b)                                         # :30-37
    assert a.symmetric_difference(b) == sa ^ sb == a.__xor__(b)                               # :40-51


def test_metrics_py(KCT):
    def cos(u, v):
        return sum(x * y for x, y in zip(u, v)) / (math.sqrt(sum(x * x for x in u)) * math.sqrt(sum(y * y for y in v)))
    names = ["AAAA", "AATT", "GGGG", "CCAA", "ATTG"]
    cases = [([5, 3, 1, 4, 6], [5, 3, 1, 4, 6]),                                              # :12-49
             ([4, 3, 1, 4, 6], [5, 3, 1, 4, 0])]                                              # :52-84
    for u, v in cases:
        a, b = assigned(KCT, 4, zip(names, u)), assigned(KCT, 4, zip(names, v))
        assert math.isclose(a.cosine(b), cos(u, v), rel_tol=1e-5) and math.isclose(b.cosine(a), cos(u, v), rel_tol=1e-5)
    a, b = assigned(KCT, 4, [("AAAA", 5), ("TTTG", 10)]), KCT(ksize=4)                        # :87-119
    assert a.cosine(b) == 0.0
    b["ATTG"] = 1
    assert a.cosine(b) == 0.0 == b.cosine(a)
    assert KCT(ksize=4).cosine(KCT(ksize=4)) == 0.0                                           # :122-132
    a = assigned(KCT, 4, [("AATT", 3), ("GGGG", 1), ("CCAA", 4), ("ATTG", 0), ("AGAT", 0)])   # :135-170
    b = assigned(KCT, 4, [("AAAA", 5), ("AATT", 4), ("GGGG", 1), ("CCAA", 4), ("ATTG", 1)])
    want = cos([0, 3, 1, 4, 0, 0], [5, 4, 1, 4, 1, 0])
    assert math.isclose(a.cosine(b), want, rel_tol=1e-5) and math.isclose(b.cosine(a), want, rel_tol=1e-5)
    same = [("AAAA", 5), ("TTTC", 2), ("AATT", 3), ("GGGG", 1)]
    assert assigned(KCT, 4, same).jaccard(assigned(KCT, 4, same)) == 1.0                      # :173-195
    a, b = assigned(KCT, 4, [("AAAA", 5), ("TTTC", 2)]), assigned(KCT, 4, [("AATT", 3), ("GGGG", 4)])
    assert a.jaccard(b) == 0.0 == b.jaccard(a)                                                # :198-216
    a = assigned(KCT, 4, [("AAAA", 5), ("AATT", 1), ("TTTC", 2)])
    b = assigned(KCT, 4, [("AAAA", 2), ("AATT", 1), ("GGGG", 4)])
    assert a.jaccard(b) == 2 / 4 == b.jaccard(a)                                              # :219-240
    a, e = assigned(KCT, 4, [("AAAA", 5), ("TTTC", 5)]), KCT(ksize=4)
    assert a.jaccard(e) == 0.0 == e.jaccard(a)                                                # :243-258
    assert KCT(ksize=4).jaccard(KCT(ksize=4)) == 1.0                                          # :261-271


def test_serialization_py(KCT, tmp_path, capfd):
    from oxli_amd import VERSION
    t = counted(KCT, 4, ["AAAA", "TTTT"])                                                     # :13-18
    d = json.loads(t.serialize_json())
    assert "counts" in d and d["ksize"] == 4 and d["version"] == t.version                    # :21-39
    f = str(tmp_path / "save.json")
    t.save(f)
    u = KCT.load(f)
    assert u.get("AAAA") == t.get("AAAA") and u.get("TTTT") == t.get("TTTT") and list(u) == list(t)   # :42-64
    with gzip.open(f, "wt") as fh:                                                            # :67-92
        json.dump(json.loads(t.serialize_json().replace(VERSION, "0.0.1")), fh)
    capfd.readouterr()
    KCT.load(f)
    err = capfd.readouterr().err
    assert "Version mismatch" in err and f"loaded version is 0.0.1, but current version is {VERSION}" in err
    bad = str(tmp_path / "bad.json")
    open(bad, "wt").write("hello, world")
    with pytest.raises(RuntimeError, match="Deserialization error:"):                         # :95-106
        KCT.load(bad)
    with pytest.raises(OSError, match="No such file or directory"):                           # :109-117
        t.save(str(tmp_path / "noexist" / "save.json"))


def test_dump_py(KCT, tmp_path):
    def fixture():  # :10-18
        return counted(KCT, 4, ["AAAA", "TTTT", "AATT", "GGGG", "GGGG"], store_kmers=True)
    t = fixture()
    with pytest.raises(ValueError, match="Cannot sort by both counts and keys at the same time."):     # :27-35
        t.dump(file=None, sortcounts=True, sortkeys=True)
    assert t.dump(file=None, sortcounts=False, sortkeys=False) == list(t)                     # :38-49
    by_count = [(AATT, 1), (GGGG, 2), (AAAA, 2)]
    by_key = [(GGGG, 2), (AATT, 1), (AAAA, 2)]
    assert t.dump(file=None, sortcounts=True, sortkeys=False) == by_count                     # :52-68 (ties: by hash)
    assert t.dump(file=None, sortkeys=True) == by_key                                         # :137-153
    single = counted(KCT, 4, ["AAAA"])
    assert single.dump(file=None, sortcounts=True, sortkeys=False) == [(AAAA, 1)]             # :71-87
    f = str(tmp_path / "dump.tsv")
    t.dump(file=f, sortcounts=True, sortkeys=False)
    assert open(f).readlines() == [f"{h}\t{c}\n" for h, c in by_count]                        # :90-112
    t.dump(file=f, sortkeys=True)
    assert open(f).readlines() == [f"{h}\t{c}\n" for h, c in by_key]                          # :115-134
    with pytest.raises(OSError):
        t.dump(file="", sortkeys=True)                                                        # :156-164
    e = KCT(ksize=4, store_kmers=True)
    assert e.dump(file=None, sortkeys=False) == []
    e.dump(file=f, sortkeys=False)
    assert open(f).readlines() == []                                                          # :167-187
    # ---- dump_kmers
    with pytest.raises(ValueError, match="Cannot sort by both counts and kmers at the same time."):    # :193-201
        t.dump_kmers(file=None, sortcounts=True, sortkeys=True)
    k_by_count = [("AATT", 1), ("AAAA", 2), ("CCCC", 2)]                                      # ties: by k-mer
    k_by_key = [("AAAA", 2), ("AATT", 1), ("CCCC", 2)]
    assert t.dump_kmers(file=None, sortcounts=True, sortkeys=False) == k_by_count             # :204-222
    assert counted(KCT, 4, ["AAAA"], store_kmers=True).dump_kmers(file=None, sortcounts=True, sortkeys=False) == [("AAAA", 1)]   # :225-239
    t.dump_kmers(file=f, sortcounts=True, sortkeys=False)
    assert open(f).readlines() == [f"{k}\t{c}\n" for k, c in k_by_count]                      # :242-264
    t.dump_kmers(file=f, sortkeys=True)
    assert open(f).readlines() == [f"{k}\t{c}\n" for k, c in k_by_key]                        # :267-286
    assert t.dump_kmers(file=None, sortkeys=True) == k_by_key                                 # :289-305
    with pytest.raises(OSError):
        t.dump_kmers(file="", sortkeys=True)                                                  # :308-316
    assert e.dump_kmers(file=None, sortkeys=False) == []
    e.dump_kmers(file=f, sortkeys=False)
    assert open(f).readlines() == []                                                          # :319-339
    # ---- removal seen through dump_kmers (:342-404)
    t = fixture(); t.drop("AATT")
    assert set(t.dump_kmers()) == {("AAAA", 2), ("CCCC", 2)}
    t = fixture(); t.drop_hash(GGGG)
    assert set(t.dump_kmers()) == {("AAAA", 2), ("AATT", 1)}
    t = fixture(); t.mincut(2)
    assert sorted(t.dump_kmers()) == [("AAAA", 2), ("CCCC", 2)]
    t = fixture(); t.maxcut(1)
    assert t.dump_kmers() == [("AATT", 1)]


def test_kmers_and_hashes_py(KCT, capfd):
    t = KCT(ksize=4)
    fwd = [("ATAA", 179996601836427478), ("TAAA", 15286642655859448092), ("AAAC", 9097280691811734508), ("AACC", 6779379503393060785)]
    assert t.kmers_and_hashes("ATAAACC", False) == fwd                                        # :6-17
    assert t.kmers_and_hashes("GGTTTAT", False) == fwd[::-1]                                  # :20-32
    mixed = [("ACGT", 2597925387403686983), ("AACG", 7952982457453691616), ("CAAC", 7315150081962684964)]
    for seq in ("ACGTTG", "acgttg"):                                                          # :35-64
        assert t.kmers_and_hashes(seq, False) == mixed
    assert all(t.hash_kmer(kmer) == h for kmer, h in mixed)                                   # :47-50
    capfd.readouterr()
    t.kmers_and_hashes("acxttg", False)
    assert "bad k-mer at position 1: ACXT" in capfd.readouterr().err                          # :67-77
    got = t.kmers_and_hashes("aattxttgg", False)
    assert "bad k-mer at position 2: ATTX" in capfd.readouterr().err                          # :80-90
    assert got == [("AATT", AATT), ("", 0), ("", 0), ("", 0), ("", 0), ("CCAA", 1798905482136869687)]   # :93-107
    assert t.kmers_and_hashes("aattxttgg", True) == [("AATT", AATT), ("CCAA", 1798905482136869687)]     # :110-120
    s = KCT(ksize=4, store_kmers=True)
    assert s.count("AAAA") == 1 and s.unhash(s.hash_kmer("AAAA")) == "AAAA"                   # :126-142
    s = KCT(ksize=4, store_kmers=True)
    s.count("TTTT")
    assert s.unhash(s.hash_kmer("TTTT")) == "AAAA"                                            # :145-161
    s = KCT(ksize=4, store_kmers=True)
    assert s.consume("ACGTTG") == 3 and all(s.unhash(s.hash_kmer(x)) == x for x in ("ACGT", "AACG", "CAAC"))   # :164-181
    s = KCT(ksize=4, store_kmers=True)
    assert (s.count("AAAA"), s.count("TTTT")) == (1, 2) and s.unhash(s.hash_kmer("AAAA")) == "AAAA"     # :184-204
    s = KCT(ksize=4, store_kmers=True)
    assert s.consume("AAAAACCCC") == 6 and s.get("AAAA") == 2                                 # :207-218
    s = KCT(ksize=4, store_kmers=True)
    s.count("AAAA")
    with pytest.raises(KeyError, match="Warning: Hash 1234567890 not found in table."):       # :221-233
        s.unhash(1234567890)
    n = KCT(ksize=3, store_kmers=False)
    n.count("AAA")
    with pytest.raises(ValueError, match="K-mer storage is not enabled."):                    # :236-252
        n.unhash(n.hash_kmer("AAA"))
    s = KCT(ksize=3, store_kmers=True)
    capfd.readouterr()
    assert s.consume("XAAAAAXGGGG") == 5 and len(s) == 2                                      # :255-283
    err = capfd.readouterr().err
    assert all(m in err for m in ("bad k-mer at position 1: XAA", "bad k-mer at position 5: AAX",
                                  "bad k-mer at position 6: AXG", "bad k-mer at position 7: XGG"))
    assert all(s.unhash(s.hash_kmer(x)) == x for x in ("AAA", "CCC"))
