"""The rest of the KmerCountTable surface (SURVEY.md 8f "next" rows: dump / save / load formats,
store_kmers, table analytics, set operations, similarity), re-expressing what the reference's tests
pin (src/python/tests/test_dump.py, test_histo.py, test_remove.py, test_serialization.py,
test_setops.py, test_metrics.py, test_kmers_and_hashes.py:126-283, test_canonicalization.py,
test_dunders.py, test_attr.py).  Hashes and counts always come from the device."""
import gzip
import json
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def KCT():
    from oxli_amd import KmerCountTable
    return KmerCountTable


AAAA, AATT, GGGG = 17832910516274425539, 382727017318141683, 73459868045630124  # test_dump.py:52-54


@pytest.fixture
def small(KCT):
    t = KCT(ksize=4)
    t.count("AAAA"); t.count("TTTT"); t.count("AATT"); t.count("GGGG"); t.count("GGGG")
    return t


# ---- dump (test_dump.py) ----------------------------------------------------------------------------------
def test_dump_file_tsv(small, tmp_path):
    f = tmp_path / "d.tsv"
    assert small.dump(file=str(f), sortkeys=True) == []
    assert f.read_text() == f"{GGGG}\t2\n{AATT}\t1\n{AAAA}\t2\n"       # "{hash}\t{count}\n" (test_dump.py:111-115)
    assert small.dump(file=str(f), sortcounts=True) == []
    assert f.read_text() == f"{AATT}\t1\n{GGGG}\t2\n{AAAA}\t2\n"
    with pytest.raises(ValueError):
        small.dump(file=str(f), sortcounts=True, sortkeys=True)
    with pytest.raises(OSError):
        small.dump(file=str(tmp_path / "no" / "such.tsv"))


def test_dump_large_sorted_on_device(KCT):
    rng = random.Random(1)
    seq = "".join(rng.choice("ACGT") for _ in range(200000))
    t = KCT(15)
    t.consume(seq); t.consume(seq[:50000])
    keys, counts = t.dump_arrays(1)
    assert np.all(keys[:-1] < keys[1:])
    k2, c2 = t.dump_arrays(2)
    assert np.all((c2[:-1] < c2[1:]) | ((c2[:-1] == c2[1:]) & (k2[:-1] < k2[1:])))
    order = np.lexsort((keys, counts))
    assert np.array_equal(k2, keys[order]) and np.array_equal(c2, counts[order])
    assert sorted(t.dump()) == list(zip(keys.tolist(), counts.tolist()))
    assert list(t) == list(zip(keys.tolist(), counts.tolist()))
    assert sorted(t.hashes) == keys.tolist()


# ---- histo / min / max (test_histo.py) -----------------------------------------------------------------------
def test_histo_min_max(KCT):
    t = KCT(4)
    assert t.min == 0 and t.max == 0 and t.histo() == [(0, 0)] and t.histo(zero=False) == []
    t.count("AAAA"); t.count("TTTT"); t.consume("CCCCCC")
    assert t.min == 2 and t.max == 3
    assert t.histo(zero=False) == [(2, 1), (3, 1)]
    assert t.histo(zero=True) == [(0, 0), (1, 0), (2, 1), (3, 1)]


# ---- removal (test_remove.py) ----------------------------------------------------------------------------------
def test_drop_mincut_maxcut(KCT):
    t = KCT(4)
    t.count("AAAA"); t.count("TTTT"); t.count("AATT"); t.count("GGGG"); t.count("GGGG"); t.count("GGGG")
    consumed = t.consumed
    t.drop("AAAA")
    assert t.get("AAAA") == 0 and len(t) == 2
    t.drop("ACGT")                      # absent: no error
    t.drop_hash(AATT)
    assert len(t) == 1 and t.get("GGGG") == 3 and t.consumed == consumed
    t.count("AAAA"); t.count("AATT"); t.count("AATT")
    assert t.mincut(2) == 1 and sorted(c for _, c in t) == [2, 3]      # removes count < 2
    assert t.maxcut(2) == 1 and [c for _, c in t] == [2]               # removes count > 2
    assert t.mincut(0) == 0 and t.maxcut(100) == 0


# ---- set operations (test_setops.py) ---------------------------------------------------------------------------
def test_set_operations(KCT):
    a, b = KCT(4), KCT(4)
    for k in ("AAAA", "AATT", "GGGG"):
        a.count(k)
    for k in ("AATT", "GGGG", "ATAT"):
        b.count(k)
    sa, sb = set(a.hashes), set(b.hashes)
    assert a.union(b) == sa | sb == (a | b)
    assert a.intersection(b) == sa & sb == (a & b)
    assert a.difference(b) == sa - sb == (a - b)
    assert a.symmetric_difference(b) == sa ^ sb == (a ^ b)


# ---- similarity (test_metrics.py) ---------------------------------------------------------------------------------
def test_jaccard_and_cosine(KCT):
    a, b, e1, e2 = KCT(4), KCT(4), KCT(4), KCT(4)
    assert e1.jaccard(e2) == 1.0 and e1.cosine(e2) == 0.0          # empty-table conventions
    for k, n in (("AAAA", 3), ("AATT", 1), ("GGGG", 2)):
        for _ in range(n):
            a.count(k)
    for k, n in (("AATT", 4), ("GGGG", 1), ("ATAT", 5)):
        for _ in range(n):
            b.count(k)
    assert a.jaccard(b) == 2 / 4 and a.jaccard(a) == 1.0 and a.jaccard(e1) == 0.0
    dot = 1 * 4 + 2 * 1
    want = dot / (math.sqrt(9 + 1 + 4) * math.sqrt(16 + 1 + 25))
    assert math.isclose(a.cosine(b), want, rel_tol=1e-12)          # the reference compares with rel_tol 1e-5
    assert math.isclose(a.cosine(a), 1.0, rel_tol=1e-12) and a.cosine(e1) == 0.0


# ---- save / load (test_serialization.py): wire format = serde_json of the struct, gzip ------------------------
def test_serialize_save_load(KCT, tmp_path, capfd):
    from oxli_amd import VERSION
    t = KCT(4)
    t.count("AAAA"); t.count("TTTT"); t.consume("GGGGG")
    d = json.loads(t.serialize_json())
    assert list(d) == ["counts", "ksize", "version", "consumed", "store_kmers", "hash_to_kmer"]   # struct order (lib.rs:32-39)
    assert d["ksize"] == 4 and d["version"] == VERSION == t.version and d["consumed"] == 13
    assert d["counts"] == {str(AAAA): 2, str(GGGG): 2} and d["store_kmers"] is False and d["hash_to_kmer"] is None
    f = str(tmp_path / "save.json")
    t.save(f)
    assert open(f, "rb").read(2) == b"\x1f\x8b"                     # gzip (niffler Format::Gzip)
    u = KCT.load(f)
    assert list(u) == list(t) and u.get("TTTT") == 2 and u.consumed == 13 and u.ksize == 4 and len(u) == 2
    # a file as the reference writes it: compact serde_json, u64 keys as strings, gzip
    ref_json = '{"counts":{"17832910516274425539":5,"73459868045630124":1},"ksize":4,"version":"0.3.0","consumed":24,"store_kmers":false,"hash_to_kmer":null}'
    g = str(tmp_path / "ref.json.gz")
    with gzip.open(g, "wt") as fh:
        fh.write(ref_json)
    r = KCT.load(g)
    assert r.get("AAAA") == 5 and r.get("CCCC") == 1 and r.consumed == 24 and r.sum_counts == 6
    p = str(tmp_path / "plain.json")                                 # niffler also accepts uncompressed input
    open(p, "w").write(ref_json.replace("0.3.0", "0.0.1"))
    capfd.readouterr()
    r2 = KCT.load(p)
    assert "Version mismatch: loaded version is 0.0.1, but current version is " + VERSION in capfd.readouterr().err
    assert r2.version == "0.0.1" and r2.get("AAAA") == 5
    open(p, "w").write("hello, world")
    with pytest.raises(RuntimeError, match="Deserialization error:"):
        KCT.load(p)
    with pytest.raises(OSError, match="No such file or directory"):
        t.save(str(tmp_path / "noexist" / "save.json"))


def test_load_of_a_hand_assembled_reference_format_file(KCT, capfd):
    """tests/golden/reference_format_save.json.gz: a file in the REFERENCE's save() format (lib.rs:274-292: serde_json of the struct at
    lib.rs:31-39, gzip as niffler / flate2 write it at Level::One), assembled by tests/golden/make_reference_save_fixture.py from the
    format's description -- unsorted HashMap order, a u64::MAX count, store_kmers with its hash -> k-mer map -- not by this repository's
    writer.  load() must take it; the hashes in it are the reference's own known answers (reference_kats.json)."""
    import os
    from oxli_amd import VERSION
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = open(os.path.join(here, "reference_format_save.json.gz"), "rb").read()
    assert raw[:10] == bytes.fromhex("1f8b0800" "00000000" "04ff")       # no name, no time stamp, XFL "fastest", OS unknown
    t = KCT.load(os.path.join(here, "reference_format_save.json.gz"))
    want = {"AACC": 3, "AAAA": 1, "ACGT": 2 ** 64 - 1, "ATAA": 2, "CCCC": 7, "AACG": 40, "AAAC": 5}
    assert t.ksize == 4 and len(t) == 7 and t.consumed == 123456789012 and t.version == VERSION and t.store_kmers is True
    for kmer, c in want.items():
        assert t.get(kmer) == c
        assert t.unhash(t.hash_kmer(kmer)) == kmer
    assert t.get("TTAT") == 2 and t.get("GGGG") == 7                      # reverse complements of ATAA / CCCC
    assert t.count("AACC") == 4 and t.count("GGTT") == 5                  # a loaded table goes on counting (GGTT = revcomp of AACC)
    assert sorted(t.hashes) == sorted(t.hash_kmer(k_) for k_ in want)
    capfd.readouterr()
    u = KCT.load(os.path.join(here, "reference_format_save_old_version.json"))   # uncompressed, older version, hash_to_kmer null
    assert "Version mismatch: loaded version is 0.2.9, but current version is " + VERSION in capfd.readouterr().err
    assert u.store_kmers is False and len(u) == 7 and u.get("ACGT") == 2 ** 64 - 1 and u.sum_counts == (sum(want.values())) % 2 ** 64


# ---- dunders / attributes (test_dunders.py, test_attr.py) ------------------------------------------------------
def test_consume_into_a_loaded_table(KCT, tmp_path):
    """load() then consume(): a loaded table takes every consume route (the per-record call glue included) and counts on
    top of what the file held -- against the oracle counting both inputs."""
    from oracle import OracleTable
    rng = random.Random(77)
    a, b = "".join(rng.choice("ACGT") for _ in range(30000)), "".join(rng.choice("ACGTN") for _ in range(30000))
    t = KCT(21)
    t.consume(a)
    f = str(tmp_path / "t.json.gz")
    t.save(f)
    u = KCT.load(f)
    ref = OracleTable(21)
    ref.consume(a)
    recs = [b[i:i + 150] for i in range(0, len(b), 150)]
    assert [u.consume(r) for r in recs] == [ref.consume(r) for r in recs]       # per-record (deferred + call glue)
    assert u.consume_batch(recs) == sum(ref.consume(r) for r in recs)           # batch
    with pytest.raises(ValueError):
        u.consume("ACGTN" * 10, skip_bad_kmers=False)
    with pytest.raises(ValueError):
        ref.consume("ACGTN" * 10, skip_bad_kmers=False)
    dk, dc = u.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert u.consumed == ref.consumed and u.sum_counts == ref.sum_counts


def test_dunders_and_attrs(KCT):
    from oxli_amd import VERSION
    t = KCT(ksize=16)
    assert len(t) == 0 and t.version == VERSION == "0.3.0" and t.consumed == 0 and t.sum_counts == 0 and t.hashes == []
    t["ACGTACGTACGTACGT"] = 5
    assert t["ACGTACGTACGTACGT"] == 5 and len(t) == 1 and t.consumed == 0
    t.consume("GCTAGCTAGCTA")                                       # shorter than k: nothing counted, consumed grows
    assert len(t) == 1 and t.consumed == 12
    with pytest.raises(ValueError):
        t["ACGT"]
    with pytest.raises(OverflowError):
        KCT(256)


# ---- canonical form (test_canonicalization.py) and store_kmers (test_kmers_and_hashes.py:126-283) ----------------
def test_canon(KCT):
    t = KCT(4)
    assert t.canon("TTTT") == "AAAA" and t.canon("acgt") == "ACGT" and t.canon("GGTA") == "GGTA" and t.canon("TACC") == "GGTA"
    with pytest.raises(ValueError, match="kmer size does not match count table ksize"):
        t.canon("ACG")
    with pytest.raises(ValueError, match="kmer contains invalid characters"):
        t.canon("ACGN")


def test_kmers_and_hashes_lists(KCT, kats, capfd):
    cg = KCT(ksize=4)
    assert cg.kmers_and_hashes("ATAAACC", False) == [("ATAA", 179996601836427478), ("TAAA", 15286642655859448092),
                                                     ("AAAC", 9097280691811734508), ("AACC", 6779379503393060785)]
    assert cg.kmers_and_hashes("GGTTTAT", False) == [("AACC", 6779379503393060785), ("AAAC", 9097280691811734508),
                                                     ("TAAA", 15286642655859448092), ("ATAA", 179996601836427478)]
    assert cg.kmers_and_hashes("acgttg", False) == [("ACGT", 2597925387403686983), ("AACG", 7952982457453691616),
                                                    ("CAAC", 7315150081962684964)]
    capfd.readouterr()
    assert cg.kmers_and_hashes("aattxttgg", False) == [("AATT", 382727017318141683), ("", 0), ("", 0), ("", 0), ("", 0),
                                                       ("CCAA", 1798905482136869687)]
    assert "bad k-mer at position 2: ATTX" in capfd.readouterr().err
    assert cg.kmers_and_hashes("aattxttgg", True) == [("AATT", 382727017318141683), ("CCAA", 1798905482136869687)]
    cg.kmers_and_hashes("acxttg", False)
    assert "bad k-mer at position 1: ACXT" in capfd.readouterr().err


def test_store_kmers(KCT, capfd):
    cg = KCT(ksize=4, store_kmers=True)
    assert cg.count("AAAA") == 1 and cg.count("TTTT") == 2
    assert cg.unhash(cg.hash_kmer("TTTT")) == "AAAA"
    with pytest.raises(KeyError, match="Warning: Hash 1234567890 not found in table."):
        cg.unhash(1234567890)
    with pytest.raises(ValueError, match="K-mer storage is not enabled."):
        KCT(3).unhash(1)
    cg = KCT(ksize=4, store_kmers=True)
    assert cg.consume("ACGTTG") == 3
    for kmer in ["ACGT", "AACG", "CAAC"]:
        assert cg.unhash(cg.hash_kmer(kmer)) == kmer
    cg = KCT(ksize=4, store_kmers=True)
    assert cg.consume("AAAAACCCC") == 6 and cg.get("AAAA") == 2
    # bad k-mers never raise on this branch, whatever the flag says (lib.rs:552-573)
    cg = KCT(ksize=3, store_kmers=True)
    capfd.readouterr()
    assert cg.consume("XAAAAAXGGGG", skip_bad_kmers=False) == 5 and len(cg) == 2
    err = capfd.readouterr().err
    for msg in ("bad k-mer at position 1: XAA", "bad k-mer at position 5: AAX", "bad k-mer at position 6: AXG", "bad k-mer at position 7: XGG"):
        assert msg in err
    assert cg.unhash(cg.hash_kmer("AAA")) == "AAA" and cg.unhash(cg.hash_kmer("GGG")) == "CCC"
    # dump_kmers and add() with string maps (test_add.py:87-109, test_dump.py)
    a, b = KCT(5, store_kmers=True), KCT(5, store_kmers=True)
    a.consume("ATGCA"); b.consume("GGCAT")
    assert a.add(b) == (1, 1)
    assert a.dump_kmers(sortkeys=True) == [("ATGCA", 1), ("ATGCC", 1)]
    assert a.dump_kmers(sortcounts=True) == [("ATGCA", 1), ("ATGCC", 1)]
    c = KCT(5, store_kmers=False)
    c.consume("GGCAT")
    capfd.readouterr()
    a.add(c)
    assert "Warning: Incoming table does not store k-mers" in capfd.readouterr().err
    with pytest.raises(ValueError, match="K-mer storage is disabled"):
        c.dump_kmers()
    # the string map survives save / load
    import os
    import tempfile
    f = os.path.join(tempfile.mkdtemp(), "t.json")
    a.save(f)
    r = KCT.load(f)
    assert r.store_kmers and r.unhash(r.hash_kmer("ATGCA")) == "ATGCA" and r.dump_kmers(sortkeys=True) == a.dump_kmers(sortkeys=True)


# ---- the boundary from plain C (what a foreign-language binding sees) ---------------------------------------
def test_c_program_through_the_abi(tmp_path):
    import os
    import subprocess

    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_example"
    libdir = os.path.join(root, "oxli_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi_example.c"),
                    "-L", libdir, "-lkct_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    rng = random.Random(99)
    seq = "".join(rng.choice("ACGTACGTN") for _ in range(5000))
    k = 4
    out = subprocess.run([str(exe), str(k), seq], check=True, capture_output=True, text=True).stdout.splitlines()
    ref = oracle.OracleTable(k)
    n = ref.consume(seq)
    ref2 = oracle.OracleTable(k)
    ref2.consume(seq)
    added, fresh = ref.add(ref2)
    keys, counts = ref.dump_arrays()
    assert out[0] == f"n {n} {n}" and out[1] == f"added {added} new {fresh}"
    assert out[2] == f"len {len(ref)}" and out[3] == f"sum {ref.sum_counts}" and out[4] == f"consumed {ref.consumed}"
    assert out[5:5 + keys.size] == [f"{h} {c}" for h, c in zip(keys.tolist(), counts.tolist())]
    assert out[-1] == "error_mode status 3 position 1"      # KCT_ERR_BAD_KMER after one good 4-mer (ACGT | CGTN is bad)


@pytest.mark.parametrize("k,R,L,G,passes", [(21, 200_000, 150, 2_000_000, 3), (51, 3_000, 10_000, 3_000_000, 4)])
def test_c_program_routes_through_rccl_with_a_world_of_one(tmp_path, k, R, L, G, passes):
    """tests/c_rccl_example.c: a plain-C process (no torch, no Python in it) drives kct_consume_device_routed with the RCCL
    exchange of libkct_rccl.so -- communicator, size all-to-all, asynchronous ncclSend / ncclRecv payload, pipelined passes -- at
    world = 1.  Its routed table, its directly counted table and the CPU oracle's must agree."""
    import os
    import subprocess

    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_rccl_example"
    libdir = os.path.join(root, "oxli_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                    os.path.join(root, "tests", "c_rccl_example.c"), "-L", libdir, "-lkct_hip", "-lkct_rccl", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    run = subprocess.run([str(exe), str(k), str(R), str(L), str(G), str(passes)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    out = [ln for ln in run.stdout.splitlines() if ln.startswith(("plain ", "routed ", "passes ", "merge ", "merged "))]   # (RCCL prints a banner of its own)
    plain, routed, tail, merge, merged = (ln.split() for ln in out)
    assert plain[0] == "plain" and routed[0] == "routed" and plain[1:] == routed[1:]
    # the late route's collective (kct_rccl_merge_across_ranks) through the same communicator: every pair sent (to itself), key 0's count
    # delivered to its owner, consumed kept, the refilled table's digests unchanged
    assert merged[0] == "merged" and merged[1:] == plain[1:]
    assert merge[2] == merge[4] == plain[4] and merge[6] == "1" and merge[8] == "1"
    ss = oracle.ShardSet(k, L, genome=oracle.synth_genome(G, 42), nreads=R, seed_r=1337, threads=8)
    d = ss.digest()
    assert [int(v) for v in routed[2::2][:3]] == [d["n"], d["len"], d["sum_counts"]]
    assert [int(v) for v in routed[8:11]] == [d["sum_hc"], d["xor_hc"], d["sum_sq"]]
    assert int(tail[1]) == passes and int(tail[3]) > 0 and int(tail[7]) == 0


# ---- analytics on the device (kct_analytics.hip) against plain dict arithmetic --------------------------
def _random_table(KCT, rng, n, kspace, with_zero, k=21):
    t, d = KCT(k), {}
    hashes = rng.integers(1, kspace, size=n, dtype=np.uint64)
    for h in hashes.tolist():
        d[h] = d.get(h, 0) + 1
    keys = np.fromiter(d.keys(), dtype=np.uint64)
    vals = np.fromiter(d.values(), dtype=np.uint64) * rng.integers(1, 50, size=keys.size, dtype=np.uint64)
    d = dict(zip(keys.tolist(), vals.tolist()))
    t._check(t._lib.kct_merge_host(t._h, keys.ctypes.data, vals.ctypes.data, keys.size, None, None))
    if with_zero:
        for _ in range(3):
            t.count_hash(0)              # key 0 lives beside the device table
        d[0] = 3
    return t, d


def test_device_analytics_match_dict_arithmetic(KCT):
    rng = np.random.default_rng(11)
    for with_zero_a, with_zero_b in ((False, False), (True, False), (True, True)):
        a, da = _random_table(KCT, rng, 30000, 40000, with_zero_a)
        b, db = _random_table(KCT, rng, 20000, 40000, with_zero_b)
        assert len(a) == len(da) and len(b) == len(db)
        assert (a.min, a.max) == (min(da.values()), max(da.values()))
        hist = {}
        for c in da.values():
            hist[c] = hist.get(c, 0) + 1
        assert a.histo(zero=False) == sorted(hist.items())
        assert a.histo(zero=True) == [(f, hist.get(f, 0)) for f in range(max(hist) + 1)]
        sa, sb = set(da), set(db)
        assert a.union(b) == sa | sb and a.intersection(b) == sa & sb
        assert a.difference(b) == sa - sb and b.difference(a) == sb - sa and a.symmetric_difference(b) == sa ^ sb
        assert a.jaccard(b) == len(sa & sb) / len(sa | sb)
        dot = sum(da[h] * db[h] for h in sa & sb)
        want = dot / (math.sqrt(sum(v * v for v in da.values())) * math.sqrt(sum(v * v for v in db.values())))
        assert math.isclose(a.cosine(b), want, rel_tol=1e-12)
        assert math.isclose(a.cosine(a), 1.0, rel_tol=1e-12)
        # cuts and drops rebuild the table on the device
        cut = 20
        assert a.mincut(cut) == sum(1 for v in da.values() if v < cut)
        da = {h: v for h, v in da.items() if v >= cut}
        assert a.maxcut(60) == sum(1 for v in da.values() if v > 60)
        da = {h: v for h, v in da.items() if v <= 60}
        victim = next(iter(da))
        a.drop_hash(victim); da.pop(victim)
        a.drop_hash(12345678901234567)   # absent: nothing happens
        if 0 in da:
            a.drop_hash(0); da.pop(0)
        keys, counts = a.dump_arrays(1)
        assert dict(zip(keys.tolist(), counts.tolist())) == da
        assert len(a) == len(da) and a.sum_counts == sum(da.values())
        assert a.get_hash(victim) == 0
        a.count_hash(victim)             # the rebuilt table keeps working
        assert a.get_hash(victim) == 1


def test_device_analytics_edge_cases(KCT):
    e1, e2 = KCT(5), KCT(5)
    assert (e1.min, e1.max, e1.histo(), e1.histo(zero=False)) == (0, 0, [(0, 0)], [])
    assert e1.union(e2) == set() and e1.jaccard(e2) == 1.0 and e1.cosine(e2) == 0.0
    assert e1.mincut(5) == 0 and e1.maxcut(0) == 0
    e1.drop_hash(7)
    t = KCT(4)
    t.consume("AAAATTTTGGGG")
    assert t.jaccard(e1) == 0.0 and t.intersection(e1) == set() and t.union(e1) == set(t.hashes)
    assert t.difference(e1) == set(t.hashes) and e1.difference(t) == set()
    t["AAAA"] = 0                        # a key whose count is 0 is still a key (lib.rs:675-681)
    assert t.min == 0 and (0, 1) in t.histo(zero=False)
    assert t.mincut(1) == 1 and t.get("AAAA") == 0 and t.hash_kmer("AAAA") not in set(t.hashes)
    only_zero = KCT(4)
    only_zero.count_hash(0); only_zero.count_hash(0)
    assert (only_zero.min, only_zero.max, only_zero.histo(zero=False)) == (2, 2, [(2, 1)])
    assert only_zero.union(t) == {0} | set(t.hashes) and only_zero.maxcut(1) == 1 and len(only_zero) == 0
    cleared = KCT(4)
    cleared.consume("ACGTACGTAC"); cleared.clear()   # lazily cleared table: stale slots must not be scanned
    assert cleared.max == 0 and cleared.histo(zero=False) == [] and t.intersection(cleared) == set() and cleared.union(t) == set(t.hashes)
    assert t.jaccard(cleared) == 0.0 and cleared.mincut(100) == 0


def test_native_save_load_many_pieces(KCT, tmp_path):
    """kct_save writes the counts as many independently deflated pieces inside ONE gzip member: any gzip reader
    (here Python's) must see exactly serialize_json(), and kct_load must bring every pair back."""
    rng = np.random.default_rng(3)
    t, d = _random_table(KCT, rng, 300000, 1 << 62, True)     # ~300k pairs: five 65536-pair pieces; key 0 included
    t.consume("ACGTACGTACGTACGTACGTACGTACGT")                  # consumed > 0
    f = str(tmp_path / "big.json.gz")
    t.save(f)
    text = gzip.open(f, "rb").read().decode()
    assert text == t.serialize_json()
    parsed = json.loads(text)
    assert len(parsed["counts"]) == len(t) and parsed["counts"]["0"] == 3
    u = KCT.load(f)
    ku, cu = u.dump_arrays(1)
    kt, ct = t.dump_arrays(1)
    assert np.array_equal(ku, kt) and np.array_equal(cu, ct) and u.consumed == t.consumed and u.ksize == t.ksize
    # members in another order, whitespace, an unknown member: a JSON reader accepts them
    p = str(tmp_path / "odd.json")
    open(p, "w").write('{ "ksize": 5, "extra": {"a": [1, "}"]}, "counts": { "7": 2 , "9":1 }, "consumed": 4, '
                       '"version": "0.3.0", "store_kmers": true, "hash_to_kmer": {"7": "AAAAC"} }')
    o = KCT.load(p)
    assert (o.ksize, len(o), o.get_hash(7), o.get_hash(9), o.consumed, o.store_kmers) == (5, 2, 2, 1, 4, True)
    assert o.unhash(7) == "AAAAC"
    e = KCT(9)
    g = str(tmp_path / "empty.json.gz")
    e.save(g)
    assert json.loads(gzip.open(g, "rb").read())["counts"] == {} and len(KCT.load(g)) == 0
    for bad in ('{"counts":{"x":1},"ksize":4}', '{"counts":{"1":-1},"ksize":4}', '{"counts":{},"ksize":400}', '{"counts":{}}', '{"counts":{"1":1}'):
        open(p, "w").write(bad)
        with pytest.raises(RuntimeError, match="Deserialization error"):
            KCT.load(p)


def test_large_pair_merges_take_the_partitioned_route_and_tally_like_add(KCT, monkeypatch):
    """A merge of >= 2^18 pairs into a table of up to 1024 blocks is radix-partitioned and merged per block in LDS.
    Zero counts, key 0 and a live table: contents and the (total added, new keys) tallies must equal
    plain dict arithmetic -- 'new' meaning the key's count was 0 before (lib.rs:801-803) -- on both routes."""
    import ctypes as C
    rng = np.random.default_rng(21)
    base_keys = rng.integers(1, 1 << 63, size=600_000, dtype=np.uint64)
    for route in ("partitioned", "atomic"):
        t = KCT(21, capacity=3_000_000)
        if route == "atomic":
            t.set_path("direct")  # atomic inserts only: merges too
        d = {}
        t.profile(True)
        for rnd in range(3):
            idx = rng.choice(base_keys.size, size=400_000, replace=False)
            keys = base_keys[idx].copy()                  # 400k distinct keys per call (add() merges a map: no repeats within a
            keys[0] = 0                                   # call), many of them seen in earlier rounds; key 0 lives beside the table
            counts = rng.integers(0, 5, size=keys.size, dtype=np.uint64)   # zero counts too
            a, b = C.c_uint64(), C.c_uint64()
            t._check(t._lib.kct_merge_host(t._h, keys.ctypes.data, counts.ctypes.data, keys.size, C.byref(a), C.byref(b)))
            tot = new = 0
            for h, c in zip(keys.tolist(), counts.tolist()):
                if d.get(h, 0) == 0:
                    new += 1
                d[h] = d.get(h, 0) + c
                tot += c
            assert (a.value, b.value) == (tot, new), (route, rnd)
        prof = t.profile_read()
        assert ("aggregate_pairs_kernel" in prof) == (route == "partitioned"), prof
        k, c = t.dump_arrays(1)
        assert dict(zip(k.tolist(), c.tolist())) == d
