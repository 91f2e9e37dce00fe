"""Property-based parity (hypothesis): arbitrary byte strings, any k, both device paths, against the
CPU oracle.  Bit-exact: per-window hashes, n, len, sum_counts, consumed, and the error position."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import OracleTable  # noqa: E402

# mostly DNA, some lower case, a sprinkling of everything else (N, IUPAC, NUL, newline, high bytes)
base = st.sampled_from(list(b"ACGT" * 12 + b"acgt" * 2 + b"NnRYxX-*\n\r\t \x00\xff\xc3\xa9"))
seqs = st.lists(base, min_size=0, max_size=700).map(bytes)
# derandomize: the same examples every run (CI must not flake); explore with KCT_HYPOTHESIS_RANDOM=1 [--hypothesis-seed=N]
import os

COMMON = dict(deadline=None, derandomize=not os.environ.get("KCT_HYPOTHESIS_RANDOM"), suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@pytest.fixture(scope="module")
def KCT():
    from oxli_amd import KmerCountTable
    return KmerCountTable


@settings(max_examples=120, **COMMON)
@given(seq=seqs, k=st.integers(1, 70))
def test_hash_windows_property(KCT, seq, k):
    want, _ = oracle.seq_to_hashes(seq, k, force=True)
    assert np.array_equal(KCT(k).hash_windows(seq), want)


@settings(max_examples=80, **COMMON)
@given(records=st.lists(seqs, min_size=0, max_size=12), k=st.integers(1, 40), path=st.sampled_from(["direct", "partitioned"]),
       batched=st.booleans())
def test_consume_property(KCT, records, k, path, batched):
    dev, ref = KCT(k, capacity=200_000), OracleTable(k)
    dev.set_path(path)
    n_ref = sum(ref.consume(r) for r in records)
    n_dev = dev.consume_batch(records) if batched else sum(dev.consume(r) for r in records)
    assert n_dev == n_ref
    dk, dc = dev.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert (len(dev), dev.sum_counts, dev.consumed) == (len(ref), ref.sum_counts, ref.consumed)


@settings(max_examples=60, **COMMON)
@given(seq=seqs, k=st.integers(1, 33))
def test_error_mode_property(KCT, seq, k):
    dev, ref = KCT(k), OracleTable(k)
    out = []
    for t in (dev, ref):
        try:
            out.append(("ok", t.consume(seq, skip_bad_kmers=False)))
        except ValueError as e:
            out.append(("err", str(e)))
    assert out[0] == out[1]
    assert (len(dev), dev.sum_counts, dev.consumed) == (len(ref), ref.sum_counts, ref.consumed)


file_base = st.sampled_from(list(b"ACGT" * 10 + b"acgtNnRYxX-*"))
file_seqs = st.lists(file_base, min_size=0, max_size=400).map(bytes)


@settings(max_examples=40, **COMMON)
@given(records=st.lists(file_seqs, min_size=0, max_size=10), width=st.integers(1, 90), fmt=st.sampled_from(["fa", "fq", "fa.gz", "fq.gz"]),
       crlf=st.booleans(), k=st.integers(1, 40), chunk=st.sampled_from([0, 1024, 1500]))
def test_consume_file_property(KCT, tmp_path_factory, monkeypatch, records, width, fmt, crlf, k, chunk):
    import gzip
    nl = b"\r\n" if crlf else b"\n"
    out = bytearray()
    for i, s in enumerate(records):
        if fmt.startswith("fa"):
            out += b">r%d desc" % i + nl
            for j in range(0, len(s), width):
                out += s[j:j + width] + nl
        else:
            out += b"@r%d" % i + nl + s + nl + b"+" + nl + b"I" * len(s) + nl
    path = tmp_path_factory.mktemp("f") / ("x." + fmt)
    path.write_bytes(gzip.compress(bytes(out)) if fmt.endswith(".gz") else bytes(out))
    if chunk:
        monkeypatch.setenv("KCT_FILE_CHUNK", str(chunk))
    else:
        monkeypatch.delenv("KCT_FILE_CHUNK", raising=False)
    dev, ref = KCT(k), OracleTable(k)
    n_ref = sum(ref.consume(r) for r in records)
    if not records:
        return  # an empty file has no format marker; the C parser accepts it, nothing to compare
    assert dev.consume_file(str(path)) == n_ref
    assert dev.last_file_records == len(records)
    dk, dc = dev.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc) and dev.consumed == ref.consumed


@settings(max_examples=40, **COMMON)
@given(a=st.lists(seqs, max_size=6), b=st.lists(seqs, max_size=6), k=st.integers(1, 33), zero=st.booleans())
def test_add_and_dump_property(KCT, a, b, k, zero):
    da, db, ra, rb = KCT(k), KCT(k), OracleTable(k), OracleTable(k)
    for t in (da, ra):
        t.consume(b"".join(a)) if False else [t.consume(s) for s in a]
    for t in (db, rb):
        [t.consume(s) for s in b]
        if zero:
            t.count_hash(0); t.count_hash(0)     # hash 0 lives host-side in the device engine
    assert da.add(db) == ra.add(rb)
    dk, dc = da.dump_arrays(1)
    rk, rc = ra.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert (len(da), da.sum_counts, da.consumed) == (len(ra), ra.sum_counts, ra.consumed)
    k2, c2 = da.dump_arrays(2)
    order = np.lexsort((rk, rc))
    assert np.array_equal(k2, rk[order]) and np.array_equal(c2, rc[order])
    assert da.get_hash_array(rk.tolist()[:50]) == rc.tolist()[:50]
