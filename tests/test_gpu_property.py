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
# derandomize: the same examples every run (CI must not flake); explore with `--hypothesis-seed=N` by hand
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


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
