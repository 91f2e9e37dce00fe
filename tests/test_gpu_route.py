"""The early route's machinery in ONE process (``kct_consume_device_routed`` with world = 1 and null callbacks: a loop-back
exchange): K1 with owner-grouped bins, region packing, K1b reading packed regions through its offset table, K2 -- every mode,
several k, skewed input, repeated passes, then plain consume() calls on the same table.  Always against the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import OracleTable  # noqa: E402

MODES = {"hash": 0, "dedupe64": 1, "compact": 2}


@pytest.fixture(scope="module")
def gpu():
    import torch
    from oxli_amd import KmerCountTable
    return torch, KmerCountTable


def routed(t, dev, nbytes, consumed, mode):
    n, stats = C.c_uint64(), (C.c_uint64 * 8)()
    t._check(t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr()), nbytes, consumed, 1, 0, MODES[mode], None, None, None,
                                              C.byref(n), stats))
    return n.value, list(stats)


@pytest.mark.parametrize("k,mode,cap", [(21, "compact", 1_000_000), (15, "compact", 6_000_000), (21, "dedupe64", 1_000_000), (31, "dedupe64", 6_000_000),
                                        (21, "hash", 1_000_000), (31, "hash", 6_000_000), (51, "hash", 1_000_000), (64, "hash", 1_000_000)])
def test_loopback_route_matches_the_oracle(gpu, k, mode, cap):
    torch, KCT = gpu
    G, R, L = 1_500_000, 200_000, 150
    genome = oracle.synth_genome(G, 11)
    reads = oracle.synth_reads_ex(genome, 0, R, L, 3, n_ppm=2_000)     # some N: windows to skip
    ref = OracleTable(k)
    tab, n_ref, _ = oracle.baseline_consume(reads, L, k, 8, native=False)
    dev = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=cap)
    half = (R // 2) * (L + 1)
    n1, s1 = routed(t, dev, half, (R // 2) * L, mode)
    n2, _ = routed(t, dev[half:], dev.numel() - half, (R - R // 2) * L, mode)     # second pass: live table / shadow
    assert n1 + n2 == n_ref and s1[0] == 0 and s1[2] == (4 if mode == "compact" else 8)
    dk, dc = t.dump_arrays(1)
    rk, rc = tab.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert t.consumed == R * L and t.sum_counts == n_ref
    # the table goes on as any other: a plain pass doubles every count
    assert t.consume_device(dev.data_ptr(), dev.numel(), R * L) == n_ref
    dk, dc = t.dump_arrays(1)
    assert np.array_equal(dk, rk) and np.array_equal(dc, 2 * rc)
    del ref


@pytest.mark.parametrize("k,mode", [(21, "compact"), (31, "dedupe64"), (41, "hash")])
def test_loopback_route_with_skewed_input(gpu, k, mode):
    """Homopolymers and tandem repeats among random reads: K1's and K1b's rings overflow for the hot bins, the entries take the
    overflow lists (bucketed by owner, exchanged, merged with the direct insert)."""
    torch, KCT = gpu
    rng = random.Random(5 + k)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    recs = [rnd(150) for _ in range(60000)] + ["A" * 200000, "AC" * 100000, "ACG" * 50000, rnd(2_000_000)]
    rng.shuffle(recs)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    stream = ("\n".join(recs) + "\n").encode()
    pad = (-len(stream)) % 16
    dev = torch.frombuffer(bytearray(stream + b"\n" * pad), dtype=torch.uint8).cuda()
    t = KCT(k, capacity=12_000_000)
    n, stats = routed(t, dev, len(stream), sum(len(r) for r in recs), mode)
    assert n == n_ref and stats[3] > 0          # overflow entries existed and were counted
    dk, dc = t.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert t.consumed == ref.consumed


def test_loopback_route_argument_checks(gpu):
    torch, KCT = gpu
    dev = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    t = KCT(31, capacity=1_000_000)
    n, stats = C.c_uint64(), (C.c_uint64 * 8)()
    call = lambda world, rank, mode: t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr()), 1024, 0, world, rank, mode, None, None, None, C.byref(n), stats)  # noqa: E731
    assert call(1, 0, 2) != 0          # compact entries need k <= 21
    assert call(2, 0, 0) != 0          # more than one rank needs callbacks
    assert call(1, 1, 0) != 0 and call(1, 0, 3) != 0
    small = KCT(21)                    # 2^16 slots: too small for the route
    assert small._lib.kct_consume_device_routed(small._h, C.c_void_p(dev.data_ptr()), 1024, 0, 1, 0, 0, None, None, None, C.byref(n), stats) != 0
    assert call(1, 0, 0) == 0 and n.value == 0 and len(t) == 0      # 1024 zero bytes: nothing to count
