"""The early multi-GPU route in ONE process (``-m gpu``): the sender's super-k-mer split alone (its wire format decoded on the host and
re-counted by the oracle), and the whole route as a loop-back (``kct_consume_device_routed`` with world = 1: split, run directory, K1's
RUNS instantiations behind every path of the table's policy) -- several k, paths, skewed input, repeated passes, then plain consume()
calls on the same table.  Always against the oracle (reference semantics: lib.rs:545-607 per record, lib.rs:778-837 for the union)."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import OracleTable  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    from oxli_amd import KmerCountTable
    return torch, KmerCountTable


def routed(t, dev, nbytes, consumed, max_windows=0):
    n, stats = C.c_uint64(), (C.c_uint64 * 16)()
    t._check(t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr()), nbytes, consumed, 1, 0, None, max_windows, C.byref(n), stats))
    return n.value, list(stats)


def split(t, dev, nbytes, world):
    """kct_superkmer_split_device -> per owner: list of (bases uint8 array, start-bit bool array, windows) per stream."""
    import torch
    ns = t._lib.kct_superkmer_streams(t._h)
    parts = C.c_void_p()
    off, nb, dirs = (C.c_uint64 * world)(), (C.c_uint64 * world)(), (C.c_uint64 * (world * ns))()
    t._check(t._lib.kct_superkmer_split_device(t._h, C.c_void_p(dev.data_ptr()), nbytes, world, C.byref(parts), off, nb, dirs))
    total = sum(nb)
    host = np.zeros(max(total, 1), dtype=np.uint8)
    if total:
        torch.cuda.synchronize()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), parts, C.c_size_t(total), 2) == 0   # 2 = device to host
    owners = []
    for o in range(world):
        part = host[off[o]: off[o] + nb[o]]
        d = [int(dirs[o * ns + s]) for s in range(ns)]
        units = [x >> 32 for x in d]
        nwin = [x & 0xFFFFFFFF for x in d]
        sunits = [(w + 127) // 128 for w in nwin]
        assert 16 * (sum(units) + sum(sunits)) == nb[o]
        streams, bpos, spos = [], 0, 16 * sum(units)
        for s in range(ns):
            words = part[bpos: bpos + 16 * units[s]].view("<u4")
            # 16 bases per word, first base in bits 31:30
            bases = ((words[:, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))) & 3).astype(np.uint8).reshape(-1)
            sw = part[spos: spos + 16 * sunits[s]].view("<u8")
            bits = ((sw[:, None] >> np.arange(64, dtype=np.uint64)) & 1).astype(bool).reshape(-1)[: nwin[s]]
            streams.append((bases, bits, nwin[s]))
            bpos += 16 * units[s]; spos += 16 * sunits[s]
        owners.append(streams)
    return owners


def runs_of(stream, k):
    """The runs of one decoded stream as ASCII strings."""
    bases, bits, nwin = stream
    if nwin == 0:
        return []
    assert bits[0], "a stream's first window must begin a run"
    starts = np.flatnonzero(bits)
    ends = np.append(starts[1:], nwin)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[bases]
    out = []
    for r, (a, b) in enumerate(zip(starts, ends)):
        o = int(a) + (k - 1) * r
        out.append(letters[o: o + int(b - a) + k - 1].tobytes().decode())
    return out


def host_owner(kmer, world):
    """The documented owner rule (include/kct.h, superkmer_kernels.h sk_scramble / sk_owner), restated on the host: the smallest scrambled
    canonical m-mer (m = min(8, k)) inside the k-mer, its 16-bit hash's top ten bits spread over the ranks."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    m = min(8, len(kmer))
    def scramble(x):
        x = (x * 0x9E3B) & 0xFFFF; x ^= x >> 7
        x = (x * 0x6A75) & 0xFFFF; x ^= x >> 9
        return x
    best = 1 << 20
    for i in range(len(kmer) - m + 1):
        f = r = 0
        for j in range(m):
            f = (f << 2) | code[kmer[i + j]]
            r = (r << 2) | (3 - code[kmer[i + m - 1 - j]])
        best = min(best, scramble(min(f, r)))
    return ((((best * 0x9E37) & 0xFFFF) >> 6) * world) >> 10


@pytest.mark.parametrize("k,world,R", [(21, 1, 12_000), (21, 8, 12_000), (31, 3, 12_000), (51, 8, 12_000), (64, 2, 12_000), (13, 4, 12_000), (5, 2, 12_000),
                                       (33, 64, 12_000),
                                       # k < 8: every window is its own minimiser, the owner changes from window to window -- full tiles
                                       # (50,000 reads: 30,000 window starts per workgroup) then hold more runs than the kernel's run list
                                       (7, 16, 50_000)])
def test_split_wire_format_decodes_to_the_input_k_mers(gpu, k, world, R):
    """Every good window of the input travels exactly once, to ONE owner that depends on the canonical k-mer only: the oracle's table of
    the decoded runs equals its table of the records, and the owners' key sets are disjoint."""
    torch, KCT = gpu
    G, L = 300_000, 150
    genome = oracle.synth_genome(G, 7)
    reads = oracle.synth_reads_ex(genome, 0, R, L, 3, n_ppm=3_000)
    dev = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=400_000)
    owners = split(t, dev, dev.numel(), world)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(bytes(r[:L]).decode()) for r in reads)
    rk, rc = ref.dump_arrays()
    tabs, n_dec, nruns, nbases = [], 0, 0, 0
    rng = np.random.default_rng(k * 1000 + world)
    for o, streams in enumerate(owners):
        tab = OracleTable(k)
        for st in streams:
            for run in runs_of(st, k):
                n = tab.consume(run, False)            # every window of a run is good
                assert n == len(run) - k + 1
                n_dec += n; nruns += 1; nbases += len(run)
                if rng.random() < 0.02:                # the DOCUMENTED owner rule, on a sample of windows (first, last and one inside the run)
                    for w in {0, n - 1, int(rng.integers(0, n))}:
                        assert host_owner(run[w: w + k], world) == o, (run[w: w + k], o)
        tabs.append(tab)
    assert n_dec == n_ref
    ks = [tb.dump_arrays() for tb in tabs]
    allk = np.concatenate([a for a, _ in ks]); allc = np.concatenate([c for _, c in ks])
    assert allk.size == np.unique(allk).size, "a k-mer went to two owners"
    order = np.argsort(allk, kind="stable")
    assert np.array_equal(allk[order], rk) and np.array_equal(allc[order], rc)
    if world > 1 and k >= 13:
        sizes = [a.size for a, _ in ks]
        assert min(sizes) > 0.4 * rk.size / world and max(sizes) < 2.2 * rk.size / world, sizes   # every owner holds about its share
    if k >= 21:
        assert nbases / n_dec < 1.0 + 2.6 * (k - 1) / (k - 8 + 2)    # runs are about as long as minimisers allow: the wire stays small


@pytest.mark.parametrize("k,path,cap", [(21, "auto", 1_000_000), (15, "dedupe", 6_000_000), (21, "partitioned", 1_000_000), (31, "dedupe", 6_000_000),
                                        (31, "auto", 1_000_000), (51, "auto", 1_000_000), (64, "partitioned", 1_000_000), (21, "direct", 400_000),
                                        (41, "partitioned", 6_000_000), (9, "auto", 300_000)])
def test_loopback_route_matches_the_oracle(gpu, k, path, cap):
    torch, KCT = gpu
    G, R, L = 1_500_000, 200_000, 150
    genome = oracle.synth_genome(G, 11)
    reads = oracle.synth_reads_ex(genome, 0, R, L, 3, n_ppm=2_000)     # some N: windows to skip
    tab, n_ref, _ = oracle.baseline_consume(reads, L, k, 8, native=False)
    dev = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=cap)
    t.set_path(path)
    half = (R // 2) * (L + 1)
    n1, s1 = routed(t, dev, half, (R // 2) * L, max_windows=1 << 22)        # four pipelined passes
    n2, s2 = routed(t, dev[half:], dev.numel() - half, (R - R // 2) * L)    # one pass, into a live table / shadow
    assert n1 + n2 == n_ref and s1[0] == 0 and s1[5] == -(-(half - k + 1) // (1 << 22)) and s2[5] == 1
    dk, dc = t.dump_arrays(1)
    rk, rc = tab.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert t.consumed == R * L and t.sum_counts == n_ref
    # the table goes on as any other: a plain pass doubles every count
    assert t.consume_device(dev.data_ptr(), dev.numel(), R * L) == n_ref
    dk, dc = t.dump_arrays(1)
    assert np.array_equal(dk, rk) and np.array_equal(dc, 2 * rc)


@pytest.mark.parametrize("k,path", [(21, "auto"), (31, "dedupe"), (41, "partitioned")])
def test_loopback_route_with_skewed_input(gpu, k, path):
    """Homopolymers and tandem repeats among random reads and one 2 Mbp record: long runs (cut at 1024 windows), tiles whose runs all go
    to one owner, K1's rings overflowing for the hot bins."""
    torch, KCT = gpu
    rng = random.Random(5 + k)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    recs = [rnd(150) for _ in range(60000)] + ["A" * 200000, "AC" * 100000, "ACG" * 50000, rnd(2_000_000), "ACGTN" * 40000]
    rng.shuffle(recs)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    stream = ("\n".join(recs) + "\n").encode()
    pad = (-len(stream)) % 16
    dev = torch.frombuffer(bytearray(stream + b"\n" * pad), dtype=torch.uint8).cuda()
    t = KCT(k, capacity=12_000_000)
    t.set_path(path)
    n, stats = routed(t, dev, len(stream), sum(len(r) for r in recs))
    assert n == n_ref
    dk, dc = t.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert t.consumed == ref.consumed


@pytest.mark.parametrize("k,world", [(21, 3), (51, 8)])
def test_owned_only_calls_partition_the_table(gpu, k, world):
    """world > 1 without an exchange: each call counts, of ALL the records, only what its rank owns.  The ranks' tables are disjoint and
    their union is the oracle's table."""
    torch, KCT = gpu
    G, R, L = 800_000, 100_000, 150
    reads = oracle.synth_reads(oracle.synth_genome(G, 13), 0, R, L, 9)
    tab, n_ref, _ = oracle.baseline_consume(reads, L, k, 8, native=False)
    dev = torch.from_numpy(reads.reshape(-1)).cuda()
    keys, counts, total = [], [], 0
    for rank in range(world):
        t = KCT(k, capacity=G // world + 100_000)
        n, stats = C.c_uint64(), (C.c_uint64 * 16)()
        t._check(t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr()), dev.numel(), R * L, world, rank, None, 1 << 22, C.byref(n), stats))
        total += n.value
        a, b = t.dump_arrays(1)
        keys.append(a); counts.append(b)
        assert t.sum_counts == n.value and abs(n.value - n_ref / world) < 0.25 * n_ref / world
    assert total == n_ref
    gk, gc = np.concatenate(keys), np.concatenate(counts)
    assert gk.size == np.unique(gk).size
    order = np.argsort(gk, kind="stable")
    rk, rc = tab.dump_arrays()
    assert np.array_equal(gk[order], rk) and np.array_equal(gc[order], rc)


def test_split_of_short_runs_overflowing_the_staging(gpu):
    """Windows that alternate between good and bad every few bases give far more runs per tile than random sequence: the split's LDS
    staging cannot take a whole tile (it goes out in pieces) and the regions sized for random input overflow (the split is redone with
    what the counts say)."""
    torch, KCT = gpu
    k, world = 21, 8
    rng = random.Random(3)
    # reads of exactly k + 1 bases: two windows each, every one a run of its own or two
    recs = ["".join(rng.choice("ACGT") for _ in range(k + 1)) for _ in range(150_000)]
    stream = ("\n".join(recs) + "\n").encode()
    pad = (-len(stream)) % 16
    dev = torch.frombuffer(bytearray(stream + b"\n" * pad), dtype=torch.uint8).cuda()
    t = KCT(k, capacity=1_000_000)
    owners = split(t, dev, len(stream), world)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    dec = OracleTable(k)
    n_dec = sum(dec.consume(run, False) for streams in owners for st in streams for run in runs_of(st, k))
    assert n_dec == n_ref == 2 * len(recs)
    a, b = dec.dump_arrays(), ref.dump_arrays()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    n, stats = routed(t, dev, len(stream), sum(len(r) for r in recs))
    assert n == n_ref
    dk, dc = t.dump_arrays(1)
    assert np.array_equal(dk, b[0]) and np.array_equal(dc, b[1])


def test_loopback_route_argument_checks(gpu):
    torch, KCT = gpu
    dev = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    t = KCT(31, capacity=1_000_000)
    n, stats = C.c_uint64(), (C.c_uint64 * 16)()
    call = lambda world, rank: t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr()), 1024, 0, world, rank, None, 0, C.byref(n), stats)  # noqa: E731
    assert call(1, 1) != 0 and call(0, 0) != 0 and call(65, 0) != 0
    big = KCT(71)                   # k > 64: not on this route
    assert big._lib.kct_consume_device_routed(big._h, C.c_void_p(dev.data_ptr()), 1024, 0, 1, 0, None, 0, C.byref(n), stats) != 0
    assert t._lib.kct_consume_device_routed(t._h, C.c_void_p(dev.data_ptr() + 8), 1000, 0, 1, 0, None, 0, C.byref(n), stats) != 0   # misaligned
    assert call(1, 0) == 0 and n.value == 0 and len(t) == 0      # 1024 zero bytes: nothing to count
    small = KCT(21)                 # a default table (2^16 slots) grows as it goes
    reads = oracle.synth_reads(oracle.synth_genome(50_000, 5), 0, 20_000, 150)
    d2 = torch.from_numpy(reads.reshape(-1)).cuda()
    assert routed(small, d2, d2.numel(), 20_000 * 150)[0] == 20_000 * 130
    ref, _, _ = oracle.baseline_consume(reads, 150, 21, 4, native=False)
    dk, dc = small.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
