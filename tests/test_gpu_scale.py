"""BASELINE.json's configurations at FULL size on one MI355X (``-m gpu``), and the device paths that only large tables
take (two partition levels, table-sized shadows, the two-level pair flush, the dedupe probe).

EXACT checks against the CPU oracle at full size (``oracle.ShardSet``: the key space sharded over the host's cores, reads
generated on the fly -- ~5x10^8 k-mers/s on the GPU box's host): C2 and the C4 shard are compared pair by pair (every
(hash, count) of the device's dump looked up in the oracle's tables, ``len`` equal: the maps are equal); the 10^10-k-mer
configurations by ``len``, ``sum_counts``, min / max count and the order-free digests sum(hash x count), xor(hash x count),
sum(count^2) mod 2^64 (``kct_digest``), plus every key of an oracle slice and as many absent keys, counts EQUAL.

Cheaper invariants beside them:

* ``n == reads x (L - k + 1) == sum_counts``, ``consumed == reads x L``, ``len`` inside the bounds the genome gives;
* consuming the same stream again doubles ``sum_counts`` and ``max``, quadruples the sum of squared counts, adds no key;
* the automatic path and the DIRECT path (one HBM atomic per k-mer -- the simplest kernel, oracle-checked on every
  small case in test_gpu_parity.py) agree on ``len``, ``sum_counts``, ``min``, ``max``, the sum of squared counts, and
  on the count of every key of a sample of 10^5-10^6 keys;

Smaller inputs into tables of more than 1024 blocks are compared with the oracle's table bit for bit.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
from oracle import OracleTable  # noqa: E402

SEED_G, SEED_R = 42, 1337

# BASELINE.json configs[2..4] on ONE GPU (8-GPU configs: one rank's shard of the reads), plus the north-star sentence's
# own workload: 100 M x 150 bp at k=21 on one MI355X.
FULL = {
    "C3": dict(reads=100_000_000, L=150, k=31, genome=500_000_000),
    "C4-shard": dict(reads=12_500_000, L=150, k=21, genome=500_000_000),
    "C5-shard": dict(reads=1_250_000, L=10_000, k=51, genome=387_500_000),
    "NS-k21": dict(reads=100_000_000, L=150, k=21, genome=500_000_000),
}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from oxli_amd import KmerCountTable, _lib
    return torch, KmerCountTable, _lib.load()


def synth(gpu, G, R, L, first=0, seed_g=SEED_G, seed_r=SEED_R):
    torch, _, lib = gpu
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.kct_synth_genome_device(g.data_ptr(), G, seed_g, stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, first, R, L, seed_r, stream) == 0
    torch.cuda.synchronize()
    return g, r


def stats(t):
    lo, hi, sq = t._count_stats()
    return len(t), t.sum_counts, lo, hi, sq


def host_mem_gib():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) / (1 << 20)
    return 0.0


def oracle_shardset(gpu, g, r, R, L, k, **model):
    """The oracle's table of reads [0, R) of the stream over the device-made genome ``g`` -- generated again on the host,
    after checking that the host generator reproduces the device's bytes on the first and the last 2000 reads."""
    genome = g.cpu().numpy()
    ns = min(2000, R)
    for first in (0, R - ns):
        dev = r[first * (L + 1): (first + ns) * (L + 1)].cpu().numpy().reshape(ns, L + 1)
        assert np.array_equal(dev, oracle.synth_reads_ex(genome, first, ns, L, SEED_R, **model)), "host and device generators differ"
    return oracle.ShardSet(k, L, genome=genome, nreads=R, seed_r=SEED_R, expect_keys=min(len(genome), R * (L - k + 1)), **model)


GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")))
DIGEST_KEYS = ("sum_hc", "xor_hc", "sum_sq", "len", "sum_counts", "min", "max", "n", "consumed")
# test name -> entry of tests/golden/config_digests.json (the CPU oracle's digests of the same input, made once on the GPU box's host
# by tests/golden/make_config_digests.py).  KCT_FULL_ORACLE=1 rebuilds the oracle's table on the spot instead (minutes per config).
GOLDEN_OF = {"C3": "C3", "C4-shard": "C4_shard", "C5-shard": "C5_shard", "NS-k21": "north_star_k21"}


def golden_digest(name, **expect):
    d = GOLDEN[name]
    for key, v in expect.items():   # the entry really is this input
        assert d[key] == v, (name, key, d[key], v)
    assert (d["seed_g"], d["seed_r"]) == (SEED_G, SEED_R)
    return d


def table_digest(t, n):
    lo, hi, _ = t._count_stats()
    return dict(zip(("sum_hc", "xor_hc", "sum_sq"), t.digest()), len=len(t), sum_counts=t.sum_counts, min=lo, max=hi, n=n, consumed=t.consumed)


def assert_digest_equal(t, ss, n):
    d = ss if isinstance(ss, dict) else ss.digest()
    lo, hi, _ = t._count_stats()
    got = dict(zip(("sum_hc", "xor_hc", "sum_sq"), t.digest()), len=len(t), sum_counts=t.sum_counts, min=lo, max=hi, n=n, consumed=t.consumed)
    assert got == {k_: d[k_] for k_ in got}, (got, d)


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_config(gpu, name):
    torch, KCT, _ = gpu
    c = FULL[name]
    R, L, k, G = c["reads"], c["L"], c["k"], c["genome"]
    free, _total = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip("needs a whole MI355X (250 GiB of free HBM)")
    if host_mem_gib() < 40:
        pytest.fail("the full-size oracle table needs ~20 GiB of host memory")
    g, r = synth(gpu, G, R, L)
    n_expect = R * (L - k + 1)
    t = KCT(k, capacity=G)
    t.profile(True)
    n = t.consume_device(r.data_ptr(), r.numel(), R * L)
    prof = t.profile_read()
    assert n == n_expect
    assert t.consumed == R * L
    distinct, total, lo, hi, sq = stats(t)
    assert total == n_expect and lo >= 1
    # every genome position is the start of one canonical k-mer; a position is missed with probability e^-(windows per position)
    assert 0.9 * G * (1.0 - np.exp(-n_expect / G)) <= distinct <= G - k + 1
    assert "count_windows_kernel" not in prof, prof            # the partitioned paths carried it
    assert "repartition_kernel" in prof or "repartition_kernel<compact>" in prof, prof   # ... through two levels
    # an oracle slice: the first 3000 reads (the first 40 long ones)
    ns = 3000 if L <= 1000 else 40
    sub = r[: ns * (L + 1)].cpu().numpy().reshape(ns, L + 1)
    ref = OracleTable(k)
    for i in range(ns):
        ref.consume(sub[i, :L])
    rk, rc = ref.dump_arrays()
    # a sample of keys: the slice's keys plus keys that are absent
    sample = np.concatenate([rk, rk ^ np.uint64(0x5555555555555555)])
    sample_counts = np.array(t.get_hash_array(sample), dtype=np.uint64)
    # THE ORACLE'S TABLE OF THE WHOLE INPUT: its committed digests (tests/golden/config_digests.json) must equal the device table's;
    # every key of the slice is there with at least the slice's count.  C4's shard -- and every config under KCT_FULL_ORACLE=1 --
    # rebuilds the oracle's table on this host: digests, every sampled key's count, and (C4's shard, 4.8x10^8 pairs) the whole
    # dump pair by pair.
    gold = golden_digest(GOLDEN_OF[name], reads=R, read_len=L, k=k, genome=G)
    assert_digest_equal(t, gold, n)
    assert np.all(sample_counts[: rk.size] >= rc) and not sample_counts[rk.size:].any()
    if name == "C4-shard" or os.environ.get("KCT_FULL_ORACLE") == "1":
        ss = oracle_shardset(gpu, g, r, R, L, k)
        assert ss.digest() == {k_: gold[k_] for k_ in ss.digest()}, "the committed digest is not the oracle's"
        assert_digest_equal(t, ss, n)
        assert sample_counts.tolist() == [ss.get_hash(int(h)) for h in sample.tolist()]
        if name == "C4-shard":
            dk, dc = t.dump_arrays(0)
            assert dk.size == ss.digest()["len"] and ss.mismatches(dk, dc) == 0
            del dk, dc
        del ss
    del g
    # the same stream again: every count doubles, no key is new
    assert t.consume_device(r.data_ptr(), r.numel(), R * L) == n_expect
    d2, total2, lo2, hi2, sq2 = stats(t)
    assert (d2, total2, lo2, hi2) == (distinct, 2 * n_expect, 2 * lo, 2 * hi)
    assert sq2 == pytest.approx(4.0 * sq, rel=1e-12)
    assert np.array_equal(np.array(t.get_hash_array(sample), dtype=np.uint64), 2 * sample_counts)
    t.release_scratch()
    del t
    torch.cuda.empty_cache()
    # the direct path on the same stream
    d = KCT(k, capacity=G)
    d.set_path("direct")
    assert d.consume_device(r.data_ptr(), r.numel(), R * L) == n_expect
    dd, dtotal, dlo, dhi, dsq = stats(d)
    assert (dd, dtotal, dlo, dhi) == (distinct, n_expect, lo, hi)
    assert dsq == pytest.approx(sq, rel=1e-12)
    assert np.array_equal(np.array(d.get_hash_array(sample), dtype=np.uint64), sample_counts)


def test_whole_C5_on_one_gpu_equals_the_oracle_digest(gpu):
    """BASELINE.json configs[4] at its OWN size on one MI355X: 10 M x 10 kbp reads of a 3.1 Gbp genome, k = 51 -- 9.95x10^10 k-mers,
    3.1x10^9 distinct, into ONE 2^33-slot table (128 GiB).  The 100 GB of reads do not sit beside that table and its scratch, so they
    are generated on the device in eight pieces (the multi-GPU job's eight rank shards, in rank order) and fed to the same table call
    after call, as a file reader would.  The table must equal the CPU oracle's table of the whole input: its committed digests
    (tests/golden/config_digests.json "C5", 13 host-minutes on 64 threads; lib.rs:545-607 per record, 778-837 for the union)."""
    torch, KCT, lib = gpu
    gold = golden_digest("C5")
    R, L, k, G = gold["reads"], gold["read_len"], gold["k"], gold["genome"]
    assert (R, L, k, G) == (10_000_000, 10_000, 51, 3_100_000_000)
    free, _total = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip("needs a whole MI355X (250 GiB of free HBM)")
    pieces = 8
    per = R // pieces
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(g.data_ptr(), G, SEED_G, stream) == 0
    r = torch.empty(per * (L + 1), dtype=torch.uint8, device="cuda")
    genome_host = g.cpu().numpy()
    t = KCT(k, capacity=G)
    assert t.capacity == 1 << 33
    t.profile(True)
    n = 0
    for p in range(pieces):
        assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, p * per, per, L, SEED_R, stream) == 0
        torch.cuda.synchronize()
        if p in (0, pieces - 1):   # the host generator makes the same bytes (what the oracle's digest was made from)
            dev = r[: 3 * (L + 1)].cpu().numpy().reshape(3, L + 1)
            assert np.array_equal(dev, oracle.synth_reads_ex(genome_host, p * per, 3, L, SEED_R)), "host and device generators differ"
        n += t.consume_device(r.data_ptr(), r.numel(), per * L)
    assert n == R * (L - k + 1)
    prof = t.profile_read()
    assert "count_windows_kernel" not in prof and "repartition_kernel" in prof, prof   # two partition levels carried it, no per-k-mer atomics
    assert_digest_equal(t, gold, n)
    # an oracle slice: the first 40 reads' keys are there with at least the slice's counts
    sub = oracle.synth_reads_ex(genome_host, 0, 40, L, SEED_R)
    ref = OracleTable(k)
    for i in range(40):
        ref.consume(sub[i, :L])
    rk, rc = ref.dump_arrays()
    got = np.array(t.get_hash_array(rk), dtype=np.uint64)
    assert np.all(got >= rc)


@pytest.mark.parametrize("route_k", [21])
def test_C4_through_the_early_route_eight_owners_in_turn(gpu, route_k):
    """BASELINE.json configs[3] (100 M x 150 bp, k = 21, 8 GPUs) through the EARLY multi-GPU route at full size, the one GPU playing the
    eight owners in turn: ``kct_consume_device_routed(world=8, rank=r, ops=NULL)`` cuts the WHOLE input into super-k-mers and counts
    what rank r owns into r's own table (kct_route.hip's owned-only mode: the sender's split kernel and the owner's K1-over-runs at
    full size, no wire).  The owners' tables are a partition of the key space: their union must equal the oracle's table of the whole
    input -- sums (xor) over owners of the digests == tests/golden/config_digests.json "north_star_k21"."""
    import ctypes as C
    torch, KCT, lib = gpu
    gold = golden_digest("north_star_k21", k=route_k)
    R, L, k, G = gold["reads"], gold["read_len"], gold["k"], gold["genome"]
    free, _total = torch.cuda.mem_get_info()
    if free < 100 * (1 << 30):
        pytest.skip("needs 100 GiB of free HBM")
    g, r = synth(gpu, G, R, L)
    del g
    world = 8
    M = (1 << 64) - 1
    tot = dict(sum_hc=0, xor_hc=0, sum_sq=0, len=0, sum_counts=0, n=0)
    lo_all, hi_all, shares = M, 0, []
    for rank in range(world):
        t = KCT(k, capacity=G // world)
        nn, stats = C.c_uint64(), (C.c_uint64 * 16)()
        t._check(lib.kct_consume_device_routed(t._h, C.c_void_p(r.data_ptr()), r.numel(), R * L if rank == 0 else 0, world, rank, None, 0,
                                               C.byref(nn), stats))
        s_hc, x_hc, s_sq = t.digest()
        lo, hi, _ = t._count_stats()
        tot["sum_hc"] = (tot["sum_hc"] + s_hc) & M
        tot["xor_hc"] ^= x_hc
        tot["sum_sq"] = (tot["sum_sq"] + s_sq) & M
        tot["len"] += len(t)
        tot["sum_counts"] += t.sum_counts
        tot["n"] += nn.value
        tot["consumed"] = tot.get("consumed", 0) + t.consumed
        lo_all, hi_all = min(lo_all, lo), max(hi_all, hi)
        shares.append(nn.value)
        del t
    tot.update(min=lo_all, max=hi_all)
    assert tot == {k_: gold[k_] for k_ in tot}, (tot, gold)
    assert max(shares) < 1.1 * sum(shares) / world and min(shares) > 0.9 * sum(shares) / world, shares   # owners' shares are even


def test_full_C2_table_equals_the_oracle_table(gpu):
    """BASELINE.json configs[1] at full size (1 M x 150 bp, k=21, genome 5 Mbp): the device's dump against the oracle's table
    of the same 1.3x10^8 k-mers, pair by pair, on the automatic path (cold: probe, compact dedupe-first, conversion), again
    after a second pass into the live table (steady state), and on the hashing path."""
    torch, KCT, _ = gpu
    G, R, L, k = 5_000_000, 1_000_000, 150, 21
    g, r = synth(gpu, G, R, L)
    ss = oracle_shardset(gpu, g, r, R, L, k)
    d = ss.digest()
    assert d["n"] == R * (L - k + 1) == d["sum_counts"]
    for path in ("auto", "partitioned", "direct"):
        t = KCT(k, capacity=G)
        t.set_path(path)
        n = t.consume_device(r.data_ptr(), r.numel(), R * L)
        dk, dc = t.dump_arrays(1)
        assert dk.size == d["len"] and ss.mismatches(dk, dc) == 0, path
        assert_digest_equal(t, ss, n)
        if path == "auto":   # the same reads again into the live table: every count doubles (the steady state's second step)
            assert t.consume_device(r.data_ptr(), r.numel(), R * L) == n
            dk2, dc2 = t.dump_arrays(1)
            assert np.array_equal(dk2, dk) and np.array_equal(dc2, 2 * dc)


@pytest.mark.parametrize("model", [dict(sub_ppm=10_000), dict(n_ppm=10_000), dict(sub_ppm=5_000, n_ppm=2_000), dict(sorted_total=1_000_000)],
                         ids=["1pct_substitutions", "1pct_N", "mixed_errors", "position_sorted"])
def test_C2_with_sequencing_errors_equals_the_oracle_table(gpu, model):
    """SURVEY 8d's secondary inputs at C2 size, from the device generator's error model (include/kct_synth.h): 1 % substitution
    errors (a quarter of the k-mers become singletons: the dedupe probe's estimate is off), 1 % N (windows skipped),
    both, and position-sorted reads.  Whatever path the table chooses: the oracle's table, pair by pair, and n."""
    torch, KCT, lib = gpu
    G, R, L, k = 5_000_000, 1_000_000, 150, 21
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    m = dict(sub_ppm=0, n_ppm=0, sorted_total=0)
    m.update(model)
    assert lib.kct_synth_genome_device(g.data_ptr(), G, SEED_G, stream) == 0
    assert lib.kct_synth_reads_device_ex(r.data_ptr(), g.data_ptr(), G, 0, R, L, SEED_R, m["sub_ppm"], m["n_ppm"], m["sorted_total"],
                                         oracle.SEED_E, stream) == 0
    torch.cuda.synchronize()
    ss = oracle_shardset(gpu, g, r, R, L, k, **m)
    d = ss.digest()
    if m["n_ppm"]:
        assert d["n"] < R * (L - k + 1)
    for path in ("auto", "dedupe", "partitioned"):
        t = KCT(k, capacity=int(d["len"]))
        t.set_path(path)
        n = t.consume_device(r.data_ptr(), r.numel(), R * L)
        dk, dc = t.dump_arrays(1)
        assert dk.size == d["len"] and ss.mismatches(dk, dc) == 0, path
        assert_digest_equal(t, ss, n)


@pytest.mark.parametrize("k", [33, 51, 64])
def test_128_bit_dedupe_first_at_C2_size_equals_the_oracle_table(gpu, k):
    """33 <= k <= 64, deep coverage of a 2 Mbp genome (1 M x 150 bp): K1 partitions mix128 pairs of the two packed words (16-byte
    entries), aggregate_blocks128_kernel counts them into the 1024 x 4096-slot shadow (two-word keys claimed by CAS + flag),
    the conversion hashes each distinct k-mer once.  Forced (set_path("dedupe")) against the oracle's table pair by pair; the automatic
    choice hashes every window at these k and must give the same table."""
    torch, KCT, _ = gpu
    G, R, L = 2_000_000, 1_000_000, 150
    g, r = synth(gpu, G, R, L, seed_g=7, seed_r=8)
    genome = g.cpu().numpy()
    ss = oracle.ShardSet(k, L, genome=genome, nreads=R, seed_r=8, expect_keys=G)
    d = ss.digest()
    assert d["n"] == R * (L - k + 1)
    t = KCT(k, capacity=G)
    t.set_path("dedupe")
    t.profile(True)
    n = t.consume_device(r.data_ptr(), r.numel(), R * L)
    prof = t.profile_read()
    assert "aggregate_blocks128_kernel" in prof and "partition_windows_kernel<raw128>" in prof and "partition_windows_kernel" not in prof, prof
    dk, dc = t.dump_arrays(1)
    assert "shadow128_flush_kernel" in t.profile_read()
    assert dk.size == d["len"] and ss.mismatches(dk, dc) == 0
    assert_digest_equal(t, ss, n)
    a = KCT(k, capacity=G)                     # automatic: both passes hash every window
    assert a.consume_device(r.data_ptr(), r.numel(), R * L) == n
    a.profile(True)
    assert a.consume_device(r.data_ptr(), r.numel(), R * L) == n
    assert "aggregate_blocks128_kernel" not in a.profile_read(), a.profile_read()   # (1.0x over hashing on its showcase: never chosen by itself since round 4)
    ak, ac = a.dump_arrays(1)
    assert np.array_equal(ak, dk) and np.array_equal(ac, 2 * dc)


def _oracle_table(reads, L, k):
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    tab, kmers, _ = oracle.baseline_consume(reads, L, k, threads, native=False)
    return tab, kmers


@pytest.mark.parametrize("k,path,G,R", [(21, "dedupe", 6_000_000, 300_000), (31, "dedupe", 6_000_000, 300_000),
                                        (21, "auto", 3_000_000, 600_000), (31, "auto", 3_000_000, 600_000),
                                        (21, "auto", 6_000_000, 600_000),
                                        (17, "partitioned", 6_000_000, 300_000), (51, "auto", 6_000_000, 300_000)])
def test_tables_of_more_than_1024_blocks_match_the_oracle(gpu, k, path, G, R):
    """2^24 slots = 2048 blocks: two partition levels.  k=21 takes the compact two-level dedupe-first path (a 4 GiB
    compact shadow, repartition_kernel<compact>), k=31 the 64-bit one.  With "auto" and k <= 32 the 8x10^7-window call
    is large enough for the dedupe PROBE (2^23 windows) to decide: ~26 windows per distinct k-mer with the 3 Mbp genome
    (dedupe-first goes on), ~13 with the 6 Mbp one (the rest of the call hashes every window).  Reading the table
    converts the pending k-mers through the two-level pair flush.  The oracle counts the same reads on the host."""
    torch, KCT, _ = gpu
    L = 150
    genome = oracle.synth_genome(G, 5)
    reads = oracle.synth_reads(genome, 0, R, L, 9)
    ref, n_ref = _oracle_table(reads, L, k)
    dev_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    t = KCT(k, capacity=6_000_000)
    assert t.capacity == 1 << 24
    t.set_path(path)
    t.profile(True)
    assert t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), R * L) == n_ref == R * (L - k + 1)
    prof = t.profile_read()
    # "auto": the probe's verdict -- dedupe-first goes on (3 Mbp genome), or the rest of the call hashes every window.  At k = 21
    # the 9x10^7-window call cannot pay for the 4 GiB compact shadow of a two-level table (0.15 windows per shadow byte): the
    # 64-bit variant, whose shadow is table-sized (256 MiB), takes over.
    dedupe_first = path == "dedupe" or (path == "auto" and k <= 32 and G == 3_000_000)
    compact = dedupe_first and k == 21 and path == "dedupe"
    if compact:
        assert "repartition_kernel<compact>" in prof and "aggregate_blocks32_kernel" in prof, prof
    elif dedupe_first:
        assert "repartition_kernel" in prof and "aggregate_blocks_kernel<shadow>" in prof, prof
    else:
        assert "repartition_kernel" in prof and "aggregate_blocks_kernel" in prof, prof
    if path == "auto" and k <= 32:
        assert ("partition_windows_kernel" in prof) == (not dedupe_first), prof
    dk, dc = t.dump_arrays(1)
    if dedupe_first:
        assert "repartition_kernel<pairs>" in t.profile_read(), t.profile_read()   # the two-level pair flush ran
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert len(t) == len(ref) and t.sum_counts == n_ref
    # a second pass into the live table and shadow, then a third after a read
    assert t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), R * L) == n_ref
    assert t.get_hash(int(rk[1234])) == 2 * int(rc[1234])
    assert t.consume_device(dev_reads.data_ptr(), dev_reads.numel() // 2, R * L // 2) == n_ref // 2
    dk, dc = t.dump_arrays(1)
    half, _ = _oracle_table(reads[: R // 2], L, k)
    hk, hc = half.dump_arrays()
    want = dict(zip(rk.tolist(), (2 * rc).tolist()))
    for h, c in zip(hk.tolist(), hc.tolist()):
        want[h] += c
    assert np.array_equal(dk, rk) and dc.tolist() == [want[h] for h in rk.tolist()]


def test_heavy_hitters_and_bad_bases_at_scale(gpu):
    """SURVEY 8d's robustness inputs at scale.  (1) `"ATGC" x 5x10^7` as one 200 Mbp record: a handful of k-mers, every lane
    of every wave holds the same one -- the partition kernels' rings overflow, the pass is abandoned and the direct kernel's
    wave combining takes over.  (2) the C2 stream with 1 % of its bases turned into N: n < reads x (L - k + 1), every window
    over an N is skipped.  The automatic path must give the direct path's table bit for bit; an oracle slice checks (2)."""
    torch, KCT, _ = gpu
    k = 21
    # (1)
    unit = torch.tensor(list(b"ATGC"), dtype=torch.uint8, device="cuda")
    rec = unit.repeat(50_000_000)
    stream = torch.cat([rec, torch.tensor([10], dtype=torch.uint8, device="cuda")])
    tables = {}
    for path in ("auto", "direct"):
        t = KCT(k, capacity=5_000_000)
        t.set_path(path)
        assert t.consume_device(stream.data_ptr(), stream.numel(), rec.numel()) == rec.numel() - k + 1
        tables[path] = t.dump_arrays(1)
        assert int(tables[path][1].sum()) == rec.numel() - k + 1
    assert np.array_equal(tables["auto"][0], tables["direct"][0]) and np.array_equal(tables["auto"][1], tables["direct"][1])
    small = OracleTable(k)
    small.consume("ATGC" * 1000)
    assert np.array_equal(small.dump_arrays()[0], tables["auto"][0])     # the same few k-mers
    del stream, rec
    # (2)
    G, R, L = 5_000_000, 1_000_000, 150
    g, r = synth(gpu, G, R, L)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    hit = (torch.rand(r.numel(), device="cuda", generator=gen) < 0.01) & (r != 10)
    r[hit] = ord("N")
    res = {}
    for path in ("auto", "direct"):
        t = KCT(k, capacity=G)
        t.set_path(path)
        res[path] = (t.consume_device(r.data_ptr(), r.numel(), R * L),) + t.dump_arrays(1)
    assert res["auto"][0] == res["direct"][0] < R * (L - k + 1)
    assert np.array_equal(res["auto"][1], res["direct"][1]) and np.array_equal(res["auto"][2], res["direct"][2])
    ns = 20_000
    sub = r[: ns * (L + 1)].cpu().numpy().reshape(ns, L + 1)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(sub[i, :L]) for i in range(ns))
    t = KCT(k, capacity=G)
    assert t.consume_device(r.data_ptr(), ns * (L + 1), ns * L) == n_ref
    dk, dc = t.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)


@pytest.mark.parametrize("k,path", [(21, "dedupe"), (31, "dedupe"), (31, "partitioned"), (51, "partitioned")])
def test_skewed_input_through_two_partition_levels(gpu, k, path):
    """Homopolymers, tandem repeats and a repeated unit among random sequence, into a table of 2048 blocks: both partition
    levels' rings overflow for the hot bins, the entries take the overflow regions and the direct insert.  Bit for bit
    against the oracle."""
    import random
    torch, KCT, _ = gpu
    rng = random.Random(31 + k)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    unit = rnd(1000)
    recs = [rnd(150) for _ in range(40000)] + ["A" * 300000, "T" * 50000, "AC" * 150000, "ACG" * 100000, unit * 200, rnd(3_000_000)]
    rng.shuffle(recs)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    t = KCT(k, capacity=6_000_000)
    assert t.capacity == 1 << 24
    t.set_path(path)
    t.profile(True)
    assert t.consume_batch(recs) == n_ref
    prof = t.profile_read()
    assert any(name.startswith("repartition_kernel") for name in prof), prof
    dk, dc = t.dump_arrays(1)
    rk, rc = ref.dump_arrays()
    assert np.array_equal(dk, rk) and np.array_equal(dc, rc)
    assert t.consumed == ref.consumed
