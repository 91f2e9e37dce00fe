"""Worker of tests/test_gpu_distributed.py: one of several ranks that SHARE one GPU (gloo between them), each with its
own device table.  Every rank counts its shard of one read stream, merge_across_ranks() makes the tables the owner
partitions of the global table, and rank 0 checks their union against the oracle's count of the whole stream.

    python -m torch.distributed.run --nproc-per-node W tests/dist_gpu_worker.py <k> <reads per rank> <genome> [route] [read length]

route: "late" (default: private tables, then merge_across_ranks) or "early:auto" / "early:dedupe" / "early:partitioned"
(consume_device_early: super-k-mers travel to the rank that owns them and are counted there by the named path of the table's policy;
two calls, the first cut into several pipelined passes, so that later passes meet live tables and shadows).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV, NATIVE = "cpu", None   # where the scalar all-reduces live; libkct_rccl.so's communicator when the run asks for it


def main():
    k, per_rank, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    route = sys.argv[4] if len(sys.argv) > 4 else "late"
    L = int(sys.argv[5]) if len(sys.argv) > 5 else 150
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # KCT_DIST_BACKEND=nccl (a box with one GPU PER RANK: tests/test_gpu_distributed.py::test_two_gpus_over_rccl): RCCL between the ranks;
    # with KCT_DIST_NATIVE=1 the exchanges are libkct_rccl.so's own (NativeRccl) instead of torch.distributed's
    backend = os.environ.get("KCT_DIST_BACKEND", "gloo")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    if route == "rccl-alone":
        rccl_alone(k, per_rank, G, L)
        return
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    global DEV, NATIVE
    DEV = "cuda" if backend == "nccl" else "cpu"
    if backend == "nccl" and os.environ.get("KCT_DIST_NATIVE") == "1":
        from oxli_amd.distributed import NativeRccl
        NATIVE = NativeRccl()
    import oracle
    from oxli_amd import KmerCountTable
    from oxli_amd.distributed import global_scalar_sum, merge_across_ranks, owner_of

    genome = oracle.synth_genome(G, 42)
    reads = oracle.synth_reads(genome, rank * per_rank, per_rank, L, 1337)
    dev_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    if route.startswith("fault"):
        faults(k, per_rank, G, L, rank, world, dev_reads)
        route = "early:auto"     # ... and then a clean job through the same processes and process group
    if route.startswith("early"):
        early(route.split(":")[1], k, per_rank, G, L, rank, world, genome, reads, dev_reads)
        return
    if route == "late:store_kmers":
        late_store_kmers(k, per_rank, L, rank, world, reads)
        return
    t = KmerCountTable(k, capacity=G)
    n = t.consume_device(dev_reads.data_ptr(), dev_reads.numel(), per_rank * L)
    assert n == per_rank * (L - k + 1)
    if rank == 1:
        for _ in range(3):
            t.count_hash(0)            # key 0 lives beside the device table; its owner is rank 0
    cap_private = t.capacity
    recv = merge_across_ranks(t, native=NATIVE)
    assert recv > 0
    keys, counts = t.dump_arrays(1)
    nz = keys != 0
    assert np.all(owner_of(torch.from_numpy(keys[nz].view(np.int64).copy()), world).numpy() == rank)   # only my slice of hash space
    assert t.capacity <= cap_private   # an owner's table is sized for its slice
    total = global_scalar_sum(t.sum_counts, DEV)
    assert total == world * per_rank * (L - k + 1) + 3, total
    assert global_scalar_sum(t.consumed, DEV) == world * per_rank * L
    parts = [None] * world
    dist.all_gather_object(parts, (keys, counts))
    if rank == 0:
        ref = oracle.OracleTable(k)
        allreads = oracle.synth_reads(genome, 0, world * per_rank, L, 1337)
        tab, _, _ = oracle.baseline_consume(allreads, L, k, min(8, len(os.sched_getaffinity(0))), native=False)
        for _ in range(3):
            tab.count_hash(0)
        rk, rc = tab.dump_arrays()
        gk = np.concatenate([p[0] for p in parts])
        gc = np.concatenate([p[1] for p in parts])
        order = np.argsort(gk, kind="stable")
        assert gk.size == np.unique(gk).size, "owner partitions overlap"
        assert np.array_equal(gk[order], rk) and np.array_equal(gc[order], rc), "union of the owner tables differs from the oracle"
        del ref
        print(f"DIST_GPU_OK world={world} distinct={rk.size}")
    dist.barrier()
    dist.destroy_process_group()


def rccl_alone(k, per_rank, G, L):
    """ONE rank with the nccl back end (= RCCL; a one-GPU box cannot hold more): the device-tensor, asynchronous branches of the
    exchanges -- the size all-to-all on device int64, the payload all-to-all on device bytes with the library counting on another
    stream meanwhile, exchange_pairs' uneven all-to-all -- run through a real communicator, everything sent to itself."""
    import oracle
    from oxli_amd import KmerCountTable
    from oxli_amd.distributed import consume_device_early, exchange_pairs, exchange_route

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    genome = oracle.synth_genome(G, 42)
    reads = oracle.synth_reads(genome, 0, per_rank, L, 1337)
    dev_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    tab, _, _ = oracle.baseline_consume(reads, L, k, min(8, len(os.sched_getaffinity(0))), native=False)
    rk, rc = tab.dump_arrays()
    t = KmerCountTable(k, capacity=G)
    n, st = consume_device_early(t, dev_reads.data_ptr(), dev_reads.numel(), per_rank * L, max_windows=1 << 22, exchange_when_alone=True)
    assert n == per_rank * (L - k + 1) and st["passes"] > 1 and st["runs"] > 0 and st["windows_sent"] == 0, st   # (sent = to OTHER ranks)
    keys, counts = t.dump_arrays(1)
    assert np.array_equal(keys, rk) and np.array_equal(counts, rc)
    # the late route's exchange: pairs on the device through all_to_all_single with uneven splits
    assert exchange_route() == "all_to_all"
    pairs = torch.stack([torch.from_numpy(keys.view(np.int64).copy()), torch.from_numpy(counts.view(np.int64).copy())], dim=1).cuda()
    got, zero = exchange_pairs(pairs, torch.tensor([pairs.shape[0]], dtype=torch.int64), zero_count=7)
    torch.cuda.synchronize()
    assert zero == 7 and got.device.type == "cuda" and torch.equal(got, pairs)
    # libkct_rccl.so IN THIS PROCESS, beside PyTorch (it binds to PyTorch's own librccl.so.1: one copy of RCCL): its own communicator
    # bootstrapped over the torch group, the early route through its kct_exchange_ops and the late route's kct_rccl_merge_across_ranks
    from oxli_amd.distributed import NativeRccl, merge_across_ranks
    rccl_maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
    nat = NativeRccl()
    assert len({os.path.realpath(p_) for p_ in rccl_maps + [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]}) == 1, "two copies of RCCL in one process"
    t2 = KmerCountTable(k, capacity=G)
    n2, st2 = consume_device_early(t2, dev_reads.data_ptr(), dev_reads.numel(), per_rank * L, max_windows=1 << 22, exchange_when_alone=True, native=nat)
    assert n2 == n and st2["passes"] == st["passes"] and st2["runs"] == st["runs"], (st, st2)
    k2, c2 = t2.dump_arrays(1)
    assert np.array_equal(k2, rk) and np.array_equal(c2, rc)
    t2.count_hash(0); t2.count_hash(0)            # key 0 lives beside the device table and must survive the merge
    nat.merge_when_alone(True)
    cap_before, consumed_before = t2.capacity, t2.consumed
    assert merge_across_ranks(t2, native=nat) == rk.size
    assert t2.get_hash(0) == 2 and t2.consumed == consumed_before and t2.capacity <= cap_before
    t2.drop_hash(0)
    k3, c3 = t2.dump_arrays(1)
    assert np.array_equal(k3, rk) and np.array_equal(c3, rc)
    nat.close()
    print(f"DIST_GPU_OK world=1 distinct={rk.size} route=rccl-alone passes={st['passes']} native=ok")
    dist.destroy_process_group()


def late_store_kmers(k, per_rank, L, rank, world, reads):
    """store_kmers tables through the late route: the hash -> k-mer map follows the keys to their owners (lib.rs:810-828)."""
    import oracle
    from oxli_amd import KmerCountTable
    from oxli_amd.distributed import global_scalar_sum, merge_across_ranks, owner_of

    t = KmerCountTable(k, store_kmers=True)
    for r in reads:
        t.consume(bytes(r[:L]).decode())
    merge_across_ranks(t)
    keys, counts = t.dump_arrays(1)
    assert np.all(owner_of(torch.from_numpy(keys.view(np.int64).copy()), world).numpy() == rank)
    pairs = t.dump_kmers(sortkeys=True)                      # [(k-mer, count)]: every owned key has its k-mer, with the global count
    assert len(pairs) == keys.size and set(t._hash_to_kmer) == set(keys.tolist())
    for kmer, c in pairs[:: max(1, len(pairs) // 200)]:
        assert t.hash_kmer(kmer) in t._hash_to_kmer and t.get(kmer) == c and t.unhash(t.hash_kmer(kmer)) == kmer
    parts = [None] * world
    dist.all_gather_object(parts, pairs)
    if rank == 0:
        ref = oracle.OracleTable(k)
        allreads = oracle.synth_reads(oracle.synth_genome(int(sys.argv[3]), 42), 0, world * per_rank, L, 1337)
        for r in allreads:
            ref.consume(bytes(r[:L]).decode())
        rk, rc = ref.dump_arrays()
        got = {}
        for p in parts:
            for kmer, c in p:
                h = ref.hash_kmer(kmer)
                assert h not in got
                got[h] = c
        assert got == dict(zip(rk.tolist(), rc.tolist()))
        print(f"DIST_GPU_OK world={world} distinct={rk.size} route=late:store_kmers")
    assert global_scalar_sum(t.sum_counts, DEV) == world * per_rank * (L - k + 1)
    dist.barrier()
    dist.destroy_process_group()


def faults(k, per_rank, G, L, rank, world, dev_reads):
    """include/kct.h's promise for the early route -- "a failure on any rank ends the call on EVERY rank with an error" -- under injected
    faults (kct_debug_inject_fault): one rank fails at one point of the protocol (the HBM query before the first collective, the split of
    pass 1, a slab allocation, the start / the wait of a payload, the owner-side count); every rank must come back from THAT call with
    an error within 30 s, none left inside a collective, the exchange's slabs released.  The next scenario (and the clean job that
    follows, checked against the oracle) runs through the same process group: a rank that had been left behind would hang it."""
    import time

    from oxli_amd import KmerCountTable, _lib
    from oxli_amd.distributed import consume_device_early

    lib = _lib.load()
    MEMINFO, SPLIT, ALLOC, START, WAIT, COUNT = 1, 2, 3, 4, 5, 6
    scenarios = [(SPLIT, 1, 2 % world), (MEMINFO, 0, 1 % world), (ALLOC, 0, 0), (START, 1, 3 % world), (WAIT, 2, 1 % world), (COUNT, 0, 2 % world), (START, 0, 0)]
    for point, pas, bad in scenarios:
        t = KmerCountTable(k, capacity=max(G // world, 400_000))
        if rank == bad:
            assert lib.kct_debug_inject_fault(t._h, point, pas) == 0
        dist.barrier()
        t0 = time.perf_counter()
        failed, text = False, ""
        try:
            consume_device_early(t, dev_reads.data_ptr(), dev_reads.numel(), per_rank * L, max_windows=1 << 22)
        except (RuntimeError, MemoryError) as e:
            failed, text = True, str(e)
        dt = time.perf_counter() - t0
        verdicts = [None] * world
        dist.all_gather_object(verdicts, (failed, round(dt, 2), text[:160]))
        assert all(v[0] for v in verdicts), f"fault {point} at pass {pas} on rank {bad}: not every rank returned an error: {verdicts}"
        assert max(v[1] for v in verdicts) < 30.0, verdicts
        if rank != bad:
            assert f"rank {bad}" in text, (point, pas, bad, text)     # the peers are told WHO failed
        else:
            assert "injected fault" in text or "failed to start" in text or "exchange of pass" in text, text
        ex = consume_device_early.last_exchanger
        assert ex is not None and len(ex.keep) == 0, f"{len(ex.keep)} exchange slabs still held after the failed call"
        t.clear()
        del t
    if rank == 0:
        print(f"FAULTS_OK world={world} scenarios={len(scenarios)}")


def early(path, k, per_rank, G, L, rank, world, genome, reads, dev_reads):
    """Two early-route calls (the rank's reads in two halves) into owner-sized tables; union of the ranks' tables == oracle."""
    import oracle
    from oxli_amd import KmerCountTable
    from oxli_amd.distributed import consume_device_early, global_scalar_sum

    t = KmerCountTable(k, capacity=max(G // world, 400_000))
    t.set_path(path)
    h = (per_rank // 2) & ~15                # (a multiple of 16 records: the second call's stream starts 16-byte aligned)
    half = h * (L + 1)
    # (the first call is cut into passes of 2^22 window starts: several exchanges in flight beside the counting, windows across the
    # cuts -- inside a record for long reads -- counted once)
    n1, s1 = consume_device_early(t, dev_reads.data_ptr(), half, h * L, max_windows=1 << 22, native=NATIVE)
    assert s1["passes"] == -(-(half - k + 1) // (1 << 22)) > 1, s1
    n2, s2 = consume_device_early(t, dev_reads.data_ptr() + half, dev_reads.numel() - half, (per_rank - h) * L, native=NATIVE)
    assert s1["windows_sent"] > 0 and s1["windows_received"] > 0 and s1["bytes_sent"] > 0
    assert s1["bytes_sent"] < (0.75 if k < 30 else 0.5) * 4 * s1["windows_sent"], s1        # far fewer bytes than one 4-byte entry per window
    total_n = global_scalar_sum(n1 + n2, DEV)
    expect = world * per_rank * (L - k + 1)
    assert total_n == expect, (total_n, expect)
    keys, counts = t.dump_arrays(1)          # (reading the table converts what the dedupe-first paths left pending)
    assert global_scalar_sum(t.sum_counts, DEV) == expect
    assert global_scalar_sum(t.consumed, DEV) == world * per_rank * L
    parts = [None] * world
    dist.all_gather_object(parts, (keys, counts))
    if rank == 0:
        allreads = oracle.synth_reads(genome, 0, world * per_rank, L, 1337)
        tab, _, _ = oracle.baseline_consume(allreads, L, k, min(8, len(os.sched_getaffinity(0))), native=False)
        rk, rc = tab.dump_arrays()
        gk = np.concatenate([p[0] for p in parts])
        gc = np.concatenate([p[1] for p in parts])
        order = np.argsort(gk, kind="stable")
        assert gk.size == np.unique(gk).size, "owner partitions overlap"
        assert np.array_equal(gk[order], rk) and np.array_equal(gc[order], rc), "union of the owner tables differs from the oracle"
        sizes = [int(p[0].size) for p in parts]
        assert min(sizes) > 0.5 * rk.size / world, sizes     # every owner holds about its share
        sent = s1["bytes_sent"] + s2["bytes_sent"]
        print(f"DIST_GPU_OK world={world} distinct={rk.size} route=early:{path} bytes_sent={sent} bytes_per_window={sent / max(1, s1['windows_sent'] + s2['windows_sent']):.3f}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
