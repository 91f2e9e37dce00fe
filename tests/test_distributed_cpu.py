"""world_size-2 gloo test of the multi-GPU shard merge logic (runs on CPU).

The device table is not available here, so each rank holds an ORACLE table for its shard of the
reads; what is under test is oxli_amd.distributed's owner function, bucketing and all-to-all
exchange -- the same code the GPU path runs over RCCL -- and that the union of the owner
partitions equals the single-table result (add() semantics, lib.rs:778-837)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir, fallback=False):
    sys.path.insert(0, ROOT)
    if fallback:
        os.environ["KCT_A2A_FALLBACK"] = "1"  # the all_gather route of exchange_pairs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oxli_amd.distributed import exchange_pairs, global_scalar_sum, owner_of, partition_by_owner

    k, L, per_rank = 21, 150, 3000
    genome = oracle.synth_genome(40000)
    reads = oracle.synth_reads(genome, rank * per_rank, per_rank, L)
    mine = oracle.OracleTable(k)
    n = sum(mine.consume(reads[i, :L]) for i in range(per_rank))
    zero = 3 if rank == 1 else 0  # pretend count_hash(0) was called on rank 1
    keys, counts = mine.dump_arrays()
    h = torch.from_numpy(keys.view(np.int64).copy())
    c = torch.from_numpy(counts.view(np.int64).copy())
    pairs, send = partition_by_owner(h, c, world)
    assert int(send.sum()) == keys.size
    assert torch.all(owner_of(pairs[:, 0], world)[1:] >= owner_of(pairs[:, 0], world)[:-1])
    recv, zero_total = exchange_pairs(pairs, send, zero)
    assert torch.all(owner_of(recv[:, 0], world) == rank)
    assert zero_total == (3 if rank == 0 else 0)
    owned = oracle.OracleTable(k)
    owned.add_pairs(recv[:, 0].contiguous().numpy().view(np.uint64), recv[:, 1].contiguous().numpy().view(np.uint64))
    ok, oc = owned.dump_arrays()
    np.save(os.path.join(outdir, f"keys{rank}.npy"), ok)
    np.save(os.path.join(outdir, f"counts{rank}.npy"), oc)
    total = global_scalar_sum(owned.sum_counts, "cpu")
    assert total == world * per_rank * (L - k + 1) == global_scalar_sum(n, "cpu")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fallback", [(2, False), (3, False), (2, True)])
def test_owner_partitioned_merge_equals_single_table(tmp_path, world, fallback):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), fallback), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import oracle
    k, L, per_rank = 21, 150, 3000
    genome = oracle.synth_genome(40000)
    reads = oracle.synth_reads(genome, 0, world * per_rank, L)
    ref = oracle.OracleTable(k)
    for i in range(reads.shape[0]):
        ref.consume(reads[i, :L])
    rk, rc = ref.dump_arrays()
    keys = np.concatenate([np.load(tmp_path / f"keys{r}.npy") for r in range(world)])
    counts = np.concatenate([np.load(tmp_path / f"counts{r}.npy") for r in range(world)])
    # owner slices are contiguous in hash space and disjoint, so concatenating the sorted
    # per-owner dumps in rank order gives the globally sorted dump
    assert np.array_equal(keys, rk)
    assert np.array_equal(counts, rc)


def test_owner_function_is_a_partition():
    from oxli_amd.distributed import owner_of
    rng = np.random.default_rng(0)
    h = torch.from_numpy(rng.integers(0, 2**63, size=20000, dtype=np.int64) * 2 + rng.integers(0, 2, size=20000))
    for world in (1, 2, 3, 4, 8):
        own = owner_of(h, world)
        assert int(own.min()) >= 0 and int(own.max()) < world
        if world > 1:
            assert len(torch.unique(own)) == world
    edge = torch.tensor([0, -1, 2**63 - 1, -(2**63)], dtype=torch.int64)  # 0, 2^64-1, 2^63-1, 2^63 as bit patterns
    assert owner_of(edge, 8).tolist() == [0, 7, 3, 4]
