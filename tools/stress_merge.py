#!/usr/bin/env python3
"""Randomised cross-check of the pair-merge routes (GPU box): for random k, genome sizes and capacities, a table built by
consuming read set A and then `add()`ing a table built from read set B must equal the table that consumed A and B itself;
the same through export -> merge_pairs_device (the multi-GPU merge's device side, pairs in source-slot order) into an
EMPTY owner-sized table, and through a shuffled copy of the pair list.  Usage: python tools/stress_merge.py [iterations] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oxli_amd import KmerCountTable, _lib

lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
stream = torch.cuda.current_stream().cuda_stream
t_start = time.time()


def signature(t):
    keys, counts = t.dump_arrays(1)
    return (keys.size, int(np.bitwise_xor.reduce(keys * counts)) if keys.size else 0, int(counts.sum()), t.consumed)


for it in range(iters):
    k = int(rng.choice([21, 31, 51, int(rng.integers(5, 65))]))
    L = int(rng.choice([100, 150, 250]))
    if L < k + 1:
        L = k + 30
    G = int(rng.choice([20_000, 400_000, 3_000_000, 12_000_000, 40_000_000]))
    cover = float(rng.choice([0.5, 2.0, 8.0]))
    N = max(16, int(G * cover / L)) & ~15
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(g.data_ptr(), G, int(rng.integers(1, 1 << 40)), stream) == 0
    sets = []
    for s in range(2):
        r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
        assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, s * N, N, L, int(rng.integers(1, 1 << 40)), stream) == 0
        sets.append(r)
    torch.cuda.synchronize()
    caps = [int(rng.choice([0, G // 4, G, 4 * G])) for _ in range(3)]

    def build(which, cap):
        t = KmerCountTable(k, capacity=cap)
        for s in which:
            t.consume_device(sets[s].data_ptr(), sets[s].numel(), N * L)
        return t

    both = build((0, 1), caps[0])
    want = signature(both)
    a, b = build((0,), caps[1]), build((1,), caps[2])
    sig_b = signature(b)
    a.add(b)
    assert signature(a) == want, f"iteration {it}: add() differs (k={k} L={L} G={G} N={N} caps={caps})"
    assert signature(b) == sig_b, f"iteration {it}: add() changed its source"
    # the multi-GPU merge's device side: bucketed export of `both`, merged into an empty table sized for the pair count
    n = len(both)
    world = int(rng.choice([2, 3, 8]))
    pairs = torch.empty((max(n, 1), 2), dtype=torch.int64, device="cuda")
    pc = np.zeros(world, dtype=np.uint64)
    got = C.c_uint64()
    assert lib.kct_export_by_owner_device(both._h, world, C.c_void_p(pairs.data_ptr()), n, pc.ctypes.data, C.byref(got)) == 0
    for shuffled in (False, True):
        src = pairs[: got.value]
        if shuffled and got.value:
            src = src[torch.randperm(got.value, device="cuda")].contiguous()
        dst = KmerCountTable(k)
        dst.resize(max(int(got.value), 1))
        x, y = C.c_uint64(), C.c_uint64()
        assert lib.kct_merge_pairs_device(dst._h, C.c_void_p(src.data_ptr()), int(got.value), C.byref(x), C.byref(y)) == 0
        z = both.get_hash(0)
        if z:
            zk, zc = np.zeros(1, dtype=np.uint64), np.array([z], dtype=np.uint64)
            assert lib.kct_merge_host(dst._h, zk.ctypes.data, zc.ctypes.data, 1, None, None) == 0
        lib.kct_add_consumed(dst._h, both.consumed)
        assert signature(dst) == want, f"iteration {it}: export -> merge differs (shuffled={shuffled}, k={k} G={G} N={N} world={world})"
    print(f"[{it}] k={k} L={L} G={G} N={N} caps={caps} world={world}: distinct={want[0]} ok ({time.time() - t_start:.0f}s)", flush=True)
    del both, a, b, sets, g
print("all merges agree")
