#!/usr/bin/env python3
"""Randomised cross-check of the device paths (GPU box): for random k, genome size, coverage, error / N rates and call
patterns, the dedupe-first paths (forced and automatic) must build exactly the table the partitioned and direct paths
build.  Usage: python tools/stress_paths.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from oxli_amd import KmerCountTable, _lib

lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
stream = torch.cuda.current_stream().cuda_stream
t_start = time.time()
for it in range(iters):
    k = int(rng.choice([int(rng.integers(1, 33)), 21, 31, 32, 16, 17, 22, 33, 51, int(rng.integers(33, 65))]))
    L = int(rng.choice([50, 100, 150, 151, 250, 1000]))
    if L < k + 1:
        L = k + 30
    big = os.environ.get("STRESS_BIG") == "1"   # bigger genomes / passes: table-sized shadows, two partition levels
    G = int(rng.choice([2_000, 50_000, 400_000, 3_000_000] + ([12_000_000, 40_000_000] if big else [])))
    N = int(rng.integers(5_000_000 // L, (300_000_000 if big else 40_000_000) // L))
    N -= N % 16
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(N * (L + 1), dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(g.data_ptr(), G, int(rng.integers(1, 1 << 40)), stream) == 0
    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, N, L, int(rng.integers(1, 1 << 40)), stream) == 0
    # sequencing errors / N / lower case, sprinkled on the device
    err = float(rng.choice([0.0, 0.0, 0.001, 0.01]))
    if err > 0:
        m = torch.rand(r.numel(), device="cuda") < err
        sub = torch.tensor(list(b"ACGTNacgt"), dtype=torch.uint8, device="cuda")[torch.randint(0, 9, (int(m.sum()),), device="cuda")]
        keep_nl = (r == 10)
        r[m] = sub
        r[keep_nl] = 10
    torch.cuda.synchronize()
    ncalls = int(rng.choice([1, 2, 3]))
    cut = [0] + sorted(int(x) - int(x) % 16 for x in rng.integers(1, N, ncalls - 1)) + [N]
    sig = None
    # the same stream as packed base arrays (cuts are multiples of 16 reads: group-aligned)
    ng = (r.numel() + 15) // 16
    pc = torch.empty(ng, dtype=torch.int32, device="cuda")
    pv = torch.empty(ng, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(r.data_ptr(), r.numel(), pc.data_ptr(), pv.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    for path in ("partitioned", "dedupe", "auto", "direct", "packed", "routed"):
        if path == "direct" and N * L > (100_000_000 if big else 30_000_000):
            continue
        if path == "routed" and k > 64:       # the early route cuts super-k-mers for k <= 64
            continue
        cap = int(rng.choice([0, G, 4 * G])) or 0
        t = KmerCountTable(k, capacity=cap)
        # the early route's loop-back (world = 1: split, run directory, K1 over runs): the owner side on a path of the table's policy
        t.set_path("auto" if path == "packed" else str(rng.choice(["auto", "partitioned", "dedupe"])) if path == "routed" else path)
        max_windows = int(rng.choice([0, 1 << 20, 1 << 22]))
        tot = 0
        for rep in range(2):
            for a, b in zip(cut[:-1], cut[1:]):
                if b <= a:
                    continue
                if path == "packed":
                    g0 = a * (L + 1) // 16
                    tot += t.consume_device_packed(pc.data_ptr() + 4 * g0, pv.data_ptr() + 2 * g0, (b - a) * (L + 1), (b - a) * L)
                elif path == "routed":
                    n_, st_ = C.c_uint64(), (C.c_uint64 * 16)()
                    t._check(lib.kct_consume_device_routed(t._h, C.c_void_p(r.data_ptr() + a * (L + 1)), (b - a) * (L + 1), (b - a) * L, 1, 0, None,
                                                           max_windows, C.byref(n_), st_))
                    tot += n_.value
                else:
                    tot += t.consume_device(r.data_ptr() + a * (L + 1), (b - a) * (L + 1), (b - a) * L)
            if rep == 0 and rng.random() < 0.5:
                t.get_hash(12345)          # a read between the two rounds
        keys, counts = t.dump_arrays(1)
        s = (tot, keys.size, int(np.bitwise_xor.reduce(keys * counts)) if keys.size else 0, int(counts.sum()), t.consumed)
        if sig is None:
            sig = s
        assert s == sig, f"iteration {it}: path {path} differs: {s} vs {sig} (k={k} L={L} G={G} N={N} err={err} cuts={cut})"
        del t
    print(f"[{it}] k={k} L={L} G={G} N={N} err={err} calls={ncalls}: n={sig[0]} distinct={sig[1]} ok ({time.time() - t_start:.0f}s)", flush=True)
print("all paths agree")
