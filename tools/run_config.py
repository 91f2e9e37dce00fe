#!/usr/bin/env python3
"""Runs one of BASELINE.json's configurations on ONE GPU at full (or scaled) size and prints a JSON line:
k-mers/s per path plus the size-independent checks (n == reads x windows, sum_counts == n, both paths
give the same number of distinct keys and the same XOR/SUM checksums over (hash, count)).

  python tools/run_config.py C3            # 100 M x 150 bp, k=31, genome 500 Mbp, table 2^30 slots (16 GiB)
  python tools/run_config.py C3 --scale 0.1
  python tools/run_config.py C5            # one GPU's share of C5: 1.25 M x 10 kbp, k=51, genome 3.1 Gbp / 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "C2": dict(reads=1_000_000, L=150, k=21, genome=5_000_000),
    "C3": dict(reads=100_000_000, L=150, k=31, genome=500_000_000),
    "C4": dict(reads=12_500_000, L=150, k=21, genome=500_000_000),     # one GPU's shard of C4
    "C5": dict(reads=1_250_000, L=10_000, k=51, genome=387_500_000),   # one GPU's shard of C5 (3.1 Gbp / 8 of key space)
    "NS": dict(reads=100_000_000, L=150, k=21, genome=500_000_000),    # the north-star sentence: 100 M x 150 bp, k=21, ONE GPU
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--paths", default="auto,direct")
    ap.add_argument("--no-dump", action="store_true", help="compare min / max / len / sum of squares instead of dumping the table")
    args = ap.parse_args()
    import numpy as np
    import torch

    from oxli_amd import KmerCountTable, _lib
    c = dict(CONFIGS[args.config])
    R, L, k = max(1, int(c["reads"] * args.scale)), c["L"], c["k"]
    G = max(L + 1, int(c["genome"] * args.scale))
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    genome = torch.empty(G, dtype=torch.uint8, device="cuda")
    reads = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(genome.data_ptr(), G, 42, stream) == 0
    assert lib.kct_synth_reads_device(reads.data_ptr(), genome.data_ptr(), G, 0, R, L, 1337, stream) == 0
    torch.cuda.synchronize()
    n_expect = R * (L - k + 1)
    out = {"config": args.config, "scale": args.scale, "reads": R, "read_len": L, "k": k, "genome": G, "kmers": n_expect,
           "stream_bytes": reads.numel(), "paths": {}}
    sig = None
    for path in args.paths.split(","):
        t = KmerCountTable(k, capacity=G)
        t.set_path(path)
        t.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = t.consume_device(reads.data_ptr(), reads.numel(), R * L)
        dt_first = time.perf_counter() - t0   # includes the one-time scratch / spill allocations
        assert n == n_expect, (n, n_expect)
        t.clear()
        t.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = t.consume_device(reads.data_ptr(), reads.numel(), R * L)
        t.sync()  # counts still pending in the dedupe-first path's shadow table are converted inside the timed region
        dt = time.perf_counter() - t0
        prof = t.profile_read()
        assert n == n_expect, (n, n_expect)
        assert t.sum_counts == n
        distinct = len(t)
        # checksum of the table without a full host sort: XOR and wrapping SUM of hash * count
        if args.no_dump:   # (a 5x10^8-key dump is 8 GB over PCIe: the statistics the device computes instead)
            lo, hi, sq = t._count_stats()
            s = (lo, hi, distinct, sq)
        else:
            keys, counts = t.dump_arrays(0)
            prod = keys * counts  # uint64 wraps
            s = (int(np.bitwise_xor.reduce(prod)), int(prod.sum(dtype=np.uint64)), distinct)
        if sig is None:
            sig = s
        assert s == sig, f"paths disagree: {s} vs {sig}"
        out["paths"][path] = {"seconds": dt, "seconds_first_call": dt_first, "kmers_per_s": n / dt, "slots": t.capacity, "distinct": distinct,
                              "kernels_ms": {kk: round(v[1], 3) for kk, v in prof.items()}}
        t.release_scratch()
        del t
        torch.cuda.empty_cache()
    out["signature"] = list(sig)
    out["distinct"] = sig[2]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
