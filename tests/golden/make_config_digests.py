#!/usr/bin/env python3
"""Generates tests/golden/config_digests.json: the CPU ORACLE's digests of BASELINE.json's configurations at full size.

For every configuration the oracle's table of the whole synthetic input (``oracle.ShardSet``: the reference's per-record loop,
lib.rs:576-600 + 100-104, key space sharded over host threads, reads generated on the fly from SEED_G = 42 / SEED_R = 1337) is reduced
to order-free digests: len, sum_counts, n, consumed, min / max count, sum(hash * count), xor(hash * count), sum(count^2), all mod 2^64.
bench.py's full-size gates and tests/test_gpu_scale.py compare the device tables with these numbers -- and the multi-GPU C4 job's
union of owner tables with NS-k21's (the same 100 M reads, split over the ranks).

Needs a host with many cores and ~40 GiB of memory (the GPU box's host: ~1 minute per 10^10 k-mers on 64 threads); nothing here
touches the GPU or /root/reference.

    python tests/golden/make_config_digests.py [--out tests/golden/config_digests.json] [names...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SEED_G, SEED_R = 42, 1337
CONFIGS = {   # name -> reads, read length, k, genome
    "C2": (1_000_000, 150, 21, 5_000_000),
    "C4_shard": (12_500_000, 150, 21, 500_000_000),
    "north_star_k21": (100_000_000, 150, 21, 500_000_000),     # = the multi-GPU C4 job's union
    "C3": (100_000_000, 150, 31, 500_000_000),
    "C5_shard": (1_250_000, 10_000, 51, 387_500_000),
    "C5": (10_000_000, 10_000, 51, 3_100_000_000),             # the multi-GPU C5 job's union (needs ~120 GiB of host memory)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "config_digests.json"))
    ap.add_argument("names", nargs="*", default=[n for n in CONFIGS if n != "C5"])
    args = ap.parse_args()
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:  # noqa: BLE001
        native = False
    out = {}
    if os.path.exists(args.out):
        out = json.load(open(args.out))
    genomes = {}
    for name in args.names:
        R, L, k, G = CONFIGS[name]
        if G not in genomes:
            genomes.clear()
            genomes[G] = oracle.synth_genome(G, SEED_G)
        t0 = time.time()
        ss = oracle.ShardSet(k, L, genome=genomes[G], nreads=R, seed_r=SEED_R, expect_keys=min(G, R * (L - k + 1)), native=native)
        d = ss.digest()
        del ss
        d.update(reads=R, read_len=L, k=k, genome=G, seed_g=SEED_G, seed_r=SEED_R)
        out[name] = d
        print(f"{name}: {d}  ({time.time() - t0:.0f} s)", flush=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
