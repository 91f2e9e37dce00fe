"""The early route's stages on ONE GPU (loop-back exchange, world = 1 ... but the split cuts for `--owners` owners, so that runs, wire
bytes and the owner side's input are what a rank of that world sees): per-kernel device time, wire bytes per window, and the same input
through the plain single-GPU path for comparison.  The numbers DESIGN.md section 6 prices an N-GPU job with.

  python tools/route_profile.py C4 --owners 8      # 12.5 M x 150 bp, k = 21 (one rank's reads of C4)
  python tools/route_profile.py NS --owners 8      # all 100 M reads (what 8 owners together have to count)
  python tools/route_profile.py C5 --owners 8 --scale 0.25
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "C2": dict(reads=1_000_000, L=150, k=21, genome=5_000_000),
    "C4": dict(reads=12_500_000, L=150, k=21, genome=500_000_000),
    "NS": dict(reads=100_000_000, L=150, k=21, genome=500_000_000),
    "C3": dict(reads=100_000_000, L=150, k=31, genome=500_000_000),
    "C5": dict(reads=1_250_000, L=10_000, k=51, genome=3_100_000_000),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--owners", type=int, default=8)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--max-windows", type=int, default=0)
    ap.add_argument("--skip-plain", action="store_true")
    ap.add_argument("--skip-loopback", action="store_true")
    ap.add_argument("--split-only", action="store_true", help="the sender's split alone (tools/split_ablate.sh; no checks: an ablated build loses windows)")
    args = ap.parse_args()
    import torch

    from oxli_amd import KmerCountTable, _lib
    lib = _lib.load()
    c = dict(CONFIGS[args.config])
    R, L, k = max(1, int(c["reads"] * args.scale)), c["L"], args.k or c["k"]
    G = max(L + 1, int(c["genome"] * args.scale)) if args.config != "C5" else c["genome"]
    s = torch.cuda.current_stream().cuda_stream
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(g.data_ptr(), G, 42, s) == 0 and lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, R, L, 1337, s) == 0
    torch.cuda.synchronize()
    n_expect = R * (L - k + 1)
    out = {"config": args.config, "reads": R, "L": L, "k": k, "genome": G, "kmers": n_expect, "owners": args.owners}

    # ---- the split alone, for `owners` owners: runs, wire bytes --------------------------------------------------------------------------
    t = KmerCountTable(k, capacity=min(G, R * (L - k + 1)))
    t.profile(True)
    ns = lib.kct_superkmer_streams(t._h)
    W = args.owners
    parts = C.c_void_p()
    off, nb, dirs = (C.c_uint64 * W)(), (C.c_uint64 * W)(), (C.c_uint64 * (W * ns))()
    step = 1 << 30      # bytes per split call (the routed call cuts passes the same way)
    tot_bytes, tot_win, per_owner = 0, 0, [0] * W
    for rep in range(2):
        t.profile_reset()
        tot_bytes, tot_win, per_owner = 0, 0, [0] * W
        for a in range(0, r.numel(), step):
            ln = min(r.numel() - a, step + k - 1)
            t._check(lib.kct_superkmer_split_device(t._h, C.c_void_p(r.data_ptr() + a), ln, W, C.byref(parts), off, nb, dirs))
            tot_bytes += sum(nb)
            for o in range(W):
                w = sum(int(dirs[o * ns + i]) & 0xFFFFFFFF for i in range(ns))
                per_owner[o] += w
                tot_win += w
    prof = t.profile_read()
    assert args.split_only or tot_win == n_expect, (tot_win, n_expect)
    out["split"] = {"kernels_ms": {kk: round(v[1], 3) for kk, v in prof.items()}, "wire_bytes": tot_bytes, "bytes_per_window": tot_bytes / tot_win,
                    "owner_share_min_max": [min(per_owner) / tot_win * W, max(per_owner) / tot_win * W]}
    t.release_scratch()
    del t
    if args.split_only:
        print(json.dumps(out))
        return

    # ---- ONE OWNER's share: rank 0 of `owners`, counting what it owns of ALL these records (world > 1 without an exchange): its table,
    # its passes, its kernels are those of a rank in a job of `owners` GPUs whose reads, all together, are this input ---------------------
    t = KmerCountTable(k, capacity=max(G // W, 1 << 20))
    t.profile(True)
    n, st = C.c_uint64(), (C.c_uint64 * 16)()
    for rep in range(2):
        t.clear()
        t.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t._check(lib.kct_consume_device_routed(t._h, C.c_void_p(r.data_ptr()), r.numel(), R * L, W, 0, None, args.max_windows, C.byref(n), st))
        t.sync()
        dt = time.perf_counter() - t0
    prof = t.profile_read()
    assert t.sum_counts == n.value and abs(n.value - n_expect / W) < 0.1 * n_expect / W
    out["owner0_of_%d" % W] = {"seconds": dt, "kmers": n.value, "passes": int(st[5]), "split_ms": st[6] / 1e3, "owner_ms": st[8] / 1e3, "slots": t.capacity,
                               "kernels_ms": {kk: round(v[1], 3) for kk, v in prof.items()}}
    t.release_scratch()
    del t
    torch.cuda.empty_cache()
    if args.skip_loopback:
        print(json.dumps(out))
        return
    # ---- the whole route as a loop-back (every window comes back to this GPU) ------------------------------------------------------------
    t = KmerCountTable(k, capacity=G)
    t.profile(True)
    for rep in range(2):
        t.clear()
        t.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t._check(lib.kct_consume_device_routed(t._h, C.c_void_p(r.data_ptr()), r.numel(), R * L, 1, 0, None, args.max_windows, C.byref(n), st))
        t.sync()
        dt = time.perf_counter() - t0
    prof = t.profile_read()
    assert n.value == n_expect and t.sum_counts == n_expect
    lo, hi, sq = t._count_stats()
    sig = (lo, hi, len(t), sq)
    out["loopback"] = {"seconds": dt, "kmers_per_s": n_expect / dt, "passes": int(st[5]), "split_ms": st[6] / 1e3, "owner_ms": st[8] / 1e3,
                       "kernels_ms": {kk: round(v[1], 3) for kk, v in prof.items()}}
    t.release_scratch()
    del t
    torch.cuda.empty_cache()
    if not args.skip_plain:
        t = KmerCountTable(k, capacity=G)
        t.profile(True)
        for rep in range(2):
            t.clear()
            t.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert t.consume_device(r.data_ptr(), r.numel(), R * L) == n_expect
            t.sync()
            dt = time.perf_counter() - t0
        prof = t.profile_read()
        lo, hi, sq = t._count_stats()
        assert (lo, hi, len(t), sq) == sig, "the route and the plain path disagree"
        out["plain"] = {"seconds": dt, "kmers_per_s": n_expect / dt, "kernels_ms": {kk: round(v[1], 3) for kk, v in prof.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
