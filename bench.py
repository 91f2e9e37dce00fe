#!/usr/bin/env python3
"""bench.py -- k-mers/sec through consume (BASELINE.json metric) on N MI355X of one node.

The job: every rank streams batches of synthetic reads (already resident in HBM) into its own
device table with kct_consume_device -- a "step" is one batch of R reads per GPU -- and, when
N > 1, ONE final owner-partitioned RCCL all-to-all turns the per-rank tables into the global table
(oxli_amd.distributed.merge_across_ranks; reference semantics: add(), lib.rs:778-837).  The table
is empty when the timed region starts, the final merge is INSIDE the timed region, and `value` =
all k-mers counted by all ranks / that time.  Work per GPU is fixed as N grows (weak scaling):
step s of rank r counts reads [(s*N + r)*R, (s*N + r + 1)*R) of one read stream, so no read is
counted twice and every step brings new reads (up to --batches distinct batches per rank are kept
resident and cycled).

Default workload = BASELINE.json configs[1] ("C2"): R = 1 M reads x 150 bp, k = 21, reads drawn
from a 5 Mbp synthetic genome (SEED_G 42, SEED_R 1337), device table sized for 5 M distinct keys.

The JSON line also carries
  roofline     : the dominant kernel (count_windows_kernel), its ALGORITHMIC bytes per launch
                 (k-mers per launch x (L/(L-k+1) + 24) B) over its average launch
                 duration measured with HIP events on the table's stream inside the timed region.
  cpu_baseline : the CPU restatement of the reference path (oracle/, "port") timed on this
                 host on a bounded sample of the same read stream (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_G, SEED_R = 42, 1337
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    p.add_argument("--read-len", type=int, default=150)
    p.add_argument("--k", type=int, default=21)
    p.add_argument("--genome", type=int, default=5_000_000)
    p.add_argument("--cpu-sample-reads", type=int, default=200_000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--batches", type=int, default=8, help="distinct read batches resident per GPU (cycled over the steps)")
    p.add_argument("--path", choices=["auto", "direct", "partitioned", "dedupe"], default="auto")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo lets several ranks share one GPU for debugging")
    return p.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/*.json), else None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel.replace("<shadow>", ""), {}).get("hbm_bytes_per_launch")  # (K2 on the shadow table is K2)
    except (OSError, ValueError):
        return None


def cpu_baseline(args, log):
    """Reference-shaped CPU path (oracle 'port') on a bounded sample of the same stream."""
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:  # noqa: BLE001 -- fall back to the portable build that travelled with the repo
        native = False
    genome = oracle.synth_genome(args.genome, SEED_G)
    cores = len(os.sched_getaffinity(0))
    n1 = min(args.cpu_sample_reads, args.reads)
    reads1 = oracle.synth_reads(genome, 0, n1, args.read_len, SEED_R)
    _, km1, s1 = oracle.baseline_consume(reads1, args.read_len, args.k, 1, native)
    log(f"cpu baseline: 1 thread {km1 / s1 / 1e6:.2f} Mk-mers/s on {n1} reads")
    # "rayon-style" best case: private table per thread over contiguous shards + parallel tree merge;
    # the best thread count is found by trying a few (more threads = more duplicate keys to merge)
    nT = min(args.reads, 1_000_000)
    readsT = reads1 if nT == n1 else oracle.synth_reads(genome, 0, nT, args.read_len, SEED_R)
    tried, tabT = {}, None
    for T in sorted({t for t in (8, 32, 128, cores) if 1 < t <= cores}):
        tabT, km, s = oracle.baseline_consume(readsT, args.read_len, args.k, T, native)
        tried[T] = km / s
        log(f"cpu baseline: {T} threads {km / s / 1e6:.2f} Mk-mers/s on {nT} reads")
    if tabT is None:  # single-core host: the verification table still has to cover the nT-read sample
        tabT, _, _ = oracle.baseline_consume(readsT, args.read_len, args.k, 1, native)
    rates = dict(tried)
    rates[1] = km1 / s1
    best_T = max(rates, key=rates.get)
    best = (rates[best_T], best_T)
    out = {"value": best[0], "unit": "k-mers/s", "cores": best[1], "kind": "port",
           "sample": f"first {nT} reads of the same stream; private table per thread + parallel tree merge with add() "
                     f"semantics, merge timed; thread counts tried {sorted(tried)} of {cores} host cores; "
                     f"1 thread on first {n1} reads: {km1 / s1:.4g} k-mers/s",
           "value_1thread": km1 / s1, "by_threads": {str(k): v for k, v in tried.items()}, "host_cores": cores,
           "native_build": native}
    return out, (readsT, tabT)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from oxli_amd import KmerCountTable, _lib
    from oxli_amd.distributed import global_scalar_sum, merge_across_ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    lib = _lib.load()
    L, k, R, G = args.read_len, args.k, args.reads, args.genome
    kmers_per_step = R * (L - k + 1)
    stream = torch.cuda.current_stream().cuda_stream
    genome = torch.empty(G, dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(genome.data_ptr(), G, SEED_G, stream) == 0
    nb = max(1, min(args.batches, args.steps))
    batches = []
    for b in range(nb):
        reads = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
        assert lib.kct_synth_reads_device(reads.data_ptr(), genome.data_ptr(), G, (b * world + rank) * R, R, L, SEED_R, stream) == 0
        batches.append(reads)
    torch.cuda.synchronize()

    table = KmerCountTable(k, capacity=G, device=local)
    table.set_stream(stream)
    table.set_path(args.path)

    def step(s):
        reads = batches[s % nb]
        return table.consume_device(reads.data_ptr(), reads.numel(), R * L)

    for s in range(args.warmup):
        step(s)
    if world > 1 and args.warmup:
        merge_across_ranks(table)  # warm the collective and the merge kernels too
    table.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_total = 0
    for s in range(args.steps):
        n_total += step(s)
    t_merge = time.perf_counter()
    if world > 1:
        merge_across_ranks(table)
    table.sync()  # counts still pending in the dedupe-first path's shadow table are converted inside the timed region
    torch.cuda.synchronize()
    merge_ms = (time.perf_counter() - t_merge) * 1e3
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n_all = global_scalar_sum(n_total, "cuda")
    else:
        n_all = n_total
    ablate = bool(os.environ.get("KCT_ABLATE"))  # timing experiments that deliberately skip work: no result checks
    assert ablate or n_total == kmers_per_step * args.steps, (n_total, kmers_per_step * args.steps)

    # invariants of the finished job (cheap, outside the timed region)
    distinct = global_scalar_sum(len(table), "cuda") if world > 1 else len(table)
    total_counts = global_scalar_sum(table.sum_counts, "cuda") if world > 1 else table.sum_counts
    assert ablate or total_counts == world * kmers_per_step * args.steps, (total_counts, world * kmers_per_step * args.steps)

    # Per-kernel device times for the roofline: the SAME steps once more with the library's HIP-event timing switched on
    # (an event pair around every launch costs ~6 % at this step size, so it stays out of the region `value` is taken from).
    table.clear()
    table.profile(True)
    table.profile_reset()
    for s in range(args.steps):
        step(s)
    table.sync()
    torch.cuda.synchronize()
    prof = table.profile_read()
    table.profile(False)

    value = n_all / elapsed
    b_alg = L / (L - k + 1) + 24.0
    # dominant kernel = the one with the most device time inside the timed region
    dom = max(prof, key=lambda n: prof[n][1]) if prof else "none"
    launches, ms = prof.get(dom, (0, 0.0))
    avg_ms = ms / launches if launches else float("nan")
    kmers_per_launch = kmers_per_step * args.steps / launches if launches else 0
    achieved = kmers_per_launch * b_alg / (avg_ms * 1e-3) / 1e9 if launches else float("nan")
    all_ms = sum(v[1] for v in prof.values())
    pipe_gbs = kmers_per_step * args.steps * b_alg / (all_ms * 1e-3) / 1e9 if all_ms else float("nan")
    # The step is a PIPELINE of kernels that each see every k-mer (partition, then count): dividing the algorithmic bytes
    # by the dominant kernel's time alone would flatter the path (and exceeds the peak once a kernel no longer hashes), so
    # `achieved` / `frac` divide by the SUM of all kernels' device time per step; the dominant kernel's own figures,
    # measured HBM traffic included, sit in `dominant_kernel`.
    step_traffic = [pmc_traffic(n) for n in prof]
    traffic_step = (sum(t * prof[n][0] for n, t in zip(prof, step_traffic) if t) / args.steps) if any(step_traffic) else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": pipe_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": pipe_gbs / HBM_PEAK_GBS, "traffic": traffic_step,
                "basis": "algorithmic bytes per step (25.15 B/k-mer x k-mers) / sum of the device time of every kernel of the step, HIP events around "
                         "every launch of a repetition of the timed steps (event timing off while `value` is taken); "
                         "traffic = PMC-measured HBM bytes per step summed over the kernels (profiles/pmc_traffic.json)",
                "alg_bytes_per_kmer": b_alg, "kmers_per_step": kmers_per_step, "kernel_ms_per_step": all_ms / args.steps,
                "dominant_kernel": {"name": dom, "avg_launch_ms": avg_ms, "launches": launches, "kmers_per_launch": kmers_per_launch,
                                    "achieved_alone": achieved, "frac_alone": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom)},
                "kernels_ms_per_step": {n: round(v[1] / args.steps, 4) for n, v in prof.items()}}

    result = {
        "metric": "k-mers/sec (consume) at k=%d, %d bp reads" % (k, L),
        "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic" if not ablate else "INVALID (KCT_ABLATE set: work skipped)",
        "config": {"workload": f"C2: {R} x {L} bp synthetic reads per GPU, k={k}, genome {G} bp (seed {SEED_G}/{SEED_R}), "
                               f"device hash table in HBM ({table.capacity} slots x 16 B)",
                   "reads_per_gpu": R, "read_len": L, "k": k, "genome": G, "distinct_kmers": distinct,
                   "distinct_batches_per_gpu": nb,
                   "step": "consume one batch of reads into the rank's table" +
                           (f"; one final RCCL owner all-to-all merge inside the timed region ({merge_ms:.3f} ms on rank 0)" if world > 1 else "")},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        base, (sample_reads, sample_table) = cpu_baseline(args, log)
        result["cpu_baseline"] = base
        result["speedup_vs_cpu_baseline"] = value / base["value"]
        if not args.no_verify:
            # the same sample through the GPU must give the oracle's table bit for bit
            ns = sample_reads.shape[0]
            table.clear()
            n = table.consume_device(batches[0].data_ptr(), ns * (L + 1), ns * L)
            dk, dc = table.dump_arrays(1)
            rk, rc = sample_table.dump_arrays()
            ok = n == ns * (L - k + 1) and np.array_equal(dk, rk) and np.array_equal(dc, rc)
            result["verified_vs_oracle"] = bool(ok)
            assert ok, "GPU table differs from the CPU oracle on the baseline sample"
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
