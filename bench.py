#!/usr/bin/env python3
"""bench.py -- k-mers/sec through consume (BASELINE.json metric) on N MI355X of one node.

HEADLINE (`value`, `ms_per_step`): BASELINE.json configs[1] ("C2") -- every rank streams K batches ("steps") of
R = 1 M synthetic 150 bp reads, already resident in HBM, into its own device table with kct_consume_device at k = 21
(reads drawn from a 5 Mbp synthetic genome, SEED_G 42 / SEED_R 1337; table sized for 5 M distinct keys), then converts
what the dedupe-first path left pending; when N > 1 ONE owner-partitioned RCCL all-to-all (oxli_amd.distributed.
merge_across_ranks; reference semantics: add(), lib.rs:778-837) turns the per-rank tables into the global table.
That K-step JOB -- table empty when the clock starts, exactly K steps, conversion and merge inside the timed region,
barrier + synchronize on both sides, max over ranks -- is REPEATED until >= 0.5 s have been timed; `value` is the
median job's rate (min / max / repeats beside it).  Step s of rank r counts reads [(s*N + r)*R, (s*N + r + 1)*R) of
one read stream: as many distinct batches as steps (up to 64) are resident, so no read is counted twice inside a job.
The table keeps what earlier passes taught it about its input (the dedupe hint), so this is the STEADY-STATE rate of a
table that is filled again and again; the hint-free rate is `configs.cold_C2`.

`configs` (rank 0, N = 1): the other workloads BASELINE.json / the north-star sentence name, each timed on a table
whose buffers exist (first call untimed) but which is empty and knows nothing about its input (`kmers_per_s`, "cold"),
each with per-kernel device times, the algorithmic HBM fraction and a correctness gate (n, sum_counts, and -- the large
ones -- len / min / max / sum of squared counts and ~10^6 sampled keys equal to the DIRECT path's table, plus an oracle
slice when the CPU checker is enabled):
  cold_C2        one 1 M-read pass into an empty, hint-free table + conversion
  packed_C2      the headline's steps with the batch resident as PACKED base arrays (2 bits + 1 validity bit per base)
  e2e_C2         the same batch from HOST memory through kct_consume_batch (SIMD pack to 0.375 B/base + H2D + count; PCIe-inclusive)
  per_record     the reference's own loop, `for rec: table.consume(rec)` (README.md:96-98), on a default table
  file_fasta / file_gz / file_bgzf   the reference's documented file loop (README.md:89-99) as ONE call, kct_consume_file: the C2 batch written
                 once to /dev/shm as FASTA (160 MB), as one gzip -1 member and as BGZF (bgzip's blocked gzip), parsed (and inflated) on the
                 host, uploaded and counted; gated on the oracle's C2 digests; `inflater` says whether libdeflate.so.0 was found on the
                 box (the zlib fallback is timed beside it: a single deflate stream cannot be inflated in parallel)
  north_star_k21 100 M x 150 bp, k=21, genome 500 Mbp, one GPU (the north-star sentence)
  C3             100 M x 150 bp, k=31, genome 500 Mbp        (BASELINE.json configs[2])
  C4_shard       one GPU's eighth of configs[3]: 12.5 M x 150 bp, k=21, same genome
  C5_shard       one GPU's eighth of configs[4]: 1.25 M x 10 kbp, k=51
  C2_sub1pct / C2_N1pct / C2_sorted / NS25_sub1pct   inputs the path heuristics were NOT tuned on (the device generator's error
                 model, include/kct_synth.h): 1 % substitution errors, 1 % N, position-sorted reads at C2 size; 25 M reads of the
                 500 Mbp genome with 1 % substitutions.  Each reports the path the table chose, the rate, the rate of
                 `--path partitioned` on the same input (`vs_partitioned`) and is gated against the partitioned and direct paths'
                 tables (n, len, sum_counts, sum / xor / square digests) and -- C2-sized -- the oracle's table pair by pair.
With N > 1 ranks (`python bench.py --gpus N` launches itself under torch.distributed.run, or the driver does) the headline is the
same weak-scaled C2 job per rank + the late merge, and
  C4 / C5        BASELINE.json configs[3] / [4] as ONE strong-scaled job over the N ranks (100 M x 150 bp k=21 / 10 M x 10 kbp k=51,
                 reads split N ways), on both multi-GPU routes: "late" (private tables, then the owner all-to-all of
                 {hash, count} pairs) and "early" (super-k-mers travel to the GPU that owns them -- about a byte per window -- and are
                 counted there; passes pipelined); exchange and conversion inside the timed region; per rank: windows and bytes sent /
                 received, ms in the split / blocked in the exchange / in the owner's counting; gated on n, sum_counts, on the routes
                 agreeing on the global len and digests AND on the CPU oracle's digests of the job's union (tests/golden/
                 config_digests.json at full size; computed on rank 0's host for the reduced sizes).  Ranks that SHARE a GPU
                 (--backend gloo on a one-GPU box) run a reduced size and say so.  C5 runs only when named (--configs C4,C5): its late
                 route needs a 128 GiB private table per rank.
  C5_whole       BASELINE.json configs[4] at its OWN size on ONE GPU: 10 M x 10 kbp, k=51, genome 3.1 Gbp -- 9.95x10^10 k-mers into one
                 2^33-slot (128 GiB) table.  The 100 GB of reads do not fit beside the table and its scratch, so they are generated on
                 the device in 12 pieces (each one pass beside that table) and consumed call after call; `seconds` = the summed wall time of
                 the twelve consume calls + the final sync (each piece resident in HBM when its call starts; generation untimed); gated on
                 the CPU oracle's committed digests of the whole input (tests/golden/config_digests.json "C5").
  north_star_streamed   the north-star run's 100 M reads fed the way a caller feeds them: 20 calls of 5 M reads and 100 calls of 1 M
                 reads into ONE table that starts empty and hint-free (one conversion at the end); rate against the one-call run,
                 the path every call took, gated on the oracle's digests.

`roofline`: bound "hbm"; `achieved` = ALGORITHMIC bytes per step (k-mers x (L/(L-k+1) + 24) B) / the summed
device time of every kernel of the step (HIP events on the table's stream, in an instrumented repetition of the job;
`value` is taken with the events off); `traffic` / `measured_frac` = PMC-measured HBM bytes per step (profiles/
pmc_r06.json, only if it was collected from THIS source tree -- stamped with a hash of the kernel sources -- else null);
`valu` = the ceiling that actually binds K1 (VALU instructions per window and issue-slot use from the same PMC file).
`cpu_baseline`: the CPU restatement of the reference path (oracle/, "port") on this host: 1 thread (the reference's
consume is single-threaded under the GIL), reads sharded over threads with a tree merge (reference-shaped "rayon"), and
the key space sharded over all cores with nothing to merge (the best CPU row; `value`), medians of 3.
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_G, SEED_R = 42, 1337
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured for a plain copy)

BIG = {  # name -> reads, read length, k, genome
    "north_star_k21": (100_000_000, 150, 21, 500_000_000),
    "C3": (100_000_000, 150, 31, 500_000_000),
    "C4_shard": (12_500_000, 150, 21, 500_000_000),
    "C5_shard": (1_250_000, 10_000, 51, 387_500_000),
}
# SURVEY 8d's secondary inputs, timed (the heuristics -- probe, dedupe hint, windows_since_read -- were tuned on error-free
# reads): name -> (reads, read length, k, genome, error model of include/kct_synth.h)
ERR = {
    "C2_sub1pct": (1_000_000, 150, 21, 5_000_000, dict(sub_ppm=10_000)),
    "C2_N1pct": (1_000_000, 150, 21, 5_000_000, dict(n_ppm=10_000)),
    "C2_sorted": (1_000_000, 150, 21, 5_000_000, dict(sorted_total=1_000_000)),
    "NS25_sub1pct": (25_000_000, 150, 21, 500_000_000, dict(sub_ppm=10_000)),
}
# N > 1 ranks only: BASELINE.json configs[3] / [4] as ONE job over all ranks (strong scaling: the reads are split N ways)
MULTI = {
    "C4": (100_000_000, 150, 21, 500_000_000),
    "C5": (10_000_000, 10_000, 51, 3_100_000_000),
}
# BASELINE.json configs[4] WHOLE on one GPU: the 100 GB of reads are generated on the device in PIECES and fed call after call
# (12 pieces: a piece's 8.3x10^9 window starts are ONE pass beside the 128 GiB table -- a pass is bounded by its ~12.5 B of scratch per window
# start -- and every pass re-reads and re-writes the whole table; with the multi-GPU job's 8 rank shards as pieces each takes two passes, 16 in all)
C5_WHOLE = (10_000_000, 10_000, 51, 3_100_000_000, 12)   # reads, read length, k, genome, pieces
FILES = ["file_fasta", "file_gz", "file_bgzf"]
ALL_CONFIGS = ["cold_C2", "packed_C2", "e2e_C2", "per_record"] + FILES + ["k51_deep"] + list(BIG) + ["north_star_streamed", "C5_whole"] + list(ERR) + list(MULTI)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    p.add_argument("--read-len", type=int, default=150)
    p.add_argument("--k", type=int, default=21)
    p.add_argument("--genome", type=int, default=5_000_000)
    p.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step job until this much has been timed")
    p.add_argument("--max-repeats", type=int, default=400)
    p.add_argument("--configs", default="all", help="'all', 'none' or a comma list of " + ",".join(ALL_CONFIGS))
    p.add_argument("--no-headline", action="store_true", help="profiling aid: run only --configs")
    p.add_argument("--no-cpu-baseline", action="store_true", help="also drops the oracle slices of the correctness gates")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--cpu-reads-1t", type=int, default=200_000)
    p.add_argument("--cpu-reads-mt", type=int, default=2_000_000)
    p.add_argument("--path", choices=["auto", "direct", "partitioned", "dedupe"], default="auto")
    p.add_argument("--verbose", action="store_true", help="print the full record instead of the compact line")
    p.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record is written")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo lets several ranks share one GPU for debugging")
    p.add_argument("--exchange", choices=["auto", "native", "torch", "both"], default="auto",
                   help="N > 1: who moves the data between ranks -- libkct_rccl.so (native: ncclSend / ncclRecv groups inside the library, what a Rust "
                        "caller links; needs --backend nccl, one rank per GPU) or torch.distributed.  auto = native for the headline's merge where "
                        "possible, and BOTH side by side in configs.C4 / C5")
    p.add_argument("--job-timeout", type=float, default=1800.0, help="N > 1 self-launch: seconds after which the parent kills the ranks (a hung collective) and exits 124")
    p.add_argument("--no-second-process", action="store_true", help="skip the headline's second sample from another cold-started process")
    p.add_argument("--headline-sample", action="store_true", help=argparse.SUPPRESS)   # (the child of the above: headline only, prints its value)
    return p.parse_args()


def source_sha():
    """Hash of the DEVICE code (the kernel headers): ties a committed PMC summary to the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "oxli_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith(".h") and name not in ("kct_internal.h", "path_policy.h", "parallel_inflate.h"):  # (host-only headers do not change what was measured)
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_summary():
    """profiles/pmc_r06.json (tools/collect_r06.sh) if it was collected from this source tree, else {} (every PMC-derived field becomes null)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_r06.json")) as f:
            d = json.load(f)
        return d if d.get("source_sha") == source_sha() else {}
    except (OSError, ValueError):
        return {}


def golden_digests():
    """tests/golden/config_digests.json: the CPU oracle's digests of the full-size configurations (tests/golden/make_config_digests.py)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


DIGEST_FIELDS = ("len", "sum_counts", "min", "max", "sum_hc", "xor_hc", "sum_sq")


def table_digest(t):
    """The device table reduced the way the oracle's digests are (one scan each: kct_count_stats, kct_digest)."""
    lo, hi, _sq = t._count_stats()
    shc, xhc, ssq = t.digest()
    return {"len": len(t), "sum_counts": t.sum_counts, "min": lo, "max": hi, "sum_hc": shc, "xor_hc": xhc, "sum_sq": ssq}


def kernel_report(prof, kmers, b_alg, pmc_cfg):
    """Per-kernel ms, the algorithmic fraction over the summed kernel time and -- when PMC data of this build exists --
    measured HBM bytes, the measured fraction of peak and K1's VALU figures."""
    all_ms = sum(v[1] for v in prof.values())
    out = {"kernels_ms": {n: round(v[1], 4) for n, v in prof.items()}, "kernel_ms_total": round(all_ms, 4)}
    if all_ms:
        out["alg_frac"] = round(kmers * b_alg / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    traffic = None
    if pmc_cfg:
        per_launch = pmc_cfg.get("kernels", {})
        base = lambda n: n.replace("<shadow>", "")  # noqa: E731  (K2 on the shadow table is K2)
        got = [(per_launch[base(n)]["hbm_bytes_per_launch"] * v[0]) for n, v in prof.items() if "hbm_bytes_per_launch" in per_launch.get(base(n), {})]
        if got:
            traffic = sum(got)
            out["hbm_bytes_measured"] = traffic
            out["hbm_bytes_per_kmer"] = round(traffic / kmers, 3)
            if all_ms:
                out["measured_frac"] = round(traffic / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out, traffic


def compact(res):
    """The one-line record: the contract's fields, the roofline / cpu_baseline objects and one short entry per config."""
    def r(x, n=4):
        return round(x, n) if isinstance(x, float) else x
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                               "dtype", "data") if k in res}
    if "value_other_process" in res:
        out["value_other_process"] = {kk: r(vv, 3) for kk, vv in res["value_other_process"].items()}
    for k in ("repeats", "timed_seconds", "value_min", "value_max", "speedup_vs_cpu_baseline", "north_star_speedup_vs_cpu_baseline", "verified_vs_oracle"):
        if k in res:
            out[k] = r(res[k], 3)
    if "config" in res:
        c = res["config"]
        out["config"] = {"workload": f"C2: {c['reads_per_gpu']} x {c['read_len']} bp reads per GPU per step, k={c['k']}, genome {c['genome']} bp, steady state",
                         "distinct_kmers": c.get("distinct_kmers"), "world": c.get("world")}
        for k in ("pairs_received_per_rank", "merge_ms_rank0_median"):
            if k in c:
                out["config"][k] = r(c[k], 3)
    if "roofline" in res:
        f = res["roofline"]
        out["roofline"] = {k: r(f.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "measured_frac", "frac_of_wall", "kernel",
                                                     "kernels_total_ms_per_step", "dominant_kernel_ms_per_step", "alg_bytes_per_kmer", "pmc_source_sha")}
        out["roofline"]["kernels_ms_per_step"] = f.get("kernels_ms_per_step")
        if f.get("valu"):
            out["roofline"]["valu"] = {"kernel": f["valu"]["kernel"], "insts_per_kmer": r(f["valu"]["valu_insts_per_window"], 1),
                                       "busy_range": [r(x_, 3) for x_ in (f["valu"].get("valu_busy_range") or [None, None])], "clock_GHz": r(f["valu"].get("clock_GHz"), 3),
                                       "wave_wait_share": r(f["valu"]["wave_wait_share"], 3), "wave_issue_stall_share": r(f["valu"].get("wave_issue_stall_share"), 3)}
        if f.get("atomics"):
            out["roofline"]["atomics_direct_path"] = {"per_kmer": r(f["atomics"]["per_kmer"], 3), "per_second": r(f["atomics"]["per_second"], 0)}
    if "cpu_baseline" in res:
        b = res["cpu_baseline"]
        out["cpu_baseline"] = {"value": r(b["value"], 0), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                               "sample": f"first {b.get('sample_reads')} reads of the same stream, median of 3; best of key-space-sharded / read-sharded",
                               "value_1thread": r(b["value_1thread"], 0), "reads_sharded_tree_merge": b["reads_sharded_tree_merge"],
                               "keyspace_sharded": b["keyspace_sharded"], "host_cores": b["host_cores"]}
    ns = res.get("configs", {}).get("north_star_k21", {})
    if "kmers_per_s" in ns:   # the number the target sentence names (100 M x 150 bp, k=21, one GPU), at the top level
        out["north_star"] = {"kmers_per_s": r(ns["kmers_per_s"], 0), "seconds": r(ns["seconds"], 5), "frac": r(ns.get("alg_frac")),
                             "gate": "ok" if all(v for kk, v in ns.get("gate", {}).items() if kk != "sampled_keys") else "FAILED"}
        st_ = res["configs"].get("north_star_streamed", {})
        if "vs_one_call" in st_:
            out["north_star"]["streamed_vs_one_call"] = r(st_["vs_one_call"], 3)
    if "configs" in res:
        out["configs"] = {}
        for name, c in res["configs"].items():
            if "skipped" in c:
                out["configs"][name] = {"skipped": c["skipped"]}
                continue
            g = c.get("gate", {})
            e = {"kmers_per_s": r(c["kmers_per_s"], 0), "seconds": r(c["seconds"], 5),
                 "gate": "ok" if all(v for kk, v in g.items() if kk != "sampled_keys") else [kk for kk, v in g.items() if not v]}
            for k in ("alg_frac", "measured_frac", "hbm_bytes_per_kmer", "kmers_per_s_warm", "us_per_call", "us_per_call_loop_only", "vs_partitioned",
                      "K1_ms_per_step", "ascii_same_timing_K1_ms_per_step", "vs_ascii_same_timing"):
                if k in c:
                    e[k] = r(c[k], 0 if k == "kmers_per_s_warm" else 4)
            if "feeds" in c:
                e["vs_one_call"] = r(c["vs_one_call"], 3)
                e["feeds"] = {kk: {"kmers_per_s": r(vv["kmers_per_s"], 0), "counting_launches": vv["counting_launches"]} for kk, vv in c["feeds"].items()}
            for k in ("path_chosen", "world", "best_route", "note", "reads_total", "oracle_digests", "pieces", "table_slots", "inflater", "variants"):
                if k in c:
                    e[k] = c[k]
            if "zlib_fallback_kmers_per_s" in c:
                e["zlib_fallback_kmers_per_s"] = r(c["zlib_fallback_kmers_per_s"], 0)
            if "single_thread_inflate_kmers_per_s" in c:
                e["single_thread_inflate_kmers_per_s"] = r(c["single_thread_inflate_kmers_per_s"], 0)
            if "partitioned_path_kmers_per_s" in c:
                e["partitioned_path_kmers_per_s"] = r(c["partitioned_path_kmers_per_s"], 0)
            if "routes" in c:
                e["routes"] = {rn: {"kmers_per_s": r(rv["kmers_per_s"], 0), "seconds": r(rv["seconds"], 5), "mode": rv["mode"], "exchange": rv.get("exchange"),
                                    "rank0": {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in (rv["per_rank"][0] or {}).items()}}
                               for rn, rv in c["routes"].items()}
            if "kernels_ms" in c:
                e["kernels_ms"] = {k: round(v, 2 if v >= 1 else 3) for k, v in c["kernels_ms"].items() if v >= 0.02}
            out["configs"][name] = e
    out["detail"] = "full record: bench_detail.json (or --verbose); fields explained in bench.py's docstring"
    return out


def cpu_baseline(args, log):
    """Reference-shaped CPU path (oracle 'port') on bounded samples of the same stream; medians of three runs."""
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:  # noqa: BLE001 -- fall back to the portable build that travelled with the repo
        native = False
    L, k = args.read_len, args.k
    genome = oracle.synth_genome(args.genome, SEED_G)
    cores = len(os.sched_getaffinity(0))
    n1 = min(args.cpu_reads_1t, args.reads)
    nT = max(n1, args.cpu_reads_mt)
    readsT = oracle.synth_reads(genome, 0, nT, L, SEED_R)
    med = statistics.median

    def rate(fn):
        runs = []
        for _ in range(3):
            _t, km, s = fn()
            runs.append(km / s)
        return med(runs), runs

    r1, runs1 = rate(lambda: oracle.baseline_consume(readsT[:n1], L, k, 1, native))
    # reference-shaped multi-thread ("rayon-style"): reads sharded, private tables, parallel tree merge with add() semantics
    by_threads = {}
    for T in sorted({t for t in (8, 32, 128) if 1 < t <= cores}):
        by_threads[T], _ = rate(lambda T=T: oracle.baseline_consume(readsT, L, k, T, native))
    # best CPU: key space sharded over all cores, no merge
    sharded = {}
    for T in sorted({t for t in (32, 64, cores) if 1 < t <= cores} or {1}):
        sharded[T], _ = rate(lambda T=T: oracle.sharded_consume(readsT, L, k, T, 262144, native))
    best_T = max(sharded, key=sharded.get)
    best_rayon = max(by_threads.values()) if by_threads else r1
    value, cores_used, shape = sharded[best_T], best_T, "key space sharded over threads, nothing to merge"
    if best_rayon > value:
        value, cores_used, shape = best_rayon, max(by_threads, key=by_threads.get), "reads sharded over threads, tree merge"
    log(f"cpu: 1 thread {r1 / 1e6:.1f}, read-sharded {best_rayon / 1e6:.0f}, key-space-sharded {sharded[best_T] / 1e6:.0f} Mk-mers/s ({best_T} of {cores} threads)")
    tab1, _, _ = oracle.baseline_consume(readsT[:n1], L, k, min(cores, 16), native)   # the verification table of the n1-read sample
    out = {"value": value, "unit": "k-mers/s", "cores": cores_used, "kind": "port", "sample_reads": nT,
           "sample": f"first {nT} reads of the same stream ({nT * (L - k + 1)} k-mers), median of 3 runs; best of: {shape}; "
                     f"1 thread on the first {n1} reads: {r1:.4g} k-mers/s; the >= 10x target is quoted against `value`",
           "value_1thread": r1, "reads_sharded_tree_merge": {str(t): round(v) for t, v in by_threads.items()},
           "keyspace_sharded": {str(t): round(v) for t, v in sharded.items()}, "host_cores": cores, "native_build": native}
    return out, (readsT[:n1], tab1)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process group -- before
    anything in this process has touched the GPU -- and exit with its code.  Hang guard: a job that has not finished after
    --job-timeout seconds (a collective some rank never joined) has its whole process group killed and the parent exits 124."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, start_new_session=True)   # its own process group: the exact group we may have to kill

    def kill_group():
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                return
            try:
                child.wait(timeout=10)
                return
            except subprocess.TimeoutExpired:
                continue

    # The ranks live in ANOTHER session, so a signal sent to this process (or to its process group: Ctrl-C, an outer `timeout`) does
    # not reach them: SIGTERM / SIGINT / SIGHUP here are turned into an exception, and every way out of the wait -- those, the hang
    # guard, any other error -- kills the ranks' group before this process exits.  (The parent has not touched the GPU: killing and
    # exiting non-zero is all there is to do.)
    class _Signalled(BaseException):
        pass

    def on_signal(signum, _frame):
        raise _Signalled(signum)

    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)
    code = None
    try:
        code = child.wait(timeout=args.job_timeout)
    except subprocess.TimeoutExpired:
        print(f"[bench] the {args.gpus}-rank job did not finish in {args.job_timeout:.0f} s: killing its process group", file=sys.stderr, flush=True)
        code = 124
    except _Signalled as e:
        print(f"[bench] signal {e.args[0]}: killing the {args.gpus}-rank job's process group", file=sys.stderr, flush=True)
        code = 128 + int(e.args[0])
    finally:
        if child.poll() is None:
            for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                signal.signal(sg, signal.SIG_IGN)   # (a second signal must not interrupt the clean-up)
            kill_group()
    raise SystemExit(code)


def second_process_sample(args):
    """The headline's `value` once more from ANOTHER cold-started process on the same box (run before this process touches the GPU):
    variance across process starts, which the repeats inside one process cannot show."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--headline-sample", "--configs", "none", "--no-cpu-baseline", "--no-verify", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--reads", str(args.reads), "--read-len", str(args.read_len), "--k", str(args.k), "--genome", str(args.genome),
           "--path", args.path, "--detail-out", os.devnull]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "ms_per_step": d["ms_per_step"], "value_min": d.get("value_min"), "value_max": d.get("value_max")}
    except Exception as e:  # noqa: BLE001 -- a missing second sample must not cost the first
        return {"error": str(e)[:200]}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    other = None
    if args.gpus == 1 and "RANK" not in os.environ and not (args.no_second_process or args.headline_sample or args.no_headline):
        other = second_process_sample(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    from oxli_amd import KmerCountTable, _lib
    from oxli_amd.distributed import global_scalar_sum, merge_across_ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    # who moves data between the ranks: libkct_rccl.so's own communicator (bootstrapped over the torch group) and / or torch.distributed
    native = None
    if world > 1 and args.exchange != "torch":
        can = args.backend == "nccl" and world <= torch.cuda.device_count()
        if can:
            # (never run on more than one GPU before the driver's own 8-GPU run: a failure to come up on ANY rank -- agreed with one
            # all-reduce -- sends every rank back to torch.distributed instead of ending the job)
            try:
                from oxli_amd.distributed import NativeRccl
                native = NativeRccl()
                ok_native = 1
            except Exception as e:  # noqa: BLE001
                log(f"libkct_rccl.so did not come up on rank {rank}: {e}")
                native, ok_native = None, 0
            flag = torch.tensor([ok_native], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                if native is not None:
                    native.close()
                native = None
                if args.exchange == "native":
                    raise SystemExit("--exchange native: libkct_rccl.so's communicator did not come up on every rank")
        elif args.exchange == "native":
            raise SystemExit("--exchange native needs --backend nccl and one rank per GPU (RCCL refuses two ranks on one device)")
    exchanges = ([("native", native)] if native is not None else []) + ([("torch", None)] if native is None or args.exchange in ("both", "auto") else [])

    lib = _lib.load()
    L, k, R, G = args.read_len, args.k, args.reads, args.genome
    kmers_per_step = R * (L - k + 1)
    b_alg = L / (L - k + 1) + 24.0
    stream = torch.cuda.current_stream().cuda_stream
    ablate = bool(os.environ.get("KCT_ABLATE"))  # timing experiments that deliberately skip work: no result checks
    pmc = pmc_summary()
    golden = golden_digests()
    med = statistics.median
    want = ALL_CONFIGS if args.configs == "all" else [] if args.configs == "none" else [c for c in args.configs.split(",") if c]
    if world > 1 and args.configs == "all":
        # C5 over N ranks wants a 2^33-slot (128 GiB) private table, a 48 GB pair export and a 43 GB receive buffer per rank on its late
        # route: it runs when asked for by name (--configs C4,C5); a rank that ran out of HBM would leave the others in a collective
        want = [c for c in want if c != "C5"]
    for c in want:
        if c not in ALL_CONFIGS:
            raise SystemExit(f"unknown config {c!r}; choose from {ALL_CONFIGS}")
    checker = rank == 0 and world == 1 and not args.no_cpu_baseline   # the CPU oracle as the checker of the gates
    if checker:
        import oracle

    def synth(G_, R_, L_, first=0):
        g = torch.empty(G_, dtype=torch.uint8, device="cuda")
        r = torch.empty(R_ * (L_ + 1), dtype=torch.uint8, device="cuda")
        assert lib.kct_synth_genome_device(g.data_ptr(), G_, SEED_G, stream) == 0
        assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G_, first, R_, L_, SEED_R, stream) == 0
        torch.cuda.synchronize()
        return g, r

    result = {
        "metric": "k-mers/sec (consume) at k=%d, %d bp reads" % (k, L), "value": None, "unit": "k-mers/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic" if not ablate else "INVALID (KCT_ABLATE set: work skipped)",
    }

    # ------------------------------------------------------------------------------------------ headline: C2
    genome = torch.empty(G, dtype=torch.uint8, device="cuda")
    assert lib.kct_synth_genome_device(genome.data_ptr(), G, SEED_G, stream) == 0
    nb = max(1, min(args.steps, 64))
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

    def job():
        """clear, K steps, conversion (+ merge): (seconds [max over ranks], k-mers of this rank, merge ms, pairs received)"""
        table.clear()
        if world > 1:
            table.resize(G)  # (the merge left an owner-sized table: back to a rank's private capacity, outside the timed region)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        n = 0
        for s in range(args.steps):
            n += step(s)
        t_merge = time.perf_counter()
        recv = merge_across_ranks(table, native=native) if world > 1 else 0
        table.sync()  # counts still pending in the dedupe-first path's shadow table are converted inside the timed region
        torch.cuda.synchronize()
        merge_ms = (time.perf_counter() - t_merge) * 1e3
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, n, merge_ms, recv

    if not args.no_headline:
        for s in range(args.warmup):
            step(s)
        if world > 1 and native is not None:
            # libkct_rccl.so's merge has never run between real GPUs before the driver's own N-GPU run: a rank on which its FIRST call
            # returns an error (an exception here, not a hang) sends every rank back to torch.distributed for the rest of the job
            try:
                merge_across_ranks(table, native=native)
                ok_merge = 1
            except Exception as e:  # noqa: BLE001
                log(f"kct_rccl_merge_across_ranks failed on rank {rank}: {e}")
                ok_merge = 0
            flag = torch.tensor([ok_merge], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                if args.exchange == "native":
                    raise SystemExit("--exchange native: kct_rccl_merge_across_ranks failed")
                native.close()
                native = None
                exchanges = [("torch", None)]
                table.clear()
                table.resize(G)
                for s in range(max(1, args.warmup)):
                    step(s)
        if world > 1 and args.warmup:
            merge_across_ranks(table, native=native)  # warm the collective and the merge kernels too
        first = job()  # (also teaches a fresh table that the dedupe-first path pays: the steady state)
        repeats = int(min(args.max_repeats, max(5 if world == 1 else 3, math.ceil(1.15 * args.min_seconds / max(first[0], 1e-6)))))
        if world > 1:  # every rank must run the same number of jobs
            t = torch.tensor([repeats], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
        runs = [job() for _ in range(repeats)]
        n_total = runs[-1][1]
        assert ablate or all(r[1] == kmers_per_step * args.steps for r in runs), (n_total, kmers_per_step * args.steps)
        n_all = global_scalar_sum(n_total, "cuda") if world > 1 else n_total
        times = sorted(r[0] for r in runs)
        elapsed = med(times)
        value = n_all / elapsed
        # invariants of the finished job (cheap, outside the timed region)
        distinct = global_scalar_sum(len(table), "cuda") if world > 1 else len(table)
        total_counts = global_scalar_sum(table.sum_counts, "cuda") if world > 1 else table.sum_counts
        assert ablate or total_counts == world * kmers_per_step * args.steps, (total_counts, world * kmers_per_step * args.steps)
        # Per-kernel device times: the SAME job once more with the library's HIP-event timing switched on (an event pair
        # around every launch costs ~6 % at this step size, so it stays out of the jobs `value` is taken from).
        table.clear()
        table.profile(True)
        table.profile_reset()
        for s in range(args.steps):
            step(s)
        table.sync()
        torch.cuda.synchronize()
        prof = table.profile_read()
        table.profile(False)
        rep, traffic_job = kernel_report(prof, kmers_per_step * args.steps, b_alg, pmc.get("C2"))
        all_ms = rep["kernel_ms_total"]
        dom = max(prof, key=lambda n_: prof[n_][1]) if prof else "none"
        launches, ms = prof.get(dom, (0, 0.0))
        pipe_gbs = kmers_per_step * args.steps * b_alg / (all_ms * 1e-3) / 1e9 if all_ms else float("nan")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": pipe_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pipe_gbs / HBM_PEAK_GBS,
                    "traffic": traffic_job / args.steps if traffic_job else None,
                    "measured_frac": (traffic_job / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_job and all_ms else None,
                    "basis": "achieved = algorithmic bytes per step (25.15 B/k-mer x k-mers) / summed device time of every kernel of the step (HIP "
                             "events, instrumented repetition); traffic = PMC HBM bytes per step and measured_frac = traffic / kernel time / peak, "
                             "from profiles/pmc_r06.json when it matches this source tree (else null)",
                    "alg_bytes_per_kmer": b_alg, "kmers_per_step": kmers_per_step, "kernels_total_ms_per_step": all_ms / args.steps,
                    "dominant_kernel_ms_per_step": ms / args.steps,
                    "frac_of_wall": kmers_per_step * b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "dominant_kernel": {"name": dom, "avg_launch_ms": ms / launches if launches else None, "launches": launches},
                    "kernels_ms_per_step": {n_: round(v[1] / args.steps, 4) for n_, v in prof.items()},
                    "valu": pmc.get("C2", {}).get("valu"), "atomics": pmc.get("C2", {}).get("atomics"), "pmc_source_sha": pmc.get("source_sha")}
        result.update({
            "value": value, "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": repeats, "timed_seconds": sum(times), "value_min": n_all / times[-1], "value_max": n_all / times[0],
            "config": {"workload": f"C2: {R} x {L} bp synthetic reads per GPU per step, k={k}, genome {G} bp (seed {SEED_G}/{SEED_R}), "
                                   f"device hash table in HBM ({table.capacity} slots x 16 B); steady state (table cleared, dedupe hint kept)",
                       "reads_per_gpu": R, "read_len": L, "k": k, "genome": G, "distinct_kmers": distinct, "distinct_batches_per_gpu": nb,
                       "job": f"{args.steps} steps into an empty table + conversion of pending counts" +
                              (f" + one RCCL owner all-to-all merge ({med([r[2] for r in runs]):.3f} ms on rank 0, {runs[-1][3]} pairs received)" if world > 1 else ""),
                       "world": world, "backend": args.backend if world > 1 else None,
                       "exchange": None if world == 1 else "libkct_rccl.so (kct_rccl_merge_across_ranks)" if native is not None else "torch.distributed (all_to_all_single)"},
            "roofline": roofline,
        })
        if other is not None:
            result["value_other_process"] = other   # the same job from another cold-started process on this box
        if world > 1:
            recv_all = [None] * world
            dist.all_gather_object(recv_all, int(runs[-1][3]))
            result["config"]["pairs_received_per_rank"] = recv_all
            result["config"]["merge_ms_rank0_median"] = med([r[2] for r in runs])

    # ------------------------------------------------------------------------------------------ the other configs
    configs = {}

    def timed_call(t, fn, cold, path=None):
        """One call on an existing table: cleared, and (cold) made to forget what it learnt about its input."""
        t.clear()
        if cold:
            t.set_path(path or args.path)  # (resets the dedupe hint)
        t.profile(True)
        t.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = fn()
        t.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = t.profile_read()
        t.profile(False)
        return dt, n, prof

    def table_stats(t):
        lo, hi, sq = t._count_stats()
        return [len(t), t.sum_counts, lo, hi, sq]

    if rank == 0 and world == 1 and want:
        del batches[1:]
        torch.cuda.empty_cache()
        reads0 = batches[0]
        if "cold_C2" in want:
            runs = [timed_call(table, lambda: table.consume_device(reads0.data_ptr(), reads0.numel(), R * L), True) for _ in range(9)]
            dt, n, prof = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
            rep, _ = kernel_report(prof, kmers_per_step, b_alg, pmc.get("cold_C2"))
            ok = all(r[1] == kmers_per_step for r in runs) and table.sum_counts == kmers_per_step
            configs["cold_C2"] = {"kmers_per_s": kmers_per_step / dt, "seconds": dt, "runs": len(runs), "gate": {"n_and_sum_counts": bool(ok)}, **rep}
            assert ablate or ok
        if "packed_C2" in want:
            # the same batch as PACKED base arrays resident in HBM (2 bits + 1 validity bit per base): steady-state steps
            ng = (reads0.numel() + 15) // 16
            pc = torch.empty(ng, dtype=torch.int32, device="cuda")
            pv = torch.empty(ng, dtype=torch.int16, device="cuda")
            assert lib.kct_pack_stream_device(reads0.data_ptr(), reads0.numel(), pc.data_ptr(), pv.data_ptr(), stream) == 0
            torch.cuda.synchronize()

            def packed_job():
                n_ = 0
                for _s in range(args.steps):
                    n_ += table.consume_device_packed(pc.data_ptr(), pv.data_ptr(), reads0.numel(), R * L)
                return n_
            def ascii_job():   # the SAME batch as ASCII bytes under the SAME timing (events on, one batch repeated): the like-for-like comparison
                n_ = 0
                for _s in range(args.steps):
                    n_ += table.consume_device(reads0.data_ptr(), reads0.numel(), R * L)
                return n_
            timed_call(table, packed_job, False)
            runs = [timed_call(table, packed_job, False) for _ in range(9)]
            dt, n, prof = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
            ok = all(r[1] == kmers_per_step * args.steps for r in runs) and table.sum_counts == kmers_per_step * args.steps
            rep, _ = kernel_report(prof, kmers_per_step * args.steps, b_alg, None)
            timed_call(table, ascii_job, False)
            aruns = [timed_call(table, ascii_job, False) for _ in range(9)]
            adt, _an, aprof = sorted(aruns, key=lambda r: r[0])[len(aruns) // 2]
            k1_of = lambda pr: sum(v[1] for kn, v in pr.items() if kn.startswith("partition_windows_kernel")) / args.steps  # noqa: E731
            configs["packed_C2"] = {"kmers_per_s": kmers_per_step * args.steps / dt, "seconds": dt, "runs": len(runs), "steps": args.steps,
                                    "what": "the headline's steps with the batch resident as packed base arrays (0.375 B per base), event timing on; "
                                            "ascii_same_timing = the same batch as ASCII bytes through the same loop under the same timing",
                                    "K1_ms_per_step": round(k1_of(prof), 4), "ascii_same_timing_kmers_per_s": kmers_per_step * args.steps / adt,
                                    "ascii_same_timing_K1_ms_per_step": round(k1_of(aprof), 4), "vs_ascii_same_timing": adt / dt,
                                    "gate": {"n_and_sum_counts": bool(ok)}, **rep}
            assert ablate or ok
            del pc, pv
        if "k51_deep" in want:
            # k > 32 with deep coverage of a small genome (1 M x 150 bp, k=51, genome 2 Mbp; steady-state steps as the headline's): what the
            # table chooses (hashing every window: the 128-bit dedupe-first variant left the automatic choice in round 4), hashing forced,
            # and the 128-bit variant forced (set_path("dedupe"))
            Gk, kk = 2_000_000, 51
            gk = torch.empty(Gk, dtype=torch.uint8, device="cuda")
            rk = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
            assert lib.kct_synth_genome_device(gk.data_ptr(), Gk, SEED_G, stream) == 0
            assert lib.kct_synth_reads_device(rk.data_ptr(), gk.data_ptr(), Gk, 0, R, L, SEED_R, stream) == 0
            torch.cuda.synchronize()
            nk_step, res51 = R * (L - kk + 1), {}
            for path in ("auto", "partitioned", "dedupe"):
                t51 = KmerCountTable(kk, capacity=Gk)
                t51.set_path(path)

                def job51():
                    n_ = 0
                    for _s in range(args.steps):
                        n_ += t51.consume_device(rk.data_ptr(), rk.numel(), R * L)
                    return n_
                timed_call(t51, job51, False)
                runs = [timed_call(t51, job51, False) for _ in range(5)]
                dt, n, prof = sorted(runs, key=lambda r_: r_[0])[len(runs) // 2]
                res51[path] = (nk_step * args.steps / dt, dt, prof, (n, len(t51), t51.sum_counts) + t51.digest())
                del t51
            rep, _ = kernel_report(res51["auto"][2], nk_step * args.steps, L / (L - kk + 1) + 24.0, None)
            ok = res51["auto"][3] == res51["partitioned"][3] == res51["dedupe"][3] and res51["auto"][3][0] == nk_step * args.steps
            configs["k51_deep"] = {"kmers_per_s": res51["auto"][0], "seconds": res51["auto"][1], "steps": args.steps,
                                   "partitioned_path_kmers_per_s": res51["partitioned"][0], "vs_partitioned": res51["auto"][0] / res51["partitioned"][0],
                                   "forced_128bit_dedupe_kmers_per_s": res51["dedupe"][0], "forced_128bit_vs_partitioned": res51["dedupe"][0] / res51["partitioned"][0],
                                   "path_chosen": "128-bit dedupe-first" if any("raw128" in kn for kn in res51["auto"][2]) else "hash every window",
                                   "what": "1 M x 150 bp per step, k=51, genome 2 Mbp, steady state (event timing on)",
                                   "gate": {"auto_partitioned_and_forced_128bit_paths_agree": bool(ok)}, **rep}
            assert ablate or ok
            del gk, rk
        host = None
        if "e2e_C2" in want or "per_record" in want or any(f_ in want for f_ in FILES):
            host = reads0.cpu().numpy().reshape(R, L + 1)
        if "e2e_C2" in want:
            # Two placements of the caller's 151 MB batch: "one_thread" = the array is first touched by ONE thread (numpy's copy: every page on
            # that thread's NUMA node -- what a single-threaded caller hands over); "spread" = first touched by 16 threads pinned to CPUs spread
            # over the host (pages on every node, as a multi-threaded reader leaves them).  The library's 16 packer threads read that memory once;
            # the rest of the call is the H2D of 57 MB of packed bases and ~0.8 ms of kernels.
            import concurrent.futures as cf
            offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
            src2d = host[:, :L]

            def first_touch_spread():
                out_ = np.empty(R * L, dtype=np.uint8)
                cpus = sorted(os.sched_getaffinity(0))
                nth = min(16, len(cpus))
                rows = [(i * R // nth, (i + 1) * R // nth) for i in range(nth)]
                def fill(i):
                    try:
                        os.sched_setaffinity(0, {cpus[i * len(cpus) // nth]})   # (this thread only)
                    except OSError:
                        pass
                    a_, b_ = rows[i]
                    out_[a_ * L:b_ * L] = src2d[a_:b_].reshape(-1)
                with cf.ThreadPoolExecutor(nth) as ex:
                    list(ex.map(fill, range(nth)))
                return out_
            variants = {}
            for vname, make in (("one_thread", lambda: np.ascontiguousarray(src2d).reshape(-1)), ("spread", first_touch_spread)):
                flat = make()
                stage, tls = [], []

                def batch_job():
                    t0_ = time.perf_counter()
                    n_ = table.consume_batch((flat, offsets))
                    stage.append(time.perf_counter() - t0_)     # the call itself: pack + H2D + passes submitted (and mostly run)
                    tls.append(table.batch_timeline())          # where inside the call (kct_batch_timeline)
                    return n_
                timed_call(table, batch_job, False)
                stage.clear(); tls.clear()
                runs = [timed_call(table, batch_job, False) for _ in range(7)]
                order = sorted(range(len(runs)), key=lambda i: runs[i][0])
                mid = order[len(order) // 2]
                dt, n, prof = runs[mid]
                rep, _ = kernel_report(prof, kmers_per_step, b_alg, None)
                ok = all(r[1] == kmers_per_step for r in runs) and table.sum_counts == kmers_per_step
                variants[vname] = {"kmers_per_s": kmers_per_step / dt, "seconds": dt, "seconds_min_max": [runs[order[0]][0], runs[order[-1]][0]],
                                   "call_ms": stage[mid] * 1e3, "sync_after_ms": (dt - stage[mid]) * 1e3, "kernels_ms_total": rep["kernel_ms_total"],
                                   "host_side_ms": stage[mid] * 1e3 - rep["kernel_ms_total"],
                                   "source_GB_per_s_over_the_call": R * L / stage[mid] / 1e9, "ok": bool(ok), "rep": rep,
                                   "timeline": {kk: round(vv, 3) for kk, vv in tls[mid].items()},
                                   "packers_GB_per_s": round(tls[mid]["source_bytes"] / max(1e-9, (tls[mid]["last_packer_end_ms"] - tls[mid]["first_packer_start_ms"]) * 1e-3) / 1e9, 1),
                                   "packer_GB_per_s_per_thread": round(tls[mid]["source_bytes"] / max(1e-9, tls[mid]["threads_busy_ms_sum"] * 1e-3) / 1e9, 2)}
                del flat
            best = max(variants, key=lambda v_: variants[v_]["kmers_per_s"])
            v1 = variants["one_thread"]
            configs["e2e_C2"] = {"kmers_per_s": v1["kmers_per_s"], "seconds": v1["seconds"], "runs": 7,
                                 "what": "kct_consume_batch from pageable host memory: 32 pool threads SIMD-pack the records to 0.375 B/base into pinned staging, "
                                         "H2D, count + conversion; `kmers_per_s` = the source array first-touched by one thread (as before); variants: see bench.py",
                                 "variants": {vn: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v_.items() if kk not in ("rep", "ok")} for vn, v_ in variants.items()},
                                 "best_variant": best, "kmers_per_s_best_variant": variants[best]["kmers_per_s"],
                                 "timeline": "call_ms = pack + H2D + kernels submitted; host_side_ms = call_ms - device kernel time: what the packers and PCIe cost; "
                                             "source_GB_per_s_over_the_call = 150 MB / call_ms; variants.*.timeline = kct_batch_timeline of the median call (ms since the "
                                             "call began: argument checks, parts cut, first packer start, last packer end, last H2D enqueued, passes submitted; threads, their "
                                             "busy time, bytes, minor faults, CPUs / NUMA nodes the packers ran on)",
                                 "gate": {"n_and_sum_counts": bool(all(v_["ok"] for v_ in variants.values()))}, **v1["rep"]}
            assert ablate or configs["e2e_C2"]["gate"]["n_and_sum_counts"]
        if any(f_ in want for f_ in FILES):
            import gzip
            import shutil
            import struct
            import tempfile
            import zlib
            tmpd = tempfile.mkdtemp(prefix="kct_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                fa = os.path.join(tmpd, "c2.fa")
                with open(fa, "wb") as f:      # ">r<i>\n<150 bases>\n": 160 MB
                    hdr = np.char.add(np.char.add(">r", np.arange(R).astype(str)), "\n").astype("S")
                    for i0 in range(0, R, 50_000):
                        f.write(b"".join(h_ + host[i, :L].tobytes() + b"\n" for i, h_ in zip(range(i0, min(R, i0 + 50_000)), hdr[i0:i0 + 50_000])))
                text = open(fa, "rb").read()
                paths = {"file_fasta": fa}
                if "file_gz" in want:
                    paths["file_gz"] = fa + ".gz"
                    with gzip.open(paths["file_gz"], "wb", compresslevel=1) as dst:
                        dst.write(text)
                if "file_bgzf" in want:
                    paths["file_bgzf"] = fa + ".bgz.gz"
                    with open(paths["file_bgzf"], "wb") as dst:   # SAM specification 4.1: members of <= 64 KiB of text with a 'BC' size field, then an empty one
                        for o in list(range(0, len(text), 65280)) + [len(text)]:
                            piece = text[o:o + 65280]
                            comp = zlib.compressobj(1, zlib.DEFLATED, -15)
                            body = comp.compress(piece) + comp.flush()
                            dst.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1) + body +
                                      struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
                del text
                gd = golden.get("C2", {})
                same = bool(gd) and (gd["reads"], gd["read_len"], gd["k"], gd["genome"]) == (R, L, k, G) and world == 1
                lib.kct_inflater_name.restype = C.c_char_p
                for name in [f_ for f_ in FILES if f_ in want]:
                    def file_job(path=paths[name]):
                        return table.consume_file(path)
                    def measure(reps):
                        runs_ = [timed_call(table, file_job, False) for _ in range(reps)]
                        dt_, n_, prof_ = sorted(runs_, key=lambda r_: r_[0])[len(runs_) // 2]
                        mine_ = table_digest(table)
                        ok_ = all(r_[1] == kmers_per_step for r_ in runs_) and table.consumed == R * L and \
                            (all(mine_[f_] == gd[f_] for f_ in DIGEST_FIELDS) if same else mine_["sum_counts"] == kmers_per_step)
                        return dt_, prof_, bool(ok_), len(runs_)
                    measure(1)   # (chunk buffers, page cache)
                    dt, prof, ok, nruns = measure(3 if name == "file_gz" else 5)
                    rep, _ = kernel_report(prof, kmers_per_step, b_alg, None)
                    entry = {"kmers_per_s": kmers_per_step / dt, "seconds": dt, "runs": nruns, "file_bytes": os.path.getsize(paths[name]),
                             "text_GB_per_s": round(os.path.getsize(fa) / dt / 1e9, 3),
                             "what": "kct_consume_file(" + name[5:] + "): the C2 batch as a file in /dev/shm -> host parser threads -> pinned chunks -> H2D -> count + conversion",
                             "gate": {"n_consumed_and_oracle_digests" if same else "n_and_sum_counts": ok}, **rep}
                    if name != "file_fasta":
                        entry["inflater"] = lib.kct_inflater_name().decode()
                        if entry["inflater"] == "libdeflate":   # the same file through the zlib fallback (what a box without libdeflate.so.0 gets)
                            os.environ["KCT_NO_LIBDEFLATE"] = "1"
                            dtz, _pz, okz, _nz = measure(2)
                            del os.environ["KCT_NO_LIBDEFLATE"]
                            entry["zlib_fallback_kmers_per_s"] = kmers_per_step / dtz
                            entry["gate"]["zlib_fallback_same_table"] = okz
                        if name == "file_gz":
                            # round 6: the ONE member is inflated by several threads (csrc/parallel_inflate.h); beside it, the same file with
                            # that switched off -- one thread's libdeflate, the bound of this entry until round 5
                            os.environ["KCT_NO_PARALLEL_GZIP"] = "1"
                            dt1, _p1, ok1, _n1 = measure(2)
                            del os.environ["KCT_NO_PARALLEL_GZIP"]
                            entry["single_thread_inflate_kmers_per_s"] = kmers_per_step / dt1
                            entry["gate"]["single_thread_inflate_same_table"] = ok1
                            entry["inflater"] += " / parallel_inflate.h (several threads on the one member, verified by length and CRC-32)"
                            entry["note"] = ("one deflate stream was one thread's work until round 5 (0.4 GB/s of text with zlib, ~0.8 with libdeflate): "
                                             "`single_thread_inflate_kmers_per_s`; `kmers_per_s` = the member entered at block boundaries by up to 64 threads")
                    configs[name] = entry
                    log(f"{name}: {entry['kmers_per_s']:.3g} k-mers/s ({entry.get('inflater', 'plain text')}), gate {entry['gate']}")
                    assert ablate or all(entry["gate"].values()), (name, entry["gate"])
            finally:
                shutil.rmtree(tmpd, ignore_errors=True)
        if "per_record" in want:
            recs = [host[i, :L].tobytes() for i in range(R)]
            runs = []
            for _ in range(3):
                t2 = KmerCountTable(k)          # default-constructed, as the README's loop would
                t0 = time.perf_counter()
                n = 0
                consume = t2.consume
                for rec in recs:
                    n += consume(rec)
                t_loop = time.perf_counter() - t0
                total = t2.sum_counts           # (the first read of the table counts what the loop buffered)
                dt = time.perf_counter() - t0
                runs.append((dt, t_loop, n == kmers_per_step == total))
            dt, t_loop, _ = sorted(runs)[1]
            n = kmers_per_step
            ok = all(r[2] for r in runs)
            configs["per_record"] = {"kmers_per_s": n / dt, "seconds": dt, "calls": R, "us_per_call": dt / R * 1e6, "us_per_call_loop_only": t_loop / R * 1e6,
                                     "runs": len(runs),
                                     "what": "for rec in reads: table.consume(rec) on KmerCountTable(21), Python loop and the final read (the device pass of "
                                             "a table that starts tiny and finds its size) included; median of 3",
                                     "gate": {"n_and_sum_counts": bool(ok)}}
            assert ablate or ok
            del t2, recs
        del host
    del batches, genome
    if not args.no_headline:
        sample_table = table
    torch.cuda.empty_cache()

    if rank == 0 and world == 1:
        for name in [c for c in want if c in BIG]:
            Rb, Lb, kb, Gb = BIG[name]
            free, _tot = torch.cuda.mem_get_info()
            if free < 230 * (1 << 30):
                configs[name] = {"skipped": f"needs a whole MI355X: {free >> 30} GiB free"}
                continue
            g, r = synth(Gb, Rb, Lb)
            del g
            n_exp = Rb * (Lb - kb + 1)
            balg = Lb / (Lb - kb + 1) + 24.0
            t = KmerCountTable(kb, capacity=Gb)
            t.set_path(args.path)
            call = lambda: t.consume_device(r.data_ptr(), r.numel(), Rb * Lb)  # noqa: E731
            t0 = time.perf_counter()
            n_first = call()
            t.sync()
            first_s = time.perf_counter() - t0   # includes every allocation
            cold = [timed_call(t, call, True) for _ in range(3)]
            dt, n, prof = sorted(cold, key=lambda x: x[0])[1]
            st = table_stats(t)
            warm = timed_call(t, call, False)
            rep, _ = kernel_report(prof, n_exp, balg, pmc.get(name))
            gate = {"n": bool(n == n_exp == n_first and all(c_[1] == n_exp for c_ in cold) and warm[1] == n_exp), "sum_counts": bool(st[1] == n_exp)}
            entry = {"kmers": n_exp, "kmers_per_s": n_exp / dt, "seconds": dt, "seconds_min_max": [min(c_[0] for c_ in cold), max(c_[0] for c_ in cold)],
                     "kmers_per_s_warm": n_exp / warm[0], "seconds_first_call": first_s, "table_slots": t.capacity, "distinct": st[0], **rep}
            if not args.no_verify:
                # sampled keys: k-mer hashes of the first reads through the device's own hash kernel, plus keys that are absent
                ns = 2000 if Lb <= 1000 else 30
                sub = r[: ns * (Lb + 1)].cpu().numpy().reshape(ns, Lb + 1)
                hs = np.unique(np.concatenate([t.hash_windows(sub[i, :Lb].tobytes()) for i in range(ns)]))
                hs = hs[hs != 0]
                sample = np.concatenate([hs, hs ^ np.uint64(0x5555555555555555)])
                got = np.array(t.get_hash_array(sample), dtype=np.uint64)
                if checker:  # an oracle slice: the device's hash kernel finds exactly the slice's keys
                    ref = oracle.OracleTable(kb)
                    for i in range(ns):
                        ref.consume(sub[i, :Lb])
                    rk, _rc = ref.dump_arrays()
                    gate["oracle_slice_keys"] = bool(np.array_equal(rk, hs))
                gd = golden.get(name)
                if gd and (gd["reads"], gd["read_len"], gd["k"], gd["genome"]) == (Rb, Lb, kb, Gb):
                    # THE ORACLE'S TABLE of the whole input (tests/golden/config_digests.json): len, sum_counts, min, max and the three
                    # order-free digests EQUAL, n and consumed equal
                    mine = table_digest(t)
                    gate["equals_oracle_digests"] = bool(all(mine[f_] == gd[f_] for f_ in DIGEST_FIELDS) and n == gd["n"] and t.consumed == gd["consumed"])
                else:
                    gate["equals_oracle_digests"] = False
                    entry["note"] = "no oracle digest of this size in tests/golden/config_digests.json"
                t.release_scratch()
                del t
                torch.cuda.empty_cache()
                d = KmerCountTable(kb, capacity=Gb)
                d.set_path("direct")
                t0 = time.perf_counter()
                nd = d.consume_device(r.data_ptr(), r.numel(), Rb * Lb)
                entry["direct_path_kmers_per_s"] = n_exp / (time.perf_counter() - t0)
                sd = table_stats(d)
                gate["equals_direct_path"] = bool(nd == n_exp and sd[:4] == st[:4] and math.isclose(sd[4], st[4], rel_tol=1e-12) and
                                                  np.array_equal(np.array(d.get_hash_array(sample), dtype=np.uint64), got))
                gate["sampled_keys"] = int(sample.size)
                del d
            else:
                del t
            entry["gate"] = gate
            configs[name] = entry
            log(f"{name}: {entry['kmers_per_s']:.3g} k-mers/s, gate {'ok' if all(v for kk, v in gate.items() if kk != 'sampled_keys') else gate}")
            assert ablate or all(v for kk, v in gate.items() if kk != "sampled_keys"), (name, gate)
            del r
            torch.cuda.empty_cache()
    # ------------------------------------------------------------------------------------------ the north-star run, fed in pieces
    if rank == 0 and world == 1 and "north_star_streamed" in want:
        Rb, Lb, kb, Gb = BIG["north_star_k21"]
        free, _tot = torch.cuda.mem_get_info()
        if free < 230 * (1 << 30):
            configs["north_star_streamed"] = {"skipped": f"needs a whole MI355X: {free >> 30} GiB free"}
        else:
            g, r = synth(Gb, Rb, Lb)
            del g
            n_exp = Rb * (Lb - kb + 1)
            gd = golden.get("north_star_k21", {})
            entry = {"kmers": n_exp, "what": "the north-star run's 100 M reads as consecutive kct_consume_device calls into ONE table that starts empty and "
                                             "hint-free; one conversion at the end (sync) inside the timed region; median of 3", "feeds": {}}
            one_call = configs.get("north_star_k21", {}).get("kmers_per_s")
            gate = {}
            for calls in (1, 20, 100):
                per = Rb // calls
                nbytes = per * (Lb + 1)
                assert per * calls == Rb and nbytes % 16 == 0
                t = KmerCountTable(kb, capacity=Gb)

                def feed():
                    n_ = 0
                    for i in range(calls):
                        n_ += t.consume_device(r.data_ptr() + i * nbytes, nbytes, per * Lb)
                    return n_
                feed(); t.sync()                               # allocations
                runs = [timed_call(t, feed, True, "auto") for _ in range(3)]
                dt, n, prof = sorted(runs, key=lambda x: x[0])[1]
                mine = table_digest(t)
                ok = bool(gd) and all(mine[f_] == gd[f_] for f_ in DIGEST_FIELDS) and all(x[1] == n_exp for x in runs) and t.consumed == gd.get("consumed")
                gate[f"{calls}_calls_equal_oracle_digests"] = bool(ok)
                k1 = {kn: v[0] for kn, v in prof.items() if kn.startswith(("partition_windows_kernel", "count_windows_kernel"))}
                entry["feeds"][str(calls)] = {"kmers_per_s": n_exp / dt, "seconds": dt, "seconds_min_max": [runs[0][0], runs[-1][0]] if False else [min(x[0] for x in runs), max(x[0] for x in runs)],
                                              "counting_launches": k1, "kernel_ms_total": round(sum(v[1] for v in prof.values()), 3)}
                t.release_scratch()
                del t
                torch.cuda.empty_cache()
            base = entry["feeds"]["1"]["kmers_per_s"]
            worst = min(entry["feeds"]["20"]["kmers_per_s"], entry["feeds"]["100"]["kmers_per_s"])
            entry.update({"kmers_per_s": worst, "seconds": n_exp / worst, "vs_one_call": worst / base, "one_call_kmers_per_s": base,
                          "one_call_kmers_per_s_in_north_star_k21": one_call, "gate": gate})
            configs["north_star_streamed"] = entry
            log(f"north_star_streamed: 1 call {base:.3g}, 20 calls {entry['feeds']['20']['kmers_per_s']:.3g}, 100 calls {entry['feeds']['100']['kmers_per_s']:.3g} k-mers/s, gate {gate}")
            assert ablate or all(gate.values()), gate
            del r
            torch.cuda.empty_cache()
    # ------------------------------------------------------------------------------------------ C5 whole, one GPU, fed in pieces
    if rank == 0 and world == 1 and "C5_whole" in want:
        Rb, Lb, kb, Gb, pieces = C5_WHOLE
        free, _tot = torch.cuda.mem_get_info()
        if free < 250 * (1 << 30):
            configs["C5_whole"] = {"skipped": f"needs a whole MI355X: {free >> 30} GiB free"}
        else:
            per = (Rb + pieces - 1) // pieces        # (the last piece is shorter)
            n_exp = Rb * (Lb - kb + 1)
            balg = Lb / (Lb - kb + 1) + 24.0
            g = torch.empty(Gb, dtype=torch.uint8, device="cuda")
            assert lib.kct_synth_genome_device(g.data_ptr(), Gb, SEED_G, stream) == 0
            r = torch.empty(per * (Lb + 1), dtype=torch.uint8, device="cuda")
            t = KmerCountTable(kb, capacity=Gb)
            t.set_path(args.path)

            def whole(profile):
                t.clear()
                t.set_path(args.path)
                t.profile(profile)
                t.profile_reset()
                n_, secs = 0, 0.0
                for p_ in range(pieces):
                    cnt_ = min(per, Rb - p_ * per)
                    assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), Gb, p_ * per, cnt_, Lb, SEED_R, stream) == 0
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n_ += t.consume_device(r.data_ptr(), cnt_ * (Lb + 1), cnt_ * Lb)
                    if p_ == pieces - 1:
                        t.sync()
                    torch.cuda.synchronize()
                    secs += time.perf_counter() - t0
                prof_ = t.profile_read() if profile else {}
                t.profile(False)
                return secs, n_, prof_
            t0 = time.perf_counter()
            first = whole(False)            # every allocation (128 GiB of table, ~100 GB of scratch)
            wall_first = time.perf_counter() - t0
            runs = [whole(False) for _ in range(2)]
            dt, n = min(runs, key=lambda x: x[0])[:2]
            _dtp, n_p, prof = whole(True)   # per-kernel device times: the same job with event timing on
            rep, _ = kernel_report(prof, n_exp, balg, pmc.get("C5_whole"))
            mine = table_digest(t)
            gd = golden.get("C5", {})
            same_input = bool(gd) and (gd["reads"], gd["read_len"], gd["k"], gd["genome"]) == (Rb, Lb, kb, Gb)
            gate = {"n": bool(n == n_exp == first[1] == n_p and all(x[1] == n_exp for x in runs)),
                    "equals_oracle_digests": bool(same_input and all(mine[f_] == gd[f_] for f_ in DIGEST_FIELDS) and n == gd["n"] and t.consumed == gd["consumed"])}
            entry = {"kmers": n_exp, "kmers_per_s": n_exp / dt, "seconds": dt, "seconds_runs": [round(x[0], 4) for x in runs], "seconds_first_run_with_allocations": first[0],
                     "wall_first_run_with_generation": wall_first, "pieces": pieces, "table_slots": t.capacity, "distinct": mine["len"],
                     "launches": {kn: v[0] for kn, v in prof.items()},
                     "what": f"{Rb} x {Lb} bp, k={kb}, genome {Gb} bp on ONE GPU: {pieces} device-generated pieces consumed call after call into one "
                             f"2^33-slot table; seconds = the consume calls + final sync (generation untimed); best of 2 after an allocating run",
                     "gate": gate, **rep}
            configs["C5_whole"] = entry
            log(f"C5_whole: {entry['kmers_per_s']:.3g} k-mers/s in {dt:.3f} s, gate {gate}")
            assert ablate or all(gate.values()), ("C5_whole", gate, mine)
            t.release_scratch()
            del t, r, g
            torch.cuda.empty_cache()
    # ------------------------------------------------------------------------------------------ inputs with sequencing errors
    if rank == 0 and world == 1:
        for name in [c for c in want if c in ERR]:
            Rb, Lb, kb, Gb, model = ERR[name]
            free, _tot = torch.cuda.mem_get_info()
            if Rb > 5_000_000 and free < 230 * (1 << 30):
                configs[name] = {"skipped": f"needs a whole MI355X: {free >> 30} GiB free"}
                continue
            m = dict(sub_ppm=0, n_ppm=0, sorted_total=0)
            m.update(model)
            g = torch.empty(Gb, dtype=torch.uint8, device="cuda")
            r = torch.empty(Rb * (Lb + 1), dtype=torch.uint8, device="cuda")
            assert lib.kct_synth_genome_device(g.data_ptr(), Gb, SEED_G, stream) == 0
            assert lib.kct_synth_reads_device_ex(r.data_ptr(), g.data_ptr(), Gb, 0, Rb, Lb, SEED_R, m["sub_ppm"], m["n_ppm"], m["sorted_total"], 7331, stream) == 0
            torch.cuda.synchronize()
            # distinct k-mers: the genome's plus ~ (k - 1) / 2 .. k new ones per substituted base
            distinct_hint = int(min(Gb, Rb * (Lb - kb + 1)) + Rb * Lb * m["sub_ppm"] / 1e6 * kb)
            balg = Lb / (Lb - kb + 1) + 24.0
            entry, tables = {"model": model}, {}
            for path in ("auto", "partitioned", "direct"):
                t = KmerCountTable(kb, capacity=distinct_hint)
                t.set_path(path)
                call = lambda: t.consume_device(r.data_ptr(), r.numel(), Rb * Lb)  # noqa: E731
                call(); t.sync()                                  # allocations
                runs = [timed_call(t, call, True, path) for _ in range(3 if path != "direct" else 1)]
                dt, n, prof = sorted(runs, key=lambda x: x[0])[len(runs) // 2]
                tables[path] = (n, len(t), t.sum_counts) + t.digest()
                if path == "auto":
                    rep, _ = kernel_report(prof, n, balg, None)
                    k1 = sorted((kn for kn in prof if kn.startswith("partition_windows_kernel")), key=lambda kn: -prof[kn][1])  # the pass's K1 first, a probe's after
                    entry.update({"kmers": n, "kmers_per_s": n / dt, "seconds": dt, "table_slots": t.capacity, "distinct": len(t),
                                  "path_chosen": ("compact dedupe-first" if "compact" in k1[0] else "64-bit dedupe-first" if "raw" in k1[0]
                                                  else "hash every window") + (" after a probe" if len(k1) > 1 else "") if k1 else "direct", **rep})
                    if checker and Rb <= 5_000_000:   # the oracle's table of the same reads, pair by pair
                        ss = oracle.ShardSet(kb, Lb, genome=g.cpu().numpy(), nreads=Rb, seed_r=SEED_R, **m)
                        dk, dc = t.dump_arrays(0)
                        entry["oracle_mismatches"] = int(ss.mismatches(dk, dc)) + abs(int(dk.size) - ss.digest()["len"])
                        del ss, dk, dc
                elif path == "partitioned":
                    entry["partitioned_path_kmers_per_s"] = n / dt
                else:
                    entry["direct_path_kmers_per_s"] = n / dt
                t.release_scratch()
                del t
                torch.cuda.empty_cache()
            gate = {"equals_partitioned_and_direct_path": bool(tables["auto"] == tables["partitioned"] == tables["direct"]),
                    "sum_counts_is_n": bool(tables["auto"][0] == tables["auto"][2])}
            if "oracle_mismatches" in entry:
                gate["equals_oracle_table"] = entry["oracle_mismatches"] == 0
            # a wrong guess of the path may cost speed, not a factor: never more than 10 % slower than hashing every window
            entry["vs_partitioned"] = entry["kmers_per_s"] / entry["partitioned_path_kmers_per_s"]
            entry["gate"] = gate
            configs[name] = entry
            log(f"{name}: {entry['kmers_per_s']:.3g} k-mers/s ({entry['path_chosen']}; partitioned {entry['partitioned_path_kmers_per_s']:.3g}), gate {gate}")
            assert ablate or all(gate.values()), (name, gate)
            del g, r
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------------------------------ N ranks: C4 / C5 as one job
    if world > 1:
        from oxli_amd.distributed import consume_device_early
        shared = world > torch.cuda.device_count()       # (debugging: several ranks on one GPU over gloo -- reduced sizes)
        cdev = "cuda" if args.backend == "nccl" else "cpu"

        def gmax(x):
            t_ = torch.tensor([x], dtype=torch.float64, device=cdev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())

        for name in [c for c in want if c in MULTI]:
            Rt, Lb, kb, Gb = MULTI[name]
            note = None
            if shared:
                Rt, Gb, note = Rt // 16 if Lb < 1000 else Rt // 64, Gb // 16, "REDUCED: ranks share one GPU (reads and genome / 16)"
            per = Rt // world
            n_exp_total = per * world * (Lb - kb + 1)
            free, _tot = torch.cuda.mem_get_info()
            need = per * (Lb + 1) * 4 + Gb + (64 << 30)
            ok_mem = torch.tensor([1 if free > need else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(ok_mem, op=dist.ReduceOp.MIN)
            if not int(ok_mem.item()):
                configs[name] = {"skipped": f"rank needs ~{need >> 30} GiB of HBM, {free >> 30} free"}
                continue
            g = torch.empty(Gb, dtype=torch.uint8, device="cuda")
            r = torch.empty(per * (Lb + 1), dtype=torch.uint8, device="cuda")
            assert lib.kct_synth_genome_device(g.data_ptr(), Gb, SEED_G, stream) == 0
            assert lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), Gb, rank * per, per, Lb, SEED_R, stream) == 0
            torch.cuda.synchronize()
            del g
            distinct_global = min(Gb, n_exp_total)
            # both routes, each through every exchange there is: "late" / "early" = libkct_rccl.so where it can run, else torch.distributed;
            # "late_torch" / "early_torch" = torch.distributed beside the native one
            routes = {}
            for xi, (xname, xobj) in enumerate(exchanges):
                suffix = "" if xi == 0 else "_" + xname
                routes["late" + suffix] = ("late", None, xname, xobj)
                routes["early" + suffix] = ("early", "super-k-mers", xname, xobj)
            entry = {"world": world, "reads_total": per * world, "reads_per_rank": per, "read_len": Lb, "k": kb, "genome": Gb, "kmers": n_exp_total,
                     "scaling": "strong", "routes": {}}
            if note:
                entry["note"] = note
            digests = {}
            for rname, (route, mode, xname, xobj) in routes.items():
                # late: a rank's private table meets k-mers from all over the genome; early: an owner holds 1 / world of the key space
                distinct_rank = min(Gb, per * (Lb - kb + 1)) if route == "late" else distinct_global // world + (1 << 16)
                t = KmerCountTable(kb, capacity=max(distinct_rank, 400_000), device=local)
                stats = {}

                def job():
                    t.clear()
                    if route == "late":
                        t.resize(max(distinct_rank, 400_000))
                    torch.cuda.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    if route == "late":
                        n = t.consume_device(r.data_ptr(), r.numel(), per * Lb)
                        t_x = time.perf_counter()
                        stats["pairs_received"] = merge_across_ranks(t, native=xobj)
                        stats["merge_ms"] = (time.perf_counter() - t_x) * 1e3
                    else:
                        n, st_ = consume_device_early(t, r.data_ptr(), r.numel(), per * Lb, native=xobj)
                        stats.update(st_)
                        stats["bytes_sent_per_window"] = st_["bytes_sent"] / max(1, st_["windows_sent"])
                    t.sync()        # the dedupe-first modes' conversion is part of the job
                    torch.cuda.synchronize()
                    dist.barrier()
                    return gmax(time.perf_counter() - t0), n

                job()   # allocations, communicator warm-up
                runs = sorted(job() for _ in range(3))
                dt, n = runs[1]
                n_all = global_scalar_sum(n, "cuda")
                per_rank_stats = [None] * world
                dist.all_gather_object(per_rank_stats, {k_: v for k_, v in stats.items() if isinstance(v, (int, float, str))})
                dg = [None] * world
                dist.all_gather_object(dg, (len(t), t.sum_counts) + t.digest())
                glen, gsum = sum(d[0] for d in dg), sum(d[1] for d in dg)
                gshc = sum(d[2] for d in dg) & ((1 << 64) - 1)
                gx = 0
                for d in dg:
                    gx ^= d[3]
                gsq = sum(d[4] for d in dg) & ((1 << 64) - 1)
                digests[rname] = (glen, gsum, gshc, gx, gsq)
                entry["routes"][rname] = {"kmers_per_s": n_exp_total / dt, "seconds": dt, "seconds_min_max": [runs[0][0], runs[-1][0]], "mode": mode,
                                          "exchange": "libkct_rccl.so" if xobj is not None else f"torch.distributed ({args.backend})",
                                          "n": n_all, "distinct_global": glen, "per_rank": per_rank_stats}
                t.release_scratch()
                del t
                torch.cuda.empty_cache()
            gate = {"n": all(v["n"] == n_exp_total for v in entry["routes"].values()), "sum_counts": all(d[1] == n_exp_total for d in digests.values()),
                    "routes_agree_on_len_and_digests": len(set(digests.values())) == 1}
            # THE ORACLE'S TABLE of the job's union: the committed digests at full size (C4's union is the north-star run's input), else
            # -- reduced sizes -- computed now on rank 0's host cores
            gd = golden.get({"C4": "north_star_k21", "C5": "C5"}[name])
            want_d = None
            if gd and (gd["reads"], gd["read_len"], gd["k"], gd["genome"]) == (per * world, Lb, kb, Gb):
                want_d = (gd["len"], gd["sum_counts"], gd["sum_hc"], gd["xor_hc"], gd["sum_sq"])
                entry["oracle_digests"] = "tests/golden/config_digests.json"
            elif n_exp_total <= 4_000_000_000 and not args.no_cpu_baseline:
                if rank == 0:
                    import oracle
                    ss = oracle.ShardSet(kb, Lb, genome=oracle.synth_genome(Gb, SEED_G), nreads=per * world, seed_r=SEED_R, expect_keys=distinct_global)
                    d_ = ss.digest()
                    del ss
                    want_d = (d_["len"], d_["sum_counts"], d_["sum_hc"], d_["xor_hc"], d_["sum_sq"])
                box = [want_d]
                dist.broadcast_object_list(box, src=0)
                want_d = box[0]
                entry["oracle_digests"] = "computed on rank 0's host (oracle.ShardSet)"
            if want_d is not None:
                gate["equals_oracle_digests"] = all(tuple(d) == tuple(want_d) for d in digests.values())
            else:
                entry["oracle_digests"] = None
            best = max(entry["routes"], key=lambda k_: entry["routes"][k_]["kmers_per_s"])
            entry.update({"kmers_per_s": entry["routes"][best]["kmers_per_s"], "seconds": entry["routes"][best]["seconds"], "best_route": best, "gate": gate})
            configs[name] = entry
            log(f"{name} on {world} ranks: " + ", ".join(f"{k_} {v['kmers_per_s']:.3g}" for k_, v in entry["routes"].items()) + f" k-mers/s, gate {gate}")
            assert ablate or all(gate.values()), (name, gate, digests)
            del r
            torch.cuda.empty_cache()

    if configs:
        result["configs"] = configs

    # ------------------------------------------------------------------------------------------ CPU baseline + final check
    if checker and not args.no_headline:
        base, (sample_reads, sample_ref) = cpu_baseline(args, log)
        result["cpu_baseline"] = base
        result["speedup_vs_cpu_baseline"] = result["value"] / base["value"]
        if "north_star_k21" in configs and "kmers_per_s" in configs["north_star_k21"]:
            result["north_star_speedup_vs_cpu_baseline"] = configs["north_star_k21"]["kmers_per_s"] / base["value"]
        if not args.no_verify:
            # the same sample through the GPU must give the oracle's table bit for bit
            ns = sample_reads.shape[0]
            dev = torch.from_numpy(np.ascontiguousarray(sample_reads).reshape(-1)).cuda()
            sample_table.clear()
            n = sample_table.consume_device(dev.data_ptr(), dev.numel(), ns * L)
            dk, dc = sample_table.dump_arrays(1)
            rk, rc = sample_ref.dump_arrays()
            ok = n == ns * (L - k + 1) and np.array_equal(dk, rk) and np.array_equal(dc, rc)
            result["verified_vs_oracle"] = bool(ok)
            assert ok, "GPU table differs from the CPU oracle on the baseline sample"
    if rank == 0:
        # The full record goes to a file (and, with --verbose, to stdout); the ONE line on stdout is a compact version of
        # it -- every figure, none of the prose -- so that it survives a size-limited capture of the output's tail.
        try:
            with open(args.detail_out, "w") as f:
                json.dump(result, f, indent=1)
        except OSError:
            pass
        print(json.dumps(result if args.verbose else compact(result), separators=(",", ":") if not args.verbose else None), flush=True)
    if native is not None:
        native.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
