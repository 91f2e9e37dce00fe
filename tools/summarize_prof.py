#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (made by tools/collect_r04.sh on the GPU box) into the small, committed files under profiles/:

  profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of that workload
  profiles/<tag>_pmc_full.json                 per workload, per kernel: launches, mean of every PMC counter, derived figures
  profiles/pmc_<tag>.json                      what bench.py reads: {source_sha, <workload>: {kernels: {name: {hbm_bytes_per_launch}}, valu, atomics}}
  profiles/<tag>_bench.json                    the un-profiled bench.py line of the same build

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streams at 64
bytes, i.e. reports half of a streaming read: kernels whose reads are wide streams get the x2 correction (STREAMING
below); the direct atomic kernel's reads are 64-byte key-line probes and are left as measured.

--on-box: run on the GPU box right after collection: writes <prof dir>/summary.json and deletes the raw per-dispatch CSVs
(hundreds of MB) so that only the summary and the stats files travel back.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
on_box = "--on-box" in sys.argv
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")

STREAMING = ("partition_windows_kernel", "aggregate_blocks", "repartition_kernel", "flush_partition_kernel", "aggregate_pairs_kernel", "split_superkmers_kernel", "gather_units_kernel", "stage_stream_kernel")
CLOCK_HZ, SIMDS = 2.4e9, 256 * 4


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("kct::", "").replace("(anonymous namespace)::", "").strip()
    base = n.split("<")[0].strip()
    targs = n[n.index("<") + 1:n.rindex(">")] if "<" in n else ""
    if base == "partition_windows_kernel":   # <KW, KC, MODE, RUNS>
        ta = [x.strip() for x in targs.split(",")] if targs else []
        mode = ta[2] if len(ta) > 2 else "0"
        runs = len(ta) > 3 and ta[3] in ("true", "1")
        tagm = {"0": "", "1": "raw", "true": "raw", "2": "compact", "3": "raw128"}.get(mode, "")
        inner = ", ".join(x for x in (tagm, "runs" if runs else "") if x)
        return base + (f"<{inner}>" if inner else "")
    if base == "split_superkmers_kernel":
        return base
    if base == "repartition_kernel":
        targs = targs.strip()
        return base + ("<pairs>" if "HIP_vector_type" in targs or "ulonglong2" in targs else "<compact>" if targs.startswith("unsigned int") else "")
    return base


def summarize(wdir):
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(wdir, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    stats = {}
    sf = sorted(glob.glob(os.path.join(wdir, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if sf:
        for row in csv.DictReader(open(sf[-1])):
            k = short(row["Name"])
            e = stats.setdefault(k, {"calls": 0, "total_ns": 0.0})
            e["calls"] += int(row["Calls"]); e["total_ns"] += float(row["TotalDurationNs"])
    out = {}
    for k, counters in pmc.items():
        if not (k.startswith(("partition", "aggregate", "repartition", "flush", "merge", "count_windows", "shadow", "split_superkmers", "gather_units", "run_directory", "stage_stream"))):
            continue
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        e = {"launches_seen": max(len(v) for v in counters.values()), "mean": mean}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            x2 = k.startswith(STREAMING)
            fetch, write = mean["FETCH_SIZE"] * 1024 * (2 if x2 else 1), mean["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = fetch + write
            e["fetch_bytes_per_launch"], e["write_bytes_per_launch"], e["fetch_x2_correction_applied"] = fetch, write, x2
        if k in stats and stats[k]["calls"]:
            e["avg_launch_ns"] = stats[k]["total_ns"] / stats[k]["calls"]
            e["calls_in_trace"] = stats[k]["calls"]
            if "hbm_bytes_per_launch" in e:
                e["hbm_GBs"] = e["hbm_bytes_per_launch"] / e["avg_launch_ns"]
            if "SQ_INSTS_VALU" in mean:
                # Round 6: no "floor" any more.  The launch's cycles come from GRBM_GUI_ACTIVE / 8 XCDs where it was
                # collected (the chip clocks down under load: ~2.2 GHz, not 2.4), and a VALU wave instruction occupies its SIMD for ~2
                # cycles (VOP2 adds / xors) to ~4 (multiplies, VOP3: tools/valu_rate.hip, 4 waves per SIMD) -- SQ_ACTIVE_INST_VALU is no
                # measurement of that: it charges every instruction one quad-cycle.  Hence a RANGE, not a figure.
                cycles = mean["GRBM_GUI_ACTIVE"] / 8 if mean.get("GRBM_GUI_ACTIVE") else e["avg_launch_ns"] * 1e-9 * CLOCK_HZ
                e["launch_cycles"] = cycles
                e["clock_GHz"] = cycles / e["avg_launch_ns"] if mean.get("GRBM_GUI_ACTIVE") else None
                e["valu_busy_range"] = [mean["SQ_INSTS_VALU"] * 2 / (SIMDS * cycles), mean["SQ_INSTS_VALU"] * 4 / (SIMDS * cycles)]
        out[k] = e
    return out, (sf[-1] if sf else None)


def main():
    full = {}
    os.makedirs(dst, exist_ok=True)
    if not on_box and os.path.exists(os.path.join(src, "summary.json")):
        full = json.load(open(os.path.join(src, "summary.json")))
        for w in full:
            sf = os.path.join(src, w, "kernel_stats.csv")
            if os.path.exists(sf):
                shutil.copyfile(sf, os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))
    else:
        for wdir in sorted(glob.glob(os.path.join(src, "*"))):
            if not os.path.isdir(wdir):
                continue
            w = os.path.basename(wdir)
            full[w], sf = summarize(wdir)
            if sf:
                shutil.copyfile(sf, os.path.join(wdir, "kernel_stats.csv"))
                if not on_box:
                    shutil.copyfile(sf, os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))
        if on_box:
            json.dump(full, open(os.path.join(src, "summary.json"), "w"))
            for wdir in glob.glob(os.path.join(src, "*")):
                for sub in ("stats", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):  # (the raw per-dispatch files stay on the box)
                    shutil.rmtree(os.path.join(wdir, sub), ignore_errors=True)
            return
    from bench import source_sha
    bench = {}
    try:
        bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
        json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
    except Exception as e:  # noqa: BLE001
        print("no bench.json:", e)
    json.dump({"_how": f"tools/collect_{tag}.sh + tools/summarize_prof.py; FETCH_SIZE/WRITE_SIZE in KiB, separate passes; x2 on streaming reads",
               "source_sha": source_sha(), "workloads": full}, open(os.path.join(dst, f"{tag}_pmc_full.json"), "w"), indent=1)
    windows = {"C2": 1.3e8, "C2_hashing": 1.3e8, "cold_C2": 1.3e8}  # k-mers of a launch
    small = {"source_sha": source_sha(), "_how": f"profiles/{tag}_pmc_full.json condensed for bench.py"}
    for w, kernels in full.items():
        e = {"kernels": {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for k, v in kernels.items() if "hbm_bytes_per_launch" in v}}
        k1 = max((k for k in kernels if k.startswith("partition_windows_kernel")), key=lambda k: kernels[k].get("avg_launch_ns", 0) * kernels[k].get("calls_in_trace", 0), default=None)
        if k1 and "SQ_INSTS_VALU" in kernels[k1]["mean"] and w in windows:
            m = kernels[k1]["mean"]
            e["valu"] = {"kernel": k1, "valu_insts_per_window": m["SQ_INSTS_VALU"] * 64 / windows[w], "salu_insts_per_window": m["SQ_INSTS_SALU"] * 64 / windows[w],
                         "lds_insts_per_window": m["SQ_INSTS_LDS"] * 64 / windows[w], "valu_busy_range": kernels[k1].get("valu_busy_range"),
                         "clock_GHz": kernels[k1].get("clock_GHz"),
                         "wave_issue_stall_share": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") and m.get("SQ_WAIT_INST_ANY") else None,
                         "wave_wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                         "note": "per window = wave instructions x 64 lanes / window starts of a launch; valu_busy_range = VALU wave instructions x [2, 4] cycles / "
                                 "(1024 SIMDs x the launch's cycles, GRBM_GUI_ACTIVE / 8): ~2 cycles for VOP2 adds / xors, ~4 for multiplies and VOP3 forms (tools/valu_rate.hip); "
                                 "wave_wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt or a barrier); profiles/r06_k1_wait_split.json says what the waits are"}
        if w == "C2_direct" and "count_windows_kernel" in kernels and "TCC_EA0_ATOMIC_sum" in kernels["count_windows_kernel"]["mean"]:
            m = kernels["count_windows_kernel"]
            e["atomics"] = {"kernel": "count_windows_kernel", "memory_side_atomics_per_launch": m["mean"]["TCC_EA0_ATOMIC_sum"],
                            "per_kmer": m["mean"]["TCC_EA0_ATOMIC_sum"] / 1.3e8,
                            "per_second": m["mean"]["TCC_EA0_ATOMIC_sum"] / (m["avg_launch_ns"] * 1e-9) if "avg_launch_ns" in m else None}
        small[w] = e
    if "C2_direct" in small and "atomics" in small["C2_direct"] and "C2" in small:
        small["C2"]["atomics"] = dict(small["C2_direct"]["atomics"], note="the DIRECT path (one HBM atomic per k-mer), measured beside the partitioned paths, which issue none per k-mer")
    json.dump(small, open(os.path.join(dst, f"pmc_{tag}.json"), "w"), indent=1)
    for w, kernels in full.items():
        for k, v in kernels.items():
            if "hbm_bytes_per_launch" in v and v.get("avg_launch_ns", 0) > 50000:
                print(f"{w:16s} {k:36s} {v['avg_launch_ns'] / 1e6:9.3f} ms  {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB  {v.get('hbm_GBs', 0):7.0f} GB/s")


if __name__ == "__main__":
    main()
