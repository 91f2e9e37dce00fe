#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (made by tools/collect_profiles.sh on the GPU box) into the small,
committed files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg ns)
  profiles/<tag>_pmc.json           per-kernel averages of every PMC counter collected, plus derived
                                    HBM bytes per launch and per k-mer
  profiles/<tag>_bench.json         the un-profiled bench.py line of the same build
  profiles/pmc_traffic.json         {kernel: {"hbm_bytes_per_launch": ...}} read by bench.py for roofline.traffic

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide coalesced streams at 64 bytes, i.e. reports half of a streaming read.  Kernels whose reads are
16-byte-per-lane streams get the x2 correction; the direct atomic kernel's reads are 64-byte random
key-line probes, which are counted exactly (one 64-B request per probe), so only its stream
component would be corrected (<2 % of its bytes, left as measured).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

STREAMING = ("partition_windows_kernel", "partition_windows_kernel<raw>", "partition_windows_kernel<compact>", "aggregate_blocks_kernel", "aggregate_blocks32_kernel")  # reads are wide coalesced streams


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("kct::", "").replace("(anonymous namespace)::", "")
    base = n.split("<")[0].strip()
    if base == "partition_windows_kernel" and n.strip().endswith(", 2>"):
        base += "<compact>"  # the compact dedupe-first variant (32-bit entries)
    elif base == "partition_windows_kernel" and (n.strip().endswith("true>") or n.strip().endswith(", 1>")):
        base += "<raw>"  # the dedupe-first variant (no MurmurHash3): a different kernel for every purpose
    return base


stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")))
if stats:
    shutil.copyfile(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))

bench = {}
try:
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print("no bench.json:", e)

kmers = bench.get("roofline", {}).get("kmers_per_launch", 1.3e8) or 1.3e8
out, traffic = {}, {}
for k, counters in pmc.items():
    avg = {c: sum(v) / len(v) for c, v in counters.items()}
    entry = {"launches_seen": {c: len(v) for c, v in counters.items()}, "avg": avg}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        fetch = avg["FETCH_SIZE"] * 1024 * (2 if k in STREAMING else 1)
        write = avg["WRITE_SIZE"] * 1024
        entry["derived"] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                            "hbm_bytes_per_launch": fetch + write, "hbm_bytes_per_kmer": (fetch + write) / kmers,
                            "fetch_x2_correction_applied": k in STREAMING}
        traffic[k] = {"hbm_bytes_per_launch": fetch + write}
    out[k] = entry
json.dump({"_how": "tools/collect_profiles.sh + tools/summarize_profiles.py; FETCH_SIZE/WRITE_SIZE in KiB, separate passes",
           "workload": bench.get("config", {}).get("workload", "C2"), "kernels": out},
          open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
if traffic:
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for k in ("partition_windows_kernel", "partition_windows_kernel<raw>", "partition_windows_kernel<compact>", "aggregate_blocks_kernel", "aggregate_blocks32_kernel", "count_windows_kernel"):
    if k in out and "derived" in out[k]:
        print(k, json.dumps(out[k]["derived"]))
