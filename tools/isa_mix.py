#!/usr/bin/env python3
"""Static instruction mix of one K1 instantiation, by what the instructions are FOR: compiles the kernel alone with line tables
(-gline-tables-only --save-temps), attributes every ISA instruction to its source line, sums by category.

    python tools/isa_mix.py [KW KC MODE]        # default 1 21 2: the compact dedupe-first K1 of the headline / north-star run

Categories are source-line ranges of oxli_amd/csrc/{k1_kernel.h, window_kernels.h, kmer_device.h} (found by name below, so that they
survive edits); "per window" divides by the 16 windows a thread walks per tile and weighs the flush code by its executions per tile.
Prints a markdown table (profiles/r04_K1_compact_isa_mix.md is this output plus commentary)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oxli_amd", "csrc")


def fn_range(path, name):
    """(first, last) line of the function or lambda whose definition line contains `name`."""
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        if name in l:
            depth, j, seen = 0, i, False
            while j < len(lines):
                depth += lines[j].count("{") - lines[j].count("}")
                seen = seen or "{" in lines[j]
                if seen and depth <= 0:
                    return i + 1, j + 1
                j += 1
    raise KeyError(name)


def main():
    kw, kc, mode = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 21, 2)
    k1, wk, kd = (os.path.join(CSRC, f) for f in ("k1_kernel.h", "window_kernels.h", "kmer_device.h"))
    cats = [  # (category, file, first, last)
        ("flush (ring_flush)", "k1_kernel.h") + fn_range(k1, "__device__ __forceinline__ bool ring_flush"),
        ("overflow route (cold)", "k1_kernel.h") + fn_range(k1, "auto overflow_hash = [&]"),
        ("tile load + prefetch", "k1_kernel.h") + fn_range(k1, "auto load_chunk = [&]"),
        ("tile load + prefetch", "k1_kernel.h") + fn_range(k1, "auto load_group = [&]"),
        ("append: commit (ring store)", "k1_kernel.h") + fn_range(k1, "auto commit = [&]"),
        ("append: bin + cursor bump", "k1_kernel.h") + fn_range(k1, "auto sink = [&]"),
        ("encode (ASCII -> 2 bits + validity)", "kmer_device.h") + fn_range(kd, "void encode4("),
        ("encode (ASCII -> 2 bits + validity)", "kmer_device.h") + fn_range(kd, "void encode16("),
        ("mix42", "kmer_device.h") + fn_range(kd, "u64 mix42("),
        ("roll (push / mask)", "kmer_device.h") + fn_range(kd, "void push_fw("),
        ("roll (push / mask)", "kmer_device.h") + fn_range(kd, "void push_rc("),
        ("roll (push / mask)", "kmer_device.h") + fn_range(kd, "void mask_k("),
        ("first window (assemble, revcomp)", "kmer_device.h") + fn_range(kd, "revcomp_packed("),
        ("first window (assemble, revcomp)", "kmer_device.h") + fn_range(kd, "void left_align("),
        ("first window (assemble, revcomp)", "kmer_device.h") + fn_range(kd, "u64 reverse_pairs64("),
        ("canonical min", "kmer_device.h") + fn_range(kd, "bool less_eq("),
        ("window walk (validity run, streams, loop)", "window_kernels.h") + fn_range(wk, "void walk_windows_encoded("),
    ]
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "k1only.hip")
        open(src, "w").write('#include "k1_kernel.h"\ntemplate __global__ void kct::partition_windows_kernel<%d, %d, %d, false>'
                             '(const unsigned char *, kct::u64, int, kct::u64, kct::PartitionArgs);\n' % (kw, kc, mode))
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-fPIC", "-DKCT_BUILDING_LIBRARY",
                        "-Wno-unused-function", "--save-temps", "-c", "-o", "k1only.o", src, "-I", CSRC], cwd=tmp, check=True, capture_output=True)
        s = open(os.path.join(tmp, "k1only-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    files = {int(m.group(1)): (m.group(3) or m.group(2)).split("/")[-1] for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
    i = s.index("partition_windows_kernel")
    i = s.index("\n", s.index(":", i))
    body = s[i: s.index("s_endpgm", i)]
    cur, tally = None, collections.defaultdict(collections.Counter)
    for line in body.split("\n"):
        t = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        cat = "kernel frame (ring / cursor init, drain, epilogue)"
        if cur:
            for name, f, a, b in cats:
                if cur[0] == f and a <= cur[1] <= b:
                    cat = name
                    break
            else:
                if cur[0].startswith("amd_") or cur[1] == 0:
                    cat = "compiler-attributed (atomics expanded, line 0)"
        tally[cat][kind] += 1
    print(f"| category (partition_windows_kernel<{kw},{kc},{mode}>) | VALU | SALU | LDS | VMEM |\n|---|---|---|---|---|")
    tot = collections.Counter()
    for cat in sorted(tally, key=lambda c: -sum(tally[c].values())):
        c = tally[cat]
        tot.update(c)
        print(f"| {cat} | {c['VALU']} | {c['SALU']} | {c['LDS']} | {c['VMEM']} |")
    print(f"| **whole kernel (static)** | {tot['VALU']} | {tot['SALU']} | {tot['LDS']} | {tot['VMEM']} |")


if __name__ == "__main__":
    main()
