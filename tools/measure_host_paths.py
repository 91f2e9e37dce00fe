#!/usr/bin/env python3
"""Measures the paths either side of the hot path on the GPU box (numbers quoted in DESIGN.md):
PCIe-inclusive consume from host buffers, FASTA file ingestion (plain / gzip), device-sorted dump,
save / load.  Run: python tools/measure_host_paths.py > gpurun_out/host_paths.json"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (only to make the synthetic reads on the host)
from oxli_amd import KmerCountTable  # noqa: E402

R, L, K, G = 1_000_000, 150, 21, 5_000_000
genome = oracle.synth_genome(G)
reads = oracle.synth_reads(genome, 0, R, L)          # uint8 [R, L+1]
flat = np.ascontiguousarray(reads[:, :L]).reshape(-1)  # CSR bytes without separators
offs = (np.arange(R + 1, dtype=np.uint64) * L)
kmers = R * (L - K + 1)
out = {"workload": f"{R} x {L} bp reads, k={K}, host-resident"}


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


t = KmerCountTable(K, capacity=G)
t.consume_batch((flat, offs))  # warm: allocations, pinned staging
def run_batch():
    t.clear(); assert t.consume_batch((flat, offs)) == kmers
s = best(run_batch)
out["consume_batch_host_csr"] = {"seconds": s, "kmers_per_s": kmers / s, "host_bytes": int(flat.size),
                                 "note": "pack into pinned staging + H2D + device passes, single-buffered"}

strs = [reads[i, :L].tobytes().decode() for i in range(R)]   # what a screed loop would hold
def run_list():
    t.clear(); assert t.consume_batch(strs) == kmers
s = best(run_list)
out["consume_batch_list_of_str"] = {"seconds": s, "kmers_per_s": kmers / s, "note": "CSR built by csrc/pyfast.c, then as above"}
def run_loop():
    t.clear()
    n = 0
    for rec in strs:
        n += t.consume(rec)
    assert n == kmers and len(t) > 0
s = best(run_loop)
out["per_record_loop"] = {"seconds": s, "kmers_per_s": kmers / s, "us_per_call": s / R * 1e6}

tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "reads.fa")
with open(fa, "wb") as f:
    for i in range(R):
        f.write(b">r%d\n" % i); f.write(reads[i, :L].tobytes()); f.write(b"\n")
def run_file():
    t.clear(); assert t.consume_file(fa) == kmers
s = best(run_file)
out["consume_file_fasta_plain"] = {"seconds": s, "kmers_per_s": kmers / s, "file_bytes": os.path.getsize(fa)}
fagz = fa + ".gz"
with open(fa, "rb") as src, gzip.open(fagz, "wb", compresslevel=1) as dst:
    dst.write(src.read())
def run_gz():
    t.clear(); assert t.consume_file(fagz) == kmers
s = best(run_gz, reps=3)
out["consume_file_fasta_gzip"] = {"seconds": s, "kmers_per_s": kmers / s, "file_bytes": os.path.getsize(fagz),
                                  "note": "single member, text < 2 GiB: inflated in one piece by libdeflate, parsed like a plain file"}
os.environ["KCT_GZIP_WHOLE_MAX"] = "0"         # the streaming reader (larger files, several members, no libdeflate)
s = best(run_gz, reps=2)
out["consume_file_fasta_gzip"]["streaming_zlib"] = kmers / s
del os.environ["KCT_GZIP_WHOLE_MAX"]
# the same text as BGZF (bgzip's blocked gzip: inflated by several threads)
import struct, zlib
fabgz = fa + ".bgz.gz"
with open(fa, "rb") as src, open(fabgz, "wb") as dst:
    while True:
        piece = src.read(65280)
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(piece) + comp.flush()
        dst.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1) + body +
                  struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
        if not piece:
            break
def run_bgz():
    t.clear(); assert t.consume_file(fabgz) == kmers
s = best(run_bgz, reps=4)
out["consume_file_fasta_bgzf"] = {"seconds": s, "kmers_per_s": kmers / s, "file_bytes": os.path.getsize(fabgz)}
for th in (4, 8, 16, 24):                      # slot threads (inflate + parse); the default is min(16, half the host's threads)
    os.environ["KCT_FILE_THREADS"] = str(th)
    s = best(run_bgz, reps=3)
    out["consume_file_fasta_bgzf"][f"threads_{th}"] = kmers / s
del os.environ["KCT_FILE_THREADS"]
os.environ["KCT_NO_LIBDEFLATE"] = "1"          # zlib's inflate instead of libdeflate's
s = best(run_bgz, reps=3)
out["consume_file_fasta_bgzf"]["zlib_inflate"] = kmers / s
del os.environ["KCT_NO_LIBDEFLATE"]

n = len(t)
for order, name in ((0, "dump_unsorted"), (1, "dump_sorted_by_hash"), (2, "dump_sorted_by_count_hash")):
    s = best(lambda: t.dump_arrays(order))
    out[name] = {"seconds": s, "pairs": n, "pairs_per_s": n / s}
sv = os.path.join(tmp, "t.json.gz")
t0 = time.perf_counter(); t.save(sv); out["save_gzip_json"] = {"seconds": time.perf_counter() - t0, "file_bytes": os.path.getsize(sv)}
t0 = time.perf_counter(); u = KmerCountTable.load(sv); out["load_gzip_json"] = {"seconds": time.perf_counter() - t0}
assert len(u) == n and u.sum_counts == t.sum_counts
print(json.dumps(out, indent=1))
