#!/usr/bin/env python3
"""Round 6: kct_consume_file on a single-member FASTQ .gz of real-world shape (2 M x 150 bp reads with qualities, zlib level 6: 627 MB of
text, 144 MB compressed) -- parallel_inflate.h's phases (PGZ_TIMING=1) and the whole call, with KCT_GZIP_THREADS = 16 / 32 / 64 / 128 and
with the parallel inflater off."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oxli_amd import KmerCountTable

N = 2_000_000
rng = np.random.default_rng(3)
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, 150), dtype=np.uint8)]
q = np.full((N, 150), ord("F"), dtype=np.uint8)
q[rng.random((N, 150)) < 0.1] = ord(",")
buf = bytearray()
for i in range(N):
    buf += b"@r%d\n" % i; buf += seqs[i].tobytes(); buf += b"\n+\n"; buf += q[i].tobytes(); buf += b"\n"
path = "/dev/shm/big.fastq.gz"
co = zlib.compressobj(6, zlib.DEFLATED, 31)
open(path, "wb").write(co.compress(bytes(buf)) + co.flush())
print("text", len(buf), "gz", os.path.getsize(path), flush=True)
os.environ["PGZ_TIMING"] = "1"
for env in ({}, {"KCT_NO_PARALLEL_GZIP": "1"}):
    for kk in ("KCT_GZIP_THREADS", "KCT_NO_PARALLEL_GZIP"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    t = KmerCountTable(21, capacity=300_000_000)
    for rep in range(3):
        t.clear()
        t0 = time.time(); n = t.consume_file(path); t.sync(); dt = time.time() - t0
    print(env, n, "%.3f s" % dt, "%.3g k-mers/s" % (n / dt), "%.2f GB/s of text" % (len(buf) / dt / 1e9), file=sys.stderr, flush=True)
os.remove(path)
