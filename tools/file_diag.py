"""kct_consume_file on a C2-sized plain FASTA in /dev/shm: the call's timeline (KCT_DEBUG=1) and its rate for several parser-thread counts."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from oxli_amd import KmerCountTable
N, L = 1_000_000, 150
rng = np.random.default_rng(3)
genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=5_000_000)]
pos = rng.integers(0, genome.size - L, size=N)
rows = genome[pos[:, None] + np.arange(L)[None, :]]
buf = bytearray()
for i in range(N):
    buf += b">r%d\n" % i; buf += rows[i].tobytes(); buf += b"\n"
path = "/dev/shm/c2.fa"
open(path, "wb").write(buf)
print("text", len(buf), flush=True)
for threads in (sys.argv[1:] or ["8"]):
    os.environ["KCT_FILE_THREADS"] = threads
    os.environ.pop("KCT_DEBUG", None)
    t = KmerCountTable(21, capacity=8_000_000)
    best = 1e9
    for rep in range(6):
        t0 = time.time(); n = t.consume_file(path); t.sync(); dt = time.time() - t0
        best = min(best, dt)
    print("threads", threads, n, "best %.2f ms" % (best * 1e3), "%.3g k-mers/s" % (n / best), "%.1f GB/s of text" % (len(buf) / best / 1e9), flush=True)
    os.environ["KCT_DEBUG"] = "1"
    t2 = KmerCountTable(21, capacity=8_000_000)
    t2.consume_file(path); t2.sync()
    sys.stderr.write("---- timeline of the next call\n")
    t0 = time.time(); t2.consume_file(path); t2.sync(); sys.stderr.write("---- call + sync %.2f ms\n" % ((time.time() - t0) * 1e3))
os.remove(path)
