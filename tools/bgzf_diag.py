"""kct_consume_file on a C2-sized BGZF FASTA in /dev/shm: rate by slot-thread count (arguments) and one call's timeline (KCT_DEBUG=1)."""
import os, struct, sys, time, zlib
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
text = bytes(buf)
path = "/dev/shm/c2.bgz.gz"
with open(path, "wb") as dst:   # SAM specification 4.1: members of <= 64 KiB of text with a 'BC' size field, then an empty one
    for o in list(range(0, len(text), 65280)) + [len(text)]:
        piece = text[o:o + 65280]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(piece) + comp.flush()
        dst.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1) + body +
                  struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
print("text", len(text), "bgzf", os.path.getsize(path), flush=True)
for threads in (sys.argv[1:] or ["32"]):
    os.environ["KCT_FILE_THREADS"] = threads
    os.environ.pop("KCT_DEBUG", None)
    t = KmerCountTable(21, capacity=5_000_000)
    best = 1e9
    for rep in range(6):
        t.clear(); t0 = time.time(); n = t.consume_file(path); t.sync(); best = min(best, time.time() - t0)
    print("threads", threads, n, "best %.2f ms" % (best * 1e3), "%.3g k-mers/s" % (n / best), "%.1f GB/s of text" % (len(text) / best / 1e9), flush=True)
os.environ["KCT_DEBUG"] = "1"
t2 = KmerCountTable(21, capacity=5_000_000)
t2.consume_file(path); t2.sync(); t2.clear()
sys.stderr.write("---- timeline of the next call\n")
t0 = time.time(); t2.consume_file(path); t2.sync(); sys.stderr.write("---- call + sync %.2f ms\n" % ((time.time() - t0) * 1e3))
os.remove(path)
