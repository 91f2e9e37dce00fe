"""One streaming-route call over a 627 MB FASTQ (.gz, one member) with the window timeline on stderr (KCT_DEBUG=1, PGZ_TIMING=1)."""
import os, sys, time, zlib
sys.path.insert(0, ".")
import numpy as np
os.environ.setdefault("KCT_GZIP_WHOLE_MAX", "0")   # (set it to 3000000000 to trace the one-piece route instead)
from oxli_amd import KmerCountTable
N = 2_000_000
rng = np.random.default_rng(3)
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, 150), dtype=np.uint8)]
q = np.full((N, 150), ord("F"), dtype=np.uint8); q[rng.random((N, 150)) < 0.1] = ord(",")
buf = bytearray()
for i in range(N):
    buf += b"@r%d\n" % i; buf += seqs[i].tobytes(); buf += b"\n+\n"; buf += q[i].tobytes(); buf += b"\n"
path = "/dev/shm/big.fastq.gz"
co = zlib.compressobj(6, zlib.DEFLATED, 31)
open(path, "wb").write(co.compress(bytes(buf)) + co.flush())
t = KmerCountTable(21, capacity=300_000_000)
t.consume_file(path); t.sync(); t.clear()
os.environ["KCT_DEBUG"] = "1"; os.environ["PGZ_TIMING"] = "1"
t2 = KmerCountTable(21, capacity=300_000_000)
t0 = time.time(); n = t2.consume_file(path); t2.sync(); dt = time.time() - t0
print("streaming", n, "%.3f s" % dt, "%.2f GB/s of text" % (len(buf) / dt / 1e9))
os.remove(path)
