import os, sys, time, zlib
sys.path.insert(0, ".")
import numpy as np
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
print("text", len(buf), "gz", os.path.getsize(path), flush=True)
# KCT_GZIP_WHOLE_MAX=0: the streaming route, as for a text beyond 2 GiB; without it: the one-piece route
for env in ({"KCT_GZIP_WHOLE_MAX": "0"}, {"KCT_GZIP_WHOLE_MAX": "0", "KCT_NO_PARALLEL_GZIP": "1"}, {}):
    os.environ.pop("KCT_NO_PARALLEL_GZIP", None); os.environ.pop("KCT_GZIP_WHOLE_MAX", None); os.environ.update(env)
    t = KmerCountTable(21, capacity=300_000_000)
    for rep in range(3):
        t.clear(); t0 = time.time(); n = t.consume_file(path); t.sync(); dt = time.time() - t0
    print("route", env, n, "%.3f s" % dt, "%.3g k-mers/s" % (n / dt), "%.2f GB/s of text" % (len(buf) / dt / 1e9), flush=True)
os.remove(path)
