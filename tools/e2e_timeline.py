"""Timeline of one kct_consume_batch call from host memory (KCT_DEBUG lines), packed and ASCII upload: python tools/e2e_timeline.py"""
import os, sys, time
os.environ["KCT_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from oxli_amd import KmerCountTable
R, L, K, G = 1_000_000, 150, 21, 5_000_000
reads = oracle.synth_reads(oracle.synth_genome(G), 0, R, L)
flat = np.ascontiguousarray(reads[:, :L]).reshape(-1)
offs = np.arange(R + 1, dtype=np.uint64) * L
for packed in (True, False):
    t = KmerCountTable(K, capacity=G)
    t.set_packed_upload(packed)
    for i in range(4):
        t.clear()
        print(f"---- packed={packed} run {i}", file=sys.stderr, flush=True)
        t0 = time.perf_counter(); n = t.consume_batch((flat, offs)); t1 = time.perf_counter(); t.sync(); t2 = time.perf_counter()
        print(f"==== packed={packed}: consume_batch {1e3*(t1-t0):.3f} ms, + sync {1e3*(t2-t1):.3f} ms", file=sys.stderr, flush=True)
