#!/usr/bin/env python3
"""Per-call latency of the point API and of per-record consume (the reference's calling pattern)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import random
from oxli_amd import KmerCountTable
rng = random.Random(1)
reads = ["".join(rng.choice("ACGT") for _ in range(150)) for _ in range(2000)]
t = KmerCountTable(21)
for r in reads[:200]:
    t.consume(r)
out = {}
def bench(name, fn, n):
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    out[name] = {"us_per_call": (time.perf_counter() - t0) / n * 1e6, "calls": n}
bench("consume(150bp read)", lambda i: t.consume(reads[i % len(reads)]), 2000)
kmer = reads[0][:21]
h = t.hash_kmer(kmer)
bench("hash_kmer", lambda i: t.hash_kmer(kmer), 1000)
bench("count(kmer)", lambda i: t.count(kmer), 1000)
bench("get(kmer)", lambda i: t.get(kmer), 1000)
bench("get_hash", lambda i: t.get_hash(h), 1000)
bench("count_hash", lambda i: t.count_hash(h), 1000)
td = KmerCountTable(21, deferred=True)
def deferred_pass(i):
    td.consume(reads[i % len(reads)])
bench("consume(150bp read), deferred", deferred_pass, 200000)
t0 = time.perf_counter(); len(td); out["deferred flush (200000 reads)"] = {"us_total": (time.perf_counter() - t0) * 1e6}
t0 = time.perf_counter(); n = t.consume_batch(reads); out["consume_batch(2000 reads)"] = {"us_total": (time.perf_counter() - t0) * 1e6}
print(json.dumps(out, indent=1))
