#!/usr/bin/env python3
"""Experiment: do a VALU-bound K1 and an LDS-bound K2 of two independent tables share the CUs when issued on two streams?
Two host threads, each with its own table and stream, consume the same resident batches; aggregate rate vs one thread."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oxli_amd import KmerCountTable, _lib
lib = _lib.load()
G, L, R, k, STEPS = 5_000_000, 150, 1_000_000, 21, 40
s0 = torch.cuda.current_stream().cuda_stream
genome = torch.empty(G, dtype=torch.uint8, device="cuda")
lib.kct_synth_genome_device(genome.data_ptr(), G, 42, s0)
batches = []
for b in range(4):
    r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    lib.kct_synth_reads_device(r.data_ptr(), genome.data_ptr(), G, b * R, R, L, 1337, s0)
    batches.append(r)
torch.cuda.synchronize()

def make():
    st = torch.cuda.Stream()
    t = KmerCountTable(k, capacity=G)
    t.set_stream(st.cuda_stream)
    t.set_path("dedupe")
    for i in range(3):
        t.consume_device(batches[i % 4].data_ptr(), batches[i % 4].numel(), R * L)
    t.sync()
    return t, st

def loop(t, out, idx):
    n = 0
    for i in range(STEPS):
        b = batches[i % 4]
        n += t.consume_device(b.data_ptr(), b.numel(), R * L)
    t.sync()
    out[idx] = n

for nthreads in (1, 2):
    tabs = [make() for _ in range(nthreads)]
    out = [0] * nthreads
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=loop, args=(tabs[i][0], out, i)) for i in range(nthreads)]
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} table(s): {sum(out) / dt:.4g} k-mers/s aggregate, {dt / STEPS * 1e3:.3f} ms per round of {nthreads} step(s)")
