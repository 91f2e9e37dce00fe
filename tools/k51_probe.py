"""k > 32 at C2 size on a small genome: hashing path against the 128-bit dedupe-first variant (measurement aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oxli_amd import KmerCountTable, _lib
lib = _lib.load()
G, R, L = 2_000_000, 1_000_000, 150
g = torch.empty(G, dtype=torch.uint8, device="cuda"); r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
lib.kct_synth_genome_device(g.data_ptr(), G, 42, s); lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, R, L, 1337, s); torch.cuda.synchronize()
for k in (33, 51, 64):
    for path in ("partitioned", "dedupe"):
        t = KmerCountTable(k, capacity=G); t.set_path(path)
        for _ in range(3): t.consume_device(r.data_ptr(), r.numel(), R * L)
        t.sync(); t.clear(); t.profile(True); t.profile_reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): n = t.consume_device(r.data_ptr(), r.numel(), R * L)
        t.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        prof = t.profile_read()
        print(k, path, f"{10 * n / dt:.3g} k-mers/s  ({dt * 100:.3f} ms per step)", {kn: round(v[1] / 10, 3) for kn, v in prof.items() if v[1] > 0.05})
