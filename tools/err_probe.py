"""Kernel breakdown of one error-model input on each path (measurement aid): python tools/err_probe.py [sub_ppm] [n_ppm] [sorted] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oxli_amd import KmerCountTable, _lib
lib = _lib.load()
sub, nn, srt = (int(sys.argv[i]) if len(sys.argv) > i else 0 for i in (1, 2, 3))
G, R, L, k = 5_000_000, 1_000_000, 150, (int(sys.argv[4]) if len(sys.argv) > 4 else 21)
g = torch.empty(G, dtype=torch.uint8, device="cuda"); r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
lib.kct_synth_genome_device(g.data_ptr(), G, 42, s)
lib.kct_synth_reads_device_ex(r.data_ptr(), g.data_ptr(), G, 0, R, L, 1337, sub, nn, R if srt else 0, 7331, s)
torch.cuda.synchronize()
hint = int(G + R * L * sub / 1e6 * k)
for path in ("auto", "partitioned", "dedupe"):
    t = KmerCountTable(k, capacity=hint); t.set_path(path)
    t.consume_device(r.data_ptr(), r.numel(), R * L); t.sync()
    for rep in range(3):
        t.clear(); t.set_path(path); t.profile(True); t.profile_reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = t.consume_device(r.data_ptr(), r.numel(), R * L); t1 = time.perf_counter(); t.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
        prof = t.profile_read(); t.profile(False)
    print(path, f"slots={t.capacity} distinct={len(t)} consume {1e3*(t1-t0):.2f} ms + sync {1e3*(t2-t1):.2f} ms; kernels {sum(v[1] for v in prof.values()):.2f} ms:",
          {kn: (v[0], round(v[1], 3)) for kn, v in prof.items()})
